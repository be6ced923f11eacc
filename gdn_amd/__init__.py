"""gdn_amd — MI355X-native (gfx950) implementation of GDN's graph-attention hot path behind
the reference's `GDN(nn.Module).forward(data, org_edge_index)` API (models/GDN.py:82-187)."""
from .model import GDN, GNNLayer, GraphLayer, OutLayer  # noqa: F401

__all__ = ["GDN", "GNNLayer", "GraphLayer", "OutLayer"]
