import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

MODEL_CASES = ["msl_demo_w5_k5", "fc64_w15_k64", "swat127_w15_k30", "mlp2_n20_w8_k6",
               "mlp3_n12_w4_k3", "dupemb_n10_k3", "wadi_stress_small_n40_w30_k16_d128"]
SCORE_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "score_T*.npz")))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Golden fixture -> (dict of numpy arrays, params as torch tensors keyed like state_dict)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    data = {k: z[k] for k in z.files}
    params = {k[2:]: torch.from_numpy(v.copy()) for k, v in data.items() if k.startswith("p/")}
    return data, params


def meta(data):
    b, n, w, k, d, layers, inter = (int(v) for v in data["meta_bnwkd"])
    return dict(b=b, n=n, w=w, k=k, d=d, out_layer_num=layers, inter=inter)


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _collect_graph_garbage(request):
    """GPU tests leave captured HIP graphs behind in reference cycles; a cyclic collection that happens to fire
    inside a LATER test's stream capture would destroy them there, which HIP forbids (the process aborts).  Collect
    at the end of every GPU test instead.  (The product's own captures are protected by harness.capture.)"""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        gc.collect()
