"""Diagnostic: where the time of one dense fused launch goes (workgroup 0, wave 0; 100 MHz real-time clock).
Needs a library built with GDN_HIPCC_EXTRA=-DGDN_STAMPS:  GDN_HIP_LIB=/path/libgdn_stamps.so python3 tools/probe_stamps.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd import _lib  # noqa: E402
from test_gpu_forward_parity import random_params  # noqa: E402

dev = torch.device("cuda:0")
model = random_params(127, 15, 30, 64, seed=0).to(dev).eval()
names = ["start", "consts loaded", "first x issued+barrier", "x stored", "B1 passed", "P done", "B2 passed",
         "S done", "M done", "E done + store", "kernel end"]
for b in (512, 4096, 32768):
    x = torch.rand((b, 127, 15), device=dev)
    with torch.no_grad():
        for _ in range(50):
            model(x, None)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib = _lib.load()
    lib.gdn_debug_read_stamps.argtypes = [ctypes.c_void_p]
    assert lib.gdn_debug_read_stamps(buf) == 0
    t = [buf[i] for i in range(11)]
    print(f"batch {b}: (10 ns ticks since kernel start)")
    for i in range(1, 11):
        print(f"   {names[i]:28s} +{(t[i] - t[i - 1]) * 10:7d} ns   at {(t[i] - t[0]) * 10:7d} ns")
