"""What this GPU delivers on a plain streaming copy (read N bytes + write N bytes), for scale next to the
roofline fractions quoted against the 8 TB/s spec peak.  python3 tools/probe_hbm_copy.py"""
import torch
dev = torch.device("cuda:0")
for mb in (128, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.rand((n,), device=dev)
    y = torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"copy {mb} MiB: {us:.1f} us -> {2 * n * 4 / us / 1e3:.0f} GB/s read+write; read-only sum:", end=" ")
    e0.record()
    for _ in range(reps):
        s = x.sum()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{us:.1f} us -> {n * 4 / us / 1e3:.0f} GB/s")
