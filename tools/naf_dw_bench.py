"""Timing of ff_dwconv3_gate_pool at the NAFNet level shapes (tuning aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from isr2_amd import ops
dev = torch.device("cuda:0")
for (H, C) in ((1024, 64), (512, 128), (256, 256), (128, 512), (64, 1024)):
    t = torch.randn(1, H, H, 2 * C, device=dev)
    w = torch.randn(9, 2 * C, device=dev)
    b = torch.randn(2 * C, device=dev)
    for _ in range(3): ops.dwconv3_gate_pool(t, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.dwconv3_gate_pool(t, w, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"H {H} C {C}: {us:8.1f} us  {12.0 * H * H * C / us / 1e6:6.2f} TB/s")
