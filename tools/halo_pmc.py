"""Runs the two CAB 3x3 convolutions a few times (profiling aid for rocprofv3 --pmc; not part of the product)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from isr2_amd import ops
dev = torch.device("cuda:0")
for Ci, Co in ((180, 60), (60, 180)):
    x = torch.randn(1, 256, 256, Ci, device=dev)
    w = torch.randn(Co, 9 * Ci, device=dev) * 0.05
    b = torch.randn(Co, device=dev)
    for _ in range(5):
        ops.conv2d(x, w, b, ksize=(3, 3), pad=(1, 1))
torch.cuda.synchronize()
