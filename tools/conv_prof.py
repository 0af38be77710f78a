"""Single-shape driver for rocprofv3 counter passes on the LDS-resident 3x3 kernel.  python3 tools/conv_prof.py Cin Cout [HW]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isr2_amd import ops  # noqa: E402

cin, cout = int(sys.argv[1]), int(sys.argv[2])
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
x = ops.empty_rows((1, hw, hw, cin), dev)
x.copy_(torch.randn(1, hw, hw, cin, device=dev))
w = torch.randn(cout, 9 * cin, device=dev) * 0.05
b = torch.randn(cout, device=dev)
for _ in range(5):
    ops.conv2d(x, w, b, ksize=(3, 3), pad=(1, 1), act="gelu")
torch.cuda.synchronize()
