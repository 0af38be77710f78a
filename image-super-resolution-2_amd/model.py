"""CompleteEnhancedFusionSR (eval mode) on the HIP kernels -- the drop-in for the reference's
`model(lr)` call at models/team29_FreqFusion/io.py:221 (src/models/enhanced_fusion.py:694-754).

    model = FreqFusionHIP(state_dict, device)     # reference-keyed state dict (real or synthetic)
    sr = model(lr)                                # lr [1,3,h,w] fp32 in [0,1] on the GPU -> [1,3,4h,4w]

There is deliberately no CPU / PyTorch compute path: without a GPU or without libff_hip.so the
constructor raises.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import lib as _lib
from .experts import HatHIP, DatHIP, NafnetHIP
from .fusion import FusionHIP

T = torch.Tensor


class FreqFusionHIP:
    def __init__(self, state_dict: Dict[str, T], device="cuda:0"):
        dev = torch.device(device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise _lib.FFError("FreqFusionHIP needs an MI355X (torch device 'cuda'); there is no CPU fallback")
        _lib.load()
        self.dev = dev
        with torch.cuda.device(dev):
            self.hat = HatHIP(state_dict, dev)
            self.dat = DatHIP(state_dict, dev)
            self.nafnet = NafnetHIP(state_dict, dev)
            self.fusion = FusionHIP(state_dict, dev)

    def experts(self, lr: T, taps: Optional[dict] = None) -> Dict[str, T]:
        """ExpertEnsemble.forward_all sequential branch (expert_loader.py:768-777)."""
        return {"hat": self.hat.forward(lr, taps), "dat": self.dat.forward(lr, taps), "nafnet": self.nafnet.forward(lr, taps)}

    @torch.no_grad()
    def forward(self, lr: T, taps: Optional[dict] = None) -> T:
        if lr.dim() != 4 or lr.shape[0] != 1 or lr.shape[1] != 3:
            raise _lib.FFError(f"expected lr of shape [1,3,h,w], got {tuple(lr.shape)}")
        lr = lr.to(self.dev, torch.float32).contiguous()
        with torch.cuda.device(self.dev):
            ex = self.experts(lr, taps)
            if taps is not None:
                taps.update({f"expert.{k}": v for k, v in ex.items()})
            return self.fusion.forward(lr, ex, taps)

    __call__ = forward
