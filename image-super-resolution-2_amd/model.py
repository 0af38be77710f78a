"""CompleteEnhancedFusionSR (eval mode) on the HIP kernels -- the drop-in for the reference's
`model(lr)` call at models/team29_FreqFusion/io.py:221 (src/models/enhanced_fusion.py:694-754).

    model = FreqFusionHIP(state_dict, device)     # reference-keyed state dict (real or synthetic)
    sr = model(lr)                                # lr [B,3,h,w] fp32 in [0,1] on the GPU -> [B,3,4h,4w]
    sr = model.graphed(lr)                        # same, replayed from a HIP graph cached per input shape

There is deliberately no CPU / PyTorch compute path: without a GPU or without libff_hip.so the
constructor raises.
"""
from __future__ import annotations

import os
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
        self.multi_stream = os.environ.get("FF_STREAMS", "1") != "0"
        self._side = None
        self._marker = None                              # plan recording: called with "fork" / "join" around the three-stream section
        self._graphs = {}                                # (B,h,w[,lane]) -> (graph, static input, static output)
        self._lanes = None                               # graphed_async: one replay stream per lane
        self.max_graphs = int(os.environ.get("FF_MAX_GRAPHS", "8"))
        with torch.cuda.device(dev):
            self.hat = HatHIP(state_dict, dev)
            self.dat = DatHIP(state_dict, dev)
            self.nafnet = NafnetHIP(state_dict, dev)
            self.fusion = FusionHIP(state_dict, dev)

    def experts(self, lr: T, taps: Optional[dict] = None, with_pre: bool = False):
        """ExpertEnsemble.forward_all (expert_loader.py:768-777).  The three experts are independent until the fusion
        stack, so each runs on its own HIP stream (fork/join on the caller's stream; the whole fan-out is captured
        into the HIP graph): latency-bound kernels of one expert overlap with bandwidth-bound kernels of another.
        with_pre: also run the LR-only part of the fusion stack (FusionHIP.pre) behind the shortest expert's stream."""
        if not self.multi_stream or taps is not None:
            ex = {"hat": self.hat.forward(lr, taps), "dat": self.dat.forward(lr, taps), "nafnet": self.nafnet.forward(lr, taps)}
            return (ex, None) if with_pre else ex
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = (torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev))
        s1, s2 = self._side
        s1.wait_stream(main)
        s2.wait_stream(main)
        if self._marker:
            self._marker("fork")
        pre = None
        with torch.cuda.stream(s1):
            dat = self.dat.forward(lr)
        with torch.cuda.stream(s2):
            naf = self.nafnet.forward(lr)
            if with_pre:
                pre = self.fusion.pre(lr)
        hat = self.hat.forward(lr)
        main.wait_stream(s1)
        main.wait_stream(s2)
        if self._marker:
            self._marker("join")
        dat.record_stream(main)
        naf.record_stream(main)
        if pre is not None:
            for t in pre.values():
                t.record_stream(main)
        ex = {"hat": hat, "dat": dat, "nafnet": naf}
        return (ex, pre) if with_pre else ex

    @torch.no_grad()
    def experts_with_features(self, lr: T):
        """ExpertEnsemble.forward_all_with_hooks (expert_loader.py:894-951): ({hat,dat,nafnet} SR outputs [1,3,4h,4w],
        {hat [1,180,h,w], dat [1,180,h,w], nafnet [1,64,h,w]} hook features bilinearly resized to the LR resolution, NCHW).
        The payload of the cached-expert files (isr2_amd/cache.py)."""
        if lr.dim() != 4 or lr.shape[0] != 1 or lr.shape[1] != 3:
            raise _lib.FFError(f"expected lr of shape [1,3,h,w], got {tuple(lr.shape)}")
        lr = lr.to(self.dev, torch.float32).contiguous()
        _, _, h, w = lr.shape
        raw = {}
        with torch.cuda.device(self.dev):
            outs = {"hat": self.hat.forward(lr, feats=raw), "dat": self.dat.forward(lr, feats=raw),
                    "nafnet": self.nafnet.forward(lr, feats=raw)}
            from . import ops
            feats = {k: ops.nhwc_to_nchw(ops.resize(v, (h, w))) for k, v in raw.items()}
        return outs, feats

    @torch.no_grad()
    def forward_with_precomputed(self, lr: T, expert_outputs, expert_features=None, taps: Optional[dict] = None) -> T:
        """CompleteEnhancedFusionSR.forward_with_precomputed (enhanced_fusion.py:756-812): the cached-mode forward.  The expert
        outputs {hat,dat,nafnet: [1,3,4h,4w]} and, optionally, their hooked features (isr2_amd/cache.py files, or
        experts_with_features()) come from the caller; only the fusion stack runs, with the collaborative block live when
        features are given.  The forward half of SURVEY 8f rank 1; no backward is built."""
        if lr.dim() != 4 or lr.shape[0] != 1 or lr.shape[1] != 3:
            raise _lib.FFError(f"expected lr of shape [1,3,h,w], got {tuple(lr.shape)}")
        lr = lr.to(self.dev, torch.float32).contiguous()
        _, _, h, w = lr.shape
        ex = {}
        for k in ("hat", "dat", "nafnet"):
            t = expert_outputs[k]
            t = t.unsqueeze(0) if t.dim() == 3 else t
            if tuple(t.shape) != (1, 3, 4 * h, 4 * w):
                raise _lib.FFError(f"expert output {k}: expected [1,3,{4 * h},{4 * w}], got {tuple(t.shape)}")
            ex[k] = t.to(self.dev, torch.float32).contiguous()
        feats = None
        if expert_features is not None:
            feats = {k: (v.unsqueeze(0) if v.dim() == 3 else v).to(self.dev, torch.float32).contiguous() for k, v in expert_features.items()}
        with torch.cuda.device(self.dev):
            return self.fusion.forward(lr, ex, taps, feats=feats)

    @torch.no_grad()
    def forward(self, lr: T, taps: Optional[dict] = None, out: Optional[T] = None) -> T:
        if lr.dim() != 4 or lr.shape[0] < 1 or lr.shape[1] != 3:
            raise _lib.FFError(f"expected lr of shape [B,3,h,w], got {tuple(lr.shape)}")
        lr = lr.to(self.dev, torch.float32).contiguous()
        if lr.shape[0] > 1:
            # Every global op of the path is per image (average pools, channel attention statistics, rFFT2, squeeze gates), so a
            # batch is B independent images: they are sequenced back to back into one output buffer (one launch stream, one
            # graph when captured through graphed()).  Bit-identical to B single forwards by construction.
            if taps is not None:
                raise _lib.FFError("taps are recorded for B == 1 only")
            if out is None:
                out = torch.empty((lr.shape[0], 3, 4 * lr.shape[2], 4 * lr.shape[3]), device=self.dev, dtype=torch.float32)
            for b in range(lr.shape[0]):
                self.forward(lr[b:b + 1], out=out[b:b + 1])
            return out
        with torch.cuda.device(self.dev):
            ex, pre = self.experts(lr, taps, with_pre=True)
            if taps is not None:
                taps.update({f"expert.{k}": v for k, v in ex.items()})
            return self.fusion.forward(lr, ex, taps, pre=pre, out=out)

    __call__ = forward

    @torch.no_grad()
    def graphed(self, lr: T) -> T:
        """forward(lr) replayed from a HIP graph cached per input shape (captured on the first call with that shape: one
        warm-up on a side stream, then the capture).  The whole launch sequence -- ~1400 kernels on three streams -- becomes
        one hipGraphLaunch; tiles of one shape (io._tiled_forward, bench.py) replay the same graph.  The returned tensor is
        the graph's static output buffer: consume it (or clone it) before the next graphed() call of the same shape."""
        graph, static_in, static_out = self._graph_for(lr, 0)
        static_in.copy_(lr, non_blocking=True)               # device-to-device memcpy into the captured input buffer
        graph.replay()
        return static_out

    def _graph_for(self, lr: T, lane: int):
        lr = lr.to(self.dev, torch.float32)
        key = tuple(lr.shape) + ((lane,) if lane else ())
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.max_graphs:          # each graph pins its own activation pool: keep a few shapes only
                torch.cuda.synchronize(self.dev)              # (a replay of the evicted graph may still be in flight on a lane)
                old = next(iter(self._graphs))
                self._graphs.pop(old)
                from . import ops as _ops
                _ops.drop_persistent(key=(id(self),) + old)   # the concat / padded-input buffers that graph's launches point at
            from . import ops
            ops.set_lane(lane)
            ops.set_capture_key((id(self),) + key)
            try:
                with torch.cuda.device(self.dev):
                    static_in = lr.clone().contiguous()
                    cur = torch.cuda.current_stream()
                    side = torch.cuda.Stream(device=self.dev)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        self.forward(static_in)
                    cur.wait_stream(side)
                    torch.cuda.synchronize(self.dev)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        static_out = self.forward(static_in)
            finally:
                ops.set_lane(0)
                ops.set_capture_key(None)
            ent = self._graphs[key] = (graph, static_in, static_out)
        return ent

    def graphed_async(self, lr: T, lane: int):
        """Two-deep tile pipeline: like graphed(), but the replay runs on the lane's own stream (lane 0 / 1, each with its own
        captured graph, static buffers and persistent scratch), so the tail of one tile -- the fusion stack's small grids after the
        three experts have joined -- overlaps the head of the next.  Returns (static output, event): wait for the event on the
        consuming stream before reading the output; work the caller has queued on the current stream (the previous consumer of
        this lane's output included) is ordered before the replay.  Tiles of one image are independent (io._tiled_forward)."""
        if lane not in (0, 1):
            raise _lib.FFError("graphed_async: lane must be 0 or 1")
        graph, static_in, static_out = self._graph_for(lr, lane + 1)  # own graph + persistent scratch per lane (0 is the synchronous path's); first use captures on the CURRENT stream
        if self._lanes is None:
            self._lanes = (torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev))
        ls = self._lanes[lane]
        ls.wait_stream(torch.cuda.current_stream())
        src = lr.to(self.dev, torch.float32)
        src.record_stream(ls)                                 # the caller may drop the tile before the lane has copied it
        with torch.cuda.stream(ls):
            static_in.copy_(src, non_blocking=True)
            graph.replay()
            ev = torch.cuda.Event()
            ev.record(ls)
        return static_out, ev
