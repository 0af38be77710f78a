"""ops.PreparedWeights: the registry of kernel-side weight images (VERDICT r1 robustness item: no ad-hoc attributes on tensors)."""
import gc
import threading

import torch


def test_registry_builds_once_enumerates_and_drops_with_the_weight():
    from isr2_amd.ops import PreparedWeights
    reg = PreparedWeights()
    w = torch.zeros(8, 8)
    calls = []

    def build():
        calls.append(1)
        return (torch.ones(4, dtype=torch.bfloat16), 7)
    a = reg.get(w, "split", build)
    b = reg.get(w, "split", build)
    assert a is b and len(calls) == 1
    reg.get(w, "halo", lambda: torch.ones(16))
    assert sorted(k for k, _ in reg.of(w)) == ["halo", "split"]
    assert reg.peek(w, "quad") is None and reg.peek(torch.zeros(8, 8), "split") is None     # keyed by identity, not by value
    assert reg.nbytes() == 4 * 2 + 16 * 4
    assert not hasattr(w, "_ff_split") and not hasattr(w, "_ff_halo")                        # nothing is hung on the tensor
    del w
    gc.collect()
    assert reg.nbytes() == 0                                                                 # entries die with their weight
    w2 = torch.zeros(3)
    reg.put(w2, "split", 1)
    reg.put(w2, "split", 2)
    assert reg.peek(w2, "split") == 2
    reg.clear()
    assert reg.peek(w2, "split") is None


def test_registry_is_safe_under_concurrent_first_use():
    from isr2_amd.ops import PreparedWeights
    reg = PreparedWeights()
    ws = [torch.zeros(4) for _ in range(64)]
    out = [[None] * 64 for _ in range(8)]

    def worker(t):
        for i, w in enumerate(ws):
            out[t][i] = reg.get(w, "k", lambda i=i: torch.full((2,), float(i)))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i, w in enumerate(ws):
        assert float(reg.peek(w, "k")[0]) == float(i)
        assert all(float(out[t][i][0]) == float(i) for t in range(8))
    assert len(reg.of(ws[0])) == 1
