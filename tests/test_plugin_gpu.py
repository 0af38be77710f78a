"""Drop-in boundary on the GPU: the plugin function the harness calls (reference test.py:50 -> io.py:188 main), its
tiled fallback (io.py:82-121, 222-228) and size-independent properties of the path at the bench size (256x256 tile,
BASELINE configs[1]) that no CPU oracle run can cover in seconds."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip_model(dev, synth_sd):
    from isr2_amd import ops
    from isr2_amd.model import FreqFusionHIP
    ops.set_gemm_mode("bf16x3")
    return FreqFusionHIP(synth_sd, dev)


def test_plugin_main_writes_reference_png(tmp_path, dev, monkeypatch):
    """BASELINE config 1 through the plugin: PNG in -> PNG out, same basenames, sorted case-insensitive glob, bytes within
    1 LSB of what the reference's own plugin wrote for the same input and weights (tests/golden/c48_u8.npz)."""
    import models.team29_FreqFusion.io as plug
    g = np.load(os.path.join(HERE, "golden", "c48_u8.npz"))
    lr_u8 = (g["lr"][0].transpose(1, 2, 0) * 255.0).round().astype(np.uint8)
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    Image.fromarray(lr_u8).save(src / "a_case.png")
    Image.fromarray(lr_u8[:32, :40]).save(src / "B_small.PNG")               # upper-case extension, non-square
    Image.fromarray(lr_u8).save(src / "ignored.jpg")                         # jpgs are only used when there is no png
    monkeypatch.setenv("FREQFUSION_PRETRAINED", str(tmp_path / "no_such_dir"))  # -> seeded synthetic weights (1234)
    monkeypatch.setenv("FF_ALLOW_SYNTH", "1")                                    # a missing fusion checkpoint raises otherwise
    plug.main(model_dir=str(tmp_path / "missing_fusion.pth"), input_path=str(src), output_path=str(dst), device=dev)
    assert sorted(os.listdir(dst)) == ["B_small.PNG", "a_case.png"]
    out = np.array(Image.open(dst / "a_case.png").convert("RGB"))
    assert out.shape == (192, 192, 3)
    diff = np.abs(out.astype(np.int16) - g["png_u8"].astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3
    assert np.array(Image.open(dst / "B_small.PNG")).shape == (128, 160, 3)


def test_plugin_main_empty_directory(tmp_path, dev, monkeypatch):
    import models.team29_FreqFusion.io as plug
    monkeypatch.setenv("FREQFUSION_PRETRAINED", str(tmp_path / "no_such_dir"))
    monkeypatch.setenv("FF_ALLOW_SYNTH", "1")
    (tmp_path / "in").mkdir()
    with pytest.raises(FileNotFoundError):                                       # reference io.py:164: torch.load of a missing file raises
        monkeypatch.setenv("FF_ALLOW_SYNTH", "0")
        plug.main(model_dir="nope.pth", input_path=str(tmp_path / "in"), output_path=str(tmp_path / "out0"), device=dev)
    monkeypatch.setenv("FF_ALLOW_SYNTH", "1")
    plug.main(model_dir="nope.pth", input_path=str(tmp_path / "in"), output_path=str(tmp_path / "out"), device=dev)
    assert os.listdir(tmp_path / "out") == []


def test_tiled_forward_matches_reference_golden(dev):
    """ff_tile_accum / ff_tile_normalize blending against the output of the reference's own _tiled_forward
    (tests/golden/tiled_37x53.npz, generated with the stand-in model below)."""
    import models.team29_FreqFusion.io as plug
    g = np.load(os.path.join(HERE, "golden", "tiled_37x53.npz"))

    def standin(t):                      # the stand-in of tests/golden/make_golden.py (test-side helper, not product code)
        up = torch.nn.functional.interpolate(t, scale_factor=4, mode="bilinear", align_corners=False)
        return up * 0.9 + 0.05 * t.mean()

    out = plug._tiled_forward(standin, torch.from_numpy(g["lr"]).to(dev), tile_size=16, overlap=4, scale=4, device=dev)
    assert (out.cpu() - torch.from_numpy(g["out"])).abs().max().item() < 2e-6


def test_oom_falls_back_to_overlap_tiles(tmp_path, dev, hip_model, monkeypatch):
    """reference io.py:222-228: an 'out of memory' RuntimeError on the whole image switches to 128/32 tiles."""
    import models.team29_FreqFusion.io as plug
    from oracle import freqfusion_oracle as O
    rng = np.random.default_rng(5)
    img = (rng.random((140, 150, 3)) * 255).astype(np.uint8)
    (tmp_path / "in").mkdir()
    Image.fromarray(img).save(tmp_path / "in" / "big.png")
    calls = []

    class Flaky:
        def __call__(self, x):
            calls.append(tuple(x.shape[-2:]))
            if x.shape[-1] > 128 or x.shape[-2] > 128:
                raise RuntimeError("HIP out of memory. Tried to allocate 1.00 GiB")
            return hip_model(x)

    monkeypatch.setattr(plug, "_build_and_load", lambda model_dir, device, rank=0, world=1: Flaky())
    plug.main(model_dir="x.pth", input_path=str(tmp_path / "in"), output_path=str(tmp_path / "out"), device=dev)
    assert calls[0] == (140, 150) and all(c == (128, 128) for c in calls[1:]) and len(calls) == 1 + 4
    got = np.array(Image.open(tmp_path / "out" / "big.png").convert("RGB")).astype(np.int16)
    lr = torch.from_numpy(img.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    ref = O.tiled_forward(lambda t: hip_model(t.to(dev)).cpu(), lr, tile=128, overlap=32, scale=4)   # oracle blending, same tiles
    ref_u8 = (ref.squeeze(0).clamp(0, 1).permute(1, 2, 0).numpy() * 255.0).round().astype(np.int16)
    d = np.abs(got - ref_u8)
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_full_size_tile_properties(dev, hip_model, synth_sd):
    """256x256 -> 1024x1024 (the bench workload).  The CPU oracle needs minutes here, so the checks are properties:
    finite and in range, bit-exact run to run, bit-exact between the three-stream and the one-stream schedule, bit-exact
    between HIP-graph replay and eager launches, and >= 100 dB between the split-bf16 and the exact-fp32 contraction."""
    from isr2_amd import ops
    from isr2_amd.model import FreqFusionHIP
    from oracle import freqfusion_oracle as O
    rng = np.random.default_rng(2)
    f = np.fft.fftfreq(256)
    amp = 1.0 / np.maximum(np.hypot(*np.meshgrid(f, f, indexing="ij")), 1.0 / 256)
    noise = np.fft.ifft2(np.fft.fft2(rng.standard_normal((3, 256, 256))) * amp).real
    noise = (noise - noise.min()) / (noise.max() - noise.min())
    lr = torch.from_numpy(noise.astype(np.float32)).unsqueeze(0).to(dev)

    a = hip_model(lr).clone()
    assert tuple(a.shape) == (1, 3, 1024, 1024) and torch.isfinite(a).all()
    assert a.min().item() >= -0.5 and a.max().item() <= 1.5
    b = hip_model(lr).clone()
    assert torch.equal(a, b), "not deterministic"
    flag = hip_model.multi_stream
    try:
        hip_model.multi_stream = not flag
        c = hip_model(lr).clone()
    finally:
        hip_model.multi_stream = flag
    assert torch.equal(a, c), "stream schedule changes the result"

    static = lr.clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gout = hip_model(static)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(a, gout), "graph replay differs from eager"

    ops.set_gemm_mode("f32")
    try:
        exact = FreqFusionHIP(synth_sd, dev)(lr)
    finally:
        ops.set_gemm_mode("bf16x3")
    psnr = O.psnr(a.cpu(), exact.cpu())
    print("256x256: PSNR(bf16x3, f32) =", psnr)
    assert psnr >= 100.0


def test_model_from_broadcast_state_dict(dev, hip_model, synth_sd):
    """The N>1 ranks of bench.py build the model from the flat device buffer the RCCL broadcast filled
    (parallel.broadcast_state_dict): same output, bit for bit, as the model built from the host state dict."""
    from isr2_amd.model import FreqFusionHIP
    from isr2_amd.parallel import broadcast_state_dict
    from isr2_amd.weights import param_spec
    sd_dev = broadcast_state_dict(synth_sd, param_spec(), 0, 1, dev)          # world 1: pack / unpack only, no collective
    assert all(v.is_cuda for v in sd_dev.values())
    lr = torch.from_numpy(np.random.default_rng(3).random((1, 3, 40, 48), dtype=np.float32)).to(dev)
    assert torch.equal(FreqFusionHIP(sd_dev, dev)(lr), hip_model(lr))


def test_image_conversions_bit_exact(dev):
    """ff_u8hwc_to_f32nchw / ff_f32nchw_to_u8hwc against the reference's numpy formulas (io.py:64-76): bit-exact, including
    values that sit exactly on a rounding tie after the multiply by 255 and values outside [0,1]."""
    from isr2_amd import ops
    rng = np.random.default_rng(9)
    u8 = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    got = ops.u8_to_f32_image(torch.from_numpy(u8).to(dev)).cpu().numpy()
    ref = (u8.astype(np.float32) / 255.0).transpose(2, 0, 1)[None]
    assert got.dtype == np.float32 and np.array_equal(got, ref)
    f = rng.random((1, 3, 41, 29), dtype=np.float32) * 1.4 - 0.2                 # some values outside [0,1]
    ties = (np.arange(0, 255, dtype=np.float32) + 0.5) / 255.0                    # k + 0.5 after * 255 (up to rounding)
    f.reshape(-1)[:ties.size] = ties
    f.reshape(-1)[ties.size:2 * ties.size] = np.nextafter(ties, np.float32(1))
    got = ops.f32_to_u8_image(torch.from_numpy(f).to(dev)).cpu().numpy()
    ref = (np.clip(f[0], 0, 1).transpose(1, 2, 0) * 255.0).round().astype(np.uint8)
    assert got.dtype == np.uint8 and np.array_equal(got, ref)


@pytest.mark.parametrize("hw", [(200, 168), (272, 204)])
def test_stream_schedule_is_bit_exact_on_ragged_sizes(dev, hip_model, hw):
    """The three-stream schedule (experts side by side, LR-only fusion work behind NAFNet) must not change a single bit
    relative to the serial schedule, also on sizes that need reflect / zero padding (a kernel that read not-yet-visible or
    uninitialised memory showed up exactly here during development)."""
    lr = torch.from_numpy(np.random.default_rng(hw[0]).random((1, 3, hw[0], hw[1]), dtype=np.float32)).to(dev)
    flag = hip_model.multi_stream
    try:
        hip_model.multi_stream = False
        a = hip_model(lr).clone()
        hip_model.multi_stream = True
        b = hip_model(lr).clone()
        c = hip_model(lr).clone()
    finally:
        hip_model.multi_stream = flag
    assert torch.equal(a, b) and torch.equal(a, c)


def test_config3_tiled_image_geometry(dev, hip_model):
    """a 510x339 LR image cut into 256-pixel tiles with 32 overlap (x in {0,224,254}, y in {0,83}: six
    tiles), blended on the device by ff_tile_accum / ff_tile_normalize -- against the oracle's restatement of the reference
    blending (io.py:82-121) fed with the same per-tile outputs."""
    import models.team29_FreqFusion.io as plug
    from oracle import freqfusion_oracle as O
    assert plug._tile_positions(510, 256, 224) == [0, 224, 254] and plug._tile_positions(339, 256, 224) == [0, 83]
    lr = torch.from_numpy(np.random.default_rng(33).random((1, 3, 339, 510), dtype=np.float32))
    got = plug._tiled_forward(hip_model, lr.to(dev), tile_size=256, overlap=32, scale=4, device=dev).cpu()
    ref = O.tiled_forward(lambda t: hip_model(t.to(dev)).cpu(), lr, tile=256, overlap=32, scale=4)
    assert tuple(got.shape) == (1, 3, 1356, 2040)
    assert (got - ref).abs().max().item() < 2e-6


def test_threaded_io_pipeline_matches_serial_loop(tmp_path, dev, hip_model, monkeypatch):
    """PNG decode / encode on host threads with pinned uint8 staging and HIP-graph replay for repeated
    shapes must write exactly the files of the reference's serial load -> forward -> save loop (io.py:214-232)."""
    import models.team29_FreqFusion.io as plug
    rng = np.random.default_rng(21)
    (tmp_path / "in").mkdir()
    sizes = [(40, 48), (40, 48), (33, 37), (40, 48), (64, 32), (40, 48)]         # a repeated shape (graph replay) and odd ones
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(tmp_path / "in" / f"im{i}.png")
    monkeypatch.setattr(plug, "_build_and_load", lambda model_dir, device, rank=0, world=1: hip_model)
    monkeypatch.setenv("FF_IO_THREADS", "0")
    plug.main(model_dir="x.pth", input_path=str(tmp_path / "in"), output_path=str(tmp_path / "serial"), device=dev)
    monkeypatch.setenv("FF_IO_THREADS", "1")
    plug.main(model_dir="x.pth", input_path=str(tmp_path / "in"), output_path=str(tmp_path / "threads"), device=dev)
    assert sorted(os.listdir(tmp_path / "serial")) == sorted(os.listdir(tmp_path / "threads")) == [f"im{i}.png" for i in range(6)]
    for i, (h, w) in enumerate(sizes):
        a = np.array(Image.open(tmp_path / "serial" / f"im{i}.png"))
        b = np.array(Image.open(tmp_path / "threads" / f"im{i}.png"))
        assert a.shape == (4 * h, 4 * w, 3) and np.array_equal(a, b), i


def test_graphed_forward_is_bit_exact_and_survives_other_sizes(dev, hip_model):
    """model.graphed: replay == eager, bit for bit; a graph captured at one size stays valid after forwards (and captures) at
    other sizes -- the persistent concat / padded-input buffers are keyed by size."""
    rng = np.random.default_rng(8)
    a = torch.from_numpy(rng.random((1, 3, 64, 48), dtype=np.float32)).to(dev)
    b = torch.from_numpy(rng.random((1, 3, 40, 72), dtype=np.float32)).to(dev)
    ea, eb = hip_model(a).clone(), hip_model(b).clone()
    ga = hip_model.graphed(a).clone()
    gb = hip_model.graphed(b).clone()
    hip_model(torch.from_numpy(rng.random((1, 3, 56, 56), dtype=np.float32)).to(dev))      # eager at a third size in between
    ga2 = hip_model.graphed(a).clone()
    a2 = torch.from_numpy(rng.random((1, 3, 64, 48), dtype=np.float32)).to(dev)
    ga3 = hip_model.graphed(a2).clone()
    assert torch.equal(ea, ga) and torch.equal(eb, gb) and torch.equal(ea, ga2)
    assert torch.equal(hip_model(a2), ga3)


def test_persistent_buffers_stay_bounded_over_many_shapes(dev, hip_model):
    """the zero-padded concat / input buffers that outlive a forward are owned by the graph entry that captured
    them (dropped on eviction) or, for eager forwards, exist once per role (replaced on a size change) -- a directory of
    differently sized images must not accumulate one set per shape."""
    from isr2_amd import ops
    rng = np.random.default_rng(9)
    shapes = [(24 + 4 * i, 20 + 8 * i) for i in range(hip_model.max_graphs + 4)]
    owners = {id(hip_model.fusion.hier), id(hip_model.nafnet)}
    for h, w in shapes:                                                        # eager forwards: one set of buffers, whatever came before
        lr = torch.from_numpy(rng.random((1, 3, h, w), dtype=np.float32)).to(dev)
        ref = hip_model(lr).clone()
        mine = {k: tuple(b.shape) for k, b in ops._PERSIST_EAGER.items() if k[0] in owners}
        assert len(mine) == 3, mine                                            # in2, in3, NAFNet's padded input -- not 3 per shape seen
        assert sorted(mine.values())[-1] == (1, 4 * h, 4 * w, 76), mine
        assert torch.equal(hip_model(lr), ref)
    hmax, wmax = max(s[0] for s in shapes), max(s[1] for s in shapes)
    one_set = 4 * (76 * (16 * hmax * wmax + 4 * hmax * wmax) + 3 * (4 * hmax + 16) * (4 * wmax + 16))
    base = ops.persistent_bytes()
    for h, w in shapes:                                                        # graphed: at most max_graphs sets are alive
        lr = torch.from_numpy(rng.random((1, 3, h, w), dtype=np.float32)).to(dev)
        out = hip_model.graphed(lr).clone()
        assert torch.equal(out, hip_model(lr))
        assert len(hip_model._graphs) <= hip_model.max_graphs
        assert ops.persistent_bytes() <= base + (hip_model.max_graphs + 1) * one_set
    n_graph_slots = sum(1 for k in ops._PERSIST_GRAPH if k[0] == id(hip_model))
    assert n_graph_slots <= hip_model.max_graphs


def test_two_lane_tile_pipeline_is_bit_exact(dev, hip_model, monkeypatch):
    """model.graphed_async / io._tiled_forward with two tiles in flight (two captured graphs on two lane streams, lane-keyed
    persistent buffers): the same bits as one tile at a time, on an image with full, right-edge, bottom-edge and corner tiles."""
    import models.team29_FreqFusion.io as plug
    lr = torch.from_numpy(np.random.default_rng(77).random((1, 3, 150, 200), dtype=np.float32)).to(dev)
    monkeypatch.setenv("FF_TILE_LANES", "0")
    ref = plug._tiled_forward(hip_model, lr, tile_size=64, overlap=8, scale=4, device=dev).clone()
    monkeypatch.setenv("FF_TILE_LANES", "1")
    got = plug._tiled_forward(hip_model, lr, tile_size=64, overlap=8, scale=4, device=dev)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    # the primitive itself: lanes return what graphed() returns, in any interleaving, and an eager forward in between is harmless
    a = lr[:, :, :64, :64].contiguous()
    b = lr[:, :, 64:128, 100:164].contiguous()
    ra, rb = hip_model.graphed(a).clone(), hip_model.graphed(b).clone()
    o0, e0 = hip_model.graphed_async(a, 0)
    o1, e1 = hip_model.graphed_async(b, 1)
    hip_model(b)
    e0.synchronize()
    e1.synchronize()
    assert torch.equal(o0, ra) and torch.equal(o1, rb)
    o0, e0 = hip_model.graphed_async(b, 0)
    e0.synchronize()
    assert torch.equal(o0, rb)


def test_config4_image_through_the_plugin(tmp_path, dev, hip_model, monkeypatch, synth_sd):
    """BASELINE config 4's unit of work at world = 1: one 2040x1356 LR image through main(), both branches of the reference's
    policy (io.py:219-228).  (a) Whole image: fits in 288 GB -- output geometry, range, and agreement of the split-bf16 path
    with the exact-fp32 contraction on a 512x512 HR crop (the CPU oracle cannot run 2.8 M tokens; whole-image parity at this
    size is pinned only through these properties).  (b) 'out of memory' on the whole image: 128-px tiles, overlap 32, blended
    on the device -- 17 x 12 = 204 replays of one captured graph; a 64x64 HR crop from the interior of a tile (no blending
    there) is compared with the CPU oracle run on that 128x128 LR tile."""
    import time
    import models.team29_FreqFusion.io as plug
    from isr2_amd import ops
    from isr2_amd.model import FreqFusionHIP
    from oracle import freqfusion_oracle as O
    import bench
    lr = (bench.make_tile(44, 2048)[0, :, :1356, :2040].permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8)   # 1/f noise
    (tmp_path / "in").mkdir()
    Image.fromarray(lr).save(tmp_path / "in" / "big_2k.png")
    # (a) whole image
    monkeypatch.setattr(plug, "_build_and_load", lambda model_dir, device, rank=0, world=1: hip_model)
    t0 = time.perf_counter()
    plug.main(model_dir="x.pth", input_path=str(tmp_path / "in"), output_path=str(tmp_path / "whole"), device=dev)
    print(f"config 4 image, whole: {time.perf_counter() - t0:.2f} s including PNG decode / encode")
    whole = np.array(Image.open(tmp_path / "whole" / "big_2k.png").convert("RGB"))
    assert whole.shape == (4 * 1356, 4 * 2040, 3)
    lr_t = torch.from_numpy(lr.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0).to(dev)
    a = hip_model(lr_t)[0, :, 2000:2512, 3000:3512].clone()
    assert torch.isfinite(a).all() and a.min().item() >= 0.0 and a.max().item() <= 1.0
    ops.set_gemm_mode("f32")
    try:
        exact = FreqFusionHIP(synth_sd, dev)(lr_t)[0, :, 2000:2512, 3000:3512].clone()
    finally:
        ops.set_gemm_mode("bf16x3")
    psnr = O.psnr(a.cpu(), exact.cpu())
    print("config 4 image, whole: PSNR(bf16x3, f32) on a 512x512 crop =", psnr)
    assert psnr >= 100.0
    del exact
    torch.cuda.empty_cache()

    # (b) the tile path
    class Limited:                                   # a 16 GB card's behaviour: the whole image does not fit
        def __call__(self, x):
            if x.shape[-1] > 128 or x.shape[-2] > 128:
                raise RuntimeError("HIP out of memory. Tried to allocate 11.30 GiB")
            return hip_model(x)

        def graphed(self, x):
            return hip_model.graphed(x)

    monkeypatch.setattr(plug, "_build_and_load", lambda model_dir, device, rank=0, world=1: Limited())
    t0 = time.perf_counter()
    plug.main(model_dir="x.pth", input_path=str(tmp_path / "in"), output_path=str(tmp_path / "tiled"), device=dev)
    print(f"config 4 image, 204 tiles of 128: {time.perf_counter() - t0:.2f} s including PNG decode / encode")
    out = np.array(Image.open(tmp_path / "tiled" / "big_2k.png").convert("RGB"))
    assert out.shape == (4 * 1356, 4 * 2040, 3)
    # tile at (y, x) = (96, 192) covers LR rows 96..224, cols 192..320; its blend ramps are 128 HR px wide, so HR rows
    # 4*96+128 .. 4*224-128 = 512..768 (cols 896..1152) belong to this tile alone
    ty, tx = 96, 192
    assert ty in plug._tile_positions(1356, 128, 96) and tx in plug._tile_positions(2040, 128, 96)
    tile = torch.from_numpy(lr[ty:ty + 128, tx:tx + 128].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    ref = O.forward(synth_sd, tile)
    ref_u8 = (ref[0].clamp(0, 1).permute(1, 2, 0).numpy() * 255.0).round().astype(np.int16)
    cy, cx = 4 * ty + 200, 4 * tx + 220
    d = np.abs(out[cy:cy + 64, cx:cx + 64].astype(np.int16) - ref_u8[200:264, 220:284])
    assert d.max() <= 1 and (d > 0).mean() < 5e-3, (d.max(), (d > 0).mean())
