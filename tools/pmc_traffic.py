"""Builds profiles/*_pmc_hbm_traffic_*.json from two rocprofv3 counter passes (profiling aid, not part of the product).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01_pmc_hbm_traffic_v10.json

Counter unit = KiB.  HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half of a wide coalesced read
stream (MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is exact.  The two counters are collected in separate passes."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0][:140]
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def product_kernel_names():
    """Names of the __global__ kernels this library defines (csrc/): everything else in a trace -- ATen element-wise / copy kernels of
    the weight preparation at model build, the runtime's buffer copies, rocBLAS -- is SETUP, not part of a forward."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "image-super-resolution-2_amd", "csrc")
    names = set()
    for f in os.listdir(d):
        txt = open(os.path.join(d, f)).read()
        names.update(re.findall(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(", txt))
    return names


SETUP_ONLY = {"split_bf16_kernel"}          # ours, but launched by the weight preparation at load (static weights), not per forward


def is_product(kernel_name: str, names) -> bool:
    base = kernel_name.replace("void ", "").split("<")[0].split("(")[0].strip()
    return base in names and base not in SETUP_ONLY


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        cf, vf = fe.get(k, [0, 0.0])
        cw, vw = wr.get(k, [0, 0.0])
        calls = max(cf, cw)
        if calls == 0:
            continue
        fk, wk = vf / max(cf, 1), vw / max(cw, 1)
        kernels[k] = {"calls": calls, "fetch_KiB_per_call_raw": fk, "write_KiB_per_call": wk,
                      "hbm_MB_per_call_corrected": (2 * fk + wk) * 1024 / 1e6}

    def family(pred):
        sel = [v for k, v in kernels.items() if pred(k)]
        calls = sum(v["calls"] for v in sel)
        mb = sum(v["calls"] * v["hbm_MB_per_call_corrected"] for v in sel)
        return {"launches": calls, "hbm_MB_per_launch": mb / max(calls, 1), "hbm_MB_total": mb}

    import hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "image-super-resolution-2_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    forwards = sum(v["calls"] for k, v in kernels.items() if "fuse_blend" in k)      # launched exactly once per forward
    names = product_kernel_names()
    for k, v in kernels.items():
        v["product"] = is_product(k, names)
    doc = {"note": __doc__.split("\n\n")[-1].replace("\n", " "), "csrc_fingerprint": h.hexdigest()[:16], "forwards": forwards,
           "git": os.environ.get("FF_GIT_HASH", "n/a"), "kernels": kernels,
           "conv_igemm_all_variants": family(lambda k: "conv_igemm" in k or "conv3x3_halo" in k),
           "token_linear_all_variants": family(lambda k: "token_linear" in k),
           "window_attn_all_variants": family(lambda k: "window_attn" in k),
           "win_attn_fused_all_variants": family(lambda k: "win_attn_fused" in k),
           "token_mlp": family(lambda k: "token_mlp" in k),
           "token_projmlp": family(lambda k: "token_projmlp" in k),
           "whole_forward": family(lambda k: kernels[k]["product"]),
           "setup_and_runtime": family(lambda k: not kernels[k]["product"]),
           "whole_forward_note": "whole_forward sums ONLY this library's kernels (csrc/ __global__ names); setup_and_runtime holds the ATen "
                                 "element-wise / copy kernels of the weight preparation at model build, runtime buffer copies and ff_split_bf16 "
                                 "(static weight planes), which run once per model, not per forward"}
    json.dump(doc, open(out, "w"), indent=1)
    for k in ("conv_igemm_all_variants", "token_linear_all_variants", "window_attn_all_variants", "win_attn_fused_all_variants", "token_mlp",
              "token_projmlp", "whole_forward", "setup_and_runtime"):
        print(k, doc[k])


if __name__ == "__main__":
    main()
