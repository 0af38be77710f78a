"""rocprofv3 --kernel-trace --stats CSV -> a per-forward table of THIS library's kernels, with setup separated.

    python tools/kernel_stats_per_forward.py <dir with *kernel_stats.csv> <out.csv> [<forwards>]

`forwards` defaults to the number of fuse_blend launches (exactly one per forward).  Rows: kernel, launches per forward, ms per
forward, average us, share of the forward; a last block lists what ran outside the forwards (ATen weight preparation at model build,
runtime copies, ff_split_bf16) so nobody mistakes it for per-tile work."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import product_kernel_names, is_product  # noqa: E402


def main():
    d, out = sys.argv[1:3]
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    names = product_kernel_names()
    fwd = int(sys.argv[3]) if len(sys.argv) > 3 else sum(int(r["Calls"]) for r in rows if "fuse_blend" in r["Name"])
    fwd = max(fwd, 1)
    prod = [r for r in rows if is_product(r["Name"], names)]
    rest = [r for r in rows if not is_product(r["Name"], names)]
    tot = sum(float(r["TotalDurationNs"]) for r in prod)
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["section", "kernel", "launches_per_forward", "ms_per_forward", "avg_us", "share_of_forward_pct"])
        for r in sorted(prod, key=lambda r: -float(r["TotalDurationNs"])):
            w.writerow(["forward", r["Name"][:160], f"{int(r['Calls']) / fwd:.2f}", f"{float(r['TotalDurationNs']) / 1e6 / fwd:.4f}",
                        f"{float(r['AverageNs']) / 1e3:.2f}", f"{100 * float(r['TotalDurationNs']) / tot:.2f}"])
        w.writerow(["forward", "TOTAL (this library's kernels)", f"{sum(int(r['Calls']) for r in prod) / fwd:.1f}", f"{tot / 1e6 / fwd:.3f}", "", "100.00"])
        for r in sorted(rest, key=lambda r: -float(r["TotalDurationNs"])):
            w.writerow(["setup (once per process, NOT per forward)", r["Name"][:160], f"calls={r['Calls']}", f"total_ms={float(r['TotalDurationNs']) / 1e6:.3f}",
                        f"{float(r['AverageNs']) / 1e3:.2f}", ""])
    print(f"{fwd} forwards; {tot / 1e6 / fwd:.2f} ms of library kernels per forward; setup {sum(float(r['TotalDurationNs']) for r in rest) / 1e6:.1f} ms in total -> {out}")


if __name__ == "__main__":
    main()
