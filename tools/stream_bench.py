"""A file beyond the HBM budget through the table functions: writes a synthetic fileset to local disk, reads it once
resident and once with PLINKING_HBM_CACHE_GB below its size (windows of half that budget), and prints the scan times.

    python tools/stream_bench.py [--variants 100000] [--samples 500000] [--budget-gb 4] [--dir /tmp]
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_child(args, prefix, budget):
    import plinking_duck_amd.lib as L  # noqa: F401
    from plinking_duck_amd import functions as F

    m, n = args.variants, args.samples
    calls = [("plink_freq", dict(columns=["ID", "ALT_FREQ", "OBS_CT"])), ("plink_hardy", dict(columns=["ID", "P_HWE"])),
             ("read_pfile sample counts", dict(orient="sample", genotypes="counts", columns=["IID", "genotypes"])),
             ("plink_score", dict(weights=[((7 * i) % 13 - 6) / 5.0 for i in range(m)], columns=["IID", "SCORE_SUM"])),
             ("read_pgen list, first 8192 variants", dict(genotypes="list", variants={"start": 0, "stop": 8192}, columns=["ID", "genotypes"])),
             ("plink_pca n_pcs=4", dict(n_pcs=4, columns=["IID", "PC1"]))]
    gb = m * ((n + 3) // 4) / 1e9
    for label, kw in calls:
        fn = label.split()[0]
        path = prefix if fn == "read_pfile" else prefix + ".pgen"
        t0 = time.perf_counter()
        r = F.query(fn, path, threads=16, drain=True, **kw)
        wall = time.perf_counter() - t0
        t = r.timing_ms
        print(f"  {label:38s} rows {len(r):8d}  bind {t['bind']:8.1f}  init {t['init']:8.1f}  scan {t['scan']:8.1f} ms   "
              f"wall {wall:6.2f} s = {gb / wall:6.1f} GB/s of the file's {gb:.1f} GB", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=100_000)
    ap.add_argument("--samples", type=int, default=500_000)
    ap.add_argument("--budget-gb", type=float, default=4.0)
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    prefix = os.path.join(args.dir, f"stream_bench_{args.variants}x{args.samples}")
    if args.child:
        return run_child(args, prefix, args.child)
    import plinking_duck_amd.lib as L

    if not os.path.exists(prefix + ".pgen"):
        t0 = time.perf_counter()
        L.synth_write_files(prefix, args.variants, args.samples, 20260807, 0.02)
        print(f"wrote {prefix}.pgen ({os.path.getsize(prefix + '.pgen') / 1e9:.1f} GB) in {time.perf_counter() - t0:.1f} s")
    for label, budget in (("resident", None), (f"streamed, PLINKING_HBM_CACHE_GB={args.budget_gb}", str(args.budget_gb))):
        env = dict(os.environ)
        if budget:
            env["PLINKING_HBM_CACHE_GB"] = budget
        print(label, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--variants", str(args.variants), "--samples", str(args.samples),
                        "--dir", args.dir, "--child", label], env=env, check=True)
    for ext in (".pgen", ".pvar", ".psam"):
        os.remove(prefix + ext)


if __name__ == "__main__":
    main()
