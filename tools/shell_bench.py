"""Phase times of the table functions (C++ shells) over a mid-size file on disk.

    python3 tools/shell_bench.py [--variants 200000] [--samples 50000] [--threads 16]

bind = companion files + header probe, init = residency (pgh_open: file -> HBM, cached per process),
scan = all scan threads until drained (tally launches + row fill; rows are serialised for the
Python harness inside scan, which DuckDB would not pay).
"""
import argparse
import os

os.environ.setdefault("PLINKING_PVAR_CACHE", "0")  # "first bind" below is the text parse, not the on-disk side-cache
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plinking_duck_amd.lib as L  # noqa: E402
from plinking_duck_amd import functions as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=200_000)
    ap.add_argument("--samples", type=int, default=50_000)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    prefix = os.path.join(tempfile.mkdtemp(), "mid")
    t0 = time.perf_counter()
    L.synth_write_files(prefix, args.variants, args.samples, 20260807, 0.02)
    size = os.path.getsize(prefix + ".pgen")
    print(f"wrote {size / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s")
    path = prefix + ".pgen"
    w = [0.001 * ((i * 7919) % 2001 - 1000) for i in range(args.variants)]
    calls = [
        ("plink_freq (cold: includes the ingest)", "plink_freq", dict(columns=["ID", "ALT_FREQ", "OBS_CT"])),
        ("plink_freq", "plink_freq", dict(columns=["ID", "ALT_FREQ", "OBS_CT"])),
        ("plink_hardy", "plink_hardy", dict(columns=["ID", "P_HWE"])),
        ("plink_missing", "plink_missing", dict(columns=["ID", "F_MISS"])),
        ("plink_missing sample", "plink_missing", dict(mode="sample", columns=["IID", "F_MISS"])),
        ("plink_score", "plink_score", dict(weights=w, columns=["IID", "SCORE_SUM"])),
        ("read_pgen counts", "read_pgen", dict(genotypes="counts", columns=["ID", "genotypes"])),
        ("read_pfile sample counts", "read_pfile", dict(orient="sample", genotypes="counts", columns=["IID", "genotypes"])),
    ]
    for label, fn, kw in calls:
        target = prefix if fn == "read_pfile" else path
        t0 = time.perf_counter()
        r = F.query(fn, target, threads=args.threads, **kw)
        wall = time.perf_counter() - t0
        t = r.timing_ms
        print(f"{label:42s} rows {len(r):7d}  bind {t['bind']:8.1f}  init {t['init']:8.1f}  scan {t['scan']:8.1f} ms"
              f"  (python round trip {wall * 1e3:8.1f} ms, {r.threads} threads)")


if __name__ == "__main__":
    main()
