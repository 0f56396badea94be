"""Phase times of the table functions (C++ shells), over a mid-size file on disk or -- with --synth -- over a
resident synthetic fileset of BASELINE's shape (1,000,000 x 500,000), which no disk file of this pool could hold.

    python3 tools/shell_bench.py [--variants 200000] [--samples 50000] [--threads 16]
    python3 tools/shell_bench.py --synth 1000000x500000 [--threads 16]

bind = companion files + header probe, init = residency (pgh_open: file -> HBM, or the generator; cached per
process) + the enqueue of the range's tally pass, scan = all scan threads until drained (waits on the pass's
batches + row fill).  Without --synth the rows are serialised for the Python harness inside scan, which DuckDB
would not pay; with --synth the chunks are drained inside the harness (every projected cell read once) and each
function's scan is also given as genotypes/s and as a fraction of the HBM roofline: variants x ceil(N/4) bytes / scan
time / 8 TB/s -- the whole SQL-side scan, not the kernel.
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
    ap.add_argument("--synth", default="", help="<variants>x<samples>: resident synthetic source, drained chunks")
    ap.add_argument("--pca-variants", type=int, default=100_000)
    ap.add_argument("--no-cache", action="store_true", help="plinking_tally_cache = false: every call walks the matrix")
    ap.add_argument("--list-variants", type=int, default=131_072,
                    help="--synth: variants of the separate resident source read_pgen's genotype lists are timed on")
    ap.add_argument("--only", default="", help="comma-separated substrings of the labels to run")
    ap.add_argument("--first", default="", help="run this function's call first (e.g. plink_hardy: the pass then "
                                                "carries the exact tests beside its tallies)")
    args = ap.parse_args()
    synth = bool(args.synth)
    if synth:
        m, n = (int(x) for x in args.synth.split("x"))
        path = prefix = f"synth:{m}x{n}:20260807:0.02"
    else:
        m, n = args.variants, args.samples
        prefix = os.path.join(tempfile.mkdtemp(), "mid")
        t0 = time.perf_counter()
        L.synth_write_files(prefix, m, n, 20260807, 0.02)
        size = os.path.getsize(prefix + ".pgen")
        print(f"wrote {size / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s")
        path = prefix + ".pgen"
    w = [0.001 * ((i * 7919) % 2001 - 1000) for i in range(m)]
    per_chrom = (m + 21) // 22
    settings = {"plinking_tally_cache": False} if args.no_cache else None
    calls = [
        ("plink_freq (first call: residency + the pass)", "plink_freq", m, dict(columns=["ID", "ALT_FREQ", "OBS_CT"])),
        ("plink_freq", "plink_freq", m, dict(columns=["ID", "ALT_FREQ", "OBS_CT"])),
        ("plink_hardy", "plink_hardy", m, dict(columns=["ID", "P_HWE"])),
        ("plink_missing", "plink_missing", m, dict(columns=["ID", "F_MISS"])),
        ("plink_missing sample", "plink_missing", m, dict(mode="sample", columns=["IID", "F_MISS"])),
        ("plink_freq, 1000-sample subset (own pass)", "plink_freq", m,
         dict(samples=list(range(0, n, max(1, n // 1000)))[:1000], columns=["ID", "ALT_FREQ", "OBS_CT"])),
        ("plink_score", "plink_score", m, dict(weights=w, columns=["IID", "SCORE_SUM"])),
        ("read_pgen counts", "read_pgen", m, dict(genotypes="counts", columns=["ID", "genotypes"])),
        ("read_pgen list", "read_pgen", m, dict(genotypes="list", columns=["ID", "genotypes"])),
        ("read_pfile sample counts", "read_pfile", m, dict(orient="sample", genotypes="counts", columns=["IID", "genotypes"])),
    ]
    if synth:
        # genotype lists: 4.5 bytes per genotype go to the host, so a source of their own size (twice: the second
        # call finds the pinned blocks and staging of the first)
        ml = min(args.list_variants, m)
        calls[8:9] = [(f"read_pgen list ({ml} variants; first call)", "read_pgen", ml, dict(genotypes="list", columns=["ID", "genotypes"])),
                      (f"read_pgen list ({ml} variants)", "read_pgen", ml, dict(genotypes="list", columns=["ID", "genotypes"]))]
    if args.first:
        # (a warm-up over a small source first, so that the first call's init is residency, not library start-up)
        lead = [c for c in calls if c[1] == args.first][:1]
        calls = [(f"{args.first} FIRST on the source (its own pass)",) + lead[0][1:]] + calls
    rec = (n + 3) // 4
    only = [x for x in args.only.split(",") if x]
    for label, fn, variants, kw in calls:
        if only and not any(x in label for x in only):
            continue
        target = prefix if fn == "read_pfile" else path
        if fn == "read_pgen" and kw.get("genotypes") == "list":
            if not synth and m * n > 4e9:
                continue
            if synth:
                target = f"synth:{variants}x{n}:20260807:0.02"
        t0 = time.perf_counter()
        r = F.query(fn, target, threads=args.threads, drain=synth, settings=settings, **kw)
        wall = time.perf_counter() - t0
        t = r.timing_ms
        line = (f"{label:48s} rows {len(r):8d}  bind {t['bind']:8.1f}  init {t['init']:8.1f}  scan {t['scan']:8.1f} ms"
                f"  ({r.threads} threads, round trip {wall * 1e3:8.1f} ms)")
        if synth:
            scan_s = max(t["scan"], 1e-3) * 1e-3
            both = max(t["scan"] + t["init"], 1e-3) * 1e-3
            line += (f"  scan {variants * n / scan_s:9.3e} genotypes/s = {variants * rec / scan_s / 8e12:5.3f} of HBM;"
                     f" init+scan {variants * rec / both / 8e12:5.3f}")
            if kw.get("genotypes") == "list":
                line += f"; {variants * (n + (n + 63) // 64 * 8) / scan_s / 1e9:6.1f} GB/s to the host"
        print(line, flush=True)
    if synth and (not only or any("pca" in x for x in only)):
        # plink_pca at BASELINE config 5's shape: its own resident source
        mp = min(args.pca_variants, m)
        spec = f"synth:{mp}x{n}:20260807:0.02"
        for label in ("plink_pca (first call: residency)", "plink_pca"):
            t0 = time.perf_counter()
            r = F.query("plink_pca", spec, threads=args.threads, drain=True, n_pcs=10, columns=["IID", "PC1", "PC10"])
            t = r.timing_ms
            print(f"{label:48s} rows {len(r):8d}  bind {t['bind']:8.1f}  init {t['init']:8.1f}  scan {t['scan']:8.1f} ms"
                  f"  ({r.threads} threads, round trip {(time.perf_counter() - t0) * 1e3:8.1f} ms)  {mp} x {n}, k = 10", flush=True)
    print(f"tally passes started by this process: {L.tally_passes_started()}")


if __name__ == "__main__":
    main()
