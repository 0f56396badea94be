#!/usr/bin/env python3
"""Times pgh_open (host parse + normalise + H2D) on a synthetic .pgen written to local disk."""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plinking_duck_amd.lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", type=int, default=64000, help="64,000 x 500,000 = 8 GB: large enough that the fixed costs of an open (allocations, header parse) do not set the rate")
ap.add_argument("--samples", type=int, default=500000)
ap.add_argument("--dosage-rate", type=float, default=0.0, help="> 0: every record carries a 0x60 dosage track")
args = ap.parse_args()
d = tempfile.mkdtemp(prefix="pgh_open_")
prefix = os.path.join(d, "syn")
t0 = time.perf_counter()
if args.dosage_rate > 0:
    L.synth_write_dosage_files(prefix, args.variants, args.samples, 1, 0.02, args.dosage_rate)
else:
    L.synth_write_files(prefix, args.variants, args.samples, 1, 0.02)
t1 = time.perf_counter()
size = os.path.getsize(prefix + ".pgen")
print(f"wrote {size / 1e9:.2f} GB in {t1 - t0:.1f} s")
for i in range(3):
    t0 = time.perf_counter()
    ds = L.Dataset.open(prefix + ".pgen")
    t1 = time.perf_counter()
    print(f"pgh_open #{i}: {t1 - t0:.3f} s = {size / (t1 - t0) / 1e9:.2f} GB/s (page cache warm)")
    c = ds.counts_range(0, 8)
    if args.dosage_rate > 0:
        print(f"  dosage variants {ds.info.dosage_variant_ct}, values {ds.info.dosage_value_ct}, sums[0] {ds.dosage_sums(0, 1)[0]}")
    ds.close()
if args.dosage_rate > 0 and args.variants <= 2000:
    os.environ["PGH_HOST_NORMALIZE"] = "1"
    t0 = time.perf_counter()
    ds = L.Dataset.open(prefix + ".pgen")
    t1 = time.perf_counter()
    print(f"pgh_open, host parse (PGH_HOST_NORMALIZE=1): {t1 - t0:.3f} s = {size / (t1 - t0) / 1e9:.2f} GB/s; sums[0] {ds.dosage_sums(0, 1)[0]}")
    ds.close()
os.remove(prefix + ".pgen"); os.remove(prefix + ".pvar"); os.remove(prefix + ".psam"); os.rmdir(d)
