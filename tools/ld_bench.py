"""pgh_ld_pairs throughput: every anchor against its next W variants.

    python3 tools/ld_bench.py [--variants 20000] [--samples 500000] [--window 64]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plinking_duck_amd.lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=500_000)
    ap.add_argument("--window", type=int, default=64)
    args = ap.parse_args()
    m, n, w = args.variants, args.samples, args.window
    ds = L.Dataset.synth(0, m, n, 20260807, 0.02)
    a = np.repeat(np.arange(m - w, dtype=np.uint32), w)
    b = a + np.tile(np.arange(1, w + 1, dtype=np.uint32), m - w)
    ds.ld_pairs(a[:4096], b[:4096])
    for rep in range(3):
        t0 = time.perf_counter()
        sums = ds.ld_pairs(a, b)
        dt = time.perf_counter() - t0
        rb = ds.info.record_bytes
        print(f"{len(a)} pairs x {n} samples: {dt * 1e3:.1f} ms = {len(a) / dt / 1e6:.2f} M pairs/s, "
              f"{len(a) * n / dt / 1e12:.2f} T sample-pairs/s, rows requested {len(a) * 1.25 * rb / dt / 1e12:.2f} TB/s")
    assert int(sums[:, 0].max()) <= n


if __name__ == "__main__":
    main()
