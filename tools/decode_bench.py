"""pgh_open on a compressed .pgen: device record decode vs the host normaliser.

    python3 tools/decode_bench.py [--variants 8192] [--samples 200000]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pgen_writer as W  # noqa: E402
import plinking_duck_amd.lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=8192)
    ap.add_argument("--samples", type=int, default=200_000)
    args = ap.parse_args()
    m, n = args.variants, args.samples
    rng = np.random.default_rng(1)
    t0 = time.perf_counter()
    geno = np.zeros((m, n), dtype=np.uint8)
    kinds = []
    for v in range(m):
        rate = float(rng.choice([0.0005, 0.005, 0.02, 0.05]))
        hit = np.flatnonzero(rng.random(n) < rate)
        if v and rng.random() < 0.3:
            geno[v] = geno[v - 1]
            kinds.append(2)
        else:
            kinds.append(int(rng.choice([1, 4, 4, 4])))
        geno[v, hit] = rng.integers(1, 4, hit.size, dtype=np.uint8)
    path = os.path.join(tempfile.mkdtemp(), "rare.pgen")
    W.write_pgen(path, geno, kinds)
    size = os.path.getsize(path)
    rows_bytes = m * ((n + 3) // 4)
    print(f"wrote {size / 1e6:.1f} MB of records for {rows_bytes / 1e6:.1f} MB of rows in {time.perf_counter() - t0:.1f} s")
    want = None
    for label, env in (("device decode", "0"), ("host normaliser", "1")):
        os.environ["PGH_HOST_NORMALIZE"] = env
        for rep in range(3):
            t0 = time.perf_counter()
            ds = L.Dataset.open(path)
            dt = time.perf_counter() - t0
            if rep == 2:
                rows = ds.copy_rows_to_host(0, m)
                if want is None:
                    want = rows
                else:
                    assert np.array_equal(rows, want), "device and host rows differ"
            ds.close()
            print(f"{label} #{rep}: {dt * 1e3:.1f} ms = {rows_bytes / dt / 1e9:.2f} GB/s of rows, "
                  f"{size / dt / 1e9:.2f} GB/s of file")


if __name__ == "__main__":
    main()
