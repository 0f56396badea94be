#!/usr/bin/env python3
"""Times pgh_pca (BASELINE config 5 shape, scaled by flags) on a synthetic matrix."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plinking_duck_amd.lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", type=int, default=20000)
ap.add_argument("--samples", type=int, default=100000)
ap.add_argument("--n-pcs", type=int, default=10)
args = ap.parse_args()

ds = L.Dataset.synth(0, args.variants, args.samples, 20260807, 0.02)
t0 = time.perf_counter()
counts = ds.counts_range().astype(np.float64)
obs = counts[:, :3].sum(axis=1)
af = (counts[:, 1] + 2 * counts[:, 2]) / (2 * obs)
keep = (obs > 0) & (af > 0) & (af < 1)
vidx = np.flatnonzero(keep).astype(np.uint32)
center = 2 * af[keep]
inv = 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep]))
t1 = time.perf_counter()
g1 = np.random.default_rng(1).standard_normal((args.samples, 2 * args.n_pcs))
ev, vecs = ds.pca(vidx, center, inv, args.n_pcs, g1)
t2 = time.perf_counter()
print(f"M_eff={len(vidx)} N={args.samples} k={args.n_pcs}: prepass {t1 - t0:.3f}s, pgh_pca {t2 - t1:.3f}s")
print("eigenvalues", np.round(ev, 6))
