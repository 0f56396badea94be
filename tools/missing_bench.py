#!/usr/bin/env python3
"""Times pgh_missing_per_sample_dev (plink_missing mode := 'sample') over a synthetic resident matrix."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import plinking_duck_amd.lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", type=int, default=1_000_000)
ap.add_argument("--samples", type=int, default=500_000)
args = ap.parse_args()
ds = L.Dataset.synth(0, args.variants, args.samples, 20260807, 0.02)
out = torch.empty((args.samples + 63) // 64 * 64, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream()
for i in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    ds.missing_per_sample_dev(0, args.variants, out.data_ptr(), st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    gb = args.variants * ds.info.record_bytes / 1e9
    print(f"missing per sample #{i}: {ms:.2f} ms = {gb / ms:.2f} TB/s = {gb / ms / 8:.3f} of 8 TB/s; sum {int(out[:args.samples].sum())}")
