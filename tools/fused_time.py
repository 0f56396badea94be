"""Launch-to-launch time of the tally kernels alone (no exact tests beside them): counts, per-sample missing, fused.

    python3 tools/fused_time.py [--variants 1000000] [--samples 500000] [--reps 10]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import plinking_duck_amd.lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=1_000_000)
    ap.add_argument("--samples", type=int, default=500_000)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    m, n = args.variants, args.samples
    torch.cuda.set_device(0)
    L.set_device(0)
    ds = L.Dataset.synth(0, m, n, 20260807, 0.02)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    d_counts = torch.empty((m, 4), dtype=torch.int32, device=dev)
    d_miss = torch.empty((n + 63) // 64 * 64, dtype=torch.int32, device=dev)
    gb = m * ds.info.record_bytes / 1e9

    def time(label, fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        print(f"{label:28s} {ms:8.3f} ms  {gb / ms:6.3f} TB/s  {gb / ms / 8:5.3f} of 8 TB/s", flush=True)

    time("counts (k_counts_block)", lambda: ds.counts_range_dev(0, m, d_counts.data_ptr(), st))
    time("missing per sample", lambda: ds.missing_per_sample_dev(0, m, d_miss.data_ptr(), st))
    time("fused (k_fused_tally)", lambda: ds.fused_tally_dev(0, m, d_counts.data_ptr(), d_miss.data_ptr(), st))
    for batch in (32768, 131072):
        def batched():
            for b in range(0, m, batch):
                ds.fused_tally_dev(b, min(m, b + batch), d_counts.data_ptr() + 16 * b, d_miss.data_ptr(), st)
        time(f"fused, {batch}-variant launches", batched)


if __name__ == "__main__":
    main()
