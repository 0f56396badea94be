"""plink_freq, then plink_hardy on the same resident source, each bracketed by a marker kernel launch so that a kernel
trace (rocprofv3 --kernel-trace) shows which kernels the second call launched: the exact tests over the pass's resident
counts and nothing that reads the matrix.  Prints the phase times of both calls."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import plinking_duck_amd.lib as L  # noqa: E402
from plinking_duck_amd import functions as F  # noqa: E402

spec = sys.argv[1] if len(sys.argv) > 1 else "synth:1000000x500000:20260807:0.02"
marker = torch.zeros(1, device="cuda")
for fn, cols in (("plink_freq", ["ID", "ALT_FREQ", "OBS_CT"]), ("plink_hardy", ["ID", "P_HWE"]), ("plink_missing", ["ID", "F_MISS"]),
                 ("plink_missing", ["IID", "F_MISS"])):
    marker.add_(1.0)  # an elementwise kernel of torch's in the trace: the boundary between the calls
    torch.cuda.synchronize()
    kw = {"mode": "sample"} if cols[0] == "IID" else {}
    r = F.query(fn, spec, threads=16, drain=True, columns=cols, **kw)
    t = r.timing_ms
    print(f"{fn:14s} {kw.get('mode', 'variant'):8s} rows {len(r):8d}  bind {t['bind']:7.1f}  init {t['init']:7.1f}  scan {t['scan']:7.1f} ms   "
          f"tally passes so far: {L.tally_passes_started()}", flush=True)
marker.add_(1.0)
torch.cuda.synchronize()
