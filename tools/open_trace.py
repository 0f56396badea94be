import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import plinking_duck_amd.lib as L
prefix = "/tmp/ot"
m, n = 32768, 500000
if not os.path.exists(prefix + ".pgen"):
    L.synth_write_files(prefix, m, n, 1, 0.02)
for k in range(4):
    t0 = time.perf_counter()
    ds = L.Dataset.open(prefix + ".pgen", variant_begin=k * 8192, variant_end=(k + 1) * 8192)
    t1 = time.perf_counter()
    ds.close()
    print(f"open of 8192 x 500000 (1.0 GB): {1e3 * (t1 - t0):.1f} ms, close {1e3 * (time.perf_counter() - t1):.1f} ms", flush=True)
