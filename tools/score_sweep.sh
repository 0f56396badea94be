#!/bin/bash
# plink_score by weight-column count, then plink_pca (run on the GPU box through gpurun): tools/score_sweep.sh <tag>
cd "$GRAFT_REPO_ROOT"
TAG=${1:-sweep}
mkdir -p gpurun_out/$TAG
for c in 1 2 4 8 16; do
  python bench.py --workload score --score-cols $c --cpu-seconds 0 > gpurun_out/$TAG/score$c.json 2> gpurun_out/$TAG/score$c.err || { tail -3 gpurun_out/$TAG/score$c.err; exit 1; }
done
python bench.py --workload pca --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/$TAG/pca.json 2> gpurun_out/$TAG/pca.err || { tail -3 gpurun_out/$TAG/pca.err; exit 1; }
python - "$TAG" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}/*.json"), key=lambda p: (len(p), p)):
    d = json.load(open(f)); r = d["roofline"]
    print(f"{d['metric']:70s} {d['ms_per_step']:8.2f} ms  verified={d['verified']}  {r['bound']} frac {r['frac']:.3f}")
PY
