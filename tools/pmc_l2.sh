#!/bin/bash
# L2 / fabric counter passes over one kernel of a bench.py workload (run on the GPU box through gpurun).
#   tools/pmc_l2.sh <out-tag> <kernel-name-substring> <bench.py args...>
# Each counter group is its own rocprofv3 run (--pmc only: no tracing domains next to it).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; KERN=$2; shift 2
OUT=gpurun_out/pmcl2_$TAG
rm -rf $OUT; mkdir -p $OUT
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $c --output-format csv -d $OUT/$tag -- python3 bench.py "$@" --cpu-seconds 0 > /dev/null 2> $OUT/err_$tag.txt || { echo "pass failed: $c"; tail -3 $OUT/err_$tag.txt; }
done
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, collections, sys
out, kern = sys.argv[1], sys.argv[2]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + '/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k, v in acc.items():
        tot[k] = (v, n[k])
with open(out + '/summary.txt', 'w') as fh:
    for k, (v, n) in tot.items():
        line = f"{k:36s} {v:16.6g}  over {n} dispatches  ({v / max(n, 1):.6g} per dispatch)"
        print(line); fh.write(line + "\n")
PY
