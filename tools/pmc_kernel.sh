#!/bin/bash
# SQ counter passes over one kernel of a bench.py workload (run on the GPU box through gpurun).
#   tools/pmc_kernel.sh <out-tag> <kernel-name-substring> <bench.py args...>
# Each counter group is its own rocprofv3 run (--pmc only: no tracing domains next to it).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; KERN=$2; shift 2
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_I8"; do
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
