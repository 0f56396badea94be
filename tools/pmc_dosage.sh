#!/bin/bash
# PMC passes over the dosage score's explicit-entry kernel (run on the GPU box through gpurun).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dos_pmc
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/dos_pmc/$tag -- python3 bench.py --workload dosagescore --variants 100000 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> gpurun_out/dos_pmc_err.txt || { tail -5 gpurun_out/dos_pmc_err.txt; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/dos_pmc/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_score_dosage_' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value'])
    print({k: f"{v:.4g}" for k, v in acc.items()})
PY
