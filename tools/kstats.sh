#!/bin/bash
# rocprofv3 --kernel-trace --stats over one bench.py workload; prints the kernels by total time.
#   tools/kstats.sh <tag> <bench.py args...>      (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/kstats_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py "$@" --cpu-seconds 0 > $OUT/bench.json 2> $OUT/err.txt
f=$(ls $OUT/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] || { echo "no stats file"; tail -5 $OUT/err.txt; exit 1; }
cp "$f" $OUT/kernel_stats.csv
python3 tools/kstats_print.py $OUT/kernel_stats.csv
