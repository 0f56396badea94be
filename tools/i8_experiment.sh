#!/bin/bash
# Where k_score_i8's time goes: rebuild score_i8.hip with one part removed at a time and time the 16-column launch.
#   bash tools/i8_experiment.sh     (run on the GPU box through gpurun)
# The knock-out builds give WRONG results by design: they are linked to /tmp and handed to the bench through
# PGENHIP_LIB -- the in-tree libpgenhip.so is never touched, whatever interrupts this script.
cd "$GRAFT_REPO_ROOT/plinking_duck_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
# every object of the regular build except the one under test (build.sh's list, taken from the build directory)
OBJS=$(ls build/*.o | grep -v -e 'build/score_i8.o' -e 'build/shell_')
X=/tmp/libpgenhip_x.so
for v in ${VARIANTS:-NONE NO_BUILD NO_DMA NO_DMA_B NO_DMA_G HOT_B}; do
  /opt/rocm/bin/hipcc $FLAGS -DPGH_I8_$v -c score_i8.hip -o /tmp/score_i8_x.o 2>/dev/null || { echo "compile failed: $v"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $X $OBJS /tmp/score_i8_x.o || { echo "link failed: $v"; continue; }
  cd "$GRAFT_REPO_ROOT"
  PGENHIP_LIB=$X rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/i8x -- python3 bench.py ${BENCH_ARGS:---workload score --score-cols ${COLS:-16}} --steps ${STEPS:-3} --warmup 1 --cpu-seconds 0 --configs none --no-sql > /dev/null 2> /tmp/i8x_err.txt
  f=$(ls -t /tmp/i8x/*/*_kernel_stats.csv 2>/dev/null | head -1)
  printf "%-28s " "$v"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_score_i8' in r['Name']:
        shape = r['Name'].split('k_score_i8')[1].split('(')[0]
        print(f"k_score_i8{shape} avg {float(r['AverageNs'])/1e6:8.2f} ms over {r['Calls']} calls", end='; ')
print()
PY
  rm -rf /tmp/i8x
  cd "$GRAFT_REPO_ROOT/plinking_duck_amd/csrc"
done
rm -f $X
