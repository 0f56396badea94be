#!/bin/bash
# k_score_i8's software-pipelined loop: builds one libpgenhip per knob combination (HERE, hipcc cross-compiles) and
# times the 16-column launch with each (ON THE GPU BOX, through gpurun):
#   bash tools/i8_variants.sh build "SKEW=0 FILL=1 HOLD=2" "SKEW=1 FILL=2 HOLD=1" ...   -> tools/build/i8v/<name>.so
#   gpurun -- bash tools/i8_variants.sh run [bench args]                                   -> one line per variant
# The in-tree libpgenhip.so is never touched; the bench loads a variant through PGENHIP_LIB and verifies its results.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/build/i8v
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p "$OUT"; rm -f "$OUT"/*.so
  cd "$ROOT/plinking_duck_amd/csrc"
  OBJS=$(ls build/*.o | grep -v -e 'build/score_i8.o' -e 'build/shell_')
  pids=()
  for v in "$@"; do
    name=$(echo "$v" | tr ' =' '_-')
    defs=""; for kv in $v; do defs="$defs -DPGH_I8_$kv"; done
    ( /opt/rocm/bin/hipcc $FLAGS $defs -c score_i8.hip -o "$OUT/$name.o" 2> "$OUT/$name.log" &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/$name.so" $OBJS "$OUT/$name.o" && rm -f "$OUT/$name.o" ) &
    pids+=($!)
    if [ ${#pids[@]} -ge 4 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  done
  wait
  ls -la "$OUT"
else
  cd "$ROOT"
  for so in "$OUT"/*.so; do
    name=$(basename "$so" .so)
    line=$(PGENHIP_LIB=$so python3 bench.py ${@:---workload score --score-cols 16} --steps ${STEPS:-10} --warmup 2 --cpu-seconds 0 --configs none --no-sql 2>/dev/null | tail -1)
    printf "%-44s " "$name"; python3 - "$line" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[1]); print(f"kernel {d['roofline']['kernel_ms_avg']:8.2f} ms  step {d['ms_per_step']:8.2f} ms  verified {d['verified']}")
except Exception as e:
    print("failed:", sys.argv[1][:200])
PY
  done
fi
