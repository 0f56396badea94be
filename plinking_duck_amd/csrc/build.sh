#!/bin/bash
# Builds libpgenhip.so for gfx950 in-tree (the .so travels to the GPU box with the repo snapshot).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p build
pids=()
for f in tally.hip unpack.hip score.hip score_i8.hip reduce.hip pca.hip pca_i8.hip decode.hip ld.hip dosage.hip phase.hip api_dataset.cpp api_analysis.cpp api_reader.cpp api_sharded.cpp api_tally.cpp pgen_file.cpp linalg.cpp; do
	o=build/${f%.*}.o
	if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -maxdepth 1 -name '*.hpp' -newer "$o")" ] || [ ../../include/pgenhip.h -nt "$o" ]; then
		# translation units are independent: compile them side by side
		if [ "${f##*.}" = "hip" ]; then
			$HIPCC $FLAGS -c "$f" -o "$o" &
		else
			$HIPCC $FLAGS -x hip -c "$f" -o "$o" &
		fi
		pids+=($!)
	fi
done
for p in "${pids[@]}"; do
	wait "$p"
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libpgenhip.so build/tally.o build/unpack.o build/score.o build/score_i8.o build/reduce.o build/pca.o build/pca_i8.o build/decode.o build/ld.o build/dosage.o build/phase.o build/api_dataset.o build/api_analysis.o build/api_reader.o build/api_sharded.o build/api_tally.o build/pgen_file.o build/linalg.o
echo "built $(cd .. && pwd)/libpgenhip.so"

# host-side table-function shells (plain C++; link against the C ABI only)
CXX=${CXX:-g++}
SHELL_SRCS="plink_common pgen_reader plink_freq plink_hardy plink_missing plink_score plink_pca plink_ld pfile_reader extension"
objs=""
for n in $SHELL_SRCS; do
	o=build/shell_$n.o
	if [ ! -f "$o" ] || [ shell/$n.cpp -nt "$o" ] || [ -n "$(find shell -maxdepth 1 -name '*.hpp' -newer "$o")" ] || [ ../../include/pgenhip.h -nt "$o" ]; then
		$CXX -O2 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -Wno-redundant-move -c shell/$n.cpp -o "$o"
	fi
	objs="$objs $o"
done
$CXX -shared -fPIC -o ../libplinking_duck_amd.so $objs -L.. -lpgenhip -Wl,-rpath,'$ORIGIN' -lpthread
echo "built $(cd .. && pwd)/libplinking_duck_amd.so"
