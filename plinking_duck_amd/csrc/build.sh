#!/bin/bash
# Builds libpgenhip.so for gfx950 in-tree (the .so travels to the GPU box with the repo snapshot).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
mkdir -p build
for f in kernels.hip api.cpp pgen_file.cpp; do
	o=build/${f%.*}.o
	if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -maxdepth 1 -name '*.hpp' -newer "$o")" ] || [ ../../include/pgenhip.h -nt "$o" ]; then
		if [ "${f##*.}" = "hip" ]; then
			$HIPCC $FLAGS -c "$f" -o "$o"
		else
			$HIPCC $FLAGS -x hip -c "$f" -o "$o"
		fi
	fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libpgenhip.so build/kernels.o build/api.o build/pgen_file.o
echo "built $(cd .. && pwd)/libpgenhip.so"
