#!/bin/bash
# Runs on the MI355X box (via gpurun): the round's benches + rocprofv3 summaries.
# Outputs under gpurun_out/profiles/; the judged copies are committed under profiles/ (rNN_ prefix,
# tools/summarise_profiles.py --round NN).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles
rm -rf $OUT; mkdir -p $OUT
run_bench() { # name args...
	local name=$1; shift
	python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
	echo "bench $name exit $?"
}
stats() { # name args...
	local name=$1; shift
	rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 bench.py "$@" --cpu-seconds 0 > /dev/null 2> $OUT/stats_$name.err
	cp $OUT/stats_$name/*/*_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null
	echo "stats $name exit $?"
}
pmc() { # name tag "counters" args...
	local name=$1; local tag=$2; local ctr=$3; shift 3
	rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${name}_$tag -- python3 bench.py "$@" --cpu-seconds 0 > /dev/null 2> $OUT/pmc_${name}_$tag.err
	cp $OUT/pmc_${name}_$tag/*/*_counter_collection.csv $OUT/${name}_pmc_$tag.csv 2>/dev/null
	echo "pmc $name $tag exit $?"
}
run_bench freq --steps 20 --warmup 3
run_bench fused --workload fused --steps 10 --warmup 2 --cpu-seconds 0
run_bench unpack --workload unpack --steps 5 --warmup 1 --cpu-seconds 0
run_bench score --workload score --steps 3 --warmup 1 --cpu-seconds 0
run_bench score1 --workload score --score-cols 1 --steps 3 --warmup 1 --cpu-seconds 0
run_bench score2 --workload score --score-cols 2 --steps 3 --warmup 1 --cpu-seconds 0
run_bench score4 --workload score --score-cols 4 --steps 3 --warmup 1 --cpu-seconds 0
run_bench score8 --workload score --score-cols 8 --steps 3 --warmup 1 --cpu-seconds 0
run_bench pca --workload pca --variants 100000 --steps 2 --warmup 1 --cpu-seconds 0
run_bench ld --workload ld --variants 20000 --steps 3 --warmup 1 --cpu-seconds 0
run_bench samplecounts --workload samplecounts --steps 3 --warmup 1 --cpu-seconds 0
run_bench missingsample --workload missingsample --steps 5 --warmup 1 --cpu-seconds 0
run_bench dosagefreq --workload dosagefreq --steps 5 --warmup 2 --cpu-seconds 0
run_bench dosagescore --workload dosagescore --steps 3 --warmup 1 --cpu-seconds 0
run_bench dosagefull --workload dosagescore --dosage-rate 1.0 --variants 50000 --steps 3 --warmup 1 --cpu-seconds 0
run_bench dosagegaps --workload dosagescore --dosage-rate 0.8 --variants 50000 --steps 3 --warmup 1 --cpu-seconds 0
stats freq --steps 10 --warmup 2 --configs none
stats fused --workload fused --steps 5 --warmup 1
stats unpack --workload unpack --steps 3 --warmup 1
stats score --workload score --steps 3 --warmup 1
stats score1 --workload score --score-cols 1 --steps 3 --warmup 1
stats pca --workload pca --variants 100000 --steps 1 --warmup 0
stats dosagescore --workload dosagescore --steps 2 --warmup 1
stats dosagefreq --workload dosagefreq --steps 3 --warmup 1
stats ld --workload ld --variants 20000 --steps 3 --warmup 1
stats samplecounts --workload samplecounts --steps 3 --warmup 1
stats missingsample --workload missingsample --steps 3 --warmup 1
for c in FETCH_SIZE WRITE_SIZE; do
	pmc freq $c $c --steps 3 --warmup 1 --configs none
	pmc fused $c $c --workload fused --steps 3 --warmup 1
	pmc unpack $c $c --workload unpack --steps 2 --warmup 1
	pmc score1 $c $c --workload score --score-cols 1 --steps 2 --warmup 1
	pmc score $c $c --workload score --steps 2 --warmup 1
	pmc dosagescore $c $c --workload dosagescore --steps 2 --warmup 1
done
# matrix-core utilisation of the int8 contraction (north star: MFMA-utilisation counters against chip peak)
MF="SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
pmc score mfma "$MF" --workload score --steps 2 --warmup 1
pmc score1 mfma "$MF" --workload score --score-cols 1 --steps 2 --warmup 1
pmc pca mfma "$MF" --workload pca --variants 100000 --steps 1 --warmup 0
# the table functions at BASELINE's shape through the SQL shells, and the kernels a plink_hardy call launches when a
# plink_freq call on the same file came first (none that reads the matrix: the tally pass is shared)
python3 tools/shell_bench.py --synth 1000000x500000 --threads 16 > $OUT/shell_bench.txt 2> $OUT/shell_bench.err
echo "shell bench exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_hardy_after_freq -- python3 tools/hardy_after_freq.py > $OUT/hardy_after_freq.txt 2> $OUT/hardy_after_freq.err
cp $OUT/stats_hardy_after_freq/*/*_kernel_trace.csv $OUT/hardy_after_freq_kernel_trace.csv 2>/dev/null
echo "hardy-after-freq trace exit $?"
python3 tools/fused_time.py > $OUT/tally_kernels_alone.txt 2>&1
ls $OUT | head -80 > $OUT/summary.txt
tail -40 $OUT/summary.txt
