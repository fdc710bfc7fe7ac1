# Round-3 rocprofv3 evidence (run on the GPU box through gpurun; program directly after `--`, counters in their own passes).
#   bash tools/profile_r03.sh [tag]      -> gpurun_out/<tag>/...   (then: python tools/pmc_summary.py gpurun_out/<tag> profiles/r03)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03prof}
mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
B="python3 bench.py --steps 3 --warmup 1 --no-knn --no-cpu --no-graph-replay --train-steps 0 --no-partitioned-check"
K="python3 bench.py --config c5 --no-cpu"
run() { name=$1; shift; timeout -k 10 240 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err || echo "$name failed rc=$?"; }
# 1. kernel time statistics of the bench command itself
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu --train-steps 0 --no-partitioned-check > $O/bench_under_rocprof.json 2> $O/stats.err || echo "stats failed"
# 2. HBM-side traffic (one counter per pass: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2)
for g in local uniform; do
  run fetch_$g --pmc FETCH_SIZE -d $O/fetch_$g -o p --output-format csv -- $B --graph $g --no-uniform
  run write_$g --pmc WRITE_SIZE -d $O/write_$g -o p --output-format csv -- $B --graph $g --no-uniform
  run tcc_$g --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc_$g -o p --output-format csv -- $B --graph $g --no-uniform
done
# 3. issue mix / busy counters of the forward's kernels (8 SQ slots per pass)
run sq1_local --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY -d $O/sq1_local -o p --output-format csv -- $B --no-uniform
run sq2_local --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/sq2_local -o p --output-format csv -- $B --no-uniform
# 4. the kNN bridge (C5)
run knn_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $O/knn_sq1 -o p --output-format csv -- $K
run knn_sq2 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/knn_sq2 -o p --output-format csv -- $K
run knn_fetch --pmc FETCH_SIZE -d $O/knn_fetch -o p --output-format csv -- $K
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/knn_stats -o knn --output-format csv -- $K > $O/knn_under_rocprof.json 2> $O/knn_stats.err || echo "knn stats failed"
find $O -name "*.csv" | head -50
du -sh $O
