# PMC passes over the training step (GPU box): issue mix / MFMA busy of the backward kernels.  bash tools/profile_train_pmc.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/trainpmc
mkdir -p $O
T="python3 bench.py --train-steps 3 --steps 2 --warmup 1 --no-knn --no-cpu --no-graph-replay --no-uniform"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY -d $O/sq1_train -o p --output-format csv -- $T > $O/sq1.out 2> $O/sq1.err || echo sq1 failed
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/sq2_train -o p --output-format csv -- $T > $O/sq2.out 2> $O/sq2.err || echo sq2 failed
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_train -o p --output-format csv -- $T > $O/fetch.out 2> $O/fetch.err || echo fetch failed
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/write_train -o p --output-format csv -- $T > $O/write.out 2> $O/write.err || echo write failed
find $O -name "*counter_collection.csv" | head
