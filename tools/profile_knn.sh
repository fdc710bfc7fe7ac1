# rocprofv3 passes over tools/knn_time.py (GPU box).  bash tools/profile_knn.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-knnprof}
mkdir -p $O
K="python3 tools/knn_time.py"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats -o knn --output-format csv -- $K > $O/stats.out 2> $O/stats.err || echo stats failed
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $O/sq1 -o p --output-format csv -- $K > $O/sq1.out 2> $O/sq1.err || echo sq1 failed
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/sq2 -o p --output-format csv -- $K > $O/sq2.out 2> $O/sq2.err || echo sq2 failed
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $O/sq3 -o p --output-format csv -- $K > $O/sq3.out 2> $O/sq3.err || echo sq3 failed
cat $O/stats/knn_kernel_stats.csv | cut -c1-220 | head -14
