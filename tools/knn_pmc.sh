set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/knnpmc
mkdir -p $O
K="python3 tools/knn_time.py"
timeout -k 10 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT -d $O/a -o p --output-format csv -- $K > $O/a.out 2> $O/a.err || echo a failed
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM SQ_INSTS_BRANCH -d $O/b -o p --output-format csv -- $K > $O/b.out 2> $O/b.err || echo b failed
