set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/train
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o t --output-format csv -- python3 bench.py --train-steps 8 --steps 2 --warmup 1 --no-knn --no-cpu --no-graph-replay --no-uniform > $O/out.txt 2> $O/err.txt
