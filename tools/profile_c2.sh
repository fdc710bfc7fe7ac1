set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c2prof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o t --output-format csv -- python3 bench.py --config c2 --no-cpu --no-graph-replay --steps 50 --warmup 10 > $O/out.txt 2> $O/err.txt
