set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rank
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-knn --no-cpu --no-graph-replay --force-dist --nodes 125000 --edges 2500000 > $O/out.json 2> $O/err.txt
ls $O/stats
