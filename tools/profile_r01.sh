set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-knn --no-cpu > $O/fetch.out 2> $O/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-knn --no-cpu > $O/write.out 2> $O/write.err
timeout -k 10 300 python3 bench.py > $O/bench_final.json 2> $O/bench.err
timeout -k 10 300 python3 bench.py --graph uniform --no-knn --no-cpu > $O/bench_uniform.json 2> $O/benchu.err
timeout -k 10 300 python3 bench.py --train-steps 3 --no-knn --no-cpu > $O/bench_train.json 2> $O/bencht.err || true
ls -la $O $O/stats $O/fetch | head -40
