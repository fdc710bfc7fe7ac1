# Round-3 bench lines and per-graph kernel statistics (GPU box, through gpurun):  bash tools/collect_r03.sh
#   -> gpurun_out/r03x/...   (copied into profiles/r03 by hand: see profiles/r03/README.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x
mkdir -p $O
F="--steps 20 --warmup 5 --no-cpu --no-knn --train-steps 0 --no-uniform --no-graph-replay --no-partitioned-check"
PROF_ROWS=8 bash tools/prof_cmd.sh r03x/c4local python3 bench.py $F > $O/c4local.txt 2>&1
PROF_ROWS=8 bash tools/prof_cmd.sh r03x/c4uniform python3 bench.py $F --graph uniform > $O/c4uniform.txt 2>&1
python3 bench.py > $O/bench_c4.json 2> $O/bench_c4.err
python3 bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
for w in 2 4 8; do WORLD=$w timeout -k 10 300 python3 tools/rank_of_8_time.py > $O/rank_of_$w.txt 2>&1; done
bash tools/profile_train.sh
tail -c 300 $O/bench_c4.json; grep -h "ms" $O/rank_of_*.txt
