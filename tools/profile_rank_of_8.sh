set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rank8
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o t --output-format csv -- python3 tools/rank_of_8_time.py > $O/out.txt 2> $O/err.txt
tail -3 $O/out.txt
