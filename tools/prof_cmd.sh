#!/bin/bash
# rocprofv3 kernel statistics of one command on the GPU box: bash tools/prof_cmd.sh <tag> python3 <script> [args]
#   -> gpurun_out/<tag>/stats/*kernel_stats.csv (+ the first rows printed).  The program comes directly after `--`.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- "$@" > $O/stdout.log 2> $O/stderr.log || echo "rocprofv3 failed rc=$?"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
tail -3 $O/stdout.log
[ -n "$f" ] && cut -d, -f1-8 "$f" | head -${PROF_ROWS:-14} | cut -c1-240
