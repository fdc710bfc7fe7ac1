cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PROF_ROWS=4 bash tools/prof_cmd.sh knn_one python3 tools/scratch/knn_rect.py tools/exp_libs/knnx_one.so 100000 100000 > gpurun_out/knn_one.txt 2>&1
PROF_ROWS=4 bash tools/prof_cmd.sh knn_two python3 tools/scratch/knn_rect.py tools/exp_libs/knnx_two.so 100000 50000 > gpurun_out/knn_two.txt 2>&1
grep "cosine_pass1_kernel<16, 1" gpurun_out/knn_one.txt gpurun_out/knn_two.txt | cut -d, -f2-4
