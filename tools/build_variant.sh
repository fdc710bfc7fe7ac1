#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAG ..." : the product library with ONE source rebuilt under extra -D flags ->
# tools/exp_libs/NAME.so (same source text => same source hash, so bridged_gnn_amd._lib loads it via SO_PATH)
set -e
cd "$(dirname "$0")/../bridged_gnn_amd/csrc"
make -s -j4
mkdir -p ../../tools/exp_libs
objs=""
for f in bgnn_api bgnn_csr bgnn_transform bgnn_transform_stream bgnn_transform_cls bgnn_aggregate bgnn_aggregate_bwd bgnn_aggregate_bwd_fast bgnn_knn bgnn_gram bgnn_norm; do
  if [ "$f.hip" == "$2" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-fast-math -Wno-unused-function $3 -c $f.hip -o ../../tools/exp_libs/$1.o
    objs="$objs ../../tools/exp_libs/$1.o"
  else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp_libs/$1.so $objs
rm -f ../../tools/exp_libs/$1.o
