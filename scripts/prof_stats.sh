#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the training step and of one K = 64 reflectance view (GPU box, repo root):
#   bash scripts/prof_stats.sh [tag]   ->   gpurun_out/stats_<tag>/{geo_train,decomp_k64}_kernel_stats.csv
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/stats_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python3 scripts/probe_train.py 2560 > $OUT/train.log 2>&1 || { tail -5 $OUT/train.log; exit 1; }
VQN_K=64 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k64 -o t -- python3 scripts/probe_decomp_glue.py 3 > $OUT/k64.log 2>&1 || { tail -5 $OUT/k64.log; exit 1; }
cp $(find $OUT/train -name '*kernel_stats.csv' | head -1) $OUT/geo_train_kernel_stats.csv
cp $(find $OUT/k64 -name '*kernel_stats.csv' | head -1) $OUT/decomp_k64_kernel_stats.csv
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
head -8 $OUT/geo_train_kernel_stats.csv | cut -c1-160; head -8 $OUT/decomp_k64_kernel_stats.csv | cut -c1-160
