#!/bin/bash
# Round-3 rocprofv3 --kernel-trace --stats summaries (GPU box, repo root):  bash scripts/prof_stats_r03.sh
#   gpurun_out/stats_r03/{geo_train,decomp_k64,decomp_train_graph2048,decomp_train_256k,geo_render_x3}_kernel_stats.csv
set -o pipefail
OUT=gpurun_out/stats_r03
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name, command...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o t -- "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
  cp "$(find $OUT/$name -name '*kernel_stats.csv' | head -1)" $OUT/${name}_kernel_stats.csv
  find $OUT/$name -name '*kernel_trace.csv' -delete; find $OUT/$name -name '*.db' -delete
  echo "== $name"; head -6 $OUT/${name}_kernel_stats.csv | cut -c1-170
}
run geo_train python3 scripts/probe_train.py 2560 &&
VQN_K=64 run decomp_k64 python3 scripts/probe_decomp_glue.py 3 &&
run decomp_train_graph2048 python3 scripts/probe_decomp_train.py 2048 24 graph &&
run decomp_train_256k python3 scripts/probe_decomp_train.py 262144 4 &&
PROBE_B=80000 run geo_render_x3 python3 scripts/probe_neus_f16s.py x3
