#!/bin/bash
# Round-5 rocprofv3 --kernel-trace --stats summaries (GPU box, repo root):  bash scripts/prof_stats_r05.sh [which...]
#   gpurun_out/stats_r05/{decomp_train_256k,decomp_train_graph2048,geo_train,...}_kernel_stats.csv
set -o pipefail
OUT=gpurun_out/stats_r05
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name, command...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o t -- "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
  cp "$(find $OUT/$name -name '*kernel_stats.csv' | head -1)" $OUT/${name}_kernel_stats.csv
  find $OUT/$name -name '*kernel_trace.csv' -delete; find $OUT/$name -name '*.db' -delete
  echo "== $name"; head -${HEAD:-14} $OUT/${name}_kernel_stats.csv | cut -c1-150
}
for w in "${@:-decomp_train_256k decomp_train_graph2048}"; do
  case $w in
    decomp_train_256k) run decomp_train_256k python3 scripts/probe_decomp_train.py 262144 6 ;;
    decomp_train_graph2048) run decomp_train_graph2048 python3 scripts/probe_decomp_train.py 2048 27 graph ;;
    geo_train) run geo_train python3 scripts/probe_train.py 2560 ;;
    decomp_k64) VQN_K=64 run decomp_k64 python3 scripts/probe_decomp_glue.py 3 ;;
    geo_render_x3) PROBE_B=80000 run geo_render_x3 python3 scripts/probe_neus_f16s.py x3 ;;
    ref_nfr_train_256k) run ref_nfr_train_256k python3 scripts/probe_decomp_train.py 262144 6 ref_nfr ;;
    decomp_train_graph2048_54) run decomp_train_graph2048_54 python3 scripts/probe_decomp_train.py 2048 54 graph ;;
    geo_render) run geo_render python3 bench.py --no-cpu-baseline --no-extras --no-traffic --no-detail ;;     # the headline command's own kernels (5 + 2 steps)
  esac || exit 1
done
