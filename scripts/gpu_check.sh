#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, rocprof kernel-trace of the same bench command.
# Usage (from the repo root on the box):  bash scripts/gpu_check.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/status.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.log; echo "pytest rc=$rc" | tee -a $OUT/status.txt
[ $rc -ne 0 ] && exit $rc
echo "== smoke" | tee -a $OUT/status.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?
tail -3 $OUT/smoke.log; echo "smoke rc=$rc" | tee -a $OUT/status.txt
[ $rc -ne 0 ] && exit $rc
echo "== bench" | tee -a $OUT/status.txt
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?
cat $OUT/bench.json; tail -3 $OUT/bench.err; echo "bench rc=$rc" | tee -a $OUT/status.txt
[ $rc -ne 0 ] && exit $rc
echo "== rocprofv3 kernel trace" | tee -a $OUT/status.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --no-cpu-baseline --no-extras --no-detail > $OUT/rocprof.log 2>&1; rc=$?
tail -3 $OUT/rocprof.log; echo "rocprof rc=$rc" | tee -a $OUT/status.txt
find $OUT/prof -name '*kernel_stats*' | head; for f in $(find $OUT/prof -name '*kernel_stats.csv'); do head -12 $f; done
# the raw trace is large: keep only the stats
find $OUT/prof -name '*kernel_trace.csv' -size +8M -delete; find $OUT/prof -name '*.db' -delete
exit $rc
