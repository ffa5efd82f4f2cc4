#!/bin/bash
# Matrix-pipe / wait counters of one 256 x 256 weight-gradient launch (f32 and bf16x3 kernels): scripts/probe_wgrad_launch.py
set -o pipefail
OUT=gpurun_out/pmcw_${1:-r02}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 scripts/probe_wgrad_launch.py > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(wgrad_\w+<[^>]*>)', r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    g = m["GRBM_GUI_ACTIVE"] / 8.0
    print(k, "cycles", int(g), "mfma_busy", round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g), 3),
          "valu_busy", round(4 * m.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * g), 3),
          "wait_any/wave_cycles", round(m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
          "wait_inst_any", round(m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
          "wait_lds", round(m.get("SQ_WAIT_INST_LDS", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
          "insts mfma/valu/lds/vmem", int(m.get("SQ_INSTS_MFMA", 0)), int(m.get("SQ_INSTS_VALU", 0)), int(m.get("SQ_INSTS_LDS", 0)), int(m.get("SQ_INSTS_VMEM_RD", 0)))
PY
