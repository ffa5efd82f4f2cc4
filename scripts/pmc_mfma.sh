#!/bin/bash
# Separate rocprofv3 --pmc passes (kernel-trace only, no other trace domains) over the fused NeuS kernels, f32 and split-precision:
# matrix-pipe busy cycles, L1->L2 read requests / L2 misses, VALU issue.  Usage (GPU box, repo root): bash scripts/pmc_mfma.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PROBE_B=20480
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 scripts/probe_neus_f16s.py f32 f16s x3 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"neus_points\\w*<[\\w, ]+>", r["Kernel_Name"])
        if not m:
            continue
        name = m.group(0)
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --kernel-trace --pmc <set> (4 separate passes) -- python3 scripts/probe_neus_f16s.py f32 f16s x3, PROBE_B=20480 rays "
                 "(SDF-only kernels at 64 samples/ray, fine kernels at 128), MI355X", "kernels": {}}
for name, cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = dict(m)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # GRBM_GUI_ACTIVE comes summed over the 8 XCDs (= 8 x kernel duration x shader clock)
        d["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
    if "TCC_REQ_sum" in m and "TCC_MISS_sum" in m:
        d["l2_hit_rate"] = 1.0 - m["TCC_MISS_sum"] / max(m["TCC_REQ_sum"], 1.0)
    d["launches_seen"] = max(len(v) for v in cs.values())
    out["kernels"][name] = d
json.dump(out, open("$OUT/pmc_mfma.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
