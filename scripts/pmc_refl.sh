#!/bin/bash
# Separate rocprofv3 --pmc passes (kernel-trace only) over one reflectance training process (scripts/probe_decomp_train.py, 262,144 points):
# matrix-pipe busy cycles, VALU issue and L2 traffic of the round-4 kernels.  Usage (GPU box, repo root): bash scripts/pmc_refl.sh [tag]
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/pmc_refl_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 scripts/probe_decomp_train.py 262144 6 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
pat = re.compile(r"(refl_train_\\w+_kernel<\\d>|wgrad_x3_lds_batched\\w*kernel(<\\w+>)?|wgrad_thin_kernel|wgrad_finalize_kernel|brdf_shade\\w*kernel<[\\w, ]+>|vq_train_bwd_kernel)")
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = pat.search(r["Kernel_Name"])
        if m:
            agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --kernel-trace --pmc <set> (3 separate passes) -- python3 scripts/probe_decomp_train.py 262144 6 (6 eager steps of the reflectance trainer, "
                 "262,144 points), MI355X", "kernels": {}}
for name, cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = dict(m)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        d["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    if "TCC_REQ_sum" in m and "TCC_MISS_sum" in m:
        d["l2_hit_rate"] = 1.0 - m["TCC_MISS_sum"] / max(m["TCC_REQ_sum"], 1.0)
    d["launches_seen"] = max(len(v) for v in cs.values())
    out["kernels"][name] = d
json.dump(out, open("$OUT/pmc_refl.json", "w"), indent=1)
print(json.dumps({k: {kk: round(vv, 4) for kk, vv in v.items() if kk in ("mfma_busy_frac", "l2_hit_rate", "launches_seen")} for k, v in out["kernels"].items()}, indent=1))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
