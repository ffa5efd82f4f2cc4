#!/bin/bash
# HBM-side traffic of the headline step: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) over
# `bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1`, aggregated per kernel into profiles/<tag>_pmc_render.json
# (read by bench.py for roofline.traffic).  Usage (GPU box, repo root): bash scripts/pmc_traffic.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/pmct_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o c -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/$c.log 2>&1 || { tail -5 $OUT/$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(neus_points\w*<\w+>|upsample_kernel|merge_kernel|section_mids_kernel|composite_fwd_kernel)', r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1, MI355X (scripts/pmc_traffic.sh)",
       "units": "counter values are KB per launch as reported; gfx950 corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 for 16 B/lane coalesced reads, WRITE_SIZE exact; both count fabric-side requests, Infinity-Cache hits included",
       "kernels": {}, "rays_per_launch": 640000}
for k, cs in agg.items():
    out["kernels"][k] = {c + "_KB_mean_per_launch": sum(v) / len(v) for c, v in cs.items()}
    out["kernels"][k]["launches"] = max(len(v) for v in cs.values())
fine = [k for k in out["kernels"] if "<true>" in k]
if fine:
    kf = out["kernels"][fine[0]]
    out["dominant_kernel"] = fine[0]
    out["dominant_kernel_traffic_bytes_per_launch"] = (2.0 * kf["FETCH_SIZE_KB_mean_per_launch"] + kf["WRITE_SIZE_KB_mean_per_launch"]) * 1024.0
json.dump(out, open("$OUT/pmc_render.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
find $OUT -name "*kernel_trace.csv" -delete
