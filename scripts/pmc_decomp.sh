#!/bin/bash
# HBM-side traffic of the decomp path's kernels: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only)
# over one full-view vq_nfr.call (scripts/probe_decomp_glue.py, 640,000 rows, 512,000 foreground) and the standalone VQ
# kernels at 1 M rows (scripts/probe_vq.py), aggregated per kernel into gpurun_out/pmcd_<tag>/pmc_decomp.json.
# Usage (GPU box, repo root): bash scripts/pmc_decomp.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/pmcd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/call_$c -o c -- python3 scripts/probe_decomp_glue.py 3 > $OUT/call_$c.log 2>&1 || { tail -5 $OUT/call_$c.log; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/vq_$c -o c -- python3 scripts/probe_vq.py > $OUT/vq_$c.log 2>&1 || { tail -5 $OUT/vq_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections, re
pat = re.compile(r'(mlp_chain_vq_kernel|mlp_chain_kernel<[^>]*>|brdf_shade_kernel<[^>]*>|vq_assign_kernel<[^>]*>|vq_assign_split_kernel<[^>]*>|l2_normalize_rows_kernel|vq_ema_mfma_kernel<[^>]*>|vq_ema_reduce2_kernel|vq_ste_loss_kernel|vq_counts_kernel)')
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), MI355X (scripts/pmc_decomp.sh)",
       "units": "KB per launch as reported; gfx950 corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 for 16 B/lane coalesced reads, WRITE_SIZE exact; fabric-side requests, Infinity-Cache hits included",
       "workloads": {}}
for wl, rows, note in (("call", 512000, "one vq_nfr.call(mode='vali') on a 640,000-row view with 512,000 foreground rows, 512 lights, visibility rows"),
                       ("vq", 1 << 20, "standalone vqn_vq_assign / vqn_vq_ema_stats, 1,048,576 x 256, K = 15 / 32 / 64 in turn")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s_*/**/*counter_collection.csv" % wl, recursive=True):
        for r in csv.DictReader(open(f)):
            m = pat.search(r["Kernel_Name"])
            if m:
                agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ks = {}
    for k, cs in agg.items():
        e = {c + "_KB_mean_per_launch": sum(v) / len(v) for c, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE_KB_mean_per_launch" in e and "WRITE_SIZE_KB_mean_per_launch" in e:
            e["traffic_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KB_mean_per_launch"] + e["WRITE_SIZE_KB_mean_per_launch"]) * 1024.0
            e["traffic_bytes_per_row"] = e["traffic_bytes_per_launch"] / rows
        ks[k] = e
    entry = {"note": note, "rows": rows, "kernels": ks}
    if wl == "call":                                  # 3 calls per pass: everything the call's kernels move, per foreground row
        entry["traffic_bytes_per_row_whole_call"] = sum(e.get("traffic_bytes_per_launch", 0.0) * e["launches"] for e in ks.values()) / 3.0 / rows
    out["workloads"][wl] = entry
json.dump(out, open("$OUT/pmc_decomp.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
find $OUT -name "*kernel_trace.csv" -delete
