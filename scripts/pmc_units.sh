#!/bin/bash
# Matrix-pipe and vector-pipe occupancy of the kernels round 1 priced without counters: separate rocprofv3 --pmc
# passes (kernel-trace only, no other trace domains) over
#   scripts/probe_decomp_glue.py 3   one full-view vq_nfr.call: mlp_chain_kernel, brdf_shade_kernel, the VQ kernels
#   scripts/probe_train.py 2560      the geo training step: tile_vm_kernel (vqn_tile_program), wgrad_kernel, neus_points2_kernel<false>
# aggregated per kernel into gpurun_out/pmcu_<tag>/pmc_units.json.  Usage (GPU box, repo root): bash scripts/pmc_units.sh [tag]
#   mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)      (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
#   valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / (1024 x GRBM_GUI_ACTIVE / 8)             (SQ_ACTIVE_INST_* count quad-cycles)
#   valu_flops     = SQ_INSTS_VALU_{ADD,MUL}_F32 x 64 + SQ_INSTS_VALU_FMA_F32 x 128 + SQ_INSTS_VALU_TRANS_F32 x 64   (per wave instr.)
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/pmcu_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/call_p$i -o c -- python3 scripts/probe_decomp_glue.py 3 > $OUT/call_p$i.log 2>&1 || { tail -5 $OUT/call_p$i.log; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/train_p$i -o c -- python3 scripts/probe_train.py 2560 > $OUT/train_p$i.log 2>&1 || { tail -5 $OUT/train_p$i.log; exit 1; }
  echo "pass $i done"
done
python3 - <<PY
import csv, glob, json, collections, re
pat = re.compile(r'(mlp_chain_vq_kernel<[^>]*>|mlp_chain_kernel<[^>]*>|brdf_shade_kernel<[^>]*>|brdf_shade_bwd_kernel<[^>]*>|vq_assign_kernel<[^>]*>|tile_vm_kernel<[^>]*>|wgrad_kernel<[^>]*>|neus_points2?_kernel<\w+>|composite_\w+_kernel)')
out = {"source": "rocprofv3 --kernel-trace --pmc <set> (4 separate passes each) -- python3 scripts/probe_decomp_glue.py 3 / python3 scripts/probe_train.py 2560, MI355X (scripts/pmc_units.sh)",
       "units": "counter means per launch; SQ_* summed over all SIMDs; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md)",
       "workloads": {}}
for wl, note in (("call", "one vq_nfr.call(mode='vali'), 640,000-row view (512,000 foreground rows), 512 lights, visibility rows, K = 15"),
                 ("train", "geo training step, 2560 rays x (64 + 64) samples, full nets")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s_p*/**/*counter_collection.csv" % wl, recursive=True):
        for r in csv.DictReader(open(f)):
            m = pat.search(r["Kernel_Name"])
            if m:
                agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ks = {}
    for k, cs in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        d = dict(m)
        g = m.get("GRBM_GUI_ACTIVE")
        if g:
            simd_cycles = 1024.0 * g / 8.0
            d["duration_cycles"] = g / 8.0
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                d["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
            if "SQ_ACTIVE_INST_VALU" in m:
                d["valu_busy_frac"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / simd_cycles
            if "SQ_INSTS_VALU_FMA_F32" in m:
                fl = 64.0 * (m.get("SQ_INSTS_VALU_ADD_F32", 0) + m.get("SQ_INSTS_VALU_MUL_F32", 0) + m.get("SQ_INSTS_VALU_TRANS_F32", 0)) + 128.0 * m["SQ_INSTS_VALU_FMA_F32"]
                d["valu_f32_flop_per_launch"] = fl
                # issue cost: 4 cycles per add / mul / fma wave instruction, 8 per transcendental (MI355X_MICROARCH.md)
                issue = 4.0 * (m.get("SQ_INSTS_VALU_ADD_F32", 0) + m.get("SQ_INSTS_VALU_MUL_F32", 0) + m["SQ_INSTS_VALU_FMA_F32"]) + 8.0 * m.get("SQ_INSTS_VALU_TRANS_F32", 0)
                d["f32_arith_issue_frac"] = issue / simd_cycles
            if "SQ_WAVE_CYCLES" in m and "SQ_ACTIVE_INST_ANY" in m:
                d["active_inst_any_frac_of_wave_cycles"] = m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_ANY" in m:
                d["note_waits"] = "SQ_WAIT_ANY (parked: s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall), in wave quad-cycles"
        d["launches_seen"] = max(len(v) for v in cs.values())
        ks[k] = d
    out["workloads"][wl] = {"note": note, "kernels": ks}
json.dump(out, open("$OUT/pmc_units.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:7000])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
