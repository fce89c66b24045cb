#!/bin/bash
# The round's rocprofv3 evidence in one go, ON the GPU box:  bash tools/profile_round.sh r05
# -> gpurun_out/<round>_{trilinear,pose_only,siddon,siddon_nx}/ (tools/profile.sh each), then the default bench line and the Siddon one.
set -u
RD=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/profile.sh ${RD}_trilinear > /dev/null 2>&1
bash $R/tools/profile.sh ${RD}_pose_only --no-voxel-grad > /dev/null 2>&1
bash $R/tools/profile.sh ${RD}_siddon --renderer siddon > /dev/null 2>&1
bash $R/tools/profile.sh ${RD}_siddon_nx --renderer siddon --drr-kwargs '{"norm_dims_offset":1}' > /dev/null 2>&1
for t in trilinear pose_only siddon siddon_nx; do
  O=$R/gpurun_out/${RD}_$t
  cp $(ls -t $O/trace/*/*_kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv 2>/dev/null
  rm -rf $O/trace $O/pmc_*/ 2>/dev/null
done
cd $R
# (the driver's own command; the line it parses -> *_bench_final_default.json, the full result -> *_bench_full_default.json)
python bench.py --steps 20 --warmup 5 --full-json gpurun_out/${RD}_bench_full_default.json > gpurun_out/${RD}_bench_final_default.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --renderer siddon --no-variants --full-json gpurun_out/${RD}_bench_full_siddon.json > gpurun_out/${RD}_bench_final_siddon.json 2>/dev/null
python - <<PY
import json
line = open("gpurun_out/${RD}_bench_final_default.json").read().strip()
d = json.loads(line)
print("the line:", len(line), "bytes")
print("headline", round(d["ms_per_step"], 3), "ms", round(d["value"], 1), "DRRs/s; clip per ray / clip batch / volume changing:", d["value_clip_per_ray"], d["value_clip_batch"], d["value_volume_changing"])
print("roofline", d["roofline"])
print("kernels_ms", d["kernels_ms"])
print("variants", d["variants"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
f = json.load(open("gpurun_out/${RD}_bench_full_default.json"))
v = f["variants"]
print("c4", {k: {a: round(b, 4) for a, b in x.items() if isinstance(b, float)} for k, x in v["c4_register_ms_per_pose_iteration"].items()})
print("c5", round(v["c5_train_step_ms"], 2), v["c5_train_step"]["min_median_max_ms"], {k: round(x, 2) for k, x in v["c5_train_step"]["phases"].items()})
print("c5 under the per-ray clip", round(v["c5_train_step_clip_ms"], 2), v["c5_train_step_clip"]["min_median_max_ms"])
PY
