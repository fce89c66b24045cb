#!/bin/bash
# SQ counters of the voxel-gradient kernels of one bench step, run ON the GPU box (via gpurun):
#   bash tools/profile_gather.sh <tag> [bench args]        (env, e.g. XVR_DRR_GATHER_TABLE=0, is inherited)
# Two PMC passes (8 SQ slots each), never combined with a trace.
set -u
TAG=${1:-gprof}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-variants $* --steps 1 --warmup 0"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
python - "$O" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
pm = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{root}/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("gather_vol", "gather_tab", "gather_clip", "gather_px", "splat")):
            pm[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in pm.items():
    print(k)
    for c, val in sorted(v.items()):
        print(f"   {c}: {val:.4g}")
    if v.get("SQ_INSTS_VALU"):
        print(f"   active lanes per VALU instruction: {v['SQ_THREAD_CYCLES_VALU'] / v['SQ_INSTS_VALU']:.1f}")
PY
find $O -type f -size +2M -delete
