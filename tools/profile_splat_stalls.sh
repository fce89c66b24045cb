#!/bin/bash
# Where k_trilinear_splat_b16's wave-cycles go: SQ busy / wait / LDS counters of the benchmark's voxel-gradient
# launch, every group in its own rocprofv3 run (no trace domains next to --pmc), plus the per-phase shader clocks of tools/splat_trace.py.
# Run ON the GPU box (via gpurun):  bash tools/profile_splat_stalls.sh <tag>
set -u
TAG=${1:-splat_stalls}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-variants --steps 1 --warmup 0 --full-json $O/full.json"
rocprofv3 -L > $O/counters.txt 2>&1
pass() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- $B > $O/$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_WAVES SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
pass d SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_WAVE32_LDS SQ_THREAD_CYCLES_VALU
python - "$O" <<'PY'
import csv, glob, re, sys, collections
O = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Kernel_Name"])
        if not m: continue
        k = m.group(0)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
for k in sorted(tot):
    if "splat" not in k and "trilinear_fwd" not in k: continue
    print("###", k)
    for c, v in sorted(tot[k].items()): print(f"- {c}: {v / max(len(disp[(k, c)]), 1):.6g}  (per dispatch, {len(disp[(k, c)])} dispatches)")
PY
python $R/tools/splat_trace.py > $O/trace.txt 2>&1
find $O -type f -size +4M -delete
