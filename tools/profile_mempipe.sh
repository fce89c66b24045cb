#!/bin/bash
# Vector-memory pipe counters (TA / TCP / TD) of the render kernels of one bench step, run ON the GPU box (via gpurun):
#   bash tools/profile_mempipe.sh <tag> [bench args]
# Answers "is the kernel waiting on the texture-address unit, on L1 tag look-ups, or on the L2's answers?" -- what the
# SQ counters of tools/profile_gather.sh cannot tell.  Four PMC passes of four counters, never combined with a trace.
set -u
TAG=${1:-mempipe}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-variants $* --steps ${STEPS:-1} --warmup ${WARMUP:-0}"
P() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$n -- $B > $O/pmc_$n.log 2>&1; }
P 1 GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
P 2 TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum
P 3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
P 4 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
P 5 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum
python - "$O" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
pm = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{root}/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("gather", "k_trilinear_fwd", "k_siddon")):
            pm[k[:90]][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k[:90]][r["Counter_Name"]] += 1
for k, v in pm.items():
    print(k)
    for c, val in sorted(v.items()):
        print(f"   {c}: {val:.4g}   ({n[k][c]} dispatches)")
PY
tail -3 $O/pmc_*.log | grep -i -B1 -A2 "error\|fail" | head -20
find $O -type f -size +2M -delete
