#!/bin/bash
# L2 / fabric counters of the trilinear forward under three diagnostics (tools/exp_forward_variants.py one <mode>), run ON the
# GPU box:   bash tools/exp_forward_counters.sh <tag>
# default = the benchmark's 116 poses on the y-pair copy; natural = the same on [x][y][z]; samepose = 116 copies of one pose
# (the launch's whole footprint fits the Infinity Cache).  One PMC group per run, never combined with a trace.
set -u
TAG=${1:-fwdcnt}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC|TCP|TA|TD|SQ|GRBM|MALL|DF|UMC|HBM)[A-Za-z0-9_]*" | sort -u > $O/counters_available.txt
for MODE in default natural samepose; do
  P() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/${MODE}_$n -- python $R/tools/exp_forward_variants.py one $MODE > $O/${MODE}_$n.log 2>&1; }
  P 1 FETCH_SIZE TCC_HIT_sum TCC_MISS_sum
  P 2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum
  P 3 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum TCC_TAG_STALL_sum
  P 4 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
done
python - "$O" <<'PY'
import collections, csv, glob, os, sys
root = sys.argv[1]
for mode in ("default", "natural", "samepose"):
    pm, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(f"{root}/{mode}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_trilinear_fwd" in r["Kernel_Name"]:
                pm[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(mode)
    for c in sorted(pm):
        print(f"   {c}: {pm[c] / n[c]:.5g} per launch ({n[c]} launches)")
PY
grep -il "error\|invalid\|not found" $O/*.log | head
find $O -type f -size +2M -delete
