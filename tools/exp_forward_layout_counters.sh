#!/bin/bash
# L1-miss / fabric line counters of the trilinear forward on the row and the tiled y-pair copies, run ON the GPU box:
#   bash tools/exp_forward_layout_counters.sh <tag>
set -u
TAG=${1:-fwdlayout}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for T in 1 0; do
  export XVR_DRR_YTILES=$T
  B="python $R/bench.py --no-cpu-baseline --no-variants --no-voxel-grad --steps 1 --warmup 0"
  timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d $O/mem_$T -- $B > $O/mem_$T.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq_$T -- $B > $O/sq_$T.log 2>&1
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/l2_$T -- $B > $O/l2_$T.log 2>&1
done
python - "$O" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
for T in ("1", "0"):
    pm, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(f"{root}/*_{T}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_trilinear_fwd<true" in r["Kernel_Name"]:
                pm[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("tiles" if T == "1" else "rows")
    for c in sorted(pm):
        print(f"   {c}: {pm[c] / n[c]:.5g} per launch ({n[c]} launches)")
PY
find $O -type f -size +2M -delete
