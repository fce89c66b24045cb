#!/bin/bash
# FETCH_SIZE / TCC_EA0_RDREQ calibration for 16-byte-per-lane gathers (tools/microbench/fetch_calib.hip), run ON the GPU box:
#   bash tools/exp_fetch_calibration.sh <tag>
# One un-profiled timing run, then one PMC group per run (never combined with a trace).  The table at the end divides every
# counter by the known number of lines of the dispatch.
set -u
TAG=${1:-fetchcal}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
mkdir -p $O; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $R/tools/microbench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
cd /tmp
/tmp/fetch_calib > $O/timing.txt 2>&1
cat $O/timing.txt
P() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$n -- /tmp/fetch_calib > $O/pmc_$n.log 2>&1; }
P 1 FETCH_SIZE TCC_HIT_sum TCC_MISS_sum
P 2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum
P 3 TCC_EA0_RDREQ_DRAM_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
python - "$O" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
# dispatch order of fetch_calib.hip: every variant twice (warm-up, timed); lines per dispatch
big, mall = (2 << 30) // 128, (64 << 20) // 128 * 32
order = [("2GiB seq G8", big), ("2GiB G8", big), ("2GiB G4", big), ("2GiB G2", big), ("2GiB G1", big),
         ("64MiBx32 seq G8", mall), ("64MiBx32 G8", mall), ("64MiBx32 G4", mall), ("64MiBx32 G1", mall)]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{root}/pmc_*/*/*_counter_collection.csv")):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_fetch" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for i, d in enumerate(sorted(per)):
        if i % 2 == 1 and i // 2 < len(order):     # the timed dispatch of each variant
            rows[i // 2].update(per[d])
names = sorted({c for v in rows.values() for c in v})
print("per 128-byte line of the working set (known count):")
print("variant".ljust(18) + "".join(n.replace("_sum", "")[-22:].rjust(24) for n in names))
for i, (tag, lines) in enumerate(order):
    print(tag.ljust(18) + "".join(f"{rows[i].get(n, float('nan')) / lines:24.3f}" for n in names))
PY
find $O -type f -size +2M -delete
