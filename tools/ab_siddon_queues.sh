#!/bin/bash
# k_siddon_splat's brick queue, A/B on one box: one queue per XCD with the bricks of a column along y kept together (round 6,
# XVR_SS_XCD_QUEUES=1, the product) against the single centre-out queue of round 5 (a diagnostic build with XVR_SS_XCD_QUEUES=0),
# at C3 under dims = shape + 1 and under the exact map through the splat; then the fabric read requests of both.
# Run ON the GPU box (via gpurun):  bash tools/ab_siddon_queues.sh > gpurun_out/r06_siddon_queue_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
OLD=$(python - <<PY
import sys; sys.path.insert(0, "$R")
from xvr_amd.build import build_diagnostic_library, ROOT
print(build_diagnostic_library(["XVR_SS_XCD_QUEUES=0"], ROOT / "tools" / "_build" / "libxvr_drr_ssq0.so", only=["drr_gather.hip"]))
PY
)
B="python $R/bench.py --renderer siddon --no-variants --no-cpu-baseline --steps 10 --warmup 3"
KW='{"norm_dims_offset": 1}'
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/step %.2f' % d['ms_per_step'], {k: v for k,v in d['kernels_ms'].items() if v>0.05})"; }
for rep in 1 2; do
$B --drr-kwargs "$KW" 2>/dev/null | show "NX  per-XCD queues (product)  "
XVR_DRR_LIBRARY=$OLD $B --drr-kwargs "$KW" 2>/dev/null | show "NX  one queue (round 5)       "
done
XVR_DRR_SIDDON_SPLAT=2 $B 2>/dev/null | show "exact map, splat, per-XCD     "
XVR_DRR_LIBRARY=$OLD XVR_DRR_SIDDON_SPLAT=2 $B 2>/dev/null | show "exact map, splat, one queue   "
cd /tmp
for tag in new old; do
  lib=""; [ $tag = old ] && lib=$OLD
  XVR_DRR_LIBRARY=$lib timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r06_ssq_$tag -- $B --drr-kwargs "$KW" --steps 1 --warmup 0 > /dev/null 2>&1
  python - $R/gpurun_out/r06_ssq_$tag $tag <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_siddon_splat" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print(sys.argv[2], "k_siddon_splat per dispatch:", {k: "%.4g" % (v / max(len(n[k]), 1)) for k, v in sorted(tot.items())})
PY
done
find $R/gpurun_out/r06_ssq_new $R/gpurun_out/r06_ssq_old -type f -size +2M -delete
