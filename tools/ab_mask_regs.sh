#!/bin/bash
# C5's masked renders: up to eight label channels summed in registers (round 6, XVR_FWD_MASK_REGS=1, the product) against the
# per-lane LDS accumulators (a diagnostic build with XVR_FWD_MASK_REGS=0).  Run ON the GPU box:  bash tools/ab_mask_regs.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
OLD=$(python - <<PY
import sys; sys.path.insert(0, "$R")
from xvr_amd.build import build_diagnostic_library, ROOT
print(build_diagnostic_library(["XVR_FWD_MASK_REGS=0"], ROOT / "tools" / "_build" / "libxvr_drr_maskregs0.so", only=["drr_trilinear.hip"]))
PY
)
for rep in 1 2; do
echo "== registers (product)"; python $R/tools/bench_training_step.py 2>/dev/null | tail -12
echo "== LDS accumulators (round 5)"; XVR_DRR_LIBRARY=$OLD python $R/tools/bench_training_step.py 2>/dev/null | tail -12
done
