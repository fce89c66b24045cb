#!/bin/bash
# Siddon under the recalled index map (norm_dims_offset = +1) at C3: the slab march on rows / on bricks, the merge walk, and the
# voxel gradient's variants.  Run on the GPU box:  bash tools/bench_siddon_nx.sh > gpurun_out/r05_siddon_nx.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $R/bench.py --renderer siddon --no-variants --no-cpu-baseline --steps 6 --warmup 2"
KW='{"norm_dims_offset": 1}'
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/step %.2f' % d['ms_per_step'], {k: v for k,v in d['kernels_ms'].items() if v>0.05})"; }
$B --drr-kwargs "$KW" 2>&1 | show "NX default              "
XVR_DRR_BRICKS_NX=0 $B --drr-kwargs "$KW" 2>&1 | show "NX slab on rows         "
XVR_DRR_SIDDON_SLAB=2 $B --drr-kwargs "$KW" 2>&1 | show "NX merge walk           "
XVR_DRR_SIDDON_SPLAT=0 $B --drr-kwargs "$KW" 2>&1 | show "NX cells gather (r4)    "
$B --drr-kwargs "$KW" --no-voxel-grad 2>&1 | show "NX pose only            "
$B 2>&1 | show "exact default           "
XVR_DRR_SIDDON_SPLAT=2 $B 2>&1 | show "exact through the splat "
