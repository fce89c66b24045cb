R=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $R/bench.py --no-cpu-baseline --no-variants --steps 10 --warmup 3"
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'step %.2f' % d['ms_per_step'], 'splat %.3f' % d['kernels_ms']['trilinear_backward[vol]'], 'fwd %.3f' % d['kernels_ms']['trilinear_forward+jac'])"; }
for rep in 1 2 3; do
for v in b16 b16_sx_odd b16_sx327 b16_sx333 b16_sx341 b16_sx325_5w; do
XVR_DRR_LIBRARY=$R/tools/_build/libxvr_drr_tune_$v.so $B 2>/dev/null | show $v
done; done
