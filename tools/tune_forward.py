"""A/B builds of the trilinear forward's occupancy cap (-DXVR_FWD_WAVES) x volume layout.  build | run"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
WAVES = [3, 4, 5, 6, 8]
lib = lambda w: ROOT / "tools" / "_build" / f"libxvr_drr_fwd_w{w}.so"
if sys.argv[1:] == ["build"]:
    from xvr_amd.build import build_diagnostic_library
    for w in WAVES:
        print(build_diagnostic_library([f"XVR_FWD_WAVES={w}"], lib(w)))
else:
    for w in WAVES:
        for yp in ("1", "0"):
            env = dict(os.environ, XVR_DRR_LIBRARY=str(lib(w)), XVR_DRR_YPAIRS=yp)
            out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-voxel-grad"],
                                 env=env, capture_output=True, text=True)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"waves <= {w}, ypairs {yp}: forward+jac {d['kernels_ms']['trilinear_forward+jac']:.3f} ms", flush=True)
