"""A/B builds of the ray-major splat (refill threshold) timed on the clip_to_volume variant:  build here, run on the GPU box."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
VARIANTS = [("r1", ["XVR_SPX_REFILL=1"]), ("r2", ["XVR_SPX_REFILL=2"]), ("r4", ["XVR_SPX_REFILL=4"]), ("r8", ["XVR_SPX_REFILL=8"])]
lib = lambda n: ROOT / "tools" / "_build" / f"libxvr_drr_tune_px_{n}.so"
if sys.argv[1:] == ["build"]:
    from xvr_amd.build import build_diagnostic_library
    for n, d in VARIANTS:
        print(build_diagnostic_library(d, lib(n)))
else:
    for n, d in VARIANTS:
        out = subprocess.run([sys.executable, str(ROOT / "tools" / "bench_variants.py"), "--steps", "3"], env=dict(os.environ, XVR_DRR_LIBRARY=str(lib(n))), capture_output=True, text=True)
        for l in out.stdout.splitlines():
            if "clip_to_volume" in l or "per-channel" in l:
                print(n, l[:200], flush=True)
