"""A/B and timing builds of the ray-major splat (list size; XVR_SPX_DIAG 1-3, which give WRONG sums: the phases' prices) timed on
the clip_to_volume and per-channel-mask variants:  `python tools/tune_splat_px.py build` here, `python tools/tune_splat_px.py` on the
GPU box (profiles/r03_splat_px_phases.txt)."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
VARIANTS = [("product", []), ("list64", ["XVR_SPX_WTAB=64"]), ("list256", ["XVR_SPX_WTAB=256"]),
            ("diag1_one_sample_per_lane_and_list", ["XVR_SPX_DIAG=1"]), ("diag2_visits_without_rays", ["XVR_SPX_DIAG=2"]), ("diag3_one_add_per_sample", ["XVR_SPX_DIAG=3"])]
lib = lambda n: ROOT / "tools" / "_build" / f"libxvr_drr_tune_px_{n}.so"
if sys.argv[1:] == ["build"]:
    from xvr_amd.build import build_diagnostic_library
    for n, d in VARIANTS:
        print(build_diagnostic_library(d, lib(n), only=["drr_gather.hip"]))
else:
    for n, d in VARIANTS:
        for only in ("trilinear clip_to_volume", "trilinear mask -> 8 channels, per"):
            out = subprocess.run([sys.executable, str(ROOT / "tools" / "bench_variants.py"), "--steps", "3", "--only", only],
                                 env=dict(os.environ, XVR_DRR_LIBRARY=str(lib(n))), capture_output=True, text=True)
            for l in out.stdout.splitlines():
                if l.startswith("| trilinear") and "batch" not in l:
                    print(f"{n:36s}", l[:200], flush=True)
