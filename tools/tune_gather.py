"""A/B builds of the table gather (waves per SIMD x table rows): builds one diagnostic library per variant HERE (hipcc
cross-compiles) or on the GPU box, and times the trilinear backward with each through bench.py.
    python tools/tune_gather.py build      # in the build container
    python tools/tune_gather.py run        # on the GPU box (gpurun)"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

VARIANTS = [(6, 13), (5, 15), (7, 11)]


def lib(w, t):
    return ROOT / "tools" / "_build" / f"libxvr_drr_tune_w{w}_t{t}.so"


if sys.argv[1:] == ["build"]:
    from xvr_amd.build import build_diagnostic_library
    for w, t in VARIANTS:
        print(build_diagnostic_library([f"XVR_TAB_WAVES={w}", f"XVR_TAB_ROWS={t}"], lib(w, t)))
else:
    for w, t in VARIANTS:
        env = dict(os.environ, XVR_DRR_LIBRARY=str(lib(w, t)))
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"waves {w} rows {t}: step {d['ms_per_step']:.2f} ms, voxel gradient {d['kernels_ms']['trilinear_backward[vol]']:.3f} ms", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"waves {w} rows {t}: failed ({e}) {out.stderr[-300:]}", flush=True)
