"""A/B builds of the brick-local splat kernels: builds one diagnostic library per variant HERE (hipcc cross-compiles)
and times the trilinear backward with each through bench.py ON the GPU box.
    python tools/tune_splat.py build      # in the build container
    python tools/tune_splat.py run        # on the GPU box (gpurun)"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

# (name, defines, XVR_DRR_GATHER_SPLAT: 0 = the table gather)
VARIANTS = [
    ("b16", [], "1"),
    # round 6: the occupancy / LDS-layout knobs once more, on the current kernel (profiles/r06_splat_tuning.txt)
    ("b16_5waves", ["XVR_S16_WAVES=5"], "1"),
    ("b16_5waves_depth1", ["XVR_S16_WAVES=5", "XVR_S16_DEPTH=1"], "1"),
    ("b16_3waves", ["XVR_S16_WAVES=3"], "1"),
    ("b16_sx_odd", ["XVR_S16_SX=325"], "1"),
    ("b16_sy19", ["XVR_S16_SY=19", "XVR_S16_SX=343"], "1"),
    ("b16_tab640", ["XVR_S16_TAB=640"], "1"),
    ("b16_quarters", ["XVR_S16_SHARES=0"], "1"),
    ("b16_pose_global", ["XVR_S16_POSE_GLOBAL=1"], "1"),
    ("b16_noadds", ["XVR_SP_ABLATE_ADDS=1"], "1"),
    ("b16_half_the_adds", ["XVR_SP_ABLATE_ADDS=2"], "1"),
    ("b16_noloads", ["XVR_S16_ABLATE_LOADS=1"], "1"),
    ("b16_noadds_noloads", ["XVR_SP_ABLATE_ADDS=1", "XVR_S16_ABLATE_LOADS=1"], "1"),
]
RENDERER = "trilinear"


def lib(name):
    return ROOT / "tools" / "_build" / f"libxvr_drr_tune_{name}.so"


if sys.argv[1:2] == ["build"]:
    from xvr_amd.build import build_diagnostic_library
    for name, defs, _ in VARIANTS:
        if sys.argv[2:] and not any(name.startswith(p) for p in sys.argv[2:]):
            continue
        print(build_diagnostic_library(defs or ["XVR_TUNE_DEFAULT=1"], lib(name), only=["drr_gather.hip"]))
else:
    for name, defs, mode in VARIANTS:
        if not lib(name).exists():
            continue
        env = dict(os.environ, XVR_DRR_LIBRARY=str(lib(name)), XVR_DRR_GATHER_SPLAT=mode)
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-variants", "--renderer", RENDERER],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"{name} {defs}: step {d['ms_per_step']:.2f} ms, voxel gradient {d['kernels_ms'][RENDERER + '_backward[vol]']:.3f} ms", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name}: failed ({e}) {out.stderr[-300:]}", flush=True)
