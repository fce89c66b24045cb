"""Loads-ablated diagnostic build of the Siddon slab march (k_siddon_slab; -DXVR_SLAB_ABLATE: voxel values made up from the
offsets, the image is WRONG by construction -- a separate library under tools/_build/, never the product's): what the march
costs without its three scattered 4-byte loads per slab.
python tools/ablate_siddon_slab.py build        (here, cross-compiles)
python tools/ablate_siddon_slab.py              (on the GPU box)"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
LIB = ROOT / "tools" / "_build" / "libxvr_drr_slabablate.so"
if sys.argv[1:] == ["build"]:
    from xvr_amd.build import build_diagnostic_library
    print(build_diagnostic_library("XVR_SLAB_ABLATE=1", LIB, only=["drr_siddon.hip"]))
else:
    for name, env in (("product", {}), ("no voxel loads", {"XVR_DRR_LIBRARY": str(LIB)})):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--renderer", "siddon", "--no-voxel-grad", "--steps", "6", "--warmup", "2",
                              "--no-cpu-baseline", "--no-variants"], env=dict(os.environ, **env), capture_output=True, text=True)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        print(f"{name:16s}: forward + jacobian {d['kernels_ms']['siddon_forward+jac']:.3f} ms", flush=True)
