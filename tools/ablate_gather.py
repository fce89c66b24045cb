"""Loads-ablated diagnostic builds of the trilinear table gather (-DXVR_GATHER_ABLATE: candidates made up in registers
instead of loaded; results are WRONG by construction -- separate libraries, never the product's).  CAUTION when reading
the numbers (round 2 learnt this the hard way): with constant candidates the compiler hoists their arithmetic out of the
trip, so "ablated = 8.7 ms, half the loads = 10.6 ms" is an UPPER bound on what memory costs, not the cost -- a variant
that really loaded a quarter of the bytes was slower (HISTORY.md section 4.1).  python tools/ablate_gather.py build | run"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
LIB = ROOT / "tools" / "_build" / "libxvr_drr_ablate.so"
LIB2 = ROOT / "tools" / "_build" / "libxvr_drr_ablate2.so"
if sys.argv[1:] == ["build"]:
    from xvr_amd.build import build_diagnostic_library
    print(build_diagnostic_library("XVR_GATHER_ABLATE=1", LIB))
    print(build_diagnostic_library("XVR_GATHER_ABLATE=2", LIB2))
else:
    for name, env in (("product", {}), ("loads ablated", {"XVR_DRR_LIBRARY": str(LIB)}), ("half the loads", {"XVR_DRR_LIBRARY": str(LIB2)})):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                             env=dict(os.environ, **env), capture_output=True, text=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{name}: voxel gradient {d['kernels_ms']['trilinear_backward[vol]']:.3f} ms", flush=True)
