"""Ablated diagnostic builds of the Siddon voxel gather (k_siddon_gather_vol2<true>; -DXVR_SG_ABLATE: results are WRONG by
construction -- separate libraries under tools/_build/, never the product's):
  1  candidates made up in registers from (i, j) instead of loaded      -> what the arithmetic and the loops cost without memory
  2  candidates loaded but not evaluated                                -> what the loads cost without the arithmetic
  3  visits without their candidates (windows, footprint copy, permutes) -> the per-visit floor
python tools/ablate_siddon_gather.py build        (here, cross-compiles)
python tools/ablate_siddon_gather.py              (on the GPU box)"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
LIBS = {k: ROOT / "tools" / "_build" / f"libxvr_drr_sgablate{k}.so" for k in (1, 2, 3)}
if sys.argv[1:] == ["build"]:
    from xvr_amd.build import build_diagnostic_library
    for k, lib in LIBS.items():
        print(build_diagnostic_library(f"XVR_SG_ABLATE={k}", lib, only=["drr_gather.hip"]))
else:
    names = {0: "product", 1: "no candidate loads", 2: "loads only", 3: "visits only"}
    for k in (0, 1, 2, 3):
        env = {"XVR_DRR_LIBRARY": str(LIBS[k])} if k else {}
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--renderer", "siddon", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-variants"],
                             env=dict(os.environ, **env), capture_output=True, text=True)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        print(f"{names[k]:20s}: voxel gradient {d['kernels_ms']['siddon_backward[vol]']:.3f} ms", flush=True)
