"""Build the HIP library in-tree: ``hipcc --offload-arch=gfx950`` -> xvr_amd/lib/libxvr_drr.so.

The .so is git-ignored (history stays source-only) but travels to the GPU box with the snapshot.
hipcc cross-compiles gfx950 without a GPU, so this also runs in the CPU-only build container.
"""

from __future__ import annotations

import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SRC = [PKG / "csrc" / "drr_kernels.hip", PKG / "csrc" / "sim_kernels.hip", PKG / "csrc" / "volume_kernels.hip",
       PKG / "csrc" / "pose_kernels.hip"]
HDR = [ROOT / "include" / "xvr_drr.h", ROOT / "include" / "xvr_sim.h", ROOT / "include" / "xvr_pose.h"]
LIB = PKG / "lib" / "libxvr_drr.so"

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-shared",
    "-fPIC",
    "-munsafe-fp-atomics",  # fp32 atomic add in hardware (global_atomic_add_f32), no CAS loops
    f"-I{ROOT / 'include'}",
]


def is_stale() -> bool:
    if not LIB.exists():
        return True
    built = LIB.stat().st_mtime
    return any(p.stat().st_mtime > built for p in SRC + HDR if p.exists())


def build_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        raise RuntimeError("hipcc not found: cannot build libxvr_drr.so (ROCm toolchain required)")
    LIB.parent.mkdir(parents=True, exist_ok=True)
    cmd = [hipcc, *HIPCC_FLAGS, "-o", str(LIB), *map(str, SRC)]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
