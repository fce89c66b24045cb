"""Build the HIP library in-tree: ``hipcc --offload-arch=gfx950`` -> xvr_amd/lib/libxvr_drr.so.

The .so is git-ignored (history stays source-only) but travels to the GPU box with the snapshot.
hipcc cross-compiles gfx950 without a GPU, so this also runs in the CPU-only build container.
"""

from __future__ import annotations

import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SRC = [PKG / "csrc" / f for f in ("drr_trilinear.hip", "drr_siddon.hip", "drr_gather.hip", "drr_rays.hip", "drr_api.hip",
                                  "sim_kernels.hip", "volume_kernels.hip", "pose_kernels.hip")]
HDR = [ROOT / "include" / "xvr_drr.h", ROOT / "include" / "xvr_sim.h", ROOT / "include" / "xvr_pose.h",
       PKG / "csrc" / "drr_common.hiph", PKG / "csrc" / "drr_splat.hiph", PKG / "csrc" / "drr_siddon_splat.hiph",
       PKG / "csrc" / "j2c_device.hiph", PKG / "csrc" / "pose_device.hiph",
       Path(__file__).resolve()]   # (this file holds the compiler flags: a change of flags makes the library stale too)
OBJ = PKG / "lib" / "obj"
LIB = PKG / "lib" / "libxvr_drr.so"

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-munsafe-fp-atomics",  # fp32 atomic add in hardware (global_atomic_add_f32), no CAS loops
    f"-I{ROOT / 'include'}",
]
# Per translation unit.  -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 operations into v_pk_{add,mul,fma}_f32, which
# on gfx950 occupy a SIMD for 7.5 clocks against 2.9 for a plain one (tools/microbench/valu_issue.hip,
# profiles/r03_microbench_valu_issue.txt): a pair costs 1.3 x two plain instructions.  The Siddon walk and the Siddon voxel
# gather are vector-issue bound and lose 16 such pairs per trip to it: 9.39 -> 8.91 ms and 11.95 -> 10.56 ms at C3 without it.
# (The trilinear forward waits on its gathers, not on the vector ALU, and measured 2 % SLOWER without the pairs -- fewer
# instructions between its loads; the splat is indifferent.)
EXTRA_FLAGS = {
    "drr_siddon.hip": ["-fno-slp-vectorize"],
    "drr_gather.hip": ["-fno-slp-vectorize"],
}


DIAG = ROOT / "tools" / "_build"   # diagnostic / tuning builds live with the tools, never next to the product library


def diagnostic_path(name: str) -> Path:
    return DIAG / f"libxvr_drr_{name}.so"


def build_diagnostic_library(define, out: Path, only=None) -> Path:
    """A separate library with extra -D's (e.g. XVR_GATHER_STATS; a string or a list of them) for the measuring tools
    under tools/; never loaded by the package itself and never written into xvr_amd/lib/ (some of these builds compute
    deliberately wrong sums).  ``only``: the translation units the defines touch (file names) -- the others are taken from
    the product build's objects, which must be up to date (build_library())."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = Path(out)
    if out.parent.resolve() == (PKG / "lib").resolve():
        raise ValueError("diagnostic libraries do not belong in xvr_amd/lib/: use xvr_amd.build.diagnostic_path(name)")
    out.parent.mkdir(parents=True, exist_ok=True)
    defines = [define] if isinstance(define, str) else list(define)
    def run(cmd):
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout[-4000:]}\n{proc.stderr[-4000:]}")

    if only is None:
        only = [s.name for s in SRC]   # (compile every unit with its own flags, then link)
    else:
        build_library()
    assert any(s.name in only for s in SRC), only
    objs = []
    for s in SRC:
        if s.name in only:   # compile, then link: hipcc takes every input of a mixed command line for HIP source
            obj = out.with_name(out.stem + "_" + s.stem + ".o")
            run([hipcc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(s.name, []), *[f"-D{d}" for d in defines], "-c", str(s), "-o", str(obj)])
            objs.append(obj)
        else:
            objs.append(OBJ / (s.stem + ".o"))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *map(str, objs)])
    return out


class HipccMissing(RuntimeError):
    """The ROCm compiler is not on this machine (as opposed to: it is, and the sources do not compile)."""


def find_hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    return hipcc if Path(hipcc).exists() else None


def is_stale() -> bool:
    if not LIB.exists():
        return True
    built = LIB.stat().st_mtime
    return any(p.stat().st_mtime > built for p in SRC + HDR if p.exists())


def build_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and not is_stale():
        return LIB
    hipcc = find_hipcc()
    if hipcc is None:
        raise HipccMissing("hipcc not found: cannot build libxvr_drr.so (ROCm toolchain required)")
    OBJ.mkdir(parents=True, exist_ok=True)
    # one builder at a time: the ranks of a multi-GPU launch import the package simultaneously, and a stale library
    # must not be rebuilt by eight processes into the same files
    import fcntl

    with open(OBJ / ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale():      # somebody else built it while we waited
            return LIB
        return _build_locked(hipcc, force, verbose)


def _build_locked(hipcc: str, force: bool, verbose: bool) -> Path:
    newest_header = max(p.stat().st_mtime for p in HDR if p.exists())

    def compile_one(src: Path):
        obj = OBJ / (src.stem + ".o")
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, newest_header):
            return obj, None
        cmd = [hipcc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, capture_output=True, text=True)
        return obj, (None if proc.returncode == 0 else f"{src.name}: hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")

    # one translation unit per kernel family: compiled in parallel (a few workers: each clang takes ~1 GB), then linked
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(compile_one, SRC))
    errors = [e for _, e in results if e]
    if errors:
        raise RuntimeError("\n".join(errors))
    tmp = LIB.with_suffix(".so.tmp")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *[str(o) for o, _ in results]]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    tmp.replace(LIB)   # atomically: a process that is loading the old library keeps a complete file
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
