"""ORACLE loader (test infrastructure): ctypes bindings for oracle/drr_scalar.c (float64, scalar).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

from .diffdrr_restated import RenderSpec, index_map

_HERE = Path(__file__).resolve().parent
_SRC = _HERE / "drr_scalar.c"
_OUT = _HERE / "_build" / "libdrr_scalar.so"


class _Spec(ctypes.Structure):
    _fields_ = [
        ("voxel_shift", ctypes.c_double),
        ("eps", ctypes.c_double),
        ("a", ctypes.c_double * 3),
        ("b", ctypes.c_double * 3),
        ("n_points", ctypes.c_int),
        ("near", ctypes.c_double),
        ("far", ctypes.c_double),
        ("denom", ctypes.c_double),
        ("clip", ctypes.c_int),
        ("per_ray_clamp", ctypes.c_int),
    ]


def build(force: bool = False) -> Path:
    """gcc the scalar oracle into oracle/_build/ (git-ignored, travels to the GPU box)."""
    if _OUT.exists() and not force and _OUT.stat().st_mtime >= _SRC.stat().st_mtime:
        return _OUT
    _OUT.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-o", str(_OUT), str(_SRC), "-lm"]
    subprocess.run(cmd, check=True)
    return _OUT


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        _lib = ctypes.CDLL(str(path))
        _lib.oracle_abi_version.restype = ctypes.c_int
        assert _lib.oracle_abi_version() == 1
    return _lib


def _c_spec(shape, spec: RenderSpec) -> _Spec:
    a, b = index_map(shape, spec)
    c = _Spec()
    c.voxel_shift, c.eps = spec.voxel_shift, spec.eps
    for i in range(3):
        c.a[i], c.b[i] = float(a[i]), float(b[i])
    c.n_points, c.near, c.far = spec.n_points, spec.near, spec.far
    c.denom = float(spec.n_points if spec.step_mode == "n_points" else spec.n_points - 1)
    if spec.clip_to_volume == "batch":
        raise NotImplementedError("the scalar oracle has no clip_to_volume = 'batch' (the torch restatement does)")
    c.clip = int(spec.clip_to_volume)
    c.per_ray_clamp = int(spec.per_ray_clamp)
    return c


def _ptr(x, ty):
    return x.ctypes.data_as(ctypes.POINTER(ty))


def render(volume, source, target, img, spec: RenderSpec, mask=None, return_segments=False):
    """float64 scalar render.  Array-likes in the reference layout; returns np.float64 [B,C,n]."""
    vol = np.ascontiguousarray(np.asarray(volume, dtype=np.float32))
    shape = (ctypes.c_int * 3)(*vol.shape)
    src = np.ascontiguousarray(np.asarray(source, dtype=np.float64).reshape(-1, 3))
    tgt = np.ascontiguousarray(np.asarray(target, dtype=np.float64))
    B, n, _ = tgt.shape
    assert src.shape[0] == B
    length = np.ascontiguousarray(np.asarray(img, dtype=np.float64).reshape(B, n))
    if mask is not None:
        msk = np.ascontiguousarray(np.asarray(mask, dtype=np.float32))
        C = int(msk.max()) + 1
        mptr = _ptr(msk, ctypes.c_float)
    else:
        C, mptr = 1, None
    out = np.zeros((B, C, n), dtype=np.float64)
    cs = _c_spec(vol.shape, spec)
    L = lib()
    if spec.renderer == "trilinear":
        rc = L.oracle_trilinear(
            _ptr(vol, ctypes.c_float), shape, _ptr(src, ctypes.c_double), _ptr(tgt, ctypes.c_double),
            _ptr(length, ctypes.c_double), B, n, ctypes.byref(cs), mptr, C, _ptr(out, ctypes.c_double),
        )
        nseg = None
    else:
        nseg_c = ctypes.c_longlong(0)
        rc = L.oracle_siddon(
            _ptr(vol, ctypes.c_float), shape, _ptr(src, ctypes.c_double), _ptr(tgt, ctypes.c_double),
            _ptr(length, ctypes.c_double), B, n, ctypes.byref(cs), mptr, C, _ptr(out, ctypes.c_double),
            ctypes.byref(nseg_c),
        )
        nseg = nseg_c.value
    if rc != 0:
        raise RuntimeError(f"scalar oracle failed with code {rc}")
    return (out, nseg) if return_segments else out
