"""ctypes binding of libxvr_drr.so (include/xvr_drr.h).  No CPU fallback: if the HIP library cannot
be loaded, every render call raises."""

from __future__ import annotations

import ctypes
import os
from pathlib import Path

import warnings

from .build import LIB, HipccMissing, build_library, is_stale

ABI_VERSION = 10
JAC_STRIDE = 8
ALPHA_WINDOW_FLOATS = 16   # XVR_DRR_ALPHA_WINDOW_FLOATS


class CSpec(ctypes.Structure):
    _fields_ = [
        ("a", ctypes.c_float * 3),
        ("b", ctypes.c_float * 3),
        ("lo", ctypes.c_float * 3),
        ("hi", ctypes.c_float * 3),
        ("plane0", ctypes.c_float * 3),
        ("eps", ctypes.c_float),
        ("n_points", ctypes.c_int32),
        ("near_", ctypes.c_float),
        ("far_", ctypes.c_float),
        ("inv_denom", ctypes.c_float),
        ("clip_to_volume", ctypes.c_int32),
        ("ray_grid_w", ctypes.c_int32),
        ("volume_layout", ctypes.c_int32),
        ("alpha_window", ctypes.c_void_p),
    ]


class CSimSpec(ctypes.Structure):
    _fields_ = [
        ("mean", ctypes.c_float),
        ("std", ctypes.c_float),
        ("std_eps", ctypes.c_float),
        ("ncc_eps", ctypes.c_float),
        ("beta", ctypes.c_float),
        ("mncc_patch", ctypes.c_int),
        ("gncc_patch", ctypes.c_int),
        ("per_image", ctypes.c_int),
        ("pre_transformed", ctypes.c_int),
    ]


class CPoseOptSpec(ctypes.Structure):   # include/xvr_pose.h: xvr_pose_opt_spec
    _fields_ = [
        ("axes", ctypes.c_int * 3),
        ("beta1", ctypes.c_float),
        ("beta2", ctypes.c_float),
        ("eps", ctypes.c_float),
        ("maximize", ctypes.c_int),
        ("factor", ctypes.c_float),
        ("patience", ctypes.c_int),
        ("threshold", ctypes.c_double),
        ("lr_eps", ctypes.c_double),
        ("max_n_plateaus", ctypes.c_int),
        ("max_iters", ctypes.c_int),
    ]


class CPoseOptState(ctypes.Structure):  # include/xvr_pose.h: xvr_pose_opt_state (device resident)
    _fields_ = [
        ("m", ctypes.c_float * 13),   # XVR_POSE_MAX_PARAMS
        ("v", ctypes.c_float * 13),
        ("lr", ctypes.c_float * 2),
        ("seen_lr", ctypes.c_float),
        ("step", ctypes.c_int),
        ("n_bad", ctypes.c_int),
        ("n_plateaus", ctypes.c_int),
        ("done", ctypes.c_int),
        ("iter", ctypes.c_int),
        ("best", ctypes.c_double),
    ]


POSE_HISTORY_COLS = 9

_P = ctypes.c_void_p
_I = ctypes.c_int
_AX = ctypes.POINTER(ctypes.c_int)
_FWD = [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, ctypes.POINTER(CSpec), _P, _P, _P, _P]
_BWD = [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, ctypes.POINTER(CSpec), _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]

EXPORTS = {
    "xvr_drr_abi_version": ([], ctypes.c_int),
    "xvr_drr_last_error": ([], ctypes.c_char_p),
    "xvr_drr_set_option": ([ctypes.c_char_p, ctypes.c_int], ctypes.c_int),
    "xvr_drr_get_option": ([ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)], ctypes.c_int),
    "xvr_drr_backward_workspace_bytes": ([_I, _I, _I, _I, _I], ctypes.c_size_t),
    "xvr_drr_siddon_backward_workspace_bytes": ([_I, _I, _I, _I, _I, ctypes.POINTER(CSpec)], ctypes.c_size_t),
    "xvr_drr_trilinear_forward": (_FWD, ctypes.c_int),
    "xvr_drr_trilinear_backward": (_BWD, ctypes.c_int),
    "xvr_drr_siddon_forward": (_FWD, ctypes.c_int),
    "xvr_drr_siddon_backward": (_BWD, ctypes.c_int),
    "xvr_drr_backward_from_jac": ([_P, _P, _I, _I, _P, _P, _P, _P], ctypes.c_int),
    "xvr_drr_alpha_window_bytes": ([_I], ctypes.c_size_t),
    "xvr_drr_alpha_window": ([_P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(CSpec), _P, _P], ctypes.c_int),
    "xvr_drr_alpha_window_backward": ([_P, _P, _P, _P, _P, _I, _I, ctypes.POINTER(CSpec), _P, _P, _P, _P], ctypes.c_int),
    "xvr_sim_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "xvr_sim_ncc_forward_backward": ([_P, _P, _P, _I, _I, _I, ctypes.POINTER(CSimSpec), _P, _P, _P, ctypes.c_size_t, _P], ctypes.c_int),
    "xvr_sim_ncc_registration_step": ([_P, _P, _P, _I, _I, _I, ctypes.POINTER(CSimSpec), _P, _P, _P, ctypes.c_size_t,
                                       _P, _P, _P, ctypes.c_size_t, _P, _P, ctypes.POINTER(CPoseOptSpec), _P, _P, _P, _P, _I, _P], ctypes.c_int),
    "xvr_sim_equalize_workspace_bytes": ([_I, _I], ctypes.c_size_t),
    "xvr_sim_equalize_forward": ([_P, _I, _I, _I, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, ctypes.c_size_t, _P],
                                 ctypes.c_int),
    "xvr_sim_equalize_backward": ([_P, _P, _P, _P, _I, _I, _I, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, ctypes.c_size_t, _P],
                                  ctypes.c_int),
    "xvr_sim_transform_state_bytes": ([_I], ctypes.c_size_t),
    "xvr_sim_transform_forward": ([_P, _I, ctypes.c_longlong, _I, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, _P], ctypes.c_int),
    "xvr_sim_transform_backward": ([_P, _P, _I, ctypes.c_longlong, _I, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, _P], ctypes.c_int),
    "xvr_sim_dice_bool": ([_P, _P, _I, _I, _I, _P, _P], ctypes.c_int),
    "xvr_sim_gaussian_blur5": ([_P, _P, _P, _I, _I, _I, ctypes.c_float, _I, _P], ctypes.c_int),
    "xvr_drr_hu_stats": ([_P, ctypes.c_longlong, _P, _P], ctypes.c_int),
    "xvr_drr_hu_to_density": ([_P, ctypes.c_longlong, _P, ctypes.c_float, _P, _P], ctypes.c_int),
    "xvr_drr_rays_forward": ([_P, _I, _I, _I, _P, _P, _P, _P], ctypes.c_int),
    "xvr_drr_foreground": ([_P, _I, _I, _I, ctypes.c_float, _P, _P, _P, _P, _P], ctypes.c_int),
    "xvr_drr_rays_backward": ([_P, _I, _I, _I, _P, _P, _P, _P, _P], ctypes.c_int),
    "xvr_drr_trilinear_forward_camera": ([_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, ctypes.POINTER(CSpec), _P, _P, _P, _P], ctypes.c_int),
    "xvr_drr_siddon_forward_camera": ([_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, ctypes.POINTER(CSpec), _P, _P, _P, _P], ctypes.c_int),
    "xvr_drr_pack_labels": ([_P, _P, ctypes.c_longlong, _P, _P], ctypes.c_int),
    "xvr_drr_ypairs_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "xvr_drr_pack_ypairs": ([_P, _I, _I, _I, _P, _P], ctypes.c_int),
    "xvr_drr_pack_labels_ypairs": ([_P, _P, _I, _I, _I, _P, _P], ctypes.c_int),
    "xvr_drr_ytiles_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "xvr_drr_pack_ytiles": ([_P, _I, _I, _I, _P, _P], ctypes.c_int),
    "xvr_drr_pack_labels_ytiles": ([_P, _P, _I, _I, _I, _P, _P], ctypes.c_int),
    "xvr_drr_pack_hu_labels_ytiles": ([_P, _P, _P, ctypes.c_float, _I, _I, _I, _P, _P], ctypes.c_int),
    "xvr_drr_bricks_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "xvr_drr_pack_bricks": ([_P, _I, _I, _I, _P, _P], ctypes.c_int),
    "xvr_drr_jac_to_camera_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "xvr_drr_jac_to_camera_backward": ([_P, _P, _P, _I, _I, _I, _P, _P, ctypes.c_size_t, _P], ctypes.c_int),
    "xvr_pose_camera_forward": ([_P, _P, _I, _AX, _P, _P, _P, _P], ctypes.c_int),
    "xvr_pose_camera_backward": ([_P, _P, _I, _AX, _P, _P, _P, _P, _P], ctypes.c_int),
    "xvr_pose_geodesic": ([_P, _P, _I, ctypes.c_float, ctypes.c_float, _P, _P, _P], ctypes.c_int),
    "xvr_pose_convert_jacobian_floats": ([_I], ctypes.c_size_t),
    "xvr_pose_convert_forward": ([_P, _P, _I, _I, _AX, _P, _P, _P], ctypes.c_int),
    "xvr_pose_convert_backward": ([_P, _P, _I, _I, _P, _P, _P], ctypes.c_int),
    "xvr_pose_multiview_forward": ([_P, _P, _I, ctypes.c_float, ctypes.c_float, _P, _P], ctypes.c_int),
    "xvr_pose_multiview_backward": ([_P, _P, _P, _I, ctypes.c_float, ctypes.c_float, _P, _P], ctypes.c_int),
    "xvr_pose_opt_state_bytes": ([], ctypes.c_size_t),
    "xvr_pose_opt_init": ([_P, _I, ctypes.c_float, ctypes.c_float, _P], ctypes.c_int),
    "xvr_pose_opt_step": ([_P, _P, _I, ctypes.POINTER(CPoseOptSpec), _P, _P, _P, _P, _P, _P], ctypes.c_int),
    "xvr_pose_camera_forward_param": ([_P, _P, _I, _I, _AX, _P, _P, _P, _P, _P], ctypes.c_int),
    "xvr_pose_opt_step_param": ([_P, _P, _I, _I, ctypes.POINTER(CPoseOptSpec), _P, _P, _P, _P, _P, _P, _P], ctypes.c_int),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def library_path() -> Path:
    # XVR_DRR_LIBRARY: load another build of the same ABI (the diagnostic builds of tools/gather_stats.py)
    override = os.environ.get("XVR_DRR_LIBRARY")
    return Path(override) if override else LIB


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load (building first if the sources are newer and hipcc is present) libxvr_drr.so."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if path == LIB and build_if_missing and is_stale():
        # A stale library is only ever used when there is NO compiler on this machine (and then loudly): a compile or
        # link error must not leave the binding running over yesterday's kernels.
        try:
            build_library()
        except HipccMissing as e:
            if not LIB.exists():
                raise HipLibraryError(f"libxvr_drr.so is missing and could not be built: {e}") from e
            warnings.warn(f"{LIB} is older than its sources and hipcc is not available to rebuild it: loading the stale "
                          "library", RuntimeWarning, stacklevel=2)
        except RuntimeError as e:
            raise HipLibraryError(f"libxvr_drr.so is stale and its rebuild failed: {e}") from e
    if not path.exists():
        raise HipLibraryError(f"{path} not found; run `python -m xvr_amd.build` (needs hipcc)")
    try:
        lib = ctypes.CDLL(str(path))
    except OSError as e:
        raise HipLibraryError(f"cannot load {path}: {e}") from e
    for name, (argtypes, restype) in EXPORTS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise HipLibraryError(f"{path} does not export {name}")
        fn.argtypes, fn.restype = argtypes, restype
    if lib.xvr_drr_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library {lib.xvr_drr_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().xvr_drr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def set_option(name: str, value: int) -> None:
    """One of the library's A/B switches (include/xvr_drr.h: xvr_drr_set_option).  Process-wide; every setting selects among
    correct kernels.  The XVR_DRR_<NAME> environment variables set the initial values when the library is loaded."""
    check(load().xvr_drr_set_option(name.encode(), int(value)), f"xvr_drr_set_option({name!r}, {value})")


def get_option(name: str) -> int:
    v = ctypes.c_int(0)
    check(load().xvr_drr_get_option(name.encode(), ctypes.byref(v)), f"xvr_drr_get_option({name!r})")
    return v.value


class option:
    """``with _lib.option("gather_splat", 0): ...`` -- the switch is put back on exit."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False
