"""``Siddon`` / ``Trilinear`` renderer modules over the HIP library (the hot call of the path).

Drop-in for the object xvr reaches as ``drr.renderer`` and calls as

    img = self.drr.renderer(tmp, source, target, img, mask=seg)      # -> [B, C, n]

(/root/reference/src/xvr/model/trainer.py:288; constructed from ``renderer="siddon"|"trilinear"``
at /root/reference/src/xvr/renderer/load.py:32-44, ``voxel_shift`` forwarded from
/root/reference/src/xvr/registrar/base.py:61).  Same argument meaning as the reference:
``volume[D0,D1,D2]``, ``source[B,1,3]`` and ``target[B,n,3]`` in voxel-index coordinates,
``img[B,1,n]`` = world-mm ray length; differentiable w.r.t. ``source``/``target``/``img`` (-> pose) and
``volume`` (-> voxels).

There is NO CPU path: tensors must live on the GPU and libxvr_drr.so must load, otherwise the call
raises (python exceptions only -- the reference's trainer catches them per step,
/root/reference/src/xvr/model/trainer.py:171-175).
"""

from __future__ import annotations

import ctypes

import torch

from . import _lib
from .spec import RenderSpec

__all__ = ["Siddon", "Trilinear", "render", "render_from_camera", "make_cspec", "invalidate_volume_cache"]


def make_cspec(shape, spec: RenderSpec, ray_grid_w: int = 0, volume_layout: int = 0) -> _lib.CSpec:
    spec.validate()
    a, b = spec.index_map(shape)
    c = _lib.CSpec()
    for i in range(3):
        c.a[i], c.b[i] = a[i], b[i]
        c.lo[i] = -spec.voxel_shift
        c.hi[i] = shape[i] - spec.voxel_shift
        c.plane0[i] = -spec.voxel_shift
    c.eps = spec.eps
    c.n_points = spec.n_points
    c.near_, c.far_ = spec.near, spec.far
    c.inv_denom = 1.0 / (spec.n_points if spec.step_mode == "n_points" else spec.n_points - 1)
    c.clip_to_volume = 2 if spec.clip_to_volume == "batch" else int(bool(spec.clip_to_volume))   # (2: + alpha_window, set by the caller)
    c.ray_grid_w = int(ray_grid_w)
    c.volume_layout = int(volume_layout)
    return c


# When set to a list, every C-ABI launch is bracketed by HIP events recorded on the launch stream and
# (name, start, end) is appended -- bench.py uses this to time the kernels live (no effect otherwise).
PROFILER = None


def _timed(name, fn, *args):
    if PROFILER is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    PROFILER.append((name, e0, e1))
    return rc


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_gpu_f32(name, t):
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the DRR renderer is a HIP kernel and has no CPU path; move it to the GPU"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype}); xvr casts the DRR module to float32")


# Device scratch for the backward pass (voxel-driven gather: packed rays + per-pose projection
# constants), one buffer per (device, stream), grown on demand and reused across calls.
# VOXEL_GATHER = False withholds it, which forces the atomic scatter fallback (tests, A/B runs).
VOXEL_GATHER = True
_WORKSPACES = {}
# The brick-local splats (k_trilinear_splat_b16 / _px, k_siddon_splat) sum in int32 fixed point inside three quarters of the range;
# a sum found in the guard band at a flush means the bound on a voxel's sum was optimistic: the voxels are poisoned with NaN (loud
# downstream, no host sync) and word 2 of the workspace is set.  XVR_DRR_CHECK_OVERFLOW=1 reads that word after every backward (one
# device -> host sync per step) and raises; last_backward_overflowed() reads it on demand.
CHECK_OVERFLOW = __import__("os").environ.get("XVR_DRR_CHECK_OVERFLOW", "0") == "1"
# (count, hook) | None.  With count > 1 the voxel gradient of a backward is computed in `count` x slabs (whole planes of 16^3 bricks),
# one launch per slab, and hook(i, grad_volume[x0:x1]) is called on the current stream's timeline right after slab i's launch: an
# async collective issued there (torch.distributed orders it behind the stream's work so far) overlaps the remaining slabs
# (bench.py at N > 1, xvr_amd.distributed.SlabAllReduce).  The sums are the same bits as the single launch's.
VOXEL_GRAD_SLABS = None
_LAST_VOL_WORKSPACE = None


def last_backward_overflowed() -> bool:
    """Did the most recent voxel-gradient launch of this process see a fixed-point sum outside its range?  (Synchronises.)"""
    ws = _LAST_VOL_WORKSPACE
    return bool(ws is not None and ws[:4].view(torch.int32)[2].item() != 0)


# Siddon under a non-exact index map gathers per plane cell into octant sums first: 32 bytes per voxel MORE scratch
# (4 GiB at 512^3).  That part is allocated for the one call that uses it and handed back to the caching allocator
# afterwards, never parked in the per-stream cache, and not at all above this cap (the atomic scatter then serves).
SIDDON_CELLS_SCRATCH_CAP = 16 << 30


def _workspace(lib, B, n, shape, device, cspec=None, cells=False):
    if not VOXEL_GATHER:
        return None, 0
    nbytes = lib.xvr_drr_backward_workspace_bytes(B, n, *shape)
    if cells and cspec is not None:
        full = lib.xvr_drr_siddon_backward_workspace_bytes(B, n, *shape, ctypes.byref(cspec))
        if nbytes < full <= nbytes + SIDDON_CELLS_SCRATCH_CAP:
            try:   # transient: lives until the caller drops it (stream-ordered reuse by the caching allocator)
                ws = torch.empty((full + 3) // 4, device=device, dtype=torch.float32)
                return ws, ws.numel() * 4
            except torch.OutOfMemoryError:
                pass   # the smaller scratch below: the dispatcher falls back to the scatter
    key = (device, torch.cuda.current_stream(device).cuda_stream)   # one scratch per stream: concurrent backwards
    ws = _WORKSPACES.get(key)                                        # on two streams must not share it
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _WORKSPACES[key] = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
    return ws, ws.numel() * 4


# Mask -> channels renders with at most 16 labels use a copy of the volume that carries every voxel's label
# in its low 4 mantissa bits (xvr_drr_pack_labels; include/xvr_drr.h): the label lookup then costs no gather.
# The density seen by the render moves by <= 15 ulp (1.8e-6 relative).  False (or XVR_DRR_PACK_LABELS=0)
# keeps the separate lookup in the mask volume.
import os as _os
import weakref

PACK_LABELS = _os.environ.get("XVR_DRR_PACK_LABELS", "1") != "0"


# Render-ready copies of a volume (label-carrying, y-pair interleaved, bricked) are cached per volume TENSOR OBJECT in a
# registry keyed by id() and validated by a weak reference -- never in the tensor's __dict__: copy.deepcopy (what
# Registrar.run does to the DRR on every call, /root/reference/src/xvr/registrar/base.py:161,192) would carry such an
# attribute to the clone, whose fresh version counter can equal a cached key taken from different data, and would
# duplicate a buffer twice the volume's size per clone.  An entry dies with its tensor.  Contents are keyed by the
# tensor's version counter, which in-place torch ops bump.  Writes that do NOT bump it -- through ``volume.data`` or through a raw
# pointer (a custom kernel) -- must be followed by ``invalidate_volume_cache(volume)``.
_VOLUME_CACHE = {}


def _cache_slot(volume):
    key = id(volume)
    slot = _VOLUME_CACHE.get(key)
    if slot is None or slot["ref"]() is not volume:
        def _drop(ref, key=key):
            cur = _VOLUME_CACHE.get(key)
            if cur is not None and cur["ref"] is ref:
                del _VOLUME_CACHE[key]
        slot = _VOLUME_CACHE[key] = {"ref": weakref.ref(volume, _drop)}
    return slot


def invalidate_volume_cache(volume=None):
    """Forget the render-ready copies (y-pair / bricked / label-carrying) of ``volume`` -- of every volume when None.
    Needed only after a write that does not bump the tensor's version counter: ``volume.data.<op>_()``, or a kernel that
    writes through ``data_ptr()``."""
    if volume is None:
        _VOLUME_CACHE.clear()
    else:
        slot = _VOLUME_CACHE.get(id(volume))
        if slot is not None and slot["ref"]() is volume:
            del _VOLUME_CACHE[id(volume)]


def _packed_volume(lib, volume, mask):
    """The label-carrying copy of ``volume``; cached per volume tensor object (an address-keyed cache goes
    stale when the allocator hands the same address to the next step's density)."""
    key = (mask.data_ptr(), mask._version, volume._version)
    slot = _cache_slot(volume)
    hit = slot.get("packed")
    if hit is not None and hit[0] == key and hit[2]() is mask:   # (weak reference: an id() can be recycled)
        return hit[1]
    packed = torch.empty_like(volume)
    rc = _timed("pack_labels", lib.xvr_drr_pack_labels, _ptr(volume), _ptr(mask), volume.numel(), _ptr(packed), _stream())
    _lib.check(rc, "xvr_drr_pack_labels")
    slot["packed"] = (key, packed, weakref.ref(mask))
    return packed


def _packed_ypair_volume(lib, volume, mask, hu_map=None):
    """The y-pair interleaved copy of the label-carrying volume, written in ONE pass over (volume, mask)
    (xvr_drr_pack_labels_ypairs): masked renders of large launches take it at once -- their volume is typically the fresh
    HU -> density map of a training step, rendered exactly twice (trainer.py:185-230), for which the "third render" rule of
    _layout_copy never fires.  0.75 ms at 512^3 against 0.36 for the labels alone; each of the two renders then saves ~1 ms."""
    D0, D1, D2 = volume.shape
    key = (mask.data_ptr(), mask._version, volume._version)
    slot = _cache_slot(volume)
    tiles = YPAIR_TILES and YPAIR_TILES_PACKED
    key = key + (tiles, None if hu_map is None else hu_map.multiplier)
    hit = slot.get("packed_ypairs")
    if hit is not None and hit[0] == key and hit[2]() is mask:
        return hit[1], (3 if tiles else 1)
    nbytes, pack = ((lib.xvr_drr_ytiles_bytes, lib.xvr_drr_pack_labels_ytiles) if tiles
                    else (lib.xvr_drr_ypairs_bytes, lib.xvr_drr_pack_labels_ypairs))
    buf = hit[1] if hit is not None and hit[1].numel() * 4 == nbytes(D0, D1, D2) and hu_map is not None else \
        torch.empty(nbytes(D0, D1, D2) // 4, device=volume.device, dtype=torch.float32)
    if hu_map is not None:   # (`volume` holds HU: the density map rides in the same pass, data.HUDensity; the step's buffer is reused
        #                          IN PLACE -- safe because every render of it is enqueued on the one stream this pass is enqueued on)
        rc = _timed("pack_hu_labels_ytiles", lib.xvr_drr_pack_hu_labels_ytiles, _ptr(volume), _ptr(mask), _ptr(hu_map.stats),
                    ctypes.c_float(hu_map.multiplier), D0, D1, D2, _ptr(buf), _stream())
    else:
        rc = _timed("pack_labels_ypairs", pack, _ptr(volume), _ptr(mask), D0, D1, D2, _ptr(buf), _stream())
    _lib.check(rc, "xvr_drr_pack_labels_ypairs")
    slot["packed_ypairs"] = (key, buf, weakref.ref(mask))
    return buf, (3 if tiles else 1)


# One-channel trilinear renders of LARGE launches march a y-pair interleaved copy of the volume (xvr_drr_pack_ypairs): two
# 16-byte gathers per sample instead of four 8-byte ones -- the march is bound by the texture-address rate per gather
# instruction -- with identical output bits.  Costs twice the volume's memory (cached ON the volume tensor object, keyed by
# its version counter, built the third time a version is rendered: see _ypair_volume).  False (or XVR_DRR_YPAIRS=0):
# natural layout.
YPAIR_LAYOUT = _os.environ.get("XVR_DRR_YPAIRS", "1") != "0"
# ... and that copy is cut into 2 x 8 tiles overlapping along z (xvr_drr_pack_ytiles, volume_layout 3; round 4): the forward is
# bound by fabric bandwidth and a tile's 128 bytes are used two to three times as densely as a z-run's.  XVR_DRR_YTILES=0: rows.
# The label-carrying copy of a training step is packed EVERY step (a fresh HU -> density map, rendered twice).  With 4 x 4 tiles at
# stride 3 the larger write (0.96 against 0.53 ms at 512^3) cost more than the two renders saved; with 2 x 8 tiles at stride 7 it
# is 0.66 ms and the renders save 0.5 (C5: 15.46-15.69 against 15.65-15.81 ms per step): tiled too, unless XVR_DRR_YTILES_PACKED=0.
YPAIR_TILES = _os.environ.get("XVR_DRR_YTILES", "1") != "0"
YPAIR_TILES_PACKED = _os.environ.get("XVR_DRR_YTILES_PACKED", "1") != "0"
YPAIR_MIN_WAVEFRONTS = 2048     # smaller launches take the sample-split kernels on the natural layout
# Siddon's counterpart: 2 x 2 x 8-voxel bricks, one per cache line (xvr_drr_pack_bricks); same caching rule.  XVR_DRR_BRICKS=0: off.
BRICK_LAYOUT = _os.environ.get("XVR_DRR_BRICKS", "1") != "0"
# renders of a volume version BEFORE its copy is built (measured with the volume changing every step, bench.py --update-volume:
# the y-pair copy costs 0.54 ms and saves 0.70 of the forward -- 14.75 -> 14.96 ms per step with the splat behind it, no gain --;
# the bricked copy saves 1.3 ms of the Siddon walk: 21.46 -> 20.51 ms per step when built at first sight)
LAYOUT_COPY_AFTER = {"ypairs": 2, "bricks": 0}
# Round 6: on the TILED y-pair copy (round 4) the forward of a large launch is 1.7 ms faster than on the natural layout (C2:
# 4.76 against 6.48 ms, profiles/r06_trilinear_rocprof_summary.md) and the copy costs 0.6 ms -- the measurement above is of round
# 2's row layout.  A launch whose saving (~0.45 ms per 1e9 nominal samples) exceeds the copy's cost (0.6 ms per 512^3 voxels) builds
# it at FIRST sight, i.e. above ~10 samples per voxel: a volume that changes every step (voxels being optimised) then renders from
# the copy as well -- 14.2 -> ms per step at the benchmark (bench.py's value_volume_changing).
YPAIR_FIRST_SIGHT_SAMPLES_PER_VOXEL = float(_os.environ.get("XVR_DRR_YPAIRS_FIRST_SIGHT", "10"))


def _layout_copy(lib, volume, kind, first_sight=False):
    """The y-pair (``kind`` = "ypairs") or bricked ("bricks") copy of ``volume``, or None the first LAYOUT_COPY_AFTER[kind]
    times a version of it is seen.  The y-pair copy costs 0.54 ms at 512^3 and saves ~0.7 ms per render: it waits for the third
    render, i.e. for a volume that is rendered again and again unchanged (registration, the benchmark, a fixed CT) -- a volume
    that changes between renders (voxels being optimised) stays on the natural layout; the fresh HU -> density map of every
    training step, rendered exactly twice with a mask (trainer.py:185-230), gets labels and y-pairs in one pass at first
    sight (_packed_ypair_volume).  The bricked copy for Siddon saves 1.3 ms per render and is built at first sight."""
    D0, D1, D2 = volume.shape
    key = (volume._version, YPAIR_TILES)
    slot = _cache_slot(volume)
    hit = slot.get(kind)      # (version, copy or None, buffer kept for reuse, renders seen)
    if hit is not None and hit[0] == key and hit[1] is not None:
        return hit[1]
    seen = hit[3] + 1 if hit is not None and hit[0] == key else 1
    buf = hit[2] if hit is not None else None
    if seen <= LAYOUT_COPY_AFTER[kind] and not first_sight:
        slot[kind] = (key, None, buf, seen)
        return None
    if kind == "ypairs":
        nbytes, pack, name = ((lib.xvr_drr_ytiles_bytes, lib.xvr_drr_pack_ytiles, "pack_ypairs") if YPAIR_TILES
                              else (lib.xvr_drr_ypairs_bytes, lib.xvr_drr_pack_ypairs, "pack_ypairs"))
    else:
        nbytes, pack, name = lib.xvr_drr_bricks_bytes, lib.xvr_drr_pack_bricks, "pack_bricks"
    if buf is not None and buf.numel() * 4 != nbytes(D0, D1, D2):
        buf = None
    if buf is None:
        buf = torch.empty(nbytes(D0, D1, D2) // 4, device=volume.device, dtype=torch.float32)
    rc = _timed(name, pack, _ptr(volume), D0, D1, D2, _ptr(buf), _stream())
    _lib.check(rc, f"xvr_drr_{name}")
    slot[kind] = (key, buf, buf, seen)
    return buf


def _ypair_volume(lib, volume, samples=0):
    """-> (copy or None, its volume_layout code: 3 = 2 x 8 tiles, 1 = rows).  ``samples``: nominal samples of the launch asking
    (B n n_points): a launch large enough to pay for the tiled copy gets it at first sight (YPAIR_FIRST_SIGHT_SAMPLES_PER_VOXEL)."""
    first = YPAIR_TILES and samples > YPAIR_FIRST_SIGHT_SAMPLES_PER_VOXEL * volume.numel()
    return _layout_copy(lib, volume, "ypairs", first), (3 if YPAIR_TILES else 1)


def _brick_volume(lib, volume):
    """(2 x 2 x 8-voxel bricks for the Siddon forward, xvr_drr_pack_bricks; cached by the y-pair copy's rule)"""
    return _layout_copy(lib, volume, "bricks")


# ... also for the non-exact index maps that the slab march serves since round 5 (norm_dims_offset = +1, align_corners: maps under
# which the volume's points look up voxels inside it).  XVR_DRR_BRICKS_NX=0: those maps march the natural layout (A/B).
BRICK_NX = _os.environ.get("XVR_DRR_BRICKS_NX", "1") != "0"


def _siddon_map_in_bounds(spec, shape) -> bool:
    """Python mirror of siddon_map_in_bounds (csrc/drr_common.hiph): idx = rint(a x + b) stays in [0, D - 1] on the volume's box."""
    a, b = spec.index_map(shape)
    for k, S in enumerate(shape):
        x0, x1 = -spec.voxel_shift + 1e-4, -spec.voxel_shift + S - 1e-4
        if round(a[k] * x0 + b[k]) < 0 or round(a[k] * x1 + b[k]) > S - 1:
            return False
    return True


def _use_bricks(spec, volume, B, n, C=1):
    """(large one-channel Siddon launches: 10.3 -> 9.7 ms at C3 on the merge walk, 6.9 -> 4.9 on the slab march.  Labels packed into
    the taps measured SLOWER with bricks -- their walk is bound by arithmetic the brick address adds to.)"""
    D0, D1, D2 = volume.shape
    waves = B * ((n + 63) // 64)
    exact = spec.norm_dims_offset == 0 and not spec.align_corners
    if not exact:
        # (the bricked copy serves these maps through the slab march ONLY -- xvr_drr_siddon_forward refuses it otherwise: the
        #  march's own conditions, mirrored; everything else keeps the natural layout and the merge walk)
        if not (BRICK_NX and _lib.get_option("siddon_slab") == 1 and _lib.get_option("fwd_split") in (0, 1)
                and _siddon_map_in_bounds(spec, (D0, D1, D2)) and D1 * D2 < 2 ** 24 and D0 * D1 * D2 < 2 ** 29
                and (D0 + D1 + D2 + 6) * 4 <= 48 * 1024 and waves > YPAIR_MIN_WAVEFRONTS):
            return False
    return (BRICK_LAYOUT and spec.renderer == "siddon" and C == 1 and waves >= YPAIR_MIN_WAVEFRONTS
            and ((D0 + 1) // 2) * ((D1 + 1) // 2) * ((D2 + 7) // 8) * 32 < 2 ** 31 and min(D0, D1, D2) >= 2)


def _use_ypairs(spec, volume, B, n):
    """(one channel, or labels packed into the volume's mantissa bits -- never with a separate mask volume)"""
    D0, D1, D2 = volume.shape
    elements = ((D0 + 1) // 2) * (D1 + 1) * ((D2 - 2) // 7 + 1) * 32 if YPAIR_TILES else D0 * (D1 + 1) * D2 * 2
    # (tiles: the kernel's z / 7 is a multiply-shift that holds below 8192 -- xvr_drr_trilinear_forward and xvr_drr_pack_ytiles
    #  refuse longer volumes, which therefore stay on the natural layout instead of raising at render time)
    return (YPAIR_LAYOUT and spec.renderer == "trilinear" and B * ((n + 63) // 64) >= YPAIR_MIN_WAVEFRONTS
            and elements < 2 ** 31 and min(D0, D1, D2) >= 2 and (D2 < 8192 or not YPAIR_TILES))


class _Render(torch.autograd.Function):
    """forward: one fused sweep (with the per-ray jacobian when a pose gradient may be needed);
    backward: elementwise-from-jacobian for the pose, a re-march with scatter for the voxels."""

    @staticmethod
    def forward(ctx, volume, source, target, img, mask, spec: RenderSpec, ray_grid_w: int, C: int, work, hu_map=None):
        lib = _lib.load()
        D0, D1, D2 = volume.shape
        B, n, _ = target.shape
        vol_c = volume.contiguous()
        src_c = source.reshape(B, 3).contiguous()
        tgt_c = target.contiguous()
        len_c = img.reshape(B, n).contiguous()
        msk_c = mask.contiguous() if mask is not None else None
        need_pose = any(ctx.needs_input_grad[1:4])
        use_jac = need_pose  # with a mask: the jacobian of the channel sum (see backward)
        out = torch.empty(B, C, n, device=volume.device, dtype=torch.float32)
        jac = torch.empty(B, n, _lib.JAC_STRIDE, device=volume.device, dtype=torch.float32) if use_jac else None
        fn = lib.xvr_drr_trilinear_forward if spec.renderer == "trilinear" else lib.xvr_drr_siddon_forward
        vol_f, msk_f = vol_c, msk_c
        pairs, pairs_layout = None, 1
        if msk_c is not None and PACK_LABELS and 2 <= C <= 16 and vol_c.data_ptr() % 16 == 0 and msk_c.data_ptr() % 16 == 0:
            if _use_ypairs(spec, vol_c, B, n):
                (pairs, pairs_layout), msk_f = _packed_ypair_volume(lib, vol_c, msk_c, hu_map), None   # labels in the taps AND the y-pair layout, one pass
            else:
                vol_f, msk_f = _packed_volume(lib, vol_c, msk_c), None     # labels ride in the taps
        if pairs is None and msk_f is None and _use_ypairs(spec, vol_c, B, n):
            pairs, pairs_layout = _ypair_volume(lib, vol_f, B * n * spec.n_points)
        bricks = _brick_volume(lib, vol_f) if msk_f is None and _use_bricks(spec, vol_c, B, n, C) else None
        if pairs is not None:
            vol_f = pairs                                              # (of the label-carrying copy when there is one)
        if bricks is not None:
            vol_f = bricks
        cs = make_cspec((D0, D1, D2), spec, ray_grid_w, volume_layout=pairs_layout if pairs is not None else (2 if bricks is not None else 0))
        window = None
        if spec.renderer == "trilinear" and spec.clip_to_volume == "batch":
            # ONE alpha window for the whole call, reduced on the device from its rays (no host round trip): the kernels read
            # near / far / scale from this buffer
            window = torch.empty(lib.xvr_drr_alpha_window_bytes(B) // 4, device=volume.device, dtype=torch.float32)
            rc = _timed("alpha_window", lib.xvr_drr_alpha_window, _ptr(src_c), _ptr(tgt_c), B, n, D0, D1, D2, ctypes.byref(cs),
                        _ptr(window), _stream())
            _lib.check(rc, "xvr_drr_alpha_window")
            cs.alpha_window = window.data_ptr()
        elif spec.renderer == "siddon":
            cs.clip_to_volume = 0
        rc = _timed(f"{spec.renderer}_forward" + ("+jac" if use_jac else ""), fn,
                    _ptr(vol_f), _ptr(msk_f), D0, D1, D2, C, _ptr(src_c), _ptr(tgt_c), _ptr(len_c), B, n,
                    ctypes.byref(cs), _ptr(out), _ptr(jac), _ptr(work), _stream())
        _lib.check(rc, f"xvr_drr_{spec.renderer}_forward")
        if hu_map is not None and pairs is None:
            raise RuntimeError("internal: a lazy HU density reached a render that does not pack its own copy")
        ctx.spec, ctx.ray_grid_w, ctx.C = spec, ray_grid_w, C
        ctx.src_shape, ctx.img_shape = source.shape, img.shape
        ctx.window, ctx.hu_map = window, hu_map
        ctx.save_for_backward(vol_c, src_c, tgt_c, len_c, msk_c, jac)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        vol_c, src_c, tgt_c, len_c, msk_c, jac = ctx.saved_tensors
        spec, C = ctx.spec, ctx.C
        D0, D1, D2 = vol_c.shape
        B, n, _ = tgt_c.shape
        dev = vol_c.device
        # With several channels the saved jacobian is that of their sum: it is the whole pose gradient
        # iff grad_out is the same for every channel.  Autograd tells us for free: the backward of
        # `img.sum(dim=1)` (all xvr does with the channels, trainer.py:292-293) is an EXPANDED tensor
        # (stride 0 along C).
        uniform = C == 1 or gout.stride(1) == 0
        g_uniform = gout[:, 0].contiguous() if uniform else None
        gout = gout.contiguous()
        need_vol = ctx.needs_input_grad[0]
        need_pose = any(ctx.needs_input_grad[1:4])
        gvol = torch.zeros_like(vol_c) if need_vol else None
        gsrc = gtgt = glen = None
        if need_pose:
            gsrc = torch.zeros(B, 3, device=dev, dtype=torch.float32)
            gtgt = torch.empty(B, n, 3, device=dev, dtype=torch.float32)
            glen = torch.empty(B, n, device=dev, dtype=torch.float32)
        from_jac = need_pose and jac is not None and uniform
        window = ctx.window
        if window is not None and ((need_pose and not from_jac) or (need_vol and msk_c is not None and not uniform)):
            raise NotImplementedError("clip_to_volume='batch': gradients are implemented for one channel (or a gradient that is "
                                      "the same for every channel)")
        if from_jac:
            rc = _timed("backward_from_jac", lib.xvr_drr_backward_from_jac,
                        _ptr(jac), _ptr(g_uniform), B, n, _ptr(gsrc), _ptr(gtgt), _ptr(glen), _stream())
            _lib.check(rc, "xvr_drr_backward_from_jac")
            if window is not None:   # the window itself depends on the pose: min / max route their gradient to two rays
                cw = make_cspec((D0, D1, D2), spec, ctx.ray_grid_w)
                cw.alpha_window = window.data_ptr()
                rc = _timed("alpha_window_backward", lib.xvr_drr_alpha_window_backward, _ptr(jac), _ptr(g_uniform), _ptr(src_c),
                            _ptr(tgt_c), _ptr(len_c), B, n, ctypes.byref(cw), _ptr(window), _ptr(gsrc), _ptr(gtgt), _stream())
                _lib.check(rc, "xvr_drr_alpha_window_backward")
        if need_vol or (need_pose and not from_jac):
            if ctx.hu_map is not None:   # (the forward rendered from HU through its packing pass; a backward kernel that reads the
                vol_c = ctx.hu_map.materialize()   # volume -- the re-march -- gets the density written after all)
            cs = make_cspec((D0, D1, D2), spec, ctx.ray_grid_w)
            if window is not None:
                cs.alpha_window = window.data_ptr()
            elif spec.renderer == "siddon":
                cs.clip_to_volume = 0
            fn = lib.xvr_drr_trilinear_backward if spec.renderer == "trilinear" else lib.xvr_drr_siddon_backward
            pose_here = need_pose and not from_jac
            ws, ws_bytes = None, 0
            tag = ("pose" if pose_here else "") + ("+vol" if need_vol else "")
            if msk_c is not None and uniform and not pose_here:
                # every sample lands in exactly one channel, so a gradient that is the same for all channels (the backward
                # of xvr's `img.sum(dim=1)`) reaches the voxels as if there were no mask: the plain one-channel gather
                msk_c, C, gout = None, 1, g_uniform
            if need_vol:
                # (the per-cell scratch only when the cells gather will really run: Siddon, no mask left, rays on a lattice)
                ws, ws_bytes = _workspace(lib, B, n, (D0, D1, D2), dev, cs,
                                          cells=spec.renderer == "siddon" and msk_c is None and ctx.ray_grid_w > 1)
            def call():
                return fn(_ptr(vol_c), _ptr(msk_c), D0, D1, D2, C, _ptr(src_c), _ptr(tgt_c), _ptr(len_c), B, n,
                          ctypes.byref(cs), _ptr(gout), _ptr(gvol),
                          _ptr(gsrc) if pose_here else None, _ptr(gtgt) if pose_here else None,
                          _ptr(glen) if pose_here else None, _ptr(ws), ws_bytes, _stream())

            def call_in_slabs(count, hook):
                # the voxel gradient in x slabs of whole brick planes, one call per slab (option gather_slab): slab i is complete
                # when call i has been issued, and the hook may hand it to a collective while call i + 1 runs
                nb0 = (D0 + 15) // 16
                try:
                    for i in range(count):
                        _lib.set_option("gather_slab", i | (count << 8))
                        rc = call()
                        if rc:
                            return rc
                        x0, x1 = min(16 * (i * nb0 // count), D0), min(16 * ((i + 1) * nb0 // count), D0)
                        if hook is not None and x1 > x0:
                            hook(i, gvol[x0:x1])
                finally:
                    _lib.set_option("gather_slab", 0)
                return 0

            slabs = VOXEL_GRAD_SLABS if (need_vol and ws is not None and gvol.dim() == 3) else None
            name = f"{spec.renderer}_backward[{tag.strip('+')}]"    # (one timed region per step, slabs or not)
            rc = _timed(name, call) if slabs is None or slabs[0] <= 1 else _timed(name, call_in_slabs, *slabs)
            _lib.check(rc, f"xvr_drr_{spec.renderer}_backward")
            if need_vol and ws is not None:
                global _LAST_VOL_WORKSPACE
                _LAST_VOL_WORKSPACE = ws
                if CHECK_OVERFLOW and last_backward_overflowed():
                    raise RuntimeError("voxel gradient: a fixed-point sum of the brick-local splat left its range (the bound on a voxel's "
                                       "sum was optimistic); the affected voxels are NaN.  Option gather_splat = 0 / siddon_splat = 0 "
                                       "selects the fp32 gathers.")
        g_source = gsrc.reshape(ctx.src_shape) if need_pose and ctx.needs_input_grad[1] else None
        g_target = gtgt if need_pose and ctx.needs_input_grad[2] else None
        g_img = glen.reshape(ctx.img_shape) if need_pose and ctx.needs_input_grad[3] else None
        return gvol, g_source, g_target, g_img, None, None, None, None, None, None


class _RenderFromCamera(torch.autograd.Function):
    """cam [B,24] -> DRRs [B,1,H*W] with the rays generated inside the render kernel; backward =
    jacobian -> camera in one fixed-order kernel (xvr_drr_jac_to_camera_backward).  Pose gradient only,
    one channel: what the reference's registration loop differentiates."""

    @staticmethod
    def forward(ctx, cam, volume, spec: RenderSpec, H: int, W: int):
        lib = _lib.load()
        cam_c, vol_c = cam.contiguous(), volume.contiguous()
        B, n = cam_c.shape[0], H * W
        pairs, layout = _ypair_volume(lib, vol_c, B * n * spec.n_points) if _use_ypairs(spec, vol_c, B, n) else (None, 0)
        if pairs is None and _use_bricks(spec, vol_c, B, n):
            pairs, layout = _brick_volume(lib, vol_c), 2               # (siddon: the bricked copy takes the same seat)
        if pairs is None:
            layout = 0
        cs = make_cspec(tuple(vol_c.shape), spec, W, volume_layout=layout)
        need = ctx.needs_input_grad[0]
        out = torch.empty(B, 1, n, device=cam.device, dtype=torch.float32)
        jac = torch.empty(B, n, _lib.JAC_STRIDE, device=cam.device, dtype=torch.float32) if need else None
        fn = lib.xvr_drr_trilinear_forward_camera if spec.renderer == "trilinear" else lib.xvr_drr_siddon_forward_camera
        rc = _timed(f"{spec.renderer}_forward" + ("+jac" if need else ""), fn, _ptr(pairs if pairs is not None else vol_c), None,
                    *vol_c.shape, 1, _ptr(cam_c),
                    B, H, W, ctypes.byref(cs), _ptr(out), _ptr(jac), None, _stream())
        _lib.check(rc, f"xvr_drr_{spec.renderer}_forward_camera")
        ctx.save_for_backward(cam_c, jac)
        ctx.hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        cam_c, jac = ctx.saved_tensors
        H, W = ctx.hw
        B = cam_c.shape[0]
        g_cam = torch.empty_like(cam_c)
        nbytes = lib.xvr_drr_jac_to_camera_workspace_bytes(B, H, W)
        key = ("j2c", cam_c.device, torch.cuda.current_stream(cam_c.device).cuda_stream)
        ws = _WORKSPACES.get(key)
        if ws is None or ws.numel() * 4 < nbytes:   # zero-filled once; every call leaves it ready for the next
            ws = _WORKSPACES[key] = torch.zeros((nbytes + 3) // 4, device=cam_c.device, dtype=torch.float32)
        rc = _timed("jac_to_camera_backward", lib.xvr_drr_jac_to_camera_backward, _ptr(jac), _ptr(gout.contiguous()), _ptr(cam_c),
                    B, H, W, _ptr(g_cam), _ptr(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "xvr_drr_jac_to_camera_backward")
        return g_cam, None, None, None, None


def render_from_camera(volume, cam, spec: RenderSpec, height: int, width: int):
    """One-channel render straight from the camera vector of ``DRR.camera`` / ``pose_camera`` ([B,24]); the
    volume is treated as a constant (no voxel gradient on this path)."""
    _check_gpu_f32("volume", volume)
    _check_gpu_f32("cam", cam)
    if volume.dim() != 3 or cam.dim() != 2 or cam.shape[1] != 24:
        raise ValueError("volume must be [D0, D1, D2] and cam [B, 24]")
    if cam.shape[0] == 0:
        return (cam.sum() * 0).expand(0, 1, height * width)
    return _RenderFromCamera.apply(cam, volume, spec, int(height), int(width))


def render(volume, source, target, img, spec: RenderSpec, mask=None, ray_grid_w: int = 0, n_channels=None, work=None):
    """Functional form.  ``work``: optional cuda uint64/int64 scalar the kernel adds its count of
    volume-touching samples (trilinear) or voxel segments (siddon) to."""
    hu_map = None
    if type(volume).__name__ == "HUDensity":   # (xvr_amd.data.HUDensity: a density that has not been written yet)
        volume.check_fresh()
        hu_map, volume = volume, volume.hu
    for name, t in (("volume", volume), ("source", source), ("target", target), ("img", img)):
        _check_gpu_f32(name, t)
    if volume.dim() != 3:
        raise ValueError(f"volume must be [D0, D1, D2], got {tuple(volume.shape)}")
    if target.dim() != 3 or target.shape[-1] != 3:
        raise ValueError(f"target must be [B, n, 3], got {tuple(target.shape)}")
    B, n, _ = target.shape
    if source.numel() != B * 3:
        raise ValueError(f"source must be [B, 1, 3] (one X-ray source per pose), got {tuple(source.shape)}")
    if img.numel() != B * n:
        raise ValueError(f"img must be [B, 1, n] ray lengths, got {tuple(img.shape)}")
    if mask is not None:
        _check_gpu_f32("mask", mask)
        if mask.shape != volume.shape:
            raise ValueError("mask and volume must have the same shape")
        C = int(n_channels) if n_channels is not None else int(mask.max().item()) + 1
    else:
        C = 1
    if ray_grid_w and n % ray_grid_w:
        ray_grid_w = 0
    if B == 0 or n == 0:
        # an empty batch (xvr's `img[keep]` can select nothing, trainer.py:202-204) renders to an empty
        # image that still hangs off the inputs' autograd graph
        return (source.sum() + target.sum() + img.sum() + 0 * volume.sum()).expand(B, C, n)
    if hu_map is not None:
        # the one consumer that takes the HU map inside its own pass: a masked trilinear launch large enough for the tiled,
        # label-carrying y-pair copy, with nothing to differentiate w.r.t. the voxels; everything else gets the density written
        if not (mask is not None and _packed_tiles_ok(spec, volume, mask, B, n, C) and YPAIR_TILES and YPAIR_TILES_PACKED):
            hu_map, volume = None, hu_map.materialize()
    return _Render.apply(volume, source, target, img, mask, spec, int(ray_grid_w), C, work, hu_map)


def _packed_tiles_ok(spec, volume, mask, B, n, C) -> bool:
    """Will _Render.forward render this masked launch from the label-carrying y-pair copy (_packed_ypair_volume)?"""
    return (PACK_LABELS and 2 <= C <= 16 and volume.is_contiguous() and mask.is_contiguous() and volume.data_ptr() % 16 == 0
            and mask.data_ptr() % 16 == 0 and _use_ypairs(spec, volume, B, n))


class _RendererBase(torch.nn.Module):
    renderer_name = ""

    def __init__(self, voxel_shift: float = 0.5, eps: float = 1e-8,
                 filter_intersections_outside_volume: bool = True, **spec_overrides):
        super().__init__()
        self.voxel_shift = voxel_shift
        self.eps = eps
        self.filter_intersections_outside_volume = filter_intersections_outside_volume
        self.spec_overrides = dict(spec_overrides)
        # (H, W) of the detector whose rays this module is fed, set by DRR; lets the kernels map
        # wavefronts to 8x8 pixel tiles.  Purely a launch-shaping hint.
        self.ray_grid = None
        self._label_cache = (None, None, None)

    def _spec(self, **kw) -> RenderSpec:
        base = dict(renderer=self.renderer_name, voxel_shift=self.voxel_shift, eps=self.eps,
                    filter_intersections_outside_volume=self.filter_intersections_outside_volume)
        base.update(self.spec_overrides)
        base.update(kw)
        return RenderSpec(**base)

    def _n_channels(self, mask):
        if mask is None:
            return None
        key = (mask.data_ptr(), mask._version)
        if self._label_cache[:2] != key:
            self._label_cache = (*key, int(mask.max().item()) + 1)
        return self._label_cache[2]

    def _grid_w(self, n):
        if self.ray_grid is not None and self.ray_grid[0] * self.ray_grid[1] == n:
            return int(self.ray_grid[1])
        return 0


class Trilinear(_RendererBase):
    """Trilinear ray-marching: ``n_points`` samples at ``linspace(near, far)`` along source->target."""

    renderer_name = "trilinear"

    def __init__(self, near: float = 0.0, far: float = 1.0, mode: str = "bilinear",
                 filter_intersections_outside_volume: bool = True, voxel_shift: float = 0.5,
                 eps: float = 1e-8, **spec_overrides):
        if mode != "bilinear":
            raise NotImplementedError("Trilinear supports mode='bilinear' only")
        super().__init__(voxel_shift, eps, filter_intersections_outside_volume, **spec_overrides)
        self.near, self.far, self.mode = near, far, mode

    def make_spec(self, n_points: int = 500, align_corners: bool = False) -> RenderSpec:
        return self._spec(near=self.near, far=self.far, n_points=n_points, align_corners=align_corners)

    def forward(self, volume, source, target, img, n_points: int = 500, align_corners: bool = False, mask=None):
        spec = self.make_spec(n_points, align_corners)
        return render(volume, source, target, img, spec, mask, self._grid_w(target.shape[1]), self._n_channels(mask))


class Siddon(_RendererBase):
    """Siddon's exact ray tracing (nearest-voxel lookup per plane-to-plane segment)."""

    renderer_name = "siddon"

    def __init__(self, mode: str = "nearest", stop_gradients_through_grid_sample: bool = False,
                 filter_intersections_outside_volume: bool = True, voxel_shift: float = 0.5,
                 eps: float = 1e-8, **spec_overrides):
        if mode != "nearest":
            raise NotImplementedError("Siddon supports mode='nearest' only")
        super().__init__(voxel_shift, eps, filter_intersections_outside_volume, **spec_overrides)
        self.mode = mode
        self.stop_gradients_through_grid_sample = stop_gradients_through_grid_sample

    def make_spec(self, align_corners: bool = False) -> RenderSpec:
        return self._spec(align_corners=align_corners)

    def forward(self, volume, source, target, img, align_corners: bool = False, mask=None):
        spec = self.make_spec(align_corners)
        if self.stop_gradients_through_grid_sample:
            volume = volume.detach()
        return render(volume, source, target, img, spec, mask, self._grid_w(target.shape[1]), self._n_channels(mask))
