"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in stock torch ops, of the DRR render algorithm that xvr calls through
``diffdrr==0.6.0``.  ** PARITY UNPINNED **: the arithmetic lives in a third-party dependency that is
pinned in the reference (``/root/reference/uv.lock:955-977``, requirement ``pyproject.toml:14``) but
is not vendored under /root/reference, is not installed here and cannot be fetched (no network), and
the reference ships no tests, golden vectors or fixtures for this path.
What is restated here is therefore the *published* DiffDRR algorithm (Gopalakrishnan & Golland,
"Fast auto-differentiable digitally reconstructed radiographs ...", and Siddon 1985), anchored on the
reference's own call sites:

* ``drr.detector(pose, None)``, ray length, ``affinv(source/target)``, ``drr.renderer(vol, source,
  target, img, mask=seg)``, ``drr.reshape_transform``  -- src/xvr/model/trainer.py:279-304
* ``DRR(subject, sdd, height, delx, width, dely, x0, y0, reverse_x_axis=, renderer=, voxel_shift=)``
  -- src/xvr/renderer/load.py:32-44, src/xvr/registrar/base.py:61
* ``Registration.forward -> DRR.forward(pose)`` -- src/xvr/registrar/base.py:249

Every constant whose value could not be pinned is an explicit
parameter of :class:`RenderSpec`; the HIP kernels take the same parameters, so pinning against the
real package later needs no kernel change.

The op sequence (linspace / sort / diff / ``grid_sample`` / ``scatter_add_``) is what the reference
executes on CPU when its DRR module is moved ``.to("cpu")``, so this file is also the "port" that
``bench.py`` times for the ``cpu_baseline`` leg.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""

from __future__ import annotations

from dataclasses import dataclass, replace

import torch
from torch.nn.functional import grid_sample

ORACLE_PARITY = "unpinned"  # see module docstring


@dataclass(frozen=True)
class RenderSpec:
    """Every numerical convention of the renderers, explicit (SURVEY.md Appendix A ambiguity set)."""

    renderer: str = "trilinear"  # "trilinear" | "siddon"
    # A4: integer coordinate i is the voxel's corner (0.0) or its centre (0.5).
    # planes of voxel i: [i - voxel_shift, i + 1 - voxel_shift]
    voxel_shift: float = 0.5
    # A3: added to (target - source) before both the alpha and the xyz computations.
    eps: float = 1e-8
    # A3': False = eps guards the alpha divisions only, the sample points are s + alpha (t - s).  The two differ by
    # alpha * eps <= 1e-8 voxels -- below fp32 resolution of any coordinate >= 0.1, so ONE kernel serves both
    # (tests/test_oracle.py bounds the difference); the knob exists so that the pin grid can name the upstream form.
    eps_in_xyz: bool = True
    # A6: grid_sample's align_corners.
    align_corners: bool = False
    # A5: xyz is normalised with dims = shape + norm_dims_offset before grid_sample.  0 makes
    # grid_sample's un-normalisation the exact inverse (index = x + voxel_shift - 0.5); the
    # recalled diffdrr values are +1 (siddon) / -1 (trilinear).
    norm_dims_offset: int = 0
    # --- trilinear only ---
    n_points: int = 500  # A1
    near: float = 0.0
    far: float = 1.0
    # A2: "n_points" -> out = L * sum / n_points ; "n_minus_1" -> out = L * sum / (n_points - 1)
    step_mode: str = "n_points"
    # False: alphas = linspace(near, far) over the whole source->target segment, shared by all rays
    #        (samples outside the volume read zeros).  True: alphas rescaled per ray to
    #        [alphamin, alphamax] and the sum additionally scaled by (alphamax - alphamin).
    # "batch": ONE window for the whole call -- alphas = A + linspace * (Z - A) with A = min over rays of alphamin and
    #        Z = max over rays of alphamax (rays that miss the volume aside), shared by all rays, the sum scaled by (Z - A).
    #        A third plausible upstream form of the same call (SURVEY App. A marks the alpha rule "uncertain"); A and Z
    #        are differentiable (min / max route the gradient to the two extremal rays), as torch autograd would have them.
    clip_to_volume: bool | str = False
    # --- siddon only ---
    # True: each ray integrates only over its own [max(alphamin,0), min(alphamax,1)], crossings
    # clipped into that interval (what a per-ray traversal does).  False: literal sort formulation, whose batch-wide column filter
    # lets a ray whose source/target lies INSIDE the volume also count segments with alpha<0 / >1.
    # Both agree whenever source and target are outside the volume (every xvr geometry).
    per_ray_clamp: bool = True
    filter_intersections_outside_volume: bool = True

    def with_(self, **kw) -> "RenderSpec":
        return replace(self, **kw)


# --------------------------------------------------------------------------------------
# geometry helpers shared by both renderers
# --------------------------------------------------------------------------------------
def index_map(shape, spec: RenderSpec, dtype=torch.float64):
    """Per-axis (a, b) such that the un-normalised grid_sample index is ``a * x + b``.

    grid_sample un-normalises u in [-1, 1] as ((u + 1) * S - 1) / 2 (align_corners=False) or
    (u + 1) / 2 * (S - 1) (True); the renderers feed u = 2 (x + shift) / dims - 1.
    """
    S = torch.tensor(list(shape), dtype=dtype)
    dims = S + spec.norm_dims_offset
    if spec.align_corners:
        a = (S - 1) / dims
        b = spec.voxel_shift * (S - 1) / dims
    else:
        a = S / dims
        b = spec.voxel_shift * S / dims - 0.5
    return a, b


def _alpha_minmax(source, target, shape, spec: RenderSpec):
    """First/last intersection of each ray with the volume's bounding planes, clamped to [0, 1]."""
    sdd = target - source + spec.eps
    lo = torch.zeros(3).to(source) - spec.voxel_shift
    hi = torch.tensor(list(shape)).to(source) - spec.voxel_shift
    alpha0 = (lo - source) / sdd
    alpha1 = (hi - source) / sdd
    alphas = torch.stack([alpha0, alpha1])
    alphamin = alphas.min(dim=0).values.max(dim=-1).values.unsqueeze(-1)
    alphamax = alphas.max(dim=0).values.min(dim=-1).values.unsqueeze(-1)
    alphamin = torch.where(alphamin < 0.0, torch.zeros_like(alphamin), alphamin)
    alphamax = torch.where(alphamax > 1.0, torch.ones_like(alphamax), alphamax)
    return alphamin, alphamax


def _filter_columns(alphas, alphamin, alphamax):
    """Drop the alpha columns that are outside the volume for ALL rays (memory optimisation)."""
    good = torch.logical_and(alphas >= alphamin, alphas <= alphamax)
    keep = good.any(dim=0).any(dim=0) if good.dim() == 3 else good
    return alphas[..., keep]


def _xyzs(alpha, source, target, shape, spec: RenderSpec):
    """Points at parameter alpha on each ray, normalised to grid_sample's [-1, 1]^3."""
    xyz = (source.unsqueeze(-2) + alpha.unsqueeze(-1) * (target - source + (spec.eps if spec.eps_in_xyz else 0.0)).unsqueeze(2)).unsqueeze(1)
    dims = torch.tensor(list(shape)).to(source) + spec.norm_dims_offset
    return 2 * (xyz + spec.voxel_shift) / dims - 1


def _lookup(volume, xyzs, mode, align_corners):
    """grid_sample wrapper: coordinate axis i of xyzs indexes volume axis i."""
    B = len(xyzs)
    out = grid_sample(
        input=volume.permute(2, 1, 0)[None, None].expand(B, -1, -1, -1, -1),
        grid=xyzs,
        mode=mode,
        padding_mode="zeros",
        align_corners=align_corners,
    )
    return out[:, 0, 0]  # [B, n, K]


def _to_channels(samples, volume, mask, xyzs, spec):
    """mask=None -> [B,1,n]; else scatter every sample into the channel of its (nearest) label."""
    if mask is None:
        return samples.sum(dim=-1).unsqueeze(1)
    B, n, _ = samples.shape
    C = int(mask.max().item() + 1)
    channels = _lookup(mask, xyzs, "nearest", spec.align_corners).long()
    out = torch.zeros(B, C, n, dtype=samples.dtype, device=samples.device)
    return out.scatter_add_(1, channels.transpose(-1, -2), samples.transpose(-1, -2))


# --------------------------------------------------------------------------------------
# renderers
# --------------------------------------------------------------------------------------
def batch_window(source, target, shape, spec: RenderSpec):
    """(A, Z): the smallest alphamin and the largest alphamax over every ray of the call that meets the volume
    ((0, 0) when none does: the image is then zero)."""
    alphamin, alphamax = _alpha_minmax(source, target, shape, spec)
    hit = alphamax > alphamin
    if not bool(hit.any()):
        z = (source.sum() + target.sum()) * 0
        return z, z
    return alphamin[hit].min(), alphamax[hit].max()


def trilinear(volume, source, target, img, spec: RenderSpec, mask=None, window=None, label_nudge=None):
    """Trilinear ray-marching.  volume[D0,D1,D2]; source[B,1,3]; target[B,n,3]; img[B,1,n] -> [B,C,n].
    ``window``: the (A, Z) of ``clip_to_volume="batch"`` when the caller has computed it over MORE rays than it passes here
    (``render`` chunks over rays; the window belongs to the whole call).
    ``label_nudge`` = (p0, p1), each broadcastable to [B, n, 3], in voxel-space units: the LABEL of a ray's first sample is looked
    up at x_0 + p0 and that of its last sample at x_{N-1} + p1 (the interpolated values stay where they are).  Test
    infrastructure for ``clip_to_volume=True`` with a mask: the first and last sample then sit exactly on a face of the volume,
    where the nearest-label lookup is decided by the last bit of the position -- a tie no two implementations break alike.  A
    nudge of a thousandth of a voxel along the face's normal, into or out of the volume, names the side explicitly
    (tests/conftest.py::resolve_face_ties)."""
    shape = volume.shape
    N = spec.n_points
    alphas = torch.linspace(spec.near, spec.far, N)[None, None].to(volume)
    alphamin, alphamax = _alpha_minmax(source, target, shape, spec)
    if spec.clip_to_volume == "batch":
        A, Z = window if window is not None else batch_window(source, target, shape, spec)
        alphas = A + alphas * (Z - A)                       # [1, 1, N], shared by all rays
    elif spec.clip_to_volume:
        alphas = alphamin + alphas * (alphamax - alphamin)  # [B, n, N]
    elif spec.filter_intersections_outside_volume:
        # numerically a no-op (dropped columns only ever sample zero padding) -- widen the window by
        # the one-voxel interpolation margin so that stays exactly true.
        sdd = (target - source + spec.eps).abs().amin(dim=-1, keepdim=True).clamp_min(1e-6)
        margin = 2.0 / sdd
        keep = torch.logical_and(alphas >= alphamin - margin, alphas <= alphamax + margin)
        alphas = alphas[..., keep.any(dim=0).any(dim=0)]
    xyzs = _xyzs(alphas, source, target, shape, spec)
    samples = _lookup(volume, xyzs, "bilinear", spec.align_corners)
    label_xyzs = xyzs
    if label_nudge is not None and mask is not None:
        dims = torch.tensor(list(shape)).to(source) + spec.norm_dims_offset
        label_xyzs = xyzs.detach().expand(source.shape[0], 1, target.shape[1], xyzs.shape[-2], 3).clone()
        label_xyzs[:, 0, :, 0] += 2 * torch.as_tensor(label_nudge[0]).to(source) / dims
        if label_xyzs.shape[-2] > 1:
            label_xyzs[:, 0, :, -1] += 2 * torch.as_tensor(label_nudge[1]).to(source) / dims
    out = _to_channels(samples, volume, mask, label_xyzs, spec)
    denom = N if spec.step_mode == "n_points" else N - 1
    scale = img / denom
    if spec.clip_to_volume == "batch":
        scale = scale * (Z - A)
    elif spec.clip_to_volume:
        scale = scale * (alphamax - alphamin).clamp_min(0).squeeze(-1).unsqueeze(1)
    return out * scale


def siddon(volume, source, target, img, spec: RenderSpec, mask=None):
    """Siddon's exact ray tracing as sort -> midpoints -> nearest lookup -> * diff(alpha)."""
    shape = volume.shape
    sdd = target - source + spec.eps
    per_axis = []
    for ax in range(3):
        planes = torch.arange(shape[ax] + 1).to(source) - spec.voxel_shift
        per_axis.append((planes.expand(len(source), 1, -1) - source[..., ax : ax + 1]) / sdd[..., ax : ax + 1])
    alphamin, alphamax = _alpha_minmax(source, target, shape, spec)
    if spec.per_ray_clamp:
        # the ends of the ray's own interval are segment boundaries too (they coincide with plane
        # crossings unless the source/target lies inside the volume, where they cut the partial
        # first/last segment that the column filter below would otherwise discard)
        per_axis += [alphamin, torch.maximum(alphamax, alphamin)]
    alphas = torch.sort(torch.cat(per_axis, dim=-1), dim=-1).values
    if spec.filter_intersections_outside_volume:
        alphas = _filter_columns(alphas, alphamin, alphamax)
    if spec.per_ray_clamp:
        # integrate over this ray's own [alphamin, alphamax] only: clip every crossing into it, so
        # segments outside collapse to zero length and a source/target inside the volume yields
        # the physically partial first/last segment.
        hi = torch.maximum(alphamax, alphamin)
        alphas = torch.minimum(torch.maximum(alphas, alphamin), hi)
    alphamid = (alphas[..., :-1] + alphas[..., 1:]) / 2
    xyzs = _xyzs(alphamid, source, target, shape, spec)
    voxels = _lookup(volume, xyzs, "nearest", spec.align_corners)
    seg = torch.diff(alphas, dim=-1)
    if spec.per_ray_clamp:
        seg = seg * (alphamax > alphamin).to(seg.dtype)
    out = _to_channels(voxels * seg, volume, mask, xyzs, spec)
    return out * img


def render(volume, source, target, img, spec: RenderSpec, mask=None, chunk: int | None = None, label_nudge=None):
    """Dispatch + optional chunking over rays (the materialised [B,n,K,3] grid is huge: SURVEY 3.3)."""
    fn = trilinear if spec.renderer == "trilinear" else siddon
    if label_nudge is not None:
        if spec.renderer != "trilinear" or (chunk is not None and target.shape[1] > chunk):
            raise ValueError("label_nudge: trilinear, unchunked")
        return trilinear(volume, source, target, img, spec, mask, label_nudge=label_nudge)
    if chunk is None or target.shape[1] <= chunk:
        return fn(volume, source, target, img, spec, mask)
    if spec.renderer == "siddon" and not spec.per_ray_clamp:
        raise ValueError("the literal (batch-filtered) siddon cannot be chunked without changing its result")
    kw = {}
    if spec.renderer == "trilinear" and spec.clip_to_volume == "batch":
        kw["window"] = batch_window(source, target, volume.shape, spec)   # of the whole call, not of a chunk
    outs = []
    for lo in range(0, target.shape[1], chunk):
        sl = slice(lo, lo + chunk)
        outs.append(fn(volume, source, target[:, sl], img[..., sl], spec, mask, **kw))
    return torch.cat(outs, dim=-1)


# --------------------------------------------------------------------------------------
# detector geometry (rows a2-a4 of SURVEY.md section 8a), restated so the oracle is self-contained
# --------------------------------------------------------------------------------------
def detector_plane(height: int, width: int, reverse_x_axis: bool = False, dtype=torch.float32):
    """Source at the origin, H x W unit-spaced grid centred on (0, 0, 1); rows vary slowest."""
    h_off = 1.0 if height % 2 else 0.5
    w_off = 1.0 if width % 2 else 0.5
    t = torch.arange(-height // 2, height // 2, dtype=dtype) + h_off
    s = torch.arange(-width // 2, width // 2, dtype=dtype) + w_off
    if reverse_x_axis:
        s = -s
    coefs = torch.cartesian_prod(t, s).reshape(-1, 2)
    target = torch.cat([coefs, torch.ones(len(coefs), 1, dtype=dtype)], dim=-1)
    source = torch.zeros(1, 3, dtype=dtype)
    return source[None], target[None]


def calibration_matrix(sdd, delx, dely, x0, y0, dtype=torch.float32):
    return torch.tensor(
        [[dely, 0, 0, y0], [0, delx, 0, x0], [0, 0, sdd, 0], [0, 0, 0, 1]],
        dtype=dtype,
    )


REORIENT = {
    "AP": [[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]],
    "PA": [[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]],
    None: [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
}


def _apply(matrix, x):
    return torch.einsum("bij,bnj->bni", matrix[..., :3, :3], x) + matrix[..., None, :3, 3]


def rays_from_pose(pose, height, width, sdd, delx, dely, x0, y0, orientation="AP", reverse_x_axis=False):
    """pose[B,4,4] (camera-to-world) -> source[B,1,3], target[B,n,3] in world mm."""
    dtype = pose.dtype
    source, target = detector_plane(height, width, reverse_x_axis, dtype)
    K = calibration_matrix(sdd, delx, dely, x0, y0, dtype)[None]
    target = _apply(K, target)
    reorient = torch.tensor(REORIENT[orientation], dtype=dtype)[None]
    total = pose @ reorient  # reorient first, then the extrinsic pose
    return _apply(total, source), _apply(total, target)


def drr_from_pose(
    volume, affine, pose, height, width, sdd, delx, dely, x0, y0, spec: RenderSpec,
    orientation="AP", reverse_x_axis=False, mask=None, chunk=None,
):
    """The 4-call sequence of src/xvr/model/trainer.py:283-289 == DRR.forward(pose)."""
    source, target = rays_from_pose(pose, height, width, sdd, delx, dely, x0, y0, orientation, reverse_x_axis)
    img = (target - source).norm(dim=-1).unsqueeze(1)
    affinv = torch.linalg.inv(affine)[None].to(pose)
    source, target = _apply(affinv, source), _apply(affinv, target)
    img = render(volume, source, target, img, spec, mask, chunk)
    return img.view(len(pose), -1, height, width)
