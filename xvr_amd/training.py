"""The render-side pieces of xvr's training step (the callers immediately around the hot call).

* ``get_random_pose``  -- /root/reference/src/xvr/model/sampler.py:5-38
* ``render_samples``   -- /root/reference/src/xvr/model/trainer.py:279-304 (the exploded 4-call
  sequence detector -> ray length -> affine_inverse -> renderer -> reshape, then the foreground
  mask / keep test with thresholds 0.10 / 0.05)
"""

from __future__ import annotations

import torch

from .pose import convert


def get_random_pose(alphamin, alphamax, betamin, betamax, gammamin, gammamax, txmin, txmax, tymin, tymax,
                    tzmin, tzmax, batch_size, generator=None):
    """A batch of random poses: uniform Euler-ZXY angles (degrees, wrapped to [-180, 180)) and translations (mm); the six columns
    are drawn one after the other, alpha first, so that a seeded generator reproduces the reference's draws
    (/root/reference/src/xvr/model/sampler.py:5-38; tests/test_reference_goldens.py holds it to the reference's output)."""
    bounds = ((alphamin, alphamax), (betamin, betamax), (gammamin, gammamax), (txmin, txmax), (tymin, tymax), (tzmin, tzmax))
    cols = [lo + torch.rand(batch_size, 1, generator=generator) * (hi - lo) for lo, hi in bounds]
    rot = torch.remainder(torch.cat(cols[:3], dim=1) + 180, 360) - 180
    return convert(rot, torch.cat(cols[3:], dim=1), parameterization="euler_angles", convention="ZXY", degrees=True)


class _Foreground(torch.autograd.Function):
    """mask = img > 0, img.sum(dim=1), keep -- the tail of render_samples -- as one HIP pass (xvr_drr_foreground,
    include/xvr_drr.h).  The gradient of the channel sum is the upstream gradient EXPANDED over the channels (stride 0): the
    renderer's backward recognises it and takes the unmasked voxel / pose gradient paths."""

    @staticmethod
    def forward(ctx, img, threshold):
        from . import _lib
        from .renderers import _ptr, _stream, _timed

        B, C, H, W = img.shape
        x = img.contiguous()
        total = torch.empty(B, 1, H, W, dtype=img.dtype, device=img.device)
        mask = torch.empty(B, C, H, W, dtype=torch.bool, device=img.device)
        keep = torch.empty(B, dtype=torch.bool, device=img.device)
        count = torch.empty(B, dtype=torch.int32, device=img.device)
        lib = _lib.load()
        _lib.check(_timed("foreground", lib.xvr_drr_foreground, _ptr(x), B, C, H * W, float(threshold), _ptr(total), _ptr(mask),
                          _ptr(count), _ptr(keep), _stream()), "xvr_drr_foreground")
        ctx.shape = img.shape
        ctx.mark_non_differentiable(mask, keep)
        return total, mask, keep

    @staticmethod
    def backward(ctx, g_total, _g_mask, _g_keep):
        return g_total.expand(ctx.shape), None


def render_samples(drr, volume, seg, affinv, pose, img_threshold=0.10, mask_threshold=0.05):
    """-> (img [B,1,H,W], mask [B,C,H,W] bool, keep [B] bool)."""
    if (volume.is_cuda and getattr(drr, "fused_rays", False)
            and affinv.matrix.data_ptr() == drr._affine_inverse.data_ptr()):   # the DRR's own inverse affine, not a subject's
        # detector -> ray length -> inverse affine as ONE launch from the pose's camera vector (same numbers:
        # tests/test_hip_parity.py::test_fused_ray_generation_matches_the_detector_path)
        from .drr import rays_from_camera

        G, c = drr._camera_affine_cached()
        cam = torch.addmm(c, pose.matrix[:, :3, :].reshape(len(pose), 12), G.T)
        source, target, img = rays_from_camera(cam, drr.detector.height, drr.detector.width)
    else:
        source, target = drr.detector(pose, None)
        img = (target - source).norm(dim=-1).unsqueeze(1)
        source, target = affinv(source), affinv(target)
    img = drr.renderer(volume, source, target, img, mask=seg)
    img = drr.reshape_transform(img, batch_size=len(pose))
    if not (img.is_cuda and img.dtype == torch.float32):
        raise RuntimeError("render_samples: the foreground / keep pass is a HIP kernel (float32 CUDA renders, no CPU path)")
    return _Foreground.apply(img, img_threshold if img.shape[1] == 1 else mask_threshold)


# ---------------------------------------------------------------------------------------------------------------
# training checkpoints: the reference's ``NNNN.pth`` schema
# ---------------------------------------------------------------------------------------------------------------
CHECKPOINT_KEYS = ("model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "itr", "model_number", "date", "config")


def save_checkpoint(outpath, model, optimizer, scheduler, itr: int, model_number: int, config: dict):
    """Write ``{outpath}/{model_number:04d}.pth`` with exactly the keys of the reference's ``Trainer._checkpoint``
    (/root/reference/src/xvr/model/trainer.py:318-332): model / optimizer / scheduler state dicts, ``itr``,
    ``model_number``, ``date`` (a ``datetime``) and the trainer's ``config`` dict.  Returns (path, model_number + 1) --
    the reference increments its counter after every save.  Works for any ``torch.nn.Module`` regressor (the timm
    network itself is out of scope; a file written by the reference loads here and vice versa)."""
    from datetime import datetime
    from pathlib import Path

    path = Path(outpath) / f"{model_number:04d}.pth"
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save({
        "model_state_dict": model.state_dict(),
        "optimizer_state_dict": optimizer.state_dict(),
        "scheduler_state_dict": scheduler.state_dict(),
        "itr": itr,
        "model_number": model_number,
        "date": datetime.now(),
        "config": config,
    }, path)
    return path, model_number + 1


def load_checkpoint(ckptpath, reuse_optimizer: bool = False):
    """The reference's ``_load_checkpoint`` (/root/reference/src/xvr/model/utils.py:176-183): -> (ckpt, start_itr,
    model_number); the iteration and file counters restart from 0 unless the optimizer is being reused."""
    if ckptpath is None:
        return None, 0, 0
    ckpt = torch.load(ckptpath, weights_only=False)   # (holds a datetime and the config dict, as the reference's files do)
    missing = [k for k in CHECKPOINT_KEYS if k not in ckpt]
    if missing:
        raise KeyError(f"{ckptpath}: not an xvr training checkpoint, missing {missing}")
    if reuse_optimizer:
        return ckpt, ckpt["itr"], ckpt["model_number"]
    return ckpt, 0, 0


def restore_from_checkpoint(ckpt, model, optimizer=None, scheduler=None, reuse_optimizer: bool = False):
    """What ``initialize_modules`` does with a loaded checkpoint (/root/reference/src/xvr/model/utils.py:132-150): the
    model weights always, optimizer and scheduler state only when ``reuse_optimizer``."""
    if ckpt is None:
        return
    model.load_state_dict(ckpt["model_state_dict"])
    if reuse_optimizer:
        if optimizer is not None:
            optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if scheduler is not None:
            scheduler.load_state_dict(ckpt["scheduler_state_dict"])
