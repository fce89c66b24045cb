"""Multiscale iterative pose refinement: the outer loop of ``xvr register`` around the render path.

Mirrors ``_RegistrarBase.run`` / ``run_test_time_optimization``
(/root/reference/src/xvr/registrar/base.py:125-292) without its file IO (DICOM parsing, pngs,
parameters.pt) and without a pose-regressor initialiser -- the caller passes the target image, the
detector intrinsics and an initial pose (what ``RegistrarFixed`` does,
/root/reference/src/xvr/registrar/fixed.py:70-81).

Schedule reproduced from the reference:
* pyramid ratios from ``_parse_scales`` (base.py:402-407); ``rescale_detector_`` per stage (:212)
* per stage Adam(maximize=True) with lr_rot / lr_xyz divided by prod 2^(stage-1) (:209,221-228),
  ReduceLROnPlateau(factor 0.1, patience, threshold, mode="max") (:229-235)
* loss = beta * mNCC([None, patch], [.5, .5]) + (1 - beta) * gradNCC(patch, sigma) on
  XrayTransforms-ed images (:115-123, :250-251)
* stop a stage after ``max_n_plateaus`` learning-rate drops, the initial lr counting as one (:238-239,270-278)

What is NOT reproduced: the reference's 4+ device->host syncs per iteration (two ``loss.item()``, the
pose -> euler -> ``tolist()``, two ``cuda.synchronize()``, base.py:246-267).  Only the plateau scheduler
needs the loss on the host; the trajectory is kept on the device and copied once per stage.
"""

from __future__ import annotations

import time
from copy import deepcopy

import torch

from . import _lib
from .drr import DRR
from .metrics import (GradientNormalizedCrossCorrelation2d, MultiscaleNormalizedCrossCorrelation2d,
                      XrayTransforms)
from .pose import RigidTransform, convert
from .pose_opt import RegistrationStage
from .registration import Registration
from .pose_opt import PARAM_KINDS
from .similarity import EqualizedSimilarity, FusedSimilarity, GeneralSimilarity
from .xray import XrayPreparation


def parse_scales(scales, crop: int, height: int):
    """Cumulative down-scaling factors -> per-stage ratios (base.py:402-407)."""
    if isinstance(scales, str):
        scales = scales.split(",")
    pyramid = [1.0] + [float(x) * (height / (height + crop)) for x in scales]
    return [pyramid[i] / pyramid[i + 1] for i in range(len(pyramid) - 1)]


class Registrar:
    def __init__(self, drr: DRR, scales="8", n_itrs="500", parameterization="euler_angles", convention="ZXY",
                 lr_rot=1e-2, lr_xyz=1e0, patience=10, threshold=1e-4, max_n_plateaus=3, crop=0, equalize=False,
                 mncc_patch_size=9, gncc_patch_size=11, sigma=0.0, beta=0.5, verbose=0, use_graph=None, fused=None,
                 device_loop=None, check_every=8, subtract_background=False, linearize=True, reducefn="max"):
        self.drr = drr
        # what turns the raw pixel array into the registration target (xvr_amd/xray.py; the reference's read_xray options,
        # /root/reference/src/xvr/registrar/base.py:128-141): `prepare_xray(raw)` applies it, parameters.pt records it.  Defaults
        # as every registrar of the reference (linearize=True: registrar/model.py:16, dicom.py:14, fixed.py:20): crop, rescale to
        # [0, 1] (always), log-linearise.  `run(gt, ...)` takes an image that is ALREADY the target (a rendered DRR, a prepared
        # X-ray) and applies none of this; only `prepare_xray` does.
        self.xray_preparation = XrayPreparation(trim=int(crop), subtract_background=bool(subtract_background),
                                                linearize=bool(linearize), frames=reducefn)
        self.scales = scales.split(",") if isinstance(scales, str) else [str(s) for s in scales]
        self.n_itrs = [int(n) for n in (n_itrs.split(",") if isinstance(n_itrs, str) else n_itrs)]
        assert len(self.scales) == len(self.n_itrs)
        self.parameterization, self.convention = parameterization, convention
        self.lr_rot, self.lr_xyz = lr_rot, lr_xyz
        self.patience, self.threshold, self.max_n_plateaus = patience, threshold, max_n_plateaus
        self.crop, self.equalize, self.verbose = crop, equalize, verbose
        self.beta = beta
        # Capture one whole iteration (render -> transforms -> similarity -> backward -> Adam) in a HIP
        # graph per pyramid stage and replay it: the reference's iteration is ~150 tiny launches and is
        # launch-bound (4.4 ms at 256^2 and at 512^2 alike on MI355X); replaying costs a fraction.
        self.use_graph = torch.cuda.is_available() if use_graph is None else use_graph
        self.mncc_patch_size, self.gncc_patch_size, self.sigma = mncc_patch_size, gncc_patch_size, sigma
        # transforms + similarity + their backward as one fused HIP call (xvr_amd/similarity.py) when the
        # configuration allows it; plain torch otherwise
        self.fused = fused
        # pose -> camera, its chain rule, Adam, the plateau scheduler and the stopping rule on the device too
        # (xvr_amd/pose_opt.py): eight launches per iteration, one host sync every `check_every` iterations.
        # Any parameterisation xvr_pose_convert_forward knows (pose_opt.PARAM_KINDS); None = use it whenever possible.
        self.device_loop, self.check_every = device_loop, check_every
        self.sim1 = MultiscaleNormalizedCrossCorrelation2d([None, mncc_patch_size], [0.5, 0.5])
        self.sim2 = GradientNormalizedCrossCorrelation2d(gncc_patch_size, sigma)

    def prepare_xray(self, raw: torch.Tensor) -> torch.Tensor:
        """raw pixel array [1, 1, (T,) H, W] -> the target image `run` takes: collimator crop, rescale, optional background
        subtraction / linearisation / frame reduction (xvr_amd/xray.py), wherever `raw` lives."""
        return self.xray_preparation(raw)

    def xray_block(self, filename=None) -> dict:
        """parameters.pt's "xray" entry (/root/reference/src/xvr/registrar/base.py:367-373): what prepare_xray really does."""
        p = self.xray_preparation
        return {"filename": filename, "crop": self.crop, "subtract_background": p.subtract_background, "linearize": p.linearize,
                "reducefn": p.frames}

    def imagesim(self, x, y):
        self.sim2.to(x.device)
        return self.beta * self.sim1(x, y) + (1 - self.beta) * self.sim2(x, y)

    def run(self, gt: torch.Tensor, init_pose: RigidTransform, intrinsics: dict | None = None):
        """gt: [1,1,H,W] target image at full resolution.  Returns a dict with final_pose, ncc history,
        per-iteration times, learning rates and the (r1,r2,r3,tx,ty,tz) trajectory."""
        device = self.drr.density.device
        *_, height, width = gt.shape
        drr = deepcopy(self.drr)
        if intrinsics is not None:
            drr.set_intrinsics_(**{**intrinsics, "height": height, "width": width})
        elif (drr.detector.height, drr.detector.width) != (height, width):
            raise ValueError("gt and the DRR detector differ in size; pass intrinsics")
        scales = parse_scales(self.scales, self.crop, height)
        rot, xyz = init_pose.convert(self.parameterization, self.convention)
        reg = Registration(drr, rot.to(device), xyz.to(device), self.parameterization, self.convention)
        gt = gt.to(device)

        traj, nccs, times, lrs = [], [], [0.0], [[self.lr_rot, self.lr_xyz]]
        step_size_scalar = 1.0
        img = transform = None
        for stage, (scale, n_itr) in enumerate(zip(scales, self.n_itrs), start=1):
            reg.drr.rescale_detector_(scale)
            transform = XrayTransforms(reg.drr.detector.height, reg.drr.detector.width, equalize=self.equalize)
            img = transform(gt)
            h, w = reg.drr.detector.height, reg.drr.detector.width
            use_fused = (self.fused if self.fused is not None else True) and device.type == "cuda" and \
                FusedSimilarity.supported(h, w, self.mncc_patch_size, self.gncc_patch_size, self.sigma, self.equalize)
            fused_sim = FusedSimilarity(img, self.mncc_patch_size, self.gncc_patch_size, self.beta) if use_fused else None
            step_size_scalar *= 2 ** (stage - 1)
            graphed = self.use_graph and device.type == "cuda"
            lr_rot = self.lr_rot / step_size_scalar
            lr_xyz = self.lr_xyz / step_size_scalar
            device_loop = (self.device_loop if self.device_loop is not None else True) and device.type == "cuda" and n_itr > 0 \
                and self.parameterization in PARAM_KINDS and (self.fused if self.fused is not None else True)
            if device_loop:
                sim_obj = self._stage_similarity(img, transform, h, w, fused_sim, per_image=False)
                stage_run = RegistrationStage(reg.drr, sim_obj, reg.rotation.data, reg.translation.data, self.convention,
                                              lr_rot, lr_xyz, self.patience, self.threshold, self.max_n_plateaus,
                                              max_iters=n_itr, parameterization=self.parameterization)
                _, stage_times = stage_run.run(n_itr, self.check_every, use_graph=graphed)
                rows = stage_run.results()[0]
                np_ = rows.shape[1] - 3            # the k rotation parameters + the translation
                nccs += rows[:, np_].tolist()
                traj += rows[:, :np_].tolist()
                lrs += rows[:, np_ + 1:np_ + 3].tolist()
                times += stage_times
                if self.verbose:
                    print(f"stage {stage}: {len(rows)} iterations on the device, ncc = {nccs[-1]:.4f}")
                continue
            if graphed:  # capturable Adam reads its learning rates from device tensors
                lr_rot, lr_xyz = torch.tensor(lr_rot, device=device), torch.tensor(lr_xyz, device=device)
            optimizer = torch.optim.Adam(
                [{"params": [reg.rotation], "lr": lr_rot}, {"params": [reg.translation], "lr": lr_xyz}],
                maximize=True, capturable=graphed,
            )
            scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(
                optimizer, factor=0.1, patience=self.patience, threshold=self.threshold, mode="max")
            n_plateaus, current_lr = 0, float("inf")
            stage_losses, stage_params = [], []

            def iteration():
                if fused_sim is not None:
                    loss = fused_sim(reg())
                else:
                    loss = self.imagesim(img, transform(reg()))
                loss.sum().backward()
                optimizer.step()
                return loss.detach()

            graph = static_loss = None
            n_warm = 3  # eager iterations before capture (they are real iterations of the optimisation)
            for itr in range(n_itr):
                t0 = time.perf_counter()
                if graphed and itr == n_warm and graph is None:
                    try:
                        optimizer.zero_grad(set_to_none=True)
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph):
                            static_loss = iteration()
                        graph.replay()   # capturing enqueues nothing: this replay IS iteration `itr`
                        loss = static_loss
                    except RuntimeError as e:
                        # capture is an optimisation, never a requirement -- but only a CAPTURE failure (torch raises
                        # RuntimeError / AcceleratorError for an op that cannot be captured) is taken for one, and never
                        # silently: a HipLibraryError, a shape error or anything else in the iteration propagates
                        if isinstance(e, _lib.HipLibraryError):
                            raise
                        import warnings

                        warnings.warn(f"Registrar: HIP-graph capture of the iteration failed ({e}); continuing eagerly", RuntimeWarning)
                        graph, graphed = None, False
                        optimizer.zero_grad()
                        loss = iteration()
                elif graph is not None:
                    graph.replay()
                    loss = static_loss
                else:
                    optimizer.zero_grad()
                    loss = iteration()
                stage_losses.append(loss.clone())
                stage_params.append(torch.cat([reg.rotation.detach(), reg.translation.detach()], dim=-1).clone())
                scheduler.step(loss.sum().item())   # the one host sync the plateau logic needs
                times.append(time.perf_counter() - t0)
                lr = [float(x) for x in scheduler.get_last_lr()]
                lrs.append(lr)
                if lr[0] < current_lr:
                    current_lr = lr[0]
                    n_plateaus += 1
                if n_plateaus == self.max_n_plateaus:
                    break
            del graph
            if stage_losses:
                nccs += torch.stack(stage_losses).reshape(-1).tolist()
                traj += torch.stack(stage_params).reshape(len(stage_params), -1).tolist()
            if self.verbose:
                print(f"stage {stage}: {len(stage_losses)} iterations, ncc = {nccs[-1]:.4f}")
        with torch.no_grad():
            final_ncc = self.imagesim(img, transform(reg())).sum().item()
        nccs.append(final_ncc)
        return dict(final_pose=RigidTransform(reg.pose.matrix.detach()), init_pose=init_pose, nccs=nccs, times=times,
                    lrs=lrs, trajectory=self._rows_as_euler_zxy(traj), runtime=sum(times), drr=reg.drr)

    def _stage_similarity(self, img, transform, h, w, fused_sim=None, per_image=False):
        """The similarity object of one pyramid stage for the device-resident loop: the single fused call, the tape-free chain for
        ``equalize``, or -- sigma > 0, patches beyond 15 -- HIP kernels strung together by autograd."""
        cfg = (h, w, self.mncc_patch_size, self.gncc_patch_size, self.sigma, self.equalize)
        if fused_sim is not None:
            return fused_sim
        if self.fused is False:   # the caller asked for the composed formulation (the A/B reference of the fused calls)
            tf = transform if not per_image else XrayTransforms(h, w, equalize=self.equalize, per_image=True)
            return GeneralSimilarity(img, tf, self.mncc_patch_size, self.gncc_patch_size, self.sigma, self.beta)
        if FusedSimilarity.supported(*cfg):
            return FusedSimilarity(img, self.mncc_patch_size, self.gncc_patch_size, self.beta, per_image=per_image)
        if EqualizedSimilarity.supported(*cfg):
            return EqualizedSimilarity(img, self.mncc_patch_size, self.gncc_patch_size, self.beta, per_image=per_image)
        tf = transform if not per_image else XrayTransforms(h, w, equalize=self.equalize, per_image=True)
        return GeneralSimilarity(img, tf, self.mncc_patch_size, self.gncc_patch_size, self.sigma, self.beta)

    def _rows_as_euler_zxy(self, rows):
        """Trajectory rows (rotation parameters + translation in THIS registrar's parameterisation, one row per
        iteration) -> the reference's logging convention: ``pose.convert("euler_angles", "ZXY")`` whatever the
        optimisation variables are (/root/reference/src/xvr/registrar/base.py:200-203, 262-266).  One batched
        conversion for the whole trajectory instead of ~60 tiny launches per iteration."""
        if not rows:
            return []
        if self.parameterization == "euler_angles" and self.convention == "ZXY":
            return [list(r) for r in rows]
        t = torch.as_tensor(rows, dtype=torch.float32)
        pose = convert(t[:, :-3], t[:, -3:], parameterization=self.parameterization, convention=self.convention)
        return torch.cat(pose.convert("euler_angles", "ZXY"), dim=-1).tolist()

    def run_batch(self, gt: torch.Tensor, init_poses: RigidTransform, intrinsics: dict | None = None) -> list:
        """Multi-start in ONE batch: the B initial poses are B independent registrations of the same target
        (``gt`` [1,1,H,W]) -- or of B different targets of the same geometry (``gt`` [B,1,H,W]) --,
        advanced together by the device-resident stage (xvr_amd/pose_opt.py) -- every pose has its own Adam
        moments, plateau scheduler, learning rates and stopping flag on the device, the similarity
        standardises every rendered image by its own min/max (``per_image``), and a stage ends when all poses
        have met the stopping rule (finished poses are left untouched).  One launch then renders B poses, which
        fills the GPU where a single 256^2 pose is one wavefront per SIMD.  Same schedule per pose as ``run``;
        needs a parameterisation of pose_opt.PARAM_KINDS.  Returns one result dict per pose."""
        device = self.drr.density.device
        *_, height, width = gt.shape
        B = len(init_poses)
        if gt.shape[0] not in (1, B):
            raise ValueError(f"gt holds {gt.shape[0]} images for {B} poses: pass one target, or one per pose")
        if self.parameterization not in PARAM_KINDS or device.type != "cuda":
            raise RuntimeError(f"run_batch needs the device-resident loop: a GPU and one of {sorted(PARAM_KINDS)}")
        drr = deepcopy(self.drr)
        if intrinsics is not None:
            drr.set_intrinsics_(**{**intrinsics, "height": height, "width": width})
        elif (drr.detector.height, drr.detector.width) != (height, width):
            raise ValueError("gt and the DRR detector differ in size; pass intrinsics")
        scales = parse_scales(self.scales, self.crop, height)
        rot, xyz = init_poses.convert(self.parameterization, self.convention)
        rot, xyz = rot.to(device).contiguous().clone(), xyz.to(device).contiguous().clone()
        gt = gt.to(device)
        per = [dict(traj=[], nccs=[], lrs=[[self.lr_rot, self.lr_xyz]], times=[0.0]) for _ in range(B)]
        step_size_scalar = 1.0
        transform = img = None
        for stage, (scale, n_itr) in enumerate(zip(scales, self.n_itrs), start=1):
            drr.rescale_detector_(scale)
            h, w = drr.detector.height, drr.detector.width
            transform = XrayTransforms(h, w, equalize=self.equalize)
            img = torch.cat([transform(gt[b:b + 1]) for b in range(gt.shape[0])])   # every target standardised on its own
            fixed = img.expand(B, -1, -1, -1) if img.shape[0] == 1 else img   # one target for all starts, or one per pose
            step_size_scalar *= 2 ** (stage - 1)
            if n_itr <= 0:   # (a stage without iterations only rescales the detector: no similarity object, no workspaces)
                continue
            sim = self._stage_similarity(fixed.contiguous(), transform, h, w, per_image=True)
            stage_run = RegistrationStage(drr, sim, rot, xyz, self.convention, self.lr_rot / step_size_scalar,
                                          self.lr_xyz / step_size_scalar, self.patience, self.threshold, self.max_n_plateaus,
                                          max_iters=n_itr, parameterization=self.parameterization)
            _, stage_times = stage_run.run(n_itr, self.check_every, use_graph=self.use_graph)
            for b, rows in enumerate(stage_run.results()):
                np_ = rows.shape[1] - 3
                per[b]["nccs"] += rows[:, np_].tolist()
                per[b]["traj"] += rows[:, :np_].tolist()
                per[b]["lrs"] += rows[:, np_ + 1:np_ + 3].tolist()
                per[b]["times"] += stage_times[: len(rows)]
            if self.verbose:
                print(f"stage {stage}: iterations per pose {[len(r) for r in stage_run.results()]}")
        final = convert(rot, xyz, parameterization=self.parameterization, convention=self.convention)
        with torch.no_grad():
            # (XrayTransforms standardises over the whole tensor it is given: one image at a time, as in run())
            final_ncc = [self.imagesim(transform(gt[b:b + 1] if gt.shape[0] == B and B > 1 else gt),
                                       transform(drr(final[b]))).sum().item() for b in range(B)]
        out = []
        for b in range(B):
            per[b]["nccs"].append(final_ncc[b])
            out.append(dict(final_pose=RigidTransform(final.matrix[b:b + 1].detach()), init_pose=init_poses[b], nccs=per[b]["nccs"],
                            times=per[b]["times"], lrs=per[b]["lrs"], trajectory=self._rows_as_euler_zxy(per[b]["traj"]),
                            runtime=sum(per[b]["times"]),
                            drr=drr))
        return out

    def parameters_dict(self, result: dict, intrinsics: dict | None = None, volume=None, mask=None, xray=None,
                        registrar_type: str = "fixed") -> dict:
        """The ``parameters.pt`` dictionary of the reference (/root/reference/src/xvr/registrar/base.py:355-394):
        same keys, 4x4 CPU poses, and the trajectory as the reference's ``_make_csv`` builds it (base.py:172-187,
        410-422): a pandas DataFrame with columns r1, r2, r3, tx, ty, tz (Euler ZXY, whatever the optimisation's
        parameterisation), ncc, times, lr_rot, lr_xyz -- or the same rows as a list of dicts where pandas is absent."""
        d = result["drr"].detector
        cols = ["r1", "r2", "r3", "tx", "ty", "tz", "ncc", "times", "lr_rot", "lr_xyz"]
        init = result["init_pose"].convert("euler_angles", "ZXY")
        rows = [torch.cat(init, dim=-1).reshape(-1).tolist()] + result["trajectory"]
        if any(len(r) != 6 for r in rows):
            raise ValueError("trajectory rows must be (r1, r2, r3, tx, ty, tz)")
        n = min(len(rows), len(result["nccs"]), len(result["times"]), len(result["lrs"]))
        table = [[*rows[i], result["nccs"][i], result["times"][i], *result["lrs"][i]] for i in range(n)]
        try:
            import pandas as pd
            traj = pd.DataFrame(table, columns=cols)
        except ImportError:   # pragma: no cover
            traj = [dict(zip(cols, r)) for r in table]
        intr = intrinsics or dict(sdd=d.sdd, height=d.height, width=d.width, delx=d.delx, dely=d.dely, x0=d.x0, y0=d.y0)
        return {
            "drr": {"volume": volume, "mask": mask, "labels": None, "orientation": self.drr.subject.orientation, **intr,
                    "reverse_x_axis": d.reverse_x_axis, "renderer": self.drr.renderer.renderer_name,
                    "read_kwargs": {}, "drr_kwargs": {"voxel_shift": self.drr.renderer.voxel_shift}},
            "xray": self.xray_block(xray),
            "optimization": {"equalize": self.equalize, "init_only": False, "scales": self.scales, "n_itrs": self.n_itrs,
                             "parameterization": self.parameterization, "convention": self.convention,
                             "lr_rot": self.lr_rot, "lr_xyz": self.lr_xyz, "patience": self.patience,
                             "max_n_plateaus": self.max_n_plateaus},
            "init_pose": result["init_pose"].matrix.detach().cpu(),
            "final_pose": result["final_pose"].matrix.detach().cpu(),
            "type": registrar_type, "runtime": result["runtime"], "trajectory": traj,
        }

    def save(self, path, result: dict, **kw) -> None:
        torch.save(self.parameters_dict(result, **kw), path)


def register_multistart(registrar: Registrar, gt: torch.Tensor, init_poses: RigidTransform, intrinsics: dict | None = None,
                        batched: bool = False):
    """Multi-start registration (configs[3]: several initial poses, one or more per GPU).  The initial
    poses are split contiguously over the ranks (``shard_bounds``); every rank refines its own starts
    independently -- no communication inside the optimisation -- and one 68-byte all-gather of
    (final similarity, 4x4 pose) lets every rank take the arg-max.  ``batched``: a rank refines its starts
    together in one batch (``Registrar.run_batch``) instead of one after the other.  Returns (best_ncc, best_pose[4,4],
    rank_of_best, local_results)."""
    from .distributed import multistart_best, shard_bounds

    lo, hi = shard_bounds(len(init_poses))
    if batched and hi > lo:   # this rank's starts as one batch (Registrar.run_batch)
        results = registrar.run_batch(gt, init_poses[lo:hi], intrinsics)
    else:
        results = [registrar.run(gt, init_poses[i], intrinsics) for i in range(lo, hi)]
    device = registrar.drr.density.device
    if results:
        best = max(results, key=lambda r: r["nccs"][-1])
        score = torch.tensor(best["nccs"][-1], device=device)
        pose = best["final_pose"].matrix.detach().reshape(4, 4).to(device)
    else:  # more ranks than starts: this rank abstains
        score = torch.tensor(float("-inf"), device=device)
        pose = torch.eye(4, device=device)
    best_score, best_pose, best_rank = multistart_best(score, pose)
    return best_score, best_pose, best_rank, results
