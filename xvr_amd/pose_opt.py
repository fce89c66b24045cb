"""Device-resident pose optimisation for the registration loop (include/xvr_pose.h).

``RegistrationStage`` runs one pyramid stage of ``_RegistrarBase.run_test_time_optimization``
(/root/reference/src/xvr/registrar/base.py:245-280) as FIVE C-ABI calls per iteration and no
autograd tape:

    pose -> camera            xvr_pose_camera_forward          (reg(): convert + detector + affine_inverse)
    camera -> DRR + jacobian  xvr_drr_{trilinear,siddon}_forward_camera   (rays generated in the kernel)
    DRR -> similarity + grad  xvr_sim_ncc_forward_backward     (transform, imagesim, backward of both)
    grad -> camera            xvr_drr_jac_to_camera_backward   (loss.backward() through renderer and rays)
    camera -> pose, Adam, ReduceLROnPlateau, stopping rule     xvr_pose_opt_step

The optimiser, the scheduler and the stopping rule keep their state on the device, so the host replays a
captured HIP graph ``check_every`` times between looks at the ``done`` flag; iterations enqueued after a
pose has met the stopping rule are no-ops, i.e. the trajectory is exactly that of a loop which checks
after every step.  Every reduction on this path adds in a fixed order (no floating-point atomics), so a
registration is reproducible bit for bit from run to run.  ``PoseCamera`` exposes the first step to autograd for callers with their own loop.
"""

from __future__ import annotations

import ctypes
import time

import numpy as np
import torch

from . import _lib
from .renderers import _ptr, _stream, _timed, make_cspec

__all__ = ["PoseCamera", "pose_camera", "RegistrationStage", "axes_of"]

MAX_PARAMS = 13   # XVR_POSE_MAX_PARAMS
FUSED_TAIL = True  # Euler + fused similarity: render -> xvr_sim_ncc_registration_step (False: the five-call iteration, A/B)
STATE_DTYPE = np.dtype([("m", "f4", MAX_PARAMS), ("v", "f4", MAX_PARAMS), ("lr", "f4", 2), ("seen_lr", "f4"), ("step", "i4"),
                        ("n_bad", "i4"), ("n_plateaus", "i4"), ("done", "i4"), ("iter", "i4"), ("best", "f8")], align=True)


# parameterisations the device loop knows (xvr_pose_convert_forward's kinds) and their number of rotation parameters
PARAM_KINDS = {"euler_angles": (0, 3), "axis_angle": (1, 3), "quaternion": (2, 4), "quaternion_adjugate": (3, 10), "rotation_6d": (4, 6),
               "se3_log_map": (5, 3), "rotation_10d": (6, 10)}


def axes_of(convention: str):
    if not isinstance(convention, str) or len(convention) != 3 or any(ch not in "XYZ" for ch in convention):
        raise ValueError(f"invalid Euler convention {convention!r}")
    return (ctypes.c_int * 3)(*["XYZ".index(ch) for ch in convention])


class PoseCamera(torch.autograd.Function):
    """(rot [B,3] Euler angles, xyz [B,3]) -> cam [B,24] for the constants (G, c) of ``DRR.camera_affine``."""

    @staticmethod
    def forward(ctx, rot, xyz, G, c, convention):
        lib = _lib.load()
        rot_c, xyz_c = rot.contiguous(), xyz.contiguous()
        B = rot_c.shape[0]
        cam = torch.empty(B, 24, device=rot.device, dtype=torch.float32)
        rc = _timed("pose_camera_forward", lib.xvr_pose_camera_forward, _ptr(rot_c), _ptr(xyz_c), B, axes_of(convention),
                    _ptr(G), _ptr(c), _ptr(cam), _stream())
        _lib.check(rc, "xvr_pose_camera_forward")
        ctx.save_for_backward(rot_c, xyz_c, G)
        ctx.convention = convention
        return cam

    @staticmethod
    def backward(ctx, g_cam):
        lib = _lib.load()
        rot_c, xyz_c, G = ctx.saved_tensors
        B = rot_c.shape[0]
        g_rot, g_xyz = torch.empty_like(rot_c), torch.empty_like(xyz_c)
        rc = _timed("pose_camera_backward", lib.xvr_pose_camera_backward, _ptr(rot_c), _ptr(xyz_c), B,
                    axes_of(ctx.convention), _ptr(G), _ptr(g_cam.contiguous()), _ptr(g_rot), _ptr(g_xyz), _stream())
        _lib.check(rc, "xvr_pose_camera_backward")
        return g_rot, g_xyz, None, None, None


def pose_camera(rot, xyz, G, c, convention="ZXY"):
    for name, t in (("rot", rot), ("xyz", xyz), ("G", G), ("c", c)):
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be a float32 CUDA tensor (HIP kernel, no CPU path)")
    if rot.shape != xyz.shape or rot.dim() != 2 or rot.shape[1] != 3:
        raise ValueError("rot and xyz must both be [B, 3]")
    return PoseCamera.apply(rot, xyz, G.contiguous(), c.contiguous(), convention)


class RegistrationStage:
    """One pyramid stage on the device.  ``drr``: the (already rescaled) DRR module; ``sim``: a
    ``FusedSimilarity`` built on the transformed target of this stage; ``rot``/``xyz``: [B,3] float32 CUDA
    tensors, updated IN PLACE."""

    def __init__(self, drr, sim, rot, xyz, convention="ZXY", lr_rot=1e-2, lr_xyz=1.0, patience=10, threshold=1e-4,
                 max_n_plateaus=3, max_iters=500, factor=0.1, betas=(0.9, 0.999), eps=1e-8, maximize=True,
                 parameterization="euler_angles"):
        self.lib = _lib.load()
        if self.lib.xvr_pose_opt_state_bytes() != STATE_DTYPE.itemsize or ctypes.sizeof(_lib.CPoseOptState) != STATE_DTYPE.itemsize:
            raise _lib.HipLibraryError("xvr_pose_opt_state layout mismatch between the library and the binding")
        if parameterization not in PARAM_KINDS:
            raise ValueError(f"the device-resident loop knows {sorted(PARAM_KINDS)}, not {parameterization!r}")
        # any parameterisation the reference's Registration takes (/root/reference/src/xvr/registrar/base.py:168-169): Euler angles
        # through the closed-form pose -> camera pair, the others through xvr_pose_camera_forward_param / xvr_pose_opt_step_param
        self.kind, self.k = PARAM_KINDS[parameterization]
        dev = drr.density.device
        for name, t, cols in (("rot", rot, self.k), ("xyz", xyz, 3)):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 2 or t.shape[1] != cols:
                raise RuntimeError(f"{name} must be a contiguous float32 CUDA tensor [B, {cols}] (HIP kernels, no CPU path)")
        self.drr, self.sim, self.rot, self.xyz = drr, sim, rot, xyz
        self.B = B = rot.shape[0]
        self.H, self.W = drr.detector.height, drr.detector.width
        # `sim`: a FusedSimilarity (one C-ABI call per iteration), or any callable raw DRRs [B,1,H,W] -> similarity [B]
        # built from differentiable torch ops (a GeneralSimilarity: Equalize, sigma > 0, patches > 15 ...), which is then
        # differentiated by autograd between the render and the optimiser step -- still no host sync, still capturable
        # ... or an EqualizedSimilarity: a chain of five HIP calls (`evaluate`), no tape either
        self.sim_kind = "chain" if hasattr(sim, "evaluate") else ("fused" if hasattr(sim, "fixed_sobel") else "general")
        if tuple(sim.fixed.shape) != (B, 1, self.H, self.W):
            raise ValueError(f"similarity target {tuple(sim.fixed.shape)} does not match {B} poses at {self.H}x{self.W}")
        n = self.n = self.H * self.W
        self.G, self.c = drr.camera_affine()
        self.axes = axes_of(convention if self.kind == 0 else "ZXY")
        self.hist_cols = self.k + 6
        self.spec = _lib.CPoseOptSpec(self.axes, betas[0], betas[1], eps, int(bool(maximize)), factor, int(patience),
                                      float(threshold), 1e-8, int(max_n_plateaus), int(max_iters))
        self.max_iters = int(max_iters)
        self.rspec = drr.renderer.make_spec()
        self._vol_version = None
        self._bind_volume()
        self.render_fn = (self.lib.xvr_drr_trilinear_forward_camera if self.rspec.renderer == "trilinear"
                          else self.lib.xvr_drr_siddon_forward_camera)
        f = dict(device=dev, dtype=torch.float32)
        self.cam, self.g_cam = torch.empty(B, 24, **f), torch.zeros(B, 24, **f)
        nbytes = self.lib.xvr_drr_jac_to_camera_workspace_bytes(B, self.H, self.W)
        self.j2c_ws = torch.zeros((nbytes + 3) // 4, **f)   # zero-filled once; every call leaves it ready for the next
        self.img, self.g_img = torch.empty(B, 1, self.H, self.W, **f), torch.empty(B, 1, self.H, self.W, **f)
        self.jac = torch.empty(B, n, _lib.JAC_STRIDE, **f)
        self.loss = torch.empty(B, **f)
        self.history = torch.zeros(B, self.max_iters, self.hist_cols, **f)
        self.pose_jac = torch.empty(self.lib.xvr_pose_convert_jacobian_floats(B), **f) if self.kind else None
        self.state = torch.zeros(B * STATE_DTYPE.itemsize, device=dev, dtype=torch.uint8)
        _lib.check(self.lib.xvr_pose_opt_init(_ptr(self.state), B, float(lr_rot), float(lr_xyz), _stream()),
                   "xvr_pose_opt_init")
        self.graph, self.graph_len = None, 1

    def _bind_volume(self):
        """The render-ready copy of the (static) volume a launch of this size marches, as the autograd path chooses it
        (renderers._RenderFromCamera): the tiled y-pair copy for trilinear launches of >= 2048 wavefronts, the bricked copy for Siddon
        where it serves; built once per volume version, at first sight -- a registration renders the same volume hundreds of times.
        (One 512^2 pose is in the latency regime, where the layout does not matter: 184-194 us on either, alternating runs,
        profiles/r06_small_batch_tiles.txt; eight starts as one batch gain 2 %: 0.241 -> 0.236 ms per pose-iteration.)"""
        from . import renderers as R
        vol = self.drr.density
        if self._vol_version == (vol.data_ptr(), vol._version):
            return
        pairs, layout = None, 0
        if R._use_ypairs(self.rspec, vol, self.B, self.n):
            pairs, layout = R._layout_copy(self.lib, vol, "ypairs", first_sight=True), (3 if R.YPAIR_TILES else 1)
        elif R._use_bricks(self.rspec, vol, self.B, self.n):
            pairs, layout = R._brick_volume(self.lib, vol), 2
        self.vol_render = pairs if pairs is not None else vol
        self.cspec = make_cspec(tuple(vol.shape), self.rspec, self.W, volume_layout=layout if pairs is not None else 0)
        if self._vol_version is not None:
            self.graph = None        # (the captured launches hold the old copy's pointer)
        self._vol_version = (vol.data_ptr(), vol._version)

    # -- the five calls --------------------------------------------------------------------------
    def camera(self):
        lib, B, s = self.lib, self.B, _stream()
        if self.kind == 0:
            _lib.check(_timed("pose_camera_forward", lib.xvr_pose_camera_forward, _ptr(self.rot), _ptr(self.xyz), B, self.axes,
                              _ptr(self.G), _ptr(self.c), _ptr(self.cam), s), "xvr_pose_camera_forward")
        else:
            _lib.check(_timed("pose_camera_forward", lib.xvr_pose_camera_forward_param, _ptr(self.rot), _ptr(self.xyz), B, self.kind, self.axes,
                              _ptr(self.G), _ptr(self.c), _ptr(self.cam), _ptr(self.pose_jac), s), "xvr_pose_camera_forward_param")

    def render(self, camera: bool = True):
        lib, B, H, W, n, s = self.lib, self.B, self.H, self.W, self.n, _stream()
        vol = self.drr.density
        if camera:
            self.camera()
        _lib.check(_timed(f"{self.rspec.renderer}_forward+jac", self.render_fn, _ptr(self.vol_render), None, *vol.shape, 1, _ptr(self.cam), B, H, W,
                          ctypes.byref(self.cspec), _ptr(self.img), _ptr(self.jac), None, s),
                   f"xvr_drr_{self.rspec.renderer}_forward_camera")

    @property
    def fused_tail(self) -> bool:
        """Euler angles + the fused similarity: the iteration is render -> xvr_sim_ncc_registration_step (round 6) -- the
        similarity's last kernel also does jacobian -> camera, the optimiser step and the NEXT iteration's camera vector; bit for
        bit the five-call sequence (tests/test_pose_opt.py), three launches fewer.  FUSED_TAIL = False: the five calls (A/B)."""
        return FUSED_TAIL and self.kind == 0 and self.sim_kind == "fused" and not getattr(self.sim.spec, "pre_transformed", 0)

    def iteration(self):
        lib, B, H, W, n = self.lib, self.B, self.H, self.W, self.n
        if self.fused_tail:
            # (the camera vector in place belongs to the current (rot, xyz): written by run() before its first iteration and by
            #  every step's tail afterwards)
            self.render(camera=False)
            s, sim = _stream(), self.sim
            _lib.check(_timed("ncc_registration_step", lib.xvr_sim_ncc_registration_step, _ptr(sim.fixed), _ptr(sim.fixed_sobel), _ptr(self.img),
                              B, H, W, ctypes.byref(sim.spec), _ptr(self.loss), _ptr(self.g_img), _ptr(sim.workspace), sim.workspace.numel() * 4,
                              _ptr(self.jac), _ptr(self.cam), _ptr(self.j2c_ws), self.j2c_ws.numel() * 4, _ptr(self.rot), _ptr(self.xyz),
                              ctypes.byref(self.spec), _ptr(self.G), _ptr(self.c), _ptr(self.state), _ptr(self.history),
                              int(getattr(sim, "_ws_armed", False)), s),
                       "xvr_sim_ncc_registration_step")
            # (every xvr_sim_ncc_* call leaves its tickets at zero: from the second call on this workspace the header's reset is not
            #  launched -- the graph is captured at the third iteration, i.e. with the flag set)
            sim._ws_armed = True
            return
        self.render()
        s, sim = _stream(), self.sim
        if self.sim_kind == "chain":
            sim.evaluate(self.img, self.loss, self.g_img)
        elif self.sim_kind == "fused":
            _lib.check(_timed("ncc_forward_backward", lib.xvr_sim_ncc_forward_backward, _ptr(sim.fixed), _ptr(sim.fixed_sobel),
                              _ptr(self.img), B, H, W, ctypes.byref(sim.spec), _ptr(self.loss), _ptr(self.g_img),
                              _ptr(sim.workspace), sim.workspace.numel() * 4, s), "xvr_sim_ncc_forward_backward")
        else:
            with torch.enable_grad():
                img = self.img.detach().requires_grad_(True)
                loss = sim(img)
                (g,) = torch.autograd.grad(loss.sum(), img)
            self.loss.copy_(loss.detach().reshape(B))
            self.g_img.copy_(g)
        _lib.check(_timed("jac_to_camera_backward", lib.xvr_drr_jac_to_camera_backward, _ptr(self.jac), _ptr(self.g_img),
                          _ptr(self.cam), B, H, W, _ptr(self.g_cam), _ptr(self.j2c_ws), self.j2c_ws.numel() * 4, s),
                   "xvr_drr_jac_to_camera_backward")
        _lib.check(_timed("pose_opt_step", lib.xvr_pose_opt_step_param, _ptr(self.rot), _ptr(self.xyz), B, self.kind,
                          ctypes.byref(self.spec), _ptr(self.G), _ptr(self.pose_jac), _ptr(self.g_cam), _ptr(self.loss), _ptr(self.state),
                          _ptr(self.history), s), "xvr_pose_opt_step_param")

    # -- host control --------------------------------------------------------------------------------
    def read_state(self) -> np.ndarray:
        """Device -> host copy of the optimiser state (the one sync of a chunk of iterations)."""
        return np.frombuffer(self.state.cpu().numpy().tobytes(), dtype=STATE_DTYPE)

    def capture(self, iterations: int = 1):
        """Two eager iterations (real ones) to warm allocations and module loading, then capture ``iterations`` of them in ONE graph:
        a replay costs 10-16 us of host + launch whatever it holds, which is a tenth of a 256^2 iteration once the tail is fused --
        the block of iterations between two looks at the `done` flag goes out as one replay (iterations past the stopping rule are
        no-ops on the device, so a block is exactly the loop that checks after every step)."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(max(int(iterations), 1)):
                self.iteration()
        self.graph, self.graph_len = g, max(int(iterations), 1)

    def run(self, n_itr: int, check_every: int = 8, use_graph: bool = True):
        """Up to ``n_itr`` iterations (never more than ``max_iters`` in total); stops at the first check
        after every pose is done.  Returns (state, seconds per iteration of each chunk as a list)."""
        n_itr = min(int(n_itr), self.max_iters - int(self.read_state()["iter"].max()))
        self._bind_volume()   # (a volume changed in place since the last run gets a fresh copy, and a fresh graph)
        if self.fused_tail:
            self.camera()   # (rot / xyz may have been set from outside since the last step; outside the captured iteration)
        times, taken = [], 0
        st = self.read_state()
        while taken < n_itr and not st["done"].all():
            k = min(check_every, n_itr - taken)
            t0 = time.perf_counter()
            if use_graph and self.graph is None and taken >= 2:
                try:
                    self.capture(check_every)   # capture enqueues nothing: the replays below are the iterations
                except RuntimeError as e:
                    # capture is an optimisation, never a requirement -- but only for the general similarity path (autograd and
                    # torch ops inside it may refuse a capture); the all-HIP iteration must capture, and anything that is not a
                    # capture error is a bug to be seen.  iteration() keeps no host-side state, so a discarded capture has
                    # advanced nothing.
                    if self.sim_kind != "general":
                        raise
                    import warnings

                    warnings.warn(f"RegistrationStage: HIP-graph capture failed ({e}); continuing eagerly", RuntimeWarning, stacklevel=2)
                    self.graph, use_graph = None, False
                    torch.cuda.synchronize()
            if self.graph is not None and k == self.graph_len:
                self.graph.replay()          # the whole block of iterations in one replay
                taken += k
            else:                            # (the first two iterations, a last block shorter than the graph, or no graph at all)
                for _ in range(k if (self.graph is not None or not use_graph) else min(k, 2 - taken)):
                    self.iteration()
                    taken += 1
            before = st["iter"].copy()
            st = self.read_state()
            done_now = int((st["iter"] - before).max())
            dt = time.perf_counter() - t0
            times += [dt / max(done_now, 1)] * done_now
        return st, times

    def results(self):
        """history rows of every pose that were actually written: list over poses of arrays [iters, k + 6] = (the k rotation
        parameters and the translation after the update, the similarity before it, lr_rot, lr_xyz)."""
        st = self.read_state()
        h = self.history.cpu().numpy()
        return [h[b, : int(st["iter"][b])] for b in range(self.B)]
