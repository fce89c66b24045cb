"""Device-resident pose arithmetic of the registration loop (include/xvr_pose.h) against the torch chain it
replaces: convert -> DRR.camera, autograd, torch.optim.Adam(maximize=True), ReduceLROnPlateau and the
stopping rule of /root/reference/src/xvr/registrar/base.py:221-280."""
import ctypes

import numpy as np
import pytest
import torch

from xvr_amd import _lib
from xvr_amd.data import make_phantom, read
from xvr_amd.drr import DRR
from xvr_amd.pose import convert


def _drr(device, size=24, det=32, **kw):
    vol, _ = make_phantom(size, n_ellipsoids=4, seed=3, device=device)
    return DRR(read(vol, spacing=(2.0, 2.5, 3.0), orientation=kw.pop("orientation", "AP")), 1020.0, det, 2.0, x0=3.0, y0=-2.0,
               renderer="trilinear", **kw).to(device)


def test_camera_is_affine_in_the_pose_matrix():
    drr = _drr("cpu")
    G, c = drr.camera_affine()
    pose = convert(torch.tensor([[0.3, -0.2, 0.5], [3.0, 0.1, -0.4]]), torch.tensor([[10.0, 700.0, -20.0], [-5.0, 850.0, 12.0]]),
                   parameterization="euler_angles", convention="ZXY")
    cam = drr.camera(pose)
    lin = pose.matrix[:, :3, :4].reshape(2, 12) @ G.T + c
    assert torch.allclose(cam, lin, rtol=1e-6, atol=1e-4)


def test_state_layout_matches_header():
    from xvr_amd.pose_opt import STATE_DTYPE
    assert ctypes.sizeof(_lib.CPoseOptState) == STATE_DTYPE.itemsize == 144
    for name, _ in _lib.CPoseOptState._fields_:
        assert getattr(_lib.CPoseOptState, name).offset == STATE_DTYPE.fields[name][1], name


@pytest.mark.gpu
@pytest.mark.parametrize("convention", ["ZXY", "ZYX", "XYZ", "ZXZ", "YXY"])
@pytest.mark.parametrize("orientation", ["AP", "PA"])
def test_pose_camera_matches_torch_chain(convention, orientation):
    from xvr_amd.pose_opt import pose_camera
    drr = _drr("cuda", orientation=orientation, reverse_x_axis=(orientation == "AP"))
    g = torch.Generator().manual_seed(5)
    rot = ((torch.rand(7, 3, generator=g) - 0.5) * 6.0).cuda().requires_grad_()
    xyz = ((torch.rand(7, 3, generator=g) - 0.5) * 100 + torch.tensor([0.0, 800.0, 0.0])).cuda().requires_grad_()
    w = torch.randn(7, 24, generator=g).cuda()
    ref = drr.camera(convert(rot, xyz, parameterization="euler_angles", convention=convention))
    g_ref = torch.autograd.grad((ref * w).sum(), [rot, xyz])
    G, c = drr.camera_affine()
    cam = pose_camera(rot, xyz, G, c, convention)
    g_hip = torch.autograd.grad((cam * w).sum(), [rot, xyz])
    # fp32 with entries up to ~1e3: a few ulp of the largest terms
    assert torch.allclose(cam, ref, rtol=2e-6, atol=2e-4), (cam - ref).abs().max()
    for a, b in zip(g_hip, g_ref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-3 * b.abs().max().item()), (a - b).abs().max()


@pytest.mark.gpu
def test_opt_step_matches_adam_and_plateau_scheduler():
    """Same synthetic camera gradients and losses through the kernel and through torch.optim.Adam(maximize) +
    ReduceLROnPlateau + the reference's plateau counting."""
    from xvr_amd.pose_opt import STATE_DTYPE, axes_of
    lib = _lib.load()
    drr = _drr("cuda")
    G, c = drr.camera_affine()
    B, T, patience, max_pl = 3, 60, 2, 3
    g = torch.Generator().manual_seed(11)
    rot0 = (torch.rand(B, 3, generator=g) - 0.5)
    xyz0 = (torch.rand(B, 3, generator=g) - 0.5) * 50 + torch.tensor([0.0, 800.0, 0.0])
    gcams = torch.randn(T, B, 24, generator=g) * torch.logspace(-3, 0, 24)
    # losses: rise, then stall (plateaus), different per pose
    losses = torch.stack([torch.minimum(torch.arange(T) * 0.01, torch.tensor(0.05 * (b + 1))) + 1e-6 * torch.randn(T, generator=g)
                          for b in range(B)], dim=1).float()

    # --- torch: one optimiser + scheduler per pose (the reference loop is B = 1)
    ref_rows, ref_iters = [], []
    for b in range(B):
        rot = rot0[b:b + 1].clone().cuda().requires_grad_()
        xyz = xyz0[b:b + 1].clone().cuda().requires_grad_()
        opt = torch.optim.Adam([{"params": [rot], "lr": 1e-2}, {"params": [xyz], "lr": 1.0}], maximize=True)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.1, patience=patience, threshold=1e-4, mode="max")
        n_pl, cur, rows = 0, float("inf"), []
        for t in range(T):
            opt.zero_grad()
            cam = drr.camera(convert(rot, xyz, parameterization="euler_angles", convention="ZXY"))
            (cam * gcams[t, b:b + 1].cuda()).sum().backward()
            opt.step()
            sch.step(losses[t, b].item())
            lr = sch.get_last_lr()
            rows.append(torch.cat([rot.detach(), xyz.detach()], 1).reshape(-1).cpu().tolist() + [losses[t, b].item(), *lr])
            if lr[0] < cur:
                cur, n_pl = lr[0], n_pl + 1
            if n_pl == max_pl:
                break
        ref_rows.append(np.array(rows))
        ref_iters.append(len(rows))

    # --- HIP
    rot, xyz = rot0.clone().cuda(), xyz0.clone().cuda()
    spec = _lib.CPoseOptSpec(axes_of("ZXY"), 0.9, 0.999, 1e-8, 1, 0.1, patience, 1e-4, 1e-8, max_pl, T)
    state = torch.zeros(B * STATE_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    hist = torch.zeros(B, T, _lib.POSE_HISTORY_COLS, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.xvr_pose_opt_init(P(state), B, 1e-2, 1.0, None) == 0
    for t in range(T):
        gc = gcams[t].clone().cuda()
        rc = lib.xvr_pose_opt_step(P(rot), P(xyz), B, ctypes.byref(spec), P(G), P(gc), P(losses[t].cuda().contiguous()), P(state),
                                   P(hist), None)
        assert rc == 0, lib.xvr_drr_last_error()
        assert float(gc.abs().max()) == 0.0        # consumed
    st = np.frombuffer(state.cpu().numpy().tobytes(), dtype=STATE_DTYPE)
    assert st["iter"].tolist() == ref_iters
    assert st["done"].tolist() == [int(n < T) for n in ref_iters]
    assert len(set(ref_iters)) > 1                 # the poses really stopped at different times
    h = hist.cpu().numpy()
    for b in range(B):
        got, want = h[b, : ref_iters[b]], ref_rows[b]
        np.testing.assert_allclose(got[:, 6:], want[:, 6:], rtol=1e-6)                 # loss, lr_rot, lr_xyz: exact up to fp32 print
        np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=0, atol=2e-5)         # angles (steps of 1e-2)
        np.testing.assert_allclose(got[:, 3:6], want[:, 3:6], rtol=0, atol=2e-3)       # mm (steps of 1)
        assert np.abs(h[b, ref_iters[b]:]).max(initial=0.0) == 0.0                      # nothing written after `done`
    # parameters frozen after done
    np.testing.assert_array_equal(torch.cat([rot, xyz], 1).cpu().numpy(), np.stack([h[b, ref_iters[b] - 1, :6] for b in range(B)]))


@pytest.mark.gpu
@pytest.mark.parametrize("renderer", ["trilinear", "siddon"])
def test_device_loop_follows_autograd_loop(renderer):
    """The whole stage on the device vs the autograd + torch.optim loop (same kernels for render and
    similarity): same trajectory for the first iterations, same optimum."""
    from xvr_amd.metrics import DoubleGeodesicSE3
    from xvr_amd.registrar import Registrar
    vol, _ = make_phantom(48, n_ellipsoids=8, seed=8, device="cuda")
    drr = DRR(read(vol, spacing=(2.5,) * 3, orientation="AP"), 1020.0, 96, 1.8, renderer=renderer, reverse_x_axis=False,
              voxel_shift=0.0 if renderer == "trilinear" else 0.5).cuda()
    true = convert(torch.tensor([[3.10, 0.05, -0.03]]), torch.tensor([[4.0, 700.0, -6.0]]), parameterization="euler_angles", convention="ZXY")
    init = convert(torch.tensor([[3.18, 0.0, 0.02]]), torch.tensor([[-6.0, 715.0, 5.0]]), parameterization="euler_angles", convention="ZXY")
    with torch.no_grad():
        gt = drr(convert(torch.tensor([[3.10, 0.05, -0.03]]).cuda(), torch.tensor([[4.0, 700.0, -6.0]]).cuda(),
                         parameterization="euler_angles", convention="ZXY"))
    kw = dict(scales="2,1", n_itrs="40,30", patience=5, max_n_plateaus=2)
    a = Registrar(drr, device_loop=True, check_every=5, **kw).run(gt, init)
    b = Registrar(drr, device_loop=False, use_graph=False, **kw).run(gt, init)
    ta, tb = np.array(a["trajectory"]), np.array(b["trajectory"])
    # Siddon's pose gradient is piecewise (nearest-voxel lookups): ulp differences grow quickly, so only the
    # first steps are compared pointwise there
    k = min(10 if renderer == "trilinear" else 3, len(ta), len(tb))
    np.testing.assert_allclose(ta[:k, :3], tb[:k, :3], atol=2e-3)     # Adam steps are lr-sized: 1e-2 rad, 1 mm
    np.testing.assert_allclose(ta[:k, 3:], tb[:k, 3:], atol=0.2)
    np.testing.assert_allclose(a["nccs"][:k], b["nccs"][:k], atol=2e-3)
    geo = DoubleGeodesicSE3(1020.0)
    ea, eb = geo(true, a["final_pose"].cpu())[2].item(), geo(true, b["final_pose"].cpu())[2].item()
    e0 = geo(true, init)[2].item()
    # (Siddon's end point moves by millimetres from run to run of the SAME loop: the similarity's double
    #  accumulators are filled by atomics in arbitrary order and the piecewise gradient amplifies it)
    assert ea < 0.25 * e0 and eb < 0.25 * e0 and (renderer == "siddon" or abs(ea - eb) < 2.0), (e0, ea, eb)
    assert len(a["trajectory"]) + 1 == len(a["nccs"]) and len(a["times"]) == len(a["nccs"]) == len(a["lrs"])


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(96, 96), (33, 47), (5, 3)])
def test_jac_to_camera_equals_from_jac_then_rays_backward(hw):
    """The fused, fixed-order kernel against the two-kernel path it replaces (autograd through DRR.forward)."""
    from xvr_amd.drr import rays_from_camera
    lib = _lib.load()
    H, W = hw
    vol, _ = make_phantom(32, n_ellipsoids=6, seed=2, device="cuda")
    drr = DRR(read(vol, spacing=(3.0,) * 3, orientation="AP"), 1020.0, H, 2.0 * 64 / max(H, W), width=W, renderer="trilinear",
              reverse_x_axis=False).cuda()
    pose = convert(torch.tensor([[3.1, 0.1, -0.05], [2.9, -0.1, 0.1]]).cuda(), torch.tensor([[3.0, 700.0, -5.0], [-8.0, 650.0, 4.0]]).cuda(),
                   parameterization="euler_angles", convention="ZXY")
    cam = drr.camera(pose).detach().requires_grad_()
    source, target, raylen = rays_from_camera(cam, H, W)
    img = drr.renderer(drr.density, source, target, raylen)
    g = torch.Generator().manual_seed(3)
    w = torch.rand(img.shape, generator=g).cuda()
    (img * w).sum().backward()
    want = cam.grad.clone()
    # the same jacobian through the C ABI
    from xvr_amd.renderers import make_cspec
    B, n = 2, H * W
    spec = drr.renderer.make_spec()
    cs = make_cspec(tuple(drr.density.shape), spec, W)
    out, jac = torch.empty(B, 1, n, device="cuda"), torch.empty(B, n, 8, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    src_c, tgt_c, len_c = source.detach().reshape(B, 3).contiguous(), target.detach().contiguous(), raylen.detach().reshape(B, n).contiguous()
    assert lib.xvr_drr_trilinear_forward(P(drr.density), None, *drr.density.shape, 1, P(src_c), P(tgt_c), P(len_c), B, n,
                                         ctypes.byref(cs), P(out), P(jac), None, None) == 0
    ws = torch.zeros((lib.xvr_drr_jac_to_camera_workspace_bytes(B, H, W) + 3) // 4, device="cuda")
    got = []
    for _ in range(3):   # the workspace is reusable without re-zeroing, and the result is the same bits every time
        g_cam = torch.full((B, 24), float("nan"), device="cuda")
        rc = lib.xvr_drr_jac_to_camera_backward(P(jac), P(w.reshape(B, n).contiguous()), P(cam.detach().contiguous()), B, H, W,
                                                P(g_cam), P(ws), ws.numel() * 4, None)
        assert rc == 0, lib.xvr_drr_last_error()
        got.append(g_cam.clone())
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])
    assert torch.allclose(got[0], want, rtol=2e-4, atol=2e-4 * want.abs().max().item()), (got[0] - want).abs().max()


@pytest.mark.gpu
def test_device_registration_is_bitwise_reproducible():
    """Every reduction on the device loop adds in a fixed order: two runs give identical trajectories."""
    from xvr_amd.registrar import Registrar
    vol, _ = make_phantom(64, n_ellipsoids=8, seed=8, device="cuda")
    drr = DRR(read(vol, spacing=(2.0,) * 3, orientation="AP"), 1020.0, 128, 1.4, renderer="trilinear", reverse_x_axis=False,
              voxel_shift=0.0).cuda()
    with torch.no_grad():
        gt = drr(convert(torch.tensor([[3.10, 0.05, -0.03]]).cuda(), torch.tensor([[4.0, 700.0, -6.0]]).cuda(),
                         parameterization="euler_angles", convention="ZXY"))
    init = convert(torch.tensor([[3.2, 0.0, 0.04]]), torch.tensor([[-8.0, 720.0, 6.0]]), parameterization="euler_angles", convention="ZXY")
    runs = [Registrar(drr, scales="2,1", n_itrs="30,20", patience=4, max_n_plateaus=2, device_loop=True).run(gt, init) for _ in range(3)]
    for r in runs[1:]:
        assert r["trajectory"] == runs[0]["trajectory"]
        assert r["nccs"][:-1] == runs[0]["nccs"][:-1]      # (the last entry is the torch-metrics re-evaluation)
        assert torch.equal(r["final_pose"].matrix, runs[0]["final_pose"].matrix)


@pytest.mark.gpu
@pytest.mark.parametrize("starts", [1, 3], ids=["one start", "three starts as a batch"])
def test_fused_iteration_tail_is_the_five_call_iteration_bit_for_bit(starts, monkeypatch):
    """Round 6: for Euler angles + the fused similarity an iteration is render -> ONE call,
    xvr_sim_ncc_registration_step, whose last kernel also contracts the image gradient with the render's jacobian, takes the
    optimiser step and writes the next camera vector (include/xvr_sim.h).  It shares every expression and every reduction order with
    the calls it replaces (csrc/j2c_device.hiph, pose_device.hiph): the trajectories -- poses, similarities, learning rates, every
    iteration of both pyramid stages, eager and graph-replayed -- are IDENTICAL, not close."""
    from xvr_amd import pose_opt
    from xvr_amd.registrar import Registrar
    vol, _ = make_phantom(64, n_ellipsoids=8, seed=8, device="cuda")
    drr = DRR(read(vol, spacing=(2.0,) * 3, orientation="AP"), 1020.0, 128, 1.4, renderer="trilinear", reverse_x_axis=False,
              voxel_shift=0.0).cuda()
    rot0, xyz0 = torch.tensor([[3.10, 0.05, -0.03]]), torch.tensor([[4.0, 700.0, -6.0]])
    with torch.no_grad():
        gt = drr(convert(rot0.cuda(), xyz0.cuda(), parameterization="euler_angles", convention="ZXY"))
    g = torch.Generator().manual_seed(2)
    inits = convert(rot0 + (torch.rand(starts, 3, generator=g) - 0.5) * 0.2, xyz0 + (torch.rand(starts, 3, generator=g) - 0.5) * 30.0,
                    parameterization="euler_angles", convention="ZXY")
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(pose_opt, "FUSED_TAIL", fused)
        reg = Registrar(drr, scales="2,1", n_itrs="30,20", patience=4, max_n_plateaus=2, device_loop=True)
        out[fused] = reg.run_batch(gt, inits) if starts > 1 else [reg.run(gt, inits)]
    for a, b in zip(out[True], out[False]):
        assert len(a["trajectory"]) > 10
        assert a["trajectory"] == b["trajectory"] and a["nccs"] == b["nccs"] and a["lrs"] == b["lrs"]
        assert torch.equal(a["final_pose"].matrix, b["final_pose"].matrix)
    # the tail really ran (and only where it applies): one timed call instead of four
    from xvr_amd import renderers
    monkeypatch.setattr(pose_opt, "FUSED_TAIL", True)
    renderers.PROFILER = []
    Registrar(drr, scales="1", n_itrs="6", max_n_plateaus=100, device_loop=True, use_graph=False).run(gt, inits[0])
    names = {e[0] for e in renderers.PROFILER}
    renderers.PROFILER = None
    assert "ncc_registration_step" in names and not names & {"ncc_forward_backward", "jac_to_camera_backward", "pose_opt_step"}, names
    renderers.PROFILER = []
    Registrar(drr, scales="1", n_itrs="6", max_n_plateaus=100, device_loop=True, use_graph=False, parameterization="axis_angle").run(gt, inits[0])
    names = {e[0] for e in renderers.PROFILER}
    renderers.PROFILER = None
    assert "ncc_registration_step" not in names and {"ncc_forward_backward", "jac_to_camera_backward", "pose_opt_step"} <= names, names


@pytest.mark.gpu
def test_batched_multistart_matches_one_by_one():
    """Registrar.run_batch: B starts as independent problems in one batch (own optimiser state, own
    standardisation) follow the same trajectories as B separate runs."""
    from xvr_amd.registrar import Registrar
    vol, _ = make_phantom(64, n_ellipsoids=8, seed=8, device="cuda")
    drr = DRR(read(vol, spacing=(2.0,) * 3, orientation="AP"), 1020.0, 128, 1.4, renderer="trilinear", reverse_x_axis=False,
              voxel_shift=0.0).cuda()
    rot0, xyz0 = torch.tensor([[3.10, 0.05, -0.03]]), torch.tensor([[4.0, 700.0, -6.0]])
    with torch.no_grad():
        gt = drr(convert(rot0.cuda(), xyz0.cuda(), parameterization="euler_angles", convention="ZXY"))
    g = torch.Generator().manual_seed(1)
    drot, dxyz = (torch.rand(3, 3, generator=g) - 0.5) * 0.2, (torch.rand(3, 3, generator=g) - 0.5) * 30.0
    inits = convert(rot0 + drot, xyz0 + dxyz, parameterization="euler_angles", convention="ZXY")
    reg = Registrar(drr, scales="2,1", n_itrs="30,20", patience=4, max_n_plateaus=2, device_loop=True)
    batch = reg.run_batch(gt, inits)
    assert len(batch) == 3
    lengths = set()
    for b in range(3):
        single = reg.run(gt, inits[b])
        tb, ts = np.array(batch[b]["trajectory"]), np.array(single["trajectory"])
        k = min(8, len(tb), len(ts))
        # (the batch renders with another kernel variant than a single pose: last-bit differences, which Adam's
        #  normalised steps amplify along the poorly conditioned source-detector axis -- steps are 1 mm there)
        np.testing.assert_allclose(tb[:k, :3], ts[:k, :3], atol=2e-3)
        np.testing.assert_allclose(tb[:k, 3:], ts[:k, 3:], atol=1.0)
        np.testing.assert_allclose(batch[b]["nccs"][:k], single["nccs"][:k], atol=2e-3)
        assert batch[b]["nccs"][-1] > 0.9 * single["nccs"][-1]
        assert len(batch[b]["times"]) == len(batch[b]["nccs"]) == len(batch[b]["lrs"]) == len(batch[b]["trajectory"]) + 1
        lengths.add(len(tb))
    # determinism carries over to the batch
    again = reg.run_batch(gt, inits)
    assert all(a["trajectory"] == b["trajectory"] for a, b in zip(again, batch))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 40, 36), (2, 64, 64)])
def test_fused_similarity_per_image_equals_one_image_at_a_time(shape):
    from xvr_amd.similarity import FusedSimilarity
    B, H, W = shape
    g = torch.Generator().manual_seed(7)
    fixed = torch.rand(B, 1, H, W, generator=g).cuda()
    moving = (torch.rand(B, 1, H, W, generator=g) * torch.tensor([1.0, 3.0, 0.5][:B]).reshape(B, 1, 1, 1)).cuda().requires_grad_()
    w = torch.tensor([1.0, -2.0, 0.5][:B]).cuda()
    sim = FusedSimilarity(fixed, per_image=True)
    loss = sim(moving)
    (loss * w).sum().backward()
    for b in range(B):
        mb = moving[b:b + 1].detach().clone().requires_grad_()
        lb = FusedSimilarity(fixed[b:b + 1])(mb)
        (lb * w[b]).sum().backward()
        assert torch.allclose(loss[b], lb[0], rtol=1e-6, atol=1e-7)
        assert torch.allclose(moving.grad[b], mb.grad[0], rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [2, 7, 40])
def test_fused_pose_loss_terms_match_torch(B):
    """Double geodesic and multiview consistency of the training loss (xvr_pose_geodesic, xvr_pose_multiview_*)
    against the torch restatement of the reference's loss (oracle/loss_restated.py): values and gradients w.r.t. the predicted pose."""
    from oracle import loss_restated as oloss
    from xvr_amd import loss as loss_mod
    from xvr_amd.training import get_random_pose

    g = torch.Generator().manual_seed(5)
    true = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, B, generator=g).cuda()
    rot, xyz = true.convert("euler_angles", "ZXY")
    res = {}
    for fused in (True, False):
        r = (rot + 0.05 * torch.randn(B, 3, generator=torch.Generator().manual_seed(1)).cuda()).requires_grad_()
        t = (xyz + 4.0 * torch.randn(B, 3, generator=torch.Generator().manual_seed(2)).cuda()).requires_grad_()
        pred = convert(r, t, parameterization="euler_angles", convention="ZXY")
        img = torch.rand(B, 1, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()
        mask = torch.rand(B, 3, 32, 32, generator=torch.Generator().manual_seed(4)).cuda() > 0.5
        if fused:
            L = loss_mod.PoseRegressionLoss(1020.0, weight_mvc=0.5).cuda()
            loss, mncc, dgeo, rgeo, tgeo, dice, mvc = L(img, mask, true, img * 0.9 + 0.01, mask, pred)
        else:
            loss, mncc, dgeo, rgeo, tgeo, dice, mvc = oloss.pose_regression_loss(img, mask.float(), true.matrix, img * 0.9 + 0.01, mask.float(),
                                                                                 pred.matrix, 1020.0, weight_mvc=0.5)
        loss.mean().backward()
        res[fused] = (loss.detach(), dgeo.detach(), rgeo.detach(), tgeo.detach(), mvc.detach(), r.grad.clone(), t.grad.clone())
    names = ("loss", "dgeo", "rgeo", "tgeo", "mvc", "d/drot", "d/dxyz")
    for a, b, name in zip(res[True], res[False], names):
        assert a.shape == b.shape, name
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-4 * max(b.abs().max().item(), 1e-6)), (name, (a - b).abs().max(), b.abs().max())
    # identical poses: zero distance, finite (zero) gradient
    r = rot.clone().requires_grad_()
    pred = convert(r, xyz, parameterization="euler_angles", convention="ZXY")
    L = loss_mod.PoseRegressionLoss(1020.0)
    mv = L.multiview_consistency(true, pred)
    mv.sum().backward()
    assert float(mv.detach().max()) < 0.05 and torch.isfinite(r.grad).all()


@pytest.mark.gpu
def test_run_batch_with_one_target_per_pose():
    """B different X-rays of one geometry registered in one batch: each equals its own single run."""
    from xvr_amd.registrar import Registrar
    vol, _ = make_phantom(64, n_ellipsoids=8, seed=8, device="cuda")
    drr = DRR(read(vol, spacing=(2.0,) * 3, orientation="AP"), 1020.0, 96, 1.8, renderer="trilinear", reverse_x_axis=False,
              voxel_shift=0.0).cuda()
    rots = torch.tensor([[3.10, 0.05, -0.03], [3.20, -0.04, 0.02]])
    xyzs = torch.tensor([[4.0, 700.0, -6.0], [-5.0, 690.0, 3.0]])
    with torch.no_grad():
        gts = drr(convert(rots.cuda(), xyzs.cuda(), parameterization="euler_angles", convention="ZXY"))
    inits = convert(rots + 0.05, xyzs + 6.0, parameterization="euler_angles", convention="ZXY")
    reg = Registrar(drr, scales="2,1", n_itrs="25,15", patience=4, max_n_plateaus=2)
    batch = reg.run_batch(gts, inits)
    for b in range(2):
        single = reg.run(gts[b:b + 1], inits[b])
        k = min(6, len(batch[b]["trajectory"]), len(single["trajectory"]))
        np.testing.assert_allclose(np.array(batch[b]["trajectory"])[:k, :3], np.array(single["trajectory"])[:k, :3], atol=2e-3)
        np.testing.assert_allclose(batch[b]["nccs"][:k], single["nccs"][:k], atol=2e-3)
        assert abs(batch[b]["nccs"][-1] - single["nccs"][-1]) < 0.05 and batch[b]["nccs"][-1] > batch[b]["nccs"][0]
    with pytest.raises(ValueError):
        reg.run_batch(gts, convert(rots[:1].repeat(3, 1), xyzs[:1].repeat(3, 1), parameterization="euler_angles", convention="ZXY"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(8)))
def test_fuzz_opt_step_against_torch_adam(seed):
    """Randomised optimiser specs (Euler convention, Adam betas / eps, ascent or descent, plateau factor / patience /
    threshold, learning rates) on random camera gradients and loss curves, against torch.optim.Adam +
    ReduceLROnPlateau stepped pose by pose."""
    from xvr_amd.pose_opt import STATE_DTYPE, axes_of
    lib = _lib.load()
    rng = np.random.default_rng(900 + seed)
    drr = _drr("cuda", orientation=str(rng.choice(["AP", "PA"])))
    G, c = drr.camera_affine()
    conv = str(rng.choice(["ZXY", "XYZ", "ZYX", "ZXZ", "YZX"]))
    B, T = int(rng.integers(1, 5)), 40
    b1, b2, eps = float(rng.uniform(0.7, 0.95)), float(rng.uniform(0.9, 0.9999)), float(10 ** rng.uniform(-9, -6))
    maximize = bool(rng.random() < 0.5)
    factor, patience, thr = float(rng.uniform(0.1, 0.7)), int(rng.integers(0, 4)), float(10 ** rng.uniform(-5, -2))
    lr_rot, lr_xyz = float(10 ** rng.uniform(-3, -1)), float(10 ** rng.uniform(-1, 0.5))
    max_pl = int(rng.integers(2, 5))
    g = torch.Generator().manual_seed(seed)
    rot0 = (torch.rand(B, 3, generator=g) - 0.5) * 2.0
    xyz0 = (torch.rand(B, 3, generator=g) - 0.5) * 50 + torch.tensor([0.0, 800.0, 0.0])
    gcams = torch.randn(T, B, 24, generator=g) * torch.logspace(-3, 0, 24)
    losses = torch.stack([torch.cumsum(torch.randn(T, generator=g) * 0.01 + (0.004 if maximize else -0.004) * (torch.arange(T) < 12 + 5 * b), 0)
                          for b in range(B)], dim=1).float()
    mode = "max"   # the reference's scheduler watches a similarity it maximises (base.py:229-235)
    ref_rows = []
    for b in range(B):
        rot = rot0[b:b + 1].clone().cuda().requires_grad_()
        xyz = xyz0[b:b + 1].clone().cuda().requires_grad_()
        opt = torch.optim.Adam([{"params": [rot], "lr": lr_rot}, {"params": [xyz], "lr": lr_xyz}], betas=(b1, b2), eps=eps, maximize=maximize)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=factor, patience=patience, threshold=thr, mode=mode)
        n_pl, cur, rows = 0, float("inf"), []
        for t in range(T):
            opt.zero_grad()
            cam = drr.camera(convert(rot, xyz, parameterization="euler_angles", convention=conv))
            (cam * gcams[t, b:b + 1].cuda()).sum().backward()
            opt.step()
            sch.step(losses[t, b].item())
            lr = sch.get_last_lr()
            rows.append(torch.cat([rot.detach(), xyz.detach()], 1).reshape(-1).cpu().tolist() + [losses[t, b].item(), *lr])
            if lr[0] < cur:
                cur, n_pl = lr[0], n_pl + 1
            if n_pl == max_pl:
                break
        ref_rows.append(np.array(rows))
    rot, xyz = rot0.clone().cuda(), xyz0.clone().cuda()
    spec = _lib.CPoseOptSpec(axes_of(conv), b1, b2, eps, int(maximize), factor, patience, thr, 1e-8, max_pl, T)
    state = torch.zeros(B * STATE_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    hist = torch.zeros(B, T, _lib.POSE_HISTORY_COLS, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.xvr_pose_opt_init(P(state), B, lr_rot, lr_xyz, None) == 0
    for t in range(T):
        gc = gcams[t].clone().cuda()
        assert lib.xvr_pose_opt_step(P(rot), P(xyz), B, ctypes.byref(spec), P(G), P(gc), P(losses[t].cuda().contiguous()), P(state), P(hist), None) == 0
    st = np.frombuffer(state.cpu().numpy().tobytes(), dtype=STATE_DTYPE)
    h = hist.cpu().numpy()
    what = f"seed {seed}: {conv} betas {b1:.3f},{b2:.4f} eps {eps:.1e} max {maximize} factor {factor:.2f} patience {patience} thr {thr:.1e}"
    for b in range(B):
        want = ref_rows[b]
        assert int(st["iter"][b]) == len(want), (what, b, int(st["iter"][b]), len(want))
        got = h[b, : len(want)]
        np.testing.assert_allclose(got[:, 6:], want[:, 6:], rtol=2e-6, err_msg=what)
        np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=0, atol=2e-3 * lr_rot * T + 1e-5, err_msg=what)
        np.testing.assert_allclose(got[:, 3:6], want[:, 3:6], rtol=0, atol=2e-3 * lr_xyz * T + 1e-3, err_msg=what)


# ----------------------------------------------------------------------------------------------
# similarity configurations beyond the single fused call: sigma > 0, Equalize
# ----------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 3), (2, 5, 7), (1, 4, 33), (3, 40, 36)])
@pytest.mark.parametrize("sigma", [0.6, 1.0, 2.5])
def test_gaussian_blur_kernels_match_torch_forward_and_adjoint(shape, sigma):
    """xvr_sim_gaussian_blur5 against the reflect-pad + two conv2d of the reference's Sobel pre-blur, and its adjoint against
    autograd through that torch formulation (every image size down to the 3 x 3 minimum of a 2-pixel reflection)."""
    from xvr_amd.metrics import Sobel
    from xvr_amd.similarity import gaussian_blur5

    B, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 1, H, W, generator=g)
    w = torch.rand(B, 1, H, W, generator=g)
    ref_mod = Sobel(sigma)
    xr = x.clone().requires_grad_(True)
    ref = ref_mod._blur(xr)
    (ref * w).sum().backward()
    xh = x.cuda().requires_grad_(True)
    out = gaussian_blur5(xh, sigma)
    (out * w.cuda()).sum().backward()
    assert torch.allclose(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(xh.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("sigma", [0.8, 1.5])
def test_gradient_ncc_with_pre_blur_runs_in_hip_and_matches_torch_and_the_oracle(sigma, monkeypatch):
    from oracle import metrics_restated as mref
    from xvr_amd import metrics, renderers

    B, H, W = 2, 48, 40
    g = torch.Generator().manual_seed(13)
    x = torch.rand(B, 1, H, W, generator=g).cuda().requires_grad_()
    y = (0.6 * x.detach() + 0.4 * torch.rand(B, 1, H, W, generator=g).cuda()).requires_grad_()
    w = torch.rand(B, generator=g).cuda() + 0.5
    sim = metrics.GradientNormalizedCrossCorrelation2d(11, sigma).cuda()
    renderers.PROFILER = []
    out = sim(x, y)
    names = {e[0] for e in renderers.PROFILER}
    renderers.PROFILER = None
    assert {"gaussian_blur5", "mncc_forward_backward"} <= names          # the HIP kernels did run
    (out * w).sum().backward()
    gx, gy = x.grad.clone(), y.grad.clone()
    x.grad = y.grad = None
    monkeypatch.setattr(metrics.GradientNormalizedCrossCorrelation2d, "FUSED", False)
    ref = sim(x, y)
    (ref * w).sum().backward()
    assert torch.allclose(out, ref, rtol=2e-5, atol=2e-6), (out - ref).abs().max()
    for a, b, name in ((gx, x.grad, "d/dx"), (gy, y.grad, "d/dy")):
        assert (a - b).abs().max() <= 2e-4 * b.abs().max(), (name, (a - b).abs().max(), b.abs().max())
    # the literal unfold formulation (oracle), value AND gradient
    xo, yo = x.detach().cpu().double().requires_grad_(True), y.detach().cpu().double().requires_grad_(True)
    oref = mref.gradient_ncc(xo, yo, 11, sigma)
    (oref * w.cpu().double()).sum().backward()
    assert torch.allclose(out.detach().cpu().double(), oref.detach(), atol=5e-5)
    assert (gx.cpu().double() - xo.grad).abs().max() <= 2e-3 * xo.grad.abs().max()
    assert (gy.cpu().double() - yo.grad).abs().max() <= 2e-3 * yo.grad.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 40, 36), (2, 64, 64)])
@pytest.mark.parametrize("beta", [0.5, 0.2])
def test_fused_similarity_gradient_matches_autograd_through_the_oracle(shape, beta):
    """The fused call's GRADIENT w.r.t. the raw DRR, directly against autograd through the literal oracle formulation
    (XrayTransforms restated line by line + the unfold-based patch NCC family, oracle/metrics_restated.py) in float64."""
    from oracle import metrics_restated as mref
    from xvr_amd.metrics import XrayTransforms
    from xvr_amd.similarity import FusedSimilarity

    B, H, W = shape
    g = torch.Generator().manual_seed(41)
    fixed_raw = torch.rand(B, 1, H, W, generator=g) * 40
    # (a unique minimum and maximum: the oracle's x.min() / x.max() then have one sub-gradient, like the kernels')
    moving = 0.7 * fixed_raw + 12 * torch.rand(B, 1, H, W, generator=g)
    fixed = mref.xray_transforms(fixed_raw, H, W)
    sim = FusedSimilarity(fixed.cuda(), 9, 11, beta)
    mv = moving.cuda().requires_grad_(True)
    loss = sim(mv)
    loss.sum().backward()
    mo = moving.double().requires_grad_(True)
    yo = mref.xray_transforms(mo, H, W)
    oref = beta * mref.multiscale_ncc(fixed.double(), yo) + (1 - beta) * mref.gradient_ncc(fixed.double(), yo, 11, 0.0)
    oref.sum().backward()
    assert torch.allclose(loss.cpu().double(), oref.detach(), atol=5e-5)
    err = (mv.grad.cpu().double() - mo.grad).abs().max() / mo.grad.abs().max()
    assert err <= 2e-3, err


@pytest.mark.gpu
@pytest.mark.parametrize("beta", [1.0, 0.5])
def test_fused_similarity_on_drr_like_images_with_a_flat_background(beta):
    """A DRR is exactly constant over most of its background.  On a flat patch the reference's two-pass (unfold) patch NCC
    is exactly 0; a one-pass variance in float32 about a tile-wide shift leaves delta^2 * 1e-7 there, which is NOT small
    against eps = 1e-5 -- round 2's kernel was off by 2-7 % of an image's patch NCC on such images (found by
    tests/test_c4_c5.py; random-noise test images have no flat patches).  The moments are summed in doubles now."""
    from oracle import metrics_restated as mref
    from xvr_amd.metrics import XrayTransforms
    from xvr_amd.similarity import FusedSimilarity

    B, H, W = 3, 64, 64
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")

    def blobs(shift):   # the same three discs per image, moved by `shift` pixels: a "moving" image close to the fixed one
        g = torch.Generator().manual_seed(43)
        img = torch.zeros(B, 1, H, W)
        for b in range(B):
            for _ in range(3):
                cy, cx, r = (torch.rand(3, generator=g) * torch.tensor([H * 0.5, W * 0.5, 9.0]) + torch.tensor([H * 0.25 + shift, W * 0.25, 5.0])).tolist()
                img[b, 0] += 20.0 * torch.clamp(1.0 - ((yy - cy) ** 2 + (xx - cx) ** 2) / r ** 2, min=0.0)
        return img
    fixed_raw, moving = blobs(0.0), blobs(1.5)
    assert (fixed_raw == 0).float().mean() > 0.4 and (moving == 0).float().mean() > 0.4      # mostly exactly-flat background
    moving[:, :, 3, 5] += 31.0   # (a unique maximum, see the test above)
    fixed = mref.xray_transforms(fixed_raw, H, W)
    sim = FusedSimilarity(fixed.cuda(), 9, 11, beta)
    mv = moving.cuda().requires_grad_(True)
    loss = sim(mv)
    loss.sum().backward()
    mo = moving.double().requires_grad_(True)
    yo = mref.xray_transforms(mo, H, W)
    oref = beta * mref.multiscale_ncc(fixed.double(), yo) + (1 - beta) * mref.gradient_ncc(fixed.double(), yo, 11, 0.0)
    oref.sum().backward()
    assert torch.allclose(loss.cpu().double(), oref.detach(), atol=1e-4), (loss.cpu(), oref)
    err = (mv.grad.cpu().double() - mo.grad).abs().max() / mo.grad.abs().max()
    assert err <= 5e-3, err


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(sigma=1.0), dict(equalize=True), dict(sigma=0.8, equalize=True)], ids=["sigma", "equalize", "both"])
def test_device_loop_and_run_batch_take_every_similarity_configuration(kw):
    """sigma > 0 and Equalize used to be refused by the device-resident loop and by run_batch (torch fallback only): now
    the stage differentiates a GeneralSimilarity (HIP NCC + blur kernels, torch transforms) between the render and the
    optimiser step.  The device loop must follow the plain autograd loop, and the batched starts their one-by-one runs."""
    from xvr_amd.registrar import Registrar
    vol, _ = make_phantom(64, n_ellipsoids=8, seed=8, device="cuda")
    drr = DRR(read(vol, spacing=(2.0,) * 3, orientation="AP"), 1020.0, 64, 2.8, renderer="trilinear", reverse_x_axis=False,
              voxel_shift=0.0).cuda()
    rot0, xyz0 = torch.tensor([[3.10, 0.05, -0.03]]), torch.tensor([[4.0, 700.0, -6.0]])
    with torch.no_grad():
        gt = drr(convert(rot0.cuda(), xyz0.cuda(), parameterization="euler_angles", convention="ZXY"))
    g = torch.Generator().manual_seed(2)
    drot, dxyz = (torch.rand(2, 3, generator=g) - 0.5) * 0.16, (torch.rand(2, 3, generator=g) - 0.5) * 24.0
    inits = convert(rot0 + drot, xyz0 + dxyz, parameterization="euler_angles", convention="ZXY")
    common = dict(scales="2,1", n_itrs="25,15", patience=4, max_n_plateaus=2, **kw)
    dev = Registrar(drr, device_loop=True, **common)
    ref = Registrar(drr, device_loop=False, fused=False, use_graph=False, **common)
    a, b = dev.run(gt, inits[0]), ref.run(gt, inits[0])
    k = min(6, len(a["trajectory"]), len(b["trajectory"]))
    np.testing.assert_allclose(np.array(a["nccs"][:k]), np.array(b["nccs"][:k]), atol=3e-3)
    np.testing.assert_allclose(np.array(a["trajectory"])[:k, :3], np.array(b["trajectory"])[:k, :3], atol=3e-3)
    assert a["nccs"][-1] > a["nccs"][0] + 0.02
    batch = dev.run_batch(gt, inits)
    assert len(batch) == 2
    for n in range(2):
        single = dev.run(gt, inits[n])
        k = min(6, len(batch[n]["trajectory"]), len(single["trajectory"]))
        np.testing.assert_allclose(np.array(batch[n]["nccs"][:k]), np.array(single["nccs"][:k]), atol=3e-3)
        assert batch[n]["nccs"][-1] > batch[n]["nccs"][0]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 40, 36), (3, 64, 64), (2, 17, 129)])
def test_equalize_hip_matches_the_torch_formulation_value_and_gradient(shape):
    """xvr_sim_equalize_forward / _backward against the line-by-line torch restatement of the reference's Equalize
    (/root/reference/src/xvr/utils/preprocess.py:34-66, [pixels x bins] weight matrix and autograd), incl. an image with
    large flat regions (many pixels on the same value, as a standardised DRR has)."""
    from oracle import metrics_restated as mref
    from xvr_amd import metrics, renderers

    B, H, W = shape
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 1, H, W, generator=g)
    x[:, :, : H // 3] = 0.0                      # background at exactly 0
    x[:, :, -2:] = 1.0                           # and a saturated border
    w = torch.rand(B, 1, H, W, generator=g)
    eq = metrics.Equalize().cuda()
    xh = x.cuda().requires_grad_(True)
    renderers.PROFILER = []
    out = eq(xh)
    assert "equalize_forward" in {e[0] for e in renderers.PROFILER}
    renderers.PROFILER = None
    (out * w.cuda()).sum().backward()
    xr = x.cuda().double().requires_grad_(True)      # the reference formulation in float64 on the same device
    ref = mref.equalize(xr)
    (ref * w.cuda().double()).sum().backward()
    assert torch.allclose(out.double(), ref.detach(), atol=2e-5), (out.double() - ref).abs().max()
    err = (xh.grad.double() - xr.grad).abs().max() / xr.grad.abs().max()
    assert err <= 2e-3, err


# ---------------------------------------------------------------------------------------------------------------
# round 4: the device-resident loop for every parameterisation (/root/reference/src/xvr/registrar/base.py:168-169)
# ---------------------------------------------------------------------------------------------------------------
NON_EULER = ["axis_angle", "quaternion", "quaternion_adjugate", "rotation_6d", "se3_log_map", "rotation_10d"]


@pytest.mark.gpu
@pytest.mark.parametrize("param", NON_EULER)
def test_param_camera_and_opt_step_match_torch_adam(param):
    """xvr_pose_camera_forward_param = drr.camera(convert(rot, xyz, param)); xvr_pose_opt_step_param = the chain rule through
    convert (forward-mode Jacobian) + torch.optim.Adam(maximize) with the reference's two parameter groups, over several steps."""
    from xvr_amd.pose_opt import PARAM_KINDS, STATE_DTYPE, axes_of
    lib = _lib.load()
    drr = _drr("cuda")
    G, c = drr.camera_affine()
    kind, k = PARAM_KINDS[param]
    B, T = 3, 12
    g = torch.Generator().manual_seed(21)
    start = convert((torch.rand(B, 3, generator=g) - 0.5), (torch.rand(B, 3, generator=g) - 0.5) * 50 + torch.tensor([0.0, 800.0, 0.0]),
                    parameterization="euler_angles", convention="ZXY")
    rot0, xyz0 = start.convert(param)
    assert rot0.shape == (B, k)
    gcams = torch.randn(T, B, 24, generator=g) * torch.logspace(-3, 0, 24)
    losses = (torch.arange(T).float()[:, None] * 0.01 + torch.zeros(T, B)).contiguous()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    # torch
    rot, xyz = rot0.clone().cuda().requires_grad_(), xyz0.clone().cuda().requires_grad_()
    opt = torch.optim.Adam([{"params": [rot], "lr": 1e-2}, {"params": [xyz], "lr": 1.0}], maximize=True)
    want_cam, rows = None, []
    for t in range(T):
        opt.zero_grad()
        cam = drr.camera(convert(rot, xyz, parameterization=param))
        if t == 0:
            want_cam = cam.detach().clone()
        (cam * gcams[t].cuda()).sum().backward()
        opt.step()
        rows.append(torch.cat([rot.detach(), xyz.detach()], 1).cpu())
    # HIP
    r, x = rot0.clone().cuda().contiguous(), xyz0.clone().cuda().contiguous()
    cam = torch.empty(B, 24, device="cuda")
    jac = torch.empty(lib.xvr_pose_convert_jacobian_floats(B), device="cuda")
    spec = _lib.CPoseOptSpec(axes_of("ZXY"), 0.9, 0.999, 1e-8, 1, 0.1, 100, 1e-4, 1e-8, 3, T)
    state = torch.zeros(B * STATE_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    hist = torch.zeros(B, T, k + 6, device="cuda")
    assert lib.xvr_pose_opt_init(P(state), B, 1e-2, 1.0, None) == 0
    for t in range(T):
        assert lib.xvr_pose_camera_forward_param(P(r), P(x), B, kind, axes_of("ZXY"), P(G), P(c), P(cam), P(jac), None) == 0, lib.xvr_drr_last_error()
        if t == 0:
            assert torch.allclose(cam, want_cam, rtol=1e-5, atol=1e-4 * float(want_cam.abs().max()))
        gc = gcams[t].clone().cuda()
        assert lib.xvr_pose_opt_step_param(P(r), P(x), B, kind, ctypes.byref(spec), P(G), P(jac), P(gc), P(losses[t].cuda()), P(state), P(hist),
                                           None) == 0, lib.xvr_drr_last_error()
        assert float(gc.abs().max()) == 0.0
        got = torch.cat([r, x], 1).cpu()
        np.testing.assert_allclose(got[:, :k].numpy(), rows[t][:, :k].numpy(), rtol=0, atol=3e-4)    # steps of 1e-2
        np.testing.assert_allclose(got[:, k:].numpy(), rows[t][:, k:].numpy(), rtol=0, atol=3e-2)    # steps of 1 mm
    h = hist.cpu().numpy()
    np.testing.assert_allclose(h[:, -1, :k + 3], torch.cat([r, x], 1).cpu().numpy())
    np.testing.assert_allclose(h[:, :, k + 3], losses.T.numpy(), rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("param", NON_EULER)
def test_device_loop_follows_autograd_loop_for_every_parameterisation(param):
    """Registrar(parameterization=...) on the device (pose -> camera with its Jacobian, Adam on k + 3 parameters) vs the
    autograd + torch.optim loop over the same kernels: same first iterations, same optimum, and bit-reproducible."""
    from xvr_amd.metrics import DoubleGeodesicSE3
    from xvr_amd.registrar import Registrar
    vol, _ = make_phantom(48, n_ellipsoids=8, seed=8, device="cuda")
    drr = DRR(read(vol, spacing=(2.5,) * 3, orientation="AP"), 1020.0, 96, 1.8, renderer="trilinear", reverse_x_axis=False,
              voxel_shift=0.0).cuda()
    true = convert(torch.tensor([[3.10, 0.05, -0.03]]), torch.tensor([[4.0, 700.0, -6.0]]), parameterization="euler_angles", convention="ZXY")
    init = convert(torch.tensor([[3.16, 0.01, 0.01]]), torch.tensor([[-4.0, 712.0, 3.0]]), parameterization="euler_angles", convention="ZXY")
    with torch.no_grad():   # (a pose is a module: .cuda() would move `true` itself)
        gt = drr(convert(torch.tensor([[3.10, 0.05, -0.03]]).cuda(), torch.tensor([[4.0, 700.0, -6.0]]).cuda(), parameterization="euler_angles",
                         convention="ZXY"))
    kw = dict(scales="2,1", n_itrs="40,30", patience=5, max_n_plateaus=2, parameterization=param, lr_rot=5e-3)
    a = Registrar(drr, device_loop=True, check_every=5, **kw).run(gt, init)
    a2 = Registrar(drr, device_loop=True, check_every=8, **kw).run(gt, init)
    b = Registrar(drr, device_loop=False, use_graph=False, **kw).run(gt, init)
    assert a["trajectory"] == a2["trajectory"] and torch.equal(a["final_pose"].matrix, a2["final_pose"].matrix)
    ta, tb = np.array(a["trajectory"]), np.array(b["trajectory"])      # logged as Euler ZXY whatever the parameterisation
    assert ta.shape[1] == 6 and tb.shape[1] == 6
    k = min(8, len(ta), len(tb))
    np.testing.assert_allclose(ta[:k, :3], tb[:k, :3], atol=3e-3)
    np.testing.assert_allclose(ta[:k, 3:], tb[:k, 3:], atol=0.3)
    np.testing.assert_allclose(a["nccs"][:k], b["nccs"][:k], atol=3e-3)
    geo = DoubleGeodesicSE3(1020.0)
    ea, eb, e0 = geo(true, a["final_pose"].cpu())[2].item(), geo(true, b["final_pose"].cpu())[2].item(), geo(true, init)[2].item()
    assert abs(ea - eb) < 0.05 * e0, (e0, ea, eb)             # the two loops end at the same place ...
    if param != "se3_log_map":                                # ... which is near the truth (a twist near the cut locus |omega| = pi
        assert ea < 0.3 * e0 and eb < 0.3 * e0, (e0, ea, eb)  #     moves slowly under Adam in EITHER loop: 67 of 70 mm left)
    assert a["nccs"][-2] > a["nccs"][0]
    assert len(a["trajectory"]) + 1 == len(a["nccs"]) and len(a["times"]) == len(a["nccs"]) == len(a["lrs"])
    # the batched multi-start takes the parameterisation too
    batch = Registrar(drr, device_loop=True, **kw).run_batch(gt, convert(torch.tensor([[3.16, 0.01, 0.01], [3.05, 0.08, -0.05]]),
                                                                          torch.tensor([[-4.0, 712.0, 3.0], [8.0, 690.0, -10.0]]),
                                                                          parameterization="euler_angles", convention="ZXY"))
    assert len(batch) == 2 and all(r["nccs"][-1] > r["nccs"][0] for r in batch)


@pytest.mark.gpu
@pytest.mark.parametrize("per_image", [False, True])
@pytest.mark.parametrize("beta", [0.5, 1.0])
def test_equalized_similarity_matches_the_oracle_value_and_gradient(per_image, beta):
    """EqualizedSimilarity (Standardize -> Equalize + Normalize -> NCC and back, five HIP calls, no tape) against float64
    autograd through the oracle's torch lines (XrayTransforms with Equalize, unfold NCC)."""
    from oracle import metrics_restated as mref
    from xvr_amd.similarity import EqualizedSimilarity

    B, H, W = 2, 40, 36
    g = torch.Generator().manual_seed(13)
    base = torch.rand(B, 1, H, W, generator=g)
    base[:, :, : H // 4] = 0.0                                       # a flat background, as a DRR has
    fixed_raw = base * 3.0
    moving = (base.roll(2, dims=-1) * 2.5 + 0.2 * torch.rand(B, 1, H, W, generator=g)) * (base.roll(2, dims=-1) > 0)
    tf = lambda x: torch.cat([mref.xray_transforms(x[b:b + 1], H, W, equalize_=True) for b in range(x.shape[0])]) if per_image \
        else mref.xray_transforms(x, H, W, equalize_=True)          # noqa: E731
    fixed = tf(fixed_raw.double()).float()
    sim = EqualizedSimilarity(fixed.cuda(), 9, 11, beta, per_image=per_image)
    m = moving.cuda().requires_grad_()
    w = torch.tensor([1.0, 1.0]) if not per_image else torch.tensor([0.7, 1.3])
    loss = sim(m)
    (loss * w.cuda()).sum().backward()
    mo = moving.double().requires_grad_()
    yo = tf(mo)
    oref = beta * mref.multiscale_ncc(fixed.double(), yo) + (1 - beta) * mref.gradient_ncc(fixed.double(), yo, 11, 0.0)
    (oref * w.double()).sum().backward()
    assert torch.allclose(loss.detach().cpu().double(), oref.detach(), atol=3e-4), (loss, oref)
    err = (m.grad.cpu().double() - mo.grad).abs().max() / mo.grad.abs().max()
    assert err < 2e-2, err
    # evaluate() twice over the same buffers: the same bits (fixed-order sums, re-entrant tickets)
    l2, g2 = torch.empty(B, device="cuda"), torch.empty(B, 1, H, W, device="cuda")
    l3, g3 = torch.empty(B, device="cuda"), torch.empty(B, 1, H, W, device="cuda")
    sim.evaluate(m.detach().contiguous(), l2, g2)
    sim.evaluate(m.detach().contiguous(), l3, g3)
    assert torch.equal(l2, l3) and torch.equal(g2, g3) and torch.equal(l2, loss.detach())
