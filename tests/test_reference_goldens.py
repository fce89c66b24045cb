"""Vectors produced by the REFERENCE's own files (src/xvr/model/sampler.py, src/xvr/model/loss.py), executed in
the build container over the diffdrr compat shim (tests/golden/make_golden_xvr.py).  They pin xvr's own
logic -- the pose sampling recipe, the Dice definition, the loss assembly and the multiview pairing."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = np.load(Path(__file__).resolve().parent / "golden" / "xvr_reference_sampler_loss.npz")


def test_get_random_pose_matches_the_reference_sampler():
    from xvr_amd.training import get_random_pose

    ranges = dict(zip((str(k) for k in GOLD["sampler_keys"]), GOLD["sampler_ranges"].tolist()))
    ranges["batch_size"] = int(ranges["batch_size"])
    torch.manual_seed(123)
    mine = get_random_pose(**ranges).matrix.numpy()
    assert np.allclose(mine, GOLD["sampler_matrix"], atol=1e-5)
    # sanity of the recipe itself: sources sit ty mm from the isocentre, looking back at it
    assert (np.linalg.norm(mine[:, :3, 3], axis=1) >= 450 - 1e-3).all()


def test_oracle_loss_matches_the_reference_loss_module():
    """oracle/loss_restated.py against the vectors the reference's own loss.py produced (tests/golden/make_golden_xvr.py)."""
    from oracle import loss_restated as oloss

    t = lambda k: torch.from_numpy(GOLD[k])  # noqa: E731
    res = oloss.pose_regression_loss(t("in_img"), t("in_mask").float(), t("in_pose"), t("in_pred_img"), t("in_pred_mask").float(),
                                     t("in_pred_pose"), 1020.0, weight_mvc=1e-3)
    for k, v in zip(("loss", "mncc", "dgeo", "rgeo", "tgeo", "dice", "mvc"), res):
        assert np.allclose(v.detach().numpy(), GOLD[f"loss_{k}"], rtol=1e-5, atol=1e-5), k
    d = oloss.dice_metric(t("in_mask").float(), t("in_pred_mask").float()).numpy()
    assert np.allclose(d, GOLD["dice_metric"], equal_nan=True)
    assert np.isnan(GOLD["dice_metric"][0, 1])  # the structure that is empty in both masks


@pytest.mark.gpu
def test_pose_regression_loss_matches_the_reference_loss_module():
    """The product's loss (HIP entry points) against the same vectors."""
    from xvr_amd.loss import DiceMetric, PoseRegressionLoss
    from xvr_amd.pose import RigidTransform

    t = lambda k: torch.from_numpy(GOLD[k]).cuda()  # noqa: E731
    fn = PoseRegressionLoss(1020.0, weight_mvc=1e-3)
    res = fn(t("in_img"), t("in_mask").bool(), RigidTransform(t("in_pose")), t("in_pred_img"), t("in_pred_mask").bool(),
             RigidTransform(t("in_pred_pose")))
    for k, v in zip(("loss", "mncc", "dgeo", "rgeo", "tgeo", "dice", "mvc"), res):
        assert np.allclose(v.detach().cpu().numpy(), GOLD[f"loss_{k}"], rtol=2e-4, atol=2e-4), k
    d = DiceMetric()(t("in_mask").bool(), t("in_pred_mask").bool()).cpu().numpy()
    assert np.allclose(d, GOLD["dice_metric"], equal_nan=True)


def test_compat_shim_exposes_the_diffdrr_names_xvr_imports():
    saved = {k: v for k, v in sys.modules.items() if k == "diffdrr" or k.startswith("diffdrr.")}
    try:
        from xvr_amd.compat import install_as_diffdrr

        assert install_as_diffdrr(force=True)
        from diffdrr.data import read, transform_hu_to_density  # noqa: F401
        from diffdrr.drr import DRR  # noqa: F401
        from diffdrr.metrics import (DoubleGeodesicSE3, GradientNormalizedCrossCorrelation2d,  # noqa: F401
                                     MultiscaleNormalizedCrossCorrelation2d)
        from diffdrr.pose import RigidTransform, convert, make_matrix  # noqa: F401
        from diffdrr.registration import N_ANGULAR_COMPONENTS, Registration  # noqa: F401

        import xvr_amd.drr

        assert DRR is xvr_amd.drr.DRR and N_ANGULAR_COMPONENTS["quaternion_adjugate"] == 10
    finally:
        for k in [k for k in sys.modules if k == "diffdrr" or k.startswith("diffdrr.")]:
            del sys.modules[k]
        sys.modules.update(saved)


REF = Path("/root/reference/src/xvr")


def _load_ref(relpath, name):
    import importlib.util

    import pytest

    if not (REF / relpath).exists():
        pytest.skip("the reference tree is not present on this machine")
    spec = importlib.util.spec_from_file_location(name, REF / relpath)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_registrar_defaults_equal_the_reference_config():
    """xvr_amd.registrar.Registrar's defaults are the reference's RegistrarArgs (src/xvr/config/registrar.py)."""
    import inspect

    from xvr_amd.registrar import Registrar

    ref = _load_ref("config/registrar.py", "xvr_ref_config_registrar").RegistrarArgs()
    mine = {k: v.default for k, v in inspect.signature(Registrar.__init__).parameters.items() if v.default is not inspect._empty}
    for key in ("scales", "n_itrs", "parameterization", "convention", "lr_rot", "lr_xyz", "patience", "threshold",
                "max_n_plateaus", "crop", "equalize"):
        assert mine[key] == getattr(ref, key), key
    tr = _load_ref("config/trainer.py", "xvr_ref_config_trainer").TrainerArgs()
    assert tr.renderer == "trilinear" and tr.batch_size == 116 and tr.parameterization == "quaternion_adjugate"


def test_reference_evaluator_runs_on_this_drr():
    """The reference's own Evaluator (src/xvr/metrics/evaluator.py: mPE, mRPE, mTRE, double geodesic) imports only
    diffdrr; over the compat shim it runs unmodified against this package's DRR (CPU-side geometry only)."""
    saved = {k: v for k, v in sys.modules.items() if k == "diffdrr" or k.startswith("diffdrr.")}
    try:
        from xvr_amd.compat import install_as_diffdrr

        install_as_diffdrr(force=True)
        ev_mod = _load_ref("metrics/evaluator.py", "xvr_ref_evaluator")
        from xvr_amd.data import make_phantom, read
        from xvr_amd.drr import DRR
        from xvr_amd.pose import convert

        vol, _ = make_phantom(16, seed=1)
        drr = DRR(read(vol, orientation="AP"), 1020.0, 32, 4.0, reverse_x_axis=False)
        fiducials = torch.tensor([[[10.0, -5.0, 3.0], [-8.0, 6.0, -2.0], [0.0, 0.0, 12.0]]])
        ev = ev_mod.Evaluator(drr, fiducials)
        true = convert(torch.tensor([[0.1, 0.05, -0.02]]), torch.tensor([[2.0, 700.0, -3.0]]), parameterization="euler_angles", convention="ZXY")
        same = ev(true, true)
        assert all(abs(x) < 1e-3 for x in same), same
        # a pure 5 mm shift along the detector's column direction: mTRE = 5 mm exactly, mPE ~ 5 mm x magnification
        moved = convert(torch.tensor([[0.1, 0.05, -0.02]]), torch.tensor([[2.0, 700.0, 2.0]]), parameterization="euler_angles", convention="ZXY")
        mpe, mrpe, mtre, dgeo = ev(true, moved)
        assert abs(mtre - 5.0) < 1e-3 and abs(dgeo - 5.0) < 1e-2
        assert 5.0 < mpe < 5.0 * 1020.0 / 600.0 and mrpe > 0
    finally:
        for k in [k for k in sys.modules if k == "diffdrr" or k.startswith("diffdrr.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_reference_initialize_drr_runs_over_the_shim(tmp_path):
    """The reference's own DRR factory (src/xvr/renderer/load.py: read(volume, mask, labels, orientation) ->
    DRR(subject, sdd, height, delx, width, dely, x0, y0, reverse_x_axis=, renderer=, **drr_kwargs).to(device))
    runs unmodified over the compat shim on NIfTI files, and hands back this package's DRR with the surface xvr
    touches afterwards (trainer.py:251-256, registrar/base.py:155-212)."""
    import struct

    saved = {k: v for k, v in sys.modules.items() if k == "diffdrr" or k.startswith("diffdrr.")}
    try:
        from xvr_amd.compat import install_as_diffdrr

        install_as_diffdrr(force=True)
        load = _load_ref("renderer/load.py", "xvr_ref_renderer_load")

        def write_nifti(path, data, affine, dtype):
            hdr = bytearray(348)
            struct.pack_into("<i", hdr, 0, 348)
            struct.pack_into("<8h", hdr, 40, 3, *data.shape, 1, 1, 1, 1)
            struct.pack_into("<h", hdr, 70, {"<i2": 4, "<f4": 16}[dtype])
            struct.pack_into("<h", hdr, 72, {"<i2": 16, "<f4": 32}[dtype])
            struct.pack_into("<8f", hdr, 76, 1.0, *[float(np.linalg.norm(affine[:3, i])) for i in range(3)], 0, 0, 0, 0)
            struct.pack_into("<f", hdr, 108, 352.0)
            struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
            struct.pack_into("<2h", hdr, 252, 0, 1)           # qform 0, sform 1
            for r in range(3):
                struct.pack_into("<4f", hdr, 280 + 16 * r, *[float(x) for x in affine[r]])
            hdr[344:348] = b"n+1\0"
            with open(path, "wb") as f:
                f.write(bytes(hdr) + b"\0" * 4 + np.asfortranarray(data.astype(dtype)).tobytes(order="F"))

        rng = np.random.default_rng(0)
        hu = (rng.random((12, 14, 10)) * 2000 - 1000).astype(np.float32)
        seg = rng.integers(0, 4, size=hu.shape).astype(np.float32)
        affine = np.diag([1.5, 1.5, 2.0, 1.0])
        affine[:3, 3] = [-9.0, -10.0, -10.0]
        write_nifti(tmp_path / "ct.nii", hu, affine, "<f4")
        write_nifti(tmp_path / "seg.nii", seg, affine, "<i2")
        drr = load.initialize_drr(str(tmp_path / "ct.nii"), str(tmp_path / "seg.nii"), "1,3", "AP", 20, 24, 1020.0, 2.0, 2.5,
                                  1.0, -2.0, True, "trilinear", read_kwargs={}, drr_kwargs={"voxel_shift": 0.0}, device="cpu")
        from xvr_amd.drr import DRR

        assert isinstance(drr, DRR) and drr.renderer.renderer_name == "trilinear" and drr.renderer.voxel_shift == 0.0
        assert (drr.detector.height, drr.detector.width, drr.detector.delx, drr.detector.dely) == (20, 24, 2.0, 2.5)
        assert tuple(drr.density.shape) == hu.shape and float(drr.density.min()) >= 0.0
        assert set(np.unique(drr.mask.numpy()).tolist()) <= {0.0, 1.0, 3.0}      # labels="1,3" keeps only those structures
        from xvr_amd.pose import convert

        pose = convert(torch.zeros(1, 3), torch.tensor([[0.0, 800.0, 0.0]]), parameterization="euler_angles", convention="ZXY")
        src, tgt = drr.detector(pose, None)
        assert src.shape == (1, 1, 3) and tgt.shape == (1, 20 * 24, 3)
        drr.rescale_detector_(0.5)
        assert (drr.detector.height, drr.detector.width) == (10, 12)
    finally:
        for k in [k for k in sys.modules if k == "diffdrr" or k.startswith("diffdrr.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_parse_scales_equals_the_reference_function():
    """registrar/base.py cannot be imported here (matplotlib, torchvision, ...), but its pure helper
    `_parse_scales` can be compiled on its own from the reference tree and compared call by call."""
    import ast

    import pytest

    path = REF / "registrar" / "base.py"
    if not path.exists():
        pytest.skip("the reference tree is not present on this machine")
    tree = ast.parse(path.read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_parse_scales")
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), str(path), "exec"), ns)
    from xvr_amd.registrar import parse_scales

    for scales, crop, height in (("8", 0, 2048), ("8,4", 0, 2048), ("24,12,6", 100, 1436), ("4,2,1", 50, 512)):
        want = ns["_parse_scales"](scales.split(","), crop, height)     # the reference passes the split list (base.py:152)
        assert parse_scales(scales, crop, height) == pytest.approx(want, rel=1e-12), (scales, crop, height)


def test_standardize_and_equalize_equal_the_reference_classes():
    """utils/preprocess.py imports torchvision (absent here), but its Standardize / Equalize classes are plain
    torch: compiled on their own from the reference tree and compared with the oracle's restatement (the checker of the HIP kernels,
    tests/test_pose_opt.py::test_equalize_hip_matches_the_torch_formulation_value_and_gradient) on random images."""
    import ast

    import pytest

    path = REF / "utils" / "preprocess.py"
    if not path.exists():
        pytest.skip("the reference tree is not present on this machine")
    tree = ast.parse(path.read_text())
    classes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("Standardize", "Identity", "Equalize")]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=classes, type_ignores=[]), str(path), "exec"), ns)
    from oracle import metrics_restated as mref

    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 1, 24, 20, generator=g) * 7.0 - 1.0
    std = ns["Standardize"]()(x)
    assert torch.allclose(mref.xray_transforms(x, 24, 20), (std - 0.15) / 0.1, rtol=1e-6, atol=1e-6)
    assert torch.allclose(mref.equalize(std), ns["Equalize"]()(std), rtol=1e-5, atol=1e-6)
    assert torch.allclose(mref.xray_transforms(x, 24, 20, equalize_=True), (ns["Equalize"]()(std) - 0.15) / 0.1, rtol=1e-5, atol=1e-5)


def test_reference_antipode_helper_puts_the_source_on_the_other_side():
    """model/inference.py:_construct_antipode (pure diffdrr.pose calls) compiled on its own over this package's
    pose module.  For a C-arm pose (xyz = (0, d, 0)) it must carry the source to the far side of the patient --
    (x, y, z) -> (x, -y, -z), a half turn about the left-right axis -- which only happens if the translation is
    applied in the camera frame, x_world = R (x_cam + t): the convention xvr_amd.pose.convert was pinned on
    (with x_world = R x_cam + t the source would not move at all).  Applying it twice is the identity."""
    import ast

    import pytest

    path = REF / "model" / "inference.py"
    if not path.exists():
        pytest.skip("the reference tree is not present on this machine")
    fn = next(n for n in ast.parse(path.read_text()).body if isinstance(n, ast.FunctionDef) and n.name == "_construct_antipode")
    from xvr_amd.pose import RigidTransform, convert

    ns = {"torch": torch, "RigidTransform": RigidTransform, "convert": convert}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), str(path), "exec"), ns)
    rot = torch.tensor([[0.4, 0.2, -0.1], [2.9, -0.3, 0.05]])
    xyz = torch.tensor([[0.0, 800.0, 0.0], [0.0, 650.0, 0.0]])
    pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    anti = ns["_construct_antipode"](pose)
    src, src_anti = pose.matrix[:, :3, 3], anti.matrix[:, :3, 3]          # camera origin in world coordinates
    assert torch.allclose(src_anti, src * torch.tensor([1.0, -1.0, -1.0]), atol=1e-3), (src, src_anti)
    assert torch.allclose(src.norm(dim=1), xyz[:, 1], atol=1e-3)           # the source orbits at distance d
    back = ns["_construct_antipode"](anti)
    assert torch.allclose(back.matrix, pose.matrix, atol=1e-4)
