"""Box-filter NCC family (xvr_amd.metrics) against the literal unfold formulation (oracle)."""
import pytest
import torch

from oracle import metrics_restated as ref
from xvr_amd import metrics


def _pair(seed=0, b=2, h=40, w=36):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.rand(b, 1, h, w, generator=g) * 3 - 0.5
    x2 = 0.6 * x1 + 0.4 * torch.rand(b, 1, h, w, generator=g)
    x2[:, :, :8, :8] = 0.25  # a flat region: variance ~ 0, eps decides
    return x1, x2


@pytest.mark.gpu
@pytest.mark.parametrize("patch", [None, 5, 9])
def test_ncc_matches_unfold_formulation(patch):
    x1, x2 = _pair()
    got = metrics.NormalizedCrossCorrelation2d(patch)(x1.cuda(), x2.cuda()).cpu()
    want = ref.ncc(x1.double(), x2.double(), patch).float()
    assert torch.allclose(got, want, atol=2e-5)


@pytest.mark.gpu
def test_multiscale_and_gradient_ncc_match_and_differentiate():
    x1, x2 = _pair(1)
    x1g, x2g = x1.cuda(), x2.cuda()
    m = metrics.MultiscaleNormalizedCrossCorrelation2d([None, 9], [0.5, 0.5])
    assert torch.allclose(m(x1g, x2g).cpu(), ref.multiscale_ncc(x1.double(), x2.double()).float(), atol=2e-5)
    for sigma in (0.0, 1.5):
        g = metrics.GradientNormalizedCrossCorrelation2d(11, sigma).cuda()
        assert torch.allclose(g(x1g, x2g).cpu(), ref.gradient_ncc(x1.double(), x2.double(), 11, sigma).float(), atol=2e-5)
    # gradients agree with autograd through the unfold formulation
    a = x2g.clone().requires_grad_(True)
    b = x2.double().clone().requires_grad_(True)
    (0.5 * m(x1g, a) + 0.5 * metrics.GradientNormalizedCrossCorrelation2d(11, 0.0).cuda()(x1g, a)).sum().backward()
    (0.5 * ref.multiscale_ncc(x1.double(), b) + 0.5 * ref.gradient_ncc(x1.double(), b, 11, 0.0)).sum().backward()
    assert torch.allclose(a.grad.cpu(), b.grad.float(), atol=1e-5 * b.grad.abs().max().item() + 1e-9)


def test_similarity_has_no_cpu_path():
    x1, x2 = _pair(2)
    for mod in (metrics.NormalizedCrossCorrelation2d(9), metrics.MultiscaleNormalizedCrossCorrelation2d([None, 9], [0.5, 0.5]),
                metrics.GradientNormalizedCrossCorrelation2d(11, 0.0)):
        with pytest.raises(RuntimeError, match="no CPU path"):
            mod(x1, x2)


@pytest.mark.gpu
def test_ncc_of_identical_images_is_one_and_of_negated_is_minus_one():
    x1, _ = _pair(2)
    x1 = x1.cuda()
    m = metrics.MultiscaleNormalizedCrossCorrelation2d([None, 9], [0.5, 0.5])
    assert torch.allclose(m(x1, x1).cpu(), torch.ones(2), atol=1e-3)
    assert torch.allclose(m(x1, -x1).cpu(), -torch.ones(2), atol=1e-3)
    assert torch.allclose(m(x1, 2.5 * x1 + 0.7).cpu(), torch.ones(2), atol=1e-3)  # affine invariance


@pytest.mark.gpu
@pytest.mark.parametrize("equalize", [False, True])
def test_xray_transforms_match_reference_lines(equalize):
    """XrayTransforms (HIP Standardize / Equalize / Normalize around torch's Resize) against the oracle's torch lines."""
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 1, 64, 64, generator=g) * 900
    t = metrics.XrayTransforms(16, equalize=equalize)
    assert torch.allclose(t(x.cuda()).cpu(), ref.xray_transforms(x, 16, equalize_=equalize), atol=2e-5)
    same = metrics.XrayTransforms(64, equalize=equalize)(x.cuda()).cpu()
    assert torch.allclose(same, ref.xray_transforms(x, 64, equalize_=equalize), atol=2e-5)
    if not equalize:   # the torch expression evaluated on the same device: the same bits
        xg = x.cuda()
        assert torch.equal(same, (((xg - xg.min()) / (xg.max() - xg.min() + 1e-6) - 0.15) / 0.1).cpu())


def test_xray_transforms_have_no_cpu_path():
    with pytest.raises(RuntimeError, match="no CPU path"):
        metrics.XrayTransforms(16)(torch.rand(1, 1, 16, 16))


def test_double_geodesic():
    from xvr_amd.pose import convert

    a = convert(torch.tensor([[0.0, 0.0, 0.0]]), torch.tensor([[0.0, 800.0, 0.0]]), parameterization="euler_angles", convention="ZXY")
    b = convert(torch.tensor([[0.1, 0.0, 0.0]]), torch.tensor([[0.0, 800.0, 0.0]]), parameterization="euler_angles", convention="ZXY")
    ang, tr, dbl = metrics.DoubleGeodesicSE3(1020.0)(a, b)
    assert abs(ang.item() - 0.1 * 510.0) < 1e-2
    assert abs(tr.item() - 2 * 800.0 * torch.sin(torch.tensor(0.05)).item()) < 1e-2
    assert abs(dbl.item() - (ang.item() ** 2 + tr.item() ** 2) ** 0.5) < 1e-3


def test_oracle_equalize_matches_its_definition_and_flattens_the_histogram():
    """oracle/metrics_restated.py::equalize (src/xvr/utils/preprocess.py:34-66, one image at a time) against the same formula
    vectorised over the batch; the HIP kernels are held to it in tests/test_pose_opt.py."""
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, 24, 20, generator=g) ** 3  # skewed towards 0
    y = ref.equalize(x, n_bins=64, tau=0.02)
    assert y.shape == x.shape and y.min() >= -1e-6 and y.max() <= 1 + 1e-6
    B = 2
    bins = torch.linspace(0, 1, 64)[None, None]
    diff = x.view(B, -1, 1) - bins
    w = (-diff.square() / (2 * 0.02**2)).exp()
    hist = w.sum(dim=1)
    hist = hist / (hist.sum(dim=1, keepdim=True) + 1e-10)
    cdf = torch.cumsum(hist, dim=1)
    cdfn = (cdf - cdf[:, 0:1]) / (1 - cdf[:, 0:1] + 1e-10)
    want = ((w / (w.sum(dim=-1, keepdim=True) + 1e-10)) * cdfn[:, None]).sum(dim=-1).view(B, 1, 24, 20)
    assert torch.allclose(y, want, atol=1e-6)
    # equalisation spreads the values: the median moves towards 0.5
    assert abs(y.median().item() - 0.5) < abs(x.median().item() - 0.5)
    assert torch.isfinite(ref.xray_transforms(x * 100, 24, 20, equalize_=True)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("per_image", [False, True])
@pytest.mark.parametrize("shape", [(5, 1, 24, 20), (3, 1, 64, 64), (2, 1, 7, 12)])
def test_fused_standardize_normalize_equals_the_torch_lines(shape, per_image):
    """xvr_sim_transform_forward / _backward (XrayTransforms without Equalize / Resize as two HIP calls) against the torch lines:
    the values bit for bit, the gradient -- including the part that flows through the min and the max, shared by tied pixels --
    to float32 summation noise."""
    from xvr_amd.metrics import XrayTransforms

    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(*shape, generator=g) * 3.0 + 0.5
    x[:, :, 0, :3] = 0.0                      # three pixels tie for the minimum of every image (air)
    x[0, 0, 1, 1] = x[0, 0, 2, 2] = 9.0       # two tie for the batch's maximum
    w = torch.randn(*shape, generator=g)
    tf = XrayTransforms(shape[2], shape[3], per_image=per_image)
    xx = x.clone().cuda().requires_grad_()
    y = tf(xx)
    (y * w.cuda()).sum().backward()
    y1, g1 = y.detach().cpu(), xx.grad.cpu()
    xo = x.clone().cuda().requires_grad_()       # the oracle's torch lines on the same device
    yo = ref.xray_transforms(xo, shape[2], shape[3], per_image=per_image)
    (yo * w.cuda()).sum().backward()
    y0, g0 = yo.detach().cpu(), xo.grad.cpu()
    assert torch.equal(y1, y0)
    scale = g0.abs().max()
    assert (g1 - g0).abs().max() <= 2e-5 * scale, ((g1 - g0).abs().max(), scale)
    # float64 autograd on the CPU: the fused backward is at least as close to it as the float32 torch chain
    xd = x.double().requires_grad_()
    (ref.xray_transforms(xd, shape[2], shape[3], per_image=per_image) * w.double()).sum().backward()
    assert (g1.double() - xd.grad).abs().max() <= 2.0 * (g0.double() - xd.grad).abs().max() + 1e-6 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("per_image", [False, True])
def test_fused_standardize_normalize_backward_runs_twice(per_image):
    """retain_graph / autograd.grad called twice: the second backward over the same forward state must give the gradient of ITS
    grad_output (the deterministic reduction's ticket counter was only reset by the forward)."""
    from xvr_amd.metrics import XrayTransforms

    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 1, 32, 28, generator=g) * 2.0 + 0.1
    x[:, :, 0, :2] = 0.0
    w1, w2 = torch.randn(4, 1, 32, 28, generator=g), torch.randn(4, 1, 32, 28, generator=g)
    tf = XrayTransforms(32, 28, per_image=per_image)
    res = {}
    for fused in (True, False):
        xx = x.clone().cuda().requires_grad_()
        y = tf(xx) if fused else ref.xray_transforms(xx, 32, 28, per_image=per_image)
        ga, = torch.autograd.grad(y, xx, w1.cuda(), retain_graph=True)
        gb, = torch.autograd.grad(y, xx, w2.cuda(), retain_graph=True)
        gc, = torch.autograd.grad(y, xx, w1.cuda())
        res[fused] = (ga.cpu(), gb.cpu(), gc.cpu())
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max()
    assert torch.equal(res[True][0], res[True][2])      # the same grad_output twice: the same bits
    assert (res[True][0] - res[True][1]).abs().max() > 1e-3
