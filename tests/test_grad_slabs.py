"""The voxel gradient in x slabs (option gather_slab of the library, xvr_amd.renderers.VOXEL_GRAD_SLABS): one launch per slab of whole
16^3-brick planes so that a caller can start exchanging slab i while slab i + 1 is computed (bench.py at N > 1).  Whatever kernel
serves the render, the slabbed backward must leave the SAME bits as the single launch, call the hook once per slab with that slab's
view of the gradient, in order, and leave the option cleared."""
import pytest
import torch

from conftest import make_case

pytestmark = pytest.mark.gpu

CASES = [
    ("trilinear", dict(n_points=60), "the brick splat on the poses' shared planes"),
    ("trilinear", dict(n_points=60, clip_to_volume=True), "the ray-major brick splat"),
    ("siddon", dict(norm_dims_offset=1), "the Siddon brick splat"),
    ("siddon", dict(), "the voxel gather: everything in the first call"),
]


def _backward(case, spec, slabs, w, grid_w):
    from xvr_amd import _lib, renderers
    from xvr_amd.renderers import render

    vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
    vol.requires_grad_(True)
    src.requires_grad_(True)
    seen = []
    renderers.VOXEL_GRAD_SLABS = (slabs, lambda i, g: seen.append((i, g.shape[0], g.data_ptr()))) if slabs else None
    try:
        out = render(vol, src, tgt, img, spec, ray_grid_w=grid_w)
        (out * w.cuda().reshape(out.shape)).sum().backward()
    finally:
        renderers.VOXEL_GRAD_SLABS = None
    torch.cuda.synchronize()
    assert _lib.get_option("gather_slab") == 0
    return vol.grad, src.grad, seen


@pytest.mark.parametrize("renderer,kw,why", CASES, ids=[c[2] for c in CASES])
@pytest.mark.parametrize("slabs", [2, 3, 7])
def test_slabbed_voxel_gradient_is_the_single_launch_bit_for_bit(renderer, kw, why, slabs):
    from xvr_amd.spec import RenderSpec

    shape, hw = (52, 37, 45), (40, 44)     # four brick planes along x: 7 slabs leaves some empty
    spec = RenderSpec(renderer=renderer, **kw)
    case = make_case(seed=23, shape=shape, height=hw[0], width=hw[1], delx=0.9 * max(shape) / max(hw))
    w = torch.randn(2, 1, hw[0] * hw[1], generator=torch.Generator().manual_seed(3))
    g1, s1, _ = _backward(case, spec, 0, w, hw[1])
    gk, sk, seen = _backward(case, spec, slabs, w, hw[1])
    assert g1.abs().max() > 0
    assert torch.equal(g1, gk), why
    # (the pose gradient is computed once, with the first slab; its float atomics make it equal to rounding, not to the bit)
    assert torch.allclose(s1, sk, rtol=1e-4, atol=1e-5 * s1.abs().max().item()), (s1, sk)
    nb0 = (shape[0] + 15) // 16
    want = [(i, min(16 * ((i + 1) * nb0 // slabs), shape[0]) - min(16 * (i * nb0 // slabs), shape[0])) for i in range(slabs)]
    assert [(i, n) for i, n, _ in seen] == [(i, n) for i, n in want if n > 0], seen
    assert sum(n for _, n, _ in seen) == shape[0]
    rows = [p for _, _, p in seen]
    assert rows == sorted(rows) and rows[0] == gk.data_ptr(), "views of the gradient autograd hands out, first slab first"
