"""For a case of tests/test_fuzz_large.py: the rays whose pose gradient differs from the float32 oracle by more than the tolerance, and for
each of them how far the HIP kernels and the float32 oracle are from the FLOAT64 oracle (near-tie rays: a piecewise function has two
one-sided derivatives there).  python tools/diag_fuzz_large.py   (on the GPU box; edit the (renderer, seed) list below)"""
import sys, torch, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from test_fuzz_large import _case
from test_hip_parity import _hip_render, _oracle_render
from conftest import to_oracle_spec
from xvr_amd.spec import RenderSpec
from xvr_amd import _lib
from oracle.diffdrr_restated import render as orender

def o64(case, spec, w, rays):
    b, r = rays
    vol = case["volume"].double().requires_grad_(True)
    src = case["source"][b:b+1].double().requires_grad_(True)
    tgt = case["target"][b:b+1, r:r+1].double().requires_grad_(True)
    img = case["img"][b:b+1, :, r:r+1].double().requires_grad_(True)
    out = orender(vol, src, tgt, img, to_oracle_spec(spec))
    (out * w[b:b+1, :, r:r+1].double()).sum().backward()
    return out.item(), tgt.grad[0, 0].tolist()

for renderer, seed in (("siddon", 61),):
    if renderer == "siddon":
        case, h, w, n = _case(seed, 8 if seed % 2 else 11)
        spec = RenderSpec(renderer="siddon", voxel_shift=0.5 if seed % 2 else 0.0)
    else:
        case, h, w, n = _case(100 + seed, 8 if seed % 2 else 9)
        kw = [dict(), dict(voxel_shift=0.0, step_mode="n_minus_1"), dict(norm_dims_offset=-1), dict(near=0.15, far=0.95)][seed % 4]
        spec = RenderSpec(renderer="trilinear", n_points=int(np.random.default_rng(seed).integers(40, 110)), **kw)
    wgt = torch.rand(n, 1, h * w, generator=torch.Generator().manual_seed(seed))
    hip = _hip_render(case, spec, grid_w=w, grads=True, w=wgt)
    with _lib.option("siddon_slab", 0), _lib.option("fwd_split", 1):
        hip2 = _hip_render(case, spec, grid_w=w, grads=True, w=wgt)
    ref = _oracle_render(case, spec, grads=True, w=wgt)
    from test_fuzz_large import _oracle64
    ref64 = _oracle64(case, spec, wgt)
    e64 = (hip[3].cpu().double() - ref64[3]).abs().amax(dim=-1) / ref64[3].abs().max().item()
    print('hip vs float64: rays beyond 2e-3:', int((e64 > 2e-3).sum()), ' beyond 1e-2:', int((e64 > 1e-2).sum()))
    e32 = (ref[3].double() - ref64[3]).abs().amax(dim=-1) / ref64[3].abs().max().item()
    print('float32 oracle vs float64: rays beyond 2e-3:', int((e32 > 2e-3).sum()), ' beyond 1e-2:', int((e32 > 1e-2).sum()))
    gt_h, gt_h2, gt_r = hip[3].cpu(), hip2[3].cpu(), ref[3]
    scale = gt_r.abs().max().item()
    err = (gt_h - gt_r).abs().amax(dim=-1) / scale
    print(f"== {renderer} seed {seed} shape {tuple(case['volume'].shape)} det {h}x{w} poses {n}; scale {scale:.3f}; rays with err > 2e-3: {(err > 2e-3).sum().item()} of {err.numel()}; hip default vs hip alt max {(gt_h - gt_h2).abs().max().item() / scale:.2e}")
    out_idx = (err.flatten() > 2e-3).nonzero().flatten().tolist()
    nh = no = 0
    for idx in out_idx:
        b, r = idx // (h * w), idx % (h * w)
        o, g64 = o64(case, spec, wgt, (b, r))
        g64 = torch.tensor(g64)
        eh = (gt_h[b, r].double() - g64).abs().max().item() / scale
        eo = (gt_r[b, r].double() - g64).abs().max().item() / scale
        nh += eh > 2e-3; no += eo > 2e-3
        s_, t_ = case["source"][b, 0], case["target"][b, r]
        d = (t_ - s_); d = d / d.norm()
        print(f"   pose {b} ray {r}: hip-o64 {eh:.2e} o32-o64 {eo:.2e} dirn {[round(x, 4) for x in d.tolist()]}")
    print(f" outliers {len(out_idx)}: hip beyond tol of float64 on {nh}, float32 oracle beyond tol of float64 on {no}")
    flat = err.flatten().topk(2).indices
    for idx in flat.tolist():
        b, r = idx // (h * w), idx % (h * w)
        o, g64 = o64(case, spec, wgt, (b, r))
        print(f" pose {b} ray {r} ({r // w},{r % w}) out hip {hip[0][b,0,r].item():.6f} o32 {ref[0][b,0,r].item():.6f} o64 {o:.6f}")
        print("   gtgt hip ", [f"{x:.5f}" for x in gt_h[b, r].tolist()], " alt ", [f"{x:.5f}" for x in gt_h2[b, r].tolist()])
        print("   gtgt o32 ", [f"{x:.5f}" for x in gt_r[b, r].tolist()], " o64 ", [f"{x:.5f}" for x in g64])
        s, t = case["source"][b, 0], case["target"][b, r]
        print("   src", [f"{x:.3f}" for x in s.tolist()], "dir", [f"{x:.4f}" for x in (t - s).tolist()])
