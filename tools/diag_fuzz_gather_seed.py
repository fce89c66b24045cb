"""One Siddon seed of tests/test_hip_parity.py::test_fuzz_voxel_gather_equals_atomic_scatter taken apart: march / merge walk forward x
splat / cells gather / scatter backward, each against the oracle and against the adjoint identity, per pose.
    python tools/diag_fuzz_gather_seed.py 120742 ['{"align_corners": false}']        (on the GPU box)"""
import sys
from pathlib import Path
import numpy as np, torch
R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
from conftest import make_case
from test_hip_parity import _oracle_render
from xvr_amd import renderers, _lib
from xvr_amd.renderers import render
from xvr_amd.spec import RenderSpec
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 70034
rng = np.random.default_rng(7000 + seed)
shape = tuple(int(x) for x in rng.integers(6, 30, size=3))
spacing = tuple(float(x) for x in rng.uniform(0.6, 2.5, size=3))
H, W = int(rng.integers(2, 40)), int(rng.integers(2, 40))
renderer = "trilinear" if rng.random() < 0.55 else "siddon"
kw = dict(renderer=renderer, voxel_shift=float(rng.choice([0.0, 0.5])), align_corners=bool(rng.random() < 0.25))
assert renderer == "siddon"
kw.update(norm_dims_offset=int(rng.choice([0, 1, 1, -1])))
extent = max(s * p for s, p in zip(shape, spacing))
inside = rng.random() < 0.15
depth = float(rng.uniform(0.1, 0.4) * extent) if inside else float(rng.uniform(1.2, 4.0) * extent)
B = int(rng.integers(1, 40))
rot = tuple(tuple(float(a) for a in rng.uniform(-180, 180, size=3) * np.array([1.0, 0.4, 0.3])) for _ in range(B))
xyz = tuple((float(rng.uniform(-0.3, 0.3) * extent), depth, float(rng.uniform(-0.3, 0.3) * extent)) for _ in range(B))
case = make_case(shape=shape, height=H, width=W, sdd=float(rng.uniform(1.5, 3.0) * depth), delx=float(rng.uniform(0.5, 3.0)),
                 n_labels=int(rng.integers(2, 6)), seed=seed, rot=rot, xyz=xyz, spacing=spacing)
import json
if len(sys.argv) > 2:
    kw.update(json.loads(sys.argv[2]))
spec = RenderSpec(**kw)
print(kw, shape, H, W, B, inside)
w = torch.rand(B, 1, H * W, generator=torch.Generator().manual_seed(seed))
ref = _oracle_render(case, spec, grads=True, w=w)
res = {}
for name, gather, slab, splat in (("march+splat", True, 1, 1), ("walk+cells", True, 2, 0), ("walk+scatter", False, 2, 0), ("march+scatter", False, 1, 1)):
    renderers.VOXEL_GATHER = gather
    with _lib.option("siddon_slab", slab), _lib.option("siddon_splat", splat):
        vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
        vol.requires_grad_(True)
        out = render(vol, src, tgt, img, spec, None, ray_grid_w=W)
        (out * w.cuda()).sum().backward()
    renderers.VOXEL_GATHER = True
    o, g = out.detach().double().cpu(), vol.grad.double().cpu()
    lhs, rhs = (o * w.double()).sum().item(), (g * case["volume"].double()).sum().item()
    eo = (o - ref[0].double()).abs().max().item() / ref[0].abs().max().item()
    eg = (g - ref[1].double()).abs().max().item() / ref[1].abs().max().item()
    per_pose = ((o - ref[0].double()).abs().amax(dim=(1, 2)) / ref[0].abs().max()).tolist()
    print(f"{name}: <Av,w> {lhs:.4f}  <v,ATw> {rhs:.4f}  out err {eo:.2e}  grad err {eg:.2e}  sum grad {g.sum().item():.4f} (oracle {ref[1].double().sum().item():.4f})")
    print("    worst poses (out):", [f"{i}:{e:.1e}" for i, e in sorted(enumerate(per_pose), key=lambda t: -t[1])[:4]])
    if name == "march+splat":
        # per pose: <A v, w>_b against <v, A_b^T w_b> (one backward per pose)
        for bb in range(B):
            vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
            vol.requires_grad_(True)
            ob = render(vol, src[bb:bb + 1], tgt[bb:bb + 1], img[bb:bb + 1], spec, None, ray_grid_w=W)
            (ob * w[bb:bb + 1].cuda()).sum().backward()
            l_, r_ = (ob.detach().double().cpu() * w[bb:bb + 1].double()).sum().item(), (vol.grad.double().cpu() * case["volume"].double()).sum().item()
            if abs(l_ - r_) > 3e-6 * abs(l_):
                print(f"      pose {bb}: <Av,w> {l_:.5f}  <v,ATw> {r_:.5f}  diff {l_ - r_:.2e}")
        pb = int(np.argmax(per_pose))
        d = ((o[pb, 0] - ref[0][pb, 0].double()).abs() / ref[0].abs().max()).reshape(H, W)
        bad = (d > 1e-3).nonzero()
        print("    pose", pb, "bad pixels", bad.shape[0], "of", H * W, bad[:12].tolist())
        src, tgt = case["source"][pb], case["target"][pb].reshape(H, W, 3)
        for (i, j) in bad[:6].tolist():
            dd = (tgt[i, j] - src.reshape(-1, 3)[0]).tolist()
            print("      px", i, j, "hip", o[pb, 0].reshape(H, W)[i, j].item(), "ref", ref[0][pb, 0].reshape(H, W)[i, j].item(), "src", src.reshape(-1, 3)[0].tolist(), "dir", dd)


if len(sys.argv) > 3:
    bb = int(sys.argv[3])
    print(f"---- pose {bb}: ray by ray, out_r against <v, A^T e_r>")
    volc = case["volume"].cuda()
    for r in range(H * W):
        vol = volc.clone().requires_grad_(True)
        ob = render(vol, case["source"][bb:bb + 1].cuda(), case["target"][bb:bb + 1].cuda(), case["img"][bb:bb + 1].cuda(), spec, None, ray_grid_w=W)
        ob.reshape(-1)[r].backward()
        l_, r_ = ob.reshape(-1)[r].item(), (vol.grad.double() * volc.double()).sum().item()
        if abs(l_ - r_) > 1e-5 * max(abs(l_), 1e-6):
            g = vol.grad
            s0 = case["source"][bb].reshape(-1, 3)[0]
            t0 = case["target"][bb].reshape(-1, 3)[r]
            print(f"  ray {r} (px {r // W},{r % W}): out {l_:.6f}  <v,ATe> {r_:.6f}  diff {l_ - r_:.3e}; sum of A^T e_r {g.sum().item():.6f}; src {s0.tolist()} tgt {t0.tolist()} raylen {case['img'][bb].reshape(-1)[r].item():.4f}")

if len(sys.argv) > 4:
    bb, r = int(sys.argv[3]), int(sys.argv[4])
    print(f"---- pose {bb} ray {r}: the row of A from the forward (one-hot volumes) against the row from the backward")
    shape3 = case["volume"].shape
    s_, t_, i_ = case["source"][bb:bb + 1].cuda(), case["target"][bb:bb + 1].cuda(), case["img"][bb:bb + 1].cuda()
    vol = case["volume"].cuda().clone().requires_grad_(True)
    ob = render(vol, s_, t_, i_, spec, None, ray_grid_w=W)
    ob.reshape(-1)[r].backward()
    Ab = vol.grad.detach().cpu().reshape(-1)
    Af = torch.zeros_like(Ab)
    n = Ab.numel()
    eye = torch.zeros(n, device="cuda")
    for v in range(n):
        eye.zero_(); eye[v] = 1.0
        with torch.no_grad():
            Af[v] = render(eye.reshape(shape3), s_, t_, i_, spec, None, ray_grid_w=W).reshape(-1)[r].item()
    d = (Af - Ab)
    idx = d.abs().nonzero().reshape(-1)
    print("   voxels where the rows differ:", [(tuple(int(x) for x in np.unravel_index(int(i), shape3)), f"fwd {Af[i].item():.5f} bwd {Ab[i].item():.5f}") for i in idx[:12]])
    print("   row sums: fwd", Af.sum().item(), "bwd", Ab.sum().item(), " nonzeros fwd", int((Af != 0).sum()), "bwd", int((Ab != 0).sum()))
    nzf = [(tuple(int(x) for x in np.unravel_index(int(i), shape3)), round(Af[i].item(), 4)) for i in (Af != 0).nonzero().reshape(-1)]
    nzb = [(tuple(int(x) for x in np.unravel_index(int(i), shape3)), round(Ab[i].item(), 4)) for i in (Ab != 0).nonzero().reshape(-1)]
    print("   fwd row:", nzf)
    print("   bwd row:", nzb)
