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
        pb = int(np.argmax(per_pose))
        d = ((o[pb, 0] - ref[0][pb, 0].double()).abs() / ref[0].abs().max()).reshape(H, W)
        bad = (d > 1e-3).nonzero()
        print("    pose", pb, "bad pixels", bad.shape[0], "of", H * W, bad[:12].tolist())
        src, tgt = case["source"][pb], case["target"][pb].reshape(H, W, 3)
        for (i, j) in bad[:6].tolist():
            dd = (tgt[i, j] - src.reshape(-1, 3)[0]).tolist()
            print("      px", i, j, "hip", o[pb, 0].reshape(H, W)[i, j].item(), "ref", ref[0][pb, 0].reshape(H, W)[i, j].item(), "src", src.reshape(-1, 3)[0].tolist(), "dir", dd)

print("---- single rays of pose 6")
pb = 6
src = case["source"][pb:pb + 1].cuda()
tg = case["target"][pb:pb + 1].reshape(1, H, W, 3)
im = case["img"][pb:pb + 1].reshape(1, 1, H, W) if case["img"].dim() == 3 else None
sel = [(8, 0), (8, 1), (9, 0), (8, 5), (0, 0), (12, 12)]
t1 = torch.stack([tg[0, i, j] for i, j in sel]).reshape(1, len(sel), 3).cuda()
L = (t1.cpu() - case["source"][pb:pb + 1].reshape(1, 1, 3)).norm(dim=-1).reshape(1, 1, len(sel)).cuda()
print("img shape", case["img"].shape, "src shape", case["source"].shape)
for slab in (1, 2):
    with _lib.option("siddon_slab", slab):
        o = render(case["volume"].cuda(), src, t1, L, spec, None, ray_grid_w=0)
    print("siddon_slab", slab, o.reshape(-1).tolist())
# whole pose alone
for slab in (1, 2):
    with _lib.option("siddon_slab", slab):
        o = render(case["volume"].cuda(), case["source"][pb:pb + 1].cuda(), case["target"][pb:pb + 1].cuda(), case["img"][pb:pb + 1].cuda(), spec, None, ray_grid_w=W)
    print("pose alone, siddon_slab", slab, [o.reshape(H, W)[i, j].item() for i, j in sel])

print("---- uncovered rays?")
vol = case["volume"].cuda()
for slab in (1, 2):
    for trial in range(2):
        junk = torch.full((B, 1, H * W), float("nan"), device="cuda"); del junk
        with _lib.option("siddon_slab", slab):
            o = render(vol, case["source"].cuda(), case["target"].cuda(), case["img"].cuda(), spec, None, ray_grid_w=W)
        nanpix = torch.isnan(o).nonzero()
        print("siddon_slab", slab, "NaN outputs:", nanpix.shape[0], nanpix[:8].tolist())
