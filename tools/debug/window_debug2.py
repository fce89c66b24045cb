import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from conftest import make_case, to_oracle_spec
from oracle.diffdrr_restated import trilinear, batch_window
from xvr_amd import _lib
from xvr_amd.renderers import make_cspec, _ptr, _stream
from xvr_amd.spec import RenderSpec
spec = RenderSpec(renderer="trilinear", clip_to_volume="batch", n_points=120)
case = make_case(seed=19, shape=(36, 40, 44), height=24, width=32, delx=1.6,
                 rot=((170.0, 25.0, 5.0), (200.0, -30.0, -8.0), (150.0, 5.0, 12.0)), xyz=((5.0, 300.0, -4.0), (-3.0, 250.0, 6.0), (0.0, 280.0, 0.0)))
w = torch.rand(3, 1, 24 * 32, generator=torch.Generator().manual_seed(6))
lib = _lib.load()
vol, src, tgt, img = (case[k].cuda().contiguous() for k in ("volume", "source", "target", "img"))
B, n = 3, 24 * 32
cs = make_cspec(vol.shape, spec, 32)
window = torch.empty(lib.xvr_drr_alpha_window_bytes(B) // 4, device="cuda")
_lib.check(lib.xvr_drr_alpha_window(_ptr(src.reshape(B, 3)), _ptr(tgt), B, n, *vol.shape, ctypes.byref(cs), _ptr(window), _stream()), "w")
cs.alpha_window = window.data_ptr()
out = torch.empty(B, 1, n, device="cuda"); jac = torch.empty(B, n, 8, device="cuda")
_lib.check(lib.xvr_drr_trilinear_forward(_ptr(vol), None, *vol.shape, 1, _ptr(src.reshape(B, 3).contiguous()), _ptr(tgt), _ptr(img.reshape(B, n).contiguous()), B, n,
                                         ctypes.byref(cs), _ptr(out), _ptr(jac), None, _stream()), "f")
torch.cuda.synchronize()
j = jac.cpu().double(); A, Z = window[0].item(), window[1].item(); W = Z - A
# oracle per-ray quantities in float64
dt = torch.float64
v, im = case["volume"].to(dt), case["img"].to(dt); s, t = case["source"].to(dt), case["target"].to(dt)
Ao, Zo = batch_window(s, t, v.shape, to_oracle_spec(spec)); Ao, Zo = Ao.detach(), Zo.detach()
sr = s.expand(B, n, 3).clone().requires_grad_(True); tr = t.clone().requires_grad_(True)
Al = Ao.clone().requires_grad_(True); Zl = Zo.clone().requires_grad_(True)
o = trilinear(v, sr.reshape(B * n, 1, 3), tr.reshape(B * n, 1, 3), im.reshape(B * n, 1, 1), to_oracle_spec(spec), window=(Al, Zl))
o.sum().backward()
js_o, jt_o = sr.grad, tr.grad
print("A,Z hip", A, Z, "oracle", Ao.item(), Zo.item())
print("out   max abs diff", (out.cpu().double().reshape(B, n) - o.detach().reshape(B, n)).abs().max().item(), "scale", o.abs().max().item())
print("js    max abs diff", (j[..., 1:4] - js_o).abs().max().item(), "scale", js_o.abs().max().item())
print("jt    max abs diff", (j[..., 4:7] - jt_o).abs().max().item(), "scale", jt_o.abs().max().item())
d = (t - s) + spec.eps
s1_h = (d * (j[..., 1:4] + j[..., 4:7])).sum(-1); s1_o = (d * (js_o + jt_o)).sum(-1)
print("s1 sums (unweighted)", s1_h.sum().item(), s1_o.sum().item(), " weighted", (w.reshape(B, n).double() * s1_h).sum().item(), (w.reshape(B, n).double() * s1_o).sum().item())
e1_o = (d * jt_o).sum(-1) - Ao * s1_o
print("E1s: hip jac[7] vs oracle  max abs diff", (j[..., 7] - e1_o).abs().max().item(), "scale", e1_o.abs().max().item(), " sums", j[..., 7].sum().item(), e1_o.sum().item())
bad = (s1_h - s1_o).abs()
top = torch.topk(bad.flatten(), 5)
print("rays with the largest s1 error", [(int(i) // n, int(i) % n, float(x), float(s1_o.flatten()[i])) for x, i in zip(top.values, top.indices)])
