import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from conftest import make_case, to_oracle_spec
from test_hip_parity import _hip_render, _oracle_render
from xvr_amd.spec import RenderSpec
from xvr_amd import renderers
for kw in (dict(n_points=120), dict(n_points=100, voxel_shift=0.0, norm_dims_offset=-1)):
    spec = RenderSpec(renderer="trilinear", clip_to_volume="batch", **kw)
    case = make_case(seed=19, shape=(36, 40, 44), height=24, width=32, delx=1.6,
                     rot=((170.0, 25.0, 5.0), (200.0, -30.0, -8.0), (150.0, 5.0, 12.0)), xyz=((5.0, 300.0, -4.0), (-3.0, 250.0, 6.0), (0.0, 280.0, 0.0)))
    w = torch.rand(3, 1, 24 * 32, generator=torch.Generator().manual_seed(6))
    ref = _oracle_render(case, spec, grads=True, w=w)
    hip = _hip_render(case, spec, grid_w=32, grads=True, w=w)
    for i, name in ((2, "grad_source"), (3, "grad_target")):
        d = (hip[i].cpu() - ref[i]).abs()
        idx = (d == d.max()).nonzero()[0].tolist()
        print(kw, name, "max abs err", d.max().item(), "at", idx, "hip", hip[i].cpu()[tuple(idx)].item(), "ref", ref[i][tuple(idx)].item())
    t = d.reshape(3, -1, 3).sum(-1)
    top = torch.topk(t.flatten(), 5)
    print("  top-5 target rays by error:", [(int(i) // t.shape[1], int(i) % t.shape[1], float(v)) for v, i in zip(top.values, top.indices)])
    torch.cuda.synchronize()
    print("  window", renderers._LAST_WINDOW[:16].cpu().tolist())
    print("  hip grad_source", hip[2].cpu().flatten().tolist()); print("  ref grad_source", ref[2].flatten().tolist())
