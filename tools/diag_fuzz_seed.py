"""One seed of tests/test_hip_parity.py::test_fuzz_random_configurations_against_the_oracle, with the per-quantity distances the
assertion hides:  python tools/diag_fuzz_seed.py 70004 70197   (on the GPU box)"""
import sys, traceback
from pathlib import Path
R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import torch
import test_hip_parity as T
orig = T._hip_render
keep = {}
def spy(*a, **k):
    r = orig(*a, **k); keep["hip"] = r; keep["case"] = a[0]; keep["spec"] = a[1]; keep["w"] = k.get("w"); keep["masked"] = k.get("mask") is not None; return r
T._hip_render = spy
orig_o = T._oracle_render
def spy_o(*a, **k):
    r = orig_o(*a, **k); keep["ref"] = r; return r
T._oracle_render = spy_o
for seed in map(int, sys.argv[1:]):
    try:
        T.test_fuzz_random_configurations_against_the_oracle(seed)
        print(seed, "passes")
    except BaseException as e:
        print(seed, "FAILED", str(e)[:400])
    names = ("out", "grad_volume", "grad_source", "grad_target", "grad_img")
    for n, h, r in zip(names, keep["hip"], keep["ref"]):
        if h is None or r is None: continue
        h, r = h.detach().double().cpu(), r.detach().double().cpu()
        e = (h - r).abs() / r.abs().max().clamp_min(1e-30)
        print(f"   {n}: max rel err {e.max().item():.2e}; entries beyond 2e-3: {int((e > 2e-3).sum())} of {e.numel()}; scale {r.abs().max().item():.3e}")
        if n == "grad_target":
            per = e.amax(dim=-1).reshape(-1)
            top = per.sort(descending=True)
            print("      worst rays:", [(int(i), f"{v:.1e}") for v, i in zip(top.values[:5].tolist(), top.indices[:5].tolist())])
    if "case" in keep:
        c = keep["case"]
        h_t, r_t = keep["hip"][3].detach().double().cpu(), keep["ref"][3].detach().double().cpu()
        per = ((h_t - r_t).abs().amax(dim=-1) / r_t.abs().max()).reshape(h_t.shape[0], -1)
        b, ray = divmod(int(per.argmax()), per.shape[1])
        s0 = c["source"][b].reshape(-1, 3)[0]; t0 = c["target"][b].reshape(-1, 3)[ray]
        print(f"      worst ray: pose {b} ray {ray}: source {s0.tolist()} direction {(t0 - s0).tolist()}; out hip {keep['hip'][0][b].reshape(-1)[ray].item():.6f} ref {keep['ref'][0][b].reshape(-1)[ray].item():.6f}"
              f"; d/d target hip {h_t[b].reshape(-1, 3)[ray].tolist()} ref {r_t[b].reshape(-1, 3)[ray].tolist()}; volume shape {tuple(c['volume'].shape)}")
        # d/d source against the FLOAT64 oracle, with the float32 oracle beside it, and the sources themselves (a source within
        # rounding of a voxel plane makes alpha = 0 a crossing: whether the first segment exists is then a tie)
        try:
            spec = keep.get("spec")
            if spec is not None:
                from conftest import to_oracle_spec
                from oracle.diffdrr_restated import render as orender
                d64 = {k: c[k].double().clone().requires_grad_(k != "mask") for k in ("volume", "source", "target", "img")}
                w = keep["w"].double()
                (orender(d64["volume"], d64["source"], d64["target"], d64["img"], to_oracle_spec(spec)) * w).sum().backward()
                g64 = d64["source"].grad
                gh, g32 = keep["hip"][2].detach().double().cpu(), keep["ref"][2].detach().double().cpu()
                sc = g64.abs().max()
                print(f"      d/d source vs FLOAT64 oracle: HIP {((gh - g64).abs().max() / sc).item():.2e}, float32 oracle {((g32 - g64).abs().max() / sc).item():.2e}")
                print(f"      sources (voxel units): {c['source'].reshape(-1, 3).tolist()}")
        except BaseException as e2:   # noqa: BLE001
            print("      (float64 diagnosis failed:", str(e2)[:200], ")")
