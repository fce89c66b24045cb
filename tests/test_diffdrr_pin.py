"""Consumes ``tests/golden/diffdrr_pin.npz`` -- outputs of the REAL diffdrr (the package xvr renders through,
/root/reference/uv.lock:955-977) on committed inputs, written by ``tools/pin_against_diffdrr.py`` on a machine where
the package is installed -- and holds the oracle (CPU) and the HIP kernels (GPU) to them with the knob set the tool
recorded.  The file cannot be produced in the build container (no diffdrr, no network): until somebody runs the tool
these tests skip and parity stays "unpinned" (DESIGN.md section 2); once it exists they are the pin."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

PIN = Path(__file__).resolve().parent / "golden" / "diffdrr_pin.npz"
needs_pin = pytest.mark.skipif(not PIN.exists(), reason="tests/golden/diffdrr_pin.npz absent: run tools/pin_against_diffdrr.py where diffdrr is installed")
PIN_FILE = {"path": PIN}      # (the rehearsal below points the consumers at the file the tool wrote against a planted renderer)

TAGS = [(r, s) for r in ("trilinear", "siddon") for s in (0.5, 0.0)]
CASES = ["case11", "case12", "c1"]


def _load():
    z = np.load(PIN_FILE["path"], allow_pickle=False)
    return {k: z[k] for k in z.files}


def _case(z, name):
    return {k: torch.from_numpy(z[f"in_{name}_{k}"]) for k in ("volume", "mask", "source", "target", "img")}


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()


def _check(a, b, tol, tie_prone, what):
    """max-norm relative error <= tol -- or, where the configuration is TIE-PRONE, for most entries, with a bound on the mean.
    A nearest-neighbour lookup whose argument sits within float32 noise of a rounding boundary is decided by the evaluation
    order, and two knob families put it there systematically: (i) Siddon under a non-exact index map -- with dims = shape + 1 the
    map sends the midpoint of the volume's CENTRE cell to a half-integer exactly ((S / 2 + shift) S / (S + 1) - 1/2 = S / 2 - 1/2
    for even S), so every ray that crosses that cell whole (a tenth of the rays of case11) credits one full-length segment to
    voxel S / 2 or S / 2 - 1 as rounding noise in x decides; the torch oracle in float32 and its float64 scalar twin already
    differ by 10 % on 2 of 240 such rays, the HIP walk on 21, all on that one segment (a volume of ones, of x- or z-indices
    renders identically; only y-indices differ, by exactly one cell length); (ii) a sample's label on the volume's face under
    per-ray clip_to_volume.  No tolerance on the maximum survives a tie; the fraction and the mean do.  (The fraction is bounded by
    the share of rays through the tie cell, not by what one machine happened to produce: the rehearsal's reference is made by the
    torch oracle where the test runs, and its float32 sums break the ties differently with the thread count -- 0.10 of case c1 in
    an 8-thread container, 0.146 on the 128-core GPU host.)"""
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    err = (a - b).abs() / b.abs().max().clamp_min(1e-12)
    if tie_prone:
        frac = (err > tol).double().mean().item()
        # (a pose gradient has three entries per pose, each the sum over all rays: there the ties show as a small error everywhere)
        assert (frac <= 0.25 or err.max().item() <= 1e-2) and err.mean().item() <= 5e-3, (what, frac, err.mean().item(), err.max().item())
    else:
        assert err.max().item() <= tol, (what, err.max().item())


def test_pin_tool_is_importable_and_refuses_to_run_without_diffdrr():
    """The recipe itself is kept healthy here: it parses, and without the package it stops with a clear message."""
    import importlib.util
    import sys

    spec = importlib.util.spec_from_file_location("pin_against_diffdrr", PIN.parents[2] / "tools" / "pin_against_diffdrr.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert callable(mod.main) and len(list(mod.knob_grid("trilinear"))) == 72 and len(list(mod.knob_grid("siddon"))) == 48
    c1 = mod.c1_case()
    assert c1["target"].shape == (1, 128 * 128, 3) and c1["img"].shape == (1, 1, 128 * 128)
    if importlib.util.find_spec("diffdrr") is None:
        with pytest.raises(SystemExit, match="diffdrr is not importable"):
            mod.main([])


@needs_pin
@pytest.mark.parametrize("renderer,shift", TAGS)
def test_oracle_reproduces_the_real_diffdrr(renderer, shift):
    from oracle.diffdrr_restated import RenderSpec, render

    z = _load()
    tag = f"{renderer}_shift{shift}"
    spec = RenderSpec(renderer=renderer, voxel_shift=shift, n_points=60, **json.loads(str(z[tag + "_knobs"])))
    for name in CASES:
        c = _case(z, name)
        v, s, t = (c[k].clone().requires_grad_(True) for k in ("volume", "source", "target"))
        out = render(v, s, t, c["img"], spec)
        (out * torch.from_numpy(z[f"{tag}_{name}_w"])).sum().backward()
        assert _rel(out, torch.from_numpy(z[f"{tag}_{name}_out"])) <= 1e-4
        for g, key in ((v.grad, "gvol"), (s.grad, "gsrc"), (t.grad, "gtgt")):
            assert _rel(g, torch.from_numpy(z[f"{tag}_{name}_{key}"])) <= 2e-3, key
        outm = render(c["volume"], c["source"], c["target"], c["img"], spec, c["mask"])
        assert _rel(outm, torch.from_numpy(z[f"{tag}_{name}_mask_out"])) <= 1e-4


@needs_pin
@pytest.mark.gpu
@pytest.mark.parametrize("renderer,shift", TAGS)
def test_hip_reproduces_the_real_diffdrr(renderer, shift):
    from xvr_amd.renderers import render
    from xvr_amd.spec import RenderSpec

    z = _load()
    tag = f"{renderer}_shift{shift}"
    knobs = json.loads(str(z[tag + "_knobs"]))
    knobs.pop("per_ray_clamp", None)   # the HIP traversal is per ray by construction
    spec = RenderSpec(renderer=renderer, voxel_shift=shift, n_points=60, **knobs)
    ties = renderer == "siddon" and bool(knobs.get("norm_dims_offset") or knobs.get("align_corners"))
    ties_mask = ties or (renderer == "trilinear" and knobs.get("clip_to_volume") is True)
    for name in CASES:
        c = {k: v.cuda() for k, v in _case(z, name).items()}
        v, s, t = (c[k].clone().requires_grad_(True) for k in ("volume", "source", "target"))
        gw = int(round(c["img"].shape[-1] ** 0.5)) if name == "c1" else 10
        out = render(v, s, t, c["img"], spec, ray_grid_w=gw)
        (out * torch.from_numpy(z[f"{tag}_{name}_w"]).cuda()).sum().backward()
        _check(out, torch.from_numpy(z[f"{tag}_{name}_out"]), 1e-4, ties, (name, "out"))
        for g, key in ((v.grad, "gvol"), (s.grad, "gsrc"), (t.grad, "gtgt")):
            _check(g, torch.from_numpy(z[f"{tag}_{name}_{key}"]), 2e-3, ties, (name, key))
        outm = render(c["volume"], c["source"], c["target"], c["img"], spec, c["mask"], ray_grid_w=gw)
        _check(outm, torch.from_numpy(z[f"{tag}_{name}_mask_out"]), 1e-4, ties_mask, (name, "masked out"))


# ---------------------------------------------------------------------------------------------------------------
# Rehearsal of the pin's first contact: a fake ``diffdrr.renderers`` whose Siddon / Trilinear are the oracle
# with PLANTED, non-default knobs (different per renderer and per voxel_shift).  The tool must run end to end against it, its grid
# search must recover exactly the planted knobs, and the consumers above, pointed at the file it wrote, must go green.  On the
# day someone has the real package the recipe then works first time.  (This proves the RECIPE; parity itself stays unpinned.)
# ---------------------------------------------------------------------------------------------------------------
PLANTED = {
    ("trilinear", 0.5): dict(norm_dims_offset=-1, align_corners=False, step_mode="n_minus_1", clip_to_volume=True),
    ("trilinear", 0.0): dict(norm_dims_offset=0, align_corners=True, step_mode="n_points", clip_to_volume="batch"),
    ("siddon", 0.5): dict(norm_dims_offset=+1, align_corners=False),
    ("siddon", 0.0): dict(norm_dims_offset=0, align_corners=True),
}


def _install_planted_diffdrr(monkeypatch):
    import sys
    import types

    from oracle.diffdrr_restated import RenderSpec, render as oracle_render

    def make(renderer):
        class _Renderer(torch.nn.Module):
            def __init__(self, voxel_shift=0.5):
                super().__init__()
                self.voxel_shift = voxel_shift

            def forward(self, volume, source, target, img, mask=None, n_points=500):
                spec = RenderSpec(renderer=renderer, voxel_shift=self.voxel_shift, n_points=n_points, **PLANTED[(renderer, self.voxel_shift)])
                return oracle_render(volume, source, target, img, spec, mask)
        _Renderer.__name__ = renderer.capitalize()
        return _Renderer

    pkg = types.ModuleType("diffdrr")
    pkg.__doc__, pkg.__version__, pkg.__file__, pkg.__path__ = "planted stand-in for the rehearsal", "0.0-planted", str(PIN), []
    ren = types.ModuleType("diffdrr.renderers")
    ren.Siddon, ren.Trilinear = make("siddon"), make("trilinear")
    pkg.renderers = ren
    monkeypatch.setitem(sys.modules, "diffdrr", pkg)
    monkeypatch.setitem(sys.modules, "diffdrr.renderers", ren)


@pytest.fixture(scope="module")
def planted_pin(tmp_path_factory):
    """tools/pin_against_diffdrr.py --quick run end to end against the planted renderers; -> path of the file it wrote."""
    import importlib.util

    mp = pytest.MonkeyPatch()
    try:
        _install_planted_diffdrr(mp)
        spec = importlib.util.spec_from_file_location("pin_against_diffdrr_rehearsal", PIN.parents[2] / "tools" / "pin_against_diffdrr.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out = tmp_path_factory.mktemp("pin") / "diffdrr_pin.npz"
        rc = mod.main(["--out", str(out), "--quick"])
        assert rc == 0 and out.exists()          # "PINNED": some knob set reproduces every vector to 1e-4
    finally:
        mp.undo()
    return out


def test_pin_recipe_recovers_planted_knobs(planted_pin):
    z = np.load(planted_pin, allow_pickle=False)
    for (renderer, shift), planted in PLANTED.items():
        tag = f"{renderer}_shift{shift}"
        found = json.loads(str(z[tag + "_knobs"]))
        assert float(z[tag + "_err"]) <= 1e-6, (tag, float(z[tag + "_err"]))
        for k, v in planted.items():      # (eps_in_xyz, per_ray_clamp and the Siddon filter are numerically neutral on these cases)
            assert found[k] == v, (tag, k, found[k], v)


@pytest.mark.parametrize("renderer,shift", TAGS)
def test_pin_consumers_go_green_on_the_rehearsal_file(planted_pin, renderer, shift, monkeypatch):
    monkeypatch.setitem(PIN_FILE, "path", planted_pin)
    test_oracle_reproduces_the_real_diffdrr(renderer, shift)


@pytest.mark.gpu
@pytest.mark.parametrize("renderer,shift", TAGS)
def test_hip_consumer_goes_green_on_the_rehearsal_file(planted_pin, renderer, shift, monkeypatch):
    monkeypatch.setitem(PIN_FILE, "path", planted_pin)
    test_hip_reproduces_the_real_diffdrr(renderer, shift)
