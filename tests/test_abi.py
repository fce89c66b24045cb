"""The C-ABI library builds for gfx950, loads without a GPU, and exports exactly what
include/xvr_drr.h declares (no compute calls here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    names = set()
    for header in sorted((ROOT / "include").glob("*.h")):      # xvr_drr.h, xvr_sim.h, xvr_pose.h
        text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
        names |= set(re.findall(r"\b(xvr_(?:drr|sim|pose)_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for want in ("xvr_drr_abi_version", "xvr_drr_last_error", "xvr_drr_trilinear_forward",
                 "xvr_drr_trilinear_backward", "xvr_drr_siddon_forward", "xvr_drr_siddon_backward",
                 "xvr_drr_backward_from_jac", "xvr_sim_ncc_forward_backward", "xvr_pose_camera_forward",
                 "xvr_pose_opt_step"):
        assert want in names


def test_library_builds_loads_and_exports_every_declared_symbol():
    from xvr_amd import _lib

    lib = _lib.load()
    assert lib.xvr_drr_abi_version() == _lib.ABI_VERSION
    raw = ctypes.CDLL(str(_lib.library_path()))
    for name in _declared():
        assert hasattr(raw, name), f"{name} declared in include/*.h but not exported"
        assert name in _lib.EXPORTS, f"{name} has no ctypes prototype"


def test_argument_errors_are_codes_not_aborts():
    from xvr_amd import _lib

    lib = _lib.load()
    rc = lib.xvr_drr_trilinear_forward(None, None, 4, 4, 4, 1, None, None, None, 1, 1, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.xvr_drr_last_error()
    with pytest.raises(RuntimeError, match="null pointer"):
        _lib.check(rc, "xvr_drr_trilinear_forward")


def test_cspec_layout_matches_header():
    from xvr_amd import _lib
    from xvr_amd.renderers import make_cspec
    from xvr_amd.spec import RenderSpec

    # 23 four-byte fields (... ray_grid_w, volume_layout), padding to 8, the alpha_window pointer
    assert ctypes.sizeof(_lib.CSpec) == (15 + 1 + 1 + 2 + 1 + 1 + 1 + 1) * 4 + 4 + 8 and _lib.CSpec.alpha_window.offset == 96
    c = make_cspec((10, 20, 30), RenderSpec(voxel_shift=0.5, n_points=200), ray_grid_w=16)
    assert list(c.a) == [1.0, 1.0, 1.0] and list(c.b) == [0.0, 0.0, 0.0]
    assert list(c.hi) == [9.5, 19.5, 29.5] and c.ray_grid_w == 16 and abs(c.inv_denom - 1 / 200) < 1e-9


def test_product_path_has_no_cpu_fallback():
    import torch

    from xvr_amd.renderers import render
    from xvr_amd.spec import RenderSpec

    v = torch.rand(4, 4, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        render(v, torch.zeros(1, 1, 3), torch.ones(1, 2, 3), torch.ones(1, 1, 2), RenderSpec())


def test_metrics_loss_and_training_glue_have_no_cpu_fallback_and_no_copied_reference_lines():
    """Round 4: the torch fall-backs that mirrored reference lines are gone from the product -- Equalize,
    XrayTransforms, DiceMetric, the loss's pose terms and render_samples' tail call their HIP entry point or raise -- and none of
    the reference's identifier runs is left in xvr_amd/ (their literal restatements live in oracle/, as the checker)."""
    import torch

    from xvr_amd.loss import DiceMetric
    from xvr_amd.metrics import Equalize, XrayTransforms

    x = torch.rand(2, 1, 16, 16)
    for call in (lambda: Equalize()(x), lambda: XrayTransforms(16)(x), lambda: XrayTransforms(16, equalize=True)(x),
                 lambda: DiceMetric()(x > 0.5, x > 0.4)):
        with pytest.raises(RuntimeError, match="no CPU path"):
            call()
    text = "\n".join(p.read_text() for p in (ROOT / "xvr_amd").rglob("*.py"))
    for run in ("cdf_normalized", "weights_norm", "Unrecognized reducefn", "(y_pred * y_true)", "pred_sum", "true_sum",
                "circle_shift", "preprocess_xray", "center_crop"):
        assert run not in text, run


def test_product_never_imports_the_oracle():
    for p in (ROOT / "xvr_amd").rglob("*.py"):
        assert "oracle" not in p.read_text().replace("oracle/", "").replace("oracle-only", ""), p


def test_options_table_is_set_through_the_abi_and_seeded_from_the_environment_once():
    """The A/B switches of the launch logic are a table behind xvr_drr_set_option / xvr_drr_get_option (include/xvr_drr.h);
    the environment seeds it when the library is loaded and is never consulted again (no getenv on the hot path)."""
    import subprocess
    import sys

    from xvr_amd import _lib

    assert _lib.get_option("gather_splat") in (0, 1)
    with _lib.option("fwd_split", 104):
        assert _lib.get_option("fwd_split") == 104
    assert _lib.get_option("fwd_split") == 0
    lib = _lib.load()
    assert lib.xvr_drr_set_option(b"no_such_option", 1) == -1
    assert lib.xvr_drr_set_option(b"tile_shape", 7) == -1 and _lib.get_option("tile_shape") == -1
    assert b"getenv" not in b"".join(p.read_bytes() for p in (ROOT / "xvr_amd" / "csrc").glob("drr_[!a]*.hip*")), \
        "a render translation unit calls getenv (only drr_api.hip's load-time table may)"
    code = ("import os; os.environ['XVR_DRR_GATHER_SPLAT'] = '0'; os.environ['XVR_DRR_ORDER_GROUP'] = '8x2';"
            "from xvr_amd import _lib; print(_lib.get_option('gather_splat'), _lib.get_option('order_group'));"
            "os.environ['XVR_DRR_GATHER_SPLAT'] = '1'; print(_lib.get_option('gather_splat'))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, check=True).stdout.split()
    assert out == ["0", str(8 | 2 << 8), "0"]


def _rank_builds(args):
    """One 'rank' of an 8-GPU launch importing the package over a stale library (worker of the test below)."""
    marker, log = args
    import time
    from pathlib import Path

    from xvr_amd import build

    build.is_stale = lambda: not Path(marker).exists()          # stale until somebody has built

    def fake_build(hipcc, force, verbose):                     # stands in for the minute of hipcc
        with open(log, "a") as f:
            f.write("build\n")
        time.sleep(0.5)
        Path(marker).write_text("built")
        return build.LIB

    build._build_locked = fake_build
    build.find_hipcc = lambda: "/bin/true"
    return str(build.build_library())


def test_eight_ranks_importing_over_a_stale_library_build_it_once(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 8 bench.py` imports the package in eight processes at once; if the
    library is stale exactly ONE of them may rebuild it (xvr_amd/build.py's file lock), the others wait and load the result
    ('the only guard against 8 ranks rebuilding at import -- untested at 8')."""
    import multiprocessing as mp

    marker, log = tmp_path / "built.marker", tmp_path / "builds.log"
    with mp.get_context("spawn").Pool(8) as pool:
        libs = pool.map(_rank_builds, [(str(marker), str(log))] * 8)
    assert len(set(libs)) == 1 and log.read_text().count("build") == 1


def _undefined_globals(path):
    """Names a module's functions read as globals that the module never defines (a NameError waiting for the first call that
    reaches the line -- bench.py's result assembly only runs on a GPU box)."""
    import builtins
    import symtable

    src = Path(path).read_text()
    top = symtable.symtable(src, str(path), "exec")
    defined = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    defined |= set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    missing = set()

    def walk(table):
        for s in table.get_symbols():
            if table.get_type() != "module" and s.is_global() and s.is_referenced() and s.get_name() not in defined:
                missing.add((table.get_name(), s.get_name()))
        for child in table.get_children():
            walk(child)

    walk(top)
    return missing


def test_driver_facing_scripts_reference_no_undefined_globals():
    for path in [ROOT / "bench.py", ROOT / "__graft_entry__.py", *sorted((ROOT / "xvr_amd").glob("*.py")), *sorted((ROOT / "tools").glob("*.py"))]:
        assert not _undefined_globals(path), (path.name, _undefined_globals(path))
