"""bench.py prints ONE JSON line with the driver's contract (run at a tiny size so it takes seconds)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _no_constants(name):
    raise AssertionError(f"{name} in bench.py's JSON line")


def _the_line(stdout):
    """The driver's view: the LAST stdout line, strict JSON (no NaN / Infinity), below 8 KB (round 5's was 23 KB and
    `BENCH_r05.parsed` came back null), and the only line that starts a JSON object."""
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert [l for l in lines if l.startswith("{")] == [lines[-1]], stdout[-3000:]
    assert len(lines[-1]) < 8192, len(lines[-1])
    return json.loads(lines[-1], parse_constant=_no_constants)


def _scalars_only(d, depth=0):
    for k, v in d.items():
        if isinstance(v, dict):
            assert depth < 1, k          # `variants` = scalars, or one level of {name: scalar}
            _scalars_only(v, depth + 1)
        else:
            assert isinstance(v, (int, float)) and not isinstance(v, bool), (k, v)


@pytest.mark.gpu
@pytest.mark.parametrize("renderer", ["trilinear", "siddon"])
def test_bench_json_contract(renderer, tmp_path):
    full_path = tmp_path / "full.json"
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--size", "64", "--det", "32",
           "--batch", "4", "--n-points", "80", "--renderer", renderer, "--cpu-rays", "256", "--c5", "--full-json", str(full_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _the_line(out.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "voxel_gradient_sums", "kernels_ms"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "render-ready copy" in d["config"]["workload"] and "before the timed region" in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"] + 1e-9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["units_per_launch"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and r["avg_launch_ms"] > 0
    assert all(not isinstance(v, (dict, list)) or k in ("binding", "hbm_physical", "step_pair") for k, v in r.items())
    assert r["step_pair"]["bytes_per_unit"] == 2 * r["bytes_per_unit"] and r["step_pair"]["kernel_ms_per_step"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "DRRs/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert c["threads"] >= c["cores"] and "median of" in c["sample"]
    assert all(v > 0 for v in d["kernels_ms"].values()) and f"{renderer}_forward+jac" in d["kernels_ms"]
    # the full result (kernel tables, every candidate floor, sources, spreads) went to the side file and to stderr
    full = json.loads(full_path.read_text())
    assert "bench_full " + json.dumps(full) in out.stderr
    assert full["value"] == d["value"] and full["ms_per_step"] == d["ms_per_step"]
    # what really binds the dominant kernel (its taps are cache-served): a floor from the committed microbenchmarks
    # (every kernel that has a committed microbenchmark behind it carries one; the dominant trilinear kernel always does)
    fwd = full["kernels"][f"{renderer}_forward+jac"]["binding"]
    assert fwd["unit"] in ("texture_address", "valu_issue", "fabric_bandwidth") and 0 < fwd["floor_ms"] and abs(fwd["frac"] - fwd["floor_ms"] / full["kernels"][f"{renderer}_forward+jac"]["avg_ms"]) < 1e-9
    assert fwd["floors_ms"][fwd["unit"]] == max(fwd["floors_ms"].values())
    if renderer == "trilinear":
        assert set(r["binding"]) == {"unit", "floor_ms", "frac"}
        assert r["binding"]["unit"] in ("lds_atomic_issue", "texture_address", "valu_issue") and 0 < r["binding"]["floor_ms"]
        assert abs(r["binding"]["frac"] - r["binding"]["floor_ms"] / r["avg_launch_ms"]) < 1e-3
        # the headline under every reading of the unpinned knobs, as top-level scalars in the headline's unit
        v = d["variants"]
        _scalars_only(v)
        assert abs(d["value_clip_per_ray"] - 4 / (v["clip_per_ray_ms"] * 1e-3)) < 2e-3 * d["value_clip_per_ray"]
        assert abs(d["value_clip_batch"] - 4 / (v["clip_batch_ms"] * 1e-3)) < 2e-3 * d["value_clip_batch"]
        assert abs(d["value_volume_changing"] - 4 / (v["volume_changing_ms"] * 1e-3)) < 2e-3 * d["value_volume_changing"]
        # every single-GPU configuration of BASELINE.json in the driver's line: C3, the pose-only steps, the recalled knob sets,
        # the C4 iteration at both pyramid levels (single and batched), the C5 step (also under the per-ray clip) -- scalars
        assert v["siddon_ms"] > 0 and v["siddon_nx_ms"] > 0
        assert set(v["pose_only_ms"]) == {"trilinear", "siddon", "trilinear_clip_per_ray", "siddon_nx"} and all(x > 0 for x in v["pose_only_ms"].values())
        assert set(v["c4_ms_per_iter"]) == {"32", "64"} == set(v["c4_ms_per_pose_iter_batched8"])
        assert v["c5_step_ms"] > 0 and v["c5_step_ms_clip"] > 0
        assert "siddon_backward[vol]" in v["siddon_nx_kernels_ms"] and "siddon_backward[vol]" in v["siddon_kernels_ms"]
        # ... and their tables in the full result
        fv = full["variants"]
        assert fv["siddon"]["roofline"]["unit_name"] == "voxel segments" and fv["siddon"]["kernels"]
        assert "trilinear_forward+jac" in fv["trilinear_pose_only"]["kernels"] and fv["trilinear_pose_only"]["roofline"]["forward_backward_pair"]["bytes_per_unit"] == 32
        rk = fv["recalled_knobs"]     # SURVEY Appendix A's recalled knob sets, step and pose-only
        assert set(rk) == {"trilinear_clip_per_ray", "siddon_dims_plus_1"}
        assert all(e["ms_per_step"] > 0 and e["pose_only_ms_per_step"] > 0 and e["kernels"] for e in rk.values())
        assert rk["siddon_dims_plus_1"]["spec"] == {"norm_dims_offset": 1}
        c4 = fv["c4_register_ms_per_pose_iteration"]
        assert set(c4) == {"32", "64"} and all(c4[k]["single"] > 0 and c4[k]["batched8"] > 0 and c4[k]["kernels"] for k in c4)
        assert all(c4[k]["ncc"][1] > c4[k]["ncc"][0] for k in c4)
        assert "clip_to_volume" in fv["c5_train_step_clip"]["config"] and fv["c5_train_step_clip"]["kernels"]
        assert r["nominal_frac"] >= r["frac"]            # nominal samples >= volume-touching samples
        assert c["c1_plumbing_128x128_DRRs_per_s"] > 0   # BASELINE.json configs[0], whole DRRs, on the CPU
        p1 = full["cpu_baseline"]["c1_plumbing"]
        assert p1["value"] > 0 and p1["reps"] >= 3 and "128x128" in p1["config"] and "batch_size 4" in p1["config"]


@pytest.mark.gpu
def test_bench_one_rank_over_rccl():
    """--force-dist --backend nccl: the process group is RCCL and the step's all_gather_into_tensor runs on it -- on ONE rank, so
    that the first RCCL call of this project does not happen on the driver's 8-GPU node (missing 4)."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--size", "64", "--det", "32", "--batch", "4",
           "--n-points", "80", "--no-cpu-baseline", "--no-variants", "--force-dist", "--backend", "nccl", "--update-volume"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _the_line(out.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["global_batch"] == 4


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--scaling", "strong"], ["--renderer", "siddon"]], ids=["weak", "strong-ragged", "siddon"])
def test_bench_dry_run_collectives_as_rank_of_eight(extra):
    """--gpus 8 --dry-run-collectives on ONE GPU: the tensors of the 8-rank step (gather buffer, padded ragged share) allocated as a
    rank would, the step run with its async all_gather_into_tensor on a one-rank RCCL group, the gathered block compared with the
    render.  The first 8-GPU run is then the measurement, not the debugging."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--dry-run-collectives", "--size", "64", "--det", "32", "--batch", "13",
           "--n-points", "80", "--backend", "nccl", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["ok"] and d["as_world_size"] == 8 and [l["as_rank"] for l in d["legs"]] == [0, 7]
    assert all(l["volume_grad_allreduce_identity"] and l["volume_grad_slabs"] == 4 for l in d["legs"])
    if "strong" in extra:   # 13 poses over 8 ranks: shares of 2 and 1, padded to 2
        assert [l["poses"] for l in d["legs"]] == [2, 1] and all(l["share_padded_to"] == 2 for l in d["legs"])
    else:
        assert all(l["poses"] == 13 and abs(l["gather_buffer_MB"] - 8 * 13 * 32 * 32 * 4 / 1e6) < 1e-9 for l in d["legs"])


def _torchrun_bench(extra, port):
    """bench.py's N > 1 branch on ONE GPU: two ranks, gloo rendezvous, both on cuda:0 (--single-device)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64",
           "--det", "32", "--n-points", "80", "--backend", "gloo", "--single-device", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return _the_line(out.stdout)          # rank 0 alone prints


@pytest.mark.gpu
def test_bench_bare_command_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with no launcher around it (WORLD_SIZE unset) re-executes itself under
    torch.distributed.run and rank 0 prints the one line (it used to exit with a usage error, which is what an
    8-GPU lease shaped like the 1-GPU step would have recorded).  And N = 1 through a launcher prints what the plain run prints."""
    import os

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--steps", "2", "--warmup", "1", "--size", "64", "--det", "32", "--n-points", "80", "--batch", "4"]
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo", "--check-gather",
                          "--full-json", str(tmp_path / "n2.json"), *common], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _the_line(out.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["gather_check"]["equal"] is True
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["volume_grad_exchange"]["in_the_timed_step"] is True and d["volume_grad_exchange"]["ms_per_step_without_it"] > 0
    one = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29641", str(ROOT / "bench.py"), "--gpus", "1", "--no-variants", "--no-cpu-baseline",
           "--full-json", str(tmp_path / "n1.json"), *common]
    out1 = subprocess.run(one, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out1.returncode == 0, out1.stderr[-3000:]
    d1 = _the_line(out1.stdout)
    plain = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-variants", "--no-cpu-baseline", "--full-json", str(tmp_path / "p.json"), *common],
                           capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    d0 = _the_line(plain.stdout)
    assert set(d1) == set(d0) and d1["n_gpus"] == d0["n_gpus"] == 1 and d1["config"] == d0["config"]
    assert d1["roofline"]["units_per_launch"] == d0["roofline"]["units_per_launch"]     # the same work; the time is a measurement


@pytest.mark.gpu
@pytest.mark.parametrize("slabs", [4, 1], ids=["overlapped slabs", "buckets after the backward"])
def test_bench_two_ranks_weak_scaling_contract(slabs):
    d = _torchrun_bench(["--batch", "4", "--check-gather", "--check-volume-grad", "--volume-grad-slabs", str(slabs)], 29611 + slabs)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8
    # the step's second exchange: the voxel gradients of the two ranks, summed inside the timed step (slab by slab
    # while the backward runs, or by bucketed async all-reduces behind it) -- equal to the gradient of the union batch rendered by one process through the HIP path --
    # and the same step timed without it next to the headline
    x = d["volume_grad_exchange"]
    assert x["in_the_timed_step"] and x["slabs"] == slabs and x["buckets"] == (8 if slabs == 1 else 0)
    assert x["ms_per_step_without_it"] > 0 and x["value_without_it"] > 0
    assert d["volume_grad_check"]["ok"] and d["volume_grad_check"]["nonzero_voxels"] > 0, d["volume_grad_check"]
    assert "all-reduce of the voxel gradient" in d["config"]["parallelism"]
    # the gathered tensor, value by value, against a single-process render of both ranks' poses through the HIP path
    assert d["gather_check"] == {"equal": True, "max_abs_diff": 0.0, "padding_zero": True, "ranks": 2}
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d and "pose-sharded x2" in d["config"]["parallelism"]
    assert d["roofline"]["units_per_launch"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [6, 5], ids=["even-split", "ragged-split"])
def test_bench_two_ranks_strong_scaling_contract(batch):
    d = _torchrun_bench(["--batch", str(batch), "--scaling", "strong", "--check-gather"], 29612 + batch)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == batch
    assert d["gather_check"] == {"equal": True, "max_abs_diff": 0.0, "padding_zero": True, "ranks": 2}   # (ragged: 3 + 2 poses, padded to 3)
    assert abs(d["value"] - batch * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


def test_the_line_is_rebuilt_from_the_committed_full_result_and_stays_small():
    """No GPU needed: bench.compact_line applied to the newest committed full result (profiles/rNN_bench_full_default.json, written
    by the driver's own command on the GPU box) reproduces the committed line's keys, is strict JSON below 8 KB -- round 5's 23 KB
    line is why BENCH_r05.parsed is null -- and carries only scalars in `variants`."""
    sys.path.insert(0, str(ROOT))
    import bench

    fulls = sorted((ROOT / "profiles").glob("r[0-9][0-9]_bench_full_default.json"))
    assert fulls, "no committed full bench result"
    full = json.loads(fulls[-1].read_text())
    line = json.dumps(bench.compact_line(full), allow_nan=False, separators=(",", ":"))
    assert len(line) < 4096, len(line)
    d = json.loads(line, parse_constant=_no_constants)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "variants", "value_clip_per_ray", "value_clip_batch", "value_volume_changing"):
        assert key in d, key
    _scalars_only(d["variants"])
    assert set(d["roofline"]["binding"]) == {"unit", "floor_ms", "frac"} and d["roofline"]["hbm_physical"]["GBps"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    committed = json.loads((fulls[-1].with_name(fulls[-1].name.replace("_full_", "_final_"))).read_text().strip(), parse_constant=_no_constants)
    assert set(committed) == set(d) and committed["value"] == d["value"] and committed["variants"].keys() == d["variants"].keys()
    assert "before the timed region" in d["config"]["workload"]


def test_pmc_traffic_reads_the_committed_counter_table():
    """`roofline.traffic` comes from profiles/traffic.json (FETCH_SIZE + WRITE_SIZE per launch of the committed PMC passes):
    the kernels of one timed call add up, the instantiations of one kernel (volume layouts) count once, and Siddon's
    forward, forward + jacobian and backward instantiations are told apart."""
    sys.path.insert(0, str(ROOT / "tools"))
    import benchlib as bench

    table = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    splat = sum(v["fetch_bytes"] + v["write_bytes"] for k, v in table.items()
                if any(t in k for t in ("k_trilinear_splat_b16", "k_trilinear_gather_tab", "k_gather_prep", "k_gather_cull", "k_trilinear_bwd")))
    assert bench.pmc_traffic("trilinear_backward[vol]") == pytest.approx(splat)
    fwd = [v["fetch_bytes"] + v["write_bytes"] for k, v in table.items() if "k_trilinear_fwd<true" in k]
    assert len(fwd) >= 1 and bench.pmc_traffic("trilinear_forward+jac") == pytest.approx(sum(fwd) / len(fwd))
    # the Siddon forward of the default path is the slab march: <true, ...> carries the jacobian, <false, ...> does not
    # ... and <.., .., true> is the instantiation for non-exact index maps (the recalled dims = shape + 1): its own leg
    jac = [v["fetch_bytes"] + v["write_bytes"] for k, v in table.items() if "k_siddon_slab<true, true, false>" in k or "k_siddon_slab<true, true>" in k]
    plain = [v["fetch_bytes"] + v["write_bytes"] for k, v in table.items() if "k_siddon_slab<false, true, false>" in k or "k_siddon_slab<false, true>" in k]
    assert jac and plain
    nx = [v["fetch_bytes"] + v["write_bytes"] for k, v in table.items() if "k_siddon_slab<true, true, true>" in k]
    assert nx and bench.pmc_traffic("siddon_forward+jac", "nx") == pytest.approx(sum(nx) / len(nx))
    sp = sum(v["fetch_bytes"] + v["write_bytes"] for k, v in table.items() if any(t in k for t in ("k_siddon_splat", "k_gather_prep", "k_gather_cull", "k_siddon<2")))
    assert bench.pmc_traffic("siddon_backward[vol]", "nx") == pytest.approx(sp, rel=0.05)
    assert bench.pmc_traffic("siddon_forward+jac") == pytest.approx(sum(jac) / len(jac))
    assert bench.pmc_traffic("siddon_forward") == pytest.approx(sum(plain) / len(plain))
    assert bench.pmc_traffic("no_such_call") is None


def test_binding_counts_are_parsed_from_the_committed_summaries_not_pasted(tmp_path):
    """`roofline.binding` prices its floors with instruction and line counts read at run time from the newest committed
    profiles/rNN_*_rocprof_summary.md, divided by the unit count of the bench run they were collected under (VERDICT r4 item
    10: they were literals in bench.py and went stale with the next kernel edit)."""
    import re

    sys.path.insert(0, str(ROOT / "tools"))
    import benchlib
    import benchlib as bench

    text = (ROOT / "bench.py").read_text() + (ROOT / "tools" / "benchlib.py").read_text()
    assert not re.search(r"\d\.\d+e[89] / \(", text), "a pasted counter literal is back in bench.py"
    for base, (kind, pattern) in bench.BINDING_KERNELS.items():
        c = benchlib.committed_counters(kind, pattern)
        assert c is not None and c["units"] > 1e8 and c["SQ_INSTS_VALU"] > 0 and c["TCC_EA0_RDREQ_sum"] > 0, base
        md = ROOT / c["file"]
        assert md.exists() and (ROOT / c["units_file"]).exists()
        # the very number printed in the committed file, per dispatch
        sec = [v for k, v in benchlib.parse_pmc_summary(md).items() if re.search(pattern, k)]
        assert any(abs(v["SQ_INSTS_VALU"] - c["SQ_INSTS_VALU"]) < 1e-6 * c["SQ_INSTS_VALU"] for v in sec)
        cands = {u: per64 for u, per64, *_ in bench.binding_candidates(base)}
        assert abs(cands["valu_issue"] - c["SQ_INSTS_VALU"] / (c["units"] / 64)) < 1e-9 * cands["valu_issue"]
        assert abs(cands["fabric_bandwidth"] - c["TCC_EA0_RDREQ_sum"] / (c["units"] / 64)) < 1e-9 * cands["fabric_bandwidth"]
        bf = bench.binding_floor(base + "[vol]", c["units"], 5.0)
        assert bf["floor_ms"] == max(bf["floors_ms"].values()) and abs(bf["frac"] - bf["floor_ms"] / 5.0) < 1e-12
    # a newer round's files win, and totals are divided by the dispatch count
    (tmp_path / "r98_trilinear_rocprof_summary.md").write_text(
        "## PMC\n\n### `(anonymous namespace)::k_trilinear_splat_b16((anonymous namespace)::GatherArgs)`  (2 dispatches)\n"
        "- SQ_INSTS_VALU: 8e+09\n- TCC_EA0_RDREQ_sum: 1e+08\n- derived: x = 1 / 64\n\n")
    (tmp_path / "r98_trilinear_bench_under_trace.json").write_text(json.dumps({"roofline": {"units_per_launch": 2.0e9}}))
    (tmp_path / "r97_trilinear_rocprof_summary.md").write_text("### `k_trilinear_splat_b16`  (1 dispatches)\n- SQ_INSTS_VALU: 1\n")
    (tmp_path / "r97_trilinear_bench_under_trace.json").write_text(json.dumps({"roofline": {"units_per_launch": 5}}))
    c = benchlib.committed_counters("trilinear", "k_trilinear_splat_b16", profiles=tmp_path)
    assert c["SQ_INSTS_VALU"] == 4e9 and c["TCC_EA0_RDREQ_sum"] == 5e7 and c["units"] == 2.0e9 and "r98" in c["file"]
