"""bench.py itself: its launcher (`--gpus N` starts N ranks), the JSON contract of a real run, and its byte accounting.

The CPU tier checks the launcher without touching a GPU (`--spawn-check`: every rank reports and exits) and the
accounting functions; the GPU tier runs the benchmark on small grids -- one rank, and two ranks sharing the one card over
the gloo backend -- and checks the line it prints."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_gpus_flag_starts_that_many_ranks():
    d = _run(["--gpus", "2", "--spawn-check"])
    assert d == {"spawn_check": True, "n_gpus": 2, "rank": 0}
    d = _run(["--gpus", "4", "--spawn-check"])
    assert d["n_gpus"] == 4


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--spawn-check"], env=env, capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "does not match" in p.stderr


def test_spmv_byte_accounting():
    sys.path.insert(0, ROOT)
    import bench
    nnz, n = 348165189, 4102893
    assert bench.spmv_csr_bytes(nnz, n) == 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n
    # the roofline is priced on what the node-block format moves: one column index per nine values
    assert bench.spmv_bytes(nnz, nnz // 9, n // 3, n) == 8.0 * nnz + 4.0 * (nnz // 9) + 8.0 * (n // 3 + 1) + 16.0 * n
    assert bench.spmv_bytes(nnz, nnz // 9, n // 3, n) < bench.spmv_csr_bytes(nnz, n)
    # P2 tets: 55 pairs per element, 6 points
    assert bench.assembly_flops(10, 10, 6, False) == 10 * 55 * 6 * 94.0


def _check_contract(d, n_gpus):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["unit"] == "elements/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["n_gpus"] == n_gpus and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert 0.0 < r["frac"] <= 1.0 and r["csr_equivalent_bytes"] > r["algorithmic_bytes"]
    assert r["traffic"] is None or r["traffic_source"].startswith("profiles/")
    assert abs(d["value"] - d["config"]["elements"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_line_single_gpu_small():
    d = _run(["--grid", "6", "--steps", "2", "--warmup", "1", "--cg-max-iter", "40", "--cpu-n", "4", "--cpu-cg-iters", "10"])
    _check_contract(d, 1)
    assert "NeoHookean" in d["config"]["workload"] and d["config"]["elements"] == 6 * 6 ** 3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["unit"] == d["unit"]
    assert "linear_elasticity_1M" in d and d["linear_elasticity_1M"]["config"]["elements"] == 998250


@pytest.mark.gpu
def test_bench_two_ranks_on_one_card_over_gloo():
    d = _run(["--gpus", "2", "--grid", "4", "--steps", "1", "--warmup", "1", "--cg-max-iter", "60", "--no-cpu-baseline"],
             {"PFX_BENCH_BACKEND": "gloo"})
    _check_contract(d, 2)
    assert d["config"]["elements"] == 2 * 6 * 4 ** 3
    assert "NOT RCCL" in d["config"]["parallelism"] and d["config"]["halo_bytes_per_exchange_rank0"] > 0
    # the converged leg: one AMG hierarchy that spans the ranks
    a = d["amg_solve"]
    assert "error" not in a and a["cg_rel_residual"] <= 1e-10 and "spans" in a["preconditioner"]


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_card_over_gloo():
    """The launch shape of the 8-GPU run, rehearsed on one card: eight ranks, the distributed iteration inside the library
    (host-staged transport), halo traffic on rank 0, the converged AMG leg."""
    d = _run(["--gpus", "8", "--grid", "6", "--steps", "1", "--warmup", "1", "--cg-max-iter", "60", "--no-cpu-baseline"],
             {"PFX_BENCH_BACKEND": "gloo"})
    _check_contract(d, 8)
    assert d["config"]["elements"] == 8 * 6 * 6 ** 3
    assert d["config"]["halo_bytes_per_exchange_rank0"] > 0 and "pfx_dist_cg_dev" in d["config"]["distributed_iteration"]
    a = d["amg_solve"]
    assert "error" not in a and a["cg_rel_residual"] <= 1e-10


@pytest.mark.gpu
def test_bench_dry_run_reports_per_rank_memory():
    d = _run(["--dry-run", "--gpus", "8", "--grid", "8", "--cells-per-rank", "2"])
    assert d["dry_run"] and d["n_gpus"] == 8 and d["fits"] and len(d["ranks"]) == 3
    assert d["elements_global"] == 6 * 8 * 8 * 16
    for r in d["ranks"]:
        assert r["bytes_with_solver_copy_and_cg_vectors"] > r["bytes_after_pattern"] > 0 and r["halo_nodes"] > 0


@pytest.mark.gpu
def test_bench_reports_its_line_when_the_optional_leg_of_a_multi_rank_run_does_not_finish():
    # a limit no AMG set-up can meet: every rank gives up, rank 0 prints the measured line once, all exit with 0
    d = _run(["--gpus", "2", "--grid", "4", "--steps", "1", "--warmup", "1", "--cg-max-iter", "60", "--no-cpu-baseline"],
             {"PFX_BENCH_BACKEND": "gloo", "PFX_BENCH_AMG_LIMIT_S": "0.001"})
    _check_contract(d, 2)
    assert "no result within" in d["amg_solve"]["error"]
