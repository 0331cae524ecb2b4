"""CPU test: the committed benchmark line (profiles/r1_bench_line.json, written by `python bench.py` on an MI355X) carries
every field of the benchmark contract, and bench.py's byte accounting is self-consistent."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", "r1_bench_line.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "elements/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes / measured launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    # value = elements * steps / time
    assert abs(d["value"] - d["config"]["elements"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_spmv_byte_accounting():
    import bench
    nnz, n = 348165189, 4102893
    assert bench.spmv_bytes(nnz, n) == 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n
    assert bench.spmv_stored_bytes(nnz, nnz // 9, n // 3, n) < bench.spmv_bytes(nnz, n)
    # P2 tets: 55 pairs per element, 6 points
    assert bench.assembly_flops(10, 10, 6, False) == 10 * 55 * 6 * 94.0
