"""bench.py: the driver's contract (one JSON line, the keys and types it reads) and the synthetic
scalar generator."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gen_scalars_are_canonical_and_deterministic():
    import bench
    for name in ("bn254_g1", "grumpkin"):
        m = bench.ORDER[name]
        a = bench.gen_scalars(5000, m, 42); b = bench.gen_scalars(5000, m, 42); c = bench.gen_scalars(5000, m, 43)
        assert a.shape == (5000, 32) and a.dtype == np.uint8
        assert np.array_equal(a, b) and not np.array_equal(a, c)
        vals = [int.from_bytes(r.tobytes(), "little") for r in a]
        assert max(vals) < m and len(set(vals)) > 4990
        assert max(vals) > m >> 3          # spread over the whole range, not a narrow band


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--workload", "lhs"]], ids=["msm", "lhs"])
def test_bench_prints_one_contract_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--logn", "12",
           "--cpu-sample-log", "10"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and "workload" in d["config"] and d["config"]["bit_exact"] is True
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-5 and ro["kernel"] == "k_accum1" and ro["kernel_ms"] > 0
    vi = ro["valu_issue"]
    assert 0 < vi["frac"] <= 1.0 and vi["peak_clock_ghz"] == 2.4          # priced at the guide's maximum clock: <= 1 by construction
    assert vi["clock_ghz_measured"] is None or 0.5 < vi["clock_ghz_measured"] <= 2.5
    assert "timed result vs walk identity" in d["config"]["bit_exact_checks"] and "sample vs oracle" in d["config"]["bit_exact_checks"]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and isinstance(cb["sample"], str)
    # the CPU leg follows the workload: the lhs line times the serial compute_lhs_witness restatement, not best_multiexp
    assert ("compute_lhs_witness" in cb["sample"]) == (extra != [])
    if extra:
        assert cb["cores"] == 1


@pytest.mark.gpu
def test_bench_lhs_witness_line():
    """--workload lhs_witness: compute_lhs_witness in full as a contract-shaped, self-verified line"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "lhs_witness", "--steps", "2", "--warmup", "1", "--logn", "10",
           "--cpu-sample-log", "7"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["unit"] == "pairs/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None
    assert d["config"]["bit_exact"] is True and "timed carry vs walk identity" in d["config"]["bit_exact_checks"]
    assert any("every coefficient" in c for c in d["config"]["bit_exact_checks"])
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and ro["traffic"] is None
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and "k_ntt_tile" in ro["kernel"] and ro["kernel_ms"] > 0
    assert 0 < ro["valu_butterflies"]["frac"] <= 1.0
    assert set(ro["phases_ms"]) == {"msm_core", "point_lists", "merge_forest", "coefficient_copy"}
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "compute_lhs_witness" in cb["sample"]
