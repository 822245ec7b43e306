"""bench.py prints one JSON line with the fields its callers read (the
benchmark contract), here on a small configuration."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", "dragon_small", "--width", "320", "--height", "180",
         "--spp-per-step", "4", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"],
        capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["unit"] == "Mrays/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["data"] == "synthetic" and j["dtype"] == "f32"
    assert isinstance(j["config"]["workload"], str) and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # fetches shared by a wave are billed once: no more node fetches than node visits
    assert 0 < r["node_fetches_per_ray"] <= r["nodes_per_ray"]
    for key in ("hbm_traffic_frac", "hbm_achievable_frac", "l2_frac", "td_busy", "lane_util", "level", "level_peak",
                "frac_at_level", "billed"):
        assert key in r, key
    # 54k triangles: the tree fits one XCD's L2, so the bytes that must move are billed, and those stay below the peak
    assert r["level"] == "l2" and r["billed"] == "compulsory" and 0 < r["frac"] <= 1
    assert j["metric"] == "Mrays/s (whole node), dragon_small 320x180"
    assert j["rank_elapsed_ms"]["spread"] == 0 and j["reduce_ms"] is None  # one rank
    assert j["primary_mrays_s"] > 0 and j["incoherent_mrays_s"] > 0
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert c["reference_lavapipe"] == "unavailable"


def test_a_tree_that_fits_l1_never_shows_more_than_the_peak():
    """The Cornell box (32 triangles, 3 nodes) lives in every CU's L1: billing 80
    bytes per node visit would count bytes that never leave the CU against HBM
    (the per-visit rate there exceeds 8 TB/s). The line bills the bytes that must
    move and says which level holds the tree."""
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", "cornell", "--width", "1024", "--height", "1024",
         "--max-depth", "8", "--spp-per-step", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
        capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    r = j["roofline"]
    assert r["level"] == "l1" and r["billed"] == "compulsory" and r["frac_at_level"] is None
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_gbs"] > r["achieved"]  # the per-visit rate is still reported, beside the billed one
    assert j["cpu_baseline"] is None and j["metric"].endswith("cornell 1024x1024")
