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
    for key in ("hbm_traffic_frac", "l2_frac", "td_busy", "lane_util"):
        assert key in r, key
    assert j["primary_mrays_s"] > 0 and j["incoherent_mrays_s"] > 0
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert c["reference_lavapipe"] == "unavailable"
