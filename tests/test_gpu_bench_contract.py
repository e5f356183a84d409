"""bench.py prints exactly one JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "1", "--pairs", "4", "--steps", "2",
                        "--warmup", "1", "--cpu-seconds", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mpixels/s" and d["value"] > 0 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert [c["pair"] for c in d["checked"]] == [0, 3]           # first and last pair of the batch
    for c in d["checked"]:
        assert c["rank"] == 0 and c["confidence_bit_exact"] is True and c["disparity_max_abs_lsb"] <= 1
    assert d["cpu_baseline"]["one_thread"]["scalar"] > 0 and d["cpu_baseline"]["one_thread"]["ref_simd"] > 0
    assert d["roofline"]["traffic_source"] is None or "kernel_sources_sha16_now" in d["roofline"]["traffic_source"]
    v = d["views_to_filtered"]["bm"]                 # extra leg: device matcher feeding the filter
    assert "error" not in v, v
    assert v["matcher_ms_per_pair"] > 0 and v["filter_ms_per_pair"] > 0 and v["num_disparities"] % 16 == 0
    ng = d["natural_guide"]                          # extra leg: the same call on a natural-image guide (round 3)
    assert "error" not in ng, ng
    assert ng["Mpixels_per_s"] > 0 and 0 < ng["table_indices_beyond_lds_head"] < 0.2
    assert ng["checked"]["confidence_bit_exact"] is True and ng["checked"]["disparity_max_abs_lsb"] <= 1
    nr = d["next_rows"]                              # extra legs: SURVEY 8(f) rows N1 / N2 (round 3)
    assert "error" not in nr, nr
    n1, n2 = nr["down_scaled_path"], nr["fgs_one_shot"]
    assert n1["ms_per_call"] > 0 and n1["maps"] == "320x240"
    assert n1["checked"]["confidence_bit_exact"] is True and n1["checked"]["disparity_max_abs_lsb"] <= 1
    assert n2["create_plus_filter_ms_per_call"] >= n2["filter_alone_ms_per_call"] > 0
    assert n2["checked"]["exact_solver_bit_exact"] is True and n2["checked"]["wave_solver_max_abs_diff"] < 0.05
