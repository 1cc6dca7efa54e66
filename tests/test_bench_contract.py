"""The bench line the driver parses: helper arithmetic on the CPU, the JSON contract on the GPU (a tiny run of bench.py
itself: every key the contract and VERDICT ask for is there and self-consistent)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_algorithmic_figures_follow_the_survey_formulas():
    import bench
    from slslam_amd import synth
    w = synth.make_window(3, num_lines=120)
    C, Cf, L, M = w["num_cameras"], w["num_free_cameras"], w["num_lines"], len(w["camera_index"])
    counts = [(C, Cf, L, M)] * 3
    assert bench.algorithmic_bytes_linearise(counts) == 3 * (72 * M + 32 * L + 48 * C + 8 * (6 * Cf) ** 2)
    assert bench.algorithmic_bytes_backsub(counts) == 3 * (72 * M + 64 * L + 96 * C + 48 * Cf)
    free = np.asarray(w["fixed_index"]).reshape(-1, 2)[:, 0] == 0
    kf = np.bincount(np.asarray(w["line_index"])[free], minlength=L).astype(float)
    assert bench.algorithmic_flops_linearise([w]) == pytest.approx(1120.0 * M + 288.0 * (kf ** 2).sum())
    assert bench.reduced_solve_mfma_count(60) == 64 and bench.reduced_solve_mfma_count(16) == 0
    assert bench.baseline_metric() == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    probe = bench.ceres_probe()
    assert set(probe) == {"ceres_available", "found"} and isinstance(probe["ceres_available"], bool)


@pytest.mark.gpu
def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--windows", "8",
           "--lines", "200", "--no-cpu-baseline", "--no-extra-configs"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_backsub", "reduced_solve_mfma", "kernel_ms_per_step"):
        assert k in d, k
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["value"] == pytest.approx(d["lm_iterations"] / (d["ms_per_step"] * d["steps"] * 1e-3), rel=1e-6)
    assert d["lm_iterations"] <= 8 * 10 * 2                   # at most max_num_iterations per window and step
    for key in ("roofline", "roofline_backsub"):
        rf = d[key]
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "binding" in rf
        assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"]) and 0 < rf["frac"] < 1
        assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9)
    # the dominant kernel's time fits inside the step
    assert d["kernel_ms_per_step"]["linearise_schur"] < d["ms_per_step"]
    # multi-rank evidence fields are there at N = 1 too; the results of the checked windows are reproduced bit for bit by a
    # solve of the same window ids in a different batch (what rank 0 does with the other ranks' results at N > 1)
    assert d["rccl_ranks_seen"] == 1 and d["distinct_devices_seen"] == 1 and len(d["per_rank_ms_per_step"]) == 1
    assert d["per_rank_ms_per_step"][0] == pytest.approx(d["ms_per_step"], rel=0.2)
    chk = d["results_check"]
    assert chk["window_ids"] == [0, 1, 2, 3] and len(chk["crc32_of_gathered_parameters"]) == 4
    assert chk["bitwise_equal_to_rank0_resolve"] is True and chk["max_abs_diff"] == 0.0
    assert "traffic_source" in d["roofline"]
