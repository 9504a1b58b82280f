"""CPU checks of the committed round-3 evidence (profiles/): the numbers a reader would re-derive by hand must agree with each
other -- the bench line with itself, with the rocprofv3 summary of the same box visit and with the PMC traffic file."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _bench():
    with open(os.path.join(P, "r03_bench.json")) as f:
        return json.load(f)


def test_bench_line_is_self_consistent():
    d = _bench()
    cfg = d["config"]
    frames = d["n_gpus"] * cfg["per_gpu_batch"] * cfg["frames"]
    assert d["value"] == pytest.approx(frames / (d["ms_per_step"] * 1e-3), rel=1e-3)
    assert cfg["ms_per_unet_call"] == pytest.approx(d["ms_per_step"] / cfg["n_timesteps"], rel=1e-3)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4)
    # executed = three bf16 MFMA passes per useful MAC
    assert r["frac"] == pytest.approx(3 * r["frac_algorithmic"], rel=2e-3)
    assert r["achieved"] == pytest.approx(3 * r["alg_gflop_per_launch"] / r["avg_us"] * 1e3, rel=2e-3)
    assert r["traffic"] >= r["alg_mb_per_launch"] * 1e6          # measured HBM bytes cannot be below the algorithmic ones
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["value"] > 0
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak"


def test_rocprof_summary_of_the_same_visit_agrees():
    d = _bench()
    for fn, roof in (("r03_rocprof_kernel_stats.csv", d["roofline"]),
                     ("r03_rocprof_kernel_stats_conv_ws.csv", d["extras"]["config2_conv_ws"]["roofline"])):
        with open(os.path.join(P, fn)) as f:
            rows = list(csv.DictReader(f))
        hit = [r for r in rows if r["Name"].replace("void ", "").startswith(roof["kernel"] + "(")]
        assert len(hit) == 1, roof["kernel"]
        assert float(hit[0]["AverageNs"]) / 1e3 == pytest.approx(roof["avg_us"], rel=0.03)


def test_traffic_file_covers_both_kernels_and_config3():
    d = _bench()
    with open(os.path.join(P, "traffic.json")) as f:
        runs = json.load(f)["runs"]
    names = {(r["workload"], r["precision"]): r for r in runs}
    assert ("gradtts", "bf16x3") in names and ("gradtts-multispk", "bf16-store") in names
    allk = {}
    for r in runs:
        allk.update(r["kernels"])
    for roof in (d["roofline"], d["extras"]["config2_conv_ws"]["roofline"], d["extras"]["config3_bf16_store"]["roofline"]):
        assert roof["kernel"] in allk, roof["kernel"]
        assert allk[roof["kernel"]]["bytes_per_launch"] == roof["traffic"]
        assert not roof["kernel"].startswith("_Z")               # demangled (rocprofv3 leaves the bf16 instances mangled)
