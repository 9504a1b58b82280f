"""CPU self-consistency checks of the committed round-5 evidence (profiles/r05_bench.json, the rocprofv3 kernel statistics and
profiles/traffic.json taken on the same box visit, tools/gpu_profiles_r05.sh): the numbers DESIGN.md quotes must follow from
each other, and the kernel names the bench reports must be the ones rocprofv3 and the PMC passes saw."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _bench():
    with open(os.path.join(P, "r05_bench.json")) as f:
        return json.load(f)


def test_headline_line_is_self_consistent():
    d = _bench()
    cfg = d["config"]
    B, T, N = cfg["per_gpu_batch"], cfg["frames"], cfg["n_timesteps"]
    assert (B, T, N, d["n_gpus"]) == (16, 1024, 50, 1) and d["unit"] == "mel-frames/s" and d["higher_is_better"] is True
    assert abs(d["value"] - B * T / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert abs(cfg["ms_per_unet_call"] - d["ms_per_step"] / N) <= 1e-3
    assert "f16" in d["dtype"] and "fp8" in d["dtype"] and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] == "weak"
    ur = d["unet_roofline"]
    hbm = ur["alg_bytes_per_frame_step"] * B * T * N / (d["ms_per_step"] * 1e-3) / 8e12
    assert abs(hbm - ur["hbm_frac"]) <= 2e-3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    # the plan of rounds 1-4 measured in the same run: the headline is the faster one, by the margin DESIGN.md quotes
    old = d["extras"]["config2_bf16x3"]["mel_frames_per_s"]
    assert 1.05 <= d["value"] / old <= 1.15


def test_roofline_block_follows_from_its_own_fields_and_rocprof_agrees():
    d = _bench()
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and r["unit"] == "TFLOP/s" and r["mfma_passes_per_mac"] == 2
    useful = r["alg_gflop_per_launch"] / r["avg_us"] * 1e3           # GFLOP / us = 1e3 TFLOP/s
    assert abs(useful - r["useful_tflops"]) <= 0.01 * useful
    assert abs(r["frac"] - useful / r["peak"]) <= 2e-3               # `frac` is the ALGORITHMIC fraction
    assert abs(r["frac_executed"] - 2 * r["frac"]) <= 2e-3
    assert "conv3x3_ws_kernel" in r["kernel"] and r["avg_us"] <= 190.0
    # rocprofv3 --kernel-trace --stats of the same command on the same box
    with open(os.path.join(P, "r05_rocprof_kernel_stats.csv")) as f:
        rows = {row["Name"]: row for row in csv.DictReader(f)}
    hit = [v for k, v in rows.items() if r["kernel"].replace("gtts::", "") in k]
    assert len(hit) == 1
    assert abs(float(hit[0]["AverageNs"]) / 1e3 - r["avg_us"]) <= 0.03 * r["avg_us"]
    # HBM traffic of that kernel from the FETCH_SIZE / WRITE_SIZE passes: at least the algorithmic bytes, not wildly more
    with open(os.path.join(P, "traffic.json")) as f:
        tj = json.load(f)
    run = [x for x in tj["runs"] if x.get("workload") == "gradtts" and x.get("precision") == "f16f8"]
    assert len(run) == 1 and run[0]["B"] == 16 and run[0]["T"] == 1024
    ent = run[0]["kernels"][r["kernel"]]
    assert 1.0 <= ent["bytes_per_launch"] / (r["alg_mb_per_launch"] * 1e6) <= 1.4


def test_extras_cover_the_other_baseline_configurations():
    ex = _bench()["extras"]
    assert ex["config3_bf16_store"]["ms_per_unet_call"] > 0
    c4 = ex["config4_diffvc_ml6"]
    assert c4["ms_per_unet_call"] <= 80.0 and c4["n30"]["ms_per_unet_call"] > 0 and c4["bf16x3"]["ms_per_unet_call"] > c4["ms_per_unet_call"]
    assert ex["batch1"]["ms_per_unet_call"] < 1.5
    assert ex["train_step"]["hip_ms"] < ex["train_step"]["torch_rocm_ms"]
