"""The driver's contract with bench.py, checked on the GPU box on a reduced volume: ONE JSON line with the metric of
BASELINE.json, the roofline of the dominant kernel measured live, the per-stage HBM rooflines, the measured parity of
the benched precision and the CPU baseline; and the training leg (configs[4]) on a reduced crop."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=900,
                       env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("precision", ["fp16", "split"])
def test_eval_line_contract(precision):
    d = _run("--steps", "1", "--warmup", "1", "--shape", "512,512,64", "--precision", precision, "--cpu-budget", "1",
             "--also-steps", "1", "--also-train-steps", "2", "--also-train-shape", "64,64,64")
    assert d["metric"].startswith("Mvoxels/s end-to-end") and d["unit"] == "Mvoxels/s" and d["n_gpus"] == 1
    assert d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 512 * 512 * 64 / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    assert d["config"]["workload"].startswith("512x512x64") and d["config"]["precision"] == precision
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and r["launches"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.02 < r["frac"] < 1.0
    assert r["layers"] and all(v["tflops"] > 0 for v in r["layers"].values())
    for k, b in (("roofline_gate", 17), ("roofline_ccl", 9), ("roofline_assign", 64)):
        assert d[k]["bound"] == "hbm" and d[k]["peak"] == 8000.0 and d[k]["algorithmic_bytes_per_voxel"] == b and d[k]["achieved"] > 0
    par = d["parity_vs_fp32_mode"]
    assert par["max_abs"] <= (1e-2 if precision == "fp16" else 1e-3)   # the line states the tolerance of what it measured
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mvoxels/s" and "936" in c["sample"]
    assert d["config"]["instances"] > 0   # the injected blob field went through stages 2-3
    assert d["box"]["mfma_probe_tflops"] > 500 and d["box"]["mfma_probe_varying_operands_tflops"] > 500
    assert d["box"]["band"] in ("slow", "typical", "fast") and d["box"]["hbm_copy_GBps"] > 500
    if precision == "split":
        assert "also" not in d
        return
    # the default (fp16) line also carries the tolerance-meeting precision and the training step, same process
    sp, tr = d["also"]["split"], d["also"]["train_bf16"]
    assert sp["value"] > 0 and sp["unit"] == "Mvoxels/s" and sp["steps"] == 1 and sp["ms_per_step"] > d["ms_per_step"]
    assert abs(sp["value"] - 512 * 512 * 64 / sp["ms_per_step"] / 1e3) < 0.01 * sp["value"]
    assert sp["parity_vs_fp32_mode"]["max_abs"] <= 1e-3 and sp["parity_vs_fp32_mode"]["north_star_tolerance"] == 1e-3
    assert sp["roofline"]["bound"] == "mfma" and 0.005 < sp["roofline"]["frac"] < 1.0 and sp["roofline"]["launches"] > 0
    mx = d["also"]["mix8"]   # split with fp8 correction products in the 3x3x3 convs: same tolerance, its layers beside split's
    assert mx["value"] > 0 and mx["parity_vs_fp32_mode"]["max_abs"] <= 1e-3 and "fp8" in mx["dtype"]
    assert {"enc0.1", "enc1.0", "mid.0", "dec1.0", "dec0.0", "dec0.1"} <= set(mx["layer_launch_ms"]) and mx["ms_per_step"] < sp["ms_per_step"]
    assert tr["dtype"] == "bf16" and tr["steps"] == 2 and tr["ms_per_step"] > 0 and tr["value"] > 0
    assert tr["roofline"]["bound"] == "mfma" and tr["roofline"]["achieved"] > 0
    assert tr["config"]["precision"] == "bf16" and all(0 < v < 3 for v in tr["config"]["losses"][:3])
    # BASELINE configs[1] in the same line: one 512x512x128 tile through the conv + GN/SiLU stack, its own roofline
    c1 = d["also"]["conv_c1"]
    assert c1["forward_ms"] > 0 and c1["dtype"] == "f16" and "512x512x128" in c1["metric"]
    assert c1["roofline"]["bound"] == "mfma" and 0.05 < c1["roofline"]["frac"] < 1.0 and 0.05 < c1["roofline"]["encoder_frac"] < 1.0
    assert {"enc0.1", "enc1.0", "mid.0", "dec1.0", "dec0.0", "dec0.1"} <= set(c1["roofline"]["layers"])
    # the timed steps allocate nothing on the device (grow-only activation buffers) and the line says so
    assert sp["allocator_in_timed_steps"]["device_mallocs"] == 0 and len(sp["step_ms_min_max"]) == 2
    assert d["config"]["inject_layout"] == "contiguous per-tile blocks"


def test_eval_line_two_ranks_gloo_rehearsal():
    """`python bench.py --gpus 2` without a launcher, rehearsed on the one-GPU box: two rank processes share the device and
    exchange through gloo (SKOOTS_DIST_BACKEND=gloo).  Rank 0 prints ONE line for the 2-rank job: the Z-sharded pipeline
    with its per-rank tile counts and per-exchange accounting, the per-rank contiguous injection, max-over-ranks timing."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SKOOTS_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--shape",
                        "512,512,128", "--no-cpu-baseline", "--no-parity", "--tile-batch", "16"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["instances"] > 0
    mg = d["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and mg["backend"] == "gloo" and len(mg["tiles_per_rank"]) == 2
    assert abs(mg["tiles_per_rank"][0] - mg["tiles_per_rank"][1]) <= 1 and sum(mg["slab_planes"]) == 128
    assert {"block_exchange", "label_seam_planes", "label_meta", "label_gather", "vector_halo"} <= set(mg["comm_rank0_per_step"])
    assert "also" not in d   # the extra legs belong to the N = 1 line


def test_eval_line_two_streams_keeps_a_roofline():
    """--streams 2: the timed steps overlap two tile batches; the per-launch roofline then comes from the single-stream
    warm-up steps and says so."""
    d = _run("--steps", "1", "--warmup", "2", "--shape", "512,512,64", "--streams", "2", "--tile-batch", "32", "--no-also", "--no-cpu-baseline")
    r = d["roofline"]
    assert d["config"]["streams"] == 2 and "warm-up" in r["timed_over"] and r["launches"] > 0 and 0.02 < r["frac"] < 1.0
    assert d["roofline_assign"]["launches"] >= 1 and d["config"]["instances"] > 0


def test_train_line_contract():
    d = _run("--config", "train", "--steps", "2", "--warmup", "1", "--shape", "64,64,64", "--cpu-budget", "1")
    assert d["dtype"] == "bf16" and d["config"]["precision"] == "bf16" and d["n_gpus"] == 1 and d["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["launches"] > 0 and d["roofline"]["achieved"] > 0
    assert set(d["roofline"]["ms_per_step"]) == {"conv_fwd_dgrad", "conv_wgrad"}
    assert all(0 < v < 3 for v in d["config"]["losses"][:3])
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
