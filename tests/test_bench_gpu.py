"""bench.py's contract on the GPU box: the default single-GPU line, and the N > 1 code path -- self-spawn, frame sharding, plan-time broadcasts,
barrier + max-over-ranks timing, the cfg4 / cfg5 legs -- exercised by two ranks that share the one GPU of the test box over gloo
(MI355CV_BENCH_SHARED_GPU=1: a code-path test, the line says so; the real multi-GPU numbers are the driver's to take)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None):
    e = dict(os.environ); e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                      # rank 0 prints ONE JSON line
    assert lines[0] == p.stdout.strip().splitlines()[-1] and len(lines[0]) < 4096, len(lines[0])      # the LAST stdout line, small enough for the driver's parser
    compact = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data", "scaling", "config", "roofline"):
        assert k in compact, k
    detail = [l for l in p.stderr.splitlines() if l.startswith('{"metric"')]
    assert len(detail) == 1                                       # the whole record goes to stderr (and bench_detail.json)
    d = json.loads(detail[0])
    assert d["value"] == compact["value"] and d["roofline"]["frac"] == compact["roofline"]["frac"]
    d["_compact"] = compact
    return d


def test_single_gpu_line():
    d = run(["--batch", "256", "--frames-per-launch", "128", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "Mpix/s" and d["dtype"] == "u8" and d["scaling"] == "weak"
    r = d["roofline"]
    # traffic is measured in the run itself (rocprofv3 --pmc passes over a child, calibrated on a copy of known size)
    t = r["traffic_detail"]
    assert r["traffic"] and 0.95 <= t["traffic_over_algorithmic"] <= 1.5 and abs(t["fetch_calibration"] - 2.0) < 0.2 and abs(t["write_calibration"] - 1.0) < 0.2, t
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["config"]["launches_per_step"] == 2 and r["launches_timed"] == 6
    assert r["algorithmic_bytes_per_launch"] == 2 * 128 * 3840 * 2160
    assert d["parity"]["result"] == "bit-exact"


def test_default_command_line_is_parseable_and_small():
    """the driver's own command shape (other configs, CPU baseline and PMC passes ON; only the step count reduced): the last stdout line must json.loads,
    stay under 4 KB and carry roofline + cpu_baseline + the summary (r04's 20.5 KB line came back "parsed": null)"""
    d = run(["--steps", "2", "--warmup", "1", "--batch", "512"])
    c = d["_compact"]
    assert c["roofline"]["bound"] == "hbm" and c["roofline"]["traffic"] and 0 < c["roofline"]["frac"] < 1
    cb = c["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and len(cb["sample"]) <= 200
    assert "headline" in c["summary"] and "cfg5" in c["summary"] and "cfg3c" in c["summary"]
    assert isinstance(d["other_configs"], list) and len(d["other_configs"]) > 40          # the long form is in the detail record only
    assert "other_configs" not in c and "traffic_detail" not in c["roofline"]
    assert os.path.exists(os.path.join(ROOT, "bench_detail.json"))


def test_two_ranks_share_one_gpu():
    d = run(["--gpus", "2", "--batch", "256", "--steps", "2", "--warmup", "1"], {"MI355CV_BENCH_SHARED_GPU": "1"})
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and "test_mode" in d
    assert d["value"] > 0 and abs(d["per_gpu_mpix_s"] * 2 - d["value"]) / d["value"] < 1e-3
    legs = d["other_configs"]
    assert len(legs) == 2 and all(l["n_gpus"] == 2 for l in legs)
    assert "128 frames / GPU" in legs[0]["config"]                # 256 frames sharded over 2 ranks


def test_eight_ranks_rehearsal_on_one_gpu():
    """the driver's --gpus 8 launch plan at a reduced batch (8 ranks share the one GPU of the test box over gloo): spawn, sharding of the 256-frame cfg4 batch
    into 32 frames per rank, the plan broadcast, max-over-ranks timing -- so that the first real 8-GPU run is not the first run of this path"""
    d = run(["--gpus", "8", "--batch", "128", "--steps", "2", "--warmup", "1", "--no-pmc"], {"MI355CV_BENCH_SHARED_GPU": "1"})
    assert d["n_gpus"] == 8 and d["config"]["ranks"] == 8 and "test_mode" in d
    assert abs(d["per_gpu_mpix_s"] * 8 - d["value"]) / d["value"] < 1e-3
    legs = d["other_configs"]
    assert len(legs) == 2 and "32 frames / GPU" in legs[0]["config"]
    assert "launch plan" in d["config"]["collective"]


def test_refuses_more_gpus_than_visible():
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "refusing" in p.stderr
