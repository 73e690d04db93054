"""bench.py's stdout contract without a GPU: emit() turns an arbitrarily large record into ONE line under 4 KB that keeps every key the driver parses
(the 20.5 KB line of round 4 came back "parsed": null), with the whole record in bench_detail.json / on stderr."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def fake_record(n_rows):
    rows = [{"config": f"cfg{i} some long configuration name " + "x" * 80, "frames": 40, "ms": 0.4, "frac": 0.7, "achieved_GBs": 5600.0} for i in range(n_rows)]
    return {"metric": "Mpix/s per GPU (4K CV_8U Gaussian5x5) + achieved HBM GB/s vs roofline, 1/2/4/8 GPUs", "value": 2.9e6, "unit": "Mpix/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 26.1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "w" * 200, "frames_per_gpu": 9216, "frames_per_launch": 512, "launches_per_step": 18, "sharding": "frames x1", "ranks": 1, "collective": "none"},
            "roofline": {"bound": "hbm", "achieved": 5840.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.73, "traffic": 9.04e9, "traffic_detail": {"k": "v" * 3000},
                         "kernel": "k_binomial_roll2<5,1,true,false,4,true>", "avg_launch_ms": 1.41, "algorithmic_bytes_per_launch": 8493465600,
                         "launches_timed": 360, "launch_ms": {"median": 1.4}, "step_ms": {"median": 25.0}, "measured_copy_GBs": 6300.0},
            "cpu_baseline": {"value": 51500.0, "unit": "Mpix/s", "cores": 16, "kind": "reference", "one_thread_Mpix_s": 4200.0, "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                             "sample": "s" * 900, "sample_short": "t" * 150},
            "parity": {"result": "bit-exact", "frames": 19},
            "other_configs": rows,
            "summary_us_per_frame_and_frac": {f"key{i}": [1.234, 0.5678] for i in range(30)}}


def test_compact_line_small_and_complete(capsys, tmp_path, monkeypatch):
    b = load_bench()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    b.emit(fake_record(200))
    cap = capsys.readouterr()
    out = cap.out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < 4096
    d = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity", "summary"):
        assert k in d, k
    assert d["roofline"]["frac"] == 0.73 and d["roofline"]["traffic"] == 9.04e9 and "traffic_detail" not in d["roofline"]
    assert d["cpu_baseline"]["cores"] == 16 and d["cpu_baseline"]["sample"] == "t" * 150
    assert "other_configs" not in d
    full = json.loads(open(tmp_path / "bench_detail.json").read())
    assert len(full["other_configs"]) == 200 and full["roofline"]["traffic_detail"]
    assert json.loads(cap.err.strip().splitlines()[-1]) == full


def test_oversize_summary_is_shed_not_fatal(capsys, tmp_path, monkeypatch):
    b = load_bench()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    r = fake_record(3)
    r["summary_us_per_frame_and_frac"] = {f"key{i}": [1.234, 0.5678] for i in range(400)}
    b.emit(r)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < 4096
    d = json.loads(out[0])
    assert "summary" not in d and d["roofline"]["frac"] == 0.73 and d["value"] == 2.9e6
