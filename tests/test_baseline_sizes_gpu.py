"""BASELINE.json's configurations at their FULL sizes, whole outputs, against the reference itself: cfg1 one 1080p 8UC3 frame, cfg2
a whole 4K frame through cvtColor / filter2D (separate and fused), cfg3 whole 8K CV_32F resize and warpAffine results, cfg4 all 32 frames of one GPU's
shard through cornerHarris and buildPyramid(4), cfg5 the full 3713x2033 matchTemplate result -- GPU vs cv::matchTemplate, and both vs an exact float64
evaluation.  The checker is oracle/_ref/libocvref.so (the real cv:: functions; it travels with the tree), the C restatement where it is absent.
tools/bench_configs.py runs the same gates before it times anything."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_baseline_config_whole_frames_against_the_reference():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    res = bench_configs.parity_gates()                       # asserts bit-exactness / 1e-4 inside
    for key in ("cfg1", "cfg2a", "cfg2b", "cfg2c", "cfg2d", "cfg2e", "cfg3a", "cfg3b", "cfg3c", "cfg4a", "cfg4b", "cfg5"):
        assert key in res, (key, res)
    import orc
    if orc.load_ref() is not None:
        assert "whole 7680x4320 result" in res["cfg3c"] and "32 whole 1080p frames" in res["cfg4a"] and "cv::matchTemplate" in res["cfg5"], res
    print(res)
