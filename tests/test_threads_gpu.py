"""Re-entrancy (SURVEY §8b "Threading"): cv:: functions may be called concurrently from many host threads, so the hooks keep
per-thread streams / staging pools.  Eight threads hammer different hooks on host arrays (staged) and device tensors at once;
every result must equal the oracle's."""
import os
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_concurrent_hooks(orc):
    import opencv_amd as cv
    rng = np.random.default_rng(99)
    img = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    k = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32)
    M = cv.getRotationMatrix2D((160.0, 120.0), 11.0, 0.9)
    want = {
        "gauss": orc.orc_gaussianBlurBinomialU8(img, 5, 4),
        "filter": orc.orc_filter2D(gray, -1, k),
        "box": orc.orc_boxFilter(img, -1, (5, 5)),
        "sobel": orc.orc_Sobel(gray, 3, 1, 0, 3),
        "gray": orc.orc_cvtColor(img, 6),
        "resize": orc.orc_resize(img, (200, 150)),
        "warp": orc.orc_warpAffine(img, M, (320, 240)),
        "dilate": orc.orc_morph(1, gray),
        "thresh": orc.orc_threshold(gray, 100, 255, 0)[1],
        "harris": orc.orc_cornerHarris(gray, 2, 3, 0.04),
    }
    calls = {
        "gauss": lambda a, g: cv.GaussianBlur(a, (5, 5), 0),
        "filter": lambda a, g: cv.filter2D(g, -1, k),
        "box": lambda a, g: cv.boxFilter(a, -1, (5, 5)),
        "sobel": lambda a, g: cv.Sobel(g, cv.CV_16S, 1, 0, 3),
        "gray": lambda a, g: cv.cvtColor(a, cv.COLOR_BGR2GRAY),
        "resize": lambda a, g: cv.resize(a, (200, 150)),
        "warp": lambda a, g: cv.warpAffine(a, M, (320, 240), cv.INTER_LINEAR | cv.WARP_INVERSE_MAP),
        "dilate": lambda a, g: cv.dilate(g),
        "thresh": lambda a, g: cv.threshold(g, 100, 255, 0)[1],
        "harris": lambda a, g: cv.cornerHarris(g, 2, 3, 0.04),
    }
    errors = []

    def worker(tid):
        try:
            names = list(calls)
            dimg, dgray = (torch.from_numpy(img).cuda(), torch.from_numpy(gray).cuda()) if tid % 2 else (img, gray)
            for it in range(12):
                n = names[(tid + it) % len(names)]
                got = calls[n](dimg, dgray)
                got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
                if got.dtype == np.float32:
                    import orc as O
                    ok = O.rel_err(got, want[n]) <= 1e-4
                else:
                    ok = np.array_equal(got, want[n])
                if not ok:
                    errors.append((tid, it, n))
        except Exception as e:                      # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_device_binding_per_thread(orc):
    """SURVEY §8e inside ONE process: a host thread binds its hooks to a device ordinal (mi355cv_setDevice), contexts are per thread and device,
    an ordinal that does not exist is refused, and the host program's current device is left as it was."""
    import ctypes
    import opencv_amd as cv
    from opencv_amd import _lib
    L = _lib.lib
    n = L.mi355cv_deviceCount()
    assert n == torch.cuda.device_count() >= 1
    assert L.mi355cv_setDevice(n) == -1 and b"visible" in L.mi355cv_lastError()        # loud, not a silent fallback to device 0
    assert L.mi355cv_setDevice(0) == 0 and L.mi355cv_getDevice() == 0
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (2 * n, 270, 480), dtype=np.uint8)
    want = [orc.orc_gaussianBlurBinomialU8(f, 5, 4) for f in frames]
    got, errors = {}, []

    def worker(d):
        try:
            assert L.mi355cv_setDevice(d) == 0 and L.mi355cv_getDevice() == d
            for f in range(d, 2 * n, n):                       # frames sharded by index over the devices, one host thread each
                src = torch.from_numpy(frames[f]).to(f"cuda:{d}")
                got[f] = cv.GaussianBlur(src, (5, 5), 0).cpu().numpy()
                # a raw C-ABI call on memory of this thread's device, own stream
                L.mi355cv_resetStream()
                p = L.mi355cv_deviceAlloc(2 * frames[f].size)
                assert p
                assert L.mi355cv_upload(ctypes.c_void_p(p), frames[f].ctypes.data, frames[f].size) == 0
                rc = L.mi355cv_gaussianBlurBinomial(ctypes.c_void_p(p), 480, ctypes.c_void_p(p + frames[f].size), 480, 480, 270, 0, 1, 0, 0, 0, 0, 5, 4)
                back = np.empty_like(frames[f])
                assert rc == 0 and L.mi355cv_download(back.ctypes.data, ctypes.c_void_p(p + frames[f].size), back.size) == 0
                assert np.array_equal(back, want[f])
                L.mi355cv_deviceFree(ctypes.c_void_p(p))
        except Exception as e:                                  # noqa: BLE001
            errors.append((d, repr(e)))

    before = torch.cuda.current_device()
    ts = [threading.Thread(target=worker, args=(d,)) for d in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert torch.cuda.current_device() == before
    for f in range(2 * n):
        assert np.array_equal(got[f], want[f]), f


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (the driver's 8-GPU node); a 1-GPU box runs test_device_binding_per_thread")
def test_image_on_another_device_is_declined():
    import opencv_amd as cv
    from opencv_amd import _lib
    import ctypes
    L = _lib.lib
    a = torch.zeros((64, 64), dtype=torch.uint8, device="cuda:1")
    d = torch.empty_like(a)
    assert L.mi355cv_setDevice(0) == 0
    L.mi355cv_resetStream()
    rc = L.mi355cv_gaussianBlurBinomial(ctypes.c_void_p(a.data_ptr()), 64, ctypes.c_void_p(d.data_ptr()), 64, 64, 64, 0, 1, 0, 0, 0, 0, 5, 4)
    assert rc == 1 and b"device 1" in L.mi355cv_lastError()
    # the Python mirror binds the thread to the image's device instead
    assert torch.equal(cv.GaussianBlur(a, (5, 5), 0), d.zero_())


def test_two_device_ordinals_in_one_process_on_a_one_gpu_box():
    """drive >= 2 device ordinals from host threads in ONE process where only one GPU exists.  The HIP runtime is asked to expose the
    GPU twice (HIP_VISIBLE_DEVICES=0,0 in a child process); if it does, the per-thread-per-device contexts are exercised on ordinals 0 and 1 through the
    raw C ABI (own streams, own scratch pools, results equal to the restatement); if the runtime de-duplicates the list the child says so and the test
    is skipped -- the real multi-ordinal run is test_device_binding_per_thread on the driver's 8-GPU node."""
    import subprocess
    import sys
    code = r'''
import ctypes, sys, threading
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import orc
from opencv_amd import _lib
L = _lib.lib
n = L.mi355cv_deviceCount()
if n < 2:
    print("ONE_ORDINAL"); sys.exit(0)
rng = np.random.default_rng(5)
frames = rng.integers(0, 256, (4, 270, 480), dtype=np.uint8)
want = [orc.orc_gaussianBlurBinomialU8(f, 5, 4) for f in frames]
errors = []
def worker(d):
    try:
        assert L.mi355cv_setDevice(d) == 0 and L.mi355cv_getDevice() == d
        for f in range(d, 4, 2):
            p = L.mi355cv_deviceAlloc(2 * frames[f].size); assert p
            assert L.mi355cv_upload(ctypes.c_void_p(p), frames[f].ctypes.data, frames[f].size) == 0
            rc = L.mi355cv_gaussianBlurBinomial(ctypes.c_void_p(p), 480, ctypes.c_void_p(p + frames[f].size), 480, 480, 270, 0, 1, 0, 0, 0, 0, 5, 4)
            back = np.empty_like(frames[f])
            assert rc == 0 and L.mi355cv_download(back.ctypes.data, ctypes.c_void_p(p + frames[f].size), back.size) == 0
            assert np.array_equal(back, want[f]), f
            L.mi355cv_deviceFree(ctypes.c_void_p(p))
    except Exception as e:
        errors.append((d, repr(e)))
ts = [threading.Thread(target=worker, args=(d,)) for d in (0, 1)]
[t.start() for t in ts]; [t.join() for t in ts]
print("ERRORS" if errors else "TWO_ORDINALS_OK", errors)
'''
    env = dict(os.environ); env["HIP_VISIBLE_DEVICES"] = "0,0"; env.pop("ROCR_VISIBLE_DEVICES", None)
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    if "ONE_ORDINAL" in p.stdout or p.returncode != 0 and "TWO_ORDINALS_OK" not in p.stdout and "ERRORS" not in p.stdout:
        pytest.skip("the HIP runtime does not expose one GPU under two ordinals: " + (p.stdout + p.stderr)[-200:])
    assert "TWO_ORDINALS_OK" in p.stdout, (p.stdout[-500:], p.stderr[-500:])


def test_roctx_ranges_under_trace_switch():
    """MI355CV_TRACE=1 (SURVEY section 5 tracing row): every entry point runs inside a roctx range -- the marker library is found with dlopen, the
    state query says so, and the hooks keep working; without the switch nothing is loaded"""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, torch, sys
        sys.path.insert(0, %r)
        import opencv_amd as cv
        from opencv_amd import _lib
        img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (270, 480), dtype=np.uint8)).cuda()
        out = cv.GaussianBlur(img, (5, 5), 0)
        torch.cuda.synchronize()
        print("STATE", _lib.lib.mi355cv_traceState(), int(out.sum()) > 0)
    """ % ROOT)
    for env_val, want in (("1", "STATE 1 True"), (None, "STATE 0 True")):
        env = dict(os.environ); env.pop("MI355CV_TRACE", None)
        if env_val: env["MI355CV_TRACE"] = env_val
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and want in p.stdout, (p.stdout[-500:], p.stderr[-1500:])
