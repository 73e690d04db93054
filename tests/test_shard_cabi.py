"""The C-ABI multi-device batch runner (csrc/shard.hip): mi355cv_shardRange is the partition every layer uses (equal to
opencv_amd.shard.frame_range and to what bench.py --gpus N gives each rank), mi355cv_runSharded runs one host thread per device slot and reports the first
failing slot.  CPU: partition, threading, error propagation (bind = 0, no device touched).  GPU: the same runner driving a sharded GaussianBlur batch through
device slots (0, 0) -- two host threads, own streams and scratch pools, one GPU -- and mi355cv_replicate of a matchTemplate template."""
import ctypes
import os
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from opencv_amd import _lib, shard

L = _lib.lib


def rng_of(n, g, s):
    a, c = ctypes.c_int(), ctypes.c_int()
    L.mi355cv_shardRange(n, g, s, ctypes.byref(a), ctypes.byref(c))
    return a.value, c.value


def test_partition_is_the_python_layers_partition():
    for n in (0, 1, 7, 8, 9, 255, 256, 640, 9216):
        for g in (1, 2, 3, 4, 8):
            got = [rng_of(n, g, s) for s in range(g)]
            assert sum(c for _, c in got) == n and all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(g - 1))
            for s in range(g):
                lo, hi = shard.frame_range(n, s, g)
                assert got[s] == (lo, hi - lo), (n, g, s)
    assert rng_of(10, 4, 9) == rng_of(10, 4, 3) and rng_of(10, 0, 0) == (0, 10)          # out-of-range slots / device counts are clamped


def test_run_sharded_threads_and_error_codes():
    seen, tids = [], set()
    lock = threading.Lock()

    def body(user, slot, device, first, count):
        with lock:
            seen.append((slot, device, first, count)); tids.add(threading.get_ident())
        return 0
    fn = _lib.SHARD_FN(body)
    devs = (ctypes.c_int * 4)(5, 6, 6, 7)
    assert L.mi355cv_runSharded(4, devs, 10, fn, None, 0) == 0
    assert sorted(seen) == [(0, 5, 0, 3), (1, 6, 3, 3), (2, 6, 6, 2), (3, 7, 8, 2)] and len(tids) == 4
    seen.clear()
    assert L.mi355cv_runSharded(8, None, 3, fn, None, 0) == 0                               # more devices than frames: empty slots are skipped
    assert sorted(s[3] for s in seen) == [1, 1, 1] and sum(1 for s in seen) == 3

    def failing(user, slot, device, first, count):
        return -7 if slot == 2 else 0
    rc = L.mi355cv_runSharded(4, None, 100, _lib.SHARD_FN(failing), None, 0)
    assert rc == -7 and b"slot 2" in L.mi355cv_lastError()

    def raising(user, slot, device, first, count):
        raise RuntimeError("boom")                                                           # ctypes turns it into a 0 return + a printed traceback: must not crash
    L.mi355cv_runSharded(2, None, 4, _lib.SHARD_FN(raising), None, 0)
    assert L.mi355cv_runSharded(0, None, 4, fn, None, 0) != 0 and L.mi355cv_runSharded(2, None, 4, _lib.SHARD_FN(), None, 0) != 0


@pytest.mark.gpu
def test_sharded_batch_through_the_c_abi_on_duplicated_ordinals(orc):
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (10, 270, 480), dtype=np.uint8)
    want = np.stack([orc.orc_gaussianBlurBinomialU8(f, 5, 4) for f in frames])
    src = torch.from_numpy(frames).cuda(); dst = torch.zeros_like(src)
    fsz = 270 * 480
    errors, slots = [], []

    def body(user, slot, device, first, count):
        try:
            assert L.mi355cv_getDevice() == device == 0
            slots.append((slot, first, count, threading.get_ident()))
            return L.mi355cv_gaussianBlurBinomialBatch(ctypes.c_void_p(src.data_ptr() + first * fsz), 480, fsz, ctypes.c_void_p(dst.data_ptr() + first * fsz), 480, fsz,
                                                       count, 480, 270, 0, 1, 5, 4)
        except Exception as e:                       # noqa: BLE001
            errors.append(repr(e)); return -1
    devs = (ctypes.c_int * 3)(0, 0, 0)
    n0 = _lib.call_count("gaussianBlurBinomialBatch")
    assert L.mi355cv_runSharded(3, devs, 10, _lib.SHARD_FN(body), None, 1) == 0, (L.mi355cv_lastError(), errors)
    assert not errors and len({s[3] for s in slots}) == 3 and sorted(s[:3] for s in slots) == [(0, 0, 4), (1, 4, 3), (2, 7, 3)]
    assert _lib.call_count("gaussianBlurBinomialBatch") == n0 + 3
    assert np.array_equal(dst.cpu().numpy(), want)
    assert L.mi355cv_runSharded(2, (ctypes.c_int * 2)(0, 99), 4, _lib.SHARD_FN(body), None, 1) != 0 and b"slot 1" in L.mi355cv_lastError()     # no device 99
    # a template replicated to every device slot (here the same GPU twice: two independent allocations)
    tpl = rng.integers(0, 256, (128, 128), dtype=np.uint8)
    out = (ctypes.c_void_p * 2)()
    assert L.mi355cv_replicate(tpl.ctypes.data, tpl.size, 2, (ctypes.c_int * 2)(0, 0), out) == 0 and out[0] and out[1] and out[0] != out[1]
    for p in out:
        back = np.empty_like(tpl)
        assert L.mi355cv_download(back.ctypes.data, ctypes.c_void_p(p), back.size) == 0 and np.array_equal(back, tpl)
        L.mi355cv_deviceFree(ctypes.c_void_p(p))
    dsrc = torch.from_numpy(tpl).cuda()                                                                                                        # device-resident source
    assert L.mi355cv_replicate(ctypes.c_void_p(dsrc.data_ptr()), tpl.size, 1, None, out) == 0
    back = np.empty_like(tpl)
    assert L.mi355cv_download(back.ctypes.data, ctypes.c_void_p(out[0]), back.size) == 0 and np.array_equal(back, tpl)
    L.mi355cv_deviceFree(ctypes.c_void_p(out[0]))


@pytest.mark.gpu
def test_replicate_over_rccl():
    """MI355CV_REPLICATE=rccl (north_star: "RCCL broadcast of shared filter weights over xGMI"): mi355cv_replicate uploads once and broadcasts with
    ncclBroadcast on communicators of its own (librccl through dlopen).  On the one-GPU test box the communicator has one rank -- the code path (library, communicator, group
    call, streams) runs, the data is checked; with two GPUs the broadcast crosses xGMI; a list that names a device twice cannot form a communicator and takes the copies.
    A process of its own: the switch is read once."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import ctypes, numpy as np, torch, sys
        sys.path.insert(0, %r)
        from opencv_amd import _lib
        L = _lib.lib
        n = min(torch.cuda.device_count(), 2)
        tpl = np.random.default_rng(5).integers(0, 256, (128, 128), dtype=np.uint8)
        out = (ctypes.c_void_p * 2)()
        devs = (ctypes.c_int * 2)(0, 1)
        rc = L.mi355cv_replicate(tpl.ctypes.data, tpl.size, n, devs, out)
        assert rc == 0, L.mi355cv_lastError()
        print("MODE", L.mi355cv_replicateMode(), "RANKS", n)
        for i in range(n):
            assert L.mi355cv_setDevice(i) == 0
            back = np.empty_like(tpl)
            assert L.mi355cv_download(back.ctypes.data, ctypes.c_void_p(out[i]), back.size) == 0 and np.array_equal(back, tpl), i
            L.mi355cv_deviceFree(ctypes.c_void_p(out[i]))
        L.mi355cv_setDevice(-1)
        rc = L.mi355cv_replicate(tpl.ctypes.data, tpl.size, 2, (ctypes.c_int * 2)(0, 0), out)          # the same device twice: copies
        assert rc == 0 and L.mi355cv_replicateMode() == 0
        print("DUP ok")
    """ % ROOT)
    env = dict(os.environ); env["MI355CV_REPLICATE"] = "rccl"
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2500:])
    assert "MODE 1" in p.stdout and "DUP ok" in p.stdout, p.stdout
