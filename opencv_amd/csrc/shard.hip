// shard.hip -- SURVEY.md §8(e) for C / C++ hosts: batched frames shard embarrassingly across the GPUs of one node.  A batch of B independent frames is split
// by index into G contiguous blocks whose sizes differ by at most one (mi355cv_shardRange: the same partition opencv_amd/shard.py frame_range and bench.py --gpus N use; for B a multiple of G it is SURVEY §8e's [g*B/G, (g+1)*B/G)), one host thread per
// device binds itself with mi355cv_setDevice and runs the caller's per-shard function there -- which calls the ordinary mi355cv_* entry points (batch or
// per-frame) on that shard's frames.  No data-path collective exists on this path (no cv:: function here has cross-frame dependencies); the only
// thing replicated is parameters: host-side arguments (filter taps, warp matrices) are re-uploaded per call by every device's own hooks, and device-resident
// parameter images (a matchTemplate template) are copied to every device by mi355cv_replicate -- plain hipMemcpy (peer-to-peer over xGMI when the source lives
// on another GPU, PCIe when it is host memory); RCCL is not used by the C ABI (the Python layer's torch.distributed broadcast is, opencv_amd/shard.py).
#include "rt.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace mi355;

extern "C" {

// frames [first, first + count) of device slot g (0 <= g < ndev): the partition every layer of this repository uses
MI355CV_API void mi355cv_shardRange(int nframes, int ndev, int g, int* first, int* count)
{
    if (ndev < 1) ndev = 1;
    if (nframes < 0) nframes = 0;
    g = std::min(std::max(g, 0), ndev - 1);
    const int base = nframes / ndev, rem = nframes % ndev;       // contiguous blocks whose sizes differ by at most one; the first `rem` slots take the longer ones
    if (first) *first = g * base + std::min(g, rem);
    if (count) *count = base + (g < rem ? 1 : 0);
}

// Runs fn(user, slot, device, first, count) once per device slot, each on its own host thread bound to devices[slot] (devices == NULL: ordinals 0 .. ndev-1;
// an ordinal may repeat -- two threads then share a GPU, each with its own streams and scratch pool).  Slots whose range is empty are not called.
// bind != 0 binds the thread with mi355cv_setDevice (fails the slot if the ordinal is not a usable gfx950 device); bind == 0 only partitions and threads
// (hosts that manage devices themselves, and the CPU-only tests).  Returns 0 when every slot returned 0; otherwise the first non-zero code in slot order,
// with mi355cv_lastError() of the CALLING thread naming the slot, its device and the failing thread's own error text.
MI355CV_API int mi355cv_runSharded(int ndev, const int* devices, int nframes, int (*fn)(void* user, int slot, int device, int first, int count), void* user, int bind)
{
    mi355::EntryGuard entry_(__func__);
    if (ndev < 1 || ndev > 64 || nframes < 0 || !fn) return setError(MI355CV_ERROR_UNKNOWN, "mi355cv_runSharded: ndev %d (1 .. 64), nframes %d, fn %p", ndev, nframes, (void*)fn);
    std::vector<int> rc(ndev, 0);
    std::vector<std::string> why(ndev);
    std::vector<std::thread> th;
    th.reserve(ndev);
    for (int g = 0; g < ndev; g++) {
        int first = 0, count = 0;
        mi355cv_shardRange(nframes, ndev, g, &first, &count);
        if (count <= 0) continue;
        const int dev = devices ? devices[g] : g;
        auto body = [&, g, dev, first, count] {
            try {
                if (bind && mi355cv_setDevice(dev) != 0) { rc[g] = MI355CV_ERROR_UNKNOWN; why[g] = mi355cv_lastError(); return; }
                rc[g] = fn(user, g, dev, first, count);
                if (rc[g] != 0) why[g] = mi355cv_lastError();
                if (bind) { if (rc[g] == 0 && mi355cv_synchronize() != 0) { rc[g] = MI355CV_ERROR_UNKNOWN; why[g] = mi355cv_lastError(); } (void)mi355cv_setDevice(-1); }
            } catch (...) { rc[g] = MI355CV_ERROR_UNKNOWN; why[g] = "exception in the shard function"; }      // nothing may cross the C boundary
        };
        // std::thread's constructor throws std::system_error when the process is out of threads (EAGAIN): the slot then runs here, on the calling thread, after
        // which the threads already started are joined as usual -- no exception leaves this extern "C" function with joinable threads behind it
        try { th.emplace_back(body); }
        catch (...) { body(); }
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < ndev; g++)
        if (rc[g] != 0) return setError(rc[g], "mi355cv_runSharded: slot %d (device %d) returned %d: %s", g, devices ? devices[g] : g, rc[g], why[g].c_str());
    return 0;
}

// A parameter image on every device of the list: out[slot] = a fresh allocation on devices[slot] holding `bytes` bytes copied from src (host memory, or device
// memory of any GPU: hipMemcpy with hipMemcpyDefault resolves the direction, peer copies travel over xGMI).  Free each with mi355cv_deviceFree on a thread bound
// to that device (or hipFree).  Returns 0, or -1 after freeing whatever it had allocated.
MI355CV_API int mi355cv_replicate(const void* src, size_t bytes, int ndev, const int* devices, void** out)
{
    mi355::EntryGuard entry_(__func__);
    if (!src || !out || ndev < 1 || ndev > 64 || !bytes) return setError(MI355CV_ERROR_UNKNOWN, "mi355cv_replicate: bad arguments");
    const int before = threadDeviceBinding();
    int done = 0, rc = 0;
    for (; done < ndev; done++) {
        out[done] = nullptr;
        if (mi355cv_setDevice(devices ? devices[done] : done) != 0) { rc = -1; break; }
        void* p = mi355cv_deviceAlloc(bytes);
        if (!p) { rc = setError(MI355CV_ERROR_UNKNOWN, "mi355cv_replicate: hipMalloc(%zu) on device %d failed", bytes, devices ? devices[done] : done); break; }
        out[done] = p;
        bool ok;
        { Stager stg; ok = ensureDevice() && hipMemcpy(p, src, bytes, hipMemcpyDefault) == hipSuccess; }      // ~Stager puts the host program's own current device back
        if (!ok) {
            rc = setError(MI355CV_ERROR_UNKNOWN, "mi355cv_replicate: copy to device %d failed: %s", devices ? devices[done] : done, hipGetErrorString(hipGetLastError()));
            done++; break;
        }
    }
    if (rc != 0) for (int i = 0; i < done; i++) if (out[i]) { (void)hipFree(out[i]); out[i] = nullptr; }
    (void)mi355cv_setDevice(before);
    return rc == 0 ? 0 : -1;
}

} // extern "C"
