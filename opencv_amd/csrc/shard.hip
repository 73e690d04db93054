// shard.hip -- SURVEY.md §8(e) for C / C++ hosts: batched frames shard embarrassingly across the GPUs of one node.  A batch of B independent frames is split
// by index into G contiguous blocks whose sizes differ by at most one (mi355cv_shardRange: the same partition opencv_amd/shard.py frame_range and bench.py --gpus N use; for B a multiple of G it is SURVEY §8e's [g*B/G, (g+1)*B/G)), one host thread per
// device binds itself with mi355cv_setDevice and runs the caller's per-shard function there -- which calls the ordinary mi355cv_* entry points (batch or
// per-frame) on that shard's frames.  No data-path collective exists on this path (no cv:: function here has cross-frame dependencies); the only
// thing replicated is parameters: host-side arguments (filter taps, warp matrices) are re-uploaded per call by every device's own hooks, and device-resident
// parameter images (a matchTemplate template) are copied to every device by mi355cv_replicate -- by default plain hipMemcpy (peer-to-peer over xGMI when the source
// lives on another GPU, PCIe when it is host memory); with MI355CV_REPLICATE=rccl one upload to the first device and an RCCL ncclBroadcast from there to the others over
// xGMI (north_star: "RCCL broadcast of shared filter weights over xGMI"; librccl.so is resolved with dlopen, the library itself links HIP only).  The Python layer
// broadcasts through torch.distributed (backend nccl = RCCL), opencv_amd/shard.py.
#include "rt.h"
#include <dlfcn.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace mi355;

namespace {

// ---- RCCL broadcast of a parameter image (MI355CV_REPLICATE=rccl).  One process, one communicator per device of the list (ncclCommInitAll; kept for the list's lifetime:
// setting one up costs tens of milliseconds), one group call: every slot posts ncclBroadcast(root = slot 0) on a stream of its own device, the data travels GPU to GPU over xGMI.
typedef void* NcclComm;
struct Rccl {
    int (*commInitAll)(NcclComm*, int, const int*) = nullptr;
    int (*broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*groupStart)() = nullptr;
    int (*groupEnd)() = nullptr;
    const char* (*errorString)(int) = nullptr;
    bool ok = false;
};
Rccl& rccl()
{
    static Rccl r = [] {
        Rccl t;
        void* h = nullptr;
        for (const char* lib : {"librccl.so.1", "librccl.so"}) { h = dlopen(lib, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) return t;
        t.commInitAll = (int (*)(NcclComm*, int, const int*))dlsym(h, "ncclCommInitAll");
        t.broadcast = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))dlsym(h, "ncclBroadcast");
        t.groupStart = (int (*)())dlsym(h, "ncclGroupStart");
        t.groupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
        t.errorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        t.ok = t.commInitAll && t.broadcast && t.groupStart && t.groupEnd;
        return t;
    }();
    return r;
}
std::mutex g_rcclMu;
std::map<std::vector<int>, std::vector<NcclComm>> g_rcclComms;
std::atomic<int> g_lastReplicateMode{0};

// 0: done over RCCL; > 0: RCCL not usable here (the caller copies instead); < 0: failed after allocating (everything freed, error recorded)
int replicateRccl(const void* src, size_t bytes, int ndev, const int* devices, void** out)
{
    Rccl& r = rccl();
    if (!r.ok) return 1;
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) devs[i] = devices ? devices[i] : i;
    { std::vector<int> u = devs; std::sort(u.begin(), u.end()); if (std::adjacent_find(u.begin(), u.end()) != u.end()) return 1; }      // a communicator needs distinct devices
    std::lock_guard<std::mutex> lk(g_rcclMu);                       // one broadcast at a time per process: communicators are shared
    std::vector<NcclComm>* comms;
    auto it = g_rcclComms.find(devs);
    if (it == g_rcclComms.end()) {
        std::vector<NcclComm> c(ndev, nullptr);
        const int e = r.commInitAll(c.data(), ndev, devs.data());
        if (e != 0) { setError(MI355CV_NOT_IMPLEMENTED, "mi355cv_replicate: ncclCommInitAll failed: %s", r.errorString ? r.errorString(e) : "?"); return 1; }
        comms = &g_rcclComms.emplace(devs, std::move(c)).first->second;
    } else comms = &it->second;
    std::vector<hipStream_t> st(ndev, nullptr);
    for (int i = 0; i < ndev; i++) out[i] = nullptr;               // the failure path frees every non-null slot: none may still hold what the caller's array contained
    int rc = 0, made = 0;
    for (; made < ndev && rc == 0; made++) {
        out[made] = nullptr;
        if (mi355cv_setDevice(devs[made]) != 0) { rc = -1; break; }
        Stager stg;
        if (!ensureDevice() || hipMalloc(&out[made], bytes) != hipSuccess || hipStreamCreateWithFlags(&st[made], hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError(); rc = setError(MI355CV_ERROR_UNKNOWN, "mi355cv_replicate: allocation on device %d failed", devs[made]); made++; break;
        }
        if (made == 0 && hipMemcpy(out[0], src, bytes, hipMemcpyDefault) != hipSuccess)                      // the one upload: host -> first device (or a peer copy)
            rc = setError(MI355CV_ERROR_UNKNOWN, "mi355cv_replicate: copy to device %d failed: %s", devs[0], hipGetErrorString(hipGetLastError()));
    }
    if (rc == 0) {
        int e = r.groupStart();
        for (int i = 0; i < ndev && e == 0; i++) e = r.broadcast(out[0], out[i], bytes, 0 /* ncclChar */, 0 /* root: slot 0 */, (*comms)[i], st[i]);
        const int e2 = r.groupEnd();
        if (e == 0) e = e2;
        for (int i = 0; i < ndev; i++) if (st[i]) { (void)mi355cv_setDevice(devs[i]); Stager stg; if (ensureDevice() && hipStreamSynchronize(st[i]) != hipSuccess && e == 0) e = -1; }
        if (e != 0) rc = setError(MI355CV_ERROR_UNKNOWN, "mi355cv_replicate: ncclBroadcast failed: %s", e > 0 && r.errorString ? r.errorString(e) : hipGetErrorString(hipGetLastError()));
    }
    for (int i = 0; i < ndev; i++) if (st[i]) (void)hipStreamDestroy(st[i]);
    if (rc != 0) for (int i = 0; i < ndev; i++) if (out[i]) { (void)hipFree(out[i]); out[i] = nullptr; }
    return rc == 0 ? 0 : -1;
}

} // namespace

extern "C" {

// how the last mi355cv_replicate of this process moved its data: 0 = hipMemcpy per device, 1 = one upload + RCCL ncclBroadcast (MI355CV_REPLICATE=rccl)
MI355CV_API int mi355cv_replicateMode(void) { return g_lastReplicateMode.load(); }

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
    // the device list is read on the CALLING thread first: a process whose very first HIP call comes from one of the worker threads below found "no ROCm-capable device"
    // on the MI355X box (tests/test_shard_cabi.py run on its own, round 5); from the calling thread the runtime initialises as everywhere else
    if (bind) (void)mi355cv_deviceCount();
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
    {
        // the transport: RCCL's ncclBroadcast over xGMI whenever more than one device is listed and librccl.so loads (north_star: "RCCL broadcast of shared filter
        // weights"; the default since round 6), one hipMemcpy per device otherwise -- MI355CV_REPLICATE=copy forces the copies, =rccl also tries RCCL for a single device
        static const int mode = [] { const char* v = getenv("MI355CV_REPLICATE"); return !v ? 0 : !strcmp(v, "rccl") ? 1 : !strcmp(v, "copy") ? -1 : 0; }();
        const bool wantRccl = mode > 0 || (mode == 0 && ndev > 1);
        g_lastReplicateMode = 0;
        if (wantRccl) {
            const int r = replicateRccl(src, bytes, ndev, devices, out);
            (void)mi355cv_setDevice(before);
            if (r == 0) { g_lastReplicateMode = 1; return 0; }
            if (r < 0) return -1;                                   // allocation / copy failure: reported, nothing left allocated
            // r > 0: RCCL not available for this device list (library missing, a device listed twice, communicator setup failed): the copies below
        }
    }
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
