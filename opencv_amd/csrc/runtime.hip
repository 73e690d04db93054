// runtime.hip -- process/thread runtime of libmi355cv.so (see rt.h).
#include "rt.h"
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace mi355 {

static std::atomic<int> g_device{-1};
static std::atomic<int> g_deviceState{0};  // 0 unknown, 1 ok, -1 unusable
static std::mutex g_mu;
static std::map<std::string, long long> g_counts;
static thread_local char t_err[512] = "";

struct Buf { void* p; size_t cap; bool busy; };

struct ThreadCtx {
    hipStream_t own = nullptr;
    hipStream_t user = nullptr;
    hipStream_t aux = nullptr;
    hipEvent_t events[64] = {};
    bool useUser = false;
    bool async = false;
    int stagerDepth = 0;              // hooks may call other hooks (adaptiveThreshold -> boxFilter): only the outermost Stager recycles the pool
    std::vector<Buf> pool;
    ~ThreadCtx() {
        // process teardown order vs. the HIP runtime is undefined: leak on purpose
    }
};

ThreadCtx& tctx() { static thread_local ThreadCtx c; return c; }

static bool envFlag(const char* name) { const char* v = getenv(name); return v && *v && strcmp(v, "0") != 0; }

bool disabled() { static bool d = envFlag("MI355CV_DISABLE"); return d; }
size_t minPixels(int cost)
{
    static const size_t v = getenv("MI355CV_MIN_PIXELS") ? strtoull(getenv("MI355CV_MIN_PIXELS"), nullptr, 10) : 0;
    static const bool autoPolicy = getenv("MI355CV_HOST_POLICY") && !strcmp(getenv("MI355CV_HOST_POLICY"), "auto");
    if (!autoPolicy) return v;
    return cost == HOST_HEAVY ? std::max<size_t>(v, 64 * 64) : (size_t)-1;
}

int setError(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(t_err, sizeof t_err, fmt, ap); va_end(ap);
    if (getenv("MI355CV_LOG")) fprintf(stderr, "[mi355cv] %s\n", t_err);
    return code;
}

// MI355CV_PRINT_COUNTS=1: at process exit, one line per entry point with the number of calls the GPU served -- how a host program that
// cannot call mi355cv_callCount (the reference's own test binary, tests/test_reference_suite.py) shows that its cv:: calls ran here
static void printCounts()
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (const auto& kv : g_counts) fprintf(stderr, "mi355cv: %s %lld\n", kv.first.c_str(), (long long)kv.second);
}

void bump(const char* entry)
{
    std::lock_guard<std::mutex> lk(g_mu);
    static const bool hooked = [] { const char* e = getenv("MI355CV_PRINT_COUNTS"); if (e && atoi(e)) atexit(printCounts); return true; }();
    (void)hooked;
    g_counts[entry]++;
}

bool ensureDevice()
{
    int st = g_deviceState.load();
    if (st == 0) {
        std::lock_guard<std::mutex> lk(g_mu);
        st = g_deviceState.load();
        if (st == 0) {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); g_deviceState = -1; setError(MI355CV_NOT_IMPLEMENTED, "no HIP device"); return false; }
            int dev = g_device.load();
            if (dev < 0) {
                if (hipGetDevice(&dev) != hipSuccess) dev = 0;   // honour a device the host already selected (e.g. torch.cuda.set_device)
                const char* e = getenv("MI355CV_DEVICE");
                if (e) dev = atoi(e);
            }
            if (dev >= n) dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { g_deviceState = -1; return false; }
            if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !envFlag("MI355CV_ANY_ARCH")) {
                g_deviceState = -1;
                setError(MI355CV_NOT_IMPLEMENTED, "device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
                return false;
            }
            g_device = dev;
            g_deviceState = st = 1;
        }
    }
    if (st != 1) return false;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != g_device.load()) {
        if (hipSetDevice(g_device.load()) != hipSuccess) return false;
    }
    return true;
}

hipStream_t stream()
{
    ThreadCtx& c = tctx();
    if (c.useUser) return c.user;
    if (!c.own) {
        if (hipStreamCreateWithFlags(&c.own, hipStreamNonBlocking) != hipSuccess) c.own = nullptr;
    }
    return c.own;
}

hipStream_t auxStream()
{
    ThreadCtx& c = tctx();
    if (!c.aux && hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking) != hipSuccess) c.aux = nullptr;
    return c.aux;
}

hipEvent_t pooledEvent(int i)
{
    ThreadCtx& c = tctx();
    if (i < 0 || i >= 64) return nullptr;
    if (!c.events[i] && hipEventCreateWithFlags(&c.events[i], hipEventDisableTiming) != hipSuccess) c.events[i] = nullptr;
    return c.events[i];
}

bool asyncMode() { return tctx().async; }

bool isDevicePtr(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }   // unregistered host memory
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// ------------------------------------------------------------------ Stager

Stager::Stager()
{
    // an error left behind by an earlier call on this thread (a failed copy, somebody else's HIP code) must not be charged to this hook
    if (tctx().stagerDepth++ == 0) (void)hipGetLastError();
}
Stager::~Stager()
{
    // buffers handed out during this (outermost) hook become reusable; safe in stream order
    if (--tctx().stagerDepth == 0)
        for (auto& b : tctx().pool) b.busy = false;
}

void* Stager::bump_(size_t bytes)
{
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~size_t(255);
    auto& pool = tctx().pool;
    int best = -1;
    for (int i = 0; i < (int)pool.size(); i++)
        if (!pool[i].busy && pool[i].cap >= bytes && (best < 0 || pool[i].cap < pool[best].cap)) best = i;
    if (best >= 0 && pool[best].cap <= 4 * bytes + (1 << 20)) { pool[best].busy = true; return pool[best].p; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); failed_ = true; setError(MI355CV_NOT_IMPLEMENTED, "hipMalloc(%zu) failed", bytes); return nullptr; }
    pool.push_back({p, bytes, true});
    return p;
}

const uchar* Stager::in(const uchar* p, size_t step, size_t rowBytes, int rows, size_t* dstep)
{
    if (!p || rows <= 0 || rowBytes == 0) { failed_ = true; return nullptr; }          // e.g. the empty dst of THRESH_DRYRUN: the hook declines
    if (isDevicePtr(p)) { *dstep = step; return p; }
    anyHost_ = true;
    size_t ds = (rowBytes + 255) & ~size_t(255);
    uchar* d = (uchar*)bump_(ds * (size_t)rows);
    if (!d) return nullptr;
    if (hipMemcpy2DAsync(d, ds, p, step, rowBytes, rows, hipMemcpyHostToDevice, stream()) != hipSuccess) {
        failed_ = true; setError(MI355CV_NOT_IMPLEMENTED, "H2D staging failed: %s", hipGetErrorString(hipGetLastError())); return nullptr;
    }
    *dstep = ds;
    return d;
}

uchar* Stager::out(uchar* p, size_t step, size_t rowBytes, int rows, size_t* dstep)
{
    if (!p || rows <= 0 || rowBytes == 0) { failed_ = true; return nullptr; }
    if (isDevicePtr(p)) { *dstep = step; return p; }
    anyHost_ = true;
    size_t ds = (rowBytes + 255) & ~size_t(255);
    uchar* d = (uchar*)bump_(ds * (size_t)rows);
    if (!d) return nullptr;
    outs_.push_back({p, step, d, ds, rowBytes, rows});
    *dstep = ds;
    return d;
}

void* Stager::param(const void* host, size_t bytes)
{
    void* d = bump_(bytes);
    if (!d) return nullptr;
    if (hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, stream()) != hipSuccess) { failed_ = true; return nullptr; }
    return d;
}

void* Stager::scratch(size_t bytes) { return bump_(bytes); }

int Stager::finish(const char* entry)
{
    hipStream_t s = stream();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return setError(MI355CV_ERROR_UNKNOWN, "%s: launch failed: %s", entry, hipGetErrorString(e));
    for (auto& o : outs_) {
        e = hipMemcpy2DAsync(o.host, o.hstep, o.dev, o.dstep, o.rowBytes, o.rows, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return setError(MI355CV_ERROR_UNKNOWN, "%s: D2H failed: %s", entry, hipGetErrorString(e));
    }
    // a hook called from inside another hook (device pointers, same stream) leaves the synchronisation to the outermost one
    if (anyHost_ || (!asyncMode() && tctx().stagerDepth == 1)) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return setError(MI355CV_ERROR_UNKNOWN, "%s: execution failed: %s", entry, hipGetErrorString(e));
    }
    bump(entry);
    return MI355CV_OK;
}

} // namespace mi355

// ------------------------------------------------------------------ exported runtime API
using namespace mi355;

extern "C" {

MI355CV_API int mi355cv_init(int device)
{
    if (device >= 0 && g_deviceState.load() == 0) g_device = device;
    return ensureDevice() ? 0 : -1;
}

MI355CV_API const char* mi355cv_version(void) { return "mi355cv 0.1 (gfx950; HAL mirror of OpenCV 4.12 imgproc hot path)"; }
MI355CV_API const char* mi355cv_lastError(void) { return t_err; }

MI355CV_API int mi355cv_setStream(void* s)
{
    ThreadCtx& c = tctx();
    c.user = (hipStream_t)s; c.useUser = true;    // NULL is HIP's null (legacy default) stream
    return 0;
}

MI355CV_API int mi355cv_resetStream(void) { tctx().useUser = false; return 0; }

MI355CV_API int mi355cv_setAsync(int enable) { tctx().async = enable != 0; return 0; }

MI355CV_API int mi355cv_synchronize(void)
{
    if (!ensureDevice()) return -1;
    return hipStreamSynchronize(stream()) == hipSuccess ? 0 : -1;
}

MI355CV_API long long mi355cv_callCount(const char* entry)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_counts.find(entry ? entry : "");
    return it == g_counts.end() ? 0 : it->second;
}

MI355CV_API void* mi355cv_deviceAlloc(size_t bytes)
{
    if (!ensureDevice()) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
MI355CV_API int mi355cv_deviceFree(void* p) { return hipFree(p) == hipSuccess ? 0 : -1; }

// frame ingest / egress: host memory the DMA engines reach directly.  kind 0 = page-locked (hipHostMalloc): staging such
// a frame through HBM is one DMA at PCIe rate instead of the driver's pageable bounce; kind 1 = managed (hipMallocManaged): the hooks run
// on it in place (isDevicePtr), pages migrate on first touch from either side.
MI355CV_API void* mi355cv_hostAlloc(size_t bytes, int kind)
{
    if (bytes == 0 || (kind != 0 && kind != 1) || !ensureDevice()) return nullptr;
    void* p = nullptr;
    const hipError_t e = kind == 0 ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMallocManaged(&p, bytes, hipMemAttachGlobal);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
MI355CV_API int mi355cv_hostFree(void* p, int kind)
{
    if (!p) return 0;
    return (kind == 0 ? hipHostFree(p) : hipFree(p)) == hipSuccess ? 0 : -1;
}
MI355CV_API int mi355cv_upload(void* d, const void* h, size_t n) { return ensureDevice() && hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
MI355CV_API int mi355cv_download(void* h, const void* d, size_t n) { return ensureDevice() && hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }

} // extern "C"
