// runtime.hip -- process/thread runtime of libmi355cv.so (see rt.h).
#include "rt.h"
#include <dlfcn.h>
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

// ---- devices.  A hook runs on the calling thread's device: the one chosen with mi355cv_setDevice on that thread, else the process default
// (MI355CV_DEVICE, mi355cv_init(dev), or whatever device the host program had current when the library was first used).  Every thread keeps
// one context (streams, events, scratch pool) PER DEVICE, so one C++ process can drive the 8 GPUs of a node from 8 host threads (SURVEY §8e:
// "one host thread + >= 2 HIP streams per device"), and the host program's own current device is put back when the outermost hook returns.
constexpr int MAX_DEV = 16;
static std::atomic<int> g_defaultDev{-1};
static std::atomic<int> g_nDev{-2};                 // -2 unknown, <= 0 none
static std::atomic<int> g_devState[MAX_DEV];        // 0 unknown, 1 usable (gfx950), -1 not
static std::mutex g_mu;
static std::map<std::string, long long> g_counts;
static std::atomic<long long> g_stagedBytes{0};   // image bytes the hooks moved over PCIe (host images staged in + results staged back)
void noteStagedBytes(long long n) { g_stagedBytes += n; }
static thread_local char t_err[512] = "";
static thread_local unsigned t_serial = 0, t_errSerial = ~0u;  // hook invocations on this thread (bumped when the outermost Stager opens and closes); the one that recorded t_err
static thread_local bool t_notDrained = false;            // the current outermost hook returned (or will return) without waiting for its stream: its scratch stays tied to that stream
static thread_local bool t_sawManaged = false;           // a managed (host-visible) image was classified during the current outermost hook
static thread_local bool t_errFresh = false;             // set by setError, taken by mi355cv_noteDecline: a reason is attributed to one declined call
static thread_local int t_dev = -1;                 // mi355cv_setDevice; -1 = process default
static thread_local int t_active = 0;               // device of the hook that is running = index of the per-device thread context
static thread_local int t_restore = -1;             // the caller's current device, to be put back by the outermost hook
static thread_local int t_depth = 0;                // hooks may call other hooks (adaptiveThreshold -> boxFilter): only the outermost Stager recycles / restores

struct Buf { void* p; size_t cap; bool busy; hipStream_t last; };

struct ThreadCtx {
    hipStream_t own = nullptr;
    hipStream_t user = nullptr;
    hipStream_t aux = nullptr;
    hipEvent_t events[64] = {};
    bool useUser = false;
    bool async = false;
    std::vector<Buf> pool;
    std::vector<Buf> pinned;       // page-locked host buffers (Stager::pinned)
    ~ThreadCtx() {
        // process teardown order vs. the HIP runtime is undefined: leak on purpose
    }
};

ThreadCtx& tctx() { static thread_local ThreadCtx c[MAX_DEV]; return c[t_active]; }
int activeDevice() { return t_active; }

static bool envFlag(const char* name) { const char* v = getenv(name); return v && *v && strcmp(v, "0") != 0; }

bool disabled() { static bool d = envFlag("MI355CV_DISABLE"); return d; }
bool hostImageTooSmall(const void* img, size_t pixels, size_t threshold)
{
    if (pixels >= threshold || isDevicePtr(img)) return false;
    setError(MI355CV_NOT_IMPLEMENTED, "host-resident image of %zu pixels, below the %zu-pixel policy threshold (two PCIe crossings cost more than the CPU path; "
             "MI355CV_MIN_PIXELS, or keep the image in device memory)", pixels, threshold);
    return true;
}

// -1: what MI355CV_HOST_POLICY says (default auto); 0 auto, 1 always -- mi355cv_setHostPolicy, for hosts above the C ABI that have no CPU path of their own to fall back to
static std::atomic<int> g_hostPolicy{-1};

size_t minPixels(int cost)
{
    static const size_t v = getenv("MI355CV_MIN_PIXELS") ? strtoull(getenv("MI355CV_MIN_PIXELS"), nullptr, 10) : 0;
    // "auto" is the default since round 5 (serving every host Mat by staging made the out-of-the-box drop-in slower than the reference's own CPU path on the
    // bandwidth-bound hooks -- 40.8 against 51.5 Gpix/s for the 8-bit Gaussian on 16 host threads); MI355CV_HOST_POLICY=always stages everything (the parity suites set it)
    static const bool autoPolicy = !(getenv("MI355CV_HOST_POLICY") && !strcmp(getenv("MI355CV_HOST_POLICY"), "always"));
    const int pol = g_hostPolicy.load(std::memory_order_relaxed);
    if (pol < 0 ? !autoPolicy : pol == 1) return v;
    return cost == HOST_HEAVY ? std::max<size_t>(v, 64 * 64) : (size_t)-1;
}

static thread_local char t_kernel[160] = "";
void noteKernel(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(t_kernel, sizeof t_kernel, fmt, ap); va_end(ap);
}

int setError(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(t_err, sizeof t_err, fmt, ap); va_end(ap);
    t_errFresh = true; t_errSerial = t_serial;
    if (getenv("MI355CV_LOG")) fprintf(stderr, "[mi355cv] %s\n", t_err);
    return code;
}

int declined(const char* fn, int line, const char* cond)
{
    if (t_errFresh && t_errSerial == t_serial) return MI355CV_NOT_IMPLEMENTED;          // the specific reason (setError during this call) stands
    if (disabled()) return setError(MI355CV_NOT_IMPLEMENTED, "%s: MI355CV_DISABLE is set", fn);
    return setError(MI355CV_NOT_IMPLEMENTED, "%s:%d: outside the GPU path because (%s)", fn, line, cond ? cond : "no branch of the dispatcher takes this argument combination");
}
void beginCall() { ++t_serial; }
static thread_local int t_entryDepth = 0;
// roctx, resolved lazily (MI355CV_TRACE=1 only): 0 = not tried, 1 = available, -1 = no library found (tracing stays off, silently)
typedef int (*RoctxPushFn)(const char*);
typedef int (*RoctxPopFn)();
static RoctxPushFn g_roctxPush = nullptr;
static RoctxPopFn g_roctxPop = nullptr;
static std::atomic<int> g_roctxState{0};
static bool roctxReady()
{
    static const bool want = envFlag("MI355CV_TRACE");
    if (!want) return false;
    int st = g_roctxState.load();
    if (st == 0) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_roctxState.load() == 0) {
            void* h = nullptr;
            for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) { h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
            if (h) { g_roctxPush = (RoctxPushFn)dlsym(h, "roctxRangePushA"); g_roctxPop = (RoctxPopFn)dlsym(h, "roctxRangePop"); }
            g_roctxState = (g_roctxPush && g_roctxPop) ? 1 : -1;
        }
        st = g_roctxState.load();
    }
    return st == 1;
}
EntryGuard::EntryGuard(const char* name)
{
    if (t_entryDepth++ == 0) ++t_serial;
    if (roctxReady()) { g_roctxPush(name ? name : "mi355cv"); traced_ = true; }
}
EntryGuard::~EntryGuard() { if (traced_) g_roctxPop(); --t_entryDepth; }
extern "C" MI355CV_API int mi355cv_traceState(void) { return roctxReady() ? 1 : g_roctxState.load(); }   // 1: ranges are being emitted; 0: MI355CV_TRACE not set; -1: no roctx library

// MI355CV_PRINT_COUNTS=1: at process exit, one line per entry point with the number of calls the GPU served -- how a host program that
// cannot call mi355cv_callCount (the reference's own test binary, tests/test_reference_suite.py) shows that its cv:: calls ran here
static std::map<std::string, std::pair<long long, std::string>> g_declines;      // hook -> (count, last reason)
static void printCounts()
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (const auto& kv : g_counts) fprintf(stderr, "mi355cv: %s %lld\n", kv.first.c_str(), (long long)kv.second);
    for (const auto& kv : g_declines) fprintf(stderr, "mi355cv: declined %s %lld (last: %s)\n", kv.first.c_str(), kv.second.first, kv.second.second.c_str());
}
// MI355CV_LEDGER=1: one line per served entry point and per declined hook on stderr, in call order -- a host program whose stdout names what it is
// doing (a gtest binary) thereby shows, call by call, which of its steps ran on the GPU and which on its own CPU path (tests/test_reference_suite.py)
static bool ledgerLog()
{
    static const bool on = [] { const char* e = getenv("MI355CV_LEDGER"); return e && atoi(e); }();
    return on;
}
static void hookPrintCounts()
{
    static const bool hooked = [] { const char* e = getenv("MI355CV_PRINT_COUNTS"); if (e && atoi(e)) atexit(printCounts); return true; }();
    (void)hooked;
}

void bump(const char* entry)
{
    std::lock_guard<std::mutex> lk(g_mu);
    hookPrintCounts();
    g_counts[entry]++;
    if (ledgerLog()) { fprintf(stderr, "[mi355cv] served %s\n", entry); fflush(stderr); }
}

static char g_nDevWhy[160] = "";                    // what hipGetDeviceCount said when it found no device (reported by mi355cv_setDevice)
static int deviceCount()
{
    int n = g_nDev.load();
    if (n == -2) {
        const hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            snprintf(g_nDevWhy, sizeof g_nDevWhy, "hipGetDeviceCount: %s, %d device(s)", hipGetErrorString(e), n);
            (void)hipGetLastError(); n = 0;
        }
        if (n > MAX_DEV) n = MAX_DEV;
        g_nDev = n;
    }
    return n;
}

// the device this thread's hooks run on (no HIP state is touched); -1 without a usable GPU
static int resolveDevice()
{
    const int n = deviceCount();
    if (n <= 0) { setError(MI355CV_NOT_IMPLEMENTED, "no HIP device"); return -1; }
    int d = t_dev;
    if (d < 0) {
        d = g_defaultDev.load();
        if (d < 0) {
            std::lock_guard<std::mutex> lk(g_mu);
            d = g_defaultDev.load();
            if (d < 0) {
                if (hipGetDevice(&d) != hipSuccess) d = 0;       // honour a device the host already selected (e.g. torch.cuda.set_device)
                const char* e = getenv("MI355CV_DEVICE");
                if (e) d = atoi(e);
                if (d < 0 || d >= n) d = 0;
                g_defaultDev = d;
            }
        }
    }
    if (d >= n) { setError(MI355CV_NOT_IMPLEMENTED, "device %d selected, %d visible", d, n); return -1; }
    int st = g_devState[d].load();
    if (st == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) { (void)hipGetLastError(); st = -1; }
        else if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !envFlag("MI355CV_ANY_ARCH")) {
            st = -1;
            setError(MI355CV_NOT_IMPLEMENTED, "device %d is %s, this library is built for gfx950 only", d, prop.gcnArchName);
        } else st = 1;
        g_devState[d] = st;
    }
    if (st != 1) return -1;
    t_active = d;
    return d;
}

bool ensureDevice()
{
    const int d = resolveDevice();
    if (d < 0) return false;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != d) {
        if (t_restore < 0 && cur >= 0) t_restore = cur;           // the host program's device: put back when the outermost hook returns
        if (hipSetDevice(d) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    return true;
}

static void restoreDevice()
{
    if (t_restore >= 0) { (void)hipSetDevice(t_restore); t_restore = -1; }
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

int ptrKind(const void* p)
{
    if (!p) return PTR_HOST;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return PTR_HOST; }   // unregistered host memory
    if (a.type == hipMemoryTypeManaged) { t_sawManaged = true; return PTR_DEVICE; }          // host-visible: results must be complete when the hook returns (Stager::finish)
    if (a.type == hipMemoryTypeDevice) return a.device == t_active ? PTR_DEVICE : PTR_FOREIGN;
    return PTR_HOST;
}

bool isDevicePtr(const void* p) { return ptrKind(p) == PTR_DEVICE; }

// ------------------------------------------------------------------ Stager

Stager::Stager()
{
    // an error left behind by an earlier call on this thread (a failed copy, somebody else's HIP code) must not be charged to this hook
    if (t_depth++ == 0) { (void)hipGetLastError(); beginCall(); t_sawManaged = false; t_notDrained = false; }
}
Stager::~Stager()
{
    // buffers handed out during this (outermost) hook become reusable: at once on the stream that used them (stream order), on any other
    // stream only after that one has drained (bump_ checks) -- two asynchronous calls under different streams never share scratch
    if (--t_depth == 0) {
        beginCall();                                         // a reason recorded during this call is not the next call's
        ThreadCtx& c = tctx();
        hipStream_t s = (c.async || t_notDrained) ? stream() : nullptr;
        for (auto& b : c.pool) if (b.busy) { b.busy = false; b.last = s; }
        for (auto& b : c.pinned) b.busy = false;          // the host has read them by now: every user synchronises before it looks
        restoreDevice();
    }
}

void* Stager::bump_(size_t bytes)
{
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~size_t(255);
    auto& pool = tctx().pool;
    hipStream_t cur = stream();
    int best = -1;
    for (int i = 0; i < (int)pool.size(); i++) {
        Buf& b = pool[i];
        if (b.busy || b.cap < bytes || (best >= 0 && b.cap >= pool[best].cap)) continue;
        if (b.last && b.last != cur) {                               // last used asynchronously on another stream
            if (hipStreamQuery(b.last) != hipSuccess) { (void)hipGetLastError(); continue; }
            b.last = nullptr;
        }
        best = i;
    }
    if (best >= 0 && pool[best].cap <= 4 * bytes + (1 << 20)) { pool[best].busy = true; return pool[best].p; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); failed_ = true; setError(MI355CV_NOT_IMPLEMENTED, "hipMalloc(%zu) failed", bytes); return nullptr; }
    pool.push_back({p, bytes, true, nullptr});
    return p;
}

// a pointer into another GPU's memory: the thread is bound to the wrong device for this image -- decline (the caller falls back / raises)
bool Stager::foreign_(const void* p)
{
    failed_ = true;
    hipPointerAttribute_t a;
    const int owner = hipPointerGetAttributes(&a, p) == hipSuccess ? a.device : -1;
    setError(MI355CV_NOT_IMPLEMENTED, "image lives on device %d, this thread runs on device %d (mi355cv_setDevice)", owner, t_active);
    return true;
}

const uchar* Stager::in(const uchar* p, size_t step, size_t rowBytes, int rows, size_t* dstep)
{
    if (!p || rows <= 0 || rowBytes == 0) { failed_ = true; return nullptr; }          // e.g. the empty dst of THRESH_DRYRUN: the hook declines
    const int kind = ptrKind(p);
    if (kind == PTR_DEVICE) { *dstep = step; return p; }
    if (kind == PTR_FOREIGN) { foreign_(p); return nullptr; }
    anyHost_ = true;
    size_t ds = (rowBytes + 255) & ~size_t(255);
    uchar* d = (uchar*)bump_(ds * (size_t)rows);
    if (!d) return nullptr;
    g_stagedBytes += (long long)(rowBytes * (size_t)rows);
    if (hipMemcpy2DAsync(d, ds, p, step, rowBytes, rows, hipMemcpyHostToDevice, stream()) != hipSuccess) {
        failed_ = true; setError(MI355CV_NOT_IMPLEMENTED, "H2D staging failed: %s", hipGetErrorString(hipGetLastError())); return nullptr;
    }
    *dstep = ds;
    return d;
}

uchar* Stager::out(uchar* p, size_t step, size_t rowBytes, int rows, size_t* dstep)
{
    if (!p || rows <= 0 || rowBytes == 0) { failed_ = true; return nullptr; }
    const int kind = ptrKind(p);
    if (kind == PTR_DEVICE) { *dstep = step; return p; }
    if (kind == PTR_FOREIGN) { foreign_(p); return nullptr; }
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

void* Stager::pinned(size_t bytes)
{
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 4095) & ~size_t(4095);
    auto& pool = tctx().pinned;
    int best = -1;
    for (int i = 0; i < (int)pool.size(); i++)
        if (!pool[i].busy && pool[i].cap >= bytes && (best < 0 || pool[i].cap < pool[best].cap)) best = i;
    if (best >= 0) { pool[best].busy = true; return pool[best].p; }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); failed_ = true; setError(MI355CV_NOT_IMPLEMENTED, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    pool.push_back({p, bytes, true, nullptr});
    return p;
}

int Stager::finish(const char* entry)
{
    hipStream_t s = stream();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return setError(MI355CV_ERROR_UNKNOWN, "%s: launch failed: %s", entry, hipGetErrorString(e));
    for (auto& o : outs_) {
        g_stagedBytes += (long long)(o.rowBytes * (size_t)o.rows);
        e = hipMemcpy2DAsync(o.host, o.hstep, o.dev, o.dstep, o.rowBytes, o.rows, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return setError(MI355CV_ERROR_UNKNOWN, "%s: D2H failed: %s", entry, hipGetErrorString(e));
    }
    // a hook called from inside another hook (device pointers, same stream) leaves the synchronisation to the outermost one
    // Device-resident images on a stream the CALLER bound (mi355cv_setStream: the caller's own work on those images is ordered on that stream, e.g. torch's current
    // stream) need not be complete at return -- nothing host-side can see them, and everything the caller enqueues next on that stream comes after (SURVEY 8b,
    // threading row): the hook returns after the enqueue, which removes the 12-15 us of hipStreamSynchronize from every per-frame call (profiles/r04_call_latency.txt).
    // Host and managed images, and calls on the library's own per-thread stream (which no caller can order against), stay synchronous.  MI355CV_DEVICE_SYNC=1: always block.
    static const bool alwaysBlock = envFlag("MI355CV_DEVICE_SYNC");
    const bool deferred = tctx().useUser && !t_sawManaged && !alwaysBlock;
    if (anyHost_ || (!asyncMode() && !deferred && t_depth == 1)) {
        // (A synchronous hook on a 4K frame is a 3-8 us kernel and hipStreamSynchronize adds 12-15 us of completion latency: 19.6 us per synchronous
        // cv_hal_gaussianBlurBinomial call against 8.2 us of enqueue + execution and 3.3 us of host time per asynchronous call, tools/ubench/call_latency.cpp,
        // profiles/r04_call_latency.txt.  Polling hipStreamQuery before blocking was tried and is SLOWER, 22.1 us: the query costs more than the wake-up it saves.)
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return setError(MI355CV_ERROR_UNKNOWN, "%s: execution failed: %s", entry, hipGetErrorString(e));
    } else if (t_depth == 1) t_notDrained = true;
    bump(entry);
    return MI355CV_OK;
}

// ------------------------------------------------------------------ pipelined host batches
bool hostBatchEligible(const void* src, const void* dst, int nframes)
{
    // no HIP state is touched here (resolveDevice names the device, ptrKind classifies the pointers): a batch this predicate rejects leaves the caller's device alone
    return nframes >= 1 && src && dst && !disabled() && resolveDevice() >= 0 && ptrKind(src) == PTR_HOST && ptrKind(dst) == PTR_HOST;
}

int runHostBatch(const char* entry, const HostBatch& hb, const HostBatchFn& run)
{
    if (disabled() || hb.nframes < 1 || hb.srows < 1 || hb.drows < 1 || !hb.srowBytes || !hb.drowBytes) return mi355::declined(__func__, __LINE__, "disabled() || hb.nframes < 1 || hb.srows < 1 || hb.drows < 1 || !hb.srowBytes || !hb.drowBytes");
    Stager stg;                                                          // outermost: the chunks' own hooks leave synchronisation to this one; first, so that a declined call also puts the host's device back
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    const size_t sp = (hb.srowBytes + 255) & ~(size_t)255, dp = (hb.drowBytes + 255) & ~(size_t)255;
    const size_t sfb = sp * (size_t)hb.srows, dfb = dp * (size_t)hb.drows;
    int cf = (int)((size_t)(64u << 20) / std::max(sfb, dfb));           // frames per chunk: <= 64 MB per buffer, <= 16 frames
    cf = cf < 1 ? 1 : cf > 16 ? 16 : cf; if (cf > hb.nframes) cf = hb.nframes;
    uchar* din[2]; uchar* dout[2];
    for (int b = 0; b < 2; b++) {
        din[b] = (uchar*)stg.scratch(sfb * cf); dout[b] = (uchar*)stg.scratch(dfb * cf);
        if (!din[b] || !dout[b]) return mi355::declined(__func__, __LINE__, "!din[b] || !dout[b]");
    }
    hipStream_t st = stream(), aux = auxStream();
    hipEvent_t inReady[2] = {pooledEvent(40), pooledEvent(41)}, bufFree[2] = {pooledEvent(42), pooledEvent(43)};
    if (!aux || !inReady[0] || !inReady[1] || !bufFree[0] || !bufFree[1]) return mi355::declined(__func__, __LINE__, "!aux || !inReady[0] || !inReady[1] || !bufFree[0] || !bufFree[1]");
    const int nchunks = (hb.nframes + cf - 1) / cf;
    auto upload = [&](int c) -> bool {
        const int b = c & 1, f0 = c * cf, nf = std::min(cf, hb.nframes - f0);
        if (c >= 2 && hipStreamWaitEvent(aux, bufFree[b], 0) != hipSuccess) return false;          // the buffers' previous chunk has been consumed and downloaded
        for (int f = 0; f < nf; f++)
            if (hipMemcpy2DAsync(din[b] + (size_t)f * sfb, sp, hb.src + (size_t)(f0 + f) * hb.sframe, hb.sstep, hb.srowBytes, hb.srows, hipMemcpyHostToDevice, aux) != hipSuccess)
                return false;
        return hipEventRecord(inReady[b], aux) == hipSuccess;
    };
    // every error exit below drains both streams first: ~Stager hands din[] / dout[] back to the pool, and DMA still in flight on the
    // auxiliary stream must not land in scratch the thread's next hook has been given
    auto fail = [&](int code) { (void)hipStreamSynchronize(aux); (void)hipStreamSynchronize(st); return code; };
    if (!upload(0)) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: H2D failed: %s", entry, hipGetErrorString(hipGetLastError())));
    std::vector<char> was;
    for (int c = 0; c < nchunks; c++) {
        const int b = c & 1, f0 = c * cf, nf = std::min(cf, hb.nframes - f0);
        if (c + 1 < nchunks && !upload(c + 1)) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: H2D failed: %s", entry, hipGetErrorString(hipGetLastError())));
        if (hipStreamWaitEvent(st, inReady[b], 0) != hipSuccess) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: %s", entry, hipGetErrorString(hipGetLastError())));
        auto& pool = tctx().pool;
        was.assign(pool.size(), 0);
        for (size_t i = 0; i < pool.size(); i++) was[i] = pool[i].busy;
        const int rc = run(din[b], sp, sfb, dout[b], dp, dfb, nf);
        // scratch the chunk's hook took from the pool is free for the next chunk's: same thread, same stream, stream order
        for (size_t i = 0; i < tctx().pool.size(); i++) if (tctx().pool[i].busy && (i >= was.size() || !was[i])) tctx().pool[i].busy = false;
        if (rc != MI355CV_OK) return fail(rc);
        for (int f = 0; f < nf; f++)
            if (hipMemcpy2DAsync(hb.dst + (size_t)(f0 + f) * hb.dframe, hb.dstep, dout[b] + (size_t)f * dfb, dp, hb.drowBytes, hb.drows, hipMemcpyDeviceToHost, st) != hipSuccess)
                return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: D2H failed: %s", entry, hipGetErrorString(hipGetLastError())));
        if (hipEventRecord(bufFree[b], st) != hipSuccess) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: %s", entry, hipGetErrorString(hipGetLastError())));
        g_stagedBytes += (long long)(hb.srowBytes * (size_t)hb.srows + hb.drowBytes * (size_t)hb.drows) * nf;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: %s", entry, hipGetErrorString(hipGetLastError())));
    return stg.finish(entry);
}

int runHostBatchN(const char* entry, const HostBatchN& hb, const HostBatchNFn& run)
{
    if (disabled() || hb.nframes < 1 || hb.srows < 1 || !hb.srowBytes || hb.nout < 1 || hb.nout > HOST_BATCH_MAX_OUT) return mi355::declined(__func__, __LINE__, "disabled() || hb.nframes < 1 || hb.srows < 1 || !hb.srowBytes || hb.nout < 1 || hb.nout > HOST_BATCH_MAX_OUT");
    for (int o = 0; o < hb.nout; o++) if (!hb.out[o].dst || hb.out[o].drows < 1 || !hb.out[o].drowBytes) return mi355::declined(__func__, __LINE__, "!hb.out[o].dst || hb.out[o].drows < 1 || !hb.out[o].drowBytes");
    Stager stg;                                                          // outermost: the chunks' own hooks leave synchronisation to this one
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    const size_t sp = (hb.srowBytes + 255) & ~(size_t)255, sfb = sp * (size_t)hb.srows;
    size_t dp[HOST_BATCH_MAX_OUT], dfb[HOST_BATCH_MAX_OUT], perFrame = sfb;
    for (int o = 0; o < hb.nout; o++) { dp[o] = (hb.out[o].drowBytes + 255) & ~(size_t)255; dfb[o] = dp[o] * (size_t)hb.out[o].drows; perFrame += dfb[o]; }
    int cf = (int)((size_t)(128u << 20) / perFrame);                     // frames per chunk: <= 128 MB per set of buffers, <= 16 frames
    cf = cf < 1 ? 1 : cf > 16 ? 16 : cf; if (cf > hb.nframes) cf = hb.nframes;
    uchar* din[2]; uchar* dout[2][HOST_BATCH_MAX_OUT]; size_t dfs[HOST_BATCH_MAX_OUT];
    for (int b = 0; b < 2; b++) {
        din[b] = (uchar*)stg.scratch(sfb * cf);
        if (!din[b]) return mi355::declined(__func__, __LINE__, "!din[b]");
        for (int o = 0; o < hb.nout; o++) { dout[b][o] = (uchar*)stg.scratch(dfb[o] * cf); dfs[o] = dfb[o]; if (!dout[b][o]) return mi355::declined(__func__, __LINE__, "!dout[b][o]"); }
    }
    hipStream_t st = stream(), aux = auxStream();
    hipEvent_t inReady[2] = {pooledEvent(44), pooledEvent(45)}, bufFree[2] = {pooledEvent(46), pooledEvent(47)};
    if (!aux || !inReady[0] || !inReady[1] || !bufFree[0] || !bufFree[1]) return mi355::declined(__func__, __LINE__, "!aux || !inReady[0] || !inReady[1] || !bufFree[0] || !bufFree[1]");
    const int nchunks = (hb.nframes + cf - 1) / cf;
    auto upload = [&](int c) -> bool {
        const int b = c & 1, f0 = c * cf, nf = std::min(cf, hb.nframes - f0);
        if (c >= 2 && hipStreamWaitEvent(aux, bufFree[b], 0) != hipSuccess) return false;          // the buffers' previous chunk has been consumed and downloaded
        for (int f = 0; f < nf; f++)
            if (hipMemcpy2DAsync(din[b] + (size_t)f * sfb, sp, hb.src + (size_t)(f0 + f) * hb.sframe, hb.sstep, hb.srowBytes, hb.srows, hipMemcpyHostToDevice, aux) != hipSuccess)
                return false;
        return hipEventRecord(inReady[b], aux) == hipSuccess;
    };
    auto fail = [&](int code) { (void)hipStreamSynchronize(aux); (void)hipStreamSynchronize(st); return code; };       // see runHostBatch
    if (!upload(0)) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: H2D failed: %s", entry, hipGetErrorString(hipGetLastError())));
    std::vector<char> was;
    for (int c = 0; c < nchunks; c++) {
        const int b = c & 1, f0 = c * cf, nf = std::min(cf, hb.nframes - f0);
        if (c + 1 < nchunks && !upload(c + 1)) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: H2D failed: %s", entry, hipGetErrorString(hipGetLastError())));
        if (hipStreamWaitEvent(st, inReady[b], 0) != hipSuccess) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: %s", entry, hipGetErrorString(hipGetLastError())));
        auto& pool = tctx().pool;
        was.assign(pool.size(), 0);
        for (size_t i = 0; i < pool.size(); i++) was[i] = pool[i].busy;
        const int rc = run(din[b], sp, sfb, dout[b], dp, dfs, nf);
        for (size_t i = 0; i < tctx().pool.size(); i++) if (tctx().pool[i].busy && (i >= was.size() || !was[i])) tctx().pool[i].busy = false;
        if (rc != MI355CV_OK) return fail(rc);
        long long bytes = (long long)(hb.srowBytes * (size_t)hb.srows) * nf;
        for (int o = 0; o < hb.nout; o++) {
            const HostBatchOut& ho = hb.out[o];
            for (int f = 0; f < nf; f++)
                if (hipMemcpy2DAsync(ho.dst + (size_t)(f0 + f) * ho.dframe, ho.dstep, dout[b][o] + (size_t)f * dfb[o], dp[o], ho.drowBytes, ho.drows, hipMemcpyDeviceToHost, st) != hipSuccess)
                    return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: D2H failed: %s", entry, hipGetErrorString(hipGetLastError())));
            bytes += (long long)(ho.drowBytes * (size_t)ho.drows) * nf;
        }
        if (hipEventRecord(bufFree[b], st) != hipSuccess) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: %s", entry, hipGetErrorString(hipGetLastError())));
        g_stagedBytes += bytes;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(setError(MI355CV_ERROR_UNKNOWN, "%s: %s", entry, hipGetErrorString(hipGetLastError())));
    return stg.finish(entry);
}

} // namespace mi355

// ------------------------------------------------------------------ exported runtime API
using namespace mi355;

extern "C" {

MI355CV_API int mi355cv_init(int device)
{
    if (device >= 0 && g_defaultDev.load() < 0 && device < deviceCount()) g_defaultDev = device;
    const bool ok = ensureDevice();
    restoreDevice();
    return ok ? 0 : -1;
}

MI355CV_API int mi355cv_deviceCount(void) { return deviceCount(); }

MI355CV_API int mi355cv_setDevice(int device)
{
    if (device < 0) { t_dev = -1; return 0; }
    if (device >= deviceCount()) { setError(MI355CV_ERROR_UNKNOWN, "mi355cv_setDevice(%d): %d device(s) visible%s%s", device, deviceCount(), g_nDevWhy[0] ? " -- " : "", g_nDevWhy); return -1; }
    const int before = t_dev;
    t_dev = device;
    if (resolveDevice() < 0) { t_dev = before; return -1; }
    return 0;
}

MI355CV_API int mi355cv_getDevice(void) { return resolveDevice(); }
} // extern "C" (reopened below)
namespace mi355 { int threadDeviceBinding() { return t_dev; } }
extern "C" {

MI355CV_API const char* mi355cv_version(void) { return "mi355cv 0.1 (gfx950; HAL mirror of OpenCV 4.12 imgproc hot path)"; }
MI355CV_API const char* mi355cv_lastError(void) { return t_err; }
MI355CV_API int mi355cv_setHostPolicy(int policy)
{
    if (policy < -1 || policy > 1) return MI355CV_ERROR_UNKNOWN;
    g_hostPolicy.store(policy, std::memory_order_relaxed);
    return MI355CV_OK;
}
MI355CV_API int mi355cv_hostPolicy(void)
{
    const int pol = g_hostPolicy.load(std::memory_order_relaxed);
    if (pol >= 0) return pol;
    const char* e = getenv("MI355CV_HOST_POLICY");
    return e && !strcmp(e, "always") ? 1 : 0;
}
MI355CV_API int mi355cv_limit(const char* key)
{
    using namespace mi355::lim;
    static const struct { const char* k; int v; } tab[] = {
        {"sep_max_taps", SEP_MAX_TAPS}, {"sep_max_taps_64f", SEP_MAX_TAPS_64F}, {"gauss8u_max_ksize", GAUSS8U_MAX_KSIZE}, {"gauss_float_max_ksize", GAUSS_FLOAT_MAX_KSIZE},
        {"adaptive_gaussian_max_block", SEP_MAX_TAPS}, {"adaptive_mean_max_block", ADAPTIVE_MEAN_MAX_BLOCK}, {"box_max_ksize", BOX_MAX_KSIZE},
        {"median8u_max_ksize", MEDIAN8U_MAX_KSIZE}, {"bilateral_max_d", 2 * BILATERAL_MAX_RADIUS + 1}, {"orb_max_levels", ORB_MAX_LEVELS}, {"filter2d_dft_taps", FILTER2D_DFT_TAPS}};
    if (!key) return -1;
    for (const auto& e : tab) if (!strcmp(key, e.k)) return e.v;
    return -1;
}
MI355CV_API const char* mi355cv_lastKernel(void) { return t_kernel; }

// Scratch buffers last used asynchronously on a caller-bound stream are tagged with it (Stager::bump_ polls that stream before handing them to another).  When the binding
// changes, the stream being left is drained once and the tags dropped: the caller may destroy it afterwards without leaving a dead handle behind for hipStreamQuery
// (ADVICE r5).  Contract: unbind (mi355cv_setStream to another stream / mi355cv_resetStream) BEFORE destroying a stream that was bound.
static void leaveUserStream(ThreadCtx& c, hipStream_t next, bool nextIsUser)
{
    if (!c.useUser || (nextIsUser && next == c.user)) return;
    bool tagged = false;
    for (auto& b : c.pool) if (b.last == c.user && b.last) tagged = true;
    if (!tagged) return;
    if (ensureDevice()) { (void)hipStreamSynchronize(c.user); (void)hipGetLastError(); restoreDevice(); }
    for (auto& b : c.pool) if (b.last == c.user) b.last = nullptr;
}

MI355CV_API int mi355cv_setStream(void* s)
{
    if (resolveDevice() < 0) return -1;
    ThreadCtx& c = tctx();
    leaveUserStream(c, (hipStream_t)s, true);
    c.user = (hipStream_t)s; c.useUser = true;    // NULL is HIP's null (legacy default) stream
    return 0;
}

MI355CV_API int mi355cv_resetStream(void)
{
    if (resolveDevice() < 0) return -1;
    ThreadCtx& c = tctx();
    leaveUserStream(c, nullptr, false);
    c.useUser = false;
    return 0;
}

MI355CV_API int mi355cv_setAsync(int enable)
{
    if (resolveDevice() < 0) return -1;
    tctx().async = enable != 0;             // per thread and device, like the stream binding
    return 0;
}

MI355CV_API int mi355cv_synchronize(void)
{
    if (!ensureDevice()) return -1;
    const bool ok = hipStreamSynchronize(stream()) == hipSuccess;
    restoreDevice();
    return ok ? 0 : -1;
}

MI355CV_API long long mi355cv_callCount(const char* entry)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_counts.find(entry ? entry : "");
    return it == g_counts.end() ? 0 : it->second;
}

MI355CV_API void mi355cv_noteDecline(const char* hook)
{
    std::lock_guard<std::mutex> lk(g_mu);
    hookPrintCounts();
    auto& d = g_declines[hook ? hook : "?"];
    d.first++;
    d.second = t_errFresh ? t_err : "no reason recorded (argument combination outside the GPU path)";
    t_errFresh = false;
    if (ledgerLog()) { fprintf(stderr, "[mi355cv] declined %s: %s\n", hook ? hook : "?", d.second.c_str()); fflush(stderr); }
}

MI355CV_API long long mi355cv_declineCount(const char* hook)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!hook) { long long n = 0; for (const auto& kv : g_declines) n += kv.second.first; return n; }
    auto it = g_declines.find(hook);
    return it == g_declines.end() ? 0 : it->second.first;
}

MI355CV_API long long mi355cv_stagedBytes(void) { return g_stagedBytes.load(); }

MI355CV_API void* mi355cv_deviceAlloc(size_t bytes)
{
    if (!ensureDevice()) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    restoreDevice();
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
    if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    restoreDevice();
    return p;
}
MI355CV_API int mi355cv_hostFree(void* p, int kind)
{
    if (!p) return 0;
    return (kind == 0 ? hipHostFree(p) : hipFree(p)) == hipSuccess ? 0 : -1;
}
MI355CV_API int mi355cv_upload(void* d, const void* h, size_t n)
{
    const bool ok = ensureDevice() && hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess;
    restoreDevice();
    return ok ? 0 : -1;
}
MI355CV_API int mi355cv_download(void* h, const void* d, size_t n)
{
    const bool ok = ensureDevice() && hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess;
    restoreDevice();
    return ok ? 0 : -1;
}

} // extern "C"
