// rt.h -- internal runtime of libmi355cv.so: per-thread HIP stream, pointer
// classification, host<->HBM staging, error/trace bookkeeping.
// Everything above this (the exported mi355cv_* hooks) speaks the HAL contract
// of the reference (hal_replacement.hpp:1342-1357): never throw, return
// OK / NOT_IMPLEMENTED / UNKNOWN.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include <functional>
#include "../../include/mi355cv.h"

typedef unsigned char uchar;

namespace mi355 {

// border codes (core/base.hpp:332)
enum { B_CONSTANT = 0, B_REPLICATE = 1, B_REFLECT = 2, B_WRAP = 3, B_REFLECT_101 = 4, B_TRANSPARENT = 5 };

// Capacity bounds of the served paths -- "how far this was built", as opposed to the cases the reference itself has no engine for.  ONE definition: the entry points
// compare against these, mi355cv_limit() reports them, tests/test_declines_cpu.py pins their values on the CPU and the -m gpu tests derive every refusal they assert from
// mi355cv_limit(), so that widening a path without widening its parity cases fails a CPU test.
namespace lim {
constexpr int SEP_MAX_TAPS = 129;          // cv_hal_sepFilter: taps per axis (seplong.hip); a Gaussian of sigma 16 on CV_32F has 97
constexpr int SEP_MAX_TAPS_64F = 33;       // ... into CV_64F (k_sepfilter_generic64)
constexpr int GAUSS8U_MAX_KSIZE = 129;     // cv_hal_gaussianBlur on CV_8U (Q8.8 taps)
constexpr int GAUSS_FLOAT_MAX_KSIZE = 129; // cv_hal_gaussianBlur on CV_16U / CV_16S / CV_32F (= the separable hook's)
constexpr int ADAPTIVE_MEAN_MAX_BLOCK = 255;
constexpr int BOX_MAX_KSIZE = 255;
constexpr int MEDIAN8U_MAX_KSIZE = 31;
constexpr int BILATERAL_MAX_RADIUS = 16;
constexpr int ORB_MAX_LEVELS = 32;
constexpr int FILTER2D_DFT_TAPS = 130;     // whole-image filter2D from this many taps on is the reference's DFT case: declined unless MI355CV_FILTER_LARGE=1
}

struct ThreadCtx;
ThreadCtx& tctx();
hipStream_t stream();
hipStream_t auxStream();            // a second per-thread stream for work that may overlap the main one (ordered with events)
hipEvent_t pooledEvent(int i);      // per-thread reusable events (timing disabled), i < 64
bool asyncMode();
bool disabled();                    // MI355CV_DISABLE=1 -> every hook answers NOT_IMPLEMENTED
// Which host-resident images a hook accepts (device / managed pointers are always served).  MI355CV_MIN_PIXELS=<n>: images below n pixels
// are declined.  MI355CV_HOST_POLICY=auto (the default): a host image is staged through HBM only by hooks whose CPU path costs more than the two PCIe
// crossings (HOST_HEAVY: the reference's single-threaded FilterEngine paths -- filter2D, sepFilter2D, Sobel, box -- and medianBlur, Canny,
// corners, cubic / Lanczos / area resize, warps, LK, Otsu, bilateralFilter, float Lab / Luv, morphology with irregular elements or on deeper images); bandwidth-bound hooks whose CPU path is multi-threaded (HOST_CHEAP: 8U Gaussian,
// colour conversions, threshold, pyrDown, equalizeHist, morphology with CV_8U rectangles, integral, bilinear resize) decline and leave the image to the CPU.
// MI355CV_HOST_POLICY=always accepts everything, which is what the parity tests and the HAL tour set (tests/conftest.py).
enum HostCost { HOST_CHEAP = 0, HOST_HEAVY = 1 };
size_t minPixels(int cost = HOST_CHEAP);
// host-resident image below the policy threshold: true, and the reason is recorded for mi355cv_lastError / the decline ledger
bool hostImageTooSmall(const void* img, size_t pixels, size_t threshold);
int  setError(int code, const char* fmt, ...);
// a hook answers NOT_IMPLEMENTED: records WHY for mi355cv_lastError and the decline ledger (mi355cv_noteDecline) -- "<entry>:<line>: <the condition that held>" --
// unless a more specific reason was already recorded during this call (staging failure, host-policy threshold, foreign device ...).  Returns MI355CV_NOT_IMPLEMENTED.
int  declined(const char* fn, int line, const char* cond);
// first line of every extern "C" entry: the outermost entry on a thread starts a new call serial, so that a reason recorded by an EARLIER call that failed before it
// opened a Stager (argument checks, runSharded / replicate errors) is never reported as this call's; nested entries keep their caller's serial
// MI355CV_TRACE=1: the guard also opens a roctx range named after the entry (roctxRangePushA / roctxRangePop of librocprofiler-sdk-roctx.so or libroctx64.so, resolved with
// dlopen at the first traced call -- the library itself links HIP only), so that rocprofv3 --marker-trace / a timeline shows every cv_hal_* hook as a range around its
// kernels (the reference's CV_INSTRUMENT_REGION / CV_TRACE_REGION role, core/private.hpp:794; SURVEY section 5 tracing row).
struct EntryGuard { explicit EntryGuard(const char* name); ~EntryGuard(); EntryGuard(const EntryGuard&) = delete; EntryGuard& operator=(const EntryGuard&) = delete; bool traced_ = false; };
void beginCall();                   // start of a hook invocation (Stager's constructor): reasons recorded by earlier calls no longer count as this call's
void bump(const char* entry);       // per-entry completed-on-GPU counter
void noteKernel(const char* fmt, ...);   // name + launch geometry of the dominant kernel the calling thread launched last (mi355cv_lastKernel)
bool ensureDevice();                // makes the calling thread's device current (mi355cv_setDevice, else the process default); false if no usable GPU
int  activeDevice();                // ordinal of that device (per-device caches key on it)
int  threadDeviceBinding();         // what mi355cv_setDevice last set on the calling thread (-1: the process default)

// where an image lives: plain / page-locked host memory (staged through HBM), this thread's device or managed memory (launched in place),
// or ANOTHER GPU's memory (the hook declines: the thread is bound to the wrong device for that image)
enum PtrKind { PTR_HOST = 0, PTR_DEVICE = 1, PTR_FOREIGN = 2 };
int  ptrKind(const void* p);
void noteStagedBytes(long long n);                 // PCIe bytes moved outside Stager::in / out (pipelined host batches)
// true if p points into this device's or managed memory (launch in place)
bool isDevicePtr(const void* p);
// src and dst are the same buffer in HBM: a stencil cannot run in place on the GPU (host images are staged into separate buffers, so they may)
inline bool inPlaceOnDevice(const void* src, const void* dst) { return src == dst && src && isDevicePtr(src); }
// the general form for hooks that get no allowInplace argument (cv_hal_sepFilter, cv_hal_filter): the device-resident rows a stencil reads,
// [s, s + sbytes) -- the ROI plus the parent rows above / below it -- intersect the rows it writes.  FilterEngine is in-place safe on the CPU
// (ring buffer of rows); GPU threads would read neighbours other threads have already overwritten, so such a call is left to the CPU.
inline bool overlapOnDevice(const void* s, size_t sbytes, const void* d, size_t dbytes)
{
    const char* a = (const char*)s; const char* b = (const char*)d;
    if (!a || !b || a + sbytes <= b || b + dbytes <= a) return false;
    return isDevicePtr(s) && isDevicePtr(d);
}

// Stages host images into HBM scratch (and results back).  Device-resident
// images pass through untouched.  One Stager per hook invocation.
class Stager {
public:
    Stager();
    ~Stager();
    // input image: `rows` rows of `rowBytes` valid bytes at `p` with pitch `step`.
    const uchar* in(const uchar* p, size_t step, size_t rowBytes, int rows, size_t* dstep);
    // output image (contents undefined until the kernel writes them)
    uchar* out(uchar* p, size_t step, size_t rowBytes, int rows, size_t* dstep);
    // small parameter blocks (filter taps, tables): always copied, 256-B aligned
    void* param(const void* host, size_t bytes);
    // scratch in HBM that lives until finish()
    void* scratch(size_t bytes);
    // page-locked host memory that lives until the outermost hook returns (a per-thread pool, like the HBM scratch): the landing zone of results whose
    // size the GPU decides (candidate lists) -- a device-to-host copy into pageable memory goes through the runtime's own bounce buffer at a fraction of
    // the PCIe rate and blocks the host meanwhile
    void* pinned(size_t bytes);
    // copies staged outputs back and synchronises when required.  Returns HAL code.
    int finish(const char* entry);
    bool anyHost() const { return anyHost_; }
    bool failed() const { return failed_; }
private:
    struct Out { uchar* host; size_t hstep; uchar* dev; size_t dstep; size_t rowBytes; int rows; };
    std::vector<Out> outs_;
    bool anyHost_ = false, failed_ = false;
    void* bump_(size_t bytes);
    bool foreign_(const void* p);
};

// A batch of whole frames that lives in HOST memory (ingest / egress): the frames cross PCIe in chunks through two sets of device
// buffers -- the upload of chunk i+1 (aux stream) overlaps the kernels and the download of chunk i (main stream) -- and `run` is the entry's own
// device-resident batch path, called once per chunk with dense device frames.  Source and destination frames may differ in geometry and type.
struct HostBatch {
    const uchar* src; size_t sstep, sframe, srowBytes; int srows;       // source frames: pitch, frame stride, valid bytes per row, rows
    uchar* dst; size_t dstep, dframe, drowBytes; int drows;
    int nframes;
};
typedef std::function<int(const uchar* s, size_t sstep, size_t sframe, uchar* d, size_t dstep, size_t dframe, int nframes)> HostBatchFn;
// both ends are plain (pageable or page-locked) host memory: the batch entries take runHostBatch then
bool hostBatchEligible(const void* src, const void* dst, int nframes);
int runHostBatch(const char* entry, const HostBatch& hb, const HostBatchFn& run);

// The same pipeline for entries with several outputs per frame (cv::buildPyramid: one image per level): every output has its own geometry, the chunk
// size is what the largest of them allows.  `run` gets the chunk's dense device frames and one (pointer, pitch, frame stride) triple per output.
constexpr int HOST_BATCH_MAX_OUT = 32;
struct HostBatchOut { uchar* dst; size_t dstep, dframe, drowBytes; int drows; };
struct HostBatchN {
    const uchar* src; size_t sstep, sframe, srowBytes; int srows;
    int nout; HostBatchOut out[HOST_BATCH_MAX_OUT];
    int nframes;
};
typedef std::function<int(const uchar* s, size_t sstep, size_t sframe, uchar* const* d, const size_t* dstep, const size_t* dframe, int nframes)> HostBatchNFn;
int runHostBatchN(const char* entry, const HostBatchN& hb, const HostBatchNFn& run);

inline int depthBytes(int depth) { return depth <= 1 ? 1 : depth <= 3 ? 2 : depth <= 5 ? 4 : 8; }     // CV_8U .. CV_64F
inline int divUp(int a, int b) { return (a + b - 1) / b; }

#if defined(__HIPCC__)
// Workgroups of a 1-D grid go to the 8 XCDs round-robin (id % 8), each XCD behind its own L2.  A kernel whose neighbouring workgroups write the two halves of
// the same 128-byte lines (rows whose pitch is not a multiple of the line: cv::integral's (W + 1)-element rows) sends every such line to memory twice as a
// partial write when the neighbours sit on different XCDs -- measured with tools/probes/fillbw2.hip: 3.1-3.9 TB/s for rows of 3841 ints against 6.5 TB/s
// when each XCD owns one contiguous run of workgroups.  This is that mapping: logical id = (id % 8) * (n / 8) + id / 8 (the last n % 8 ids keep their own).
__device__ __forceinline__ unsigned xcdContiguous(unsigned id, unsigned n)
{
    const unsigned per = n >> 3;
    return id < (per << 3) ? (id & 7u) * per + (id >> 3) : id;
}
#endif

#define MI355_CHECK_LAUNCH(entry)                                                        \
    do { hipError_t e__ = hipGetLastError();                                             \
         if (e__ != hipSuccess) return mi355::setError(MI355CV_ERROR_UNKNOWN, "%s: launch failed: %s", entry, hipGetErrorString(e__)); } while (0)

#ifdef __HIPCC__
// (a << n) + b as ONE v_lshl_add_u32, and a + 4 b + 6 c on packed u16 pairs in four of them: written as plain C the optimiser folds the
// 4 c + 2 c into a 32-bit multiply by 6 (v_mul_lo_u32), which is not a full-rate instruction
__device__ __forceinline__ uint32_t lshlAdd(uint32_t a, int n, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(n), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sum146(uint32_t a, uint32_t b, uint32_t c) { return lshlAdd(c, 1, lshlAdd(b + c, 2, a)); }
#endif
} // namespace mi355

// borderInterpolate (core/src/copy.cpp:748-793), device+host. Returns -1 for CONSTANT.
__host__ __device__ inline int mi355_borderInterpolate(int p, int len, int borderType)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (borderType == mi355::B_REPLICATE) return p < 0 ? 0 : len - 1;
    if (borderType == mi355::B_REFLECT || borderType == mi355::B_REFLECT_101) {
        int delta = borderType == mi355::B_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (borderType == mi355::B_WRAP) {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1;
}
