// hist.hip -- SURVEY.md §8 f1: cv_hal_equalize_hist (hal_replacement.hpp:1120; caller histogram.cpp:3455) and cv_hal_threshold_otsu
// (:1077; caller thresh.cpp:1568).  Both are "histogram -> a few hundred scalar operations -> per-pixel table / compare":
//   k_hist<T>   per-workgroup LDS histogram (CV_8U: 256 bins, one sub-histogram per wave to spread the atomics) or global atomics
//               (CV_16U: 65 536 bins), merged into HBM with one atomicAdd per non-empty bin
//   host        the cumulative table (float scale, cvRound) / the Otsu scan in double precision, exactly the reference's loops --
//               256 or 65 536 iterations are not worth a kernel, and the double arithmetic stays bit-identical to the CPU's
//   k_lut / mi355cv_threshold   the per-pixel pass
#include "rt.h"
#include <cfloat>
#include <cmath>
#include <vector>

using namespace mi355;

extern "C" MI355CV_API int mi355cv_threshold(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                             int depth, int cn, double thresh, double maxValue, int thresholdType);

namespace {

__global__ __launch_bounds__(256) void k_hist_u8(const uchar* __restrict__ src, size_t sstep, int W, int H, int rowsPerBlock, unsigned* __restrict__ hist)
{
    __shared__ unsigned h[4][256];
    for (int i = threadIdx.x; i < 1024; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    unsigned* mine = h[threadIdx.x >> 6];
    const int y0 = blockIdx.x * rowsPerBlock, y1 = min(y0 + rowsPerBlock, H);
    for (int y = y0; y < y1; y++) {
        const uchar* row = src + (size_t)y * sstep;
        const int head = min((int)((16 - ((uintptr_t)row & 15)) & 15), W);      // bytes before the first 16-byte boundary
        if ((int)threadIdx.x < head) atomicAdd(&mine[row[threadIdx.x]], 1u);
        const int nvec = (W - head) / 16;
        const uint4* v = (const uint4*)(row + head);
        for (int i = threadIdx.x; i < nvec; i += 256) {
            const uint4 q = v[i];
            const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&mine[w4[k] & 255], 1u); atomicAdd(&mine[(w4[k] >> 8) & 255], 1u);
                atomicAdd(&mine[(w4[k] >> 16) & 255], 1u); atomicAdd(&mine[w4[k] >> 24], 1u);
            }
        }
        const int tail0 = head + nvec * 16;
        if (tail0 + (int)threadIdx.x < W) atomicAdd(&mine[row[tail0 + threadIdx.x]], 1u);
    }
    __syncthreads();
    const unsigned s = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
    if (s) atomicAdd(&hist[threadIdx.x], s);
}

__global__ __launch_bounds__(256) void k_hist_u16(const uchar* __restrict__ src, size_t sstep, int W, int H, unsigned* __restrict__ hist)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    atomicAdd(&hist[((const unsigned short*)(src + (size_t)y * sstep))[x]], 1u);
}

__global__ __launch_bounds__(256) void k_lut_u8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                const uchar* __restrict__ lut)
{
    __shared__ uchar t[256];
    t[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= W || y >= H) return;
    const uchar* s = src + (size_t)y * sstep + x4;
    uchar* d = dst + (size_t)y * dstep + x4;
#pragma unroll
    for (int k = 0; k < 4; k++) if (x4 + k < W) d[k] = t[s[k]];
}

// the cumulative table of cv::equalizeHist (histogram.cpp:3472-3489) from the 256-bin histogram, one wave, bin i on lane i % 64: the
// integer prefix sums are exact in any order, the float scale and the cvRound are single IEEE operations, so this equals the host loop
__global__ __launch_bounds__(64) void k_equalize_lut(const unsigned* __restrict__ hist, int total, uchar* __restrict__ lut)
{
    __shared__ unsigned h[256];
    __shared__ int first;
    for (int i = threadIdx.x; i < 256; i += 64) h[i] = hist[i];
    if (threadIdx.x == 0) first = 256;
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) if (h[i]) atomicMin(&first, i);
    __syncthreads();
    const int i0 = first;
    if ((int)h[i0] == total) { for (int i = threadIdx.x; i < 256; i += 64) lut[i] = (uchar)i0; return; }        // dst.setTo(i)
    const float scale = __fdiv_rn(255.f, (float)(total - (int)h[i0]));
    __shared__ __attribute__((aligned(4))) uchar l[256];
    if (threadIdx.x == 0) {                                                 // 255 dependent integer adds: run them against LDS, not HBM
        int sum = 0;
        for (int i = 0; i <= i0; i++) l[i] = 0;
        for (int i = i0 + 1; i < 256; i++) {
            sum += (int)h[i];
            l[i] = (uchar)min(max(__float2int_rn((float)sum * scale), 0), 255);
        }
    }
    __syncthreads();
    reinterpret_cast<unsigned*>(lut)[threadIdx.x] = reinterpret_cast<const unsigned*>(l)[threadIdx.x];
}

// histogram of a device-resident image into `host` (nbins entries); synchronises the stream
bool histogramToHost(Stager& stg, const uchar* ds, size_t dss, int width, int height, int depth, std::vector<int>& host)
{
    const int nbins = depth == MI355CV_8U ? 256 : 65536;
    unsigned* dh = (unsigned*)stg.scratch((size_t)nbins * 4);
    if (!dh) return false;
    hipStream_t st = stream();
    if (hipMemsetAsync(dh, 0, (size_t)nbins * 4, st) != hipSuccess) return false;
    if (depth == MI355CV_8U) {
        // every workgroup ends with one global atomic per non-empty bin, all workgroups onto the same 256 words: keep the number of
        // workgroups near two per CU (1080 of them spent 20 us of a 24 us 4K histogram queueing on those atomics)
        const int rowsPerBlock = std::max(1, divUp(height, 512));
        hipLaunchKernelGGL(k_hist_u8, dim3(divUp(height, rowsPerBlock)), dim3(256), 0, st, ds, dss, width, height, rowsPerBlock, dh);
    } else
        hipLaunchKernelGGL(k_hist_u16, dim3(divUp(width, 64), divUp(height, 4)), dim3(256), 0, st, ds, dss, width, height, dh);
    host.resize(nbins);
    if (hipMemcpyAsync(host.data(), dh, (size_t)nbins * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    return hipStreamSynchronize(st) == hipSuccess;
}

} // namespace

extern "C" {

MI355CV_API int mi355cv_equalize_hist(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0 || (long long)width * height > 0x7fffffffLL) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || (long long)width * height > 0x7fffffffLL");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    // histogram -> table -> per-pixel pass, all in stream order: no host round trip, so the hook stays asynchronous for images in HBM
    unsigned* dh = (unsigned*)stg.scratch(256 * 4);
    uchar* dl = (uchar*)stg.scratch(256);
    if (!dh || !dl) return mi355::declined(__func__, __LINE__, "!dh || !dl");
    hipStream_t st = stream();
    if (hipMemsetAsync(dh, 0, 256 * 4, st) != hipSuccess) return MI355CV_ERROR_UNKNOWN;
    const int rowsPerBlock = std::max(1, divUp(height, 512));
    hipLaunchKernelGGL(k_hist_u8, dim3(divUp(height, rowsPerBlock)), dim3(256), 0, st, ds, dss, width, height, rowsPerBlock, dh);
    hipLaunchKernelGGL(k_equalize_lut, dim3(1), dim3(64), 0, st, dh, width * height, dl);
    hipLaunchKernelGGL(k_lut_u8, dim3(divUp(divUp(width, 4), 64), divUp(height, 4)), dim3(256), 0, stream(), ds, dss, dd, dds, width, height, dl);
    return stg.finish("equalize_hist");
}

// `depth` receives src.type() from cv::threshold (thresh.cpp:1568): CV_8UC1 == 0 or CV_16UC1 == 2
MI355CV_API int mi355cv_threshold_otsu(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int depth,
                                       double maxValue, int thresholdType, double* thresh)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || !thresh || width <= 0 || height <= 0 || (depth != MI355CV_8U && depth != MI355CV_16U) || thresholdType < 0 || thresholdType > 4)
        return mi355::declined(__func__, __LINE__, "disabled() || !thresh || width <= 0 || height <= 0 || (depth != MI355CV_8U && depth != MI355CV_16U) || thresholdType < 0 || thresholdType > 4");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if ((long long)width * height > 0x7fffffffLL || !ensureDevice()) return mi355::declined(__func__, __LINE__, "(long long)width * height > 0x7fffffffLL || !ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
    const int e = depth == MI355CV_8U ? 1 : 2;
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * e, height, &dss);
    // THRESH_DRYRUN (thresh.cpp:1550-1557): cv::threshold wants the level only and hands an EMPTY destination (data == NULL): the histogram and the level, no pass over the image
    uchar* dd = dst_data ? stg.out(dst_data, dst_step, (size_t)width * e, height, &dds) : nullptr;
    if (!ds || (dst_data && !dd)) return mi355::declined(__func__, __LINE__, "!ds || (dst_data && !dd)");
    std::vector<int> h;
    if (!histogramToHost(stg, ds, dss, width, height, depth, h)) return MI355CV_ERROR_UNKNOWN;
    // getThreshVal_Otsu, thresh.cpp:1158-1192
    const int N = (int)h.size();
    double mu = 0, scale = 1. / (width * height);
    for (int i = 0; i < N; i++) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < N; i++) {
        const double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    // the fixed-level pass as cv::threshold sets it up (thresh.cpp:1580-1590 / :1640-1650); the Otsu level is always below the type's maximum
    const int hi = depth == MI355CV_8U ? 255 : 65535;
    const int ithresh = (int)max_val;
    int imaxval = (int)std::lrint(maxValue);
    if (thresholdType == 2) imaxval = ithresh;
    imaxval = std::min(std::max(imaxval, 0), hi);
    if (dd) {
        const int rc = mi355cv_threshold(ds, dss, dd, dds, width, height, depth, 1, ithresh, imaxval, thresholdType);
        if (rc != MI355CV_OK) return rc;
    }
    *thresh = max_val;
    return stg.finish("threshold_otsu");
}

} // extern "C"
