// gftt_sort.hip -- the one place a vendor primitive is used: rocPRIM's device-wide radix sort orders the corner candidates of
// cv::goodFeaturesToTrack (64-bit keys: response, then pixel index, both descending = the reference's std::sort with greaterThanPtr,
// featureselect.cpp:55-60, :447).  Kept in its own translation unit because the rocPRIM headers dominate its compile time.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "rt.h"

namespace mi355 {

size_t sortKeysDescTemp(unsigned n)
{
    size_t bytes = 0;
    if (rocprim::radix_sort_keys_desc(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n, 0, 64, (hipStream_t)0) != hipSuccess) return 0;
    return bytes;
}

bool sortKeysDesc(void* temp, size_t bytes, const unsigned long long* in, unsigned long long* out, unsigned n, hipStream_t st)
{
    return rocprim::radix_sort_keys_desc(temp, bytes, in, out, n, 0, 64, st) == hipSuccess;
}

} // namespace mi355
