// Includes the reference's imgproc hal_replacement.hpp the way modules/imgproc does, with the custom_hal.hpp cmake generated from the HAL package:
// the cv_hal_* names below are therefore whatever the package bound them to.  Prints which implementation each resolves to and calls two hooks.
#include "opencv2/core.hpp"
#include "opencv2/core/hal/interface.h"
#include "opencv2/imgproc/hal/interface.h"
#include "hal_replacement.hpp"
#include <cstdio>
#include <vector>
#define STR2(x) #x
#define STR(x) STR2(x)
int main()
{
    // the bindings are function-like macros (every call goes through the decline-counting shim of mi355cv_hal.hpp): expand one call of each
    std::printf("cv_hal_gaussianBlurBinomial -> %s\n", STR(cv_hal_gaussianBlurBinomial(ARGS)));
    std::printf("cv_hal_resize -> %s\n", STR(cv_hal_resize(ARGS)));
    std::printf("cv_hal_cvtBGRtoGray -> %s\n", STR(cv_hal_cvtBGRtoGray(ARGS)));
    std::vector<uchar> src(64 * 48, 7), dst(64 * 48, 0);
    // CV_HAL_ERROR_OK on a gfx950 host, CV_HAL_ERROR_NOT_IMPLEMENTED (the caller falls back) anywhere else: both are the contract
    const int rc = cv_hal_gaussianBlurBinomial(src.data(), 64, dst.data(), 64, 64, 48, CV_8U, 1, 0, 0, 0, 0, 5, 4);
    std::printf("gaussianBlurBinomial rc=%d dst[100]=%d declined=%lld\n", rc, (int)dst[100], mi355cv_declineCount("gaussianBlurBinomial"));
    if ((rc == CV_HAL_ERROR_NOT_IMPLEMENTED) != (mi355cv_declineCount("gaussianBlurBinomial") == 1)) return 3;      // the ledger follows the return code
    if (rc != CV_HAL_ERROR_OK && rc != CV_HAL_ERROR_NOT_IMPLEMENTED) return 1;
    if (rc == CV_HAL_ERROR_OK && dst[100] != 7) return 2;
    std::printf("version: %s\n", mi355cv_version());
    return 0;
}
