// mi355cv_hal.hpp -- OpenCV custom-HAL header: routes the reference's imgproc cv_hal_* hooks to
// libmi355cv.so.  Register it the way samples/hal/c_hal does (reference: samples/hal/c_hal/impl.h,
// CMakeLists.txt:925-1043): configure OpenCV with -DOpenCV_HAL_DIR=<dir with OpenCV_HALConfig.cmake>
// whose OpenCV_HAL_HEADERS lists this file and OpenCV_HAL_LIBRARIES lists libmi355cv.so.
// It is included by the generated custom_hal.hpp from modules/imgproc/src/hal_replacement.hpp:1339,
// after the hal_ni_* stubs, so each #undef/#define pair below swaps one stub for our entry point.
// See INTEGRATION.md.
#ifndef MI355CV_HAL_HPP
#define MI355CV_HAL_HPP

#include "mi355cv.h"

// hal_replacement.hpp:1169 / caller smooth.dispatch.cpp:696
#undef  cv_hal_gaussianBlurBinomial
#define cv_hal_gaussianBlurBinomial mi355cv_gaussianBlurBinomial

#endif
