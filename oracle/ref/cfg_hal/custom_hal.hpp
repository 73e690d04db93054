/* custom_hal.hpp for the HAL-enabled build of the reference (oracle/_ref/libocvref_hal.so):
 * what cmake generates from cmake/templates/custom_hal.hpp.in when OpenCV is configured with
 * -DOpenCV_HAL_DIR=<dir>: one #include per registered HAL header.  Ours routes the imgproc
 * cv_hal_* hooks to libmi355cv.so (include/mi355cv_hal.hpp). */
#ifndef _CUSTOM_HAL_INCLUDED_
#define _CUSTOM_HAL_INCLUDED_
#include "mi355cv_hal.hpp"
#endif
