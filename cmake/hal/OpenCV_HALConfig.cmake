# OpenCV_HALConfig.cmake -- registers libmi355cv.so as the custom HAL of an OpenCV build (the package the reference looks for with
# find_package(OpenCV_HAL NO_MODULE), CMakeLists.txt:946-948, :1033-1039; same four variables as samples/hal/c_hal/config.cmake).
#
#   make -C <this repo>/opencv_amd/csrc                                  # builds opencv_amd/libmi355cv.so (hipcc --offload-arch=gfx950)
#   cmake -S <opencv> -B build -DOpenCV_HAL_DIR=<this repo>/cmake/hal [-DWITH_IPP=OFF ...]
#
# OpenCV then writes `#include "mi355cv_hal.hpp"` into its generated custom_hal.hpp (cmake/templates/custom_hal.hpp.in), which every
# module's hal_replacement.hpp includes (imgproc :1339, video, features2d), adds include/ to the include path and links libmi355cv.so into
# the modules.  On a host without a gfx950 device every hook answers CV_HAL_ERROR_NOT_IMPLEMENTED and the stock paths run.
get_filename_component(_mi355cv_root "${CMAKE_CURRENT_LIST_DIR}/../.." ABSOLUTE)
set(OpenCV_HAL_VERSION 0.2.0)
set(OpenCV_HAL_LIBRARIES "${_mi355cv_root}/opencv_amd/libmi355cv.so")
set(OpenCV_HAL_HEADERS "mi355cv_hal.hpp")
set(OpenCV_HAL_INCLUDE_DIRS "${_mi355cv_root}/include")
if(EXISTS "${OpenCV_HAL_LIBRARIES}" AND EXISTS "${OpenCV_HAL_INCLUDE_DIRS}/mi355cv_hal.hpp")
  set(OpenCV_HAL_FOUND TRUE)
else()
  set(OpenCV_HAL_FOUND FALSE)
  message(WARNING "mi355cv HAL: ${OpenCV_HAL_LIBRARIES} is missing -- build it first (make -C ${_mi355cv_root}/opencv_amd/csrc)")
endif()
unset(_mi355cv_root)
