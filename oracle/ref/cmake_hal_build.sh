#!/bin/bash
# TEST INFRASTRUCTURE.  The reference built by ITS OWN build system (cmake + ninja, offline) with this repository registered as its custom HAL:
#   cmake -S /root/reference ... -DOpenCV_HAL_DIR=<repo>/cmake/hal      (CMakeLists.txt:946-948, :1033-1039 find_package(OpenCV_HAL NO_MODULE))
# The reference's cmake writes `#include "mi355cv_hal.hpp"` into its generated custom_hal.hpp and links libmi355cv.so into its modules; its own
# opencv_test_imgproc / opencv_test_video / opencv_test_features2d then run with every cv_hal_* hook bound to this library (tests/test_cmake_reference_build.py).
# Build tree: $BUILD (default /tmp/ocv_cmake_hal, outside the repository); what has to travel to the GPU box is copied to oracle/_ref/cmake_hal/.
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
REF=${REF:-/root/reference}
BUILD=${BUILD:-/tmp/ocv_cmake_hal}
OUT=$REPO/oracle/_ref/cmake_hal
test -f $REPO/opencv_amd/libmi355cv.so || make -C $REPO/opencv_amd/csrc -j8
cmake -S $REF -B $BUILD -GNinja -DCMAKE_BUILD_TYPE=Release -DOpenCV_HAL_DIR=$REPO/cmake/hal \
  -DBUILD_LIST=core,imgproc,imgcodecs,videoio,highgui,ts,video,features2d,flann -DWITH_IPP=OFF -DWITH_OPENCL=OFF -DWITH_ITT=OFF -DWITH_FFMPEG=OFF -DWITH_GSTREAMER=OFF \
  -DWITH_V4L=OFF -DWITH_GTK=OFF -DWITH_QT=OFF -DWITH_1394=OFF -DWITH_OPENEXR=OFF -DWITH_JASPER=OFF -DWITH_OPENJPEG=OFF -DWITH_WEBP=OFF -DWITH_TIFF=OFF \
  -DWITH_PROTOBUF=OFF -DWITH_ADE=OFF -DWITH_LAPACK=OFF -DWITH_EIGEN=OFF -DBUILD_ZLIB=ON -DBUILD_PNG=ON -DBUILD_JPEG=ON -DBUILD_TESTS=ON -DBUILD_PERF_TESTS=OFF -DBUILD_EXAMPLES=OFF \
  -DBUILD_opencv_apps=OFF -DBUILD_JAVA=OFF -DBUILD_opencv_python3=OFF -DBUILD_opencv_python2=OFF -DOPENCV_GENERATE_SETUPVARS=OFF > $BUILD.configure.log 2>&1
grep -n "Custom HAL\|mi355cv" $BUILD.configure.log | head
grep -n "mi355cv_hal.hpp" $BUILD/custom_hal.hpp
ninja -C $BUILD -j${JOBS:-8} opencv_test_imgproc opencv_test_video > $BUILD.build.log 2>&1
mkdir -p $OUT/lib $OUT/bin
cp -a $BUILD/lib/libopencv_*.so* $OUT/lib/
cp $BUILD/bin/opencv_test_imgproc $BUILD/bin/opencv_test_video $OUT/bin/
cp $BUILD/custom_hal.hpp $OUT/custom_hal.hpp
grep -i "custom hal" $BUILD.configure.log > $OUT/configure_summary.txt || true
strip $OUT/lib/*.so.4.* $OUT/bin/* 2>/dev/null || true
du -sh $OUT
