#!/bin/bash
# Run the reference's own imgproc accuracy tests (oracle/_ref/opencv_test_imgproc_hal, `make -C oracle/ref reftests`) with the hooks served
# by the GPU and summarise: which tests failed, how many hook calls each entry point served.  On the GPU box, from the repo root:
#     bash tools/run_reference_suite.sh [gtest filter, default: everything that needs no image files]
# Output: gpurun_out/reference_suite.log (full), a summary on stdout.  Round 1 ran a 70-test subset in 2.4 s (profiles/r01h_*.log).
FILTER=${1:-*}
NEEDS_DATA='Canny_Modes.*:GaussianBlurVsBitexact.*:GaussianBlur_Bitexact.regression_9863:ImgProc_Bayer2RGBA.*:ImgProc_BayerEdgeAwareDemosaicing.*:Imgproc_AdaptiveThreshold.*:Imgproc_ColorBayer.*:Imgproc_ColorBayerVNG.*:Imgproc_ColorBayerVNG_Strict.*:Imgproc_GoodFeatureToT.accuracy:Imgproc_sepFilter2D.*:Imgproc_sepFilter2D_outTypes.*:Imgproc_sepFilter2D_types.*'
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp
MI355CV_PRINT_COUNTS=1 $REPO/oracle/_ref/opencv_test_imgproc_hal --gtest_color=no --gtest_filter="$FILTER-$NEEDS_DATA" > $REPO/gpurun_out/reference_suite.log 2>&1
grep -E "tests? from .* ran|^\[  PASSED|^\[  FAILED  \] [A-Za-z_0-9/.]+$" $REPO/gpurun_out/reference_suite.log | sort -u | head -80
grep -E "^mi355cv:" $REPO/gpurun_out/reference_suite.log
