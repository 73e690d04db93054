/* modules present in the oracle build */
#define HAVE_OPENCV_CORE
#define HAVE_OPENCV_IMGPROC
