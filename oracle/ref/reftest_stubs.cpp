// reftest_stubs.cpp -- TEST INFRASTRUCTURE ONLY.  The reference's own imgproc test sources (built by `make reftests` into
// oracle/_ref/opencv_test_imgproc_hal) reference a few imgcodecs / highgui entry points; the oracle build has neither module, so they get
// inert stand-ins here.  Tests that need image files (opencv_extra, not available) are excluded on the command line.
#include "opencv2/core.hpp"
#include <string>
#include <vector>

namespace cv {
Mat imread(const String&, int) { return Mat(); }
bool imwrite(const String&, InputArray, const std::vector<int>&) { return false; }
void namedWindow(const String&, int) {}
void imshow(const String&, InputArray) {}
int waitKey(int) { return -1; }
}
