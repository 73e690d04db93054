/* no installed data files in the oracle build */
#define OPENCV_INSTALL_PREFIX "/nonexistent"
#define OPENCV_DATA_INSTALL_PATH "share/opencv4"
#define OPENCV_BUILD_DIR "/nonexistent"
#define OPENCV_DATA_BUILD_DIR_SEARCH_PATHS ""
#define OPENCV_INSTALL_DATA_DIR_RELATIVE "../share/opencv4"
