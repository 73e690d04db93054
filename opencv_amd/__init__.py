"""opencv_amd -- MI355X-native imgproc hot path behind OpenCV's own API names.

Host-side mirror (Python, like the reference's cv2 binding) of the reference's
dispatcher layer for the hot path (modules/imgproc/src/*.dispatch.cpp): argument
checks, dst allocation and the choice of hook follow the reference; the pixels
are produced by hand-written HIP kernels in libmi355cv.so through the C ABI of
include/mi355cv.h (the cv_hal_* replacement boundary).

Images are H x W[ x C] arrays: `torch.Tensor` on a CUDA(ROCm) device (processed in
place in HBM on torch's current stream) or `numpy.ndarray` / CPU tensors (staged
through HBM by the library).  The result has the same kind as the input.
"""
from . import _lib
from ._lib import call_count, limit, Mi355cvError  # noqa: F401
from .core import *  # noqa: F401,F403
from .imgproc import *  # noqa: F401,F403
from .video import *  # noqa: F401,F403
from .features2d import *  # noqa: F401,F403

__version__ = "0.1"
