"""Array plumbing shared by the host-side mirror: type codes, borders, buffers."""
import ctypes
import threading
import numpy as np

try:  # torch is plumbing (device memory + streams), not the product
    import torch
except Exception:  # pragma: no cover
    torch = None

from . import _lib

# depth codes (core/hal/interface.h:66-80)
CV_8U, CV_8S, CV_16U, CV_16S, CV_32S, CV_32F, CV_64F = range(7)
# border codes (core/base.hpp:332-345)
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101, BORDER_TRANSPARENT = range(6)
BORDER_REFLECT101 = BORDER_DEFAULT = BORDER_REFLECT_101
BORDER_ISOLATED = 16
# interpolation flags (imgproc.hpp:248-294)
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4, INTER_LINEAR_EXACT, INTER_NEAREST_EXACT = range(7)
WARP_INVERSE_MAP = 16

_NP_DEPTH = {np.dtype(np.uint8): CV_8U, np.dtype(np.int8): CV_8S, np.dtype(np.uint16): CV_16U,
             np.dtype(np.int16): CV_16S, np.dtype(np.int32): CV_32S, np.dtype(np.float32): CV_32F,
             np.dtype(np.float64): CV_64F}
_DEPTH_NP = {v: k for k, v in _NP_DEPTH.items()}
if torch is not None:
    _T_DEPTH = {torch.uint8: CV_8U, torch.int8: CV_8S, torch.int16: CV_16S, torch.int32: CV_32S,
                torch.float32: CV_32F, torch.float64: CV_64F}
    if hasattr(torch, "uint16"):
        _T_DEPTH[torch.uint16] = CV_16U
    _DEPTH_T = {v: k for k, v in _T_DEPTH.items()}
    _T_DEPTH_ESZ = {k: (v, torch.empty(0, dtype=k).element_size()) for k, v in _T_DEPTH.items()}


def CV_MAKETYPE(depth, cn):
    return depth + ((cn - 1) << 3)


def _row_step(stride, rowb, h):
    """Row step in bytes of a view.  A one-row view may carry any stride (numpy / torch report what they like for a length-1 axis): it gets
    at least one row of bytes, like a one-row cv::Mat submatrix keeps its parent's step.  For h > 1 the rows must not overlap or run
    backwards -- negative, zero (broadcast) or short strides would make the hooks read and write outside the buffer."""
    if h <= 1:
        return stride if stride >= rowb else rowb
    if stride < rowb:
        raise ValueError(f"image rows overlap or run backwards (row stride {stride} B < {rowb} B per row): pass a contiguous copy")
    return stride


class Img:
    """A 2-D image view handed to the C ABI: pointer, step (bytes), width, height, depth, channels."""
    __slots__ = ("obj", "ptr", "step", "w", "h", "depth", "cn", "device", "esz")

    def __init__(self, a):
        self.obj = a
        if torch is not None and isinstance(a, torch.Tensor):
            sh, st = a.shape, a.stride()                      # one call each: this constructor runs twice per hook call
            nd = len(sh)
            if nd not in (2, 3):
                raise ValueError("image must be HxW or HxWxC")
            if nd == 3 and st[2] != 1 or nd == 2 and sh[1] > 1 and st[1] != 1:
                raise ValueError("image rows must be dense (channel-interleaved, unit stride)")
            self.h, self.w = sh[0], sh[1]
            self.cn = sh[2] if nd == 3 else 1
            if nd == 3 and self.cn > 1 and st[1] != self.cn:
                raise ValueError("pixels must be contiguous within a row")
            self.depth, self.esz = _T_DEPTH_ESZ[a.dtype]
            # a one-row view keeps its parent's step like a cv::Mat submatrix does (the hooks reach real rows above / below a ROI through it)
            rowb = self.w * self.cn * self.esz
            self.step = _row_step(st[0] * self.esz, rowb, self.h)
            self.ptr = a.data_ptr()
            self.device = a.is_cuda
        else:
            a = np.asarray(a)
            self.obj = a
            if a.ndim not in (2, 3):
                raise ValueError("image must be HxW or HxWxC")
            self.h, self.w = a.shape[0], a.shape[1]
            self.cn = a.shape[2] if a.ndim == 3 else 1
            self.depth = _NP_DEPTH[a.dtype]
            self.esz = a.itemsize
            if a.ndim == 3 and (a.strides[2] != self.esz or (self.cn > 1 and a.strides[1] != self.cn * self.esz)):
                raise ValueError("pixels must be contiguous within a row")
            if a.ndim == 2 and self.w > 1 and a.strides[1] != self.esz:
                raise ValueError("image rows must be dense")
            rowb = self.w * self.cn * self.esz
            self.step = _row_step(a.strides[0], rowb, self.h)
            self.ptr = a.ctypes.data
            self.device = False

    @property
    def type(self):
        return CV_MAKETYPE(self.depth, self.cn)


def empty_like_kind(ref, h, w, cn, depth):
    """Allocate an output of the same kind (torch-cuda / torch-cpu / numpy) as `ref`."""
    shape = (h, w) if cn == 1 and (not hasattr(ref, "ndim") or ref.ndim == 2) else (h, w, cn)
    if torch is not None and isinstance(ref, torch.Tensor):
        return torch.empty(shape, dtype=_DEPTH_T[depth], device=ref.device)
    return np.empty(shape, dtype=_DEPTH_NP[depth])


_raw_stream = getattr(getattr(torch, "_C", None), "_cuda_getCurrentRawStream", None) if torch is not None else None


_tls = threading.local()   # .dev: device ordinal the calling thread's hooks were last bound to (the library's binding is per thread too)


def bind_stream(*imgs):
    """Launch on torch's current stream when the images live on a torch CUDA device."""
    if torch is None:
        return
    for im in imgs:
        if im is not None and im.device:
            dev = im.obj.device
            # the hooks of this thread run on the device that owns the image (one context per thread and device inside the library), on
            # torch's current stream for THAT device; torch's own current device is left alone (the library restores it after each hook)
            if getattr(_tls, "dev", None) != dev.index:
                if _lib.lib.mi355cv_setDevice(dev.index) != 0:
                    raise RuntimeError("mi355cv_setDevice(%d): %s" % (dev.index, _lib.lib.mi355cv_lastError().decode()))
                _tls.dev = dev.index
            # the raw handle of torch's current stream (the public route builds a Stream object per call)
            h = _raw_stream(dev.index) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream
            _lib.lib.mi355cv_setStream(ctypes.c_void_p(h))
            return
    _lib.lib.mi355cv_resetStream()


def set_async(enable: bool):
    """Device-resident calls return after enqueue on the bound stream (caller synchronises)."""
    _lib.lib.mi355cv_setAsync(1 if enable else 0)


def synchronize():
    _lib.lib.mi355cv_synchronize()
