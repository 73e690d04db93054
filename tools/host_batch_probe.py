"""PCIe-inclusive rate of cv::GaussianBlur 5x5 on 4K CV_8UC1 frames that live in page-locked host memory: per-frame hook calls (stage in, filter, stage out,
synchronise) vs the pipelined batch entry (chunks through two sets of device buffers, upload / filter / download overlapped)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
N = int(os.environ.get("N", 64))
host = torch.randint(0, 256, (N, 2160, 3840), dtype=torch.uint8).pin_memory()
out = torch.empty_like(host).pin_memory()
for _ in range(2):
    cv.GaussianBlurBatch(host, 5, dst=out)
t0 = time.perf_counter(); cv.GaussianBlurBatch(host, 5, dst=out); t1 = time.perf_counter()
npv, npo = host.numpy(), out.numpy()
for f in range(2):
    cv.GaussianBlur(npv[f], (5, 5), 0, dst=npo[f])
t2 = time.perf_counter()
for f in range(N):
    cv.GaussianBlur(npv[f], (5, 5), 0, dst=npo[f])
t3 = time.perf_counter()
px = N * 3840 * 2160
print(f"pipelined batch entry, {N} x 4K 8UC1 from page-locked host memory: {(t1 - t0) / N * 1e6:7.1f} us / frame = {px / (t1 - t0) / 1e9:6.1f} Gpix/s = {2 * px / (t1 - t0) / 1e9:6.1f} GB/s over PCIe (both ways)")
print(f"per-frame hook calls (stage in, kernel, stage out, sync):          {(t3 - t2) / N * 1e6:7.1f} us / frame = {px / (t3 - t2) / 1e9:6.1f} Gpix/s")
