"""Batch sharding across the GPUs of one node.

Frames are independent units: a batch of B frames is split by index, one process per GPU (launched by
`torch.distributed.run`), no collective on the data path.  The only collective is a broadcast of the shared, tiny
parameters (filter taps, warp matrices, the template) from rank 0 at plan time -- RCCL over xGMI on GPUs, gloo on CPU.
"""
import os

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def world():
    """(rank, world_size, local_rank) from the torchrun environment (single process when absent)"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def frame_range(nframes, rank, world_size):
    """frames [lo, hi) owned by `rank`: contiguous blocks, sizes differ by at most one (SURVEY.md §8e partitioning)"""
    base, rem = divmod(nframes, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def spawn_ranks(n, script, argv, need_gpus=True, port=None):
    """`python <script> --gpus N` without a launcher: run the same command line as N ranks -- one process per GPU -- under
    torch.distributed.run on this node (rendezvous on 127.0.0.1) and return its exit code.  With need_gpus the request is refused when fewer
    than N devices are visible: a multi-GPU number is never reported from fewer GPUs than it names."""
    import subprocess
    import sys
    if need_gpus:
        have = torch.cuda.device_count() if torch is not None and torch.cuda.is_available() else 0
        if have < n:
            sys.stderr.write(f"{os.path.basename(script)}: --gpus {n} requested but only {have} GPU(s) are visible -- refusing to report a "
                             "multi-GPU number from fewer devices\n")
            return 2
    port = port or os.environ.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(script)] + list(argv)
    return subprocess.call(cmd)


def init(backend=None):
    """initialise torch.distributed when launched with WORLD_SIZE > 1 (nccl == RCCL on ROCm, gloo on CPU)"""
    rank, ws, local = world()
    if ws == 1 or dist is None:
        return rank, ws, local
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=ws)
    return rank, ws, local


def broadcast_params(arr, src=0, device=None):
    """plan-time broadcast of a small numpy parameter block from `src` to every rank; returns the array"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(arr)
    rank = dist.get_rank()
    a = np.ascontiguousarray(arr)
    meta = torch.tensor([a.ndim] + list(a.shape) + [0] * (8 - a.ndim - 1) if rank == src else [0] * 8, dtype=torch.int64)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    meta = meta.to(dev)
    dist.broadcast(meta, src)
    nd = int(meta[0]); shape = [int(v) for v in meta[1:1 + nd]]
    buf = torch.from_numpy(a.astype(np.float64)).to(dev) if rank == src else torch.empty(shape, dtype=torch.float64, device=dev)
    dist.broadcast(buf, src)
    return buf.cpu().numpy()


def barrier():
    if dist is not None and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX-reduce one float over ranks (the timing reduction of bench.py)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_counts(n, device=None):
    """SUM-reduce an integer (frames processed) over ranks"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(n)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    t = torch.tensor([int(n)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t[0])


def gather_ragged(rows, dst=0, device=None, dtype=None):
    """Egress of a detector's output when the frames are sharded (SURVEY.md section 8e + f3): every rank holds one variable-length record array per frame it owns
    (keypoints, descriptors: the number of rows is the GPU's decision), rank `dst` wants all of them in frame order.  `rows`: a list of 2-D uint8-viewable
    numpy arrays of the same row width (one per owned frame, in frame order; ranks own contiguous frame blocks -- frame_range).  One all_gather of the
    per-frame row counts, one padded all_gather of the bytes: two collectives, off the data path of the kernels.  Returns the list of all frames' arrays on
    `dst`, None elsewhere (single process: the input).  1-D arrays of records (a structured dtype such as the keypoint records of ORB.detectAndCompute) are taken
    as n rows of one record and come back 1-D.  `dtype`: the element type of the output; needed only where `dst` may own no frame (it cannot read it off its own list)."""
    arrs = [np.ascontiguousarray(a) for a in rows]
    one_d = bool(arrs) and arrs[0].ndim == 1
    if dtype is None and arrs:
        dtype = arrs[0].dtype
    dtype = np.dtype(dtype) if dtype is not None else None
    one_d = one_d or (dtype is not None and dtype.names is not None)         # records travel as rows of one record
    arrs = [a.reshape(-1, 1) if a.ndim == 1 else a for a in arrs]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [a.reshape(-1) for a in arrs] if one_d else arrs
    ws, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    width = 0
    for a in arrs:
        if a.ndim != 2:
            raise ValueError("gather_ragged: 2-D record arrays expected")
        width = max(width, a.shape[1] * a.itemsize)
    # 1. how many frames and rows everybody has (frames per rank differ by at most one: pad the count vectors to the maximum)
    meta = torch.tensor([len(arrs), width], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(ws)]
    dist.all_gather(metas, meta)
    nfr = [int(m[0]) for m in metas]
    width = max(int(m[1]) for m in metas)
    maxf = max(nfr) if nfr else 0
    cnt = torch.zeros(max(maxf, 1), dtype=torch.int64, device=dev)
    for i, a in enumerate(arrs):
        cnt[i] = a.shape[0]
    cnts = [torch.zeros_like(cnt) for _ in range(ws)]
    dist.all_gather(cnts, cnt)
    totals = [int(c[:n].sum()) for c, n in zip(cnts, nfr)]
    # 2. the bytes, padded to the largest rank's total
    maxb = max(max(totals) * width, 1)
    mine = np.zeros(maxb, np.uint8)
    if arrs and totals[rank]:
        flat = np.concatenate([a.view(np.uint8).reshape(a.shape[0], -1) for a in arrs if a.shape[0]], axis=0)
        mine[:flat.size] = flat.reshape(-1)
    buf = torch.from_numpy(mine).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(ws)]
    dist.all_gather(bufs, buf)
    if rank != dst:
        return None
    out = []
    for r in range(ws):
        raw = bufs[r].cpu().numpy()
        off = 0
        for f in range(nfr[r]):
            n = int(cnts[r][f])
            block = raw[off:off + n * width].reshape(n, width)
            off += n * width
            rec = block.view(dtype).copy() if dtype is not None and dtype.itemsize > 1 and width % dtype.itemsize == 0 else block.copy()
            out.append(rec.reshape(-1) if one_d else rec)
    return out
