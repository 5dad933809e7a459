"""Image-range data parallelism of the label-generation path, one process per GPU.

The reference fans out with bash: `step = n_data / N_GPUS + 1`, rank r takes images
[r*step, min((r+1)*step, n_data)), each process runs the batch loop on its range and all of
them append to one result.json (utils/create_random300_labels.sh:37-51,
batch_spalign_kmeans.py:407-422, :538-544).  Images are independent units (k-means is joint
only inside a batch), so the native equivalent keeps the partition, exchanges nothing during
the run, and replaces the shared-file append by ONE all_gather of fixed-size per-image records
at the end (torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests); rank 0 then writes result.json in index order.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

# one record per processed image: index, TN, FP, FN, TP, n_superpixels, kmeans_iters, kmeans_status,
# then the stage timers of its batch (result.json's time_* keys, batch_spalign_kmeans.py:428-458, and
# elapsed_time, :421) as float64 bit patterns carried in the same int64 row, so that the lines rank 0
# writes for images of other ranks have the same schema as its own
RECORD_FIELDS = ('index', 'TN', 'FP', 'FN', 'TP', 'n_superpixels', 'kmeans_iters', 'kmeans_status')
TIMER_FIELDS = ('time_superpixel', 'time_roialign', 'time_prior', 'time_kmeans', 'time_feature_maps',
                'elapsed_time', 'gpu', 'time_device')      # time_device: the batch's own device time (cli.py), every rank's lines carry it
RECORD_WIDTH = len(RECORD_FIELDS) + len(TIMER_FIELDS)


def pack_timers(times):
    """{timer key: seconds} -> len(TIMER_FIELDS) int64 words (float64 bit patterns; missing = NaN)."""
    v = np.array([float(times.get(k, np.nan)) for k in TIMER_FIELDS], np.float64)
    return v.view(np.int64).tolist()


def unpack_timers(row):
    """Inverse of pack_timers on one gathered record row; timers that were not recorded are dropped."""
    v = np.asarray(row[len(RECORD_FIELDS):RECORD_WIDTH], np.int64).view(np.float64)
    return {k: float(x) for k, x in zip(TIMER_FIELDS, v) if not np.isnan(x)}


def world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process otherwise)."""
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)),
            int(os.environ.get('LOCAL_RANK', 0)))


def init(backend=None):
    rank, ws, local = world()
    # SPA_DIST_FORCE=1: create the process group (and run every collective) with ONE rank too — the RCCL path of a single-GPU box
    force = os.environ.get('SPA_DIST_FORCE') == '1'
    if (ws > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # SPA_DIST_BACKEND=gloo lets two ranks share ONE GPU (tests of the N > 1 code path on
            # a single-GPU box); production uses nccl (= RCCL over xGMI)
            backend = os.environ.get('SPA_DIST_BACKEND') or \
                ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            # bind this process to its GPU and create its spa_ctx BEFORE the process group exists:
            # RCCL then initialises on the device the label kernels already use, and no rank ever
            # touches device 0 by accident (SPA_BENCH_SAME_DEVICE: several ranks share GPU 0 in tests)
            # (ranks isolated by device visibility see one device each: ordinal 0)
            dev = 0 if os.environ.get('SPA_BENCH_SAME_DEVICE') == '1' else local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(dev)
            from .engine import default_engine
            default_engine()
        kw = {}
        if backend == 'nccl' and torch.cuda.is_available():
            kw['device_id'] = torch.device('cuda', torch.cuda.current_device())     # RCCL binds to this rank's GPU, no guessing from the rank
        dist.init_process_group(backend, rank=rank, world_size=ws, **kw)
        import atexit
        atexit.register(lambda: dist.is_initialized() and dist.destroy_process_group())
    return rank, ws, local


def shard_range(n_data, n_shards, rank, balanced=False):
    """[start, end) of `rank`.  Default = the launcher's rule (step = n_data // n_shards + 1,
    later ranks may get fewer or no images); balanced=True spreads the remainder instead."""
    if balanced:
        base, rem = divmod(n_data, n_shards)
        start = rank * base + min(rank, rem)
        return start, start + base + (1 if rank < rem else 0)
    step = n_data // n_shards + 1
    start = min(n_data, rank * step)
    return start, min(n_data, start + step)


def batch_ranges(start, end, batchsize):
    """The reference batch loop (batch_spalign_kmeans.py:538-544): the last batch is shifted
    back to keep the batch size, so it overlaps its predecessor (and may start below `start`,
    even below 0 -> python slicing semantics, when the range is shorter than one batch)."""
    out = []
    for i in range(start, end, batchsize):
        if i + batchsize >= end:
            out.append((end - batchsize, end))
        else:
            out.append((i, i + batchsize))
    return out


def gather_records(records, device=None):
    """records: (n, RECORD_WIDTH) int64 array of this rank -> concatenation over ranks, on every
    rank (rows of rank 0 first).  One all_gather of the counts + one of the padded payload."""
    rec = np.asarray(records, dtype=np.int64).reshape(-1, RECORD_WIDTH)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and os.environ.get('SPA_DIST_FORCE') != '1'):
        return rec
    ws = dist.get_world_size()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) \
            if dist.get_backend() == 'nccl' else torch.device('cpu')
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    pad = torch.zeros((cap, RECORD_WIDTH), dtype=torch.int64, device=device)
    pad[:rec.shape[0]] = torch.from_numpy(rec).to(device)
    parts = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def all_values(value, device=None):
    """One float per rank -> the list over ranks, on every rank (bench.py: per-rank rates and gather times)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and os.environ.get('SPA_DIST_FORCE') != '1'):
        return [float(value)]
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) \
            if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def max_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and os.environ.get('SPA_DIST_FORCE') != '1'):
        return float(value)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) \
            if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
