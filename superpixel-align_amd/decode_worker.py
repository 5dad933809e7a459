"""PNG decoding in worker PROCESSES for the labelled driver (input stage of batch_spalign_kmeans.py:486-548:
ResizeImageDataset / ZippedCityscapesRoadDataset decode one PNG per image on the main thread).

Threads scale the decode itself (zlib releases the GIL) but every thread still runs PIL's Python-level chunk loop,
and with 32 of them the main thread — which issues ~400 kernel launches per batch from Python — waits 40-100 ms per
batch for the interpreter lock (measured, HISTORY.md section 5).  Worker processes have their own interpreters: they
decode straight into shared-memory slabs the parent has registered as pinned host memory, so a batch goes from PNG
to the GPU with one asynchronous DMA and no copy in the parent.

Imports nothing heavier than numpy and Pillow (a worker must not pull in torch)."""
import os
import zipfile
from multiprocessing import shared_memory

import numpy as np

_SHM = {}            # name -> SharedMemory (attached once per worker)
_ZIP = {}            # path -> ZipFile (opened once per worker)


def _attach(name):
    shm = _SHM.get(name)
    if shm is None:
        shm = shared_memory.SharedMemory(name=name)
        _SHM[name] = shm
    return shm


def _open(src):
    """src: a path, or (zip path, member name)."""
    if isinstance(src, tuple):
        zf = _ZIP.get(src[0])
        if zf is None:
            zf = zipfile.ZipFile(src[0])
            _ZIP[src[0]] = zf
        return zf.open(src[1])
    return src


def decode_into(task):
    """(shm name, byte offset, (H, W, C) expected or None for a 2-D label image, source) -> the decoded shape.
    The frame is written at the offset when its shape is the expected one; the caller falls back otherwise."""
    from PIL import Image
    name, offset, shape, src = task
    with Image.open(_open(src)) as f:
        a = np.asarray(f, dtype=np.uint8)
    if len(shape) == 2:
        a = a if a.ndim == 2 else a[:, :, 0]
    else:
        if a.ndim == 2:
            a = a[:, :, None]
        a = a[:, :, :3]
    if tuple(a.shape) != tuple(shape):
        return tuple(a.shape)
    dst = np.ndarray(shape, dtype=np.uint8, buffer=_attach(name).buf, offset=offset)
    dst[...] = a
    return tuple(a.shape)


def warm(_):
    """first task of every worker: import Pillow's PNG plugin now, not inside the first batch"""
    from PIL import Image, PngImagePlugin  # noqa: F401
    return os.getpid()
