"""Multi-GPU plumbing of the sampling path: image-batch sharding + the ONE collective.

SURVEY.md 8(e): the path shards naturally over independent images -- GroupNorm and attention
are per sample, DDIM is pointwise, the timestep and the prompt are shared read-only.  So:

  * one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI; "gloo" in the
    CPU tests);
  * rank r owns the contiguous global image indices [r*B/R, (r+1)*B/R); the initial noise of
    image i is keyed by its GLOBAL index, so results do not depend on the GPU count;
  * the only data-path collective is one broadcast of the packed prompt embeddings
    [cond (T x ctx_dim) | uncond (Tu x ctx_dim)] from rank 0 (473 088 bytes at T = Tu = 77) --
    in the reference they come out of CLIP once per prompt (stablediffusion/mod.rs:194-211);
  * u8 images stay on their rank (the caller gathers on the host if it wants them together).

torch is used only for the process group and device buffers (plumbing).
"""
from __future__ import annotations


def shard_range(global_batch: int, rank: int, world: int) -> range:
    """Contiguous block partition of image indices; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def pack_prompt(cond, uncond):
    """[cond | uncond] as one flat float32 buffer + the two row counts (torch tensors)."""
    import torch
    if cond.dim() != 2 or uncond.dim() != 2 or cond.shape[1] != uncond.shape[1]:
        raise ValueError("cond [T,C] and uncond [Tu,C] expected")
    return torch.cat([cond.reshape(-1), uncond.reshape(-1)]).contiguous().float(), cond.shape[0], uncond.shape[0]


def broadcast_prompt(packed, src: int = 0):
    """The single collective of the path.  No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(packed, src=src)
    return packed


def unpack_prompt(packed, t: int, tu: int, ctx_dim: int):
    cond = packed[: t * ctx_dim].reshape(t, ctx_dim)
    uncond = packed[t * ctx_dim: (t + tu) * ctx_dim].reshape(tu, ctx_dim)
    return cond, uncond


def share_flat_array(make, rank: int, world: int, barrier, tag: str, directory: str = "/dev/shm", fallback_dirs=None):
    """One host copy of a large read-only float32 array for all ranks of a node (bench.py: the 3.6 GB synthetic weight image, which costs
    seconds of numpy RNG per rank on shared cores).  Rank 0 calls `make()` and writes the result to `<dir>/sdmi_<tag>.f32` in the first of
    `directory`, then `fallback_dirs` (default: the temporary directory), that exists, is writable and has room; after `barrier()` the other
    ranks map the file rank 0 wrote -- named by its sidecar `<file>.len` (the byte count; leftovers of a crashed run with the same tag are removed by
    rank 0 before it writes, so a stale file is never mapped) -- read-only (np.memmap: the page cache holds ONE copy); after a second barrier rank 0 unlinks the
    names (the mappings stay valid).  If NO candidate takes the file (a container with a 64 MB /dev/shm and a read-only /tmp), every rank finds no file behind the
    barrier and calls `make()` itself -- slower, never wrong, and no rank is left waiting.  world == 1: returns make().  Not part of the data
    path: no collective, only two barriers."""
    import os
    import shutil
    import tempfile
    import numpy as np
    if world == 1:
        return np.ascontiguousarray(make(), dtype=np.float32)
    dirs = [directory] + list(fallback_dirs if fallback_dirs is not None else [tempfile.gettempdir()])
    paths = [os.path.join(d, f"sdmi_{tag}.f32") for d in dirs]
    arr = None
    written = None
    if rank == 0:
        # names left behind by a crashed earlier run (same tag = same MASTER_PORT) are removed FIRST: behind the barrier a rank maps only what this call wrote --
        # the sidecar `<path>.len` names the byte count, and a mapping of any other size is refused
        for path in paths:
            for leftover in (path, path + ".len", path + ".tmp", path + ".len.tmp"):
                try:
                    os.unlink(leftover)
                except OSError:
                    pass
        arr = np.ascontiguousarray(make(), dtype=np.float32)
        for path in paths:
            tmp = path + ".tmp"
            try:
                if shutil.disk_usage(os.path.dirname(path)).free < arr.nbytes + (64 << 20):
                    continue
                fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)   # a fresh file of our own, never through a planted link
                with os.fdopen(fd, "wb") as f:
                    arr.tofile(f)
                os.replace(tmp, path)          # the name appears only once the file is complete
                # the sidecar the same way (own fresh file, no planted link, renamed when complete): a rank that sees `<path>.len` sees a complete data file
                fd = os.open(path + ".len.tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
                with os.fdopen(fd, "w") as f:
                    f.write(str(arr.nbytes))
                os.replace(path + ".len.tmp", path + ".len")
                written = path
                break
            except OSError:
                for name in (tmp, path + ".len.tmp", path + ".len", path):   # nothing of a failed attempt stays behind (the multi-GB data file included)
                    try:
                        os.unlink(name)
                    except OSError:
                        pass
    barrier()
    if rank != 0:
        for path in paths:
            try:
                with open(path + ".len") as f:
                    nbytes = int(f.read().strip())
                have = os.path.getsize(path)
            except (OSError, ValueError):
                continue                       # no sidecar, or a sidecar without its data file: the next directory, then make()
            if have != nbytes:
                raise RuntimeError(f"share_flat_array: {path} has {have} bytes, rank 0 wrote {nbytes}")
            arr = np.memmap(path, dtype=np.float32, mode="r")
            break
        else:
            arr = np.ascontiguousarray(make(), dtype=np.float32)     # rank 0 found no place for the file
    barrier()
    if rank == 0 and written is not None:
        for name in (written, written + ".len"):
            try:
                os.unlink(name)
            except OSError:
                pass
    return arr
