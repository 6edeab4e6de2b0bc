"""Multi-process (world_size 2, gloo, CPU) test of the N>1 path's plumbing: the single broadcast of
the packed prompt embedding and the image sharding by global index (stable_diffusion_burn_amd/sharding.py,
used verbatim by bench.py with backend "nccl" == RCCL on the GPU box)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stable_diffusion_burn_amd import sharding, synthetic as syn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, global_batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, Tu, C = 7, 2, 64
    packed = torch.zeros((T + Tu) * C)
    if rank == 0:  # only rank 0 has the prompt embedding (CLIP output in the reference)
        p0, _, _ = sharding.pack_prompt(torch.from_numpy(syn.cond_context(0, T, C)), torch.from_numpy(syn.uncond_context(Tu, C)))
        packed.copy_(p0)
    sharding.broadcast_prompt(packed, src=0)
    cond, uncond = sharding.unpack_prompt(packed, T, Tu, C)
    mine = list(sharding.shard_range(global_batch, rank, world))
    lat = np.stack([syn.initial_latent(i, 8, 8) for i in mine]) if mine else np.zeros((0, 4, 8, 8), np.float32)
    # barrier + max-over-ranks timing, as bench.py does
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, mine, cond.numpy().copy(), uncond.numpy().copy(), lat, float(t.item())))
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world, global_batch = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_c, ref_u = syn.cond_context(0, 7, 64), syn.uncond_context(2, 64)
    covered = []
    for rank, mine, cond, uncond, lat, tmax in got:
        assert np.array_equal(cond, ref_c) and np.array_equal(uncond, ref_u)   # every rank has rank 0's prompt
        assert tmax == float(world)                                             # MAX over ranks
        for j, i in enumerate(mine):                                            # noise keyed by GLOBAL index
            assert np.array_equal(lat[j], syn.initial_latent(i, 8, 8))
        covered += mine
    assert covered == list(range(global_batch))


def _share_worker(rank, world, port, tag, q, directory="/dev/shm", fallback_dirs=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def make():
        calls.append(rank)
        return np.random.default_rng(7).standard_normal(100_003).astype(np.float32)

    arr = sharding.share_flat_array(make, rank, world, dist.barrier, tag, directory, fallback_dirs)
    dist.barrier()
    left = [d for d in [directory] + list(fallback_dirs or []) if os.path.exists(os.path.join(d, f"sdmi_{tag}.f32"))]
    q.put((rank, len(calls), float(np.asarray(arr, np.float64).sum()), arr.shape, bool(left), type(arr).__name__))
    dist.destroy_process_group()


def _run_share(directory, fallback_dirs):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = f"test_{os.getpid()}_{abs(hash((directory, tuple(fallback_dirs or ())))) % 10 ** 6}"
    port = _free_port()
    procs = [ctx.Process(target=_share_worker, args=(r, world, port, tag, q, directory, fallback_dirs)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = float(np.random.default_rng(7).standard_normal(100_003).astype(np.float32).astype(np.float64).sum())
    assert all(g[2] == want and g[3] == (100_003,) for g in got)
    assert not any(g[4] for g in got)                       # no file is left behind
    return got


def test_weight_image_falls_back_to_the_next_directory(tmp_path):
    """the first directory does not exist (or is full: same branch): rank 0 writes to the fallback, rank 1 maps it"""
    got = _run_share(str(tmp_path / "no_such_dir"), [str(tmp_path)])
    assert [g[1] for g in got] == [1, 0] and got[1][5] == "memmap"


def test_a_stale_weight_image_of_a_crashed_run_is_never_mapped(tmp_path):
    """round 4's advice: a leftover `sdmi_<tag>.f32` (a crashed earlier run with the same MASTER_PORT; wrong size, wrong content) sits in the first directory.
    Rank 0 removes leftovers before it writes and names what it wrote in a sidecar with the byte count; the other ranks map only that -- every rank ends up with
    THIS run's array and no name is left behind."""
    first, second = tmp_path / "first", tmp_path / "second"
    first.mkdir(); second.mkdir()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = f"stale_{os.getpid()}"
    stale = first / f"sdmi_{tag}.f32"
    np.full(17, 123.0, dtype=np.float32).tofile(stale)            # wrong size, wrong content
    port = _free_port()
    procs = [ctx.Process(target=_share_worker, args=(r, world, port, tag, q, str(first), [str(second)])) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = float(np.random.default_rng(7).standard_normal(100_003).astype(np.float32).astype(np.float64).sum())
    assert all(g[2] == want and g[3] == (100_003,) for g in got)   # every rank holds THIS run's array
    assert not stale.exists() and not any(g[4] for g in got)       # the leftover was removed, nothing new is left behind


def test_weight_image_is_generated_per_rank_when_no_directory_takes_it(tmp_path):
    """a container whose /dev/shm is too small and whose temporary directory is read-only: nobody waits, every rank generates its own copy"""
    got = _run_share(str(tmp_path / "no_such_dir"), [str(tmp_path / "nor_this")])
    assert [g[1] for g in got] == [1, 1] and got[1][5] == "ndarray"


def test_weight_image_is_generated_once_and_mapped_by_the_other_ranks():
    """bench.py --gpus N: rank 0 generates the synthetic weight image, the others map it (sharding.share_flat_array)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = f"test_{os.getpid()}"
    port = _free_port()
    procs = [ctx.Process(target=_share_worker, args=(r, world, port, tag, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = float(np.random.default_rng(7).standard_normal(100_003).astype(np.float32).astype(np.float64).sum())
    assert [g[1] for g in got] == [1, 0]                    # generated on rank 0 only
    assert all(g[2] == want and g[3] == (100_003,) for g in got)
    assert not any(g[4] for g in got)                       # the name is gone once everybody has mapped it
    assert got[1][5] == "memmap"


def test_single_process_broadcast_is_noop():
    p = torch.arange(6, dtype=torch.float32)
    assert sharding.broadcast_prompt(p) is p
    c, u = sharding.unpack_prompt(p, 2, 1, 2)
    assert c.shape == (2, 2) and u.shape == (1, 2)


def test_c_abi_partition_rule_matches_the_launcher():
    """sdmi_sample_image_sharded (one process, one thread per device) and bench.py / sharding.py (one process per device)
    must hand the same global image indices to the same device: sdmi_shard_range == sharding.shard_range."""
    import ctypes as C

    from stable_diffusion_burn_amd import _capi, sharding
    lib = _capi.load_library()
    for n in (0, 1, 3, 7, 8, 16, 64, 65, 128):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                b, e = C.c_int32(), C.c_int32()
                assert lib.sdmi_shard_range(n, r, world, C.byref(b), C.byref(e)) == 0
                assert list(range(b.value, e.value)) == list(sharding.shard_range(n, r, world))
                covered += list(range(b.value, e.value))
            assert covered == list(range(n))
    b, e = C.c_int32(), C.c_int32()
    assert lib.sdmi_shard_range(4, 4, 4, C.byref(b), C.byref(e)) != 0


def test_rank_runner_surfaces_a_failing_rank_after_the_others_have_finished():
    """sdmi_sample_image_sharded runs one host thread per device (csrc/multi_ranks.hpp); a rank that fails must not strand the others or
    leave work in flight behind the caller's back: the library checks (and reports through its status) that every healthy rank ran to
    completion and that the drain hook -- MultiEngine: hipStreamSynchronize on every device -- ran exactly once BEFORE the failing rank's
    error was rethrown.  No device needed (sdmi_selftest_rank_errors)."""
    from stable_diffusion_burn_amd import _capi
    lib = _capi.load_library()
    assert lib.sdmi_selftest_rank_errors(8, -1) == 0                  # nobody fails
    for n, bad in ((1, 0), (2, 1), (8, 0), (8, 5), (8, 7)):
        rc = lib.sdmi_selftest_rank_errors(n, bad)
        msg = lib.sdmi_last_error().decode()
        assert rc == -2, (n, bad, rc)                                  # SDMI_ERR_HIP: the status the failing rank threw
        assert f"rank {bad}: injected failure" in msg, msg             # the failing rank is named, with its own message
    assert lib.sdmi_selftest_rank_errors(0, 0) < 0                    # bad arguments are errors, not crashes
