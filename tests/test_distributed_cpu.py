"""world_size=2 gloo test of the multi-GPU split: contiguous byte-balanced ranges,
no data-path collective, one all_gather of the int32 size vector.  The per-rank "work" here is a
stand-in (sizes derived from the lengths) because kernels cannot run without a GPU; the real
per-rank work is exercised by the gpu tests and bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from k4os.compression.lz4_amd.sharding import byte_balanced_ranges, gather_sizes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lens, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ranges = byte_balanced_ranges(lens, world)
    lo, hi = ranges[rank]
    local = torch.tensor([(int(l) * 7 + i) % 100003 for i, l in zip(range(lo, hi), lens[lo:hi])], dtype=torch.int32)
    full = gather_sizes(local, ranges)
    q.put((rank, full.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_size_vector_gather_gloo(world):
    rng = np.random.default_rng(9)
    lens = np.exp(rng.uniform(np.log(1024), np.log(1 << 20), 301)).astype(np.int64)
    want = np.array([(int(l) * 7 + i) % 100003 for i, l in enumerate(lens)], dtype=np.int32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, full in got:
        assert np.array_equal(full, want)


# ---- the one-batch driver (bench.py --strong) with the kernels' emulator build standing in for the GPU ----------------
class _EmuPickleBackend:
    """per-rank work of sharding.sharded_pickle_roundtrip on the CPU: the product's pickle / unpickle KERNEL SOURCE compiled
    against the wave emulator (tests/emu).  Test infrastructure; the product's backend is device.DevicePickleBackend."""

    def __init__(self):
        from emu_lib import Emu
        self.emu = Emu()

    def pickle_unpickle(self, data, off, lens):
        import time
        from emu_lib import arena
        n = lens.size
        if n == 0:
            return torch.zeros(0, dtype=torch.int32), 0.0, 0.0, True
        src = np.ascontiguousarray(data)
        env, eoff, ecap = arena([int(l) + 5 for l in lens])
        t = time.perf_counter()
        plen = self.emu.pickle_batch(src, off, lens, env, eoff, ecap, threads=2)
        t_p = time.perf_counter() - t
        back, boff, bcap = arena([int(l) for l in lens])
        t = time.perf_counter()
        ulen = self.emu.unpickle_batch(env, eoff, plen, back, boff, bcap, threads=2)
        t_u = time.perf_counter() - t
        ok = bool((ulen == lens).all()) and all(
            back[int(boff[i]):int(boff[i]) + int(lens[i])].tobytes() == src[int(off[i]):int(off[i]) + int(lens[i])].tobytes() for i in range(n))
        return torch.from_numpy(plen.astype(np.int32)), t_p, t_u, ok


def _strong_worker(rank, world, port, lens_all, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from k4os.compression.lz4_amd.sharding import sharded_pickle_roundtrip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ranges, sizes, mine = sharded_pickle_roundtrip(_EmuPickleBackend(), lens_all, rank, world)
    q.put((rank, ranges, sizes.numpy().copy(), mine))
    dist.barrier()
    dist.destroy_process_group()


def test_one_batch_over_two_ranks_matches_a_single_rank_run(oracle):
    """the code path of `bench.py --gpus N --strong`: every rank takes its byte-balanced range of ONE batch, only the size
    vector travels; the gathered vector equals what one rank alone (here: the oracle over the whole batch) produces"""
    from k4os.compression.lz4_amd import corpus
    lens_all = corpus.config4_lengths(48, hi=48 << 10)
    want = np.array([len(oracle.pickle(corpus.config4_share(lens_all, i, i + 1)[0])) for i in range(lens_all.size)], np.int32)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, lens_all, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    covered = 0
    for rank, ranges, sizes, mine in got:
        assert np.array_equal(sizes, want), rank
        assert mine["roundtrip_ok"] and mine["messages"] == ranges[rank][1] - ranges[rank][0]
        covered += mine["messages"]
    assert covered == lens_all.size
    b = [int(lens_all[lo:hi].sum()) for lo, hi in got[0][1]]
    assert abs(b[0] - b[1]) <= int(lens_all.max())          # byte-balanced, not count-balanced
