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
