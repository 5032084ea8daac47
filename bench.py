#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on MI355X: GiB/s encode+decode on batched 64 KiB blocks.

Workload (config.workload): BASELINE.json configs[1] -- 4096 independent 64 KiB blocks of the
12-class "Silesia-like" mix (corpus.silesia_like_blocks, seed = 2 + rank), L00_FAST encode then
decode, inputs and outputs resident in HBM before the timed region.  A step = one encode pass +
one decode pass over the batch.  `value` = uncompressed bytes through the round trip per second:
N_gpus * sum(U) / max-over-ranks(step time), in GiB/s.  Multi-GPU: one process per GPU, every
rank works on its own 4096 blocks (weak scaling), no data-path collective; the int32 size vector
is all-gathered once outside the timed region.

Extra objects on the JSON line:
  roofline      dominant kernel (the encoder: k4_parse_kernel, which decides the sequences and writes the blocks
                out) -- algorithmic bytes per launch (sum(U)+sum(C)+12N) / median launch duration measured
                with HIP events on the launch stream
  roofline_decode  the same for the decode kernel (the BASELINE.json 50%-of-HBM target)
  cpu_baseline  kind "reference": the reference's own LL64 engine files compiled here (oracle/_ref, built by
                oracle/make_ref.py + g++) on the host cores, same blocks, T threads; the C restatement
                (oracle/k4lz4_oracle.c, "port") and liblz4 are kept as columns beside it
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s HBM3E


def strong(args):
    """ONE batch sharded over the ranks (SURVEY.md 8e / BASELINE.json configs[3]): rank r pickles + unpickles the byte-balanced
    range r of the --messages batch in its own HBM; the int32 envelope-size vector is all-gathered over RCCL (the only
    collective); rank 0 checks the gathered vector against the oracle on a sample drawn from EVERY rank's range and -- batches
    of up to 8 GiB -- the WHOLE vector against a single-rank run of the same batch.  Also printed: the critical path (the
    longest message of a range alone), every rank's busy time, and the time more ranks can at best make of the batch."""
    import torch
    import torch.distributed as dist
    from k4os.compression.lz4_amd import corpus
    from k4os.compression.lz4_amd.device import DevicePickleBackend
    from k4os.compression.lz4_amd.sharding import sharded_pickle_roundtrip

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # K4LZ4_RANK_DEVICE=0 puts every rank on GPU 0: the real multi-rank backend (a context per rank, the gathered size vector)
    # on a one-GPU box.  RCCL refuses two ranks on one device, so the size vector then travels over gloo.
    shared = os.environ.get("K4LZ4_RANK_DEVICE")
    device = int(shared) if shared is not None else local_rank
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1 and shared is None:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        size_backend = "nccl (RCCL all_gather)"
    else:
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        size_backend = "gloo" + (f" ({world} ranks share GPU {device})" if world > 1 else "")
    lens_all = corpus.config4_lengths(args.messages)
    backend = DevicePickleBackend(device)
    ranges, sizes, mine = sharded_pickle_roundtrip(backend, lens_all, rank, world)
    on = backend.dc.device if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([mine["pickle_s"], mine["unpickle_s"], 0.0 if mine["roundtrip_ok"] else 1.0, mine["longest_s"]], dtype=torch.float64, device=on)
    per_rank = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(per_rank, t)
    per_rank = [[float(v) for v in p.tolist()] for p in per_rank]
    t_p, t_u = max(p[0] for p in per_rank), max(p[1] for p in per_rank)
    bad = max(p[2] for p in per_rank)
    if rank == 0:
        from oracle_lib import Oracle
        oracle = Oracle()
        sizes_h = sizes.cpu().numpy()
        total = int(np.asarray(lens_all, dtype=np.int64).sum())
        rng = np.random.default_rng(1)
        sample = sorted({int(i) for lo, hi in ranges if hi > lo for i in rng.integers(lo, hi, size=8)})
        equal = all(len(oracle.pickle(corpus.config4_share(lens_all, i, i + 1)[0])) == int(sizes_h[i]) for i in sample)
        busy = [round((p[0] + p[1]) * 1e3, 3) for p in per_rank]
        crit = max(p[3] for p in per_rank)
        check = {"size_vector_sample_equals_oracle": equal, "sampled_messages": len(sample)}
        if total <= (8 << 30):
            # ... and the WHOLE vector against one rank doing the whole batch alone (same GPU, a fresh context)
            data, off, lens = corpus.config4_share(np.asarray(lens_all), 0, len(lens_all))
            whole, _, _, _ = DevicePickleBackend(device).pickle_unpickle(data, off, lens)
            check["size_vector_equals_single_rank_run"] = bool(np.array_equal(whole.cpu().numpy(), sizes_h))
        print(json.dumps({
            "metric": "GiB/s LZ4Pickler.Pickle + Unpickle over ONE batch of variable-length messages, byte-balanced over the GPUs",
            "value": round(total / 2 ** 30 / (t_p + t_u), 3), "unit": "GiB/s", "n_gpus": world, "steps": 1, "warmup": 1,
            "ms_per_step": round((t_p + t_u) * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[3]: {args.messages} messages 1 KiB-4 MiB (log-uniform), random / text alternating, "
                                   f"{total} bytes, split into {world} contiguous byte-balanced ranges", "messages": args.messages,
                       "bytes_per_rank": [int(np.asarray(lens_all[lo:hi], dtype=np.int64).sum()) for lo, hi in ranges],
                       "pickle_GiBs": round(total / 2 ** 30 / t_p, 3), "unpickle_GiBs": round(total / 2 ** 30 / t_u, 3),
                       "envelope_bytes_total": int(sizes_h.astype(np.int64).sum()), "size_vector_over": size_backend},
            # a ragged batch is as slow as its longest message on its single wavefront: that message alone, each rank's own
            # pickle + unpickle time, and what more ranks can at best make of it
            "critical_path_ms": round(crit * 1e3, 3), "rank_busy_ms": busy,
            "predicted_ms_at_ranks": {str(k): round(max(crit, sum(busy) / 1e3 / k) * 1e3, 3) for k in (1, 2, 4, 8)},
            "roundtrip_ok_all_ranks": bad == 0.0, **check}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=4096)
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--strong", action="store_true",
                    help="ONE configs[3] batch (LZ4Pickler over --messages variable-length messages) split over the ranks by bytes; "
                         "not the headline metric, reported as its own line")
    ap.add_argument("--messages", type=int, default=100000)
    args = ap.parse_args()
    if args.strong:
        return strong(args)

    import torch
    import torch.distributed as dist
    from k4os.compression.lz4_amd import LZ4Codec, corpus
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    from k4os.compression.lz4_amd.sharding import gather_sizes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # K4LZ4_RANK_DEVICE=0 puts every rank on GPU 0 (the headline path with world > 1 on a one-GPU box, tests/test_gpu_configs_full.py):
    # RCCL refuses two ranks on one device, so the barrier / max / size vector then travel over gloo
    shared = os.environ.get("K4LZ4_RANK_DEVICE")
    device = int(shared) if shared is not None else local_rank
    torch.cuda.set_device(device)
    on = torch.device("cuda", device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared is None:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=on)
        else:
            os.environ.setdefault("MASTER_PORT", "29534")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            on = torch.device("cpu")

    n, bs = args.blocks, args.block_size
    corpus_dir = os.environ.get("K4LZ4_CORPUS_DIR")
    blocks = corpus.silesia_blocks(n, bs, corpus_dir) if corpus_dir and os.path.isdir(corpus_dir) else None
    data_kind = "silesia" if blocks is not None else "synthetic"
    if blocks is None:
        blocks = corpus.silesia_like_blocks(n, bs, seed=2 + rank)
    elif rank:
        blocks = np.roll(blocks, -rank * 97, axis=0)       # every rank its own order of the same corpus
    dc = DeviceCodec(device)
    lens = np.full(n, bs, np.int32)
    off = np.arange(n, dtype=np.uint64) * bs
    bound = LZ4Codec.MaximumOutputSize(bs)
    src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, dc.device)
    comp = DeviceBatch.empty_slots(np.full(n, bound), dc.device)
    back = DeviceBatch.empty_slots(lens, dc.device)
    clen = dc.new_out_len(n)
    dlen = dc.new_out_len(n)
    comp_src = DeviceBatch(comp.data, comp.off, clen)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        dc.encode(src, comp, clen)
        if ev is not None:
            ev[1].record()
        dc.decode(comp_src, back, dlen)
        if ev is not None:
            ev[2].record()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])          # events sit on the launch stream (torch's current stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    enc_ms = np.array([e[0].elapsed_time(e[1]) for e in events])
    dec_ms = np.array([e[1].elapsed_time(e[2]) for e in events])
    clen_h = clen.cpu().numpy().astype(np.int64)
    dlen_h = dlen.cpu().numpy()
    sum_u, sum_c = int(n) * bs, int(clen_h.sum())
    ok_roundtrip = bool((dlen_h == bs).all()) and bool(torch.equal(back.data[:n * bs], src.data[:n * bs]))

    # the only collective: the size vector (RCCL all_gather), outside the timed region
    if world > 1:
        ranges = [(r * n, (r + 1) * n) for r in range(world)]
        all_sizes = gather_sizes(clen, ranges)
        total_c = int(all_sizes.sum().item())
    else:
        total_c = sum_c

    # Bit-exactness is every rank's to show for its OWN blocks: each rank encodes its blocks with the oracle (and with the compiled
    # reference where oracle/_ref is built), compares every block's bytes with what its GPU wrote, and the verdicts are reduced
    # with MIN -- `bit_exact` in the line is all ranks' data, not rank 0's share.  (K4LZ4_TEST_CORRUPT_RANK=r: rank r flips one
    # byte of its copy of the GPU's output first -- the test that the reduction really carries a failure through.)
    from k4os.compression.lz4_amd import make_arena
    src_h = blocks.reshape(-1)
    caps = np.full(n, bound, np.int32)
    # verification runs on every rank at once: each takes its share of the host's cores (the CPU baseline, rank 0's alone, is timed
    # afterwards on all of them while the other ranks wait at a barrier -- VERDICT round 5, item 7)
    all_threads = os.cpu_count() or 1
    threads = max(1, all_threads // world)
    oracle = None
    my_exact, my_ref_equal = None, None
    ref_dst = ref_off = ref_len = comp_h = coff_h = None
    t_enc = []
    if not args.no_verify or (rank == 0 and not args.no_cpu_baseline):
        from oracle_lib import Oracle
        oracle = Oracle()
        ref_dst, ref_off = make_arena(caps)
        for _ in range(2):
            t = time.perf_counter()
            ref_len = oracle.encode_batch(src_h, off, lens, ref_dst, ref_off, caps, threads=threads)
            t_enc.append(time.perf_counter() - t)
        comp_h = comp.data.cpu().numpy()
        coff_h = comp.off.cpu().numpy()
        if os.environ.get("K4LZ4_TEST_CORRUPT_RANK") == str(rank):
            comp_h = comp_h.copy()
            comp_h[int(coff_h[n // 2]) + 7] ^= 0x20

        def same_bytes(e_len, e_dst):
            return bool(np.array_equal(e_len, clen_h.astype(np.int32))) and all(
                np.array_equal(comp_h[coff_h[i]:coff_h[i] + clen_h[i]], e_dst[int(ref_off[i]):int(ref_off[i]) + int(e_len[i])]) for i in range(n))

        my_exact = same_bytes(ref_len, ref_dst) and ok_roundtrip
        if not args.no_verify:
            try:
                from oracle_lib import RefEngine
                ref_engine = RefEngine()
                ref2_dst, _ = make_arena(caps)
                ref2_len = ref_engine.encode_batch(src_h, off, lens, ref2_dst, ref_off, caps, threads=threads)
                my_ref_equal = same_bytes(ref2_len, ref2_dst)
            except FileNotFoundError:
                my_ref_equal = None
    all_exact, all_ref_equal, all_roundtrip = my_exact, my_ref_equal, ok_roundtrip
    if world > 1:
        # -1 = this rank did not check; the line's flag is true only when every rank checked and every rank's check held
        v = torch.tensor([-1 if my_exact is None else int(my_exact), -1 if my_ref_equal is None else int(my_ref_equal), int(ok_roundtrip)],
                         dtype=torch.int32, device=on)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        all_exact = None if int(v[0]) < 0 else bool(int(v[0]))
        all_ref_equal = None if int(v[1]) < 0 else bool(int(v[1]))
        all_roundtrip = bool(int(v[2]))

    if world > 1:
        dist.barrier()                       # every rank's verification is over: the host is idle for rank 0's baseline
    result = None
    if rank == 0:
        bit_exact = all_exact
        cpu = None
        threads = all_threads
        if oracle is not None:
            if not args.no_cpu_baseline:
                out_h, out_off = make_arena(lens)
                gib = sum_u / 2 ** 30
                k1 = min(n, 256)

                def timed(engine, label):
                    """encode + decode of the whole batch on all host threads (best of 2) and of 256 blocks on one thread"""
                    te, td = [], []
                    for _ in range(2):
                        t = time.perf_counter(); el = engine.encode_batch(src_h, off, lens, ref_dst, ref_off, caps, threads=threads); te.append(time.perf_counter() - t)
                    for _ in range(2):
                        t = time.perf_counter(); engine.decode_batch(ref_dst, ref_off, el, out_h, out_off, lens, threads=threads); td.append(time.perf_counter() - t)
                    t1e = t1d = 1e9
                    for _ in range(2):
                        t = time.perf_counter(); engine.encode_batch(src_h, off[:k1], lens[:k1], ref_dst, ref_off[:k1], caps[:k1], threads=1); t1e = min(t1e, time.perf_counter() - t)
                        t = time.perf_counter(); engine.decode_batch(ref_dst, ref_off[:k1], el[:k1], out_h, out_off[:k1], lens[:k1], threads=1); t1d = min(t1d, time.perf_counter() - t)
                    return {"value": round(gib / (min(te) + min(td)), 3), "encode_GiBs": round(gib / min(te), 3), "decode_GiBs": round(gib / min(td), 3),
                            "one_thread_encode_GiBs": round(k1 * bs / 2 ** 30 / t1e, 3), "one_thread_decode_GiBs": round(k1 * bs / 2 ** 30 / t1d, 3),
                            "what": label}

                port = timed(oracle, "oracle/k4lz4_oracle.c: C restatement of the reference's LL64 engine, gcc -O2")
                ref_cpu = None
                try:
                    from oracle_lib import RefEngine
                    ref_engine = RefEngine()
                    ref_cpu = timed(ref_engine, "oracle/_ref/libk4ref.so: the reference's own LL64 engine files (Engine/x64/LL64.fast.cs, LL64.dec.cs) "
                                                "respelled as C++ by oracle/make_ref.py, g++ -O2 -fwrapv")
                    # its bytes against the GPU's, block by block, on EVERY rank (reduced above; the restatement's likewise)
                    ref_cpu["equals_gpu_bytes"] = all_ref_equal
                except FileNotFoundError:
                    pass
                # "as sanity, liblz4.so.1" -- one thread (ctypes calls hold no batch entry point), 256 blocks
                lz = None
                try:
                    from oracle_lib import SystemLZ4
                    sl = SystemLZ4()
                    if sl.available:
                        import ctypes as C
                        u8p = C.POINTER(C.c_uint8)
                        lz_dst = np.empty((k1, bound), np.uint8)           # buffers made beforehand: the loop below is the calls, nothing else
                        lz_out = np.empty(bs, np.uint8)
                        ptr = lambda a: a.ctypes.data_as(u8p)
                        t1e = t1d = 1e9
                        for _ in range(2):          # best of 2: the first pass touches the buffers' pages
                            t = time.perf_counter(); lz_len = [sl.lib.LZ4_compress_fast(ptr(blocks[i]), ptr(lz_dst[i]), bs, bound, 1) for i in range(k1)]; t1e = min(t1e, time.perf_counter() - t)
                            t = time.perf_counter(); [sl.lib.LZ4_decompress_safe(ptr(lz_dst[i]), ptr(lz_out), lz_len[i], bs) for i in range(k1)]; t1d = min(t1d, time.perf_counter() - t)
                        lz = {"version": sl.version, "one_thread_encode_GiBs": round(k1 * bs / 2 ** 30 / t1e, 3), "one_thread_decode_GiBs": round(k1 * bs / 2 ** 30 / t1d, 3),
                              "what": "liblz4.so.1 through ctypes, per-block calls, 256 blocks (sanity column, not the reference)"}
                except Exception:
                    lz = None
                head = ref_cpu or port
                cpu = {
                    "value": head["value"], "unit": "GiB/s", "cores": threads,
                    "kind": "reference" if ref_cpu else "port",
                    "sample": f"all {n} blocks x {bs} B of this workload, encode+decode, best of 2, {threads} threads; {head['what']}",
                    "encode_GiBs": head["encode_GiBs"], "decode_GiBs": head["decode_GiBs"],
                    "one_thread_encode_GiBs": head["one_thread_encode_GiBs"], "one_thread_decode_GiBs": head["one_thread_decode_GiBs"],
                    "port": port, "liblz4": lz,
                }
                if ref_cpu:
                    cpu["equals_gpu_bytes"] = ref_cpu["equals_gpu_bytes"]
        # the same batch through the host-pointer entry points (pageable host memory -> GPU -> host memory): PCIe and staging
        # inclusive, reported beside the metric, never as `value`
        host_to_host = None
        if not args.no_host_path:
            from k4os.compression.lz4_amd import make_arena as _arena
            h_dst, h_doff = _arena(np.full(n, bound, np.int32))
            h_back, h_boff = _arena(lens)
            te, td = [], []
            for _ in range(3):
                t = time.perf_counter(); h_len = LZ4Codec.EncodeBatchPacked(blocks.reshape(-1), off, lens, h_dst, h_doff, np.full(n, bound, np.int32)); te.append(time.perf_counter() - t)
                t = time.perf_counter(); h_dl = LZ4Codec.DecodeBatchPacked(h_dst, h_doff, h_len, h_back, h_boff, lens); td.append(time.perf_counter() - t)
            host_to_host = {"encode_GiBs": round(sum_u / 2 ** 30 / min(te), 2), "decode_GiBs": round(sum_u / 2 ** 30 / min(td), 2),
                            "roundtrip_ok": bool((h_dl == bs).all()) and bool(np.array_equal(h_back[:n * bs], blocks.reshape(-1))),
                            "note": "k4lz4_encode_batch / k4lz4_decode_batch on pageable host buffers, best of 3, PCIe + staging inclusive; not the metric"}
            # the same with the caller's three buffers page-locked once (k4lz4_host_register): what a service that reuses its
            # buffers gets; registering is not timed
            from k4os.compression.lz4_amd import host_register, host_unregister
            src_flat = np.ascontiguousarray(blocks.reshape(-1))
            regs = []
            try:
                for arr in (src_flat, h_dst, h_back):
                    host_register(arr); regs.append(arr)
                h_back[:] = 0
                te, td = [], []
                for _ in range(3):
                    t = time.perf_counter(); h_len2 = LZ4Codec.EncodeBatchPacked(src_flat, off, lens, h_dst, h_doff, np.full(n, bound, np.int32)); te.append(time.perf_counter() - t)
                    t = time.perf_counter(); h_dl = LZ4Codec.DecodeBatchPacked(h_dst, h_doff, h_len2, h_back, h_boff, lens); td.append(time.perf_counter() - t)
                host_to_host["registered"] = {
                    "encode_GiBs": round(sum_u / 2 ** 30 / min(te), 2), "decode_GiBs": round(sum_u / 2 ** 30 / min(td), 2),
                    "roundtrip_ok": bool(np.array_equal(h_len2, h_len)) and bool((h_dl == bs).all()) and bool(np.array_equal(h_back[:n * bs], src_flat)),
                    "note": "same calls, the three host buffers registered beforehand (k4lz4_host_register), best of 3"}
            except Exception as ex:       # registration is the host's to refuse (locked-memory limit): the pageable numbers stand
                host_to_host["registered"] = {"error": str(ex)[:200]}
            finally:
                for arr in regs:
                    host_unregister(arr)
        ms_per_step = elapsed / args.steps * 1e3
        alg_bytes = sum_u + sum_c + 12 * n          # U read + C written (+12 B metadata) per block
        # median of the timed launches (the mean is kept beside it)
        enc_avg, dec_avg = float(np.median(enc_ms)), float(np.median(dec_ms))

        # HBM traffic per launch: rocprofv3 PMC passes (scripts/pmc_round.sh: the L2's DRAM read and write requests in 32-byte units,
        # separate passes; the same two counters reproduce the known byte counts of scripts/ubench/pmc_calib.hip exactly, which
        # FETCH_SIZE / WRITE_SIZE do not).  The file names the kernel sources it was measured on; figures from other
        # sources (or for another workload) are not reported: traffic = null.
        traffic, issue_doc, l2_doc = {}, {}, {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path) and n == 4096 and bs == 65536:
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                doc = json.load(open(pmc_path))
                import hashlib
                h = hashlib.sha256()
                csrc = os.path.join(ROOT, "k4os", "compression", "lz4_amd", "csrc")
                for name in sorted(os.listdir(csrc)):
                    if name.endswith((".hpp", ".hip")):
                        h.update(name.encode()); h.update(open(os.path.join(csrc, name), "rb").read())
                if doc.get("source_sha") == h.hexdigest()[:16]:
                    traffic = doc.get("traffic_bytes_per_launch", {})
                    issue_doc = doc.get("issue", {})
                    l2_doc = doc.get("l2", {})
            except Exception:
                traffic = {}

        def issue(kernels, units, per, med_ms):
            """the limit that binds these kernels: how busy the vector and scalar issue ports are over the timed call
            and how many wave instructions a unit of work costs -- from the same hashed PMC file as `traffic`
            (scripts/pmc_summary.py: SQ_ACTIVE_INST_VALU / _SCA in quad-cycles, SQ_INSTS_*), else null"""
            rows = [issue_doc.get(k) for k in kernels]
            if not rows or any(r is None for r in rows) or not rows[0].get("kernel_ns_alone"):
                return None
            insts = sum(r["insts_total"] for r in rows)
            clock_ghz = rows[0]["kernel_cycles_alone"] / rows[0]["kernel_ns_alone"]
            call_cycles = med_ms * 1e6 * clock_ghz                     # kernels that run side by side share the call's cycles
            return {"valu_busy": round(4.0 * sum(r["active_valu_quadcycles"] for r in rows) / 1024 / call_cycles, 3),
                    "salu_busy": round(4.0 * sum(r["active_scalar_quadcycles"] for r in rows) / 1024 / call_cycles, 3),
                    "wave_insts_per_launch": int(insts), per: round(insts / units, 2) if units else None,
                    "note": "busy = SQ_ACTIVE_INST_{VALU,SCA} x 4 cycles / (1024 SIMDs x the timed call's cycles); the issue ports, "
                            "not HBM, are the binding limit (DESIGN.md 5)"}

        def roof(med_ms, mean_ms, kernels, timed, iss):
            ach = alg_bytes / (med_ms * 1e-3) / 1e9
            tr = [traffic.get(k) for k in kernels]
            hits, misses = sum((l2_doc.get(k) or {}).get("hit", 0.0) for k in kernels), sum((l2_doc.get(k) or {}).get("miss", 0.0) for k in kernels)
            return {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": sum(tr) if tr and all(t is not None for t in tr) else None,
                    # the L2's hit rate over the launch (TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), same hashed PMC file), else null
                    "l2_hit": round(hits / (hits + misses), 4) if hits + misses > 0 else None,
                    "median_launch_ms": round(med_ms, 4), "mean_launch_ms": round(mean_ms, 4),
                    "avg_launch_ms": round(med_ms, 4), "avg_is": "median of the timed launches (= median_launch_ms; the rates above derive from it)",
                    "algorithmic_bytes_per_launch": alg_bytes, "issue": iss,
                    "kernel": " || ".join(kernels), "timed": timed}

        n_seq = None
        if not args.no_verify or not args.no_cpu_baseline:
            try:
                n_seq = sum(oracle.count_sequences(ref_dst[int(ref_off[i]):int(ref_off[i]) + int(ref_len[i])]) for i in range(n))
            except Exception:
                n_seq = None

        cus = torch.cuda.get_device_properties(device).multi_processor_count or 256
        dec_kernel = "k4_decode_pair_kernel" if n <= 16 * cus and not os.environ.get("K4LZ4_NO_PAIR") else "k4_decode_kernel"
        if os.environ.get("K4LZ4_NO_PARSE"):
            enc_kernels = ["k4_encode_fast_gtab_kernel", "k4_encode_fast_kernel"]
            enc_timed = "k4lz4_encode_batch_device call, HIP events on the launch stream (cost + order kernels, then both one-kernel encoders concurrently)"
        else:
            enc_kernels = ["k4_parse_kernel"] + ([] if not os.environ.get("K4LZ4_NO_INLINE_EMIT") else ["k4_emit_kernel"])
            enc_timed = ("k4lz4_encode_batch_device call, HIP events on the launch stream: k4_cost_kernel + k4_order_kernel (0.09 ms), then "
                         "k4_parse_kernel -- sequences and bytes of every block, one wavefront per block, 16 blocks per workgroup")
        result = {
            "metric": "GiB/s encode+decode on batched 64 KiB blocks; bit-exact vs C# ref",
            "value": round(world * sum_u / 2 ** 30 / (elapsed / args.steps), 3),
            "unit": "GiB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": data_kind,
            "config": {"workload": f"BASELINE.json configs[1]: {n} independent {bs} B blocks per GPU, " +
                                   ("real Silesia corpus cut into consecutive blocks (K4LZ4_CORPUS_DIR)" if data_kind == "silesia" else
                                    "Silesia-like 12-class mix (seed 2+rank)") + ", L00_FAST encode + decode, HBM-resident",
                       "blocks_per_gpu": n, "block_bytes": bs, "level": "L00_FAST",
                       "ratio": round(sum_c / sum_u, 4), "total_compressed_bytes_all_gpus": total_c,
                       "encode_GiBs_per_gpu": round(sum_u / 2 ** 30 / (enc_avg * 1e-3), 3),
                       "decode_GiBs_per_gpu": round(sum_u / 2 ** 30 / (dec_avg * 1e-3), 3),
                       "parallelism": f"{world} x independent block ranges, no data-path collective"},
            "bit_exact": bit_exact,
            "bit_exact_scope": f"all {world} rank(s): every rank compares its own {n} blocks with the oracle" +
                               (" and with the compiled reference" if all_ref_equal is not None else "") + ", verdicts reduced with MIN",
            "roundtrip_all_ranks": all_roundtrip,
            # what the events bracket: the whole library call on its launch stream (enc_timed says which kernels that is)
            "roofline": roof(enc_avg, float(enc_ms.mean()), enc_kernels, enc_timed,
                             issue(enc_kernels, n * bs / 64.0, "wave_insts_per_64_positions", enc_avg)),
            # batches of up to 16 blocks per CU are decoded by the two-waves-per-block kernel (k4lz4_capi.hip launch())
            "roofline_decode": roof(dec_avg, float(dec_ms.mean()), [dec_kernel],
                                    "k4lz4_decode_batch_device call, HIP events on the launch stream (one kernel)",
                                    issue([dec_kernel], n_seq, "wave_insts_per_sequence", dec_avg)),
            "sequences_per_launch": n_seq,
            "cpu_baseline": cpu,
            "host_to_host": host_to_host,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
