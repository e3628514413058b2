"""N>1 path on CPU: world_size-2 gloo processes shard a batch by packet range with no data-path collective.
The decode itself is stood in for by the oracle (there is no GPU here); what is under test is the partition,
the per-rank independence and the status gather that bench.py / parallel.py use."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly(pkg):
    par = importlib.import_module("saprobe-alac_amd.parallel")
    for n in (0, 1, 7, 64, 65, 4096, 262144):
        for world in (1, 2, 3, 4, 8):
            spans = [par.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert all(lo <= hi for lo, hi in spans)
    assert par.shard_range(262144, 8, 3) == (98304, 131072)  # BASELINE config e: 32768 per GPU
    with pytest.raises(ValueError):
        par.shard_range(10, 2, 2)


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from oracle import oracle
    par = importlib.import_module("saprobe-alac_amd.parallel")
    synth = importlib.import_module("saprobe-alac_amd.synth")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = oracle.make_config(256, 16, 2)
        n = 101  # ragged: 51 + 50
        lo, hi = par.shard_range(n, world, rank)
        # every rank builds ONLY its slice of the seeded stream, like bench.py does (first_index = lo)
        b = synth.gen_batch(cfg, hi - lo, first_index=lo, threads=2)
        out, frames, status = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes)
        ok = bool((status == 0).all()) and bool(np.array_equal(out, b.pcm))
        fr_all, st_all = par.gather_status(frames, status, n, world, rank, dist)
        # the whole batch decoded in one go must agree with the gathered per-rank results
        full = synth.gen_batch(cfg, n, threads=2)
        _, fr_ref, st_ref = oracle.decode_batch(cfg, full.blob, full.offsets, full.sizes)
        ok = ok and bool(np.array_equal(fr_all, fr_ref)) and bool(np.array_equal(st_all, st_ref))
        ok = ok and bool(np.array_equal(full.pcm[lo:hi], b.pcm))
        q.put((rank, ok, int(frames.sum())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(pkg, synth, oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res)


def test_sharded_decoder_single_process(pkg, synth, oracle):
    """ShardedDecoder (hipSetDevice per slice) with a stand-in per-device decoder: slices reassemble in order."""
    par = importlib.import_module("saprobe-alac_amd.parallel")
    cfg = oracle.make_config(128, 16, 2)
    b = synth.gen_batch(cfg, 37, threads=2)
    pk = [b.packet(i) for i in range(b.n)]
    offs = np.zeros(len(pk) + 1, np.uint64)
    offs[1:] = np.cumsum([len(p) for p in pk], dtype=np.uint64)
    blob = np.frombuffer(b"".join(pk) + b"\0", np.uint8)

    class FakeDecoder:
        def __init__(self, config, device):
            self.device = device

        def decode_batch(self, blob, offsets):
            n = len(offsets) - 1
            sizes = np.diff(offsets).astype(np.uint32)
            pad, o2, s2 = __import__("conftest").pack_packets(
                [blob[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(n)])
            return oracle.decode_batch(cfg, pad, o2, s2)

        def close(self):
            pass

    sd = par.ShardedDecoder(cfg, [0, 1, 2], make_decoder=FakeDecoder)
    out, frames, status = sd.decode_batch(blob, offs)
    assert (status == 0).all() and np.array_equal(frames, b.frames) and np.array_equal(out, b.pcm)
    sd.close()
