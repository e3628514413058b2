"""Multi-GPU readiness that a one-GPU box can prove (BASELINE config e is 8 devices).

The path shards by independent packet ranges with no collective, so what has to be right is: one handle per slice, one
host thread per handle, nothing shared between handles (device, streams, workspace, staging, error text). These tests
run several REAL handles concurrently — on the same device twice when the box has one GPU, on every device when it has
more — through the Python ShardedDecoder (parallel.py) and the C++ one (host/sharded_decoder.hpp). No scaling curve is
claimed from them: the curve is the driver's to measure on an 8-GPU node."""
import ctypes
import importlib
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _devices():
    import torch
    n = torch.cuda.device_count()
    return [0, 0] if n < 2 else list(range(min(n, 6)))


def _dense(b):
    pk = [b.packet(i) for i in range(b.n)]
    offs = np.zeros(len(pk) + 1, np.uint64)
    offs[1:] = np.cumsum([len(p) for p in pk], dtype=np.uint64)
    return np.frombuffer(b"".join(pk), np.uint8), offs


@pytest.mark.parametrize("depth,ch,fl,n", [(16, 2, 4096, 3001), (24, 6, 512, 777)])
def test_sharded_decoder_on_real_handles(pkg, oracle, synth, helpers, depth, ch, fl, n):
    """parallel.ShardedDecoder with real PacketDecoders: [0, 0] on a one-GPU box (two handles, two threads, one device),
    every device otherwise; ragged split; checked against the oracle, error packets included."""
    par = importlib.import_module("saprobe-alac_amd.parallel")
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    b = synth.gen_batch(cfg, n, threads=8)
    rng = np.random.default_rng(n)
    packets = [b.packet(i) for i in range(b.n)]
    for k in rng.integers(0, n, 40):  # damaged packets in every slice
        packets[int(k)] = packets[int(k)][:max(1, len(packets[int(k)]) // 3)]
    blob, offs, sizes = helpers.pack_dense(packets)
    offs1 = np.concatenate([offs, [np.uint64(len(blob))]]).astype(np.uint64)
    ref_blob, ref_offs, ref_sizes = helpers.pack_packets(packets)
    ref = oracle.decode_batch(cfg, ref_blob, ref_offs, ref_sizes, threads=8)
    pcfg = pkg.PacketConfig(FrameLength=fl, BitDepth=depth, NumChannels=ch)
    devs = _devices()
    sd = par.ShardedDecoder(pcfg, devs)
    try:
        for _ in range(2):
            got = sd.decode_batch(blob, offs1)
            helpers.assert_same_decode(cfg, ref, got, bpf, "devices %s" % devs)
    finally:
        sd.close()


def test_handles_share_nothing_across_threads(pkg, oracle, synth, helpers):
    """Two handles driven from two threads at once, many rounds, different configurations: every result must be its own
    (no shared stream, workspace or staging), and a failing call on one thread must not leak its error text into the
    other (alacgpu_last_error is thread-local)."""
    devs = _devices()
    jobs = []
    for i, (depth, ch, fl) in enumerate([(16, 2, 1024), (24, 2, 512)]):
        cfg = oracle.make_config(fl, depth, ch)
        b = synth.gen_batch(cfg, 900 + 64 * i, threads=8)
        blob, offs = _dense(b)
        jobs.append((cfg, b, blob, offs, devs[i % len(devs)]))
    errors = []
    barrier = threading.Barrier(2)

    def work(i):
        cfg, b, blob, offs, dev = jobs[i]
        try:
            pcfg = pkg.PacketConfig(FrameLength=cfg.frame_length, BitDepth=cfg.bit_depth, NumChannels=cfg.num_channels)
            with pkg.NewPacketDecoder(pcfg, dev) as dec:
                barrier.wait()
                for r in range(6):
                    out, fr, st = dec.decode_batch(blob, offs)
                    assert (st == 0).all() and np.array_equal(fr, b.frames), "thread %d round %d" % (i, r)
                    full = b.frames == cfg.frame_length
                    assert np.array_equal(out[full], b.pcm[full]), "thread %d round %d" % (i, r)
                    if i == 0:  # a call that fails on this thread only
                        rc = dec._lib.alacgpu_decode_batch(dec._h, None, 0, None, 5, None, 0, None, None)
                        assert rc == -2 and b"null" in dec._lib.alacgpu_last_error()
                    else:
                        assert dec._lib.alacgpu_last_error() in (b"", None) or b"null" not in dec._lib.alacgpu_last_error()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def test_cpp_sharded_decoder(pkg, oracle, synth, helpers):
    """host/sharded_decoder.hpp (one std::thread + handle per device) through the test shim."""
    from test_container import _build_shim
    L = _build_shim(True, pkg)
    cfg = oracle.make_config(2048, 16, 2)
    b = synth.gen_batch(cfg, 1501, threads=8)
    blob, offs = _dense(b)
    ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
    pcfg = pkg.PacketConfig(FrameLength=2048, BitDepth=16, NumChannels=2)
    for devs in ([0], _devices(), _devices() + [0]):
        stride = 2048 * 4
        out = np.zeros((b.n, stride), np.uint8)
        fr = np.zeros(b.n, np.uint32)
        st = np.full(b.n, -1, np.int32)
        d = (ctypes.c_int * len(devs))(*devs)
        L.shim_sharded_decode.restype = ctypes.c_long
        rc = L.shim_sharded_decode(ctypes.byref(pcfg), d, ctypes.c_size_t(len(devs)), ctypes.c_void_p(blob.ctypes.data),
                                   ctypes.c_size_t(len(blob)), ctypes.c_void_p(offs.ctypes.data), ctypes.c_size_t(b.n), ctypes.c_void_p(out.ctypes.data),
                                   ctypes.c_size_t(stride), ctypes.c_void_p(fr.ctypes.data), ctypes.c_void_p(st.ctypes.data))
        assert rc == 0, L.shim_last_error()
        helpers.assert_same_decode(cfg, ref, (out, fr, st), 4, "devices %s" % devs)
