"""Shared fixtures. `-m "not gpu"` runs here on CPU; `-m gpu` runs on a real MI355X.

CPU suite: oracle vs golden vectors, round trips, the kernel's lane logic compiled for the host
(tests/host_sim) vs the oracle, host-side API, C-ABI symbol check, gloo sharding.
GPU suite: parity of the HIP path with the oracle, always through the C ABI.
"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has a hyphen)."""
    return importlib.import_module("saprobe-alac_amd")


@pytest.fixture(scope="session")
def synth():
    m = importlib.import_module("saprobe-alac_amd.synth")
    m.build()
    return m


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def lane_sim(oracle):
    """Host build of csrc/alac_wave.h (test-only; see tests/host_sim/lane_sim.cpp)."""
    d = os.path.join(ROOT, "tests", "host_sim")
    so = os.path.join(d, "liblane_sim.so")
    srcs = [os.path.join(d, "lane_sim.cpp")] + [os.path.join(ROOT, "saprobe-alac_amd", "csrc", h)
                                                for h in ("alac_wave.h", "alac_regular.h", "alac_split.h", "alac_duo.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-fwrapv", "-fPIC", "-std=c++17", "-Wno-unknown-pragmas", "-shared",
                               "-o", so, srcs[0]])
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    L.lane_sim_decode_batch.argtypes = [vp, vp, ctypes.c_size_t, vp, vp, ctypes.c_size_t, vp, ctypes.c_size_t, vp, vp,
                                        ctypes.c_int, ctypes.c_int, vp, ctypes.c_int]

    def run(cfg, blob, offsets, sizes, variant=-1, stride_pad=0, want_classes=False, blob_bytes=None, guard=False):
        """blob_bytes: the readable bytes of the blob (default: all of it); guard: run on a copy that ends at an
        inaccessible page, so that any read past the blob is fatal."""
        n = len(offsets)
        stride = (oracle.frame_bytes(cfg) + 15) // 16 * 16 + stride_pad
        out = np.zeros((n, stride), np.uint8)
        classes = np.zeros(n, np.uint32)
        fr = np.zeros(n, np.uint32)
        st = np.zeros(n, np.int32)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        sizes = np.ascontiguousarray(sizes, np.uint32)
        rc = L.lane_sim_decode_batch(ctypes.byref(cfg), blob.ctypes.data, len(blob) if blob_bytes is None else blob_bytes,
                                     offsets.ctypes.data, sizes.ctypes.data, n, out.ctypes.data, stride, fr.ctypes.data,
                                     st.ctypes.data, 1, variant, classes.ctypes.data, 1 if guard else 0)
        assert rc == 0
        if want_classes:
            return out, fr, st, classes
        return out, fr, st

    return run


def pack_packets(packets, pad=64):
    """List of packet bytes -> (blob, offsets[n], sizes[n]), every packet followed by `pad` zero bytes and aligned to
    16 (the round-1 device layout; the decoder needs neither any more, see pack_dense)."""
    offs, sizes, buf = [], [], bytearray()
    for q in packets:
        offs.append(len(buf))
        sizes.append(len(q))
        buf += q
        buf += bytes(pad)
        while len(buf) % 16:
            buf.append(0)
    buf += bytes(64)
    return (np.frombuffer(bytes(buf), np.uint8).copy(), np.array(offs, np.uint64), np.array(sizes, np.uint32))


def pack_dense(packets, lead=0):
    """Packets back to back as in an mdat (internal/mp4/mp4.go:382-420): no padding, no alignment, the blob ends with
    the last packet's last byte. lead: bytes of 0xFF in front (shifts every packet's alignment)."""
    offs, sizes, buf = [], [], bytearray(b"\xff" * lead)
    for q in packets:
        offs.append(len(buf))
        sizes.append(len(q))
        buf += q
    return (np.frombuffer(bytes(buf) if buf else b"\0", np.uint8).copy()[:len(buf)], np.array(offs, np.uint64),
            np.array(sizes, np.uint32))


def mutate_packets(batch, rng, n_out):
    """Corrupt valid packets: bit flips, truncation, header damage, garbage, constant tails."""
    pk = []
    for _ in range(n_out):
        src = bytearray(batch.packet(int(rng.integers(batch.n))))
        mode = int(rng.integers(6))
        if mode == 0:
            for _ in range(int(rng.integers(1, 8))):
                k = int(rng.integers(len(src)))
                src[k] ^= 1 << int(rng.integers(8))
        elif mode == 1:
            src = src[:int(rng.integers(0, len(src) + 1))]
        elif mode == 2:
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(min(len(src), 12)))
                src[k] = int(rng.integers(256))
        elif mode == 3:
            src = bytearray(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes())
        elif mode == 4:
            src = src[:int(rng.integers(1, len(src) + 1))]
            k = int(rng.integers(len(src)))
            src[k] ^= 0xff
        else:
            k = int(rng.integers(len(src)))
            src[k:] = bytes([0xff if rng.integers(2) else 0]) * (len(src) - k)
        pk.append(bytes(src))
    return pk


def antiphase_packets(synth, cfg, n, seed=1, order=4, every=2):
    """Multi-channel packets whose PAIRS are loud and in anti-phase (L = -R near full scale, a slow sine plus noise): with the
    pair matrixed (mixRes 1) the difference channel v = L - R (matrix.go:40-41 inverted) needs all of its chanBits = depth - shift
    + 1 = 17 bits, and the mid / side arithmetic its whole range. Every `every`-th packet is one of those, the others are quiet.
    -> list of (packet bytes, expected PCM bytes)."""
    rng = np.random.default_rng(seed)
    depth, ch, fl = cfg.bit_depth, cfg.num_channels, cfg.frame_length
    bs = {16: 0, 20: 0, 24: 1, 32: 2}[depth]
    top = 1 << (depth - 8 * bs - 1)
    ne = synth.num_elements(ch)
    out = []
    for k in range(n):
        t = np.arange(fl)[:, None]
        amp = (top - 40) if k % every == 0 else top // 64
        base = (amp * np.sin(t / 9.0 + rng.uniform(0, 6.28, size=(1, ch)))).astype(np.int64) + rng.integers(-30, 31, size=(fl, ch))
        hi = np.clip(base, -top, top - 1)
        for c in range(0, ch - 1, 2):  # whatever the layout makes of it: neighbouring output channels in anti-phase
            hi[:, c + 1] = np.clip(-hi[:, c] - 1, -top, top - 1)
        if ch >= 3:
            hi[:, 2] = np.clip(-hi[:, 1] - 1, -top, top - 1)
        pcm = ((hi << (8 * bs)) | rng.integers(0, 1 << (8 * bs), size=(fl, ch))) if bs else hi
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        elems = [synth.default_elem(order=order, mix_res=1, mix_bits=2, bytes_shifted=bs, never_escape=1) for _ in range(ne)]
        out.append((synth.encode_packet(cfg, elems, pcm), synth.pack_pcm(cfg, pcm)))
    return out


def loud_packets(synth, cfg, n, seed=1, pb_factor=7, order=0):
    """Packets whose residuals keep the Golomb mean at the top of its range: compressed elements (never the escape
    form) whose folded residuals n = 2|r| (- 1) sit in 56 000..65 535 sample after sample, with pbFactor 7 — the largest
    effective pb = PB * 7 / 4 a cookie byte PB can give (decoder.go:296-299) — so that mean approaches 512 * 65 535 and
    pb * mean (golomb.go:215) its largest product; one sample in sixteen is the most negative value of the channel
    (n = 65 535 exactly, or beyond it and into the clamp of golomb.go:216-218 where the channel is wider than 16 bits).
    order 0: the samples ARE the residuals (predictor.go:53-56). -> list of (packet bytes, expected PCM bytes)."""
    rng = np.random.default_rng(seed)
    depth, ch, fl = cfg.bit_depth, cfg.num_channels, cfg.frame_length
    bs = {16: 0, 20: 0, 24: 1, 32: 2}[depth]
    ne = synth.num_elements(ch)
    out = []
    for _ in range(n):
        mag = rng.integers(28000, 32768, size=(fl, ch))
        sgn = rng.choice([-1, 1], size=(fl, ch))
        hi = mag * sgn
        hi[rng.integers(0, 16, size=(fl, ch)) == 0] = -32768 if depth == 16 else -(1 << (depth - 8 * bs - 1))
        pcm = (hi.astype(np.int64) << (8 * bs)) | rng.integers(0, 1 << (8 * bs), size=(fl, ch)) if bs else hi
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        elems = [synth.default_elem(order=order, pb_factor=pb_factor, mix_res=0, mix_bits=0, bytes_shifted=bs, never_escape=1)
                 for _ in range(ne)]
        out.append((synth.encode_packet(cfg, elems, pcm), synth.pack_pcm(cfg, pcm)))
    return out


def assert_same_decode(cfg, ref, got, bpf, what=""):
    """(out, frames, status) triples must agree: status, frame count, and PCM bytes of the frames."""
    o1, f1, s1 = ref
    o2, f2, s2 = got
    assert np.array_equal(s1, s2), "%s status differs at %s" % (what, np.nonzero(s1 != s2)[0][:8])
    assert np.array_equal(f1, f2), "%s frame count differs at %s" % (what, np.nonzero(f1 != f2)[0][:8])
    for i in range(len(f1)):
        nb = int(f1[i]) * bpf
        if not np.array_equal(o1[i, :nb], o2[i, :nb]):
            bad = np.nonzero(o1[i, :nb] != o2[i, :nb])[0]
            raise AssertionError("%s packet %d: PCM differs at byte %d (of %d)" % (what, i, bad[0], nb))


@pytest.fixture(scope="session")
def helpers():
    class H:
        pass

    H.pack_packets = staticmethod(pack_packets)
    H.pack_dense = staticmethod(pack_dense)
    H.mutate_packets = staticmethod(mutate_packets)
    H.assert_same_decode = staticmethod(assert_same_decode)
    H.loud_packets = staticmethod(loud_packets)
    H.antiphase_packets = staticmethod(antiphase_packets)
    return H


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_decoder_factory(pkg):
    """PacketDecoder factory for the gpu-marked tests; fails loudly if the extension is missing."""
    if not _have_gpu():
        pytest.fail("gpu-marked test on a machine without a GPU")
    assert os.path.exists(pkg.lib_path()), "libalacgpu.so missing: build() did not run"

    def make(cfg_like):
        c = pkg.PacketConfig(FrameLength=cfg_like.frame_length, BitDepth=cfg_like.bit_depth,
                             NumChannels=cfg_like.num_channels, PB=cfg_like.pb, MB=cfg_like.mb, KB=cfg_like.kb,
                             MaxRun=cfg_like.max_run, SampleRate=cfg_like.sample_rate)
        return pkg.NewPacketDecoder(c, 0)

    return make
