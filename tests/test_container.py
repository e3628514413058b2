"""the MP4 sample table -> batch descriptor (saprobe-alac_amd/mp4.py) and the streaming façade
(stream.py). The demuxer is host-only and is checked here on CPU against files written by tests/m4a.py, with the
layouts and the damage the reference's own tests use (tests/error_test.go:146-442: empty, garbage, truncated before
/ inside moov, corrupted stsd, corrupted cookie, zeroed stsz, truncated packet). The façade needs the GPU."""
import importlib

import numpy as np
import pytest

from tests import m4a


@pytest.fixture(scope="module")
def mp4(pkg):
    return importlib.import_module("saprobe-alac_amd.mp4")


@pytest.fixture(scope="module")
def stream(pkg):
    return importlib.import_module("saprobe-alac_amd.stream")


def _packets(synth, cfg, n, seed=7):
    b = synth.gen_batch(cfg, n, base_seed=seed, threads=4)
    return b, [b.packet(i) for i in range(n)]


LAYOUTS = [dict(), dict(co64=True), dict(per_chunk=[3]), dict(per_chunk=[1, 4, 2], gap=5), dict(wrapped=True),
           dict(qt_version=1), dict(decoy_track=True), dict(large_mdat=True, per_chunk=[7], co64=True)]


@pytest.mark.parametrize("layout", LAYOUTS)
def test_sample_table_matches_what_was_written(pkg, mp4, synth, oracle, layout):
    cfg = oracle.make_config(256, 16, 2)
    b, packets = _packets(synth, cfg, 23)
    data = m4a.write_m4a(cfg, packets, **layout)
    t = mp4.find_alac_track(data)
    assert len(t) == len(packets)
    assert t.sizes.tolist() == [len(p) for p in packets]
    for i, p in enumerate(packets):
        o = int(t.offsets[i])
        assert data[o:o + len(p)] == p
    c = pkg.ParseMagicCookie(t.cookie)
    assert (c.FrameLength, c.BitDepth, c.NumChannels, c.SampleRate) == (256, 16, 2, cfg.sample_rate)
    assert t.contiguous() == (layout.get("gap", 0) == 0)
    # numpy arrays and memoryviews are accepted as well
    t2 = mp4.find_alac_track(np.frombuffer(data, np.uint8))
    assert np.array_equal(t2.offsets, t.offsets)


def test_constant_sample_size_and_short_tables(mp4, oracle):
    cfg = oracle.make_config(64, 16, 1)
    packets = [bytes([i]) * 10 for i in range(9)]
    t = mp4.find_alac_track(m4a.write_m4a(cfg, packets, const_size=True, per_chunk=[4]))
    assert t.sizes.tolist() == [10] * 9 and len(set(t.offsets.tolist())) == 9
    # the stsc runs cover fewer samples than stsz declares: the table ends with the chunks (mp4.go:398-412)
    data = bytearray(m4a.write_m4a(cfg, packets, per_chunk=[2, 2, 2, 2, 1]))
    k = data.find(b"stco")
    data[k + 8:k + 12] = (3).to_bytes(4, "big")  # keep 3 of the 5 chunks
    assert len(mp4.find_alac_track(bytes(data))) == 6


def test_stsc_runs_are_walked_in_file_order(mp4):
    """lookupSamplesPerChunk stops at the first run that starts beyond the chunk, sorted or not (mp4.go:579-591)."""
    import struct
    cfg = type("C", (), dict(frame_length=16, bit_depth=16, pb=40, mb=10, kb=14, num_channels=1, max_run=255,
                             max_frame_bytes=0, avg_bit_rate=0, sample_rate=8000))
    entry = m4a.sample_entry(b"alac", m4a.cookie_bytes(cfg))
    # runs (1,2) (5,1) (3,4): chunks 1-4 take 2; chunk 5+ stop at (5,1) -> 1 ... and (3,4) is reached only from chunk 5
    tr = m4a.trak(entry, [100, 200, 300, 400, 500, 600], [(1, 2), (5, 1), (3, 4)], [1] * 40)
    data = m4a.box(b"ftyp", b"M4A ") + m4a.box(b"moov", tr) + bytes(700)
    t = mp4.find_alac_track(data)
    per_chunk = [int((t.offsets // 100 == c).sum()) for c in range(1, 7)]
    assert per_chunk == [2, 2, 2, 2, 4, 4]
    assert struct.calcsize(">I") == 4


def test_container_errors(mp4, synth, oracle):
    cfg = oracle.make_config(128, 16, 2)
    _, packets = _packets(synth, cfg, 5)
    good = m4a.write_m4a(cfg, packets)

    def sentinel(data):
        with pytest.raises(mp4.Mp4Error) as e:
            mp4.find_alac_track(data)
        return e.value.sentinel

    assert sentinel(b"") == mp4.ErrNoALACTrack                                    # TestDecode_EmptyReader
    assert sentinel(bytes(range(256)) * 4) in (mp4.ErrNoALACTrack, mp4.ErrInvalidBoxSize)  # TestDecode_GarbageData
    moov = good.find(b"moov") - 4
    assert sentinel(good[:moov]) == mp4.ErrNoALACTrack                            # TestDecode_TruncatedBeforeMoov
    assert sentinel(good[:moov + 60]) == mp4.ErrNoALACTrack                       # TestDecode_TruncatedMoov
    bad = bytearray(good)
    k = bad.find(b"alac", bad.find(b"stsd"))
    bad[k:k + 4] = b"XXXX"
    assert sentinel(bytes(bad)) == mp4.ErrNoALACTrack                             # TestDecode_CorruptedStsd
    for name, err in ((b"stsc", mp4.ErrNoStsc), (b"stsz", mp4.ErrNoStsz), (b"stco", mp4.ErrNoChunkOffset)):
        bad = bytearray(good)
        k = bad.find(name)
        bad[k:k + 4] = b"free"
        assert sentinel(bytes(bad)) == err
    bad = bytearray(good)
    k = bad.find(b"stsz")
    bad[k + 12:k + 16] = (1 << 30).to_bytes(4, "big")   # more entries than the file holds
    assert sentinel(bytes(bad)) == mp4.ErrInvalidStsz
    bad = bytearray(good)
    bad[moov + 8:moov + 12] = (4).to_bytes(4, "big")    # first child of moov with a size below its header
    assert sentinel(bytes(bad)) == mp4.ErrInvalidBoxSize


def test_corrupted_cookie_is_a_config_error(pkg, mp4, synth, oracle):
    """TestDecode_CorruptedALACCookie (tests/error_test.go:257-334): the track is found, the config is refused."""
    cfg = oracle.make_config(128, 16, 2)
    _, packets = _packets(synth, cfg, 3)
    data = bytearray(m4a.write_m4a(cfg, packets))
    k = data.find(b"alac", data.find(b"stsd"))
    data[k + 4 + 28 + 4] = 9  # compatibleVersion
    t = mp4.find_alac_track(bytes(data))
    with pytest.raises(pkg.ErrConfig):
        pkg.ParseMagicCookie(t.cookie)


@pytest.mark.gpu
@pytest.mark.parametrize("depth,ch,layout", [(16, 2, dict()), (24, 2, dict(per_chunk=[5, 9], gap=3, co64=True)),
                                             (16, 1, dict(wrapped=True, qt_version=1)), (24, 6, dict(decoy_track=True))])
def test_streaming_decoder_reads_the_source_pcm(pkg, stream, synth, oracle, depth, ch, layout):
    cfg = oracle.make_config(512, depth, ch)
    b, packets = _packets(synth, cfg, 150, seed=depth + ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    want = b"".join(b.pcm[i, :int(b.frames[i]) * bpf].tobytes() for i in range(b.n))
    data = m4a.write_m4a(cfg, packets, **layout)
    with stream.NewDecoder(data, window=64) as d:
        f = d.Format()
        assert (f.SampleRate, f.BitDepth, f.Channels) == (cfg.sample_rate, depth, ch)
        assert abs(d.Duration() - 150 * 512 / cfg.sample_rate) < 1e-6
        got = bytearray()
        while True:                      # odd read sizes: packet and window boundaries fall inside reads
            part = d.Read(7919)
            if not part:
                break
            got += part
        assert bytes(got) == want
        assert d.Read(10) == b""
        # packet-aligned seek (decode.go:103-124)
        pos = d.Seek(40.5 * 512 / cfg.sample_rate)
        assert abs(pos - 40 * 512 / cfg.sample_rate) < 1e-9 and abs(d.Position() - pos) < 1e-9
        off = sum(int(b.frames[i]) * bpf for i in range(40))
        assert d.Read(1000) == want[off:off + 1000]
        assert d.Seek(-3.0) == 0.0 and d.Read(64) == want[:64]
        assert d.Seek(1e9) == d.Duration() and d.Read(1) == b""
    with stream.NewDecoder(np.frombuffer(data, np.uint8)) as d:   # one window covers the file
        assert d.ReadAll() == want


@pytest.mark.gpu
def test_streaming_decoder_errors(pkg, stream, synth, oracle, tmp_path):
    cfg = oracle.make_config(256, 16, 2)
    b, packets = _packets(synth, cfg, 40)
    bpf = 4
    want = [b.pcm[i, :int(b.frames[i]) * bpf].tobytes() for i in range(b.n)]
    with pytest.raises(stream.ErrNoTrack):
        stream.NewDecoder(b"")                                   # TestNewDecoder_EmptyReader
    # a packet that does not decode: everything before it is delivered, then the error, again and again
    broken = list(packets)
    broken[17] = bytes([0xA0, 0x00, 0x00])                       # an unsupported element (PCE, decoder.go:176-179)
    data = m4a.write_m4a(cfg, broken)
    with stream.NewDecoder(data, window=8) as d:
        got = d.Read(10 ** 9)
        assert got == b"".join(want[:17])
        for _ in range(2):
            with pytest.raises(pkg.ErrDecode) as e:
                d.Read(100)
            assert "decoding packet 17" in str(e.value)
        d.Seek(18 * 256 / cfg.sample_rate)
        assert d.Read(10 ** 9) == b"".join(want[18:])
    # TestDecode_TruncatedPacket (tests/error_test.go:413-442): the file ends inside the last packets
    data = m4a.write_m4a(cfg, packets)
    cut = data[:len(data) - len(packets[-1]) - 3]
    path = tmp_path / "cut.m4a"
    path.write_bytes(cut)
    with stream.NewDecoder(str(path), window=16) as d:           # a path: the file is mapped
        got = d.Read(10 ** 9)
        assert got == b"".join(want[:38])
        with pytest.raises(pkg.AlacError) as e:
            d.Read(1)
        assert "reading sample 38" in str(e.value)


# ---- the C++ host mirrors (saprobe-alac_amd/host/mp4_demux.hpp, stream_decoder.hpp) through a ctypes shim ---------
import ctypes
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SENTINELS = ["ErrNoALACTrack", "ErrInvalidEntry", "ErrInvalidBoxSize", "ErrNoChunkOffset", "ErrInvalidCo64",
              "ErrNoStsc", "ErrInvalidStsc", "ErrNoStsz", "ErrInvalidStsz"]  # enum order of mp4_demux.hpp


def _build_shim(with_decoder, pkg=None):
    d = os.path.join(_ROOT, "tests", "host_sim")
    so = os.path.join(d, "libhost_shim_gpu.so" if with_decoder else "libhost_shim_cpu.so")
    host = os.path.join(_ROOT, "saprobe-alac_amd", "host")
    srcs = [os.path.join(d, "host_shim.cpp")] + [os.path.join(host, h) for h in os.listdir(host)]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-pthread", "-shared", "-o", so, srcs[0]]
        if with_decoder:
            libdir = os.path.dirname(pkg.lib_path())
            cmd += ["-DSHIM_WITH_DECODER", "-L" + libdir, "-lalacgpu", "-Wl,-rpath," + libdir]
        subprocess.check_call(cmd)
    return ctypes.CDLL(so)


def _cpp_demux(L, data):
    cookie = (ctypes.c_uint8 * 256)()
    clen = ctypes.c_size_t(0)
    cap = 1 << 16
    offs = np.zeros(cap, np.uint64)
    sizes = np.zeros(cap, np.uint32)
    L.demux_track.restype = ctypes.c_long
    buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
    n = L.demux_track(buf, ctypes.c_size_t(len(data)), cookie, ctypes.c_size_t(256), ctypes.byref(clen),
                      offs.ctypes.data_as(ctypes.c_void_p), sizes.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
    if n < 0:
        return _SENTINELS[-n - 1]
    return bytes(cookie[:clen.value]), offs[:n].copy(), sizes[:n].copy()


def test_cpp_demuxer_agrees_with_the_python_one(mp4, synth, oracle):
    L = _build_shim(False)
    cfg = oracle.make_config(256, 16, 2)
    _, packets = _packets(synth, cfg, 23)
    files = [m4a.write_m4a(cfg, packets, **layout) for layout in LAYOUTS]
    files.append(m4a.write_m4a(cfg, [bytes([i]) * 10 for i in range(9)], const_size=True, per_chunk=[4]))
    good = files[0]
    moov = good.find(b"moov") - 4
    files += [b"", bytes(range(256)) * 4, good[:moov], good[:moov + 60], good[:len(good) // 2]]
    for name in (b"stsd", b"stsc", b"stsz", b"stco", b"alac", b"stbl", b"minf"):
        bad = bytearray(good)
        k = bad.find(name)
        bad[k:k + 4] = b"free"
        files.append(bytes(bad))
    rng = np.random.default_rng(5)
    for _ in range(300):  # random damage inside moov: both must land on the same answer
        bad = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            bad[moov + int(rng.integers(0, good.find(b"mdat") - moov))] = int(rng.integers(256))
        files.append(bytes(bad))
    agree = 0
    for data in files:
        got = _cpp_demux(L, data)
        try:
            t = mp4.find_alac_track(data)
            if len(t) > (1 << 16):
                continue
            assert not isinstance(got, str), got
            assert got[0] == bytes(t.cookie) and np.array_equal(got[1], t.offsets) and np.array_equal(got[2], t.sizes)
        except mp4.Mp4Error as e:
            assert got == [k for k in _SENTINELS if getattr(mp4, k) == e.sentinel][0]
        agree += 1
    assert agree > 300


@pytest.mark.gpu
def test_cpp_streaming_decoder(pkg, synth, oracle):
    L = _build_shim(True, pkg)
    for f in (L.shim_read, L.shim_open):
        f.restype = ctypes.c_long
    for f in (L.shim_seek, L.shim_duration, L.shim_position):
        f.restype = ctypes.c_longlong
    L.shim_last_error.restype = ctypes.c_char_p
    cfg = oracle.make_config(512, 24, 2)
    b, packets = _packets(synth, cfg, 120, seed=3)
    bpf = 6
    want = [b.pcm[i, :int(b.frames[i]) * bpf].tobytes() for i in range(b.n)]
    broken = list(packets)
    broken[77] = bytes([0xA0, 0, 0])
    for pk, stop in ((packets, None), (broken, 77)):
        data = m4a.write_m4a(cfg, pk, per_chunk=[11, 4], gap=2)
        buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
        h = ctypes.c_void_p()
        assert L.shim_open(buf, ctypes.c_size_t(len(data)), ctypes.c_size_t(32), ctypes.byref(h)) == 0, L.shim_last_error()
        assert L.shim_duration(h) == 120 * 512 * 10 ** 9 // cfg.sample_rate
        out = (ctypes.c_uint8 * 100003)()
        got = bytearray()
        while True:
            n = L.shim_read(h, out, ctypes.c_size_t(100003))
            if n <= 0:
                break
            got += bytes(out[:n])
        if stop is None:
            assert n == 0 and bytes(got) == b"".join(want)
            assert L.shim_seek(h, ctypes.c_longlong(int(50.5 * 512 / cfg.sample_rate * 1e9))) == L.shim_position(h)
            n = L.shim_read(h, out, ctypes.c_size_t(1000))
            assert bytes(out[:n]) == b"".join(want[50:])[:1000]
        else:
            assert n == -3 and b"decoding packet 77" in L.shim_last_error()   # ErrDecode, again on every Read
            assert L.shim_read(h, out, ctypes.c_size_t(10)) == -3
            assert bytes(got) == b"".join(want[:77])
        L.shim_close(h)
    # TestDecode_TruncatedPacket (tests/error_test.go:413-442): the file ends inside the last packets. Everything in front of
    # the first sample that cannot be read is delivered (it sits in the middle of a read-ahead window), then the read error,
    # on every further Read (decode.go:157-186; the Go twin go/alacgpu_decoder.go: lostIdx does the same)
    data = m4a.write_m4a(cfg, packets)
    cut = data[:len(data) - len(packets[-1]) - len(packets[-2]) - 5]
    buf = (ctypes.c_uint8 * len(cut)).from_buffer_copy(cut)
    h = ctypes.c_void_p()
    assert L.shim_open(buf, ctypes.c_size_t(len(cut)), ctypes.c_size_t(32), ctypes.byref(h)) == 0, L.shim_last_error()
    got = bytearray()
    while True:
        n = L.shim_read(h, out, ctypes.c_size_t(100003))
        if n <= 0:
            break
        got += bytes(out[:n])
    assert n == -4 and b"reading sample 117" in L.shim_last_error()                     # ErrRead
    assert bytes(got) == b"".join(want[:117])
    assert L.shim_read(h, out, ctypes.c_size_t(10)) == -4
    L.shim_close(h)
    h = ctypes.c_void_p()
    junk = (ctypes.c_uint8 * 64)()
    assert L.shim_open(junk, ctypes.c_size_t(64), ctypes.c_size_t(8), ctypes.byref(h)) == -1     # ErrNoTrack
