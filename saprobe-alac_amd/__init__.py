"""saprobe-alac_amd — MI355X (gfx950) batch ALAC packet decoder.

Host-side mirror of the reference's packet layer (mycophonic/saprobe-alac, root package `alac`)
over the C ABI of include/alacgpu.h:

    reference (Go)                                   here
    ------------------------------------------------ ----------------------------------
    PacketConfig             config.go:27-38          PacketConfig
    PCMFormat                format.go:20-24          PCMFormat
    NewPacketDecoder         decoder.go:90            NewPacketDecoder / PacketDecoder(...)
    (*PacketDecoder).Format  decoder.go:112           PacketDecoder.Format()
    (*PacketDecoder).DecodePacket  decoder.go:117     PacketDecoder.DecodePacket(packet) -> bytes
    DecodePackets (new batch entry, north star)       PacketDecoder.DecodePackets(packets)
    ErrConfig / ErrDecode    errors.go:22-34          ErrConfig / ErrDecode (+ .sentinel)
    internal sentinels       internal/alac/errors.go  ErrBitstreamOverrun, ErrSampleOverrun, ...

The reference is Go; this image has no Go toolchain, so the host side above the C ABI is Python
(ctypes) for the tests/bench and C++ (host/packet_decoder.hpp) for native callers; INTEGRATION.md
shows the cgo binding. Every decode call runs the HIP kernels in csrc/: there is no CPU decode
path, and loading fails loudly when libalacgpu.so is missing.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = None

__all__ = [
    "PacketConfig", "PCMFormat", "PacketDecoder", "NewPacketDecoder", "ParseMagicCookie",
    "ErrConfig", "ErrDecode", "AlacError", "build", "lib", "lib_path", "trim",
]

PACKET_PAD = 0  # ALACGPU_PACKET_PAD: blobs are dense since 0.3.0


def trim():
    """alacgpu_trim(): free what destroyed decoders left in the per-process handle pool (see PacketDecoder.close)."""
    lib().alacgpu_trim()


# ---- errors: errors.go:22-34 and internal/alac/errors.go:24-33 -------------------------------------
class AlacError(Exception):
    """Base of the package's errors; `sentinel` names the wrapped internal sentinel (errors.Is target)."""

    sentinel = None


class ErrConfig(AlacError):
    """errors.go:25 — invalid or unsupported configuration."""


class ErrDecode(AlacError):
    """errors.go:33 — failure during packet decoding; .status is the C ABI status word."""

    def __init__(self, msg, status=0, sentinel=None):
        super().__init__(msg)
        self.status = status
        self.sentinel = sentinel


class HipError(RuntimeError):
    """The HIP runtime or the extension failed (no reference counterpart)."""


ErrInvalidCookie = "alac: invalid magic cookie"
ErrUnsupportedVersion = "alac: unsupported compatible version"
ErrUnsupportedElement = "alac: unsupported element type (CCE/PCE)"
ErrInvalidHeader = "alac: invalid frame header"
ErrInvalidShift = "alac: invalid bytesShifted value"
ErrBitstreamOverrun = "alac: bitstream overrun"
ErrSampleOverrun = "alac: sample count exceeds buffer"
ErrBitDepth = "alac: unsupported bit depth"
ErrMalformed = "alac: malformed packet (the reference panics)"
ErrRange = "alac: packet outside the blob"

_CODE_SENTINEL = {1: ErrBitstreamOverrun, 2: ErrSampleOverrun, 3: ErrInvalidHeader, 4: ErrInvalidShift,
                  5: ErrUnsupportedElement, 6: ErrMalformed, 7: ErrRange}
_CTX = {0: None, 1: "SCE/LFE", 2: "CPE", 3: "DSE", 4: "FIL"}
_STAGE = {0: None, 1: "entropy decode", 2: "entropy decode U", 3: "entropy decode V"}


def status_error(status):
    """Rebuild the reference's error chain text from a status word (decoder.go:144-189,303,468,482)."""
    code, ctx, stage = status & 0xff, (status >> 8) & 0xf, (status >> 12) & 0x3
    parts = ["decode failed"]
    if _CTX.get(ctx):
        parts.append(_CTX[ctx])
    if _STAGE.get(stage):
        parts.append(_STAGE[stage])
    sentinel = _CODE_SENTINEL.get(code, "alac: unknown status %d" % code)
    parts.append(sentinel)
    return ErrDecode(": ".join(parts), status=status, sentinel=sentinel)


# ---- data contract ------------------------------------------------------------------------------------
class PacketConfig(ctypes.Structure):
    """PacketConfig (config.go:27-38) as the POD alacgpu_config."""

    _fields_ = [
        ("FrameLength", ctypes.c_uint32),
        ("BitDepth", ctypes.c_uint8),
        ("NumChannels", ctypes.c_uint8),
        ("PB", ctypes.c_uint8),
        ("MB", ctypes.c_uint8),
        ("KB", ctypes.c_uint8),
        ("_reserved0", ctypes.c_uint8),
        ("MaxRun", ctypes.c_uint16),
        ("MaxFrameBytes", ctypes.c_uint32),
        ("AvgBitRate", ctypes.c_uint32),
        ("SampleRate", ctypes.c_uint32),
    ]

    def __init__(self, FrameLength=4096, BitDepth=16, NumChannels=2, PB=40, MB=10, KB=14, MaxRun=255,
                 MaxFrameBytes=0, AvgBitRate=0, SampleRate=44100):
        super().__init__(FrameLength, BitDepth, NumChannels, PB, MB, KB, 0, MaxRun, MaxFrameBytes, AvgBitRate,
                         SampleRate)

    # C-side field names (alacgpu_config) as read-only aliases
    frame_length = property(lambda s: s.FrameLength)
    bit_depth = property(lambda s: s.BitDepth)
    num_channels = property(lambda s: s.NumChannels)
    pb = property(lambda s: s.PB)
    mb = property(lambda s: s.MB)
    kb = property(lambda s: s.KB)
    max_run = property(lambda s: s.MaxRun)
    max_frame_bytes = property(lambda s: s.MaxFrameBytes)
    avg_bit_rate = property(lambda s: s.AvgBitRate)
    sample_rate = property(lambda s: s.SampleRate)


class PCMFormat(ctypes.Structure):
    """PCMFormat (format.go:20-24)."""

    _fields_ = [("SampleRate", ctypes.c_int32), ("BitDepth", ctypes.c_int32), ("Channels", ctypes.c_int32)]

    def __repr__(self):
        return "PCMFormat(SampleRate=%d, BitDepth=%d, Channels=%d)" % (self.SampleRate, self.BitDepth, self.Channels)


def ParseMagicCookie(cookie):
    """ParseMagicCookie (config.go:47-81): 24-byte ALACSpecificConfig, optional 'frma'/'alac' wrappers."""
    data = bytes(cookie or b"")
    if len(data) >= 12 and data[4:8] == b"frma":
        data = data[12:]
    if len(data) >= 12 and data[4:8] == b"alac":
        data = data[12:]
    if len(data) < 24:
        e = ErrConfig("invalid configuration: " + ErrInvalidCookie)
        e.sentinel = ErrInvalidCookie
        raise e
    if data[4] > 0:
        e = ErrConfig("invalid configuration: %s: %d" % (ErrUnsupportedVersion, data[4]))
        e.sentinel = ErrUnsupportedVersion
        raise e
    be = lambda b: int.from_bytes(b, "big")  # noqa: E731
    return PacketConfig(FrameLength=be(data[0:4]), BitDepth=data[5], PB=data[6], MB=data[7], KB=data[8],
                        NumChannels=data[9], MaxRun=be(data[10:12]), MaxFrameBytes=be(data[12:16]),
                        AvgBitRate=be(data[16:20]), SampleRate=be(data[20:24]))


# ---- native library --------------------------------------------------------------------------------------
def lib_path():
    # ALACGPU_LIB: another build of the same library (kernel A/B experiments under profiles/)
    return os.environ.get("ALACGPU_LIB") or os.path.join(_CSRC, "libalacgpu.so")


def build(force=False):
    """Compile the translation units of csrc/ for gfx950 (hipcc cross-compiles without a GPU) and link libalacgpu.so."""
    so = lib_path()
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h", ".inc"))] + [
        os.path.join(_HERE, "..", "include", "alacgpu.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        jobs = str(max(1, min(6, os.cpu_count() or 1)))
        subprocess.check_call(["make", "-C", _CSRC, "-j", jobs, "libalacgpu.so"], stdout=subprocess.DEVNULL)
    return so


def csrc_sha256():
    """Fingerprint of the kernel sources (csrc/*.hip, *.h, *.inc): profiles/*/traffic.json is stamped with it."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(_CSRC)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(_CSRC, f), "rb").read())
    return h.hexdigest()


class Dispatch(ctypes.Structure):
    """alacgpu_dispatch (include/alacgpu.h)."""

    _fields_ = [("packets_per_slot", ctypes.c_uint32), ("slots", ctypes.c_uint32), ("irregular_slots", ctypes.c_uint32),
                ("wide_slots", ctypes.c_uint32), ("narrow_slots", ctypes.c_uint32), ("keys", ctypes.c_uint32),
                ("gated", ctypes.c_uint32), ("lanes_per_packet", ctypes.c_uint32), ("narrow_kernel", ctypes.c_char * 32),
                ("wide_kernel", ctypes.c_char * 32), ("irregular_kernels", ctypes.c_char * 96), ("workgroups_per_cu", ctypes.c_uint32)]


_EXPORTS = {
    "alacgpu_create": (ctypes.c_int, [ctypes.POINTER(PacketConfig), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "alacgpu_destroy": (None, [ctypes.c_void_p]),
    "alacgpu_trim": (None, []),
    "alacgpu_host_alloc": (ctypes.c_void_p, [ctypes.c_size_t]),
    "alacgpu_host_free": (None, [ctypes.c_void_p]),
    "alacgpu_last_dispatch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "alacgpu_get_format": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(PCMFormat)]),
    "alacgpu_frame_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "alacgpu_decode_packet": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                             ctypes.POINTER(ctypes.c_int32)]),
    "alacgpu_decode_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                            ctypes.c_void_p]),
    "alacgpu_decode_batch_start": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                  ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                  ctypes.c_void_p]),
    "alacgpu_decode_batch_wait": (ctypes.c_int, [ctypes.c_void_p]),
    "alacgpu_decode_batch_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "alacgpu_reserve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "alacgpu_last_kernel_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "alacgpu_timing_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "alacgpu_kernel_times": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_size_t)]),
    "alacgpu_pair_placement": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.POINTER(ctypes.c_size_t)]),
    "alacgpu_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "alacgpu_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "alacgpu_last_error": (ctypes.c_char_p, []),
    "alacgpu_version": (ctypes.c_char_p, []),
}


def lib():
    """Load libalacgpu.so. Raises if the HIP extension has not been built: there is no fallback."""
    global _LIB
    if _LIB is None:
        so = lib_path()
        if not os.path.exists(so):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); this package has no CPU decode path" % so)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
        # /opt/rocm's). If torch is going to be used in this process it must be loaded FIRST so that
        # libalacgpu.so binds to the runtime torch initialises; loaded the other way round the process
        # ends up with two runtimes and torch.cuda.is_available() turns False. The library only needs
        # the stable hip_4.2 symbol set, so either runtime serves it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(so)
        for name, (res, args) in _EXPORTS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def _check(rc):
    if rc == 0:
        return
    msg = (lib().alacgpu_last_error() or b"").decode("utf-8", "replace")
    if rc == -1:
        e = ErrConfig(msg or "invalid configuration")
        e.sentinel = ErrBitDepth if "bit depth" in msg else None
        raise e
    if rc == -2:
        raise ValueError(msg or "bad argument")
    raise HipError(msg or "HIP failure %d" % rc)


def bytes_per_sample(depth):
    """BytesPerSample (internal/alac/format.go:23-34)."""
    try:
        return {16: 2, 20: 3, 24: 3, 32: 4}[depth]
    except KeyError:
        raise ValueError("alac: BytesPerSample called with unsupported bit depth %d" % depth)


# ---- PacketDecoder (decoder.go:79-128) ------------------------------------------------------------------
class PacketDecoder:
    """Decodes ALAC packets into interleaved LE signed PCM on one MI355X (decoder.go:79).

    Like the reference's, a PacketDecoder is single-caller. It is bound to one HIP device and one
    stream; multi-GPU callers make one decoder per device (see parallel.py).
    """

    def __init__(self, config, device=0):
        self._h = ctypes.c_void_p()
        self._lib = lib()
        self.config = config
        _check(self._lib.alacgpu_create(ctypes.byref(config), device, ctypes.byref(self._h)))
        self.device = device
        self.frame_bytes = self._lib.alacgpu_frame_bytes(self._h)

    def close(self):
        """alacgpu_destroy: the handle's streams, events and small buffers go to a per-process pool for the next decoder on
        this device (at most 128 MB of device memory and 64 MB of pinned memory per pooled handle, four per device);
        trim() gives that back too."""
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.alacgpu_destroy(self._h)
            self._h = ctypes.c_void_p()

    @staticmethod
    def trim():
        """alacgpu_trim: free what destroyed handles left in the pool (every device)."""
        lib().alacgpu_trim()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def Format(self):
        """(*PacketDecoder).Format (decoder.go:112)."""
        f = PCMFormat()
        _check(self._lib.alacgpu_get_format(self._h, ctypes.byref(f)))
        return f

    def DecodePacket(self, packet):
        """(*PacketDecoder).DecodePacket (decoder.go:117): one packet -> PCM bytes; raises ErrDecode."""
        packet = bytes(packet)
        out = np.empty(max(self.frame_bytes, 1), dtype=np.uint8)
        n = ctypes.c_size_t()
        st = ctypes.c_int32()
        buf = np.frombuffer(packet, dtype=np.uint8) if packet else np.zeros(1, np.uint8)
        rc = self._lib.alacgpu_decode_packet(self._h, buf.ctypes.data, len(packet), out.ctypes.data, out.size,
                                             ctypes.byref(n), ctypes.byref(st))
        if rc == -4:
            raise status_error(st.value)
        _check(rc)
        return out[:n.value].tobytes()

    def DecodePackets(self, packets):
        """New batch entry: list of packets -> (list of PCM bytes or ErrDecode per packet)."""
        packets = [bytes(p) for p in packets]
        n = len(packets)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        if n:
            offsets[1:] = np.cumsum([len(p) for p in packets], dtype=np.uint64)
        blob = np.frombuffer(b"".join(packets) or b"\0", dtype=np.uint8)
        out, frames, status = self.decode_batch(blob, offsets)
        bpf = self.config.NumChannels * bytes_per_sample(self.config.BitDepth)
        res = []
        for i in range(n):
            res.append(status_error(int(status[i])) if status[i] else out[i, :int(frames[i]) * bpf].tobytes())
        return res

    def decode_batch(self, blob, offsets, out_stride=None):
        """alacgpu_decode_batch: host blob + offsets[n+1] -> (out[n, stride] uint8, frames, status). The offsets may
        come from an untrusted sample table: packets that leave the blob get ALACGPU_ERR_RANGE and are never read."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        stride = out_stride or self.frame_bytes
        # (empty, not zeros: the entry itself zeroes failing packets' slots and the bytes behind partial frames, and touching
        # 50 MB twice is a third of a long file's decode time)
        out = np.empty((max(n, 0), stride), dtype=np.uint8)
        frames = np.zeros(max(n, 0), dtype=np.uint32)
        status = np.zeros(max(n, 0), dtype=np.int32)
        if n > 0:
            _check(self._lib.alacgpu_decode_batch(self._h, blob.ctypes.data, blob.size, offsets.ctypes.data, n,
                                                  out.ctypes.data, stride, frames.ctypes.data, status.ctypes.data))
        return out, frames, status

    def decode_batch_start(self, blob, offsets, out_stride=None):
        """alacgpu_decode_batch_start: the same decode on a thread of the library's own (no Python thread, no GIL to fight
        for). -> a token for decode_batch_wait; nothing else may be called on this decoder until then."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        stride = out_stride or self.frame_bytes
        out = np.empty((max(n, 0), stride), dtype=np.uint8)
        frames = np.zeros(max(n, 0), dtype=np.uint32)
        status = np.zeros(max(n, 0), dtype=np.int32)
        if n > 0:
            _check(self._lib.alacgpu_decode_batch_start(self._h, blob.ctypes.data, blob.size, offsets.ctypes.data, n,
                                                        out.ctypes.data, stride, frames.ctypes.data, status.ctypes.data))
        return (blob, offsets, out, frames, status, n > 0)  # the token keeps every buffer alive

    def decode_batch_wait(self, token):
        """-> (out, frames, status) of the decode decode_batch_start began; raises what decode_batch would have."""
        if token[5]:
            _check(self._lib.alacgpu_decode_batch_wait(self._h))
        return token[2], token[3], token[4]

    def decode_batch_device(self, d_blob, blob_bytes, d_offsets, d_sizes, n, d_out, out_stride, d_frames, d_status,
                            sync=True):
        """alacgpu_decode_batch_device: raw device pointers (ints), e.g. torch tensors' data_ptr(); blob_bytes =
        readable bytes at d_blob (packets may lie densely); d_sizes may be None (offsets then has n+1 entries).
        The handle's stream does not order against torch's: synchronize the inputs first."""
        _check(self._lib.alacgpu_decode_batch_device(self._h, d_blob, blob_bytes, d_offsets, d_sizes, n, d_out,
                                                     out_stride, d_frames, d_status, 1 if sync else 0))

    def reserve(self, n_packets):
        _check(self._lib.alacgpu_reserve(self._h, n_packets))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        _check(self._lib.alacgpu_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def timing_reset(self):
        _check(self._lib.alacgpu_timing_reset(self._h))

    def kernel_times_ms(self, max_n=64):
        """Durations (ms) of the most recent decode kernel launches, HIP events on the handle's stream."""
        ms = np.zeros(max_n, dtype=np.float32)
        got = ctypes.c_size_t()
        _check(self._lib.alacgpu_kernel_times(self._h, ms.ctypes.data, max_n, ctypes.byref(got)))
        return ms[:got.value].copy()

    def pair_placement(self, max_slots=1 << 16):
        """Diagnostics of the last device decode (include/alacgpu.h: alacgpu_pair_placement): an (n_slots, 4) uint32
        array, one row per wave slot of the launch plan — [tag, clock at start, clock at end, 0]; all zero for slots
        that no gated wave pair decoded."""
        raw = np.zeros((max_slots, 4), dtype=np.uint32)
        got = ctypes.c_size_t()
        _check(self._lib.alacgpu_pair_placement(self._h, raw.ctypes.data, raw.size, ctypes.byref(got)))
        return raw[:got.value].copy()

    def last_dispatch(self):
        """What the last device decode dispatched, read back from the plan the device built (include/alacgpu.h:
        alacgpu_last_dispatch) -> dict."""
        d = Dispatch()
        _check(self._lib.alacgpu_last_dispatch(self._h, ctypes.byref(d)))
        out = {k: int(getattr(d, k)) for k, _ in Dispatch._fields_[:8]}
        out.update({k: getattr(d, k).decode() for k in ("narrow_kernel", "wide_kernel", "irregular_kernels")})
        out["workgroups_per_cu"] = int(d.workgroups_per_cu)
        return out

    def synchronize(self):
        _check(self._lib.alacgpu_synchronize(self._h))


def NewPacketDecoder(config, device=0):
    """NewPacketDecoder (decoder.go:90): raises ErrConfig for bit depths outside {16,20,24,32}."""
    return PacketDecoder(config, device)


def NewDecoder(source, device=0, window=1024):
    """NewDecoder (decode.go:50-76): streaming façade over the batch path, see stream.py."""
    from . import stream
    return stream.NewDecoder(source, device=device, window=window)


def FindALACTrack(data):
    """internal/mp4 FindALACTrack (mp4.go:233-298) on a file in memory -> mp4.Track (cookie, offsets, sizes)."""
    from . import mp4
    return mp4.find_alac_track(data)
