"""Synthetic ALAC packet generator (ctypes binding of synth/alac_synth.c).

Makes benchmark and test inputs: a seeded signal source plus the build's own ALAC encoder (the
reference is decode-only, README.md:35). Host-only; never on the decode path.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PROFILE_MUSIC, PROFILE_NOISE, PROFILE_QUIET, PROFILE_STRESS, PROFILE_MUSIC_LE8 = 0, 1, 2, 3, 4
PROFILE_MUSIC_NOSHIFT, PROFILE_MUSIC_MIXED = 5, 6
COEF_WARM, COEF_RANDOM, COEF_GIVEN = 0, 1, 2
FLAG_LEADING_FIL, FLAG_MID_DSE, FLAG_NO_END = 1, 2, 4
PACKET_PAD = 64
BASE_SEED = 0x5A9B0BE


class Elem(ctypes.Structure):
    _fields_ = [
        ("order_u", ctypes.c_uint8), ("order_v", ctypes.c_uint8), ("den_shift", ctypes.c_uint8),
        ("mode_u", ctypes.c_uint8), ("mode_v", ctypes.c_uint8), ("pb_factor", ctypes.c_uint8),
        ("mix_bits", ctypes.c_uint8), ("mix_res", ctypes.c_int8), ("bytes_shifted", ctypes.c_uint8),
        ("force_escape", ctypes.c_uint8), ("never_escape", ctypes.c_uint8), ("coef_mode", ctypes.c_uint8),
        ("partial", ctypes.c_uint8), ("pad0", ctypes.c_uint8 * 3),
        ("coefs_u", ctypes.c_int16 * 32), ("coefs_v", ctypes.c_int16 * 32), ("seed", ctypes.c_uint64),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libalac_synth.so")
    src = os.path.join(_HERE, "alac_synth.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libalac_synth.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libalac_synth.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.alac_synth_num_elements.argtypes = [ctypes.c_int]
        L.alac_synth_encode_packet.restype = sz
        L.alac_synth_encode_packet.argtypes = [vp, vp, vp, ctypes.c_uint32, ctypes.c_uint32, vp, sz]
        L.alac_synth_signal.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, vp]
        L.alac_synth_params.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, vp, vp, vp]
        L.alac_synth_pack_pcm.argtypes = [vp, vp, ctypes.c_uint32, vp]
        L.alac_synth_slot_bytes.restype = sz
        L.alac_synth_slot_bytes.argtypes = [vp]
        L.alac_synth_gen_batch.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, sz, sz, vp, sz, vp, vp, vp, sz,
                                           ctypes.c_int]
        L.alac_synth_compact.restype = sz
        L.alac_synth_compact.argtypes = [vp, sz, vp, sz, sz, vp]
        _LIB = L
    return _LIB


def num_elements(num_channels):
    return lib().alac_synth_num_elements(num_channels)


def default_elem(order=6, den_shift=9, mix_bits=2, mix_res=1, pb_factor=4, bytes_shifted=0, **kw):
    e = Elem()
    e.order_u = e.order_v = order
    e.den_shift, e.mix_bits, e.mix_res, e.pb_factor, e.bytes_shifted = den_shift, mix_bits, mix_res, pb_factor, bytes_shifted
    for k, v in kw.items():
        if k in ("coefs_u", "coefs_v"):
            arr = getattr(e, k)
            for i, c in enumerate(v):
                arr[i] = c
        else:
            setattr(e, k, v)
    return e


def signal(cfg, profile, seed, num_frames):
    pcm = np.zeros((max(num_frames, 1), cfg.num_channels), dtype=np.int32)
    lib().alac_synth_signal(ctypes.byref(cfg), profile, seed, num_frames, pcm.ctypes.data)
    return pcm[:num_frames]


def pack_pcm(cfg, pcm):
    """Expected decoder output (interleaved LE PCM bytes) of an int32 [frames][channels] block."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    bps = {16: 2, 20: 3, 24: 3, 32: 4}[cfg.bit_depth]
    out = np.zeros(max(pcm.size * bps, 1), dtype=np.uint8)
    lib().alac_synth_pack_pcm(ctypes.byref(cfg), pcm.ctypes.data, pcm.shape[0], out.ctypes.data)
    return out[:pcm.size * bps].tobytes()


def encode_packet(cfg, elems, pcm, flags=0):
    """pcm: int32 [frames][channels] in output channel order; elems: list of Elem in bitstream order."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    ne = num_elements(cfg.num_channels)
    if len(elems) != ne:
        raise ValueError("need %d element settings" % ne)
    arr = (Elem * ne)(*elems)
    cap = lib().alac_synth_slot_bytes(ctypes.byref(cfg))
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().alac_synth_encode_packet(ctypes.byref(cfg), arr, pcm.ctypes.data if pcm.size else None,
                                       pcm.shape[0], flags, out.ctypes.data, cap)
    if n == 0:
        raise RuntimeError("encode failed")
    return out[:n].tobytes()


class Batch:
    """A generated batch in the device blob layout of include/alacgpu.h."""

    def __init__(self, cfg, blob, offsets, sizes, frames, pcm, pcm_stride):
        self.cfg, self.blob, self.offsets, self.sizes, self.frames = cfg, blob, offsets, sizes, frames
        self.pcm, self.pcm_stride = pcm, pcm_stride

    @property
    def n(self):
        return len(self.sizes)

    @property
    def compressed_bytes(self):
        return int(self.sizes.astype(np.int64).sum())

    def packet(self, i):
        o = int(self.offsets[i])
        return self.blob[o:o + int(self.sizes[i])].tobytes()


def gen_batch(cfg, n, profile=PROFILE_MUSIC, base_seed=BASE_SEED, first_index=0, threads=None, want_pcm=True):
    """n packets of the seeded stream; returns a Batch (blob padded/aligned for the device entry)."""
    L = lib()
    threads = threads or min(os.cpu_count() or 1, 32)
    slot = L.alac_synth_slot_bytes(ctypes.byref(cfg))
    slots = np.empty(n * slot + 64, dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    frames = np.zeros(n, dtype=np.uint32)
    bps = {16: 2, 20: 3, 24: 3, 32: 4}[cfg.bit_depth]
    stride = cfg.frame_length * cfg.num_channels * bps
    pcm = np.zeros((n, stride), dtype=np.uint8) if want_pcm else None
    rc = L.alac_synth_gen_batch(ctypes.byref(cfg), profile, base_seed, first_index, n, slots.ctypes.data, slot,
                                sizes.ctypes.data, frames.ctypes.data, pcm.ctypes.data if want_pcm else None,
                                stride, threads)
    if rc != 0 or (n and int(sizes.min()) == 0):
        raise RuntimeError("packet generation failed")
    offsets = np.zeros(n, dtype=np.uint64)
    total = L.alac_synth_compact(slots.ctypes.data, slot, sizes.ctypes.data, n, PACKET_PAD, offsets.ctypes.data)
    blob = slots[:total + 64].copy()
    blob[total:] = 0
    return Batch(cfg, blob, offsets, sizes, frames, pcm, stride)
