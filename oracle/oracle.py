"""ctypes binding of the CPU oracle (oracle/alac_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Config(ctypes.Structure):
    """POD mirror of PacketConfig (config.go:27-38) == alacgpu_config (include/alacgpu.h)."""

    _fields_ = [
        ("frame_length", ctypes.c_uint32),
        ("bit_depth", ctypes.c_uint8),
        ("num_channels", ctypes.c_uint8),
        ("pb", ctypes.c_uint8),
        ("mb", ctypes.c_uint8),
        ("kb", ctypes.c_uint8),
        ("reserved0", ctypes.c_uint8),
        ("max_run", ctypes.c_uint16),
        ("max_frame_bytes", ctypes.c_uint32),
        ("avg_bit_rate", ctypes.c_uint32),
        ("sample_rate", ctypes.c_uint32),
    ]


def make_config(frame_length=4096, bit_depth=16, num_channels=2, pb=40, mb=10, kb=14, max_run=255,
                sample_rate=44100):
    return Config(frame_length, bit_depth, num_channels, pb, mb, kb, 0, max_run, 0, 0, sample_rate)


def build(force=False):
    so = os.path.join(_HERE, "libalac_oracle.so")
    src = os.path.join(_HERE, "alac_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libalac_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libalac_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.alac_oracle_create.restype = ctypes.c_void_p
        L.alac_oracle_create.argtypes = [ctypes.POINTER(Config)]
        L.alac_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.alac_oracle_frame_bytes.restype = ctypes.c_size_t
        L.alac_oracle_frame_bytes.argtypes = [ctypes.c_void_p]
        L.alac_oracle_decode_packet.restype = ctypes.c_int32
        L.alac_oracle_decode_packet.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        L.alac_oracle_decode_batch.restype = ctypes.c_int
        L.alac_oracle_decode_batch.argtypes = [ctypes.POINTER(Config), ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                               ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _LIB = L
    return _LIB


def bytes_per_sample(depth):
    return {16: 2, 20: 3, 24: 3, 32: 4}[depth]


def frame_bytes(cfg):
    return cfg.frame_length * cfg.num_channels * bytes_per_sample(cfg.bit_depth)


def decode_packet(cfg, packet):
    """-> (status, frames, pcm bytes trimmed like DecodePacket's output[:n])."""
    L = lib()
    d = L.alac_oracle_create(ctypes.byref(cfg))
    if not d:
        raise ValueError("unsupported config")
    try:
        n = L.alac_oracle_frame_bytes(d)
        out = np.zeros(max(n, 1), dtype=np.uint8)
        fr = ctypes.c_uint32()
        pkt = np.frombuffer(bytes(packet), dtype=np.uint8) if len(packet) else np.zeros(1, np.uint8)
        st = L.alac_oracle_decode_packet(d, pkt.ctypes.data, len(packet), out.ctypes.data, ctypes.byref(fr))
        nb = fr.value * cfg.num_channels * bytes_per_sample(cfg.bit_depth)
        return st, fr.value, out[:nb].tobytes() if st == 0 else b""
    finally:
        L.alac_oracle_destroy(d)


def decode_batch(cfg, blob, offsets, sizes, out_stride=None, threads=1, want_output=True):
    """blob: uint8 array; offsets uint64[n]; sizes uint32[n]. -> (out[n, stride] or None, frames, status)."""
    L = lib()
    n = len(offsets)
    stride = out_stride or frame_bytes(cfg)
    out = np.zeros((n, stride), dtype=np.uint8) if want_output else None
    frames = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    L.alac_oracle_decode_batch(ctypes.byref(cfg), blob.ctypes.data, offsets.ctypes.data, sizes.ctypes.data, n,
                               out.ctypes.data if want_output else None, stride, frames.ctypes.data,
                               status.ctypes.data, threads)
    return out, frames, status
