#!/usr/bin/env python3
"""ab_bench.py — A/B of several builds of libalacgpu.so in ONE process on ONE device (devices differ by up to 12 % on
compute-bound kernels, so builds are only comparable inside a run): the builds take turns, `--rounds` times, on the same
device-resident batch; prints the per-build median of the HIP-event decode time.

    python tools/ab_bench.py [--packets 65536 --depth 16 --channels 2 --profile 0] libA.so libB.so ...

A build may be given as path@NAME=value[,NAME=value]: those environment variables are set while its handle is made (the
library reads ALACGPU_SIDE, ALACGPU_LANES_MIN, ... per handle), so one binary can be compared with itself under two settings.

Builds with the round-1 ABI (alacgpu 0.2.x: no blob_bytes argument, 64 zero bytes behind every packet) are driven
through their own signature; the batch is laid out with padding so both kinds can read it."""
import argparse
import ctypes
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--packets", type=int, default=65536)
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--frame-length", type=int, default=4096)
    ap.add_argument("--profile", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("saprobe-alac_amd")
    synth = importlib.import_module("saprobe-alac_amd.synth")
    synth.build()
    P, FL, depth, ch = args.packets, args.frame_length, args.depth, args.channels
    cfg = pkg.PacketConfig(FrameLength=FL, BitDepth=depth, NumChannels=ch)
    bps = pkg.bytes_per_sample(depth)
    stride = FL * ch * bps
    b = synth.gen_batch(cfg, P, profile=args.profile, threads=min(os.cpu_count() or 1, 32))
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(b.blob).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
    d_out = torch.zeros((P, stride), dtype=torch.uint8, device=dev)
    d_fr = torch.zeros(P, dtype=torch.int32, device=dev)
    d_st = torch.full((P,), -1, dtype=torch.int32, device=dev)
    pcm = torch.from_numpy(b.pcm).to(dev)
    torch.cuda.synchronize()
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    builds = []
    for spec in args.libs:
        path, _, envs = spec.partition("@")
        saved = {}
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        L = ctypes.CDLL(os.path.abspath(path))
        L.alacgpu_version.restype = ctypes.c_char_p
        ver = L.alacgpu_version().decode()
        old = " 0.2." in ver
        L.alacgpu_create.argtypes = [ctypes.POINTER(pkg.PacketConfig), ctypes.c_int, ctypes.POINTER(vp)]
        L.alacgpu_decode_batch_device.argtypes = ([vp, vp, vp, vp, sz, vp, sz, vp, vp, ctypes.c_int] if old else
                                                  [vp, vp, sz, vp, vp, sz, vp, sz, vp, vp, ctypes.c_int])
        L.alacgpu_kernel_times.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
        L.alacgpu_timing_reset.argtypes = [vp]
        L.alacgpu_reserve.argtypes = [vp, sz]
        L.alacgpu_destroy.argtypes = [vp]
        h = vp()
        assert L.alacgpu_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == 0
        L.alacgpu_reserve(h, P)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        builds.append({"path": spec, "version": ver, "L": L, "h": h, "old": old, "ms": []})

    def run(bd, steps):
        L, h = bd["L"], bd["h"]
        L.alacgpu_timing_reset(h)
        for _ in range(steps):
            a = [h, d_blob.data_ptr()] + ([] if bd["old"] else [d_blob.numel()]) + [
                d_off.data_ptr(), d_sz.data_ptr(), P, d_out.data_ptr(), stride, d_fr.data_ptr(), d_st.data_ptr(), 1]
            assert L.alacgpu_decode_batch_device(*a) == 0
        ms = np.zeros(64, np.float32)
        got = sz()
        L.alacgpu_kernel_times(h, ms.ctypes.data, 64, ctypes.byref(got))
        return ms[:got.value]

    for bd in builds:  # warm-up + correctness of every build
        d_st.fill_(-1)
        run(bd, 2)
        full = torch.from_numpy(b.frames.astype(np.int64) == FL).to(dev)
        bd["bit_exact"] = bool(int(d_st.abs().sum()) == 0 and torch.equal(d_out[full], pcm[full]))
    for _ in range(args.rounds):
        for bd in builds:
            bd["ms"] += list(run(bd, args.steps))
    for bd in builds:
        m = np.array(bd["ms"])
        print(json.dumps({"lib": bd["path"], "version": bd["version"], "median_ms": round(float(np.median(m)), 4),
                          "min_ms": round(float(m.min()), 4), "max_ms": round(float(m.max()), 4), "n": len(m),
                          "bit_exact": bd["bit_exact"], "all_ms": [round(float(x), 3) for x in m] if os.environ.get("AB_DUMP") else None, "note": "0.2.x brackets exclude the sort pre-pass (~20 us)" if bd["old"] else ""}))
        bd["L"].alacgpu_destroy(bd["h"])


if __name__ == "__main__":
    main()
