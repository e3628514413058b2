#!/usr/bin/env python3
"""pair_placement.py — how the wave pairs of the last decode were spread over the CUs (alacgpu_pair_placement):
pairs per CU, slots taken by a pair after its first, and the keys each CU got.

    python tools/pair_placement.py [--packets 65536 --depth 16 --channels 2 --profile 0] [lib.so]"""
import argparse
import collections
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(ROOT, "saprobe-alac_amd", "csrc", "libalacgpu.so"))
    ap.add_argument("--packets", type=int, default=65536)
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--frame-length", type=int, default=4096)
    ap.add_argument("--profile", type=int, default=0)
    ap.add_argument("--warm", type=int, default=12, help="decodes before the one that is looked at")
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("saprobe-alac_amd")
    synth = importlib.import_module("saprobe-alac_amd.synth")
    synth.build()
    P, FL = args.packets, args.frame_length
    cfg = pkg.PacketConfig(FrameLength=FL, BitDepth=args.depth, NumChannels=args.channels)
    stride = FL * args.channels * pkg.bytes_per_sample(args.depth)
    b = synth.gen_batch(cfg, P, profile=args.profile, threads=min(os.cpu_count() or 1, 32))
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(b.blob).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
    d_out = torch.zeros((P, stride), dtype=torch.uint8, device=dev)
    d_fr = torch.zeros(P, dtype=torch.int32, device=dev)
    d_st = torch.full((P,), -1, dtype=torch.int32, device=dev)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    L = ctypes.CDLL(os.path.abspath(args.lib))
    L.alacgpu_create.argtypes = [ctypes.POINTER(pkg.PacketConfig), ctypes.c_int, ctypes.POINTER(vp)]
    L.alacgpu_decode_batch_device.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, vp, vp, ctypes.c_int]
    L.alacgpu_pair_placement.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
    L.alacgpu_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    h = vp()
    assert L.alacgpu_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == 0
    for _ in range(args.warm):
        assert L.alacgpu_decode_batch_device(h, d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr(), d_sz.data_ptr(), P,
                                             d_out.data_ptr(), stride, d_fr.data_ptr(), d_st.data_ptr(), 1) == 0
    ms = ctypes.c_float()
    L.alacgpu_last_kernel_ms(h, ctypes.byref(ms))
    raw = np.zeros(4 * (P // 8 + 4096), np.uint32)
    got = sz()
    assert L.alacgpu_pair_placement(h, raw.ctypes.data, raw.size, ctypes.byref(got)) == 0
    raw = raw[:4 * got.value].reshape(-1, 4)
    tags = raw[:, 0]
    owned_rows = raw[tags != 0]
    owned = owned_rows[:, 0]
    cu = (owned >> 8) & 0x1ff
    slot = (owned >> 4) & 15
    swept = owned & 1
    per_cu = collections.Counter(cu.tolist())
    print("decode %.3f ms; wave slots %d, owned by pairs %d, CUs seen %d" % (ms.value, tags.size, owned.size, len(per_cu)))
    print("slots per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
    print("arrival number histogram:", sorted(collections.Counter(slot.tolist()).items()))
    print("slots taken by a pair after its first:", int(swept.sum()))
    sa, sb = (owned >> 17) & 3, (owned >> 19) & 3
    per_simd = collections.defaultdict(lambda: [0, 0])
    for c, a, b_ in zip(cu.tolist(), sa.tolist(), sb.tolist()):
        per_simd[(c, a)][0] += 1
        per_simd[(c, b_)][1] += 1
    print("SIMD loads (entropy waves, predictor waves) histogram:", sorted(collections.Counter(tuple(v) for v in per_simd.values()).items()))
    # timing: s_memtime ticks (about 2.1 GHz on MI355X), shown in thousands, relative to the earliest start
    first = int(np.argmax(tags != 0))
    t0 = owned_rows[:, 1].astype(np.int64)
    t1 = owned_rows[:, 2].astype(np.int64)
    base = t0.min()
    t0 = np.where(t0 == 0, t0.max(), t0)
    base = t0.min()
    beg, end = ((t0 - base) % (1 << 32)) / 1000.0, ((t1 - base) % (1 << 32)) / 1000.0
    idx = np.nonzero(tags != 0)[0] - first
    print("pairs (kiloticks): first start 0, last start %.0f, last end %.0f" % (beg.max(), end.max()))
    edges = np.linspace(0, idx.max() + 1, 9).astype(int)
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (idx >= lo) & (idx < hi)
        if m.any():
            print("  items %4d..%4d: duration mean %.0f max %.0f, end max %.0f" % (lo, hi - 1, (end - beg)[m].mean(), (end - beg)[m].max(), end[m].max()))
    late = np.argsort(-end)[:8]
    by_cu = collections.defaultdict(list)
    for k in range(owned.size):
        by_cu[int(cu[k])].append(k)
    print("the pairs that end last, with the others on their CU (item:arrival:entropy SIMD:predictor SIMD:end):")
    for k in late:
        mates = by_cu[int(cu[k])]
        print("  item %4d ends %.0f on CU %3d:" % (idx[k], end[k], cu[k]),
              " ".join("%d:%d:%d:%d:%.0f" % (idx[m], slot[m], sa[m], sb[m], end[m]) for m in mates))


if __name__ == "__main__":
    main()
