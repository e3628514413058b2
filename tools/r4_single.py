#!/usr/bin/env python3
"""r4_single.py — kernel time of single packets of the benchmark stream, one decode each (a batch of one: one workgroup, one
live lane per wave), beside the packet's predictor orders: which packets are the slow ones of a small batch, and why.
usage: python tools/r4_single.py [first] [count]   (ALACGPU_LANES_MIN as usual)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("saprobe-alac_amd")
synth = importlib.import_module("saprobe-alac_amd.synth")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = pkg.PacketConfig(FrameLength=4096, BitDepth=16, NumChannels=2)
b = synth.gen_batch(cfg, count, first_index=first, threads=8)
dev = torch.device("cuda:0")
d_blob = torch.from_numpy(b.blob).to(dev)
d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
stride = 4096 * 4
d_out = torch.zeros((count, stride), dtype=torch.uint8, device=dev)
d_fr = torch.zeros(count, dtype=torch.int32, device=dev)
d_st = torch.zeros(count, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
rows = []
with pkg.NewPacketDecoder(cfg, 0) as dec:
    for i in range(count):
        ms = []
        for _ in range(4):
            dec.decode_batch_device(d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr() + 8 * i, d_sz.data_ptr() + 4 * i, 1,
                                    d_out.data_ptr() + stride * i, stride, d_fr.data_ptr() + 4 * i, d_st.data_ptr() + 4 * i, sync=True)
            ms.append(dec.last_kernel_ms())
        p = b.packet(i)
        bits = "".join(format(x, "08b") for x in p[:64])
        partial = int(bits[19])
        pos = 23 + 32 * partial + 16
        nu = int(bits[pos + 11:pos + 16], 2)
        den_u, pbf_u = int(bits[pos + 4:pos + 8], 2), int(bits[pos + 8:pos + 11], 2)
        pos2 = pos + 16 + 16 * nu
        nv = int(bits[pos2 + 11:pos2 + 16], 2)
        rows.append((min(ms[1:]), i, nu, nv, len(p), int(b.frames[i])))
ok = bool(int(d_st.abs().sum()) == 0 and torch.equal(d_out.cpu(), torch.from_numpy(b.pcm)))
rows.sort()
print("bit_exact", ok)
for ms, i, nu, nv, sz, fr in rows:
    print("packet %3d  %.4f ms  orders %2d %2d  %5d bytes  %d frames" % (i, ms, nu, nv, sz, fr))
