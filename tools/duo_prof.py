#!/usr/bin/env python3
"""duo_prof.py — per-role cycle breakdown of the wave-pair decoder (profiling build, -DALAC_DUO_PROF).
usage: ALACGPU_LIB=profiles/exp_bin/duo_prof.so python tools/duo_prof.py [packets]"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("saprobe-alac_amd")
synth = importlib.import_module("saprobe-alac_amd.synth")
synth.build()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = pkg.PacketConfig(FrameLength=4096, BitDepth=DEPTH, NumChannels=2)
b = synth.gen_batch(cfg, P, profile=0, first_index=0, threads=16)
dev = torch.device("cuda", 0)
stride = 4096 * 2 * pkg.bytes_per_sample(DEPTH)
d_blob = torch.from_numpy(b.blob).to(dev)
d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
d_out = torch.zeros((P, stride), dtype=torch.uint8, device=dev)
d_fr = torch.zeros(P, dtype=torch.int32, device=dev)
d_st = torch.full((P,), -1, dtype=torch.int32, device=dev)
dec = pkg.NewPacketDecoder(cfg, 0)
dec.reserve(P)
L = ctypes.CDLL(pkg.lib_path())
buf = (ctypes.c_ulonglong * 32)()
def step():
    dec.decode_batch_device(d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr(), d_sz.data_ptr(), P, d_out.data_ptr(), stride,
                            d_fr.data_ptr(), d_st.data_ptr(), sync=True)
step(); step(); step()
L.alacgpu_debug_prof(buf)
v = list(buf)
names = ["A:fetch+golomb", "B:predict", "A/C:emit", "barrier wait"]
waves = (P + 63) // 64
for role, off in (("A", 0), ("B", 16), ("C", 8)):
    for phase, po in (("U phase", 0), ("last phase", 4)):
        tot = sum(v[off + po:off + po + 4])
        print("role %s, %s (ticks of s_memtime per wave, %d waves; %.1f per step of 4096):" % (role, phase, waves, tot / waves / 4096))
        for k in range(4):
            print("   %-16s %12.0f  %5.1f %%" % (names[k], v[off + po + k] / waves, 100.0 * v[off + po + k] / max(tot, 1)))
print("kernel ms:", dec.kernel_times_ms()[-1:])
