import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle
pkg = importlib.import_module("saprobe-alac_amd"); synth = importlib.import_module("saprobe-alac_amd.synth")
depth, ch, fl = 16, 2, 4096
prof = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = oracle.make_config(fl, depth, ch)
b = synth.gen_batch(cfg, n, profile=prof, threads=8)
ref_out, ref_fr, ref_st = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
pc = pkg.PacketConfig(FrameLength=fl, BitDepth=depth, NumChannels=ch)
dec = pkg.NewPacketDecoder(pc, 0)
offs = np.concatenate([b.offsets.astype(np.uint64), np.array([b.offsets[-1] + b.sizes[-1]], np.uint64)])
out, fr, st = dec.decode_batch(b.blob, offs)
def bits(p, pos, nb):
    v = int.from_bytes(p[pos // 8: pos // 8 + 8].ljust(8, b"\0"), "big")
    return (v >> (64 - (pos % 8) - nb)) & ((1 << nb) - 1)
bad = 0
for i in range(n):
    p = bytes(b.packet(i))
    hdr = bits(p, 19, 4); pos = 23 + (32 if hdr >> 3 else 0) + 16
    hu = bits(p, pos, 16); nu = hu & 31; pos += 16 + 16 * nu; hv = bits(p, pos, 16); nv = hv & 31
    same = np.array_equal(out[i], ref_out[i]) and fr[i] == ref_fr[i] and st[i] == ref_st[i]
    if not same:
        bad += 1
        d = np.nonzero(out[i] != ref_out[i])[0]
        if bad <= 12:
            print("pkt %d nu %d nv %d frames %d/%d st %x/%x first diff byte %s (frame %s, chan %s) ndiff %d" % (
                i, nu, nv, fr[i], ref_fr[i], st[i] & 0xffffffff, ref_st[i] & 0xffffffff, d[:1], d[:1] // 4, (d[:1] % 4) // 2, len(d)))
print("bad", bad, "of", n)
