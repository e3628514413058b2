"""fuzz_repro.py depth ch fl prof n ppw kb [mutate] [seed] — one configuration of tools/gpu_fuzz.py, differences in detail"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa
pkg = importlib.import_module("saprobe-alac_amd"); synth = importlib.import_module("saprobe-alac_amd.synth")
from oracle import oracle
from conftest import mutate_packets, pack_packets
depth, ch, fl, prof, n = map(int, sys.argv[1:6])
ppw, kb = sys.argv[6], int(sys.argv[7])
mut = int(sys.argv[8]) if len(sys.argv) > 8 else 0
seed = int(sys.argv[9]) if len(sys.argv) > 9 else 1
if ppw != "0":
    os.environ["ALACGPU_PPW"] = ppw
rng = np.random.default_rng(abs(seed))
cfg = oracle.make_config(fl, depth, ch, kb=kb)
b = synth.gen_batch(cfg, n, profile=prof, threads=8) if seed < 0 else synth.gen_batch(cfg, n, profile=prof, base_seed=int(rng.integers(1 << 30)), threads=8)
blob, offs, sizes = b.blob, b.offsets, b.sizes
if mut:
    blob, offs, sizes = pack_packets(mutate_packets(b, rng, n))
ref_out, ref_fr, ref_st = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
c = pkg.PacketConfig(FrameLength=fl, BitDepth=depth, NumChannels=ch, PB=cfg.pb, MB=cfg.mb, KB=cfg.kb, MaxRun=cfg.max_run, SampleRate=cfg.sample_rate)
with pkg.NewPacketDecoder(c, 0) as dec:
    pk = [bytes(blob[int(offs[i]):int(offs[i]) + int(sizes[i])]) for i in range(len(offs))]
    o = np.zeros(len(offs) + 1, np.uint64); o[1:] = np.cumsum([len(p) for p in pk])
    out, fr, st = dec.decode_batch(np.frombuffer(b"".join(pk) or b"\0", np.uint8), o)
def bits(p, pos, nb):
    v = int.from_bytes(p[pos // 8: pos // 8 + 8].ljust(8, b"\0"), "big")
    return (v >> (64 - (pos % 8) - nb)) & ((1 << nb) - 1)
bad = np.nonzero((st != ref_st) | (fr != ref_fr) | (out != ref_out).any(axis=1))[0]
print("n", len(pk), "bad", len(bad))
for i in bad[:16]:
    p = pk[i]
    hdr = bits(p, 19, 4); pos = 23 + (32 if hdr >> 3 else 0) + 16
    hu = bits(p, pos, 16); nu = hu & 31; pos2 = pos + 16 + 16 * nu; hv = bits(p, pos2, 16); nv = hv & 31
    print("pkt %d size %d tag %d hdr %x nu %d nv %d modeu %d: st %08x ref %08x frames %d ref %d pcmdiff %d" % (
        i, len(p), bits(p, 0, 3), hdr, nu, nv, hu >> 12, int(st[i]) & 0xffffffff, int(ref_st[i]) & 0xffffffff, fr[i], ref_fr[i], int((out[i] != ref_out[i]).sum())))
    print("   got", bytes(out[i][:12]).hex(), "ref", bytes(ref_out[i][:12]).hex(), "packet", p[:40].hex())
