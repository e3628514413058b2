"""Debug helper: multi-chunk host entry vs the source PCM; prints where they differ."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("saprobe-alac_amd")
synth = importlib.import_module("saprobe-alac_amd.synth")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfg = pkg.PacketConfig(FrameLength=4096, BitDepth=16, NumChannels=2)
b = synth.gen_batch(cfg, P, threads=8)
pk_off = np.zeros(P + 1, np.uint64)
pk_off[1:] = np.cumsum(b.sizes.astype(np.uint64))
dense = np.empty(int(pk_off[-1]), np.uint8)
for i in range(P):
    o = int(b.offsets[i]); dense[int(pk_off[i]):int(pk_off[i + 1])] = b.blob[o:o + int(b.sizes[i])]
with pkg.NewPacketDecoder(cfg) as dec:
    for rep in range(3):
        out, fr, st = dec.decode_batch(dense, pk_off)
        bad_st = np.nonzero(st != 0)[0]
        bad_fr = np.nonzero(fr != b.frames)[0]
        rows = np.array([i for i in range(P) if not np.array_equal(out[i, :int(b.frames[i]) * 4], b.pcm[i, :int(b.frames[i]) * 4])])
        print("rep", rep, "status!=0:", len(bad_st), bad_st[:10], st[bad_st[:5]], "frames!=:", len(bad_fr), bad_fr[:10], "pcm rows differ:", len(rows), rows[:20])
