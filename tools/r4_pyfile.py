#!/usr/bin/env python3
"""r4_pyfile.py — where the Python file decoder's time goes: the host entry alone (pageable numpy buffers) at several
window sizes, and the whole NewDecoder + Read(64 KiB) loop. usage: python tools/r4_pyfile.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
pkg = importlib.import_module("saprobe-alac_amd")
synth = importlib.import_module("saprobe-alac_amd.synth")
import m4a
cfg = pkg.PacketConfig(FrameLength=4096, BitDepth=16, NumChannels=2, SampleRate=44100)
n = 3230
b = synth.gen_batch(cfg, n, threads=16, want_pcm=False)
pk = [b.packet(i) for i in range(n)]
data = m4a.write_m4a(cfg, pk)
offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(p) for p in pk])
blob = np.frombuffer(b"".join(pk), np.uint8)
for w in (256, 1024, 3230):
    with pkg.NewPacketDecoder(cfg) as dec:
        ts = []
        for it in range(6):
            t0 = time.perf_counter()
            for lo in range(0, n, w):
                hi = min(n, lo + w)
                dec.decode_batch(blob[int(offs[lo]):int(offs[hi])], offs[lo:hi + 1] - offs[lo])
            ts.append(time.perf_counter() - t0)
        print("decode_batch in windows of %4d: best %.2f ms median %.2f ms (whole file)" % (w, min(ts) * 1e3, float(np.median(ts)) * 1e3), flush=True)
for w in (1024, 4096):
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        d = pkg.NewDecoder(data, window=w)
        while d.Read(65536):
            pass
        d.close()
        ts.append(time.perf_counter() - t0)
    print("NewDecoder + Read loop, window %4d: best %.2f ms median %.2f ms" % (w, min(ts) * 1e3, float(np.median(ts)) * 1e3), flush=True)
