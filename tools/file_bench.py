#!/usr/bin/env python3
"""file_bench.py — file to PCM, in the shape of the reference's own benchmark (tests/benchmark_test.go:261-286,
benchDecodeSaprobe: NewDecoder on the .m4a, then Read with a 64 KiB buffer until EOF, best / median of 10 iterations),
on the files it uses (tests/benchmark_test.go:39-51: 44.1 kHz / 16-bit and 96 kHz / 24-bit stereo, 10 s and 300 s).

The files are written here (tests/m4a.py around this repo's encoder: music-like signal, not the reference's white noise,
which ALAC stores as escape packets, docs/QA.md:140-147). Two hosts are timed over the same C ABI:
  python   saprobe-alac_amd.stream.NewDecoder + Read(65536) loop
  c++      host/stream_decoder.hpp (alac::NewDecoder + Read) through tests/host_sim/host_shim.cpp
plus DecodePacket latency (one packet per call, decoder.go:117) p50 / p99 over 200 calls.
Prints one JSON line per file. Reference numbers for orientation (docs/QA.md:122-125, hardware unstated, white noise):
CD 10 s 4 ms, 96k/24 10 s 12 ms, CD 300 s 114 ms, 96k/24 300 s 346 ms; real music 32-39 Msamples/s (docs/QA.md:178-179)."""
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    pkg = importlib.import_module("saprobe-alac_amd")
    synth = importlib.import_module("saprobe-alac_amd.synth")
    import m4a
    from test_container import _build_shim
    iters = int(os.environ.get("FILE_BENCH_ITERS", "10"))
    shim = _build_shim(True, pkg)
    shim.shim_open.restype = ctypes.c_long
    shim.shim_read.restype = ctypes.c_long
    for name, rate, depth, seconds in (("CD 44.1k/16 10s", 44100, 16, 10), ("96k/24 10s", 96000, 24, 10),
                                       ("CD 44.1k/16 300s", 44100, 16, 300), ("96k/24 300s", 96000, 24, 300)):
        cfg = pkg.PacketConfig(FrameLength=4096, BitDepth=depth, NumChannels=2, SampleRate=rate)
        n = (rate * seconds + 4095) // 4096
        b = synth.gen_batch(cfg, n, threads=min(os.cpu_count() or 1, 32), want_pcm=True)
        data = m4a.write_m4a(cfg, [b.packet(i) for i in range(n)])
        bps = pkg.bytes_per_sample(depth)
        total = int(b.frames.astype(np.int64).sum()) * 2 * bps
        res = {"file": name, "packets": n, "file_MB": round(len(data) / 1e6, 2), "pcm_MB": round(total / 1e6, 2)}
        # ---- python façade
        times = []
        ok = True
        for it in range(iters + 1):
            t0 = time.perf_counter()
            dec = pkg.NewDecoder(data)
            got = 0
            h = __import__("hashlib").sha256() if it == 0 else None
            while True:
                chunk = dec.Read(65536)
                if not chunk:
                    break
                got += len(chunk)
                if h:
                    h.update(chunk)
            dec.close()
            dt = time.perf_counter() - t0
            if it:
                times.append(dt)
            else:  # first pass: correctness (and warm-up)
                exp = __import__("hashlib").sha256()
                for i in range(n):
                    exp.update(b.pcm[i, :int(b.frames[i]) * 2 * bps].tobytes())
                ok = ok and got == total and h.digest() == exp.digest()
        res["python_ms_median"] = round(float(np.median(times)) * 1e3, 2)
        res["python_ms_best"] = round(min(times) * 1e3, 2)
        if os.environ.get("FILE_BENCH_VERBOSE"):
            res["python_ms_all"] = [round(t * 1e3, 2) for t in times]
        # ---- C++ façade
        buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
        out = (ctypes.c_uint8 * 65536)()
        times = []
        for it in range(iters + 1):
            t0 = time.perf_counter()
            hnd = ctypes.c_void_p()
            rc = shim.shim_open(buf, ctypes.c_size_t(len(data)), ctypes.c_size_t(0), ctypes.byref(hnd))  # 0: the decoder's own window size
            assert rc == 0, shim.shim_last_error()
            got = 0
            while True:
                k = shim.shim_read(hnd, out, ctypes.c_size_t(65536))
                if k <= 0:
                    break
                got += k
            shim.shim_close(hnd)
            dt = time.perf_counter() - t0
            if it:
                times.append(dt)
            ok = ok and got == total
        res["cpp_ms_median"] = round(float(np.median(times)) * 1e3, 2)
        res["cpp_ms_best"] = round(min(times) * 1e3, 2)
        res["cpp_Msamples_per_s"] = round(total / bps / np.median(times) / 1e6, 1)
        res["bit_exact"] = bool(ok)
        print(json.dumps(res), flush=True)
    # ---- DecodePacket latency (BASELINE config a shape: one 16-bit stereo 4096-frame packet per call)
    cfg = pkg.PacketConfig(FrameLength=4096, BitDepth=16, NumChannels=2)
    b = synth.gen_batch(cfg, 64, threads=4)
    with pkg.NewPacketDecoder(cfg) as dec:
        lat = []
        for i in range(220):
            p = b.packet(i % 64)
            t0 = time.perf_counter()
            pcm = dec.DecodePacket(p)
            lat.append(time.perf_counter() - t0)
            assert pcm == b.pcm[i % 64, :int(b.frames[i % 64]) * 4].tobytes()
        lat = np.array(lat[20:]) * 1e3
        print(json.dumps({"DecodePacket_ms": {"p50": round(float(np.percentile(lat, 50)), 3),
                                              "p99": round(float(np.percentile(lat, 99)), 3), "n": len(lat)},
                          "note": "one packet is one serial chain of 8192 steps on the GPU; the reference's pure-Go "
                                  "path decodes it in ~0.23 ms on one core (docs/QA.md:178)"}), flush=True)


if __name__ == "__main__":
    main()
