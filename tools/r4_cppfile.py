#!/usr/bin/env python3
"""r4_cppfile.py — where the C++ file decoder's time goes on a long 96 kHz / 24-bit file: copy speed out of pinned and
pageable memory, the host entry alone per window size, NewDecoder + Read loops at several window and Read sizes.
usage: python tools/r4_cppfile.py"""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("saprobe-alac_amd")
synth = importlib.import_module("saprobe-alac_amd.synth")
import m4a
from test_container import _build_shim
shim = _build_shim(True, pkg)
shim.shim_open.restype = ctypes.c_long; shim.shim_read.restype = ctypes.c_long
L = pkg.lib()
L.alacgpu_host_alloc.restype = ctypes.c_void_p; L.alacgpu_host_alloc.argtypes = [ctypes.c_size_t]
L.alacgpu_host_free.argtypes = [ctypes.c_void_p]
# (a) copy speed
nb = 64 << 20
p = L.alacgpu_host_alloc(nb)
pin = np.ctypeslib.as_array((ctypes.c_uint8 * nb).from_address(p)); pin[:] = 1
page = np.ones(nb, np.uint8); dst = np.empty(1 << 16, np.uint8); big = np.empty(nb, np.uint8)
for name, src in (("pinned", pin), ("pageable", page)):
    t0 = time.perf_counter(); big[:] = src; t1 = time.perf_counter()
    ctypes.memmove(big.ctypes.data, src.ctypes.data, nb); t2 = time.perf_counter()
    print("copy 64 MB out of %-8s memory: numpy %.2f ms (%.1f GB/s), memmove %.2f ms (%.1f GB/s)" % (name, (t1 - t0) * 1e3, nb / (t1 - t0) / 1e9, (t2 - t1) * 1e3, nb / (t2 - t1) / 1e9), flush=True)
del pin; L.alacgpu_host_free(p)
cfg = pkg.PacketConfig(FrameLength=4096, BitDepth=24, NumChannels=2, SampleRate=96000)
n = 7032
b = synth.gen_batch(cfg, n, threads=16, want_pcm=False)
pk = [b.packet(i) for i in range(n)]
data = m4a.write_m4a(cfg, pk)
buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
if os.environ.get("TRACE"):
    import subprocess
    d = os.path.join(ROOT, "tests", "host_sim"); so = os.path.join(d, "libhost_shim_trace.so"); libdir = os.path.dirname(pkg.lib_path())
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-pthread", "-shared", "-DALAC_STREAM_TRACE", "-DSHIM_WITH_DECODER", "-o", so,
                           os.path.join(d, "host_shim.cpp"), "-L" + libdir, "-lalacgpu", "-Wl,-rpath," + libdir])
    tr = ctypes.CDLL(so); tr.shim_open.restype = ctypes.c_long; tr.shim_read.restype = ctypes.c_long
    out = (ctypes.c_uint8 * 65536)()
    for it in range(3):
        sys.stderr.write("---- pass %d\n" % it); sys.stderr.flush()
        t0 = time.perf_counter()
        h = ctypes.c_void_p()
        assert tr.shim_open(buf, ctypes.c_size_t(len(data)), ctypes.c_size_t(1024), ctypes.byref(h)) == 0
        while tr.shim_read(h, out, ctypes.c_size_t(65536)) > 0:
            pass
        tr.shim_close(h)
        sys.stderr.write("---- pass %d took %.2f ms\n" % (it, (time.perf_counter() - t0) * 1e3)); sys.stderr.flush()
    sys.exit(0)
for window in (512, 1024, 2048, 4096, 8192):
    for rd in (65536, 1 << 20):
        out = (ctypes.c_uint8 * rd)()
        ts, topen = [], []
        for it in range(5):
            t0 = time.perf_counter()
            h = ctypes.c_void_p()
            assert shim.shim_open(buf, ctypes.c_size_t(len(data)), ctypes.c_size_t(window), ctypes.byref(h)) == 0
            t1 = time.perf_counter()
            got = 0
            while True:
                k = shim.shim_read(h, out, ctypes.c_size_t(rd))
                if k <= 0: break
                got += k
            shim.shim_close(h)
            ts.append(time.perf_counter() - t0); topen.append(t1 - t0)
        print("C++ window %5d Read %7d: median %.2f ms best %.2f (open %.2f ms) %d MB" % (window, rd, float(np.median(ts[1:])) * 1e3, min(ts[1:]) * 1e3, float(np.median(topen[1:])) * 1e3, got >> 20), flush=True)
offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(x) for x in pk])
blob = np.frombuffer(b"".join(pk), np.uint8)
for w in (1024, 4096, 7032):
    with pkg.NewPacketDecoder(cfg) as dec:
        ts = []
        for it in range(5):
            t0 = time.perf_counter()
            for lo in range(0, n, w):
                hi = min(n, lo + w)
                dec.decode_batch(blob[int(offs[lo]):int(offs[hi])], offs[lo:hi + 1] - offs[lo])
            ts.append(time.perf_counter() - t0)
        print("decode_batch (pageable in and out) in windows of %4d: best %.2f ms median %.2f ms (whole file)" % (w, min(ts) * 1e3, float(np.median(ts)) * 1e3), flush=True)
for w in (1024, 4096):
    ts = []
    for it in range(5):
        t0 = time.perf_counter()
        d = pkg.NewDecoder(data, window=w)
        while d.Read(65536):
            pass
        d.close()
        ts.append(time.perf_counter() - t0)
    print("python NewDecoder + Read loop, window %4d: best %.2f ms median %.2f ms" % (w, min(ts) * 1e3, float(np.median(ts)) * 1e3), flush=True)
