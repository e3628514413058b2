#!/usr/bin/env python3
"""gpu_fuzz.py — one-off randomized parity sweep on the GPU: random frame lengths (now and then 8 192 .. 70 000 frames),
depths, channel counts, signal profiles, batch sizes, wave widths, workgroups per CU and cookie bytes KB / PB / MB (config.go:72-74), intact,
`loud` and corrupted packets, HIP path vs oracle through the C ABI.
usage: python tools/gpu_fuzz.py [rounds] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
DRY = os.environ.get("ALACGPU_FUZZ_DRY") == "1"  # generator and oracle only (no GPU): the same rounds, to tell a fault of theirs apart
if not DRY:
    import torch  # noqa: F401  (one HIP runtime per process: torch first)
pkg = importlib.import_module("saprobe-alac_amd")
synth = importlib.import_module("saprobe-alac_amd.synth")
from oracle import oracle
from conftest import mutate_packets, pack_packets, assert_same_decode, loud_packets
synth.build(); oracle.build()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for r in range(rounds):
    depth = int(rng.choice([16, 16, 16, 20, 24, 24, 32]))
    ch = int(rng.choice([1, 2, 2, 2, 3, 6, 8]))
    fl = int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(70, 600)), int(rng.integers(600, 5000)), 4096, 352]))
    prof = int(rng.choice([synth.PROFILE_MUSIC, synth.PROFILE_NOISE, synth.PROFILE_QUIET, synth.PROFILE_STRESS,
                           synth.PROFILE_STRESS, synth.PROFILE_MUSIC_NOSHIFT, synth.PROFILE_MUSIC_MIXED]))
    n = int(rng.choice([1, 7, 64, 65, 200, 700]))
    if rng.integers(12) == 0 and fl < 70:  # batches big enough for the gated pair kernel (16-bit) / a second round (others)
        n = int(rng.choice([66000, 70000, 82000, 99000, 132000, 150000]))
        ch = int(rng.choice([1, 2, 2]))
    ppw = rng.choice(["", "64", "16", "2"])
    if ppw:
        os.environ["ALACGPU_PPW"] = str(ppw)
    else:
        os.environ.pop("ALACGPU_PPW", None)
    lm = rng.choice(["", "", "3", "6"])  # from how many taps on the second predictor wave works (alac_duo.h: duo_phase_lanes; read per handle)
    if lm:
        os.environ["ALACGPU_LANES_MIN"] = str(lm)
    else:
        os.environ.pop("ALACGPU_LANES_MIN", None)
    fit = rng.choice(["", "", "", "4", "5"])  # four / five four-wave workgroups per CU whatever the batch size (alac_gpu.h: decode_mode; read per handle)
    if fit:
        os.environ["ALACGPU_FIT"] = str(fit)
    else:
        os.environ.pop("ALACGPU_FIT", None)
    if rng.integers(10) == 0 and fl < 300 and ch <= 2:  # full 64-packet workgroups with one or less per CU: both predictor waves at work
        n = int(rng.choice([16400, 16500, 17000]))
    kb = int(rng.choice([14, 14, 14, 14, 3, 32, 255, 0]))
    # PB / MB: cookie bytes too. PB <= 73 keeps the lean Golomb step (alac_regular.h: lean_config), above it the whole-packet decoder
    pb = int(rng.choice([40, 40, 40, 0, 1, 20, 39, 41, 72, 73, 74, 100, 127, 128, 255, int(rng.integers(0, 256))]))
    mb = int(rng.choice([10, 10, 10, 0, 1, 127, 128, 255, int(rng.integers(0, 256))]))
    if rng.integers(25) == 0 and n <= 700:  # long frames: Apple's encoder goes to 16 384 (docs/research/ENCODERS.md:79); > 65 536: the scan route
        fl = int(rng.choice([8192, 16384, int(rng.integers(4097, 20000)), 65536, 65537, 70000]))
        n = int(rng.choice([1, 7, 64, 65])) if fl < 60000 else int(rng.choice([1, 7, 20]))
    cfg = oracle.make_config(fl, depth, ch, pb=pb, mb=mb, kb=kb)
    bpf = ch * oracle.bytes_per_sample(depth)
    base_seed = int(rng.integers(1 << 30))
    if os.environ.get("ALACGPU_FUZZ_VERBOSE") == "2":
        print("round %d (before the generator): depth %d ch %d fl %d prof %d n %d kb %d pb %d mb %d seed %d" % (r, depth, ch, fl, prof, n, kb, pb, mb, base_seed), flush=True)
    try:
        b = synth.gen_batch(cfg, n, profile=prof, base_seed=base_seed, threads=8)
    except RuntimeError:
        continue  # the encoder has no code for this residual under this KB
    blob, offs, sizes = b.blob, b.offsets, b.sizes
    what = rng.integers(6)
    if what < 2:
        blob, offs, sizes = pack_packets(mutate_packets(b, rng, n))
    elif what == 2 and fl <= 5000 and n <= 700:  # the mean at the top of its range (conftest.loud_packets) among the others
        try:
            lp = [q for q, _ in loud_packets(synth, cfg, max(1, n // 2), seed=int(rng.integers(1 << 30)), pb_factor=int(rng.choice([7, 7, 4, 5])),
                                             order=int(rng.choice([0, 0, 4, 31])))]
        except RuntimeError:
            lp = []
        blob, offs, sizes = pack_packets(lp + [b.packet(i) for i in range(n - len(lp))])
    ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
    if os.environ.get("ALACGPU_FUZZ_VERBOSE"):
        print("round %d: depth %d ch %d fl %d prof %d n %d ppw %r kb %d pb %d mb %d lanes_min %r fit %r what %d" % (r, depth, ch, fl, prof, len(offs), ppw, kb, pb, mb, lm, fit, what), flush=True)
    if DRY:
        rng.integers(0, 4)
        if os.environ.get("ALACGPU_FUZZ_DUMP") == str(r):  # this round's batch and the oracle's answer, for tests/host_sim or a debugger
            np.savez(os.environ.get("ALACGPU_FUZZ_DUMP_TO", "/tmp/fuzz_round.npz"), blob=blob, offs=offs, sizes=sizes, out=ref[0], frames=ref[1], status=ref[2],
                     cfg=np.array([fl, depth, ch, pb, mb, kb]))
            print("dumped round %d" % r)
            sys.exit(0)
        continue
    c = pkg.PacketConfig(FrameLength=fl, BitDepth=depth, NumChannels=ch, PB=cfg.pb, MB=cfg.mb, KB=cfg.kb,
                         MaxRun=cfg.max_run, SampleRate=cfg.sample_rate)
    with pkg.NewPacketDecoder(c, 0) as dec:
        o = np.zeros(len(offs) + 1, np.uint64)
        # host entry wants back-to-back packets: re-pack
        pk = [bytes(blob[int(offs[i]):int(offs[i]) + int(sizes[i])]) for i in range(len(offs))]
        o[1:] = np.cumsum([len(p) for p in pk])
        # dense: nothing behind the packets, a random lead shifts every alignment
        lead = int(rng.integers(0, 4))
        out, fr, st = dec.decode_batch(np.frombuffer(b"\xff" * lead + b"".join(pk) or b"\0", np.uint8), o + np.uint64(lead))
    try:
        assert_same_decode(cfg, ref, (out, fr, st), bpf, "round %d" % r)
    except AssertionError as e:
        bad += 1
        print("MISMATCH depth %d ch %d fl %d prof %d n %d ppw %r kb %d pb %d mb %d lanes_min %r fit %r what %d: %s" % (depth, ch, fl, prof, n, ppw, kb, pb, mb, lm, fit, what, e), flush=True)
    if r % 50 == 49:
        print("round %d" % (r + 1), flush=True)
print("%d rounds, %d mismatches" % (rounds, bad))
sys.exit(1 if bad else 0)
