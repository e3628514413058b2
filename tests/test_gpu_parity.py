"""Parity of the HIP path with the oracle on a real MI355X — every call goes through the C ABI
(include/alacgpu.h via the ctypes mirror). Bit-exact: PCM bytes, frame counts and status words."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gpu_decode(dec, b_blob, offsets, sizes):
    """Host entry (alacgpu_decode_batch) on the packets laid out DENSELY (back to back, as in an mdat): every test
    that goes through here also checks that a packet's neighbour is never read as its zero pad."""
    pk = [b_blob[int(o):int(o) + int(s)].tobytes() for o, s in zip(offsets, sizes)]
    offs = np.zeros(len(pk) + 1, np.uint64)
    offs[1:] = np.cumsum([len(p) for p in pk], dtype=np.uint64)
    blob = np.frombuffer(b"".join(pk) or b"\0", np.uint8)
    return dec.decode_batch(blob, offs)


def test_known_answer_packets_on_gpu(pkg):
    k = json.load(open(os.path.join(HERE, "golden", "kat.json")))
    c = k["config_common"]
    for v in k["vectors"]:
        cfg = pkg.PacketConfig(FrameLength=v["frame_length"], BitDepth=c["bit_depth"],
                               NumChannels=v["num_channels"], PB=c["pb"], MB=c["mb"], KB=c["kb"], MaxRun=c["max_run"])
        with pkg.NewPacketDecoder(cfg) as dec:
            pcm = dec.DecodePacket(bytes.fromhex(v["packet"].replace(" ", "")))
            assert pcm.hex().upper() == v["pcm"].upper(), v["name"]
            f = dec.Format()
            assert (f.SampleRate, f.BitDepth, f.Channels) == (44100, 16, v["num_channels"])


def test_hand_derived_predictor_and_matrix_packets_on_gpu(pkg):
    """K5..K13, K14..K19 and K20..K23 (other cookie bytes: PB 20 / 73 / 255, MB 0 / 255 / 1, KB 32; tests/golden/kat_derivation.md)
    through DecodePacket on the GPU."""
    k = json.load(open(os.path.join(HERE, "golden", "kat2.json")))
    c = k["config_common"]
    more = [json.load(open(os.path.join(HERE, "golden", n)))["vectors"] for n in ("kat3.json", "kat4.json")]
    for v in k["vectors"] + more[0] + more[1]:
        cfg = pkg.PacketConfig(FrameLength=v["frame_length"], BitDepth=v["bit_depth"], NumChannels=v["num_channels"],
                               PB=v.get("pb", c["pb"]), MB=v["mb"], KB=v.get("kb", c["kb"]), MaxRun=c["max_run"])
        with pkg.NewPacketDecoder(cfg) as dec:
            pcm = dec.DecodePacket(bytes.fromhex(v["packet"]))
            assert pcm.hex().upper() == v["pcm"].upper(), v["name"]
            # the same packet inside a batch of copies (regular path: full waves, same key)
            res = dec.DecodePackets([bytes.fromhex(v["packet"])] * 70)
            assert all(r == pcm for r in res), v["name"]


def test_golden_packets_on_gpu(pkg):
    g = json.load(open(os.path.join(HERE, "golden", "golden_packets.json")))
    c = g["config_common"]
    decs = {}
    for v in g["vectors"]:
        key = (v["frame_length"], v["bit_depth"], v["num_channels"])
        if key not in decs:
            decs[key] = pkg.NewPacketDecoder(pkg.PacketConfig(FrameLength=key[0], BitDepth=key[1], NumChannels=key[2],
                                                              PB=c["pb"], MB=c["mb"], KB=c["kb"], MaxRun=c["max_run"]))
        dec = decs[key]
        try:
            pcm = dec.DecodePacket(bytes.fromhex(v["packet"]))
            st = 0
        except pkg.ErrDecode as e:
            pcm, st = b"", e.status
        assert st == v["status"], v
        assert hashlib.sha256(pcm).hexdigest() == v["pcm_sha256"]
    for d in decs.values():
        d.close()


CONFIGS = [(16, 2, 4096), (24, 2, 4096), (16, 1, 4096), (24, 8, 1024), (20, 2, 512), (32, 2, 512), (16, 6, 300),
           (20, 3, 300), (32, 1, 100), (16, 7, 40), (24, 4, 77), (16, 5, 33), (16, 2, 1)]


@pytest.mark.parametrize("depth,ch,fl", CONFIGS)
def test_batch_matches_oracle(pkg, oracle, synth, helpers, gpu_decoder_factory, depth, ch, fl):
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    with gpu_decoder_factory(cfg) as dec:
        for prof, n in ((synth.PROFILE_MUSIC, 200), (synth.PROFILE_NOISE, 70), (synth.PROFILE_QUIET, 70),
                        (synth.PROFILE_STRESS, 330)):
            b = synth.gen_batch(cfg, n, profile=prof, threads=8)
            ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
            got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
            helpers.assert_same_decode(cfg, ref, got, bpf, "profile %d" % prof)
            if fl > 8 and prof != synth.PROFILE_STRESS:  # fl <= order: the reference panics in the warm-up
                # lossless: the decoder gives back the source PCM (tests/conformance_test.go:282-291)
                assert (got[2] == 0).all()
                for i in range(b.n):
                    nb = int(b.frames[i]) * bpf
                    assert np.array_equal(got[0][i, :nb], b.pcm[i, :nb])


@pytest.mark.parametrize("depth,ch,fl,n", [(24, 2, 48, 70000), (32, 2, 40, 99000), (20, 1, 64, 132000), (16, 2, 36, 132000),
                                              (16, 1, 50, 200000)])
def test_more_than_one_round_of_four_wave_workgroups(pkg, oracle, synth, helpers, gpu_decoder_factory, depth, ch, fl, n):
    """Batches of more than 4 x CUs wave slots that are not the gated 16-bit twin's: the four-wave kernels run them in as
    many rounds as it takes (k_dec16q / k_dec24q / k_dec32q; until the end of round 3 wave pairs did), the spare wave of
    every workgroup rotating over its CU's SIMDs round after round. PCM, frame counts and status words are the oracle's, and
    the decoder gives back the source PCM (tests/conformance_test.go:282-291)."""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    with gpu_decoder_factory(cfg) as dec:
        b = synth.gen_batch(cfg, n, profile=synth.PROFILE_MUSIC, threads=16)
        ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=16)
        got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
        helpers.assert_same_decode(cfg, ref, got, bpf, "%d packets" % n)
        assert (got[2] == 0).all()
        for i in range(0, b.n, 997):
            nb = int(b.frames[i]) * bpf
            assert np.array_equal(got[0][i, :nb], b.pcm[i, :nb])


@pytest.mark.parametrize("depth,ch,fl,n,ppw", [(24, 2, 52, 17000, "16"), (16, 2, 41, 99000, ""), (20, 2, 42, 70000, "16"),
                                                  (24, 2, 227, 16400, "2")])
def test_large_batches_with_few_narrow_regular_slots(pkg, oracle, synth, helpers, gpu_decoder_factory, monkeypatch, depth, ch,
                                                     fl, n, ppw):
    """Large STRESS batches: most wave slots belong to irregular packets and wide keys, so the count of narrow regular
    slots — what decides on the device between the four-wave kernels and the wave pairs (k_decode_body.inc: three_waves)
    — is small although the batch is not. Round 3's first version guessed on the HOST, from the upper bound of all slots,
    whether the four-wave kernel had to be launched: it was not, the wave pairs left the slots to it, and the packets
    stayed undecoded (found by tools/gpu_fuzz.py; these are its cases)."""
    if ppw:
        monkeypatch.setenv("ALACGPU_PPW", ppw)
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    with gpu_decoder_factory(cfg) as dec:
        b = synth.gen_batch(cfg, n, profile=synth.PROFILE_STRESS, threads=8)
        ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
        got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
        helpers.assert_same_decode(cfg, ref, got, bpf, "STRESS, %d packets" % n)


@pytest.mark.parametrize("depth,ch,fl,n,profiles", [(16, 2, 4096, 300, "ms"), (16, 1, 700, 300, "ms"), (24, 2, 1024, 300, "ms"),
                                                      (20, 2, 512, 300, "ms"), (16, 2, 64, 16500, "m"), (24, 2, 48, 16500, "m")])
def test_two_lane_predictor_waves_match_oracle(pkg, oracle, synth, helpers, gpu_decoder_factory, monkeypatch, depth, ch, fl, n,
                                               profiles):
    """The second predictor wave of the four-wave workgroups (k_dec16q / k_dec24q, alac_duo.h: duo_phase_lanes: a packet's
    taps on two lanes) forced on for EVERY key with orders 3..16 (ALACGPU_LANES_MIN is read when a handle is made; by
    default only keys with nine taps or more take it): odd and even orders, int16-wrapping and int32 coefficient orders,
    every denShift, partial frames, escape codes and zero runs beside it (STRESS), and batches of full 64-packet workgroups
    (16 500 packets: both predictor waves at work, each taking its number of steps from the other's packets too)."""
    monkeypatch.setenv("ALACGPU_LANES_MIN", "3")
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    with gpu_decoder_factory(cfg) as dec:
        for prof in [synth.PROFILE_MUSIC] + ([synth.PROFILE_STRESS, synth.PROFILE_QUIET] if "s" in profiles else []):
            b = synth.gen_batch(cfg, n, profile=prof, threads=8)
            ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
            got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
            helpers.assert_same_decode(cfg, ref, got, bpf, "lanes, profile %d" % prof)
            if prof != synth.PROFILE_STRESS:
                assert (got[2] == 0).all()
                for i in range(0, b.n, max(1, b.n // 500)):
                    nb = int(b.frames[i]) * bpf
                    assert np.array_equal(got[0][i, :nb], b.pcm[i, :nb])
            if prof == synth.PROFILE_MUSIC and n <= 1000:
                # damaged packets beside intact ones: the entropy wave parks a failed lane, the predictor waves keep going
                rng = np.random.default_rng(7 + depth + ch)
                blob, offs, sizes = helpers.pack_packets(helpers.mutate_packets(b, rng, 400) + [b.packet(i) for i in range(60)])
                ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
                got = _gpu_decode(dec, blob, offs, sizes)
                helpers.assert_same_decode(cfg, ref, got, bpf, "lanes, damaged packets")


@pytest.mark.parametrize("depth,ch,fl,kb", [(16, 2, 256, 14), (24, 2, 128, 14), (16, 1, 64, 14), (24, 8, 32, 14),
                                            (20, 3, 50, 14), (32, 2, 64, 14), (16, 2, 256, 0), (16, 2, 8, 255),
                                            (16, 2, 256, 32), (16, 2, 256, 255), (24, 5, 64, 40)])
def test_corrupt_packets_match_oracle_and_do_not_poison_the_batch(pkg, oracle, synth, helpers, gpu_decoder_factory,
                                                                  depth, ch, fl, kb):
    cfg = oracle.make_config(fl, depth, ch, kb=kb)
    bpf = ch * oracle.bytes_per_sample(depth)
    rng = np.random.default_rng(99 + depth + ch + kb)
    with gpu_decoder_factory(cfg) as dec:
        # KB >= 32 (a cookie byte, untrusted): zero runs (QUIET) are where WB = (1 << KB) - 1 matters (golomb.go:60,227)
        b = synth.gen_batch(cfg, 64, profile=synth.PROFILE_QUIET if kb >= 32 else synth.PROFILE_MUSIC, threads=8)
        good = [b.packet(i) for i in range(b.n)]
        bad = helpers.mutate_packets(b, rng, 700)
        mixed = []
        for i, p in enumerate(bad):  # interleave intact packets: a bad neighbour must not affect them
            mixed.append(p)
            if i % 5 == 0:
                mixed.append(good[i % len(good)])
        blob, offs, sizes = helpers.pack_packets(mixed)
        ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
        got = _gpu_decode(dec, blob, offs, sizes)
        helpers.assert_same_decode(cfg, ref, got, bpf, "fuzz")
        assert (ref[2] != 0).sum() > 100 and (ref[2] == 0).sum() > 100


@pytest.mark.parametrize("pb", [0, 1, 20, 39, 41, 73, 74, 127, 255])
def test_other_cookie_bytes_pb_and_mb_match_oracle(pkg, oracle, synth, helpers, gpu_decoder_factory, pb):
    """PB and MB are bytes of an untrusted file's magic cookie (config.go:72-73); effective pb = PB * pbFactor / 4
    (decoder.go:296-299), first mean MB (golomb.go:157). PB <= 73 takes the lean Golomb step (alac_regular.h: lean_config:
    pb <= 127, mean < 2^25 + 512, pb * mean does not wrap), larger ones the whole-packet decoder with the reference's uint32
    arithmetic (golomb.go:215): both sides of that edge and the extremes, MB 0 / 1 / 10 / 255, on MUSIC / QUIET / NOISE /
    STRESS (pbFactor 0..7) streams of 1, 2 and 6 channels, damaged packets beside intact ones, and `loud` packets that hold
    the mean at the top of its range with pbFactor 7 (conftest.loud_packets)."""
    rng = np.random.default_rng(1000 + pb)
    for mb in (0, 1, 10, 255):
        for depth, ch, fl in ((16, 2, 300), (16, 1, 200), (24, 2, 128), (16, 6, 64), (20, 2, 96), (32, 1, 80)):
            cfg = oracle.make_config(fl, depth, ch, pb=pb, mb=mb)
            bpf = ch * oracle.bytes_per_sample(depth)
            with gpu_decoder_factory(cfg) as dec:
                for prof in (synth.PROFILE_MUSIC, synth.PROFILE_QUIET, synth.PROFILE_STRESS, synth.PROFILE_NOISE):
                    b = synth.gen_batch(cfg, 150, profile=prof, base_seed=pb * 1000 + mb, threads=8)
                    ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
                    got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
                    helpers.assert_same_decode(cfg, ref, got, bpf, "pb %d mb %d %d-bit %d-ch profile %d" % (pb, mb, depth, ch, prof))
                    if prof != synth.PROFILE_STRESS:  # lossless under any cookie (tests/conformance_test.go:282-291)
                        assert (got[2] == 0).all()
                        for i in range(b.n):
                            nb = int(b.frames[i]) * bpf
                            assert np.array_equal(got[0][i, :nb], b.pcm[i, :nb])
                    mixed = helpers.mutate_packets(b, rng, 150) + [b.packet(i) for i in range(0, b.n, 5)]
                    blob, offs, sizes = helpers.pack_packets(mixed)
                    ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
                    helpers.assert_same_decode(cfg, ref, _gpu_decode(dec, blob, offs, sizes), bpf,
                                               "damaged, pb %d mb %d %d-bit %d-ch profile %d" % (pb, mb, depth, ch, prof))
                if mb in (10, 255):
                    loud = helpers.loud_packets(synth, cfg, 80, seed=pb + depth + mb)
                    blob, offs, sizes = helpers.pack_packets([q for q, _ in loud])
                    ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
                    assert (ref[2] == 0).all()
                    got = _gpu_decode(dec, blob, offs, sizes)
                    helpers.assert_same_decode(cfg, ref, got, bpf, "loud, pb %d mb %d %d-bit %d-ch" % (pb, mb, depth, ch))
                    for i, (_, pcm) in enumerate(loud):
                        assert got[0][i, :len(pcm)].tobytes() == pcm


@pytest.mark.parametrize("depth,ch,fl,n", [(16, 2, 8192, 200), (16, 2, 16384, 140), (24, 2, 8192, 140), (24, 2, 16384, 100),
                                              (24, 8, 8192, 40), (16, 6, 16384, 24), (32, 2, 16384, 70), (20, 1, 16384, 130),
                                              (16, 2, 65536, 66), (16, 2, 65537, 40), (24, 2, 70000, 24), (16, 1, 100000, 30)])
def test_long_frames_match_oracle(pkg, oracle, synth, helpers, gpu_decoder_factory, depth, ch, fl, n):
    """FrameLength is any uint32 of the cookie (config.go:70); Apple's encoder goes to 16 384 (docs/research/ENCODERS.md:79).
    Up to 65 536 frames a mono / stereo packet stays regular (alac_regular.h: classify_regular): the U tile, the rings'
    `near` logic and bit positions of 16 384 x 2 x 33 bits at sizes four and sixteen times the benchmark's; above 65 536 the
    scan route; 8 and 6 channels through scan -> predictor pass -> interleave with rows of 16 384 samples. MUSIC and QUIET
    give back the source PCM (tests/conformance_test.go:282-291); STRESS (partial frames, every order, escape elements) and
    damaged packets are the oracle's byte for byte."""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    rng = np.random.default_rng(fl + depth)
    with gpu_decoder_factory(cfg) as dec:
        for prof, k in ((synth.PROFILE_MUSIC, n), (synth.PROFILE_STRESS, max(8, n // 2)), (synth.PROFILE_QUIET, max(8, n // 3)),
                        (synth.PROFILE_NOISE, max(4, n // 8))):
            b = synth.gen_batch(cfg, k, profile=prof, base_seed=fl + prof, threads=16)
            ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=16)
            got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
            helpers.assert_same_decode(cfg, ref, got, bpf, "%d frames, profile %d" % (fl, prof))
            if prof != synth.PROFILE_STRESS and not (depth == 32 and ch == 2 and prof == synth.PROFILE_NOISE):
                assert (got[2] == 0).all()
                for i in range(b.n):
                    nb = int(b.frames[i]) * bpf
                    assert np.array_equal(got[0][i, :nb], b.pcm[i, :nb])
            if prof == synth.PROFILE_MUSIC:
                mixed = helpers.mutate_packets(b, rng, 40) + [b.packet(i) for i in range(0, b.n, 7)]
                blob, offs, sizes = helpers.pack_packets(mixed)
                ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=16)
                helpers.assert_same_decode(cfg, ref, _gpu_decode(dec, blob, offs, sizes), bpf, "%d frames, damaged" % fl)


@pytest.mark.parametrize("depth,ch", [(32, 1), (24, 2), (24, 1), (20, 1)])
def test_runs_of_escape_codes_up_to_a_truncation_point(pkg, oracle, synth, helpers, gpu_decoder_factory, depth, ch):
    """The GPU twin of tests/test_lane_logic.py's test of the same name: escape code after escape code (24 / 32-bit
    streams without shift bytes: up to 41 bits a step, inside the lean step) in packets cut at every byte of their last
    stretch, full 64-lane waves of them; status words and the frames in front are the oracle's (golomb.go:168,86-108)."""
    fl = 96
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    b = synth.gen_batch(cfg, 12, profile=synth.PROFILE_MUSIC_NOSHIFT, threads=4)
    packets = []
    for i in range(b.n):
        p = b.packet(i)
        packets += [p[:k] for k in range(max(1, len(p) - 90), len(p) + 1)]
    blob, offs, sizes = helpers.pack_packets(packets)
    ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
    assert len(np.unique(ref[2])) >= 2
    with gpu_decoder_factory(cfg) as dec:
        helpers.assert_same_decode(cfg, ref, _gpu_decode(dec, blob, offs, sizes), bpf, "truncated escape runs")


def test_ragged_and_empty_batches(pkg, oracle, synth, helpers, gpu_decoder_factory):
    cfg = oracle.make_config(512, 16, 2)
    with gpu_decoder_factory(cfg) as dec:
        assert dec.DecodePackets([]) == []
        for n in (1, 63, 64, 65, 129):  # around the 64-packet wave
            b = synth.gen_batch(cfg, n, threads=4)
            ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes)
            helpers.assert_same_decode(cfg, ref, _gpu_decode(dec, b.blob, b.offsets, b.sizes), 4, "n=%d" % n)
        res = dec.DecodePackets([b.packet(0), b"", bytes([0xE0]), b.packet(1)[:20]])
        assert res[0] == b.pcm[0, :int(b.frames[0]) * 4].tobytes()
        assert isinstance(res[1], pkg.ErrDecode) and res[1].sentinel == pkg.ErrBitstreamOverrun
        assert res[2] == bytes(512 * 4)
        assert isinstance(res[3], pkg.ErrDecode)
        with pytest.raises(pkg.ErrDecode):
            dec.DecodePacket(b"")


def test_device_resident_entry_full_size_round_trip(pkg, synth, oracle, gpu_decoder_factory):
    """BASELINE config b: 4096 x 16-bit stereo 4096-frame packets through alacgpu_decode_batch_device,
    checked by the size-independent property decode(encode(pcm)) == pcm, plus a slice against the oracle."""
    import torch
    cfg = oracle.make_config(4096, 16, 2)
    n = 4096
    b = synth.gen_batch(cfg, n, threads=16)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(b.blob).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
    stride = 4096 * 4
    d_out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    d_fr = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with gpu_decoder_factory(cfg) as dec:
        dec.decode_batch_device(d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr(), d_sz.data_ptr(), n, d_out.data_ptr(), stride,
                                d_fr.data_ptr(), d_st.data_ptr(), sync=True)
        assert dec.last_kernel_ms() > 0
    assert int(d_st.abs().sum()) == 0
    assert np.array_equal(d_fr.cpu().numpy().astype(np.uint32), b.frames)
    assert torch.equal(d_out.cpu(), torch.from_numpy(b.pcm))
    ref = oracle.decode_batch(cfg, b.blob, b.offsets[:256], b.sizes[:256], threads=8)
    assert np.array_equal(ref[0], d_out[:256].cpu().numpy())


@pytest.mark.parametrize("depth,ch,fl", [(16, 2, 256), (24, 2, 128), (24, 8, 48), (32, 1, 64)])
def test_dense_device_entry_hostile_neighbours_and_bad_descriptors(pkg, oracle, synth, helpers, gpu_decoder_factory,
                                                                   depth, ch, fl):
    """alacgpu_decode_batch_device on a dense blob: truncated packets whose neighbours are non-zero bytes decode as
    if followed by the reference's zero pad (bitbuffer.go:33); descriptors that leave the blob get ALACGPU_ERR_RANGE
    and do not disturb their neighbours; d_sizes == NULL takes the sizes from offsets[n+1]."""
    import torch
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    rng = np.random.default_rng(depth + ch)
    b = synth.gen_batch(cfg, 80, profile=synth.PROFILE_MUSIC, threads=8)
    q = synth.gen_batch(cfg, 40, profile=synth.PROFILE_QUIET, threads=8)
    packets = []
    for src in (b, q):
        for i in range(src.n):
            p = src.packet(i)
            packets += [p, p[:int(rng.integers(1, len(p)))], p[:max(1, len(p) - int(rng.integers(1, 9)))]]
    packets += helpers.mutate_packets(b, rng, 100)
    packets.append(b.packet(3)[:len(b.packet(3)) // 2])  # a truncated packet ends the blob
    ref_blob, ref_offs, ref_sizes = helpers.pack_packets(packets)
    ref = oracle.decode_batch(cfg, ref_blob, ref_offs, ref_sizes, threads=8)
    dev = torch.device("cuda:0")
    stride = (fl * bpf + 15) // 16 * 16
    for lead in (0, 3):
        blob, offs, sizes = helpers.pack_dense(packets, lead=lead)
        n = len(packets)
        offs1 = np.concatenate([offs, [np.uint64(len(blob))]]).astype(np.uint64)
        d_blob = torch.from_numpy(blob).to(dev)
        d_off = torch.from_numpy(offs1.astype(np.int64)).to(dev)
        d_sz = torch.from_numpy(sizes.astype(np.int32)).to(dev)
        with gpu_decoder_factory(cfg) as dec:
            for use_sizes in (True, False):
                d_out = torch.full((n, stride), 0xA5, dtype=torch.uint8, device=dev)
                d_fr = torch.full((n,), -1, dtype=torch.int32, device=dev)
                d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                dec.decode_batch_device(d_blob.data_ptr(), len(blob), d_off.data_ptr(),
                                        d_sz.data_ptr() if use_sizes else None, n, d_out.data_ptr(), stride,
                                        d_fr.data_ptr(), d_st.data_ptr(), sync=True)
                got = (d_out.cpu().numpy(), d_fr.cpu().numpy().astype(np.uint32), d_st.cpu().numpy())
                helpers.assert_same_decode(cfg, ref, got, bpf, "dense lead %d sizes %s" % (lead, use_sizes))
            # descriptors that leave the blob: flagged, neighbours unaffected
            bad_off = offs.copy()
            bad_sz = sizes.copy()
            bad_off[5] = len(blob) + 17               # starts behind the blob
            bad_sz[9] = len(blob)                     # reaches beyond it
            bad_off[11] = np.uint64(2 ** 63)          # nowhere near
            bad_sz[13] = 0x7fffffff
            d_off2 = torch.from_numpy(bad_off.astype(np.int64)).to(dev)
            d_sz2 = torch.from_numpy(bad_sz.astype(np.int32)).to(dev)
            d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            dec.decode_batch_device(d_blob.data_ptr(), len(blob), d_off2.data_ptr(), d_sz2.data_ptr(), n, d_out.data_ptr(),
                                    stride, d_fr.data_ptr(), d_st.data_ptr(), sync=True)
            st = d_st.cpu().numpy()
            fr = d_fr.cpu().numpy()
            for k in (5, 9, 11, 13):
                assert st[k] == 7 and fr[k] == 0, (k, st[k])  # ALACGPU_ERR_RANGE
            keep = np.ones(n, bool)
            keep[[5, 9, 11, 13]] = False
            assert np.array_equal(st[keep], ref[2][keep]) and np.array_equal(fr[keep].astype(np.uint32), ref[1][keep])


def test_host_entry_many_chunks(pkg, oracle, synth, helpers, gpu_decoder_factory, monkeypatch):
    """alacgpu_decode_batch cuts a batch into chunks that are in flight on three streams at once: force small chunks
    (slots are reused many times) and check every packet, pageable and pinned caller memory."""
    import torch
    monkeypatch.setenv("ALACGPU_CHUNK_MB", "1")
    cfg = oracle.make_config(1024, 16, 2)
    b = synth.gen_batch(cfg, 3000, threads=8)
    ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
    with gpu_decoder_factory(cfg) as dec:
        for _ in range(2):
            helpers.assert_same_decode(cfg, ref, _gpu_decode(dec, b.blob, b.offsets, b.sizes), 4, "pageable")
        blob, offs, sizes = helpers.pack_dense([b.packet(i) for i in range(b.n)])
        offs1 = np.concatenate([offs, [np.uint64(len(blob))]]).astype(np.uint64)
        t_blob = torch.empty(len(blob), dtype=torch.uint8, pin_memory=True)
        t_blob.numpy()[:] = blob
        t_out = torch.empty((b.n, dec.frame_bytes), dtype=torch.uint8, pin_memory=True)
        t_fr = torch.empty(b.n, dtype=torch.int32, pin_memory=True)
        t_st = torch.empty(b.n, dtype=torch.int32, pin_memory=True)
        pkg._check(dec._lib.alacgpu_decode_batch(dec._h, t_blob.data_ptr(), len(blob), offs1.ctypes.data, b.n, t_out.data_ptr(),
                                                 dec.frame_bytes, t_fr.data_ptr(), t_st.data_ptr()))
        got = (t_out.numpy(), t_fr.numpy().view(np.uint32), t_st.numpy())
        helpers.assert_same_decode(cfg, ref, got, 4, "pinned")


def test_24bit_shift_and_8ch_full_frames(pkg, synth, oracle, helpers, gpu_decoder_factory):
    """BASELINE configs c and d at reduced batch: 24-bit stereo with shift buffer; 7.1 24-bit."""
    for ch, n in ((2, 512), (8, 256)):
        cfg = oracle.make_config(4096, 24, ch)
        b = synth.gen_batch(cfg, n, threads=16)
        with gpu_decoder_factory(cfg) as dec:
            got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
        assert (got[2] == 0).all() and np.array_equal(got[1], b.frames)
        assert np.array_equal(got[0], b.pcm)


def test_bytes_behind_a_partial_frame_read_as_zero_on_a_reused_handle(pkg, synth, oracle, helpers, gpu_decoder_factory):
    """Host entry: the kernels leave the bytes behind a partial frame alone, and a handle that comes back from the pool
    (alacgpu_destroy keeps device buffers) still holds an earlier batch's PCM in its staging: the caller must get zeros
    there, as from DecodePacket's zeroed frame buffer (decoder.go:120,127), not somebody else's samples. (Round 3: found
    when a large batch ran before test_24bit_shift_and_8ch_full_frames in the same process.)"""
    cfg = oracle.make_config(512, 24, 2)
    bpf = 6
    loud = synth.gen_batch(cfg, 3000, profile=synth.PROFILE_NOISE, threads=8)
    with gpu_decoder_factory(cfg) as dec:
        _gpu_decode(dec, loud.blob, loud.offsets, loud.sizes)  # fills the staging with non-zero PCM
    b = synth.gen_batch(cfg, 3000, profile=synth.PROFILE_STRESS, threads=8)  # a fifth of its packets have partial frames
    assert (b.frames < 512).sum() > 100
    with gpu_decoder_factory(cfg) as dec:  # the pooled handle
        out, fr, st = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
    ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
    helpers.assert_same_decode(cfg, ref, (out, fr, st), bpf, "reused handle")
    for i in range(b.n):
        assert not out[i, int(fr[i]) * bpf:].any(), "packet %d: bytes behind its %d frames" % (i, fr[i])


@pytest.mark.parametrize("depth,ch,fl", [(16, 2, 33), (16, 2, 47), (16, 2, 1000), (16, 1, 4095), (24, 2, 129),
                                         (20, 1, 65), (32, 2, 200), (16, 2, 4097)])
def test_wave_pair_chunk_tails_and_full_waves(pkg, oracle, synth, helpers, gpu_decoder_factory, monkeypatch, depth, ch, fl):
    """The wave pair of alac_duo.h works in chunks of 16 steps and groups of 4 / 8: frame lengths around those
    multiples, partial frames inside a wave, and full 64-lane waves (forced: a small batch is normally spread over
    narrow waves) with several sort keys per batch."""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    for ppw in ("64", "16", None):
        if ppw is None:
            monkeypatch.delenv("ALACGPU_PPW", raising=False)
        else:
            monkeypatch.setenv("ALACGPU_PPW", ppw)
        with gpu_decoder_factory(cfg) as dec:
            for prof, n in ((synth.PROFILE_MUSIC, 333), (synth.PROFILE_STRESS, 260)):
                b = synth.gen_batch(cfg, n, profile=prof, base_seed=fl * 131 + depth, threads=8)
                ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
                got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
                helpers.assert_same_decode(cfg, ref, got, bpf, "ppw %s profile %d" % (ppw, prof))


@pytest.mark.parametrize("ch,n", [(2, 65536), (8, 16384)])
def test_baseline_configs_c_and_d_at_full_size(pkg, synth, oracle, gpu_decoder_factory, ch, n):
    """BASELINE config c (65 536 x 24-bit stereo with shift bytes) and d (16 384 x 24-bit 7.1) at their full batch
    sizes through the device-resident entry: several rounds of workgroups per CU, the sort with every key present.
    Checked by the size-independent property decode(encode(pcm)) == pcm (tests/conformance_test.go:282-291), frame
    counts and status words; the oracle decodes a slice."""
    import torch
    cfg = oracle.make_config(4096, 24, ch)
    b = synth.gen_batch(cfg, n, threads=16)
    dev = torch.device("cuda:0")
    stride = 4096 * ch * 3
    d_blob = torch.from_numpy(b.blob).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
    d_out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    d_fr = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with gpu_decoder_factory(cfg) as dec:
        dec.decode_batch_device(d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr(), d_sz.data_ptr(), n, d_out.data_ptr(),
                                stride, d_fr.data_ptr(), d_st.data_ptr(), sync=True)
    assert int(d_st.abs().sum()) == 0
    assert np.array_equal(d_fr.cpu().numpy().astype(np.uint32), b.frames)
    for lo in range(0, n, 4096):  # nothing is written behind a partial frame either: whole slots compare equal
        exp = torch.from_numpy(b.pcm[lo:lo + 4096]).to(dev)
        assert torch.equal(d_out[lo:lo + 4096], exp), lo
        del exp
    ref = oracle.decode_batch(cfg, b.blob, b.offsets[:128], b.sizes[:128], threads=8)
    assert np.array_equal(ref[0], d_out[:128].cpu().numpy())


@pytest.mark.parametrize("n,fl,profile", [(70000, 64, 0), (98304, 40, 0), (150000, 33, 3), (66000, 48, 2)])
def test_batches_between_the_rounds_take_the_gated_pairs(pkg, synth, oracle, helpers, gpu_decoder_factory, n, fl, profile):
    """16-bit batches of more than 4 x CUs wave slots but less than the next multiple (k_decode_body.inc): the gated
    kernel decodes them with five or six pairs per CU. Every regular slot is decoded exactly once by some pair, no CU
    admits more pairs than its quota — and the PCM is the oracle's (DynDecomp / UnpcBlock / WriteStereo16, golomb.go:148, predictor.go:45, matrix.go:30)."""
    import torch
    cfg = oracle.make_config(fl, 16, 2)
    b = synth.gen_batch(cfg, n, profile=profile, threads=16)
    dev = torch.device("cuda:0")
    stride = fl * 4
    d_blob = torch.from_numpy(b.blob).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
    d_out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    d_fr = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with gpu_decoder_factory(cfg) as dec:
        dec.decode_batch_device(d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr(), d_sz.data_ptr(), n, d_out.data_ptr(),
                                stride, d_fr.data_ptr(), d_st.data_ptr(), sync=True)
        place = dec.pair_placement()
    ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=16)
    got = (d_out.cpu().numpy(), d_fr.cpu().numpy().astype(np.uint32), d_st.cpu().numpy())
    helpers.assert_same_decode(cfg, ref, got, 4, "n=%d" % n)
    tags = place[:, 0]
    owned = tags[tags != 0]
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    slots = len(tags)
    if slots > 4 * n_cu and slots <= 6 * n_cu:  # otherwise pair_quota leaves the batch to the ungated kernel
        assert len(owned) >= slots - 24 and (owned >> 31).all()  # all but the irregular slots, each tagged once
        cu = (owned >> 8) & 0x1ff
        first = owned[(owned & 1) == 0]  # a pair's first item; bit 0 marks the ones it took afterwards
        per_cu = np.bincount(((first >> 8) & 0x1ff).astype(np.int64), minlength=512)
        # (with frames this short the first pairs are done before the last ones arrive, so only the ceiling is firm)
        assert len(np.unique(cu)) > n_cu // 2 and per_cu.max() <= 6
        assert (place[tags != 0, 2] != place[tags != 0, 1]).all()  # start and end clocks recorded


@pytest.mark.parametrize("depth,ch,n,fl,profile,force", [(24, 2, 70000, 48, 0, None), (32, 2, 70000, 40, 0, None), (20, 1, 75000, 37, 3, None),
                                                         (24, 2, 140000, 35, 2, None), (16, 2, 70000, 64, 0, "5"), (16, 1, 20000, 100, 3, "5"),
                                                         (24, 2, 9000, 130, 3, "5"), (16, 2, 70000, 64, 0, "4")])
def test_five_four_wave_workgroups_per_cu(pkg, synth, oracle, helpers, gpu_decoder_factory, monkeypatch, depth, ch, n, fl, profile, force):
    """Round 4: the four-wave kernels launched without their dynamic-LDS pad ("fit 5": five workgroups share a CU, alac_gpu.h:
    decode_mode) — batches whose narrow wave slots number 4..5 x CUs or more than 7 x CUs, the widths without a gated twin, the
    48-dword stager rows of the 3-byte writer, partial frames, and the shape forced on (ALACGPU_FIT) for batches of other sizes
    and for the 16-bit kernel, whose batches of that size the gated twin takes otherwise. PCM, frame counts and status words are
    the oracle's (golomb.go:148, predictor.go:45, matrix.go:30); the dispatch read back from the device names the shape."""
    import torch
    if force is None:
        monkeypatch.delenv("ALACGPU_FIT", raising=False)
    else:
        monkeypatch.setenv("ALACGPU_FIT", force)
    cfg = oracle.make_config(fl, depth, ch)
    b = synth.gen_batch(cfg, n, profile=profile, base_seed=depth * 1000 + fl, threads=16)
    bpf = ch * oracle.bytes_per_sample(depth)
    with gpu_decoder_factory(cfg) as dec:
        got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
        disp = dec.last_dispatch()
    ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=16)
    helpers.assert_same_decode(cfg, ref, got, bpf, "n=%d fit %s" % (n, force))
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    q = disp["narrow_slots"] / n_cu
    if force is not None:
        assert disp["workgroups_per_cu"] == int(force) and not disp["gated"]
    else:  # widths without a gated twin (alac_gpu.h: decode_mode)
        want = 4 if q <= 4 else 5 if q <= 5 else 4 if q <= 7.5 else 5
        assert disp["workgroups_per_cu"] == want and not disp["gated"], (q, disp)
        if n_cu == 256 and profile != 3:  # the sizes above are chosen for an MI355X (STRESS batches hold irregular packets too)
            assert want == 5, q


def test_24bit_batch_just_above_four_rounds_at_full_frame_length(pkg, synth, oracle, gpu_decoder_factory):
    """66 000 x 24-bit stereo 4096-frame packets — eight wave slots more than 4 x CUs, which used to
    cost a whole second round (3.88 ms against 2.62 for 65 536) and now run as a fifth workgroup on some CUs. Checked by
    decode(encode(pcm)) == pcm (tests/conformance_test.go:282-291), frame counts and status words; the oracle decodes a slice."""
    import torch
    n, ch = 66000, 2
    cfg = oracle.make_config(4096, 24, ch)
    b = synth.gen_batch(cfg, n, threads=16)
    dev = torch.device("cuda:0")
    stride = 4096 * ch * 3
    d_blob = torch.from_numpy(b.blob).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
    d_out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    d_fr = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with gpu_decoder_factory(cfg) as dec:
        dec.decode_batch_device(d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr(), d_sz.data_ptr(), n, d_out.data_ptr(),
                                stride, d_fr.data_ptr(), d_st.data_ptr(), sync=True)
        disp = dec.last_dispatch()
    assert int(d_st.abs().sum()) == 0
    assert np.array_equal(d_fr.cpu().numpy().astype(np.uint32), b.frames)
    for lo in range(0, n, 4096):
        exp = torch.from_numpy(b.pcm[lo:lo + 4096]).to(dev)
        assert torch.equal(d_out[lo:lo + 4096], exp), lo
        del exp
    ref = oracle.decode_batch(cfg, b.blob, b.offsets[-128:], b.sizes[-128:], threads=8)
    assert np.array_equal(ref[0], d_out[-128:].cpu().numpy())
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert disp["workgroups_per_cu"] == 5 and disp["narrow_kernel"] == "alac_decode_24q", disp


@pytest.mark.parametrize("depth,ch,fl", [(16, 3, 200), (16, 6, 96), (24, 8, 80), (32, 4, 64), (16, 8, 33), (20, 5, 70), (24, 8, 4096),
                                         (16, 2, 4096), (24, 2, 512)])
def test_pairs_whose_difference_channel_needs_17_bits_on_gpu(pkg, synth, oracle, helpers, gpu_decoder_factory, depth, ch, fl):
    """Loud pairs in anti-phase: the difference channel of a matrixed pair uses all of its chanBits = depth - shift + 1 = 17 bits
    (matrix.go:40-41 inverted; UnpcBlock's sign extension, predictor.go:46), a range music-like signals never reach — through the
    split pipeline (more than two channels: scan, predictor pass over int32 rows, interleave) and through the wave workgroups of
    the stereo kernels, between quiet packets, in batches of several waves per key. (Written for round 4's int16 rows, which
    were measured and not kept: profiles/r04_final/experiments/rows16.txt.)"""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    n = 40 if fl == 4096 else 300
    pk = helpers.antiphase_packets(synth, cfg, n, seed=depth * 7 + ch, every=3)
    packets = [p for p, _ in pk]
    blob, offs, sizes = helpers.pack_packets(packets)
    ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
    for i, (_, pcm) in enumerate(pk):
        assert ref[0][i, :len(pcm)].tobytes() == pcm and ref[2][i] == 0, "oracle lost packet %d" % i
    with gpu_decoder_factory(cfg) as dec:
        got = _gpu_decode(dec, blob, offs, sizes)
    helpers.assert_same_decode(cfg, ref, got, bpf, "anti-phase pairs")


def test_gated_pairs_with_corrupt_packets_and_two_handles_at_once(pkg, synth, oracle, helpers, gpu_decoder_factory):
    """The gated kernel under the conditions the small-batch tests never reach it in: a batch of 84 000 packets of which
    every tenth is damaged (the status words of DynDecomp's error paths, golomb.go:157-163,196-199,239-245, must be the
    oracle's and the neighbours unharmed), decoded by two handles from two threads at the same time — their workgroups
    compete for the CUs, so neither finds the residency its gate assumes; the sweep has to pick up the rest."""
    import threading
    fl, n = 40, 84000
    cfg = oracle.make_config(fl, 16, 2)
    rng = np.random.default_rng(7)
    b = synth.gen_batch(cfg, n, profile=synth.PROFILE_MUSIC, threads=16)
    packets = [b.packet(i) for i in range(n)]
    bad = helpers.mutate_packets(b, rng, n // 10)
    for k, p in enumerate(bad):
        packets[k * 10 + 3] = p
    blob, offs, sizes = helpers.pack_dense(packets, lead=1)
    ref = oracle.decode_batch(cfg, *helpers.pack_packets(packets), threads=16)
    assert len(np.unique(ref[2])) > 3
    results, gated = [None, None], [False, False]

    def work(k):
        with gpu_decoder_factory(cfg) as dec:
            for _ in range(3):
                results[k] = _gpu_decode(dec, blob, offs, sizes)
            gated[k] = bool((dec.pair_placement()[:, 0] != 0).any())

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(2):
        helpers.assert_same_decode(cfg, ref, results[k], 4, "handle %d" % k)
    assert all(gated)  # the batch really went through alac_decode_16g


def test_host_entry_checks_untrusted_offsets(pkg, oracle, synth, helpers, gpu_decoder_factory, monkeypatch):
    """alacgpu_decode_batch takes its offsets from a sample table (internal/mp4/mp4.go:382-420: stco / stsz of a file
    nobody vouches for). Descriptors that leave the blob, or end before they start, must never be read: they get
    ALACGPU_ERR_RANGE, the packets around them decode as if nothing had happened. Small chunks, so that bad descriptors
    fall on chunk borders too; the blob sits at the end of its allocation in pageable and in pinned memory."""
    import torch
    cfg = oracle.make_config(256, 16, 2)
    b = synth.gen_batch(cfg, 600, threads=8)
    ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
    blob, offs, sizes = helpers.pack_dense([b.packet(i) for i in range(b.n)])
    offs1 = np.concatenate([offs, [np.uint64(len(blob))]]).astype(np.uint64)
    rng = np.random.default_rng(7)
    bad = offs1.copy()
    big = np.uint64(len(blob))
    # a table whose tail runs past the file, entries far outside, an entry that ends before it starts
    bad[590:] += np.uint64(3 * len(blob))
    for i in rng.choice(np.arange(5, 580), 40, replace=False):
        bad[i] = [big + np.uint64(1), np.uint64(2**63), np.uint64(2**64 - 1), bad[i - 1] - np.uint64(1) if bad[i - 1] else big * np.uint64(2)][int(rng.integers(0, 4))]
    lo, hi = bad[:-1], bad[1:]
    ok = (lo <= hi) & (hi <= big)
    assert 0 < (~ok).sum() < 200 and ok.sum() > 300
    # a good packet is one whose bytes are still exactly its own (both ends untouched)
    same = ok & (lo == offs1[:-1]) & (hi == offs1[1:])
    monkeypatch.setenv("ALACGPU_CHUNK_MB", "1")
    with gpu_decoder_factory(cfg) as dec:
        for pinned in (False, True):
            if pinned:
                t_blob = torch.empty(len(blob), dtype=torch.uint8, pin_memory=True)
                t_blob.numpy()[:] = blob
                hb = t_blob.numpy()
            else:
                hb = blob
            out, fr, st = dec.decode_batch(hb, bad)
            assert (st[~ok] == 7).all() and (fr[~ok] == 0).all(), "pinned %s" % pinned  # ALACGPU_ERR_RANGE
            got = (out[same], fr[same], st[same])
            want = (ref[0][same], ref[1][same], ref[2][same])
            helpers.assert_same_decode(cfg, want, got, 4, "pinned %s" % pinned)
            assert (st[ok] != 7).all()
        # and the handle is as good as new afterwards
        helpers.assert_same_decode(cfg, ref, dec.decode_batch(blob, offs1), 4, "after")


def test_decode_packet_in_the_shape_of_baseline_config_a(pkg, oracle, synth, gpu_decoder_factory):
    """BASELINE config a: PacketDecoder.DecodePacket (decoder.go:117) on ONE 16-bit / 44.1 kHz stereo packet of 4096
    frames — a batch of one through the same kernels — against the oracle, for a handful of packets of every
    signal profile (the reference reaches this call from Decoder.Read, decode.go:179)."""
    cfg = oracle.make_config(4096, 16, 2)
    pc = pkg.PacketConfig(FrameLength=4096, BitDepth=16, NumChannels=2)
    with pkg.NewPacketDecoder(pc) as dec:
        assert dec.Format().SampleRate == 44100
        for prof in (synth.PROFILE_MUSIC, synth.PROFILE_QUIET, synth.PROFILE_NOISE, synth.PROFILE_STRESS):
            b = synth.gen_batch(cfg, 6, profile=prof, threads=4)
            ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=4)
            for i in range(b.n):
                try:
                    pcm, st = dec.DecodePacket(b.packet(i)), 0
                except pkg.ErrDecode as e:
                    pcm, st = b"", e.status
                assert st == int(ref[2][i]), (prof, i)
                if st == 0:
                    assert len(pcm) == int(ref[1][i]) * 4
                    assert pcm == ref[0][i, :len(pcm)].tobytes(), (prof, i)
                    if prof != synth.PROFILE_STRESS:  # lossless (tests/conformance_test.go:282-291)
                        assert pcm == b.pcm[i, :len(pcm)].tobytes()


@pytest.mark.parametrize("depth,ch,fl", [(24, 2, 4096), (32, 1, 4096), (32, 2, 1024), (24, 1, 700)])
def test_wide_channels_on_the_wave_pair(pkg, oracle, synth, helpers, lane_sim, gpu_decoder_factory, depth, ch, fl):
    """24- and 32-bit streams WITHOUT shift bytes have chanBits 24..33 (decoder.go:371): the literal 32-bit predictor
    (predictor.go:46 with its wrapping products), sorted under the wide keys and decoded by alac_decode_w24 / _w32.
    NOSHIFT: every element without shift bytes; MIXED: independent predictor orders per channel (many sort keys)."""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    with gpu_decoder_factory(cfg) as dec:
        for prof, n in ((synth.PROFILE_MUSIC_NOSHIFT, 300), (synth.PROFILE_MUSIC_MIXED, 400)):
            b = synth.gen_batch(cfg, n, profile=prof, threads=8)
            ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=8)
            assert (ref[2] == 0).all()
            got = _gpu_decode(dec, b.blob, b.offsets, b.sizes)
            helpers.assert_same_decode(cfg, ref, got, bpf, "profile %d" % prof)
            if not (depth == 32 and ch == 2):  # 32-bit pairs without shift bytes: chanBits 33, every sample 0 (predictor.go:46)
                for i in range(b.n):
                    nb = int(b.frames[i]) * bpf
                    assert np.array_equal(got[0][i, :nb], b.pcm[i, :nb])
            # these packets are sorted under the wide keys (the classifier is the kernel's own code, built for the host),
            # which only alac_decode_w24 / _w32 decode
            keys = lane_sim(cfg, b.blob, b.offsets, b.sizes, variant=-1, want_classes=True)[3]
            wide = (keys >= 1024) & (keys < 2048)
            if prof == synth.PROFILE_MUSIC_NOSHIFT:
                assert wide.sum() > n // 2, "no wide keys"
            else:  # independent predictor orders per channel: many keys, i.e. many partly filled waves
                assert len(np.unique(keys[keys < 2048])) >= (6 if ch == 2 else 3)


def test_gated_pairs_on_full_length_packets(pkg, synth, oracle, helpers, gpu_decoder_factory):
    """70 000 x 16-bit stereo packets of 4096 frames: between four and five pairs per CU, the gated kernel's case, at
    the benchmark's frame length (the other gated tests use short frames). Checked by decode(encode(pcm)) == pcm over
    the whole batch and against the oracle on two slices."""
    import torch
    n, fl = 70000, 4096
    cfg = oracle.make_config(fl, 16, 2)
    b = synth.gen_batch(cfg, n, threads=16)
    dev = torch.device("cuda:0")
    stride = fl * 4
    d_blob = torch.from_numpy(b.blob).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
    d_out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    d_fr = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with gpu_decoder_factory(cfg) as dec:
        dec.decode_batch_device(d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr(), d_sz.data_ptr(), n, d_out.data_ptr(),
                                stride, d_fr.data_ptr(), d_st.data_ptr(), sync=True)
        place = dec.pair_placement()
    assert int(d_st.abs().sum()) == 0
    assert np.array_equal(d_fr.cpu().numpy().astype(np.uint32), b.frames)
    for lo in range(0, n, 8192):
        exp = torch.from_numpy(b.pcm[lo:lo + 8192]).to(dev)
        assert torch.equal(d_out[lo:lo + 8192], exp), lo
        del exp
    for lo in (0, n - 200):
        ref = oracle.decode_batch(cfg, b.blob, b.offsets[lo:lo + 200], b.sizes[lo:lo + 200], threads=8)
        assert np.array_equal(ref[0], d_out[lo:lo + 200].cpu().numpy())
    tags = place[:, 0]
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    if 4 * n_cu < len(tags) <= 6 * n_cu:
        owned = tags[tags != 0]
        assert len(owned) >= len(tags) - 24 and (owned >> 31).all()


@pytest.mark.parametrize("side", ["2", "1", "0"])
def test_irregular_packets_beside_the_regular_ones_back_to_back(pkg, synth, oracle, helpers, gpu_decoder_factory, monkeypatch,
                                                                side):
    """launch() forks the irregular packets' kernels (alac_scan, alac_interleave, alac_legacy) onto a stream of their own
    beside the regular packets' workgroups and joins before the stop event (ALACGPU_SIDE=0: all in line, the order of
    rounds 1-3; 1: batches of up to 6 x CUs wave slots; 2: always). Two different batches with escape, damaged and order-17+ packets among regular ones go through ONE handle's
    device entry six times back to back without a host synchronisation in between (each launch clears and re-sorts the
    plan the previous launch's side kernels were still reading if the join did not hold); every result must be the
    oracle's (decoder.go:142-203: element walk, escape elements; status words of the damaged packets)."""
    import torch
    monkeypatch.setenv("ALACGPU_SIDE", side)
    fl, n = 352, 3000
    cfg = oracle.make_config(fl, 16, 2)
    rng = np.random.default_rng(11)
    dev = torch.device("cuda:0")
    batches = []
    for k, prof in enumerate((synth.PROFILE_STRESS, synth.PROFILE_MUSIC)):
        b = synth.gen_batch(cfg, n, profile=prof, base_seed=900 + k, threads=8)
        packets = [b.packet(i) for i in range(n)]
        noise = synth.gen_batch(cfg, n // 8, profile=synth.PROFILE_NOISE, base_seed=77 + k, threads=8)
        for j in range(n // 8):
            packets[j * 8 + 5] = noise.packet(j)  # escape elements: scan + interleave
        for j, p in enumerate(helpers.mutate_packets(b, rng, n // 16)):
            packets[j * 16 + 2] = p
        blob, offs, sizes = helpers.pack_packets(packets)
        ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=8)
        stride = fl * 4
        t = dict(blob=torch.from_numpy(np.ascontiguousarray(blob)).to(dev), off=torch.from_numpy(offs.astype(np.int64)).to(dev),
                 sz=torch.from_numpy(sizes.astype(np.int32)).to(dev), ref=ref, stride=stride,
                 outs=[(torch.zeros((n, stride), dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                        torch.full((n,), -1, dtype=torch.int32, device=dev)) for _ in range(3)])
        batches.append(t)
    torch.cuda.synchronize()
    with gpu_decoder_factory(cfg) as dec:
        for r in range(3):
            for t in batches:
                o, f, s = t["outs"][r]
                dec.decode_batch_device(t["blob"].data_ptr(), t["blob"].numel(), t["off"].data_ptr(), t["sz"].data_ptr(), n,
                                        o.data_ptr(), t["stride"], f.data_ptr(), s.data_ptr(), sync=False)
        dec.synchronize()
    for k, t in enumerate(batches):
        assert len(np.unique(t["ref"][2])) > 2  # damaged packets among them
        for r in range(3):
            o, f, s = t["outs"][r]
            got = (o.cpu().numpy(), f.cpu().numpy().astype(np.uint32), s.cpu().numpy())
            helpers.assert_same_decode(cfg, t["ref"], got, 4, "batch %d round %d side %s" % (k, r, side))
