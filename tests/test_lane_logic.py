"""The kernel's decode logic (csrc/alac_wave.h, compiled for the host by tests/host_sim)
against the oracle: valid streams of every shape, then corrupted ones (status words must agree too)."""
import numpy as np
import pytest

CONFIGS = [(16, 2, 4096), (24, 2, 1024), (16, 1, 512), (24, 8, 256), (20, 3, 300), (32, 2, 512), (16, 5, 33),
           (32, 1, 100), (20, 2, 64), (16, 7, 40), (24, 4, 77), (16, 2, 1),
           # every layout whose frame is a whole number of dwords (interleave_frame_packed)
           (16, 4, 40), (16, 6, 64), (16, 8, 48), (32, 3, 30), (32, 4, 36), (32, 5, 24), (32, 6, 28), (32, 7, 20), (32, 8, 24)]


@pytest.mark.parametrize("depth,ch,fl", CONFIGS)
def test_lane_matches_oracle_on_valid_streams(oracle, synth, lane_sim, helpers, depth, ch, fl):
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    for prof in (synth.PROFILE_MUSIC, synth.PROFILE_NOISE, synth.PROFILE_QUIET, synth.PROFILE_STRESS):
        b = synth.gen_batch(cfg, 96, profile=prof, threads=4)
        ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=4)
        # every class variant must decode every packet (the order class is a speed choice only) ...
        for variant in (-1, -2, 0, 1, 2, 3):
            got = lane_sim(cfg, b.blob, b.offsets, b.sizes, variant=variant)
            helpers.assert_same_decode(cfg, ref, got, bpf, "profile %d variant %d" % (prof, variant))
        # ... and so must the unaligned-output path (direct stores instead of the LDS stager)
        got = lane_sim(cfg, b.blob, b.offsets, b.sizes, stride_pad=4)
        helpers.assert_same_decode(cfg, ref, got, bpf, "profile %d unaligned" % prof)


@pytest.mark.parametrize("depth,ch,fl,kb", [(16, 2, 256, 14), (24, 2, 128, 14), (16, 1, 64, 14), (24, 8, 32, 14),
                                            (20, 3, 50, 14), (32, 2, 64, 14), (16, 5, 33, 14), (16, 2, 256, 0),
                                            (24, 6, 16, 3), (16, 2, 8, 255), (32, 8, 5, 14), (16, 2, 256, 32),
                                            (16, 2, 256, 255), (24, 5, 64, 40), (16, 6, 40, 14),
                                            (16, 8, 24, 14), (32, 4, 20, 14), (24, 4, 32, 14), (32, 7, 16, 14)])
def test_lane_matches_oracle_on_corrupt_packets(oracle, synth, lane_sim, helpers, depth, ch, fl, kb):
    cfg = oracle.make_config(fl, depth, ch, kb=kb)
    bpf = ch * oracle.bytes_per_sample(depth)
    rng = np.random.default_rng(depth * 1000 + ch * 10 + kb)
    # QUIET: zero runs, whose multiplier is masked with WB = (1 << KB) - 1 (golomb.go:60,227): KB >= 32 must give all ones
    for prof in (synth.PROFILE_MUSIC, synth.PROFILE_STRESS, synth.PROFILE_QUIET):
        b = synth.gen_batch(cfg, 48, profile=prof, threads=4)
        if kb >= 32:  # the intact packets too: they reach the lean wave-pair path (fl > 32) with KB >= 32
            ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=4)
            helpers.assert_same_decode(cfg, ref, lane_sim(cfg, b.blob, b.offsets, b.sizes, variant=-1), bpf, "kb %d" % kb)
            assert prof == synth.PROFILE_STRESS or fl <= 32 or (ref[2] == 0).all()
        blob, offs, sizes = helpers.pack_packets(helpers.mutate_packets(b, rng, 400))
        ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=4)
        for variant in (-1, -2, 0, 3):
            got = lane_sim(cfg, blob, offs, sizes, variant=variant)
            helpers.assert_same_decode(cfg, ref, got, bpf, "fuzz profile %d variant %d" % (prof, variant))
        assert len(np.unique(ref[2])) > 3  # the corpus really reaches several error classes


@pytest.mark.parametrize("pb", [0, 1, 20, 39, 41, 73, 74, 127, 255])
def test_lane_matches_oracle_on_other_cookie_bytes(oracle, synth, lane_sim, helpers, pb):
    """PB and MB come from the file's magic cookie (config.go:72-73) and are not 40 / 10 by any law. lean_config
    (alac_regular.h) lets PB <= 73 take the lean Golomb step — effective pb = PB * pbFactor / 4 <= 127
    (decoder.go:296-299), where mean < 2^25 + 512 and pb * mean cannot wrap — and sends larger ones to the whole-packet
    decoder with the reference's literal uint32 arithmetic (golomb.go:215): both sides of that edge, the extremes, and
    MB 0 / 1 / 255 as the first mean (golomb.go:157), on MUSIC / QUIET / NOISE / STRESS (pbFactor 0..7) streams, damaged
    packets, and `loud` packets that hold the mean at the top of its range with pbFactor 7."""
    rng = np.random.default_rng(pb)
    for mb in (0, 1, 10, 255):
        for depth, ch, fl in ((16, 2, 300), (16, 1, 200), (24, 2, 128), (16, 6, 64), (20, 2, 96), (32, 1, 80)):
            cfg = oracle.make_config(fl, depth, ch, pb=pb, mb=mb)
            bpf = ch * oracle.bytes_per_sample(depth)
            for prof in (synth.PROFILE_MUSIC, synth.PROFILE_QUIET, synth.PROFILE_STRESS, synth.PROFILE_NOISE):
                b = synth.gen_batch(cfg, 32, profile=prof, base_seed=pb * 1000 + mb, threads=4)
                ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=4)
                for variant in (-1, -2, 3):
                    got = lane_sim(cfg, b.blob, b.offsets, b.sizes, variant=variant)
                    helpers.assert_same_decode(cfg, ref, got, bpf, "pb %d mb %d profile %d variant %d" % (pb, mb, prof, variant))
                if prof != synth.PROFILE_STRESS:  # lossless under any cookie (tests/conformance_test.go:282-291)
                    assert (ref[2] == 0).all()
                    for i in range(b.n):
                        assert np.array_equal(ref[0][i, :int(b.frames[i]) * bpf], b.pcm[i, :int(b.frames[i]) * bpf])
                blob, offs, sizes = helpers.pack_packets(helpers.mutate_packets(b, rng, 100))
                ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=4)
                for variant in (-1, 3):
                    got = lane_sim(cfg, blob, offs, sizes, variant=variant)
                    helpers.assert_same_decode(cfg, ref, got, bpf, "damaged, pb %d mb %d profile %d variant %d" % (pb, mb, prof, variant))
            if mb == 10:
                loud = helpers.loud_packets(synth, cfg, 12, seed=pb + depth)
                blob, offs, sizes = helpers.pack_packets([q for q, _ in loud])
                ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=4)
                assert (ref[2] == 0).all()
                for i, (_, pcm) in enumerate(loud):
                    assert ref[0][i, :len(pcm)].tobytes() == pcm
                for variant in (-1, -2, 3):
                    helpers.assert_same_decode(cfg, ref, lane_sim(cfg, blob, offs, sizes, variant=variant), bpf,
                                               "loud, pb %d variant %d" % (pb, variant))


@pytest.mark.parametrize("depth,ch", [(32, 1), (24, 2), (24, 1), (20, 1)])
def test_runs_of_escape_codes_up_to_a_truncation_point(oracle, synth, lane_sim, helpers, depth, ch):
    """Streams without shift bytes at 24 / 32 bits are escape code after escape code (9 + chanBits = up to 41 bits each,
    taken inside the lean step: alac_regular.h: gol_step, ESC); the lean step's bounds live in `near`, which is refreshed
    every eight steps for the GOL_REACH = 8 x 41 bits a lane can move in between. Packets cut at every byte of their last
    stretch: the status (overrun where the reference's loop test fails, malformed where getStreamBits would read outside,
    golomb.go:168,86-108) and the frames in front must be the oracle's."""
    fl = 96
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    b = synth.gen_batch(cfg, 6, profile=synth.PROFILE_MUSIC_NOSHIFT, threads=2)
    packets = []
    for i in range(b.n):
        p = b.packet(i)
        packets += [p[:k] for k in range(max(1, len(p) - 90), len(p) + 1)]
    blob, offs, sizes = helpers.pack_packets(packets)
    ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=4)
    assert len(np.unique(ref[2])) >= 2
    for variant in (-1, 3):
        helpers.assert_same_decode(cfg, ref, lane_sim(cfg, blob, offs, sizes, variant=variant), bpf, "variant %d" % variant)


@pytest.mark.parametrize("depth,ch,fl", [(16, 2, 33), (16, 2, 47), (16, 2, 1000), (16, 1, 4095), (24, 2, 129),
                                         (20, 1, 65), (32, 2, 200), (16, 2, 4097)])
def test_wave_pair_chunk_tails(oracle, synth, lane_sim, helpers, depth, ch, fl):
    """alac_duo.h works in chunks of 16 steps and groups of 4 / 8: frame lengths around those multiples."""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    for prof in (synth.PROFILE_MUSIC, synth.PROFILE_STRESS):
        b = synth.gen_batch(cfg, 48, profile=prof, base_seed=fl * 131 + depth, threads=4)
        ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=4)
        got = lane_sim(cfg, b.blob, b.offsets, b.sizes, variant=-1)
        helpers.assert_same_decode(cfg, ref, got, bpf, "profile %d" % prof)


@pytest.mark.parametrize("depth,ch,fl", [(16, 2, 256), (24, 2, 128), (16, 1, 64), (24, 8, 48), (32, 2, 64), (20, 3, 50)])
def test_dense_blob_with_hostile_neighbours(oracle, synth, lane_sim, helpers, depth, ch, fl):
    """Packets back to back as in an mdat (internal/mp4/mp4.go:382-420): no zero pad, any alignment, the blob ends with
    the last byte of the last packet. Truncated packets then have the NEXT packet's (non-zero) bytes where the
    reference sees its 4 zero pad bytes (bitbuffer.go:33): results must not change. The blob sits against an
    inaccessible page (guard), so a read beyond it is fatal."""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    rng = np.random.default_rng(depth * 7 + ch)
    for prof in (synth.PROFILE_MUSIC, synth.PROFILE_STRESS, synth.PROFILE_QUIET):
        b = synth.gen_batch(cfg, 40, profile=prof, threads=4)
        packets = []
        for i in range(b.n):
            p = b.packet(i)
            packets.append(p)
            packets.append(p[:int(rng.integers(1, len(p)))])          # truncated: its neighbour starts right behind it
            packets.append(p[:max(1, len(p) - int(rng.integers(1, 9)))])  # cut inside the last bytes
        packets += helpers.mutate_packets(b, rng, 60)
        packets.append(b.packet(0))                                    # an intact packet ends the blob
        ref_blob, ref_offs, ref_sizes = helpers.pack_packets(packets)
        ref = oracle.decode_batch(cfg, ref_blob, ref_offs, ref_sizes, threads=4)
        for lead in (0, 1, 2, 3):
            blob, offs, sizes = helpers.pack_dense(packets, lead=lead)
            for variant in (-1, -2, 3):
                got = lane_sim(cfg, blob, offs, sizes, variant=variant, guard=True)
                helpers.assert_same_decode(cfg, ref, got, bpf, "dense lead %d variant %d profile %d" % (lead, variant, prof))
        # ... and with a truncated packet as the very last thing in the blob
        packets.append(b.packet(1)[:len(b.packet(1)) // 2])
        ref_blob, ref_offs, ref_sizes = helpers.pack_packets(packets)
        ref = oracle.decode_batch(cfg, ref_blob, ref_offs, ref_sizes, threads=4)
        blob, offs, sizes = helpers.pack_dense(packets, lead=1)
        got = lane_sim(cfg, blob, offs, sizes, variant=-1, guard=True)
        helpers.assert_same_decode(cfg, ref, got, bpf, "dense, truncated tail")


@pytest.mark.parametrize("depth,ch,fl", [(24, 2, 300), (32, 2, 200), (24, 1, 128), (32, 1, 100)])
def test_wide_channels_take_the_wave_pair(oracle, synth, lane_sim, helpers, depth, ch, fl):
    """24/32-bit streams without shift bytes: chanBits 24/25 and 32/33 (decoder.go:230,371; chanShift wraps for 33,
    predictor.go:46). classify_regular keys them KEY_WIDE and they run on the wave pair with predict_wide."""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    for prof in (synth.PROFILE_MUSIC_NOSHIFT, synth.PROFILE_MUSIC_MIXED):
        b = synth.gen_batch(cfg, 64, profile=prof, threads=4)
        ref = oracle.decode_batch(cfg, b.blob, b.offsets, b.sizes, threads=4)
        got = lane_sim(cfg, b.blob, b.offsets, b.sizes, variant=-1, want_classes=True)
        helpers.assert_same_decode(cfg, ref, got[:3], bpf, "profile %d" % prof)
        keys = got[3]
        if prof == synth.PROFILE_MUSIC_NOSHIFT:
            assert ((keys >= 1024) & (keys < 2048)).sum() > 48  # the wide keys, not the scan / whole-packet route
            if depth == 24 or ch == 1:  # 32-bit pairs have chanBits 33: every sample is 0 there (predictor.go:46)
                assert (ref[2] == 0).all() and np.array_equal(ref[0][:, :fl * bpf], b.pcm[:, :fl * bpf])
        else:
            assert len(np.unique(keys[keys < 1024])) >= (6 if ch == 2 else 3)


@pytest.mark.parametrize("depth,ch,fl", [(16, 3, 200), (16, 6, 96), (24, 8, 80), (32, 4, 64), (16, 8, 33), (20, 5, 70)])
def test_pairs_whose_difference_channel_needs_17_bits(oracle, synth, lane_sim, helpers, depth, ch, fl):
    """A matrixed pair's channels have chanBits = depth - shift + 1: loud pairs in anti-phase use all 17 bits of the difference
    channel (matrix.go:40-41 inverted) and the whole range of the mid / side arithmetic, which music-like signals never reach
    (UnpcBlock's sign extension to chanBits, predictor.go:46,99-127; WriteStereo*, matrix.go:30). Multi-channel streams: the split
    pipeline's int32 rows. (Written for round 4's int16 rows — profiles/r04_final/experiments/rows16.txt —, kept as a parity case.)"""
    cfg = oracle.make_config(fl, depth, ch)
    bpf = ch * oracle.bytes_per_sample(depth)
    pk = helpers.antiphase_packets(synth, cfg, 24, seed=depth + ch)
    blob, offs, sizes = helpers.pack_packets([p for p, _ in pk])
    ref = oracle.decode_batch(cfg, blob, offs, sizes, threads=4)
    for i, (_, pcm) in enumerate(pk):
        assert ref[0][i, :len(pcm)].tobytes() == pcm and ref[2][i] == 0, "oracle lost packet %d" % i
    for variant in (-1, -2, 3):
        got = lane_sim(cfg, blob, offs, sizes, variant=variant)
        helpers.assert_same_decode(cfg, ref, got, bpf, "anti-phase pairs, variant %d" % variant)


def test_lane_logic_reproduces_the_hand_derived_packets(oracle, lane_sim):
    """K1..K23 (tests/golden/kat*.json, derived on paper from the reference source) through the kernel's decode logic
    as built for the host: every routing the GPU library can take for them."""
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("kat.json", "kat2.json", "kat3.json", "kat4.json"):
        k = json.load(open(os.path.join(here, "golden", name)))
        c = k["config_common"]
        for v in k["vectors"]:
            depth = v.get("bit_depth", c.get("bit_depth"))
            cfg = oracle.make_config(v["frame_length"], depth, v["num_channels"], v.get("pb", c.get("pb")), v.get("mb", c.get("mb")), v.get("kb", c["kb"]),
                                     c["max_run"])
            pkt = np.frombuffer(bytes.fromhex(v["packet"].replace(" ", "")), np.uint8)
            want = bytes.fromhex(v["pcm"].replace(" ", ""))
            for variant in (-1, -2, 3):
                out, fr, st = lane_sim(cfg, pkt, np.array([0], np.uint64), np.array([len(pkt)], np.uint32), variant=variant)
                assert st[0] == 0 and fr[0] == v.get("frames", v["frame_length"]), (v["name"], variant)
                assert out[0, :len(want)].tobytes() == want, (v["name"], variant)
