"""oracle/goref.py (the pure-Python transliteration of the reference, written independently of the C oracle) against
the committed vectors, the hand-derived KATs and the C oracle on seeded random packets. Four readings of the
reference — oracle, goref, hand derivations, HIP kernels — must agree; this file covers the first three on the CPU
(`tests/golden/crosscheck_goref.py` runs the same comparison over 10 000+ packets)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import goref

HERE = os.path.dirname(os.path.abspath(__file__))


def test_goref_reproduces_the_golden_vectors():
    g = json.load(open(os.path.join(HERE, "golden", "golden_packets.json")))
    c = g["config_common"]
    for v in g["vectors"]:
        cfg = goref.PacketConfig(v["frame_length"], v["bit_depth"], v["num_channels"], c["pb"], c["mb"], c["kb"], c["max_run"])
        pcm, frames, st = goref.decode_packet(cfg, bytes.fromhex(v["packet"]))
        assert (st, frames) == (v["status"], v["frames"]), v
        assert hashlib.sha256(pcm).hexdigest() == v["pcm_sha256"]


def test_goref_reproduces_the_hand_derived_kats():
    k = json.load(open(os.path.join(HERE, "golden", "kat.json")))
    c = k["config_common"]
    for v in k["vectors"]:
        cfg = goref.PacketConfig(v["frame_length"], c["bit_depth"], v["num_channels"], c["pb"], c["mb"], c["kb"], c["max_run"])
        pcm, frames, st = goref.decode_packet(cfg, bytes.fromhex(v["packet"].replace(" ", "")))
        assert st == 0 and pcm.hex().upper() == v["pcm"].upper(), v["name"]
    for name in ("kat2.json", "kat3.json", "kat4.json"):  # kat4: PB per vector
        k = json.load(open(os.path.join(HERE, "golden", name)))
        c = k["config_common"]
        for v in k["vectors"]:
            cfg = goref.PacketConfig(v["frame_length"], v["bit_depth"], v["num_channels"], v.get("pb", c.get("pb")), v["mb"], v.get("kb", c["kb"]),
                                     c["max_run"])
            pcm, frames, st = goref.decode_packet(cfg, bytes.fromhex(v["packet"]))
            assert st == 0 and frames == v.get("frames", v["frame_length"]), v["name"]
            assert pcm.hex().upper() == v["pcm"].upper(), v["name"]


@pytest.mark.parametrize("fl,depth,ch,kb", [(64, 16, 2, 14), (48, 24, 2, 14), (40, 16, 1, 14), (32, 20, 2, 14),
                                            (32, 32, 2, 14), (24, 24, 8, 14), (33, 24, 3, 14), (16, 32, 5, 14),
                                            (64, 16, 2, 32), (64, 16, 2, 255), (40, 16, 2, 0)])
def test_goref_agrees_with_the_c_oracle(oracle, synth, helpers, fl, depth, ch, kb):
    """Status word, frame count and every PCM byte, on valid and on corrupted packets (error statuses included)."""
    cfg_o = oracle.make_config(fl, depth, ch, kb=kb)
    cfg_g = goref.PacketConfig(fl, depth, ch, 40, 10, kb, 255)
    rng = np.random.default_rng(fl * 1000 + depth * 10 + ch)
    n_ok = 0
    for prof in (synth.PROFILE_STRESS, synth.PROFILE_MUSIC, synth.PROFILE_QUIET, synth.PROFILE_NOISE):
        if kb == 0 and prof != synth.PROFILE_STRESS:
            continue
        b = synth.gen_batch(cfg_o, 12, profile=prof, base_seed=0xC0FFEE + prof, threads=2)
        for p in [b.packet(i) for i in range(b.n)] + helpers.mutate_packets(b, rng, 12):
            st, frames, pcm = oracle.decode_packet(cfg_o, p)
            info = {}
            g_pcm, g_frames, g_st = goref.decode_packet(cfg_g, p, info=info)
            if st == 6 and info["cpe_last_slot"]:
                continue  # documented deviation: a pair that does not fit the frame
            assert (st, frames) == (g_st, g_frames), p.hex()
            assert pcm == g_pcm, p.hex()
            n_ok += st == 0
    assert n_ok >= 12


@pytest.mark.parametrize("pb,mb", [(0, 10), (1, 0), (20, 255), (39, 1), (41, 10), (73, 0), (74, 255), (127, 10), (255, 1)])
def test_goref_agrees_with_the_c_oracle_on_other_cookie_bytes(oracle, synth, helpers, pb, mb):
    """PB and MB are bytes of the magic cookie (config.go:72-73): an untrusted file sets them. Effective pb =
    PB * pbFactor / 4 (decoder.go:296-299) reaches 446 for PB 255 with pbFactor 7 (STRESS varies pbFactor 0..7), where
    pb * mean wraps in uint32 (golomb.go:215) — both restatements must wrap the same way. NOISE keeps the mean high,
    QUIET walks in and out of zero runs with MB as the first mean (golomb.go:157)."""
    rng = np.random.default_rng(pb * 256 + mb)
    n_ok = 0
    for fl, depth, ch in ((48, 16, 2), (40, 24, 2), (32, 16, 1), (24, 20, 6), (32, 32, 1)):
        cfg_o = oracle.make_config(fl, depth, ch, pb=pb, mb=mb)
        cfg_g = goref.PacketConfig(fl, depth, ch, pb, mb, 14, 255)
        for prof in (synth.PROFILE_STRESS, synth.PROFILE_MUSIC, synth.PROFILE_QUIET, synth.PROFILE_NOISE):
            b = synth.gen_batch(cfg_o, 6, profile=prof, base_seed=0xBEE5 + prof + pb, threads=2)
            for p in [b.packet(i) for i in range(b.n)] + helpers.mutate_packets(b, rng, 6):
                st, frames, pcm = oracle.decode_packet(cfg_o, p)
                info = {}
                g_pcm, g_frames, g_st = goref.decode_packet(cfg_g, p, info=info)
                if st == 6 and info["cpe_last_slot"]:
                    continue
                assert (st, frames) == (g_st, g_frames), p.hex()
                assert pcm == g_pcm, p.hex()
                n_ok += st == 0
    assert n_ok >= 60
