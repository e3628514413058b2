"""Regenerates tests/golden/golden_packets.json.

The reference (Go) cannot be built or run in this image and ships no fixtures, so these vectors are
produced by THIS repo's tools: packets by the synthetic encoder (saprobe-alac_amd/synth), expected
PCM / frame count / status by the CPU oracle (oracle/alac_oracle.c), which is itself pinned by
tests/golden/kat.json and by the lossless round trip. They freeze today's behaviour so that a later
change to oracle, encoder or kernel that alters any byte is caught. Run: python tests/golden/make_golden.py
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

synth = importlib.import_module("saprobe-alac_amd.synth")

CASES = [
    # (frame_length, depth, channels, profile, n)
    (64, 16, 2, synth.PROFILE_MUSIC, 4), (64, 16, 2, synth.PROFILE_STRESS, 6), (48, 24, 2, synth.PROFILE_MUSIC, 3),
    (48, 24, 2, synth.PROFILE_STRESS, 4), (40, 16, 1, synth.PROFILE_QUIET, 3), (32, 20, 2, synth.PROFILE_STRESS, 3),
    (32, 32, 2, synth.PROFILE_MUSIC, 3), (24, 24, 8, synth.PROFILE_MUSIC, 2), (24, 16, 6, synth.PROFILE_STRESS, 3),
    (40, 16, 2, synth.PROFILE_NOISE, 2), (33, 24, 3, synth.PROFILE_STRESS, 3), (16, 32, 5, synth.PROFILE_STRESS, 3),
]


def main():
    vectors = []
    rng = np.random.default_rng(20260101)
    for fl, depth, ch, prof, n in CASES:
        cfg = oracle.make_config(fl, depth, ch)
        b = synth.gen_batch(cfg, n, profile=prof, base_seed=0x601DE2, threads=1)
        packets = [b.packet(i) for i in range(n)]
        # one corrupted copy per case so error statuses are frozen too
        bad = bytearray(packets[0])
        bad[int(rng.integers(len(bad)))] ^= 0x5a
        packets.append(bytes(bad))
        packets.append(packets[0][: max(1, len(packets[0]) // 2)])
        for j, p in enumerate(packets):
            st, frames, pcm = oracle.decode_packet(cfg, p)
            vectors.append({
                "frame_length": fl, "bit_depth": depth, "num_channels": ch, "profile": prof, "index": j,
                "packet": p.hex(), "status": int(st), "frames": int(frames),
                "pcm_sha256": hashlib.sha256(pcm).hexdigest(),
                "pcm": pcm.hex() if len(pcm) <= 512 else None,
            })
    out = {"_generator": "tests/golden/make_golden.py (oracle + synthetic encoder of this repo)",
           "config_common": {"pb": 40, "mb": 10, "kb": 14, "max_run": 255}, "vectors": vectors}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_packets.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, len(vectors), "vectors", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
