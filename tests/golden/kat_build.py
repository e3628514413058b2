"""Assembles the hand-built known-answer packets K5..K13 into tests/golden/kat2.json, K14..K19 into kat3.json and K20..K22
(cookie bytes PB / MB other than 40 / 10) into kat4.json.

This script only PACKS BITS: every field below was chosen by hand, every expected PCM byte string was worked out on
paper from the reference source (tests/golden/kat_derivation.md holds the derivations, step by step, with the reference
lines they follow) and is typed in here as a constant. Nothing in this file decodes anything; no decoder (oracle,
goref, kernel) was used to produce the expectations. tests/test_oracle.py, tests/test_goref.py and
tests/test_gpu_parity.py then require the C oracle, the Python transliteration and the HIP kernels to reproduce them.

    python tests/golden/kat_build.py        # rewrites kat2.json, kat3.json and kat4.json
"""
import json
import os


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, value, n):
        value &= (1 << n) - 1
        for i in range(n - 1, -1, -1):
            self.bits.append((value >> i) & 1)
        return self

    def raw(self, s):
        self.bits += [int(c) for c in s if c in "01"]
        return self

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)
        return self

    def hex(self):
        self.align()
        out = bytearray()
        for i in range(0, len(self.bits), 8):
            b = 0
            for j in range(8):
                b = (b << 1) | self.bits[i + j]
            out.append(b)
        return out.hex().upper()


# static code used by most KATs: cookie MB=255 and pbFactor=0 keep the mean at 255, so m = 0, k = 1 and no zero run is
# ever tested (255 << 2 >= 512): a residual is `nd` ones and a zero (nd <= 8), or nine ones and nd as a chanBits-bit
# literal; nd = 2*del for del >= 0, -2*del - 1 for del < 0 (golomb.go:172-218; kat_derivation.md §0)
def unary(bw, dels, chan_bits):
    for d in dels:
        nd = 2 * d if d >= 0 else -2 * d - 1
        if nd <= 8:
            bw.raw("1" * nd + "0")
        else:
            bw.raw("1" * 9).put(nd, chan_bits)


def elem_header(bw, tag, partial=0, bs=0, escape=0, num_samples=None):
    bw.put(tag, 3).put(0, 4).put(0, 12).put((partial << 3) | (bs << 1) | escape, 4)
    if partial:
        bw.put(num_samples, 32)


def chan_header(bw, mode, den_shift, pb_factor, coefs):
    bw.put(mode, 4).put(den_shift, 4).put(pb_factor, 3).put(len(coefs), 5)
    for c in coefs:
        bw.put(c, 16)


def le(values, nbytes):
    out = bytearray()
    for v in values:
        v &= (1 << (8 * nbytes)) - 1
        out += v.to_bytes(nbytes, "little")
    return out.hex().upper()


def build():
    kats = []

    # ---- K5: 16-bit CPE, order 4 on both channels, negative mixRes (kat_derivation.md §K5) -----------------------
    bw = BitWriter()
    elem_header(bw, 1)
    bw.put(2, 8).put(-3, 8)                         # mixBits 2, mixRes -3
    chan_header(bw, 0, 1, 0, [3, -2, 1, 1])          # U
    chan_header(bw, 0, 1, 0, [-1, 2, 0, 1])          # V
    unary(bw, [50, 4, -3, 2, -4, 3, -40, 0, 100, -2], 17)
    unary(bw, [3, 1, 1, -2, 3, -1, 2, 0, -3, 4], 17)
    bw.put(7, 3)
    lr = [56, 53, 61, 57, 60, 55, 59, 56, 60, 54, 53, 52, 26, 16, -43, -38, -2, -19, -92, -80]
    kats.append({"name": "K5 CPE order 4 both signs, negative mixRes", "frame_length": 10, "bit_depth": 16,
                 "num_channels": 2, "mb": 255, "packet": bw.hex(), "pcm": le(lr, 2)})

    # ---- K6: 16-bit SCE, order 6 -----------------------------------------------------------------------------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 2, 0, [5, -3, 2, -1, 1, -2])
    unary(bw, [10, 3, -2, 4, -1, 2, -3, 4, -3, -30, 2], 16)
    bw.put(7, 3)
    kats.append({"name": "K6 SCE order 6", "frame_length": 11, "bit_depth": 16, "num_channels": 1, "mb": 255,
                 "packet": bw.hex(), "pcm": le([10, 13, 11, 15, 14, 16, 13, 13, 14, -14, -11], 2)})

    # ---- K7: 3 channels: SCE order 8, then CPE with order 0 (U) and 31 (V), mixRes 0 ---------------------------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 3, 0, [8, -7, 6, -5, 4, -3, 2, -1])
    unary(bw, [5, -1, 2, -2, 3, -3, 4, -4, 1, 3, -9, 2], 16)
    elem_header(bw, 1)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 0, [])
    chan_header(bw, 0, 5, 0, list(range(1, 32)))     # numActive 31: 31 coefficient words are read and ignored
    unary(bw, [1, -1, 2, -2, 3, -3, 4, -4, 0, 1, -1, 2], 17)
    unary(bw, [100, 1, 1, 1, -2, -2, 3, 0, -4, 4, -1, 2], 17)
    bw.put(7, 3)
    c = [5, 4, 6, 4, 7, 4, 8, 4, 5, 14, -1, -2]
    l = [1, -1, 2, -2, 3, -3, 4, -4, 0, 1, -1, 2]
    r = [100, 101, 102, 103, 101, 99, 102, 102, 98, 102, 101, 103]
    frames = []
    for i in range(12):
        frames += [l[i], r[i], c[i]]                 # SMPTE order L R C (decoder.go:58)
    kats.append({"name": "K7 3-channel layout: SCE order 8, CPE orders 0 / 31", "frame_length": 12, "bit_depth": 16,
                 "num_channels": 3, "mb": 255, "packet": bw.hex(), "pcm": le(frames, 2)})

    # ---- K8: general predictor, order 2, int16 coefficient wrap in both directions ---------------------------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 15, 0, [32767, -32768])
    unary(bw, [3, 2, -1, 2, -1, -3, 1, 0], 16)
    bw.put(7, 3)
    kats.append({"name": "K8 general order 2 with int16 wrap", "frame_length": 8, "bit_depth": 16, "num_channels": 1,
                 "mb": 255, "packet": bw.hex(), "pcm": le([3, 5, 4, 4, 6, -1, -2, 5], 2)})

    # ---- K9: mode != 0 double pass, adaptive Golomb with k = 1 and k = 2 (standard cookie) -------------------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 1, 1, 4, [2, -1, 1, -1])
    bw.raw("111111110")      # s0: k=1, nd=8
    bw.raw("111111110")      # s1: k=1, nd=8
    bw.raw("10 10")          # s2: k=2, pre=1, v=2 -> n=4
    bw.raw("110 0")          # s3: k=2, pre=2, v<2 -> n=6, one bit consumed
    bw.raw("10 11")          # s4: k=2, pre=1, v=3 -> n=5
    bw.raw("0 0")            # s5: k=2, pre=0, v<2 -> n=0
    bw.raw("110 10")         # s6: k=2, pre=2, v=2 -> n=7
    bw.raw("0 10")           # s7: k=2, pre=0, v=2 -> n=1
    bw.put(7, 3)
    kats.append({"name": "K9 mode 1 double pass, adaptive Golomb k=2", "frame_length": 8, "bit_depth": 16,
                 "num_channels": 1, "mb": 10, "packet": bw.hex(), "pcm": le([4, 12, 22, 35, 45, 45, 58, 93], 2)})

    # ---- K10: 24-bit CPE, bytesShifted 1, partial frame (6 of 8), general order 1 ----------------------------------
    bw = BitWriter()
    elem_header(bw, 1, partial=1, bs=1, num_samples=6)
    bw.put(1, 8).put(1, 8)
    chan_header(bw, 0, 2, 0, [3])
    chan_header(bw, 0, 0, 0, [])
    for b in [0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88, 0x99, 0xAA, 0xBB, 0xCC]:
        bw.put(b, 8)
    unary(bw, [-2, 1, -1, 3, 0, -2], 17)
    unary(bw, [1, -1, 2, 0, -3, 4], 17)
    bw.put(7, 3)
    pcm = "11FFFF22FEFF" "33FFFF440000" "55FFFF66FDFF" "770200880200" "99FEFFAA0100" "BB0100CCFDFF"
    kats.append({"name": "K10 24-bit shift merge, partial frame, order 1", "frame_length": 8, "frames": 6,
                 "bit_depth": 24, "num_channels": 2, "mb": 255, "packet": bw.hex(), "pcm": pcm})

    # ---- K11: FIL (short and extended count), DSE with alignment, then a 20-bit SCE --------------------------------
    bw = BitWriter()
    bw.put(6, 3).put(2, 4).put(0xBEEF, 16)                    # FIL, count 2
    bw.put(6, 3).put(15, 4).put(1, 8)                         # FIL, count 15 + 1 - 1 = 15
    for i in range(15):
        bw.put(0xA0 + i, 8)
    bw.put(4, 3).put(0, 4).put(1, 1).put(3, 8).align()        # DSE, align flag, 3 bytes
    bw.put(0xDEAD42, 24)
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 0, [])
    unary(bw, [1, -1, 2, -2, 3, -3], 20)
    bw.put(7, 3)
    kats.append({"name": "K11 FIL + DSE prefix, 20-bit SCE", "frame_length": 6, "bit_depth": 20, "num_channels": 1,
                 "mb": 255, "packet": bw.hex(), "pcm": "100000F0FFFF200000E0FFFF300000D0FFFF"})

    # ---- K12: 32-bit CPE, bytesShifted 2, negative mixRes ----------------------------------------------------------
    bw = BitWriter()
    elem_header(bw, 1, bs=2)
    bw.put(3, 8).put(-5, 8)
    chan_header(bw, 0, 0, 0, [])
    chan_header(bw, 0, 0, 0, [])
    for w in [0x1234, 0xABCD, 0x0001, 0xFFFF, 0x8000, 0x7FFF, 0x0F0F, 0xF0F0]:
        bw.put(w, 16)
    unary(bw, [3, -3, 4, -1], 17)
    unary(bw, [-2, 4, 1, -4], 17)
    bw.put(7, 3)
    pcm = "34120000CDAB0200" "01000400FFFF0000" "00800600FF7F0500" "0F0FF9FFF0F0FDFF"
    kats.append({"name": "K12 32-bit shift merge (2 bytes), negative mixRes", "frame_length": 4, "bit_depth": 32,
                 "num_channels": 2, "mb": 255, "packet": bw.hex(), "pcm": pcm})

    # ---- K13: 24-bit CPE escape element (chanBits > 16 path) -------------------------------------------------------
    bw = BitWriter()
    elem_header(bw, 1, escape=1)
    for hi, lo in [(0x1234, 0x56), (0xFFFF, 0xFF), (0x8000, 0x00), (0x7FFF, 0xFF), (0x0000, 0x01), (0x00FF, 0x00)]:
        bw.put(hi, 16).put(lo, 8)
    bw.put(7, 3)
    kats.append({"name": "K13 24-bit CPE escape", "frame_length": 3, "bit_depth": 24, "num_channels": 2, "mb": 255,
                 "packet": bw.hex(), "pcm": "563412FFFFFF000080FFFF7F01000000FF00"})
    return kats


def build3():
    """K14..K19 (round 3): the branches K1..K13 do not reach. Derivations: kat_derivation.md, second part."""
    kats = []

    # ---- K14: 16-bit SCE, order 5 (unpcBlock5, predictor.go:198-310): early stops and full walks, both signs -------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 1, 0, [2, -1, 1, 0, 1])
    unary(bw, [7, 2, -3, 1, 2, -1, 4, -5, 20, -30, 2], 16)
    bw.put(7, 3)
    kats.append({"name": "K14 SCE order 5 (unpcBlock5)", "frame_length": 11, "bit_depth": 16, "num_channels": 1, "mb": 255,
                 "packet": bw.hex(), "pcm": le([7, 9, 6, 7, 9, 8, 12, 5, 31, -3, 41], 2)})

    # ---- K15: the mean clamp for n > 0xffff (golomb.go:216-218), then k = 7 codes; V with two zero-length runs -----
    bw = BitWriter()
    elem_header(bw, 1)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 4, [])
    chan_header(bw, 0, 0, 4, [])
    bw.raw("111111111").put(0x10001, 17)   # U0: escape code, n = 65537 -> del = -32769; mean clamped to 0xffff
    bw.raw("10 0000101")                    # U1: k = 7, m = 127: pre 1, v 5 -> n = 131 -> del = -66
    bw.raw("0 000000")                      # U2: k = 7: pre 0, v < 2 -> n = 0, six bits behind the prefix
    bw.raw("110 00 0 00 10")                # V: +1, run of 0, -1 (zmode), run of 0, +1 (zmode)
    bw.put(7, 3)
    kats.append({"name": "K15 mean clamp n > 0xffff, k = 7, zero-length runs", "frame_length": 3, "bit_depth": 16,
                 "num_channels": 2, "mb": 10, "packet": bw.hex(), "pcm": "FF7F0100BEFFFFFF00000100"})

    # ---- K16: 32-bit SCE without shift bytes: chanBits 32, getStreamBits' fifth byte (golomb.go:90-99) -------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 0, [])
    bw.raw("110")                            # s0: k = 1, n = 2 -> +1
    bw.raw("111111111").put(0x89ABCDEE, 32)  # s1: escape code read at bit offset 19 of the stream: 32 + 3 > 32
    bw.raw("0 0000010")                      # s2: the clamp left mean = 0xffff: k = 7, v = 2 -> n = 1 -> -1
    bw.put(7, 3)
    kats.append({"name": "K16 32-bit literal across five bytes", "frame_length": 3, "bit_depth": 32, "num_channels": 1,
                 "mb": 255, "packet": bw.hex(), "pcm": "01000000F7E6D544FFFFFFFF"})

    # ---- K17: zero run whose length is the 16-bit literal of dynGet (golomb.go:121-129) ----------------------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 4, [])
    bw.raw("0")                              # s0: n = 0; mean 10 -> zero run, k = 4
    bw.raw("111111111").put(17, 16)          # run of 17 as a literal
    bw.raw("110")                            # s18: zmode: n = 2 -> nd = 3 -> -2; mean 120 -> zero run, k = 3
    bw.raw("0 00")                           # run of 0
    bw.raw("0")                              # s19: zmode: nd = 1 -> -1
    bw.put(7, 3)
    kats.append({"name": "K17 zero run with a 16-bit literal length", "frame_length": 20, "bit_depth": 16,
                 "num_channels": 1, "mb": 10, "packet": bw.hex(), "pcm": le([0] * 18 + [-2, -1], 2)})

    # ---- K18: 8 channels: SCE CPE CPE CPE LFE -> L R C LFE Ls Rs Lc Rc (decoder.go:63) -----------------------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 0, [])
    unary(bw, [10, 11], 16)                                     # C
    for u, v in (([20, 21], [30, 31]), ([40, 41], [50, 51]), ([60, 61], [-70, -71])):   # Lc Rc, L R, Ls Rs
        elem_header(bw, 1)
        bw.put(0, 8).put(0, 8)
        chan_header(bw, 0, 0, 0, [])
        chan_header(bw, 0, 0, 0, [])
        unary(bw, u, 17)
        unary(bw, v, 17)
    elem_header(bw, 3)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 0, [])
    unary(bw, [-1, -2], 16)                                     # LFE
    bw.put(7, 3)
    frames = [40, 50, 10, -1, 60, -70, 20, 30, 41, 51, 11, -2, 61, -71, 21, 31]
    kats.append({"name": "K18 8-channel layout", "frame_length": 2, "bit_depth": 16, "num_channels": 8, "mb": 255,
                 "packet": bw.hex(), "pcm": le(frames, 2)})

    # ---- K19: FIL of 0 bytes, DSE with the 255 + n count and alignment (decoder.go:555-574), then an SCE -----------
    bw = BitWriter()
    bw.put(6, 3).put(0, 4)                                      # FIL, count 0: the DSE starts at bit 7
    bw.put(4, 3).put(5, 4).put(1, 1).put(255, 8).put(2, 8)      # DSE: align flag, count 255 + 2 = 257
    bw.align()                                                  # bit 31 -> 32
    for i in range(257):
        bw.put((i * 7 + 1) & 0xFF, 8)
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 0, [])
    unary(bw, [3, -4], 16)
    bw.put(7, 3)
    kats.append({"name": "K19 DSE with extended count", "frame_length": 2, "bit_depth": 16, "num_channels": 1, "mb": 255,
                 "packet": bw.hex(), "pcm": "0300FCFF"})
    return kats


def build4():
    """K20..K23 (round 4): cookie bytes PB and MB (K23: KB too) other than 40 / 10 (/ 14) (config.go:72-73), the mean trajectories worked on
    paper from golomb.go:172-246 with pb = PB * pbFactor / 4 (decoder.go:296-299). Derivations: kat_derivation.md, third
    part. All three are 16- / 20-bit SCEs with numActive 0 (the samples are the residuals, predictor.go:53-60) and
    FrameLength 40, so that the GPU library sorts them under a regular key (alac_regular.h: classify_regular)."""
    kats = []

    # ---- K20: PB 20, pbFactor 6 -> pb 30; MB 0: first mean 0, a zero run at once, k = 1 -> 2, an escape code, then
    #      the fixed point mean = 3584 = 512 * 7 with n = 7 for ever ---------------------------------------------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 6, [])
    bw.raw("1110")                           # s0: k = 1, n = 3 -> -2; mean 90 -> zero run, k32 = 2, mz = 3
    bw.raw("0 10")                           # run of 1 (s1 = 0)
    bw.raw("111111110")                      # s2: zmode: n = 8, nd = 9 -> -5; mean 270
    bw.raw("11111110")                       # s3: n = 7 -> -4; mean 465
    bw.raw("111111110")                      # s4: n = 8 -> +4; mean 678
    bw.raw("110 11")                         # s5: k = 2: pre 2, v 3 -> n = 8 -> +4; mean 879
    bw.raw("10 0")                           # s6: k = 2: pre 1, v < 2 -> n = 3 -> -2; mean 918
    bw.raw("0 10")                           # s7: k = 2: pre 0, v 2 -> n = 1 -> -1; mean 895
    bw.raw("0 0")                            # s8: n = 0; mean 843
    bw.raw("111111111").put(93, 16)          # s9: escape code, n = 93 -> -47; mean 3584
    for _ in range(30):
        bw.raw("10 00")                      # s10..s39: k = 3, m = 7: pre 1, v < 2 -> n = 7 -> -4; mean stays 3584
    bw.put(7, 3)
    kats.append({"name": "K20 PB 20 (pb 30), MB 0", "frame_length": 40, "bit_depth": 16, "num_channels": 1, "pb": 20, "mb": 0,
                 "packet": bw.hex(), "pcm": le([-2, 0, -5, -4, 4, 4, -2, -1, 0, -47] + [-4] * 30, 2)})

    # ---- K21: PB 73, pbFactor 7 -> pb 127, the largest the lean Golomb step of the kernels takes; MB 255; the mean climbs
    #      to 21 005 824 = 512 * 41 027 and stays there -----------------------------------------------------------------
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 7, [])
    bw.raw("111111111").put(0xFFFF, 16)      # s0: k = 1, escape code, n = 65535 -> -32768; mean 8 323 137
    bw.raw("11111110").put(8191, 13)         # s1: k = 13: pre 7, v 8191 -> n = 65527 -> -32764; mean 14 580 538
    bw.raw("1110").put(16383, 14)            # s2: k = 14: pre 3, v 16383 -> n = 65531 -> -32766; mean 19 286 319
    bw.raw("1110").put(2060, 14)             # s3: pre 3, v 2060 -> n = 51208 -> +25604; mean 21 005 824
    for _ in range(36):
        bw.raw("110").put(8262, 14)          # s4..s39: pre 2, v 8262 -> n = 41027 -> -20514; mean stays
    bw.put(7, 3)
    kats.append({"name": "K21 PB 73 (pb 127), MB 255", "frame_length": 40, "bit_depth": 16, "num_channels": 1, "pb": 73,
                 "mb": 255, "packet": bw.hex(), "pcm": le([-32768, -32764, -32766, 25604] + [-20514] * 36, 2)})

    # ---- K22: PB 255, pbFactor 7 -> pb 446: pb * mean wraps in uint32 (golomb.go:215); MB 1; 20-bit, so that an escape code
    #      can carry n = 65536 > 0xffff and the clamp (golomb.go:216-218) brings the mean back; then down into a zero run ---
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 7, [])
    bw.raw("111111111").put(0xFFFF, 20)      # s0: k = 1, escape code, n = 65535 -> -32768; mean 29 228 611
    bw.raw("0").put(2, 14)                   # s1: k = 14: pre 0, v 2 -> n = 1 -> -1; 446 * mean wraps; mean 28 934 021
    bw.raw("10").put(16383, 14)              # s2: pre 1, v 16383 -> n = 32765 -> -16383; wraps; mean 43 508 791
    bw.raw("0").put(0, 13)                   # s3: pre 0, v < 2 -> n = 0; wraps; mean 39 162 988
    bw.raw("111111111").put(0x10000, 20)     # s4: escape code, n = 65536 -> +32768; mean clamped to 65535
    bw.raw("0").put(0, 6)                    # s5: k = 7: n = 0; mean 8448
    bw.raw("0").put(0, 3)                    # s6: k = 4: n = 0; mean 1089
    bw.raw("0 0")                            # s7: k = 2: n = 0; mean 141
    bw.raw("0")                              # s8: k = 1: n = 0; mean 19 -> zero run, k32 = 3, mz = 7
    bw.raw("11110 100")                      # run of 4 * 7 + 4 - 1 = 31: s9..s39
    bw.put(7, 3)
    vals = [-32768, -1, -16383, 0, 32768, 0, 0, 0, 0] + [0] * 31
    kats.append({"name": "K22 PB 255 (pb 446: uint32 wrap), MB 1, 20-bit", "frame_length": 40, "bit_depth": 20,
                 "num_channels": 1, "pb": 255, "mb": 1, "packet": bw.hex(), "pcm": le([v << 4 for v in vals], 3)})

    # ---- K23: the reference's 32-bit window has 32 - (bitPos & 7) stream bits (golomb.go:179-180): a code of 9 + 17 bits at
    #      bit offset 7 ends in the zero fill. KB 32 lets k reach 17 once the mean has passed 2^26 (pb 446: no contraction)
    bw = BitWriter()
    elem_header(bw, 0)
    bw.put(0, 8).put(0, 8)
    chan_header(bw, 0, 0, 7, [])
    bw.raw("111111111").put(0xFFFF, 20)      # s0 (bit 55): k = 1, escape code, n = 65535 -> -32768; mean 29 228 611
    bw.raw("110").put(2, 15)                 # s1 (bit 84): k = 15: pre 2, v 2 -> n = 65535; mean 58 162 185
    bw.raw("10").put(0, 15)                  # s2 (bit 102): k = 16: pre 1, v < 2 -> n = 65535; mean 87 057 728
    bw.raw("111111110").put(5, 17)           # s3 (bit 119, offset 7): k = 17: pre 8; the stream says v = 5, the window 4
    bw.raw("0").put(0, 6)                    # s4: k = 7: n = 0; mean 8448
    bw.raw("0").put(0, 3)                    # s5: k = 4: n = 0; mean 1089
    bw.raw("0 0")                            # s6: k = 2: n = 0; mean 141
    bw.raw("0")                              # s7: k = 1: n = 0; mean 19 -> zero run, k32 = 3, mz = 7
    bw.raw("11110 101")                      # run of 4 * 7 + 5 - 1 = 32: s8..s39
    bw.put(7, 3)
    vals = [-32768, -32768, -32768, -524286, 0, 0, 0, 0] + [0] * 32
    kats.append({"name": "K23 KB 32, PB 255: a code that ends in the window's zero fill", "frame_length": 40, "bit_depth": 20,
                 "num_channels": 1, "pb": 255, "mb": 1, "kb": 32, "packet": bw.hex(), "pcm": le([v << 4 for v in vals], 3)})
    return kats


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = {"_source": "hand-built packets; expected PCM derived on paper from the reference source in "
                      "tests/golden/kat_derivation.md (no decoder was run to produce it)",
           "config_common": {"pb": 40, "kb": 14, "max_run": 255}, "vectors": build()}
    path = os.path.join(here, "kat2.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["vectors"]), "vectors")
    out["vectors"] = build3()
    path = os.path.join(here, "kat3.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["vectors"]), "vectors")
    out["config_common"] = {"kb": 14, "max_run": 255}  # PB and MB per vector
    out["vectors"] = build4()
    path = os.path.join(here, "kat4.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["vectors"]), "vectors")


if __name__ == "__main__":
    main()
