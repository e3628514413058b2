/*
 * alac_lane.h — the per-packet ALAC decode state machine executed by ONE LANE of a wavefront.
 *
 * Design: lane-per-packet. A 64-wide wavefront decodes 64 independent packets in
 * lock step. Every lane produces exactly one residual per loop iteration (a zero run is a per-lane
 * countdown, not a burst), so all lanes of a wave sit at the same sample index i: the scratch row
 * scr[i][lane] is one coalesced 256-B access and control flow only diverges on rare paths (escape
 * codes, zero-run starts, the general predictor). Nothing here is a port: the reference decodes one
 * packet at a time with whole-block passes (Golomb block -> predictor block -> writer); this fuses
 * them per sample and keeps all state in registers.
 *
 * Bit-exactness contract: identical PCM bytes, frame count and status word to the reference
 * (mycophonic/saprobe-alac) for every input, including Go's shift/wrap semantics. Reference lines
 * are cited at each step (paths relative to the reference tree).
 *
 * The file is plain C++ with ALAC_DEV in front of every function: alacgpu.hip compiles it for
 * gfx950 (ALAC_DEV = __device__ __forceinline__); tests/host_sim compiles the same text with g++
 * to check the LOGIC against the oracle without a GPU. It is not a CPU decode path of the product:
 * libalacgpu.so contains no host decoder.
 */
#ifndef ALAC_LANE_H
#define ALAC_LANE_H

#include <stdint.h>

#include "../../include/alacgpu.h"

#ifndef ALAC_DEV
#error "define ALAC_DEV before including alac_lane.h"
#endif

namespace alac {

struct DevCfg {
    uint32_t frame_length;
    uint32_t bit_depth;
    uint32_t num_channels;
    uint32_t pb, mb, kb;
    uint32_t bps;        /* BytesPerSample, internal/alac/format.go:23-34 */
    uint32_t fast16s;    /* 1: 16-bit stereo, 16-byte aligned output -> batched 16-B stores */
};

/* channelLayoutOffsets (decoder.go:55-64) packed 4 bits per entry, entry k at bits 4k */
ALAC_DEV uint32_t layout_offset(uint32_t num_chan, uint32_t chan_idx) {
    const uint32_t tbl[8] = {0x0u,        0x10u,       0x102u,      0x3102u,
                             0x43102u,    0x354102u,   0x3654102u,  0x35410762u};
    return (tbl[num_chan - 1] >> (4 * chan_idx)) & 0xfu;
}

/* ---- Go shift semantics (SURVEY.md §8a trap 1): counts >= 32 give 0 / sign fill ------------- */
ALAC_DEV uint32_t go_shl(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x << n; }
ALAC_DEV uint32_t go_shr(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x >> n; }
ALAC_DEV int32_t go_sar(int32_t x, uint32_t n) { return x >> (n >= 32 ? 31u : n); }
/* (x << chanShift) >> chanShift, predictor.go:68,78,130 */
ALAC_DEV int32_t sext_cs(int32_t x, uint32_t cs) {
    return cs >= 32 ? 0 : (int32_t)((uint32_t)x << cs) >> cs;
}
/* signOfInt, predictor.go:35-39 */
ALAC_DEV int32_t sign_of(int32_t v) { return (int32_t)((uint32_t)(-v) >> 31) | (v >> 31); }
ALAC_DEV uint32_t clz32(uint32_t x) { return x ? (uint32_t)__builtin_clz(x) : 32u; }
ALAC_DEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
ALAC_DEV uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

/* ---- bit access ------------------------------------------------------------------------------
 * Stateless reads at an absolute bit position: 64-bit big-endian window whose MSB is stream bit
 * `pos` (>= 57 valid bits). The byte offset is clamped to size+8 so a corrupt position can never
 * leave the packet's zero pad (ALACGPU_PACKET_PAD = 16); every consumer of such a position raises
 * a status before the data could matter. */
struct Bits {
    const uint8_t* p;
    uint32_t size;

    ALAC_DEV uint64_t window(uint32_t pos) const {
        uint32_t b = umin(pos >> 3, size + 8u);
        uint64_t raw;
        __builtin_memcpy(&raw, p + b, 8);
        return __builtin_bswap64(raw) << (pos & 7u);
    }
    /* n bits (0..32) at pos: BitBuffer.Read / ReadSmall / ReadOne all reduce to this
     * (bitbuffer.go:55-96) */
    ALAC_DEV uint32_t get(uint32_t pos, uint32_t n) const {
        return n == 0 ? 0u : (uint32_t)(window(pos) >> (64u - n));
    }
    /* where the Go code panics on a slice bound (fresh decoder: len(Buf) = size+4):
     * Read: Buf[Pos:Pos+3] (bitbuffer.go:58); ReadSmall: Buf[Pos:Pos+2] (:75) */
    ALAC_DEV bool read_panics(uint32_t pos) const { return (pos >> 3) > size + 1u; }
    ALAC_DEV bool read_small_panics(uint32_t pos) const { return (pos >> 3) > size + 2u; }
    ALAC_DEV bool past_end(uint32_t pos) const { return (pos >> 3) >= size; } /* bitbuffer.go:115 */
};

/* Advance (bitbuffer.go:99-103): BitIdx is uint32 and wraps; positions far past the packet are
 * clamped (every later use of them errors the same way wherever they are). */
ALAC_DEV uint32_t advance(uint32_t pos, uint32_t nbits) {
    uint64_t np = (uint64_t)(pos & ~7u) + (uint64_t)(uint32_t)((pos & 7u) + nbits);
    return np > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)np;
}

constexpr int32_t ST_OVERRUN = ALACGPU_ERR_BITSTREAM_OVERRUN;
constexpr int32_t ST_SAMPLE_OVERRUN = ALACGPU_ERR_SAMPLE_OVERRUN;
constexpr int32_t ST_HEADER = ALACGPU_ERR_INVALID_HEADER;
constexpr int32_t ST_SHIFT = ALACGPU_ERR_INVALID_SHIFT;
constexpr int32_t ST_UNSUPPORTED = ALACGPU_ERR_UNSUPPORTED_ELEMENT;
constexpr int32_t ST_MALFORMED = ALACGPU_ERR_MALFORMED;

/* little-endian store of the low `bps` bytes (matrix.go:43-48 etc.) */
ALAC_DEV void store_le(uint8_t* dst, int32_t v, uint32_t bps) {
    dst[0] = (uint8_t)v;
    dst[1] = (uint8_t)(v >> 8);
    if (bps > 2) dst[2] = (uint8_t)(v >> 16);
    if (bps > 3) dst[3] = (uint8_t)(v >> 24);
}

/*
 * Decode one packet. `scr` is this lane's column of the wave's scratch tile: element i lives at
 * scr[i * SCR_STRIDE] (SCR_STRIDE = 64 on the GPU: row i of the wave is contiguous).
 * Returns the status word; *frames_out = numSamples of the last element (decoder.go:206).
 */
template <uint32_t SCR_STRIDE>
ALAC_DEV int32_t decode_lane(const DevCfg& cfg, const uint8_t* pkt, uint32_t size, uint8_t* out, int32_t* scr,
                             uint32_t* frames_out) {
    const Bits bits{pkt, size};
    const uint32_t num_chan = cfg.num_channels;
    const uint32_t bps = cfg.bps;
    const uint32_t frame_stride = num_chan * bps;
    const uint32_t depth = cfg.bit_depth;
    const uint32_t wb = go_shl(1u, cfg.kb) - 1u; /* SetAGParams golomb.go:60 */

    uint32_t pos = 0;
    uint32_t num_samples = cfg.frame_length; /* decoder.go:136 */
    uint32_t chan_idx = 0;
    uint32_t written[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* frames written per output channel slot */
    int32_t st = 0;

    for (;;) {
        /* ---- element dispatch, decoder.go:142-203 ------------------------------------------ */
        if (bits.past_end(pos)) {
            st = ALACGPU_STATUS(ST_OVERRUN, 0, 0);
            break;
        }
        const uint32_t tag = bits.get(pos, 3);
        pos += 3;
        if (tag == 2 || tag == 5) { /* CCE / PCE, decoder.go:179-180 */
            st = ALACGPU_STATUS(ST_UNSUPPORTED, 0, 0);
            break;
        }
        if (tag == 4) { /* skipDSE, decoder.go:555-574 */
            uint32_t align = bits.get(pos + 4, 1);
            uint32_t count = bits.get(pos + 5, 8);
            pos += 13;
            if (count == 255) {
                count += bits.get(pos, 8);
                pos += 8;
            }
            if (align && (pos & 7u)) pos = advance(pos, 8u - (pos & 7u));
            pos = advance(pos, count * 8u);
            if (bits.past_end(pos)) {
                st = ALACGPU_STATUS(ST_OVERRUN, ALACGPU_CTX_DSE, 0);
                break;
            }
            continue;
        }
        if (tag == 6) { /* skipFIL, decoder.go:538-552 */
            uint32_t count = bits.get(pos, 4);
            pos += 4;
            if (count == 15) {
                count += bits.get(pos, 8) - 1u;
                pos += 8;
            }
            pos = advance(pos, (count & 0xffffu) * 8u);
            if (bits.past_end(pos)) {
                st = ALACGPU_STATUS(ST_OVERRUN, ALACGPU_CTX_FIL, 0);
                break;
            }
            continue;
        }
        if (tag == 7) break; /* END, decoder.go:192-195 */

        const bool cpe = tag == 1;
        if (cpe && chan_idx + 2 > num_chan) break; /* decoder.go:163-165 */
        const uint32_t ctx = cpe ? ALACGPU_CTX_CPE : ALACGPU_CTX_SCE;
        const uint32_t nch_e = cpe ? 2u : 1u;
        const uint32_t out_chan = layout_offset(num_chan, chan_idx);
        if (out_chan + nch_e > num_chan) {
            /* a pair that does not fit the frame: the reference writes outside the frame (and
             * panics on a full frame); oracle and kernel both report it as malformed */
            st = ALACGPU_STATUS(ST_MALFORMED, ctx, 0);
            break;
        }

        /* ---- element header, decoder.go:213-235 / 351-376 ------------------------------------ */
        /* ReadSmall(4) instance tag, Read(12) unused, Read(4) header nibble */
        if (bits.read_small_panics(pos) || bits.read_panics(pos + 4)) {
            st = ALACGPU_STATUS(ST_MALFORMED, ctx, 0);
            break;
        }
        if (bits.get(pos + 4, 12) != 0) {
            st = ALACGPU_STATUS(ST_HEADER, ctx, 0);
            break;
        }
        if (bits.read_panics(pos + 16)) {
            st = ALACGPU_STATUS(ST_MALFORMED, ctx, 0);
            break;
        }
        const uint32_t hdr = bits.get(pos + 16, 4);
        pos += 20;
        const uint32_t partial = hdr >> 3;
        uint32_t bytes_shifted = (hdr >> 1) & 3u;
        if (bytes_shifted == 3) {
            st = ALACGPU_STATUS(ST_SHIFT, ctx, 0);
            break;
        }
        const bool escape = (hdr & 1u) != 0;
        uint32_t chan_bits = depth - bytes_shifted * 8u + (cpe ? 1u : 0u);
        uint32_t ns = num_samples;
        bool bad = false;
        if (partial) {
            bad = bits.read_panics(pos) || bits.read_panics(pos + 16);
            ns = bits.get(pos, 32);
            pos += 32;
        }

        int32_t mix_bits = 0, mix_res = 0;
        uint32_t hdr_pos = 0;   /* first per-channel header (compressed) */
        uint32_t shift_pos = 0; /* start of the shift block */
        uint32_t data_pos = 0;  /* escape: first raw sample */
        if (!escape) {
            /* decodeSCECompressed / decodeCPECompressed field walk, decoder.go:272-293 / 421-457 */
            mix_bits = (int32_t)bits.get(pos, 8);
            mix_res = (int32_t)(int8_t)bits.get(pos + 8, 8);
            hdr_pos = pos + 16;
            uint32_t q = hdr_pos;
            uint32_t last_read = pos + 8;
            for (uint32_t c = 0; c < nch_e; ++c) {
                uint32_t num = bits.get(q + 11, 5);
                last_read = q + 16 + (num ? (num - 1u) * 16u : 0u) - (num ? 0u : 8u);
                q += 16u + 16u * num;
            }
            /* header reads are sequential and only panic: test the last one */
            bad = bad || bits.read_panics(last_read);
            shift_pos = q;
            pos = q;
            if (bytes_shifted != 0) pos = advance(pos, bytes_shifted * 8u * nch_e * ns);
            /* DynDecomp entry: input := Buf[Pos:] (golomb.go:149), predCoefs[:numSamples] (:155) */
            bad = bad || (pos >> 3) > size + 4u || ns > cfg.frame_length;
        } else {
            if (cpe) chan_bits = depth; /* decoder.go:388 */
            data_pos = pos;
            /* mixU[:numSamples:numSamples] (decoder.go:328,509) and the Read()s of the raw samples:
             * positions grow monotonically, so only the last read can be the first to panic */
            bad = bad || ns > cfg.frame_length;
            if (!bad && ns != 0) {
                uint32_t total = nch_e * ns * chan_bits;
                uint32_t last_w = chan_bits > 16 ? chan_bits - 16u : chan_bits;
                bad = bits.read_panics(pos + total - last_w);
            }
            pos = advance(pos, nch_e * ns * chan_bits);
        }
        if (bad) {
            st = ALACGPU_STATUS(ST_MALFORMED, ctx, 0);
            break;
        }
        const bool use_shift = !escape && bytes_shifted != 0 && (depth == 24 || depth == 32);
        const uint32_t shift_bits = bytes_shifted * 8u;
        const uint32_t chan_shift = 32u - chan_bits; /* wraps for chanBits 33, predictor.go:46 */
        const uint32_t mix_sh = (uint32_t)mix_bits > 31u ? 31u : (uint32_t)mix_bits;

        /* ---- channels of the element: U then V ------------------------------------------------ */
        uint8_t* const obase = out + out_chan * bps;
        int32_t err = 0;
        uint32_t err_chan = 0;
        for (uint32_t c = 0; c < nch_e; ++c) {
            const bool last_chan = c + 1 == nch_e;
            /* per-channel header, decoder.go:275-286 */
            uint32_t mode = 0, den_shift = 0, na = 0, pb_local = 0;
            int32_t coef[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int32_t coef_hi[32]; /* general predictor only (orders 9..30): spills to scratch */
            if (!escape) {
                const uint32_t h = bits.get(hdr_pos, 16);
                mode = h >> 12;
                den_shift = (h >> 8) & 0xfu;
                pb_local = (cfg.pb * ((h >> 5) & 7u)) / 4u; /* decoder.go:299 */
                na = h & 0x1fu;
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j)
                    if (j < na) coef[j] = (int32_t)(int16_t)bits.get(hdr_pos + 16u + 16u * j, 16);
                if (na > 8 && na != 31)
                    for (uint32_t j = 0; j < na; ++j)
                        coef_hi[j] = (int32_t)(int16_t)bits.get(hdr_pos + 16u + 16u * j, 16);
                hdr_pos += 16u + 16u * na;
                /* DynDecomp entry again for V: Buf[Pos:] can only have grown legally */
                if ((pos >> 3) > size + 4u || (ns != 0 && (pos >> 3) > size)) {
                    err = ST_MALFORMED; /* Buf[Pos:] out of range, or maxPos wrapped: first read32bit panics */
                    break;
                }
            }
            const bool wrap16 = !(na == 4 || na == 5 || na == 6 || na == 8);
            const int32_t den_half = den_shift ? (int32_t)(1u << (den_shift - 1u)) : 0;
            const uint32_t max_pos = size * 8u;

            uint32_t mean = cfg.mb, zmode = 0, zrem = 0;
            int32_t hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* hist[j] = out[i-1-j] */
            int32_t ring[32];                               /* general predictor history */
            int32_t dprev = 0;                              /* delta pre-pass state (mode != 0) */

            for (uint32_t i = 0; i < ns; ++i) {
                int32_t o;
                if (!escape) {
                    /* ---- one residual: DynDecomp, golomb.go:167-247 ------------------------------ */
                    int32_t del;
                    if (zrem != 0) {
                        del = 0; /* inside a zero run (golomb.go:236-239) */
                        --zrem;
                    } else {
                        if (pos >= max_pos) {
                            err = ST_OVERRUN; /* golomb.go:168-170 */
                            break;
                        }
                        uint32_t m = mean >> 9;
                        uint32_t k = umin(31u - clz32(m + 3u), cfg.kb);
                        m = go_shl(1u, k) - 1u;
                        const uint64_t w = bits.window(pos);
                        uint32_t n = clz32(~(uint32_t)(w >> 32));
                        if (n >= 9) {
                            /* escape code: getStreamBits(bitPos+9, maxSize), golomb.go:184-186,86-108 */
                            const uint32_t gpos = pos + 9u;
                            const uint32_t gb = gpos & 7u;
                            const bool five = chan_bits + gb > 32u;
                            if ((gpos >> 3) > size || (five && (gpos >> 3) >= size)) {
                                err = ST_MALFORMED; /* read32bit / input[byteOffset+4] out of range */
                                break;
                            }
                            const uint64_t w2 = w << 9;
                            if (chan_bits == 0) n = 0;
                            else if (chan_bits <= 32) n = (uint32_t)(w2 >> (64u - chan_bits));
                            else n = (uint32_t)(w2 >> 31) & ((2u << gb) - 1u); /* numBits 33: only byte 5 survives */
                            pos += 9u + chan_bits;
                        } else {
                            pos += n + 1u;
                            if (k != 1) {
                                const uint32_t v = k == 0 ? 0u : (uint32_t)((w << (n + 1u)) >> (64u - k));
                                if (v >= 2) {
                                    n = n * m + v - 1u;
                                    pos += k;
                                } else {
                                    n *= m;
                                    pos += k - 1u;
                                }
                            }
                        }
                        const uint32_t nd = n + zmode;
                        del = (int32_t)((nd + 1u) >> 1) * (-(int32_t)(nd & 1u) | 1); /* golomb.go:206-209 */
                        mean = pb_local * nd + mean - ((pb_local * mean) >> 9);      /* golomb.go:215 */
                        if (n > 0xffffu) mean = 0xffffu;
                        zmode = 0;
                        if ((mean << 2) < 512u && i + 1u < ns) { /* golomb.go:223 */
                            zmode = 1;
                            int32_t k32 = (int32_t)clz32(mean) - 24 + (int32_t)((mean + 16u) >> 6);
                            if (k32 < 0) k32 = 0;
                            const uint32_t kz = (uint32_t)k32;
                            const uint32_t mz = (go_shl(1u, kz) - 1u) & wb;
                            if ((pos >> 3) > size) { /* dynGet's read32bit, golomb.go:115 */
                                err = ST_MALFORMED;
                                break;
                            }
                            const uint64_t wz = bits.window(pos);
                            uint32_t pre = clz32(~(uint32_t)(wz >> 32));
                            uint32_t run;
                            if (pre >= 9) {
                                run = (uint32_t)((wz << 9) >> 48);
                                pos += 25u;
                            } else {
                                pos += pre + 1u;
                                const uint32_t val = kz == 0 ? 0u : (uint32_t)((wz << (pre + 1u)) >> (64u - kz));
                                pos += kz;
                                if (val < 2) {
                                    run = pre * mz;
                                    pos -= 1u;
                                } else {
                                    run = pre * mz + val - 1u;
                                }
                            }
                            if ((uint64_t)i + 1u + run > ns) {
                                err = ST_SAMPLE_OVERRUN; /* golomb.go:232-234 */
                                break;
                            }
                            zrem = run;
                            if (run >= 65535u) zmode = 0;
                            mean = 0;
                        }
                    }
                    /* ---- delta pre-pass when mode != 0 (decoder.go:307-309: numActive 31, denShift 0) */
                    if (mode != 0) {
                        dprev = i == 0 ? del : sext_cs(del + dprev, chan_shift);
                        del = dprev;
                    }
                    /* ---- one predictor step: UnpcBlock, predictor.go:45-94 ---------------------------- */
                    const int32_t prev = hist[0];
                    if (i == 0 || na == 0) {
                        o = del; /* out[0] = pc1[0]; numActive 0 copies */
                    } else if (na == 31 || i <= na) {
                        o = sext_cs(del + prev, chan_shift); /* delta mode / warm-up, predictor.go:63-79 */
                    } else if (na <= 8) {
                        /* unpcBlock4/5/6/8 and the general form for 1,2,3,7 (predictor.go:99-684) */
                        int32_t top = hist[1];
#pragma unroll
                        for (uint32_t j = 2; j <= 8; ++j) top = na == j ? hist[j] : top;
                        int32_t d[8];
                        int32_t acc = den_half;
#pragma unroll
                        for (uint32_t j = 0; j < 8; ++j) {
                            d[j] = top - hist[j];
                            acc -= coef[j] * d[j];
                        }
                        o = sext_cs(del + top + (acc >> den_shift), chan_shift);
                        const int32_t sg = sign_of(del);
                        if (sg != 0) {
                            int32_t del0 = del;
                            bool go = true;
#pragma unroll
                            for (int32_t j = 7; j >= 0; --j) {
                                const bool act = go && (uint32_t)j < na;
                                const int32_t sgn = sg > 0 ? sign_of(d[j]) : -sign_of(d[j]);
                                int32_t cj = coef[j] - sgn;
                                if (wrap16) cj = (int32_t)(int16_t)cj; /* predictor.go:664,675 */
                                coef[j] = act ? cj : coef[j];
                                del0 -= act ? (int32_t)(na - (uint32_t)j) * ((sgn * d[j]) >> den_shift) : 0;
                                if (act && (sg > 0 ? del0 <= 0 : del0 >= 0)) go = false;
                            }
                        }
                    } else {
                        /* unpcBlockGeneral, orders 9..30 (predictor.go:623-684); rare */
                        const int32_t top = ring[(i - 1u - na) & 31u];
                        int32_t sum1 = 0;
                        for (uint32_t j = 0; j < na; ++j) sum1 += coef_hi[j] * (ring[(i - 1u - j) & 31u] - top);
                        o = sext_cs(del + top + ((sum1 + den_half) >> den_shift), chan_shift);
                        const int32_t sg = sign_of(del);
                        if (sg != 0) {
                            int32_t del0 = del;
                            for (int32_t j = (int32_t)na - 1; j >= 0; --j) {
                                const int32_t dd = top - ring[(i - 1u - (uint32_t)j) & 31u];
                                const int32_t sgn = sg > 0 ? sign_of(dd) : -sign_of(dd);
                                coef_hi[j] = (int32_t)(int16_t)(coef_hi[j] - sgn);
                                del0 -= (int32_t)(na - (uint32_t)j) * ((sgn * dd) >> den_shift);
                                if (sg > 0 ? del0 <= 0 : del0 >= 0) break;
                            }
                        }
                    }
#pragma unroll
                    for (uint32_t j = 8; j >= 1; --j) hist[j] = hist[j - 1];
                    hist[0] = o;
                    if (na > 8) ring[i & 31u] = o;
                } else {
                    /* decodeSCEEscape / decodeCPEEscape, decoder.go:326-345 / 507-535 */
                    o = sext_cs((int32_t)bits.get(data_pos + (i * nch_e + c) * chan_bits, chan_bits), chan_shift);
                }

                /* ---- hand-off / unmix / PCM store ---------------------------------------------------- */
                if (!last_chan) {
                    scr[(uint64_t)i * SCR_STRIDE] = o; /* U waits for V */
                } else {
                    uint8_t* dst = obase + (uint64_t)i * frame_stride;
                    if (cpe) {
                        const int32_t u = scr[(uint64_t)i * SCR_STRIDE];
                        const int32_t v = o;
                        int32_t l, r;
                        if (mix_res != 0) { /* matrix.go:40-41 */
                            l = u + v - ((mix_res * v) >> mix_sh);
                            r = l - v;
                        } else {
                            l = u;
                            r = v;
                        }
                        if (depth == 20) { /* matrix.go:77-78 */
                            l = (int32_t)((uint32_t)l << 4);
                            r = (int32_t)((uint32_t)r << 4);
                        }
                        if (use_shift) { /* matrix.go:129-132 */
                            const uint32_t sp = shift_pos + i * 2u * shift_bits;
                            l = (int32_t)((uint32_t)l << shift_bits) | (int32_t)bits.get(sp, shift_bits);
                            r = (int32_t)((uint32_t)r << shift_bits) | (int32_t)bits.get(sp + shift_bits, shift_bits);
                        }
                        if (cfg.fast16s) {
                            *(uint32_t*)dst = ((uint32_t)l & 0xffffu) | ((uint32_t)r << 16);
                        } else {
                            store_le(dst, l, bps);
                            store_le(dst + bps, r, bps);
                        }
                    } else {
                        int32_t val = o;
                        if (depth == 20) val = (int32_t)((uint32_t)val << 4);
                        if (use_shift) /* matrix.go:266-268 */
                            val = (int32_t)((uint32_t)val << shift_bits) |
                                  (int32_t)bits.get(shift_pos + i * shift_bits, shift_bits);
                        store_le(dst, val, bps);
                    }
                }
            }
            if (err) {
                err_chan = c;
                break;
            }
            /* UnpcBlock warm-up indexes 1..numActive of the frame-length buffers (predictor.go:76-79) */
            if (!escape && na != 0 && na != 31 && na >= cfg.frame_length) {
                err = ST_MALFORMED;
                break;
            }
        }
        if (err) {
            const uint32_t stage = (escape || err == ST_MALFORMED) ? (uint32_t)ALACGPU_STAGE_NONE
                                   : cpe ? (uint32_t)(err_chan == 0 ? ALACGPU_STAGE_ENTROPY_U : ALACGPU_STAGE_ENTROPY_V)
                                         : (uint32_t)ALACGPU_STAGE_ENTROPY;
            st = ALACGPU_STATUS(err, ctx, stage);
            break;
        }
        if (st) break;

#pragma unroll
        for (uint32_t s = 0; s < 8; ++s)
            if (s >= out_chan && s < out_chan + nch_e) written[s] = umax(written[s], ns);
        num_samples = ns;
        chan_idx += nch_e;
        if (chan_idx >= num_chan) break; /* decoder.go:200-202 */
    }

    if (st) {
        /* a Go panic carries no wrapping context: report the bare code */
        if (ALACGPU_STATUS_CODE(st) == ST_MALFORMED) st = ST_MALFORMED;
        *frames_out = 0;
        return st;
    }
    /* DecodePacket hands back output[:n] of a zeroed frame buffer (decoder.go:120,127): slots no
     * element wrote, or wrote for fewer frames than the last element, read as zero */
#pragma unroll
    for (uint32_t s = 0; s < 8; ++s) {
        if (s < num_chan && written[s] < num_samples) {
            for (uint32_t i = written[s]; i < num_samples; ++i) {
                uint8_t* dst = out + (uint64_t)i * frame_stride + s * bps;
                for (uint32_t b = 0; b < bps; ++b) dst[b] = 0;
            }
        }
    }
    *frames_out = num_samples;
    return 0;
}

} /* namespace alac */
#endif
