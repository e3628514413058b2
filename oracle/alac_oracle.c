/*
 * alac_oracle.c — CPU restatement of the reference ALAC packet decoder.
 *
 * TEST INFRASTRUCTURE ONLY. This file is the parity oracle and the CPU baseline
 * ("port") for bench.py. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (saprobe-alac_amd/) never does.
 *
 * It restates, function by function, the pure-Go reference mycophonic/saprobe-alac
 * (paths below are relative to the reference tree):
 *     internal/alac/bitbuffer.go:25-123   -> bb_*            (a1)
 *     internal/alac/golomb.go:28-253      -> dyn_decomp etc. (a2-a5)
 *     internal/alac/predictor.go:35-684   -> unpc_block*     (a6-a9)
 *     internal/alac/matrix.go:30-301      -> write_stereo / write_mono (a10-a11)
 *     internal/alac/format.go:23-34       -> bytes_per_sample (a12)
 *     decoder.go:55-76,117-574            -> decode_packet_into etc. (a13-a17)
 * Go integer semantics are kept explicitly: shifts by >= 32 (go_shl/go_shr/go_sar),
 * wrapping int32/uint32 arithmetic (compile with -fwrapv), uint16 truncation in
 * ReadSmall, int16 wrap in the general predictor only.
 *
 * Where the Go code would PANIC (slice bounds), this restatement longjmps out and
 * reports ALACGPU_ERR_MALFORMED: the reference has no defined result there. The model is a
 * FRESH PacketDecoder per packet (cap(bits.Buf) == len(packet)+4, bitbuffer.go:36-51).
 *
 * PARITY UNPINNED by the reference itself: it ships no golden vectors and cannot be built here (Go,
 * no toolchain), so nothing below has been compared with the Go binary. What pins this oracle
 * instead: the hand-derived known-answer packets K1-K4 of SURVEY.md §8(c) (tests/golden/kat.json),
 * K5-K13 (tests/golden/kat2.json) and K14-K19 (tests/golden/kat3.json), each decoded on paper from the
 * reference source in tests/golden/kat_derivation.md; a second, independent restatement in Python
 * (oracle/goref.py) that agrees with it on ~20 000 valid and corrupted packets
 * (tests/golden/crosscheck_goref.py); and the lossless round trip decode(encode(pcm)) == pcm, the
 * property the reference's conformance test asserts (tests/conformance_test.go:282-291).
 *
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg are the
 * only callers; the product (libalacgpu.so, the Python / C++ / Go hosts) never links or loads it.
 */
#include "alac_oracle.h"

#include <pthread.h>
#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

/* ---- Go shift semantics -------------------------------------------------------------- */
static inline uint32_t go_shl(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x << n; }
static inline uint32_t go_shr(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x >> n; }
static inline int32_t go_sar(int32_t x, uint32_t n) { return n >= 32 ? (x < 0 ? -1 : 0) : x >> n; }
static inline int32_t go_shl_i(int32_t x, uint32_t n) { return (int32_t)go_shl((uint32_t)x, n); }

/* ---- decoder state (decoder.go:79-87) ------------------------------------------------ */
typedef struct {
    const uint8_t* buf; /* padded copy: size + 4 zero bytes (bitbuffer.go:33,44-47) */
    int64_t pos;
    uint32_t bit_idx;
    int64_t size;
} bitbuf;

struct alac_oracle {
    alacgpu_config cfg;
    int32_t* mix_u;
    int32_t* mix_v;
    int32_t* predictor;
    uint16_t* shift_buf;
    uint8_t* padded;
    size_t padded_cap;
    jmp_buf panic_jmp;
    uint8_t* out;
    int64_t out_len;
};

#ifdef ORACLE_TRACE_PANIC
#include <stdio.h>
#include <execinfo.h>
#define GO_PANIC(d) do { void* bt_[8]; int n_ = backtrace(bt_, 8); fprintf(stderr, "go panic at oracle line %d\n", __LINE__); backtrace_symbols_fd(bt_, n_, 2); longjmp((d)->panic_jmp, 1); } while (0)
#else
#define GO_PANIC(d) longjmp((d)->panic_jmp, 1)
#endif

/* ---- BitBuffer (bitbuffer.go) ------------------------------------------------------- */
/* Read: bitbuffer.go:55-69. Buf[Pos:Pos+3:Pos+3] panics unless Pos+3 <= len(Buf). */
static uint32_t bb_read(alac_oracle* d, bitbuf* b, uint8_t num_bits) {
    if (b->pos < 0 || b->pos + 3 > b->size + 4) GO_PANIC(d);
    const uint8_t* w = b->buf + b->pos;
    uint32_t r = (uint32_t)w[0] << 16 | (uint32_t)w[1] << 8 | (uint32_t)w[2];
    r = (r << b->bit_idx) & 0x00FFFFFFu;
    r = go_shr(r, 24u - (uint32_t)num_bits);
    b->bit_idx += num_bits;
    b->pos += (int64_t)(b->bit_idx >> 3);
    b->bit_idx &= 7;
    return r;
}

/* ReadSmall: bitbuffer.go:73-85 (uint16 arithmetic). */
static uint8_t bb_read_small(alac_oracle* d, bitbuf* b, uint8_t num_bits) {
    if (b->pos < 0 || b->pos + 2 > b->size + 4) GO_PANIC(d);
    const uint8_t* w = b->buf + b->pos;
    uint16_t r = (uint16_t)((uint16_t)w[0] << 8 | (uint16_t)w[1]);
    r = (uint16_t)(r << b->bit_idx);
    r = (uint16_t)(r >> (16 - (uint16_t)num_bits));
    b->bit_idx += num_bits;
    b->pos += (int64_t)(b->bit_idx >> 3);
    b->bit_idx &= 7;
    return (uint8_t)r;
}

/* ReadOne: bitbuffer.go:88-96. */
static uint8_t bb_read_one(alac_oracle* d, bitbuf* b) {
    if (b->pos < 0 || b->pos >= b->size + 4) GO_PANIC(d);
    uint8_t r = (uint8_t)((b->buf[b->pos] >> (7 - b->bit_idx)) & 1);
    b->bit_idx++;
    b->pos += (int64_t)(b->bit_idx >> 3);
    b->bit_idx &= 7;
    return r;
}

/* Advance: bitbuffer.go:99-103 (BitIdx is uint32: the sum wraps). */
static void bb_advance(bitbuf* b, uint32_t num_bits) {
    b->bit_idx += num_bits;
    b->pos += (int64_t)(b->bit_idx >> 3);
    b->bit_idx &= 7;
}

/* ByteAlign: bitbuffer.go:106-112. */
static void bb_byte_align(bitbuf* b) {
    if (b->bit_idx == 0) return;
    bb_advance(b, 8 - b->bit_idx);
}

/* PastEnd: bitbuffer.go:115-117. */
static int bb_past_end(const bitbuf* b) { return b->pos >= b->size; }

/* ---- Golomb (golomb.go) -------------------------------------------------------------- */
#define QBSHIFT 9
#define QB (1u << QBSHIFT)
#define MMULSHIFT 2
#define MDENSHIFT (QBSHIFT - MMULSHIFT - 1)
#define MOFF (1u << (MDENSHIFT - 2))
#define BITOFF 24
#define MAX_PREFIX_16 9
#define MAX_PREFIX_32 9
#define MAX_DATATYPE_BITS_16 16
#define N_MAX_MEAN_CLAMP 0xffffu
#define N_MEAN_CLAMP_VAL 0xffffu
#define MAX_ZERO_RUN 65535u

/* lead: golomb.go:69-71. */
static inline int32_t lead(int32_t m) { return m == 0 ? 32 : (int32_t)__builtin_clz((uint32_t)m); }
/* lg3a: golomb.go:74-76. */
static inline int32_t lg3a(int32_t x) { return 31 - lead(x + 3); }

typedef struct {
    const uint8_t* p; /* input = bitBuf.Buf[bitBuf.Pos:] */
    int64_t len;      /* len(input) */
} gslice;

/* read32bit: golomb.go:81-83. buf[offset:] then a 4-byte load: panics unless offset+4 <= len. */
static inline uint32_t read32bit(alac_oracle* d, gslice in, int64_t offset) {
    if (offset < 0 || offset + 4 > in.len) GO_PANIC(d);
    const uint8_t* q = in.p + offset;
    return (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | (uint32_t)q[3];
}

/* getStreamBits: golomb.go:86-108. */
static uint32_t get_stream_bits(alac_oracle* d, gslice in, uint32_t bit_offset, uint32_t num_bits) {
    uint32_t byte_offset = bit_offset / 8;
    uint32_t load1 = read32bit(d, in, (int64_t)byte_offset);
    if (num_bits + (bit_offset & 7) > 32) {
        uint32_t result = load1 << (bit_offset & 7);
        if ((int64_t)byte_offset + 4 >= in.len) GO_PANIC(d);
        uint32_t load2 = (uint32_t)in.p[byte_offset + 4];
        uint32_t load2shift = 8 - (num_bits + (bit_offset & 7) - 32);
        load2 = go_shr(load2, load2shift);
        result = go_shr(result, 32 - num_bits);
        result |= load2;
        return result;
    }
    uint32_t result = go_shr(load1, 32 - num_bits - (bit_offset & 7));
    if (num_bits < 32) result &= go_shl(1, num_bits) - 1;
    return result;
}

/* dynGet: golomb.go:112-144. */
static uint32_t dyn_get(alac_oracle* d, gslice in, uint32_t bit_pos, uint32_t m, uint32_t k,
                        uint32_t* new_bit_pos) {
    uint32_t temp_bits = bit_pos;
    uint32_t stream_long = read32bit(d, in, (int64_t)(temp_bits >> 3));
    stream_long <<= temp_bits & 7;
    uint32_t pre = (uint32_t)lead((int32_t)~stream_long);
    uint32_t result;
    if (pre >= MAX_PREFIX_16) {
        pre = MAX_PREFIX_16;
        temp_bits += pre;
        stream_long <<= pre;
        result = stream_long >> (32 - MAX_DATATYPE_BITS_16);
        temp_bits += MAX_DATATYPE_BITS_16;
        *new_bit_pos = temp_bits;
        return result;
    }
    temp_bits += pre + 1;
    stream_long <<= pre + 1;
    uint32_t val = go_shr(stream_long, 32 - k);
    temp_bits += k;
    if (val < 2) {
        result = pre * m;
        temp_bits--;
    } else {
        result = pre * m + val - 1;
    }
    *new_bit_pos = temp_bits;
    return result;
}

/* DynDecomp: golomb.go:148-253. pb = params.PB, kb = params.KB, mb0 = params.MB0
 * (SetAGParams golomb.go:55-65; WB = (1<<KB)-1). Returns an alacgpu_code. */
static int dyn_decomp(alac_oracle* d, bitbuf* bits, uint32_t mb0, uint32_t pb, uint32_t kb,
                      int32_t* pred_coefs, int64_t num_samples, int max_size) {
    /* input := bitBuf.Buf[bitBuf.Pos:]  — panics if Pos > len(Buf) */
    if (bits->pos < 0 || bits->pos > bits->size + 4) GO_PANIC(d);
    gslice in = {bits->buf + bits->pos, bits->size + 4 - bits->pos};
    uint32_t start_pos = bits->bit_idx;
    uint32_t max_pos = (uint32_t)(bits->size - bits->pos) * 8u;
    uint32_t bit_pos = start_pos;

    /* predCoefs = predCoefs[:numSamples:numSamples] */
    if (num_samples < 0 || num_samples > (int64_t)d->cfg.frame_length) GO_PANIC(d);

    uint32_t mean = mb0;
    int32_t zmode = 0;
    int64_t count = 0;
    uint32_t wb = go_shl(1, kb) - 1;
    uint32_t residual;

    while (count < num_samples) {
        if (bit_pos >= max_pos) return ALACGPU_ERR_BITSTREAM_OVERRUN;

        uint32_t m = mean >> QBSHIFT;
        int32_t k = lg3a((int32_t)m);
        if ((int32_t)kb < k) k = (int32_t)kb;
        m = go_shl(1, (uint32_t)k) - 1;

        {
            uint32_t stream_long = read32bit(d, in, (int64_t)(bit_pos >> 3));
            stream_long <<= bit_pos & 7;
            residual = (uint32_t)lead((int32_t)~stream_long);
            if (residual >= MAX_PREFIX_32) {
                residual = get_stream_bits(d, in, bit_pos + MAX_PREFIX_32, (uint32_t)max_size);
                bit_pos += MAX_PREFIX_32 + (uint32_t)max_size;
            } else {
                bit_pos += residual + 1;
                if (k != 1) {
                    stream_long <<= residual + 1;
                    uint32_t v = go_shr(stream_long, 32 - (uint32_t)k);
                    if (v >= 2) {
                        residual = residual * m + v - 1;
                        bit_pos += (uint32_t)k;
                    } else {
                        residual *= m;
                        bit_pos += (uint32_t)k - 1;
                    }
                }
            }
        }

        uint32_t ndecode = residual + (uint32_t)zmode;
        int32_t multiplier = -(int32_t)(ndecode & 1);
        multiplier |= 1;
        int32_t del = (int32_t)((ndecode + 1) >> 1) * multiplier;

        pred_coefs[count] = del;
        count++;

        mean = pb * (residual + (uint32_t)zmode) + mean - ((pb * mean) >> QBSHIFT);
        if (residual > N_MAX_MEAN_CLAMP) mean = N_MEAN_CLAMP_VAL;

        zmode = 0;

        if ((mean << MMULSHIFT) < QB && count < num_samples) {
            zmode = 1;
            int32_t k32 = lead((int32_t)mean) - BITOFF + (int32_t)((mean + MOFF) >> MDENSHIFT);
            if (k32 < 0) k32 = 0;
            uint32_t mz = (go_shl(1, (uint32_t)k32) - 1) & wb;

            residual = dyn_get(d, in, bit_pos, mz, (uint32_t)k32, &bit_pos);

            if (count + (int64_t)residual > num_samples) return ALACGPU_ERR_SAMPLE_OVERRUN;

            int64_t end = count + (int64_t)residual;
            memset(pred_coefs + count, 0, (size_t)(end - count) * sizeof(int32_t));
            count = end;

            if (residual >= MAX_ZERO_RUN) zmode = 0;
            mean = 0;
        }
    }

    bb_advance(bits, bit_pos - start_pos);
    return ALACGPU_OK;
}

/* ---- Predictor (predictor.go) -------------------------------------------------------- */
/* signOfInt: predictor.go:35-39. */
static inline int32_t sign_of_int(int32_t v) {
    int32_t negi = (int32_t)((uint32_t)(-v) >> 31);
    return negi | (v >> 31);
}

/* unpcBlock4/5/6/8: predictor.go:99-193,198-310,315-446,449-618. Coefficients live in int32
 * locals for the whole block (no int16 wrap), taps walked from the highest down, the
 * last tap (coef0) updated without touching del0. */
static inline __attribute__((always_inline)) void unpc_block_fixed(const int order, const int32_t* pc1,
                                                                   int32_t* out, int64_t num,
                                                                   int16_t* coefs, uint32_t chan_shift,
                                                                   uint32_t den_shift, int32_t den_half) {
    const int lim = order + 1;
    int32_t c[8];
    for (int j = 0; j < order; j++) c[j] = (int32_t)coefs[j];

    for (int64_t idx = lim; idx < num; idx++) {
        const int32_t* w = out + idx - lim; /* w[0] = top, w[lim-1] = out[idx-1] */
        int32_t top = w[0];
        int32_t diff[8];
        int32_t acc = den_half;
        for (int j = 0; j < order; j++) {
            diff[j] = top - w[lim - 1 - j];
            acc -= c[j] * diff[j];
        }
        int32_t sum1 = go_sar(acc, den_shift);

        int32_t del = pc1[idx];
        int32_t del0 = del;
        int32_t sign = sign_of_int(del);
        del += top + sum1;
        out[idx] = go_sar(go_shl_i(del, chan_shift), chan_shift);

        if (sign > 0) {
            int j;
            for (j = order - 1; j >= 1; j--) {
                int32_t sgn = sign_of_int(diff[j]);
                c[j] -= (int32_t)(int16_t)sgn;
                del0 -= (order - j) * go_sar(sgn * diff[j], den_shift);
                if (del0 <= 0) break;
            }
            if (j == 0) c[0] -= (int32_t)(int16_t)sign_of_int(diff[0]);
        } else if (sign < 0) {
            int j;
            for (j = order - 1; j >= 1; j--) {
                int32_t sgn = -sign_of_int(diff[j]);
                c[j] -= (int32_t)(int16_t)sgn;
                del0 -= (order - j) * go_sar(sgn * diff[j], den_shift);
                if (del0 >= 0) break;
            }
            if (j == 0) c[0] += (int32_t)(int16_t)sign_of_int(diff[0]);
        }
    }
    for (int j = 0; j < order; j++) coefs[j] = (int16_t)c[j];
}

/* unpcBlockGeneral: predictor.go:623-684 (int16 coefficients wrap in place). */
static void unpc_block_general(const int32_t* pc1, int32_t* out, int64_t num, int16_t* coefs,
                               int32_t num_active, int lim, uint32_t chan_shift, uint32_t den_shift,
                               int32_t den_half) {
    int active = (int)num_active;
    for (int64_t idx = lim; idx < num; idx++) {
        const int32_t* hist = out + idx - lim;
        int32_t top = hist[0];
        int32_t sum1 = 0;
        for (int k = 0; k < active; k++) sum1 += (int32_t)coefs[k] * (hist[active - k] - top);

        int32_t del = pc1[idx];
        int32_t del0 = del;
        int32_t sign = sign_of_int(del);
        del += top + go_sar(sum1 + den_half, den_shift);
        out[idx] = go_sar(go_shl_i(del, chan_shift), chan_shift);

        if (sign > 0) {
            for (int k = active - 1; k >= 0; k--) {
                int32_t dd = top - hist[active - k];
                int32_t sgn = sign_of_int(dd);
                coefs[k] = (int16_t)(coefs[k] - (int16_t)sgn);
                del0 -= (int32_t)(active - k) * go_sar(sgn * dd, den_shift);
                if (del0 <= 0) break;
            }
        } else if (sign < 0) {
            for (int k = active - 1; k >= 0; k--) {
                int32_t dd = top - hist[active - k];
                int32_t sgn = sign_of_int(dd);
                coefs[k] = (int16_t)(coefs[k] + (int16_t)sgn);
                del0 -= (int32_t)(active - k) * go_sar(-sgn * dd, den_shift);
                if (del0 >= 0) break;
            }
        }
    }
}

/* UnpcBlock: predictor.go:45-94. pc1/out have len frame_length (decoder.go:104-106). */
static void unpc_block(alac_oracle* d, int32_t* pc1, int32_t* out, int64_t num, int16_t* coefs,
                       int32_t num_active, uint32_t chan_bits, uint32_t den_shift) {
    int64_t buf_len = (int64_t)d->cfg.frame_length;
    uint32_t chan_shift = 32u - chan_bits; /* uint32 wrap when chanBits > 32 */
    int32_t den_half = 0;
    if (den_shift > 0) den_half = (int32_t)go_shl(1, den_shift - 1);

    if (buf_len < 1) GO_PANIC(d); /* out[0] = pc1[0] */
    out[0] = pc1[0];

    if (num_active == 0) {
        if (num > 1 && pc1 != out) memcpy(out + 1, pc1 + 1, (size_t)(num - 1) * sizeof(int32_t));
        return;
    }
    if (num_active == 31) {
        int32_t prev = out[0];
        for (int64_t idx = 1; idx < num; idx++) {
            int32_t del = pc1[idx] + prev;
            prev = go_sar(go_shl_i(del, chan_shift), chan_shift);
            out[idx] = prev;
        }
        return;
    }
    /* warm-up: idx runs to numActive regardless of num; indexes the full-length buffers */
    for (int64_t idx = 1; idx <= (int64_t)num_active; idx++) {
        if (idx >= buf_len) GO_PANIC(d);
        int32_t del = pc1[idx] + out[idx - 1];
        out[idx] = go_sar(go_shl_i(del, chan_shift), chan_shift);
    }
    switch (num_active) {
        case 4: unpc_block_fixed(4, pc1, out, num, coefs, chan_shift, den_shift, den_half); break;
        case 5: unpc_block_fixed(5, pc1, out, num, coefs, chan_shift, den_shift, den_half); break;
        case 6: unpc_block_fixed(6, pc1, out, num, coefs, chan_shift, den_shift, den_half); break;
        case 8: unpc_block_fixed(8, pc1, out, num, coefs, chan_shift, den_shift, den_half); break;
        default:
            unpc_block_general(pc1, out, num, coefs, num_active, (int)num_active + 1, chan_shift,
                               den_shift, den_half);
    }
}

/* ---- matrix.go ------------------------------------------------------------------------ */
/* BytesPerSample: format.go:23-34. */
static int bytes_per_sample(uint8_t depth) {
    switch (depth) {
        case 16: return 2;
        case 20:
        case 24: return 3;
        case 32: return 4;
        default: return 0;
    }
}

/* dst := out[off : off+n : off+n] — panics unless off+n <= len(out). */
static inline uint8_t* out_slice(alac_oracle* d, int64_t off, int n) {
    if (off < 0 || off + n > d->out_len) GO_PANIC(d);
    return d->out + off;
}

static inline void put_le(uint8_t* dst, int32_t v, int bps) {
    for (int b = 0; b < bps; b++) dst[b] = (uint8_t)((uint32_t)v >> (8 * b));
}

/* WriteStereo16/20/24/32: matrix.go:30-215. The 16- and 20-bit writers ignore the shift
 * buffer (matrix.go:30,66); 20-bit shifts left by 4 (matrix.go:77-78,95,101). */
static void write_stereo(alac_oracle* d, int chan_idx, int num_chan, int64_t num_samples, int32_t mix_bits,
                         int32_t mix_res, int bytes_shifted) {
    int depth = d->cfg.bit_depth;
    int bps = bytes_per_sample((uint8_t)depth);
    int64_t stride = (int64_t)num_chan * bps;
    int64_t off = (int64_t)chan_idx * bps;
    uint32_t shift = (uint32_t)bytes_shifted * 8;
    int use_shift = (depth == 24 || depth == 32) && bytes_shifted != 0;
    const int32_t* mix_u = d->mix_u;
    const int32_t* mix_v = d->mix_v;
    if (num_samples > (int64_t)d->cfg.frame_length) GO_PANIC(d); /* mixU[:numSamples:numSamples] */

    for (int64_t idx = 0; idx < num_samples; idx++) {
        int32_t left, right;
        if (mix_res != 0) {
            left = mix_u[idx] + mix_v[idx] - go_sar(mix_res * mix_v[idx], (uint32_t)mix_bits);
            right = left - mix_v[idx];
        } else {
            left = mix_u[idx];
            right = mix_v[idx];
        }
        if (depth == 20) {
            left = go_shl_i(left, 4);
            right = go_shl_i(right, 4);
        }
        if (use_shift) {
            left = go_shl_i(left, shift) | (int32_t)d->shift_buf[idx * 2 + 0];
            right = go_shl_i(right, shift) | (int32_t)d->shift_buf[idx * 2 + 1];
        }
        uint8_t* dst = out_slice(d, off, 2 * bps);
        put_le(dst, left, bps);
        put_le(dst + bps, right, bps);
        off += stride;
    }
}

/* WriteMono16/20/24/32: matrix.go:220-301. */
static void write_mono(alac_oracle* d, int chan_idx, int num_chan, int64_t num_samples, int bytes_shifted) {
    int depth = d->cfg.bit_depth;
    int bps = bytes_per_sample((uint8_t)depth);
    int64_t stride = (int64_t)num_chan * bps;
    int64_t off = (int64_t)chan_idx * bps;
    uint32_t shift = (uint32_t)bytes_shifted * 8;
    int use_shift = (depth == 24 || depth == 32) && bytes_shifted != 0;
    const int32_t* mix_u = d->mix_u;
    if (num_samples > (int64_t)d->cfg.frame_length) GO_PANIC(d);

    for (int64_t idx = 0; idx < num_samples; idx++) {
        int32_t val = mix_u[idx];
        if (depth == 20) val = go_shl_i(val, 4);
        if (use_shift) val = go_shl_i(val, shift) | (int32_t)d->shift_buf[idx];
        uint8_t* dst = out_slice(d, off, bps);
        put_le(dst, val, bps);
        off += stride;
    }
}

/* ---- decoder.go ----------------------------------------------------------------------- */
/* channelLayoutOffsets: decoder.go:55-64. */
static const int channel_layout_offsets[8][8] = {
    {0}, {0, 1}, {2, 0, 1}, {2, 0, 1, 3}, {2, 0, 1, 3, 4},
    {2, 0, 1, 4, 5, 3}, {2, 0, 1, 4, 5, 6, 3}, {2, 6, 7, 0, 1, 4, 5, 3},
};

enum { ELEM_SCE = 0, ELEM_CPE = 1, ELEM_CCE = 2, ELEM_LFE = 3, ELEM_DSE = 4, ELEM_PCE = 5, ELEM_FIL = 6, ELEM_END = 7 };

typedef struct {
    uint32_t mode, den_shift, pb_factor, num;
    int16_t coefs[32];
} chan_params;

/* per-channel predictor header: decoder.go:275-286, 425-450 */
static void read_chan_params(alac_oracle* d, bitbuf* bits, chan_params* p) {
    uint32_t h = bb_read(d, bits, 8);
    p->mode = h >> 4;
    p->den_shift = h & 0xf;
    h = bb_read(d, bits, 8);
    p->pb_factor = h >> 5;
    p->num = h & 0x1f;
    memset(p->coefs, 0, sizeof(p->coefs));
    for (uint32_t i = 0; i < p->num; i++) p->coefs[i] = (int16_t)bb_read(d, bits, 16);
}

/* one channel: SetAGParams + DynDecomp + UnpcBlock (decoder.go:298-311, 464-489) */
static int decode_channel(alac_oracle* d, bitbuf* bits, chan_params* p, int32_t* mix, uint32_t chan_bits,
                          int64_t num_samples) {
    uint32_t pred_bound = d->cfg.pb;
    int err = dyn_decomp(d, bits, d->cfg.mb, (pred_bound * p->pb_factor) / 4, d->cfg.kb, d->predictor,
                         num_samples, (int)chan_bits);
    if (err) return err;
    if (p->mode != 0) unpc_block(d, d->predictor, d->predictor, num_samples, NULL, 31, chan_bits, 0);
    unpc_block(d, d->predictor, mix, num_samples, p->coefs, (int32_t)p->num, chan_bits, p->den_shift);
    return 0;
}

/* decodeSCECompressed: decoder.go:267-324. */
static int32_t decode_sce_compressed(alac_oracle* d, bitbuf* bits, uint32_t chan_bits, int bytes_shifted,
                                     int64_t num_samples) {
    (void)bb_read(d, bits, 8);
    (void)bb_read(d, bits, 8);
    chan_params pu;
    read_chan_params(d, bits, &pu);

    bitbuf shift_bits = *bits;
    if (bytes_shifted != 0) bb_advance(bits, (uint32_t)bytes_shifted * 8u * (uint32_t)num_samples);

    int err = decode_channel(d, bits, &pu, d->mix_u, chan_bits, num_samples);
    if (err) return ALACGPU_STATUS(err, 0, ALACGPU_STAGE_ENTROPY);

    if (bytes_shifted != 0) {
        uint8_t shift = (uint8_t)(bytes_shifted * 8);
        for (int64_t i = 0; i < num_samples; i++) d->shift_buf[i] = (uint16_t)bb_read(d, &shift_bits, shift);
    }
    return 0;
}

/* decodeSCEEscape: decoder.go:326-345. */
static void decode_sce_escape(alac_oracle* d, bitbuf* bits, uint32_t chan_bits, int64_t num_samples) {
    uint32_t shift = 32u - chan_bits;
    if (num_samples > (int64_t)d->cfg.frame_length) GO_PANIC(d);
    if (chan_bits <= 16) {
        for (int64_t i = 0; i < num_samples; i++) {
            int32_t val = (int32_t)bb_read(d, bits, (uint8_t)chan_bits);
            d->mix_u[i] = go_sar(go_shl_i(val, shift), shift);
        }
    } else {
        uint32_t extra = chan_bits - 16;
        for (int64_t i = 0; i < num_samples; i++) {
            int32_t val = (int32_t)bb_read(d, bits, 16);
            val = go_sar(go_shl_i(val, 16), shift);
            d->mix_u[i] = val | (int32_t)bb_read(d, bits, (uint8_t)extra);
        }
    }
}

/* decodeSCE: decoder.go:210-265. Returns a status word; *ns_out = numSamples. */
static int32_t decode_sce(alac_oracle* d, bitbuf* bits, int chan_idx, int num_chan, uint32_t num_samples,
                          uint32_t* ns_out) {
    (void)bb_read_small(d, bits, 4);
    if (bb_read(d, bits, 12) != 0) return ALACGPU_STATUS(ALACGPU_ERR_INVALID_HEADER, 0, 0);
    uint32_t hdr = bb_read(d, bits, 4);
    uint32_t partial = hdr >> 3;
    int bytes_shifted = (int)((hdr >> 1) & 3);
    if (bytes_shifted == 3) return ALACGPU_STATUS(ALACGPU_ERR_INVALID_SHIFT, 0, 0);
    uint32_t escape = hdr & 1;
    uint32_t chan_bits = (uint32_t)d->cfg.bit_depth - (uint32_t)bytes_shifted * 8;
    if (partial != 0) {
        num_samples = bb_read(d, bits, 16) << 16;
        num_samples |= bb_read(d, bits, 16);
    }
    if (escape == 0) {
        int32_t st = decode_sce_compressed(d, bits, chan_bits, bytes_shifted, (int64_t)num_samples);
        if (st) return st;
    } else {
        decode_sce_escape(d, bits, chan_bits, (int64_t)num_samples);
        bytes_shifted = 0;
    }
    write_mono(d, chan_idx, num_chan, (int64_t)num_samples, bytes_shifted);
    *ns_out = num_samples;
    return 0;
}

/* decodeCPECompressed: decoder.go:416-505. */
static int32_t decode_cpe_compressed(alac_oracle* d, bitbuf* bits, uint32_t chan_bits, int bytes_shifted,
                                     int64_t num_samples, int32_t* mix_bits, int32_t* mix_res) {
    *mix_bits = (int32_t)bb_read(d, bits, 8);
    *mix_res = (int32_t)(int8_t)bb_read(d, bits, 8);
    chan_params pu, pv;
    read_chan_params(d, bits, &pu);
    read_chan_params(d, bits, &pv);

    bitbuf shift_bits = *bits;
    if (bytes_shifted != 0) bb_advance(bits, (uint32_t)bytes_shifted * 8u * 2u * (uint32_t)num_samples);

    int err = decode_channel(d, bits, &pu, d->mix_u, chan_bits, num_samples);
    if (err) return ALACGPU_STATUS(err, 0, ALACGPU_STAGE_ENTROPY_U);
    err = decode_channel(d, bits, &pv, d->mix_v, chan_bits, num_samples);
    if (err) return ALACGPU_STATUS(err, 0, ALACGPU_STAGE_ENTROPY_V);

    if (bytes_shifted != 0) {
        uint8_t shift = (uint8_t)(bytes_shifted * 8);
        for (int64_t i = 0; i < num_samples * 2; i += 2) {
            d->shift_buf[i] = (uint16_t)bb_read(d, &shift_bits, shift);
            d->shift_buf[i + 1] = (uint16_t)bb_read(d, &shift_bits, shift);
        }
    }
    return 0;
}

/* decodeCPEEscape: decoder.go:507-535. */
static void decode_cpe_escape(alac_oracle* d, bitbuf* bits, uint32_t chan_bits, int64_t num_samples) {
    uint32_t shift = 32u - chan_bits;
    if (num_samples > (int64_t)d->cfg.frame_length) GO_PANIC(d);
    if (chan_bits <= 16) {
        for (int64_t i = 0; i < num_samples; i++) {
            int32_t val = (int32_t)bb_read(d, bits, (uint8_t)chan_bits);
            d->mix_u[i] = go_sar(go_shl_i(val, shift), shift);
            val = (int32_t)bb_read(d, bits, (uint8_t)chan_bits);
            d->mix_v[i] = go_sar(go_shl_i(val, shift), shift);
        }
    } else {
        uint32_t extra = chan_bits - 16;
        for (int64_t i = 0; i < num_samples; i++) {
            int32_t val = (int32_t)bb_read(d, bits, 16);
            val = go_sar(go_shl_i(val, 16), shift);
            d->mix_u[i] = val | (int32_t)bb_read(d, bits, (uint8_t)extra);
            val = (int32_t)bb_read(d, bits, 16);
            val = go_sar(go_shl_i(val, 16), shift);
            d->mix_v[i] = val | (int32_t)bb_read(d, bits, (uint8_t)extra);
        }
    }
}

/* decodeCPE: decoder.go:348-414. */
static int32_t decode_cpe(alac_oracle* d, bitbuf* bits, int chan_idx, int num_chan, uint32_t num_samples,
                          uint32_t* ns_out) {
    (void)bb_read_small(d, bits, 4);
    if (bb_read(d, bits, 12) != 0) return ALACGPU_STATUS(ALACGPU_ERR_INVALID_HEADER, 0, 0);
    uint32_t hdr = bb_read(d, bits, 4);
    uint32_t partial = hdr >> 3;
    int bytes_shifted = (int)((hdr >> 1) & 3);
    if (bytes_shifted == 3) return ALACGPU_STATUS(ALACGPU_ERR_INVALID_SHIFT, 0, 0);
    uint32_t escape = hdr & 1;
    uint32_t chan_bits = (uint32_t)d->cfg.bit_depth - (uint32_t)bytes_shifted * 8 + 1;
    if (partial != 0) {
        num_samples = bb_read(d, bits, 16) << 16;
        num_samples |= bb_read(d, bits, 16);
    }
    int32_t mix_bits = 0, mix_res = 0;
    if (escape == 0) {
        int32_t st = decode_cpe_compressed(d, bits, chan_bits, bytes_shifted, (int64_t)num_samples, &mix_bits,
                                           &mix_res);
        if (st) return st;
    } else {
        chan_bits = (uint32_t)d->cfg.bit_depth;
        decode_cpe_escape(d, bits, chan_bits, (int64_t)num_samples);
        bytes_shifted = 0;
    }
    write_stereo(d, chan_idx, num_chan, (int64_t)num_samples, mix_bits, mix_res, bytes_shifted);
    *ns_out = num_samples;
    return 0;
}

/* skipFIL: decoder.go:538-552. */
static int skip_fil(alac_oracle* d, bitbuf* bits) {
    int16_t count = (int16_t)bb_read_small(d, bits, 4);
    if (count == 15) count = (int16_t)(count + (int16_t)bb_read_small(d, bits, 8) - 1);
    bb_advance(bits, (uint32_t)count * 8u);
    return bb_past_end(bits) ? ALACGPU_ERR_BITSTREAM_OVERRUN : 0;
}

/* skipDSE: decoder.go:555-574. */
static int skip_dse(alac_oracle* d, bitbuf* bits) {
    (void)bb_read_small(d, bits, 4);
    uint8_t align = bb_read_one(d, bits);
    uint16_t count = (uint16_t)bb_read_small(d, bits, 8);
    if (count == 255) count = (uint16_t)(count + (uint16_t)bb_read_small(d, bits, 8));
    if (align != 0) bb_byte_align(bits);
    bb_advance(bits, (uint32_t)count * 8u);
    return bb_past_end(bits) ? ALACGPU_ERR_BITSTREAM_OVERRUN : 0;
}

/* decodePacketInto: decoder.go:133-207, with output = make([]byte, FrameLength*numChan*bps)
 * as DecodePacket allocates it (decoder.go:120). */
static int32_t decode_packet_into(alac_oracle* d, bitbuf* bits, uint32_t* ns_out) {
    uint32_t num_samples = d->cfg.frame_length;
    int num_chan = d->cfg.num_channels;
    int chan_idx = 0;
    const int* offsets = channel_layout_offsets[num_chan - 1];

    for (;;) {
        if (bb_past_end(bits)) return ALACGPU_STATUS(ALACGPU_ERR_BITSTREAM_OVERRUN, 0, 0);
        uint8_t tag = bb_read_small(d, bits, 3);
        switch (tag) {
            case ELEM_SCE:
            case ELEM_LFE: {
                uint32_t ns;
                int32_t st = decode_sce(d, bits, offsets[chan_idx], num_chan, num_samples, &ns);
                if (st) return st | (ALACGPU_CTX_SCE << 8);
                num_samples = ns;
                chan_idx++;
                break;
            }
            case ELEM_CPE: {
                if (chan_idx + 2 > num_chan) goto done;
                /* DOCUMENTED DEVIATION: a pair mapped to the last output slot (only reachable with an
                 * element order that does not match NumChannels) makes the reference write outside
                 * the frame and panic on a full frame; oracle and kernel both call it malformed. */
                if (offsets[chan_idx] + 2 > num_chan) GO_PANIC(d);
                uint32_t ns;
                int32_t st = decode_cpe(d, bits, offsets[chan_idx], num_chan, num_samples, &ns);
                if (st) return st | (ALACGPU_CTX_CPE << 8);
                num_samples = ns;
                chan_idx += 2;
                break;
            }
            case ELEM_CCE:
            case ELEM_PCE: return ALACGPU_STATUS(ALACGPU_ERR_UNSUPPORTED_ELEMENT, 0, 0);
            case ELEM_DSE: {
                int e = skip_dse(d, bits);
                if (e) return ALACGPU_STATUS(e, ALACGPU_CTX_DSE, 0);
                break;
            }
            case ELEM_FIL: {
                int e = skip_fil(d, bits);
                if (e) return ALACGPU_STATUS(e, ALACGPU_CTX_FIL, 0);
                break;
            }
            case ELEM_END: bb_byte_align(bits); goto done;
        }
        if (chan_idx >= num_chan) break;
    }
done:
    *ns_out = num_samples;
    return 0;
}

/* ---- public API ------------------------------------------------------------------------- */
alac_oracle* alac_oracle_create(const alacgpu_config* cfg) {
    if (!cfg) return NULL;
    /* NewPacketDecoder: decoder.go:90-93 rejects depths outside {16,20,24,32}. NumChannels
     * outside 1..8 index-panics on first use (decoder.go:140); rejected here up front. */
    if (bytes_per_sample(cfg->bit_depth) == 0) return NULL;
    if (cfg->num_channels < 1 || cfg->num_channels > 8) return NULL;
    /* FrameLength 0: every compressed element panics at out[0] = pc1[0] (predictor.go:53);
     * rejected up front like the channel count (include/alacgpu.h, ALACGPU_E_CONFIG). */
    if (cfg->frame_length == 0) return NULL;
    alac_oracle* d = (alac_oracle*)calloc(1, sizeof(*d));
    if (!d) return NULL;
    d->cfg = *cfg;
    size_t n = cfg->frame_length ? cfg->frame_length : 1;
    d->mix_u = (int32_t*)calloc(n, sizeof(int32_t));
    d->mix_v = (int32_t*)calloc(n, sizeof(int32_t));
    d->predictor = (int32_t*)calloc(n, sizeof(int32_t));
    d->shift_buf = (uint16_t*)calloc(n * 2, sizeof(uint16_t));
    return d;
}

void alac_oracle_destroy(alac_oracle* d) {
    if (!d) return;
    free(d->mix_u);
    free(d->mix_v);
    free(d->predictor);
    free(d->shift_buf);
    free(d->padded);
    free(d);
}

size_t alac_oracle_frame_bytes(const alac_oracle* d) {
    return (size_t)d->cfg.frame_length * d->cfg.num_channels * (size_t)bytes_per_sample(d->cfg.bit_depth);
}

int32_t alac_oracle_decode_packet(alac_oracle* d, const uint8_t* packet, size_t len, uint8_t* out,
                                  uint32_t* frames_out) {
    size_t need = len + 4;
    if (d->padded_cap < need) {
        free(d->padded);
        d->padded_cap = need * 2 + 64;
        d->padded = (uint8_t*)malloc(d->padded_cap);
    }
    /* Reset: bitbuffer.go:36-51 */
    if (len) memcpy(d->padded, packet, len);
    memset(d->padded + len, 0, 4);
    bitbuf bits = {d->padded, 0, 0, (int64_t)len};

    d->out = out;
    d->out_len = (int64_t)alac_oracle_frame_bytes(d);
    memset(out, 0, (size_t)d->out_len); /* make([]byte, ...) decoder.go:120 */
    *frames_out = 0;

    if (setjmp(d->panic_jmp)) {
        *frames_out = 0;
        return ALACGPU_STATUS(ALACGPU_ERR_MALFORMED, 0, 0);
    }
    uint32_t ns = 0;
    int32_t st = decode_packet_into(d, &bits, &ns);
    if (st) return st;
    /* output[:n] (decoder.go:127) panics when n > len(output) */
    int64_t nbytes = (int64_t)ns * d->cfg.num_channels * bytes_per_sample(d->cfg.bit_depth);
    if (nbytes > d->out_len) return ALACGPU_STATUS(ALACGPU_ERR_MALFORMED, 0, 0);
    *frames_out = ns;
    return 0;
}

typedef struct {
    const alacgpu_config* cfg;
    const uint8_t* blob;
    const uint64_t* offsets;
    const uint32_t* sizes;
    size_t lo, hi;
    uint8_t* out;
    size_t out_stride;
    uint32_t* frames_out;
    int32_t* status;
    int no_output;
} batch_job;

static void* batch_worker(void* arg) {
    batch_job* j = (batch_job*)arg;
    alac_oracle* d = alac_oracle_create(j->cfg);
    if (!d) return NULL;
    uint8_t* tmp = j->no_output ? (uint8_t*)malloc(alac_oracle_frame_bytes(d) + 1) : NULL;
    for (size_t i = j->lo; i < j->hi; i++) {
        size_t len = j->sizes ? j->sizes[i] : (size_t)(j->offsets[i + 1] - j->offsets[i]);
        uint8_t* o = j->no_output ? tmp : j->out + i * j->out_stride;
        j->status[i] = alac_oracle_decode_packet(d, j->blob + j->offsets[i], len, o, &j->frames_out[i]);
    }
    free(tmp);
    alac_oracle_destroy(d);
    return NULL;
}

int alac_oracle_decode_batch(const alacgpu_config* cfg, const uint8_t* blob, const uint64_t* offsets,
                             const uint32_t* sizes, size_t n, uint8_t* out, size_t out_stride,
                             uint32_t* frames_out, int32_t* status, int threads) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    batch_job* jobs = (batch_job*)calloc((size_t)threads, sizeof(batch_job));
    pthread_t* tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; t++) {
        size_t lo = (size_t)t * per, hi = lo + per;
        if (lo > n) lo = n;
        if (hi > n) hi = n;
        jobs[t] = (batch_job){cfg, blob, offsets, sizes, lo, hi, out, out_stride, frames_out, status, out == NULL};
        if (threads == 1) batch_worker(&jobs[t]);
        else pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
    }
    if (threads > 1)
        for (int t = 0; t < threads; t++) pthread_join(tids[t], NULL);
    free(jobs);
    free(tids);
    return 0;
}
