/*
 * alac_synth.c — synthetic ALAC packet generator (encoder + seeded signal source).
 *
 * The reference (mycophonic/saprobe-alac) is decode-only (README.md:35) and ships no
 * fixtures; its tests make packets with external encoders (tests/conformance_test.go:427-497)
 * that do not exist in this image. This file is the build's own encoder: the exact inverse
 * of the decode path, derived from the decoder lines cited at each step. It
 * produces the benchmark and test inputs; it is host-only C and never on the decode path.
 *
 *   bit layout      decoder.go:133-207 (element walk), :210-235 / :348-376 (element header),
 *                   :267-293 / :416-457 (per-channel header, shift block), :538-574 (FIL/DSE)
 *   entropy coder   inverse of DynDecomp / dynGet  internal/alac/golomb.go:112-253
 *   predictor       inverse of UnpcBlock*          internal/alac/predictor.go:45-684
 *   mix             inverse of WriteStereo*        internal/alac/matrix.go:40-41
 *   shift split     inverse of                     internal/alac/matrix.go:129-132
 */
#include "alac_synth.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- helpers ------------------------------------------------------------------------------- */
static inline uint32_t go_shl(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x << n; }
static inline uint32_t go_shr(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x >> n; }
static inline int32_t go_sar(int32_t x, uint32_t n) { return n >= 32 ? (x < 0 ? -1 : 0) : x >> n; }
static inline int32_t sext(int32_t x, uint32_t chan_shift) {
    return go_sar((int32_t)go_shl((uint32_t)x, chan_shift), chan_shift);
}
static inline int32_t sign_of_int(int32_t v) { return (int32_t)((uint32_t)(-v) >> 31) | (v >> 31); }
static inline int32_t lead(uint32_t m) { return m == 0 ? 32 : (int32_t)__builtin_clz(m); }

static inline uint64_t splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t* s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

static int bytes_per_sample(int depth) { return depth == 16 ? 2 : depth == 32 ? 4 : 3; }

/* channelLayoutOffsets, decoder.go:55-64 */
static const int layout_offsets[8][8] = {
    {0}, {0, 1}, {2, 0, 1}, {2, 0, 1, 3}, {2, 0, 1, 3, 4},
    {2, 0, 1, 4, 5, 3}, {2, 0, 1, 4, 5, 6, 3}, {2, 6, 7, 0, 1, 4, 5, 3},
};
/* element tags per channel count in MPEG order (decoder.go:41-50): 0=SCE 1=CPE 3=LFE */
static const int layout_elems[8][6] = {
    {0, -1}, {1, -1}, {0, 1, -1}, {0, 1, 0, -1}, {0, 1, 1, -1},
    {0, 1, 1, 3, -1}, {0, 1, 1, 0, 3, -1}, {0, 1, 1, 1, 3, -1},
};

int alac_synth_num_elements(int num_channels) {
    if (num_channels < 1 || num_channels > 8) return 0;
    int n = 0;
    while (layout_elems[num_channels - 1][n] >= 0) n++;
    return n;
}

/* ---- MSB-first bit writer -------------------------------------------------------------------- */
typedef struct {
    uint8_t* buf;
    size_t cap;
    uint64_t bitpos;
    int overflow;
} bitw;

static void bw_put(bitw* w, uint32_t val, uint32_t nbits) {
    if (nbits == 0) return;
    if (nbits > 32) { /* no such field exists: a caller's parameter is out of range (see ag_encode) */
        w->overflow = 1;
        return;
    }
    if (nbits < 32) val &= (1u << nbits) - 1u;
    size_t byte = (size_t)(w->bitpos >> 3);
    uint32_t used = (uint32_t)(w->bitpos & 7);
    if (byte + 8 > w->cap) { /* need room for the widest spill */
        w->overflow = 1;
        w->bitpos += nbits; /* keep counting: callers align with `while (bitpos & 7)` */
        return;
    }
    /* merge into a 40-bit big-endian window starting at `byte` */
    uint64_t acc = used ? ((uint64_t)(w->buf[byte] >> (8 - used)) << (8 - used)) << 32 : 0;
    acc |= (uint64_t)val << (40 - used - nbits);
    uint32_t total = used + nbits;
    for (uint32_t i = 0; i * 8 < total; i++) w->buf[byte + i] = (uint8_t)(acc >> (32 - 8 * i));
    w->bitpos += nbits;
}
static void bw_put_ones(bitw* w, uint32_t n) {
    while (n >= 16) {
        bw_put(w, 0xffff, 16);
        n -= 16;
    }
    if (n) bw_put(w, (1u << n) - 1, n);
}

/* ---- adaptive Golomb encoder: inverse of golomb.go:148-253 ------------------------------------- */
/* One code with parameters (m, k): inverse of the inlined dynGet32Bit (golomb.go:178-203) when
 * is_run = 0 (escape literal of maxSize bits), and of dynGet (golomb.go:112-144) when is_run = 1. */
static void ag_put(bitw* w, uint32_t x, uint32_t m, uint32_t k, uint32_t esc_bits, int is_run) {
    uint32_t q = m ? x / m : 9, r = m ? x % m : 0;
    if (m == 0 && x == 0) q = 0; /* m == 0: only x == 0 has a regular code */
    if (q >= 9) {
        bw_put_ones(w, 9);
        bw_put(w, esc_bits >= 32 ? x : (x & (go_shl(1, esc_bits) - 1)), esc_bits);
        return;
    }
    bw_put_ones(w, q);
    bw_put(w, 0, 1);
    if (k == 1 && !is_run) return; /* golomb.go:188: k == 1 carries no remainder */
    if (r == 0) {
        if (k >= 1) bw_put(w, 0, k - 1); /* decoder sees v < 2 and gives back one bit */
    } else {
        bw_put(w, r + 1, k);
    }
}

static void ag_encode(bitw* w, const int32_t* res, uint32_t n, uint32_t mb, uint32_t pb, uint32_t kb,
                      uint32_t chan_bits) {
    uint32_t mean = mb, zmode = 0, wb = go_shl(1, kb) - 1;
    uint32_t c = 0;
    while (c < n) {
        uint32_t m = mean >> 9;
        int32_t k = 31 - lead(m + 3);
        if ((int32_t)kb < k) k = (int32_t)kb;
        m = go_shl(1, (uint32_t)k) - 1;

        int32_t del = res[c];
        uint32_t nn = del >= 0 ? 2u * (uint32_t)del : 2u * (uint32_t)(-(int64_t)del) - 1u;
        uint32_t x = nn - zmode;
        /* a value the regular code cannot carry takes the 9-ones escape with chan_bits literal */
        ag_put(w, x, m, (uint32_t)k, chan_bits, 0);
        c++;

        mean = pb * nn + mean - ((pb * mean) >> 9);
        if (x > 0xffff) mean = 0xffff;
        zmode = 0;

        if ((mean << 2) < 512 && c < n) {
            zmode = 1;
            int32_t k32 = lead(mean) - 24 + (int32_t)((mean + 16) >> 6);
            if (k32 < 0) k32 = 0;
            /* With pb > 127 the product pb * mean wraps (golomb.go:215), the mean no longer contracts and can pass 2^30,
             * where mean << 2 wraps too and a "zero run" starts with k32 in the millions: the reference then jumps
             * millions of bits ahead (dynGet, golomb.go:131-139) and fails. Nothing can be encoded for that state. */
            if (k32 > 24) {
                w->overflow = 1;
                return;
            }
            uint32_t mz = (go_shl(1, (uint32_t)k32) - 1) & wb;
            uint32_t run = 0;
            while (c + run < n && res[c + run] == 0 && run < 65535) run++;
            ag_put(w, run, mz, (uint32_t)k32, 16, 1);
            c += run;
            if (run >= 65535) zmode = 0;
            mean = 0;
        }
    }
}

/* ---- forward predictor: inverse of predictor.go:45-94 ------------------------------------------- */
static void pc_block(const int32_t* in, int32_t* res, uint32_t num, int16_t* coefs, int num_active,
                     uint32_t chan_bits, uint32_t den_shift) {
    uint32_t chan_shift = 32u - chan_bits;
    int32_t den_half = den_shift ? (int32_t)go_shl(1, den_shift - 1) : 0;
    if (num == 0) return;
    res[0] = in[0];
    if (num_active == 0) {
        for (uint32_t i = 1; i < num; i++) res[i] = in[i];
        return;
    }
    if (num_active == 31) {
        for (uint32_t i = 1; i < num; i++) res[i] = sext(in[i] - in[i - 1], chan_shift);
        return;
    }
    for (uint32_t i = 1; i <= (uint32_t)num_active && i < num; i++) res[i] = sext(in[i] - in[i - 1], chan_shift);

    /* orders 4/5/6/8 keep int32 coefficients (predictor.go:107-110); others wrap int16 (:664,:675) */
    int wrap16 = !(num_active == 4 || num_active == 5 || num_active == 6 || num_active == 8);
    int32_t c[32];
    for (int j = 0; j < num_active; j++) c[j] = coefs[j];
    uint32_t lim = (uint32_t)num_active + 1;
    for (uint32_t idx = lim; idx < num; idx++) {
        const int32_t* w = in + idx - lim;
        int32_t top = w[0];
        int32_t acc = den_half;
        for (int j = 0; j < num_active; j++) acc -= c[j] * (top - w[lim - 1 - (uint32_t)j]);
        int32_t sum1 = go_sar(acc, den_shift);
        int32_t del = sext(in[idx] - top - sum1, chan_shift);
        res[idx] = del;
        int32_t del0 = del;
        int32_t sign = sign_of_int(del);
        if (sign == 0) continue;
        for (int j = num_active - 1; j >= 0; j--) {
            int32_t dd = top - w[lim - 1 - (uint32_t)j];
            int32_t sgn = sign > 0 ? sign_of_int(dd) : -sign_of_int(dd);
            c[j] -= sgn;
            if (wrap16) c[j] = (int16_t)c[j];
            if (j == 0 && !wrap16) break; /* fixed orders: last tap leaves del0 alone */
            del0 -= (num_active - j) * go_sar(sgn * dd, den_shift);
            if (sign > 0 ? del0 <= 0 : del0 >= 0) break;
        }
    }
    for (int j = 0; j < num_active; j++) coefs[j] = (int16_t)c[j];
}

/* ---- one channel: header + residual stream -------------------------------------------------------- */
typedef struct {
    int16_t coefs[32];
    int32_t* res;
} chan_work;

static void prepare_channel(const int32_t* in, uint32_t num, const alac_synth_elem* ep, int which, uint32_t chan_bits,
                            chan_work* cw, int32_t* tmp) {
    int order = which ? ep->order_v : ep->order_u;
    int mode = which ? ep->mode_v : ep->mode_u;
    const int16_t* given = which ? ep->coefs_v : ep->coefs_u;
    uint32_t chan_shift = 32u - chan_bits;
    memset(cw->coefs, 0, sizeof(cw->coefs));
    if (order > 0 && order < 31) {
        if (ep->coef_mode == ALAC_SYNTH_COEF_GIVEN) {
            memcpy(cw->coefs, given, sizeof(int16_t) * (size_t)order);
        } else if (ep->coef_mode == ALAC_SYNTH_COEF_RANDOM) {
            uint64_t s = ep->seed ^ (which ? 0xA5A5A5A5ull : 0x5A5A5A5Aull);
            for (int j = 0; j < order; j++) cw->coefs[j] = (int16_t)(splitmix64(&s) & 0xffff);
        } else {
            /* Apple-style start (AINIT 38, BINIT -29, CINIT -2 scaled by 2^denShift/16) [ext],
             * then one adaptation pass over this block so the header carries warm coefficients */
            int32_t den = (int32_t)go_shl(1, ep->den_shift);
            cw->coefs[0] = (int16_t)((38 * den) >> 4);
            if (order > 1) cw->coefs[1] = (int16_t)((-29 * den) >> 4);
            if (order > 2) cw->coefs[2] = (int16_t)((-2 * den) >> 4);
            int16_t warm[32];
            memcpy(warm, cw->coefs, sizeof(warm));
            pc_block(in, tmp, num, warm, order, chan_bits, ep->den_shift);
            memcpy(cw->coefs, warm, sizeof(warm));
        }
    }
    int16_t run[32];
    memcpy(run, cw->coefs, sizeof(run));
    pc_block(in, cw->res, num, run, order, chan_bits, ep->den_shift);
    if (mode != 0 && num > 0) {
        /* decoder.go:307-309: a delta pass (numActive 31, denShift 0) runs before the coefficient pass */
        int32_t prev = cw->res[0];
        for (uint32_t i = 1; i < num; i++) {
            int32_t cur = cw->res[i];
            cw->res[i] = sext(cur - prev, chan_shift);
            prev = cur;
        }
    }
}

static void put_chan_header(bitw* w, const alac_synth_elem* ep, int which, const chan_work* cw) {
    int order = which ? ep->order_v : ep->order_u;
    int mode = which ? ep->mode_v : ep->mode_u;
    bw_put(w, (uint32_t)(mode & 0xf), 4);
    bw_put(w, (uint32_t)(ep->den_shift & 0xf), 4);
    bw_put(w, (uint32_t)(ep->pb_factor & 7), 3);
    bw_put(w, (uint32_t)(order & 0x1f), 5);
    for (int j = 0; j < order; j++) bw_put(w, (uint16_t)cw->coefs[j], 16);
}

/* ---- one element ------------------------------------------------------------------------------------ */
typedef struct {
    int32_t *a, *b, *ra, *rb, *tmp; /* per-thread scratch, each >= frame_length */
} scratch;

static void put_elem_header(bitw* w, int tag, int instance, int partial, int bytes_shifted, int escape,
                            uint32_t num) {
    bw_put(w, (uint32_t)tag, 3);
    bw_put(w, (uint32_t)instance & 0xf, 4);
    bw_put(w, 0, 12);
    bw_put(w, (uint32_t)((partial << 3) | (bytes_shifted << 1) | escape), 4);
    if (partial) {
        bw_put(w, num >> 16, 16);
        bw_put(w, num & 0xffff, 16);
    }
}

static void put_escape_samples(bitw* w, const int32_t* a, const int32_t* b, uint32_t num, uint32_t chan_bits) {
    uint32_t mask = chan_bits >= 32 ? 0xffffffffu : (go_shl(1, chan_bits) - 1);
    for (uint32_t i = 0; i < num; i++) {
        bw_put(w, (uint32_t)a[i] & mask, chan_bits);
        if (b) bw_put(w, (uint32_t)b[i] & mask, chan_bits);
    }
}

/* Encode one SCE/LFE (right == NULL) or CPE. left/right are the element's channels in the PCM domain. */
static void encode_element(bitw* w, const alacgpu_config* cfg, const alac_synth_elem* ep, int tag, int instance,
                           const int32_t* left, const int32_t* right, uint32_t num, scratch* s) {
    int depth = cfg->bit_depth;
    int stereo = right != NULL;
    int partial = ep->partial || num != cfg->frame_length;
    int bs = ep->bytes_shifted;
    uint32_t chan_bits = (uint32_t)depth - 8u * (uint32_t)bs + (stereo ? 1u : 0u);
    uint32_t chan_shift = 32u - chan_bits;
    uint32_t pb = ((uint32_t)cfg->pb * (uint32_t)(ep->pb_factor & 7)) / 4;

    int escape = ep->force_escape;
    size_t esc_bits = (size_t)num * (size_t)depth * (stereo ? 2 : 1);
    bitw save = *w;

    if (!escape) {
        uint32_t smask = go_shl(1, 8u * (uint32_t)bs) - 1;
        /* shift split: matrix.go:129-132 rebuilds (x<<8bs)|low */
        for (uint32_t i = 0; i < num; i++) {
            s->a[i] = go_sar(left[i], 8u * (uint32_t)bs);
            if (stereo) s->b[i] = go_sar(right[i], 8u * (uint32_t)bs);
        }
        if (stereo) {
            /* inverse of matrix.go:40-41: v = L-R, u = R + ((mixRes*v)>>mixBits) */
            for (uint32_t i = 0; i < num; i++) {
                int32_t l = s->a[i], r = s->b[i];
                int32_t v = l - r;
                int32_t u = ep->mix_res != 0 ? r + go_sar((int32_t)ep->mix_res * v, ep->mix_bits) : l;
                if (ep->mix_res == 0) v = r;
                s->a[i] = sext(u, chan_shift);
                s->b[i] = sext(v, chan_shift);
            }
        } else {
            for (uint32_t i = 0; i < num; i++) s->a[i] = sext(s->a[i], chan_shift);
        }
        chan_work cu = {{0}, s->ra}, cv = {{0}, s->rb};
        prepare_channel(s->a, num, ep, 0, chan_bits, &cu, s->tmp);
        if (stereo) prepare_channel(s->b, num, ep, 1, chan_bits, &cv, s->tmp);

        put_elem_header(w, tag, instance, partial, bs, 0, num);
        bw_put(w, stereo ? ep->mix_bits : 0, 8);
        bw_put(w, stereo ? (uint8_t)ep->mix_res : 0, 8);
        put_chan_header(w, ep, 0, &cu);
        if (stereo) put_chan_header(w, ep, 1, &cv);
        if (bs) {
            for (uint32_t i = 0; i < num; i++) {
                bw_put(w, (uint32_t)left[i] & smask, 8u * (uint32_t)bs);
                if (stereo) bw_put(w, (uint32_t)right[i] & smask, 8u * (uint32_t)bs);
            }
        }
        uint64_t ent_start = w->bitpos;
        ag_encode(w, cu.res, num, cfg->mb, pb, cfg->kb, chan_bits);
        if (stereo) ag_encode(w, cv.res, num, cfg->mb, pb, cfg->kb, chan_bits);
        /* escape whenever the compressed form is not smaller than raw (as real encoders do) */
        if (!ep->never_escape && (w->bitpos - ent_start) + (uint64_t)bs * 8u * num * (stereo ? 2 : 1) >= esc_bits)
            escape = 1;
    }
    if (escape) {
        *w = save;
        put_elem_header(w, tag, instance, partial, 0, 1, num);
        put_escape_samples(w, left, right, num, (uint32_t)depth);
    }
}

/* ---- packet ------------------------------------------------------------------------------------------- */
size_t alac_synth_encode_packet(const alacgpu_config* cfg, const alac_synth_elem* elems, const int32_t* pcm,
                                uint32_t num_frames, uint32_t flags, uint8_t* out, size_t out_cap) {
    int nch = cfg->num_channels;
    if (nch < 1 || nch > 8 || num_frames > cfg->frame_length) return 0;
    size_t fl = cfg->frame_length ? cfg->frame_length : 1;
    int32_t* mem = (int32_t*)malloc(sizeof(int32_t) * fl * 7);
    if (!mem) return 0;
    scratch s = {mem, mem + fl, mem + 2 * fl, mem + 3 * fl, mem + 4 * fl};
    int32_t* left = mem + 5 * fl;
    int32_t* right = mem + 6 * fl;
    bitw w = {out, out_cap, 0, 0};

    if (flags & ALAC_SYNTH_FLAG_LEADING_FIL) {
        /* FIL, decoder.go:538-552: 4-bit count (15 -> +8-bit-1), then count bytes */
        bw_put(&w, 6, 3);
        bw_put(&w, 3, 4);
        bw_put(&w, 0xABCDEF, 24);
    }
    int chan_idx = 0;
    for (int e = 0; layout_elems[nch - 1][e] >= 0; e++) {
        int tag = layout_elems[nch - 1][e];
        int stereo = tag == 1;
        int o = layout_offsets[nch - 1][chan_idx];
        for (uint32_t i = 0; i < num_frames; i++) {
            left[i] = pcm[(size_t)i * (size_t)nch + (size_t)o];
            if (stereo) right[i] = pcm[(size_t)i * (size_t)nch + (size_t)o + 1];
        }
        if (e == (layout_elems[nch - 1][1] >= 0 ? 1 : 0) && (flags & ALAC_SYNTH_FLAG_MID_DSE)) {
            /* DSE, decoder.go:555-574: tag, align flag, 8-bit count (255 -> +8 bits), align, bytes */
            bw_put(&w, 4, 3);
            bw_put(&w, 0, 4);
            bw_put(&w, 1, 1);
            bw_put(&w, 2, 8);
            while (w.bitpos & 7) bw_put(&w, 0, 1);
            bw_put(&w, 0xBEEF, 16);
        }
        encode_element(&w, cfg, &elems[e], tag, e, left, stereo ? right : NULL, num_frames, &s);
        chan_idx += stereo ? 2 : 1;
    }
    if (!(flags & ALAC_SYNTH_FLAG_NO_END)) bw_put(&w, 7, 3);
    while (w.bitpos & 7) bw_put(&w, 0, 1);
    free(mem);
    if (w.overflow) return 0;
    return (size_t)(w.bitpos >> 3);
}

/* ---- seeded signal source (SURVEY.md §8d signal model) ------------------------------------------------- */
/* "music-like": six sinusoids (60 Hz - 8 kHz, log-uniform) + AR(2)-shaped noise, correlated channels. */
static void gen_music(uint64_t seed, const alacgpu_config* cfg, uint32_t num_frames, int32_t* pcm) {
    int nch = cfg->num_channels, depth = cfg->bit_depth;
    double full = ldexp(1.0, depth - 1) - 1.0;
    double rate = cfg->sample_rate ? (double)cfg->sample_rate : 44100.0;
    uint64_t s = seed;
    double f[6], a[6], ph[6];
    double asum = 0;
    for (int j = 0; j < 6; j++) {
        f[j] = 60.0 * pow(8000.0 / 60.0, u01(&s));
        a[j] = 0.2 + u01(&s);
        asum += a[j];
        ph[j] = 6.283185307179586 * u01(&s);
    }
    double peak = 0.25 * full; /* about -12 dBFS */
    for (int j = 0; j < 6; j++) a[j] *= peak / asum;
    double noise_amp = full * pow(10.0, (-50.0 + 20.0 * u01(&s)) / 20.0);
    double common[2] = {0, 0};
    double chs[8][2];
    double w[8][6], chph[8][6];
    memset(chs, 0, sizeof(chs));
    for (int c = 0; c < nch; c++)
        for (int j = 0; j < 6; j++) {
            w[c][j] = 0.6 + 0.4 * u01(&s);
            chph[c][j] = 0.3 * u01(&s);
        }
    double lo = -ldexp(1.0, depth - 1), hi = full;
    /* oscillators by complex rotation: state (cs,sn) of channel c, partial j */
    double rc[6], rs[6], cs[8][6], sn[8][6];
    for (int j = 0; j < 6; j++) {
        double w0 = 6.283185307179586 * f[j] / rate;
        rc[j] = cos(w0);
        rs[j] = sin(w0);
        for (int c = 0; c < nch; c++) {
            cs[c][j] = cos(ph[j] + chph[c][j]);
            sn[c][j] = sin(ph[j] + chph[c][j]);
        }
    }
    for (uint32_t i = 0; i < num_frames; i++) {
        /* AR(2) low-passed noise shared by all channels + a smaller independent part per channel */
        uint64_t r0 = splitmix64(&s);
        double e = ((double)(r0 & 0xfffff) + (double)((r0 >> 20) & 0xfffff) + (double)((r0 >> 40) & 0xfffff)) *
                       (2.0 / 1048576.0) - 3.0;
        double n0 = 1.6 * common[0] - 0.68 * common[1] + 0.08 * e;
        common[1] = common[0];
        common[0] = n0;
        for (int c = 0; c < nch; c++) {
            double x = 0;
            for (int j = 0; j < 6; j++) {
                x += w[c][j] * a[j] * sn[c][j];
                double ncs = cs[c][j] * rc[j] - sn[c][j] * rs[j];
                sn[c][j] = sn[c][j] * rc[j] + cs[c][j] * rs[j];
                cs[c][j] = ncs;
            }
            uint64_t r1 = splitmix64(&s);
            double ei = ((double)(r1 & 0xffffff) + (double)((r1 >> 24) & 0xffffff)) * (2.0 / 16777216.0) - 2.0;
            double ni = 1.2 * chs[c][0] - 0.4 * chs[c][1] + 0.2 * ei;
            chs[c][1] = chs[c][0];
            chs[c][0] = ni;
            x += noise_amp * (n0 * 3.0 + 0.3 * ni);
            double q = floor(x + 0.5);
            if (q < lo) q = lo;
            if (q > hi) q = hi;
            pcm[(size_t)i * (size_t)nch + (size_t)c] = (int32_t)q;
        }
    }
}

/* uniform white noise at full scale: incompressible, drives the escape path (docs/QA.md:140-147) */
static void gen_noise(uint64_t seed, const alacgpu_config* cfg, uint32_t num_frames, int32_t* pcm) {
    int nch = cfg->num_channels, depth = cfg->bit_depth;
    uint64_t s = seed;
    for (size_t i = 0; i < (size_t)num_frames * (size_t)nch; i++) {
        uint32_t r = (uint32_t)splitmix64(&s);
        pcm[i] = (int32_t)(r << (32 - depth)) >> (32 - depth);
    }
}

/* small-amplitude signal with silent stretches: exercises zero runs (golomb.go:223-246) */
static void gen_quiet(uint64_t seed, const alacgpu_config* cfg, uint32_t num_frames, int32_t* pcm) {
    int nch = cfg->num_channels;
    uint64_t s = seed;
    int lowbits = cfg->bit_depth > 16 ? cfg->bit_depth - 16 : 0;
    for (uint32_t i = 0; i < num_frames; i++) {
        int silent = ((i / 97) % 3) == 1;
        for (int c = 0; c < nch; c++) {
            int32_t v = 0;
            if (!silent) {
                uint32_t r = (uint32_t)splitmix64(&s);
                v = (int32_t)(r % 7) - 3;
                if ((r >> 8) % 61 == 0) v *= 40;
            }
            pcm[(size_t)i * (size_t)nch + (size_t)c] = (int32_t)((uint32_t)v << lowbits);
        }
    }
}

void alac_synth_signal(const alacgpu_config* cfg, int profile, uint64_t seed, uint32_t num_frames, int32_t* pcm) {
    if (profile == ALAC_SYNTH_PROFILE_NOISE) gen_noise(seed, cfg, num_frames, pcm);
    else if (profile == ALAC_SYNTH_PROFILE_QUIET) gen_quiet(seed, cfg, num_frames, pcm);
    else gen_music(seed, cfg, num_frames, pcm);
}

/* Encoder settings for packet `seed` under a profile. */
void alac_synth_params(const alacgpu_config* cfg, int profile, uint64_t seed, alac_synth_elem* elems,
                       uint32_t* num_frames, uint32_t* flags) {
    int ne = alac_synth_num_elements(cfg->num_channels);
    uint64_t s = seed ^ 0xC0FFEE1234ull;
    *num_frames = cfg->frame_length;
    *flags = 0;
    uint32_t pick = (uint32_t)(splitmix64(&s) % 1000);
    if (profile != ALAC_SYNTH_PROFILE_STRESS) {
        if (pick < 10 && cfg->frame_length > 1) /* 1 % partial frames */
            *num_frames = 1 + (uint32_t)(splitmix64(&s) % (cfg->frame_length - 1));
    } else if (pick < 200 && cfg->frame_length > 1) {
        *num_frames = (uint32_t)(splitmix64(&s) % (cfg->frame_length + 1));
    }
    for (int e = 0; e < ne; e++) {
        alac_synth_elem* ep = &elems[e];
        memset(ep, 0, sizeof(*ep));
        ep->seed = splitmix64(&s);
        uint32_t r = (uint32_t)splitmix64(&s);
        if (profile == ALAC_SYNTH_PROFILE_STRESS) {
            /* parity-only distribution: random orders incl. 0, 31 and the general path, random
             * int16 coefficients, denShift 0-15, mixRes -128..127, modes, shifts, escapes */
            ep->order_u = (uint8_t)(r % 32);
            ep->order_v = (uint8_t)((r >> 5) % 32);
            if ((r >> 10) % 3 == 0) {
                static const uint8_t common[] = {0, 4, 5, 6, 8, 31, 1, 2, 3, 7, 9, 12, 30};
                ep->order_u = common[(r >> 12) % 13];
                ep->order_v = common[(r >> 16) % 13];
            }
            ep->den_shift = (uint8_t)((r >> 20) % 16);
            ep->mode_u = (uint8_t)(((r >> 24) % 4 == 0) ? 1 + (r >> 26) % 15 : 0);
            ep->mode_v = (uint8_t)(((r >> 28) % 4 == 0) ? 1 : 0);
            uint32_t r2 = (uint32_t)splitmix64(&s);
            ep->pb_factor = (uint8_t)(r2 % 8);
            ep->mix_bits = (uint8_t)((r2 >> 3) % 5 == 0 ? (r2 >> 8) % 256 : (r2 >> 8) % 8);
            ep->mix_res = (int8_t)((r2 >> 16) & 0xff);
            int maxbs = cfg->bit_depth >= 24 ? 2 : (cfg->bit_depth == 20 ? 1 : (cfg->bit_depth == 16 ? 1 : 0));
            ep->bytes_shifted = (uint8_t)((r2 >> 24) % 3 == 0 ? (r2 >> 26) % (uint32_t)(maxbs + 1) : 0);
            ep->coef_mode = (r2 >> 28) % 2 ? ALAC_SYNTH_COEF_RANDOM : ALAC_SYNTH_COEF_WARM;
            ep->force_escape = (uint8_t)((r2 >> 30) % 4 == 0 && (r % 7 == 0));
            ep->never_escape = (uint8_t)((r % 5) != 0);
            ep->partial = (uint8_t)((r2 >> 29) & 1 & (r % 11 == 0));
        } else {
            uint32_t oc = r % 100;
            int order = oc < 30 ? 4 : oc < 60 ? 6 : oc < 90 ? 8 : (oc < 95 || profile == ALAC_SYNTH_PROFILE_MUSIC_LE8) ? 5 : 12;
            ep->order_u = ep->order_v = (uint8_t)order;
            if (profile == ALAC_SYNTH_PROFILE_MUSIC_MIXED) { /* every channel picks its own order, as ffmpeg's encoder does */
                ep->order_u = (uint8_t)(4 + (r >> 20) % 3);
                ep->order_v = (uint8_t)(4 + (r >> 24) % 3);
            }
            ep->den_shift = 9;
            ep->pb_factor = 4;
            ep->mix_bits = 2;
            ep->mix_res = (int8_t)((r >> 8) % 3);
            ep->bytes_shifted = (uint8_t)(cfg->bit_depth == 24 ? 1 : cfg->bit_depth == 32 ? 2 : 0);
            if (profile == ALAC_SYNTH_PROFILE_MUSIC_NOSHIFT) {
                ep->bytes_shifted = 0; /* chanBits 24/25, 32/33 */
                ep->never_escape = 1;  /* the live low bytes make the element larger than raw PCM: keep it compressed */
            }
            ep->coef_mode = ALAC_SYNTH_COEF_WARM;
            ep->force_escape = (uint8_t)(((r >> 16) % 200) == 0); /* 0.5 % escape */
        }
    }
    if (profile == ALAC_SYNTH_PROFILE_STRESS) {
        uint32_t r = (uint32_t)splitmix64(&s);
        if (r % 9 == 0) *flags |= ALAC_SYNTH_FLAG_LEADING_FIL;
        if (r % 7 == 0) *flags |= ALAC_SYNTH_FLAG_MID_DSE;
        if (r % 13 == 0) *flags |= ALAC_SYNTH_FLAG_NO_END;
    }
}

/* expected decoder output for a source block: interleaved LE PCM (matrix.go byte order) */
void alac_synth_pack_pcm(const alacgpu_config* cfg, const int32_t* pcm, uint32_t num_frames, uint8_t* out) {
    int bps = bytes_per_sample(cfg->bit_depth);
    size_t n = (size_t)num_frames * cfg->num_channels;
    for (size_t i = 0; i < n; i++) {
        uint32_t v = (uint32_t)pcm[i];
        if (cfg->bit_depth == 20) v <<= 4;
        for (int b = 0; b < bps; b++) out[i * (size_t)bps + (size_t)b] = (uint8_t)(v >> (8 * b));
    }
}

/* ---- threaded batch generation ---------------------------------------------------------------------------- */
typedef struct {
    const alacgpu_config* cfg;
    int profile;
    uint64_t base_seed;
    size_t first_index, lo, hi;
    uint8_t* slots;
    size_t slot_bytes;
    uint32_t* sizes;
    uint32_t* frames;
    uint8_t* pcm_out;
    size_t pcm_stride;
} gen_job;

static void* gen_worker(void* arg) {
    gen_job* j = (gen_job*)arg;
    const alacgpu_config* cfg = j->cfg;
    size_t fl = cfg->frame_length ? cfg->frame_length : 1;
    int32_t* pcm = (int32_t*)malloc(sizeof(int32_t) * fl * cfg->num_channels);
    alac_synth_elem elems[8];
    for (size_t i = j->lo; i < j->hi; i++) {
        uint64_t s = j->base_seed ^ (uint64_t)(j->first_index + i);
        uint64_t seed = splitmix64(&s);
        uint32_t nf, flags;
        alac_synth_params(cfg, j->profile, seed, elems, &nf, &flags);
        int sig = j->profile;
        if (j->profile == ALAC_SYNTH_PROFILE_STRESS) sig = (int)(seed % 3); /* music / noise / quiet */
        alac_synth_signal(cfg, sig, seed, nf, pcm);
        size_t len = alac_synth_encode_packet(cfg, elems, pcm, nf, flags, j->slots + i * j->slot_bytes, j->slot_bytes);
        j->sizes[i] = (uint32_t)len;
        j->frames[i] = nf;
        if (j->pcm_out) alac_synth_pack_pcm(cfg, pcm, nf, j->pcm_out + i * j->pcm_stride);
    }
    free(pcm);
    return NULL;
}

size_t alac_synth_slot_bytes(const alacgpu_config* cfg) {
    /* worst case of the compressed form, which the stress profile may be told to keep however large it gets: an
     * escape-coded sample (9 + chanBits <= 42 bits, shift byte included) or a regular one followed by an
     * escape-coded zero-run length (<= 23 + 25 bits), so 9 bytes per sample; + headers + FIL/DSE extras */
    size_t raw = (size_t)cfg->frame_length * cfg->num_channels * 9u;
    return (raw + 8u * 80u + 256u + 15u) & ~(size_t)15u;
}

int alac_synth_gen_batch(const alacgpu_config* cfg, int profile, uint64_t base_seed, size_t first_index, size_t n,
                         uint8_t* slots, size_t slot_bytes, uint32_t* sizes, uint32_t* frames, uint8_t* pcm_out,
                         size_t pcm_stride, int threads) {
    if (cfg->num_channels < 1 || cfg->num_channels > 8) return -1;
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    gen_job* jobs = (gen_job*)calloc((size_t)threads, sizeof(gen_job));
    pthread_t* tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; t++) {
        size_t lo = (size_t)t * per, hi = lo + per;
        if (lo > n) lo = n;
        if (hi > n) hi = n;
        jobs[t] = (gen_job){cfg, profile, base_seed, first_index, lo, hi, slots, slot_bytes, sizes, frames, pcm_out, pcm_stride};
        if (threads == 1) gen_worker(&jobs[t]);
        else pthread_create(&tids[t], NULL, gen_worker, &jobs[t]);
    }
    if (threads > 1)
        for (int t = 0; t < threads; t++) pthread_join(tids[t], NULL);
    free(jobs);
    free(tids);
    return 0;
}

/* Pack slots into the device blob layout: packet i at offsets[i] (16-byte aligned), followed by
 * >= `pad` zero bytes. In place, ascending (packed offset <= slot offset). Returns total bytes. */
size_t alac_synth_compact(uint8_t* slots, size_t slot_bytes, const uint32_t* sizes, size_t n, size_t pad,
                          uint64_t* offsets) {
    size_t w = 0;
    for (size_t i = 0; i < n; i++) {
        size_t len = sizes[i];
        if (w != i * slot_bytes) memmove(slots + w, slots + i * slot_bytes, len);
        offsets[i] = w;
        size_t end = w + len;
        size_t next = (end + pad + 15u) & ~(size_t)15u;
        size_t limit = (i + 1) * slot_bytes; /* never clobber the next unread slot */
        if (next > limit) next = limit;
        memset(slots + end, 0, next - end);
        w = next;
    }
    return w;
}
