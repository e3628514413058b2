/*
 * alac_regular.h — the lean wave decoder for REGULAR packets (the overwhelming majority of real streams).
 *
 * A packet is regular when classify_regular() accepts it: 16-bit, mono or stereo, its first tag is the one
 * element that covers the whole frame (SCE/LFE for 1 channel, CPE for 2), compressed, no shift bytes,
 * mode 0 on both channels, predictor orders in {4,5,6,8} (the reference's unrolled int32-coefficient
 * predictors, predictor.go:81-93), header inside the packet. Everything else takes decode_wave (alac_wave.h),
 * which decodes any packet. The two produce identical bytes and status words; which one runs is a speed
 * choice made per packet by the classifier, and waves are built from packets with the same (numU, numV).
 *
 * What "lean" buys: with the orders wave-uniform the predictor is compiled for exactly NA
 * taps and the warm-up test is scalar; the per-sample step has no data-dependent branch on its common path
 * (zero-run countdown, the in-range test and dead lanes are selects; only an escape code, the start of a
 * zero run or an error take the one slow branch); |diff| + rounding is one v_sad_u32 on a sign-biased history;
 * all products are 24-bit (v_mad_i32_i24 / v_mad_u32_u24), exact because chanBits <= 17 here.
 */
#ifndef ALAC_REGULAR_H
#define ALAC_REGULAR_H

#include "alac_wave.h"

#ifndef ALAC_NOINLINE
#define ALAC_NOINLINE
#endif
#ifndef ALAC_SAD
/* |a - b| + c on unsigned operands: v_sad_u32 on the GPU */
#define ALAC_SAD(a, b, c) (((a) > (b) ? (a) - (b) : (b) - (a)) + (c))
#endif
#ifndef ALAC_LOAD4
/* four consecutive dwords from a 4-byte aligned address: one global_load_dwordx4 on the GPU */
#define ALAC_LOAD4(q, a, b, c, d) \
    do {                         \
        (a) = (q)[0];            \
        (b) = (q)[1];            \
        (c) = (q)[2];            \
        (d) = (q)[3];            \
    } while (0)
#endif
#ifndef ALAC_PICK
/* dst = src, opaque to the optimiser on the GPU: a chain of these under scalar tests must stay a chain of
 * v_mov (written plainly the compiler turns it into an indexed load from a scratch copy of the array) */
#define ALAC_PICK(dst, src) ((dst) = (src))
#endif
#ifndef ALAC_SIGN
/* -1 / 0 / +1: one v_med3_i32 on the GPU */
#define ALAC_SIGN(x) (((x) > 0) - ((x) < 0))
#endif
#ifndef ALAC_SUBSAT
/* unsigned a - b, 0 when b > a: v_sub_u32 ... clamp on the GPU */
#define ALAC_SUBSAT(a, b) ((a) > (b) ? (a) - (b) : 0u)
#endif
#ifndef ALAC_CLAMP01
/* 0 for x <= 0, 1 for x >= 1: one v_med3_i32 on the GPU */
#define ALAC_CLAMP01(x) ((x) > 0 ? 1 : 0)
#endif
#ifndef ALAC_MAD24
/* a * b + c for 24-bit a, b: one v_mad_i32_i24 on the GPU, opaque to the optimiser (a sum of such products written
 * plainly is re-associated into multiplies and a tree of adds: more instructions for latency nobody is waiting on) */
#define ALAC_MAD24(a, b, c) ((int32_t)(a) * (int32_t)(b) + (int32_t)(c))
#endif
#ifndef ALAC_XAD
/* (a ^ b) + c: one v_xad_u32 on the GPU */
#define ALAC_XAD(a, b, c) ((((uint32_t)(a)) ^ ((uint32_t)(b))) + (uint32_t)(c))
#endif
#ifndef ALAC_MSUB24
/* acc - a * c for 24-bit a and a small wave-uniform c: one v_mad_i32_i24 with -c as its scalar operand on the GPU
 * (written as a product the compiler turns multiplications by 2, 4, 8 into two shifts and a subtraction) */
#define ALAC_MSUB24(acc, a, c) ((int32_t)(acc) - (int32_t)(a) * (int32_t)(c))
#endif
#ifndef ALAC_MULU24
/* exact when both operands fit 24-bit unsigned: v_mul_u32_u24 / v_mad_u32_u24 on the GPU */
#define ALAC_MULU24(a, b) ((uint32_t)(a) * (uint32_t)(b))
#endif

namespace alac {

constexpr uint32_t KEY_IRREGULAR = 1024; /* sort key of packets for decode_wave; regular: numU*32 + numV */
constexpr uint32_t NUM_KEYS = 1025;

/* orders the lean decoder runs: 4/5/6/8 on exactly NA taps, the others (general form, int16 coefficient wrap) on
 * 16 register taps with wave-uniform skips; 0 copies and 31 is delta mode. 17..30 exist only on paper. */
ALAC_DEV bool regular_order(uint32_t na) { return na <= 16 || na == 31; }

/* Sort key of a packet; no entropy decoding, reads only the element header. */
ALAC_DEV uint32_t classify_regular(const DevCfg& cfg, const uint8_t* pkt, uint32_t size) {
    if (cfg.num_channels > 2 || cfg.aligned16 == 0 || cfg.kb == 0 ||
        cfg.frame_length > 65536u || cfg.frame_length <= 32u)
        return KEY_IRREGULAR;
    const Bits bits{pkt, size};
    if (size < 12) return KEY_IRREGULAR;
    const uint32_t tag = bits.get(0, 3);
    const bool cpe = cfg.num_channels == 2;
    if (cpe ? tag != 1 : !(tag == 0 || tag == 3)) return KEY_IRREGULAR;
    if (bits.get(7, 12) != 0) return KEY_IRREGULAR;
    const uint32_t hdr = bits.get(19, 4);
    if (hdr & 1u) return KEY_IRREGULAR; /* escape element */
    const uint32_t bs = (hdr >> 1) & 3u;
    if (bs == 3) return KEY_IRREGULAR;
    /* 24-bit products need chanBits <= 23 (16-bit; 20-bit; 24/32-bit with their usual shift bytes) */
    const uint32_t chan_bits = cfg.bit_depth - 8u * bs + (cpe ? 1u : 0u);
    if (chan_bits < 1 || chan_bits > 23 || cfg.bit_depth < 8u * bs) return KEY_IRREGULAR;
    uint32_t pos = 23;
    uint32_t ns = cfg.frame_length;
    if (hdr >> 3) {
        ns = bits.get(pos, 32);
        pos += 32;
    }
    if (ns == 0 || ns > cfg.frame_length) return KEY_IRREGULAR;
    pos += 16; /* mixBits, mixRes */
    const uint32_t hu = bits.get(pos, 16);
    const uint32_t nu = hu & 0x1fu;
    if ((hu >> 12) != 0 || !regular_order(nu)) return KEY_IRREGULAR;
    pos += 16u + 16u * nu;
    uint32_t nv = 0;
    if (cpe) {
        const uint32_t hv = bits.get(pos, 16);
        nv = hv & 0x1fu;
        if ((hv >> 12) != 0 || !regular_order(nv)) return KEY_IRREGULAR;
        pos += 16u + 16u * nv;
    }
    /* header and shift block must be wholly inside the packet and the entropy stream must start inside it
     * (anything else is an error or panic case: decode_wave reports those) */
    const uint64_t ent = (uint64_t)pos + (uint64_t)bs * 8u * (cpe ? 2u : 1u) * ns;
    if ((ent >> 3) >= size) return KEY_IRREGULAR;
    return nu * 32u + nv;
}

/* ---- the lean path's bit reader: an LDS ring per lane, refilled ahead of time --------------------------------
 * w0,w1 hold stream dwords widx, widx+1 and w2 the next one, as in FastRd, but they are fed from a 32-dword ring
 * in LDS (W::ring_*), never straight from HBM. The ring is topped up 16 bytes at a time on a wave-uniform
 * schedule (every 4th step): tick() first commits the block whose global load was issued 4 steps earlier, then
 * issues the next one. So no step ever waits on an HBM/L2 round trip: the data a step needs left memory at
 * least four steps ago, and each packet byte is fetched from L2 exactly once. A plain step consumes <= 32 bits,
 * so 4 dwords per 4 steps sustain it (reseek() covers the slow path); start() prefills 16 dwords. Loads stay inside size +
 * ALACGPU_PACKET_PAD; positions past size*8 + 66 bits are never decoded here. */
template <class W>
struct RingRd {
    const uint32_t* base; /* packet start rounded down to a dword */
    uint32_t bias;        /* stream bit 0 is bit `bias` of base[0] */
    uint32_t limit;       /* first dword index that may not be loaded */
    uint32_t w0, w1, w2, widx;
    uint32_t fill;        /* ring holds dwords [fill-32, fill); multiple of 4 */
    uint32_t p0, p1, p2, p3;
    bool pend;

    ALAC_DEV void init(const uint8_t* pkt, uint32_t size) {
        /* pointer arithmetic, not an integer round trip: the compiler keeps the global address space */
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(pkt) & 3u);
        base = reinterpret_cast<const uint32_t*>(pkt - mis);
        bias = mis * 8u;
        limit = ((bias >> 3) + size + ALACGPU_PACKET_PAD - 16u) >> 2; /* a 16-byte block starting below it stays inside the pad */
        w0 = w1 = w2 = widx = fill = 0;
        p0 = p1 = p2 = p3 = 0;
        pend = false;
    }
    ALAC_DEV void load4(uint32_t at) {
        /* one 16-byte load, 4-byte aligned. The dwords stay RAW (little-endian) in p0..p3: touching them here
         * would make the wave wait for the load on the spot; commit() swaps them four steps later. */
        ALAC_LOAD4(base + at, p0, p1, p2, p3);
    }
    ALAC_DEV void commit(W& wv) {
        wv.ring_write4(fill & 31u, __builtin_bswap32(p0), __builtin_bswap32(p1), __builtin_bswap32(p2),
                       __builtin_bswap32(p3));
        fill += 4u;
        pend = false;
    }
    /* channel start: synchronous prefill from the block holding `pos` */
    ALAC_DEV void start(W& wv, uint32_t pos) {
        const uint32_t ni = (pos + bias) >> 5;
        fill = ni & ~3u;
        pend = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (fill < limit) {
                load4(fill);
                commit(wv);
            }
        }
        reseek(wv, pos);
    }
    ALAC_DEV void reseek(W& wv, uint32_t pos) {
        widx = (pos + bias) >> 5;
        /* a slow-path step (escape code + zero-run code) can eat more than one dword, more than tick() puts
         * back: top the ring up on the spot whenever it runs low. Plain steps take <= 32 bits (prefix + 1 + k,
         * k <= 23), which the 4 dwords per 4 steps of tick() cover. */
        while (fill < widx + 12u && fill < limit) {
            if (!pend) load4(fill);
            commit(wv);
        }
        w0 = wv.ring_read(widx & 31u);
        w1 = wv.ring_read((widx + 1u) & 31u);
        w2 = wv.ring_read((widx + 2u) & 31u);
    }
    ALAC_DEV uint32_t window(uint32_t pos) const {
        const uint32_t r = (pos + bias) & 31u;
        return (uint32_t)(((((uint64_t)w0) << 32) | w1) << r >> 32);
    }
    /* no branch: the cache moves by 0 or 1 dword (the slow path reseeks), w2 is re-read from LDS every step */
    ALAC_DEV void slide(W& wv, uint32_t pos) {
        const uint32_t ni = (pos + bias) >> 5;
        const bool cross = ni != widx;
        w0 = cross ? w1 : w0;
        w1 = cross ? w2 : w1;
        widx = ni;
        w2 = wv.ring_read((ni + 2u) & 31u);
#ifdef ALAC_RING_DEBUG
        if (ni + 2u < limit && w2 != __builtin_bswap32(base[ni + 2u]))
            fprintf(stderr, "ring mismatch at dword %u (fill %u widx %u limit %u pend %d)\n", ni + 2u, fill, widx, limit, (int)pend);
#endif
    }
    /* slide() without a compare: the move (0 or 1 dwords) becomes a bit mask */
    ALAC_DEV void slide_mask(W& wv, uint32_t pos) {
        const uint32_t ni = (pos + bias) >> 5;
        const uint32_t cm = 0u - (ni - widx);
        w0 = (w1 & cm) | (w0 & ~cm);
        w1 = (w2 & cm) | (w1 & ~cm);
        widx = ni;
        w2 = wv.ring_read((ni + 2u) & 31u);
    }
    /* every 4th step, wave-uniform */
    ALAC_DEV void tick(W& wv) {
        if (pend) commit(wv);
        if (fill + 4u <= widx + 32u && fill < limit) {
            load4(fill);
            pend = true;
        }
    }
};

/* per-lane Golomb + reader state of one channel */
template <class W>
struct RegLane {
    RingRd<W> rd;
    uint32_t pos, mean, zmode, zrem, pb, max_pos;
    int32_t err;
};

/* The rare part of DynDecomp (golomb.go:167-247) for one lane: overrun, an escape code, and/or the start of a
 * zero run. Redoes the sample from its start with the stateless reader; returns the residual. Works on local
 * copies and writes the lane state back once (stores into the state from several exits make the compiler keep
 * it in scratch memory). */
template <class W>
ALAC_DEV int32_t golomb_slow(const Bits& bits, RegLane<W>& s, uint32_t size, uint32_t kb, uint32_t wb,
                             uint32_t chan_bits, uint32_t i, uint32_t ns) {
    uint32_t pos = s.pos, mean = s.mean, zmode = s.zmode, zrem = s.zrem;
    int32_t err = 0, del = 0;
    if (pos >= s.max_pos) {
        err = ST_OVERRUN; /* golomb.go:168-170 */
    } else {
        uint32_t m = mean >> 9;
        const uint32_t k = umin(31u - clz32(m + 3u), kb);
        m = (1u << k) - 1u;
        const uint32_t w = (uint32_t)(bits.window(pos) >> 32);
        uint32_t n = clz32(~w);
        if (n >= 9) { /* getStreamBits(bitPos+9, maxSize), golomb.go:184-186,86-108 */
            const uint32_t gpos = pos + 9u;
            const uint32_t gb = gpos & 7u;
            const bool five = chan_bits + gb > 32u;
            if ((gpos >> 3) > size || (five && (gpos >> 3) >= size)) err = ST_MALFORMED;
            const uint64_t w2 = bits.window(gpos);
        if (chan_bits == 0) n = 0;
        else if (chan_bits <= 32) n = (uint32_t)(w2 >> (64u - chan_bits));
        else n = (uint32_t)(w2 >> 31) & ((2u << gb) - 1u); /* numBits 33: only byte 5 survives (golomb.go:90-99) */
            pos += 9u + chan_bits;
        } else {
            const uint32_t v = (w << (n + 1u)) >> (32u - k);
            pos += n + 1u + k - (v >= 2 ? 0u : 1u);
            n = v >= 2 ? n * m + v - 1u : n * m;
        }
        if (err == 0) {
            const uint32_t nd = n + zmode;
            const int32_t half = (int32_t)((nd + 1u) >> 1);
            del = (nd & 1u) ? -half : half;
            mean = s.pb * nd + mean - ((s.pb * mean) >> 9);
            if (n > 0xffffu) mean = 0xffffu;
            zmode = 0;
            if ((mean << 2) < 512u && i + 1u < ns) { /* golomb.go:223-246 */
                zmode = 1;
                int32_t k32 = (int32_t)clz32(mean) - 24 + (int32_t)((mean + 16u) >> 6);
                if (k32 < 0) k32 = 0;
                const uint32_t kz = (uint32_t)k32;
                const uint32_t mz = ((1u << kz) - 1u) & wb;
                if ((pos >> 3) > size) { /* dynGet's read32bit, golomb.go:115 */
                    err = ST_MALFORMED;
                } else {
                    const uint32_t wz = (uint32_t)(bits.window(pos) >> 32);
                    const uint32_t pre = clz32(~wz);
                    uint32_t rl;
                    if (pre >= 9) {
                        rl = (wz << 9) >> 16;
                        pos += 25u;
                    } else {
                        const uint32_t val = kz == 0 ? 0u : (wz << (pre + 1u)) >> (32u - kz);
                        pos += pre + kz + (val < 2 ? 0u : 1u);
                        rl = val < 2 ? pre * mz : pre * mz + val - 1u;
                    }
                    if ((uint64_t)i + 1u + rl > ns) err = ST_SAMPLE_OVERRUN; /* golomb.go:232-234 */
                    zrem = rl;
                    if (rl >= 65535u) zmode = 0;
                    mean = 0;
                }
            }
        }
    }
    const bool ok = err == 0;
    s.pos = ok ? pos : s.pos;
    s.mean = ok ? mean : s.mean;
    s.zmode = ok ? zmode : s.zmode;
    s.zrem = ok ? zrem : s.zrem;
    s.err = err;
    return del;
}

/*
 * One residual (DynDecomp, golomb.go:167-247), the form the entropy wave of alac_duo.h runs: a lone wave pays for
 * EVERY instruction it issues (scalar ones and branches included, ~4-5 cycles each) and a v_cmp / v_cndmask pair
 * costs ~17, so the step is written as plain integer arithmetic on bit masks: no compare, no select and no
 * exec-mask juggling on the common path; the one branch left is the rare part (golomb_slow).
 * ns_live: the lane's sample count, 0 once the lane has failed (i >= ns_live: a dead step, nothing moves).
 * on_mask: ~0 when i < ns_live, carried from the previous step (updated here for step i+1).
 */
template <class W>
ALAC_DEV int32_t gol_step(W& wv, const Bits& bits, RegLane<W>& s, uint32_t size, uint32_t kb, uint32_t wb,
                          uint32_t chan_bits, uint32_t i, uint32_t ns, uint32_t& ns_live, uint32_t& on_mask) {
    const uint32_t k = umin(31u - clz32((s.mean >> 9) + 3u), kb); /* 1..23 */
    const uint32_t w = s.rd.window(s.pos);
    const uint32_t pre = clz32(~w);
    const uint32_t v = (w << ((pre + 1u) & 31u)) >> (32u - k);
    const uint32_t vm1 = ALAC_SUBSAT(v, 1u);           /* v >= 2: value v - 1 and k bits; else 0 and k - 1 bits */
    const uint32_t n = (pre << k) - pre + vm1;          /* pre * (2^k - 1) + ... */
    const uint32_t nd = n + s.zmode;
    const uint32_t mean2 = s.pb * nd + s.mean - ((s.pb * s.mean) >> 9); /* golomb.go:215 */
    /* masks: lane decodes a code in this step (alive, not inside a zero run) */
    const uint32_t next_on = (uint32_t)((int32_t)(i + 1u - ns_live) >> 31); /* i + 1 < ns_live (both < 2^31) */
    const uint32_t norun = (uint32_t)((int32_t)(s.zrem - 1u) >> 31);        /* zrem == 0 (zrem <= 65535) */
    const uint32_t okm = on_mask & norun;
    /* rare cases, as nonzero-means-true flags: overrun (golomb.go:168), escape code (:184), n > 0xffff (:216),
     * start of a zero run (:223: mean * 4 < 512 with a sample left; mean2 < 2^25 here, the shift cannot wrap) */
    const uint32_t rare = (ALAC_SUBSAT(s.pos + 1u, s.max_pos) | ((pre + 7u) >> 4) | (n >> 16) |
                           (ALAC_SUBSAT(128u, mean2) & next_on)) & okm;
    const uint32_t hm = (nd + 1u) >> 1; /* golomb.go:206-209 */
    const uint32_t sg = 0u - (nd & 1u);
    int32_t del = (int32_t)(((hm ^ sg) - sg) & norun);
    const uint32_t o_pos = s.pos, o_mean = s.mean, o_zmode = s.zmode, o_zrem = s.zrem;
    s.pos = o_pos + ((pre + k + umin(vm1, 1u)) & okm); /* prefix + 1, then k bits (v >= 2) or k - 1 */
    s.mean = (mean2 & okm) | (o_mean & ~okm);
    s.zmode = o_zmode & ~okm;
    s.zrem = ALAC_SUBSAT(o_zrem, 1u);
    on_mask = next_on;
    if (wv.any(rare != 0u)) {
        if (rare != 0u) {
            s.pos = o_pos;
            s.mean = o_mean;
            s.zmode = o_zmode;
            s.zrem = o_zrem;
            del = golomb_slow(bits, s, size, kb, wb, chan_bits, i, ns);
            s.rd.reseek(wv, s.pos);
            ns_live = s.err ? 0u : ns_live;
            on_mask = (uint32_t)((int32_t)(i + 1u - ns_live) >> 31);
        }
    }
    s.rd.slide_mask(wv, s.pos);
    return del;
}

/*
 * One channel of a regular element, all lanes in lock step. NA = this channel's predictor order (wave-uniform).
 * LAST: this channel completes the frame (V of a pair, or the mono channel): unmix and emit PCM.
 */
/* tentative result of the branch-free part of one Golomb sample (see regular_phase) */
struct GolTent {
    uint32_t pos2, mean2;
    int32_t del;
    bool dec, slow, on, inrun;
};

/* ---- one residual (DynDecomp, golomb.go:167-247) in two halves ------------------------------------------------
 * gol_tentative(): pure ALU, no branch, no state change. gol_commit(): the one rare branch (escape code, start of
 * a zero run, overrun), then the state update by selects and the reader slide. i = sample index, ns = the lane's
 * sample count (i >= ns: a dead step). */
template <class W>
ALAC_DEV void gol_tentative(const RegLane<W>& s, uint32_t kb, uint32_t i, uint32_t ns, GolTent& t) {
    t.on = i < ns && s.err == 0;
    t.inrun = s.zrem != 0;
    t.dec = t.on && !t.inrun;
    uint32_t m = s.mean >> 9;
    const uint32_t k = umin(31u - clz32(m + 3u), kb);
    m = (1u << k) - 1u;
    const uint32_t w = s.rd.window(s.pos);
    const uint32_t pre = clz32(~w);
    const uint32_t v = (w << ((pre + 1u) & 31u)) >> (32u - k);
    const bool big = v >= 2;
    const uint32_t n = pre * m + (big ? v - 1u : 0u);
    const uint32_t nd = n + s.zmode;
    t.mean2 = s.pb * nd + s.mean - ((s.pb * s.mean) >> 9); /* golomb.go:215 */
    if (n > 0xffffu) t.mean2 = 0xffffu;
    t.slow = t.dec && (s.pos >= s.max_pos || pre >= 9 || ((t.mean2 << 2) < 512u && i + 1u < ns));
    const int32_t half = (int32_t)((nd + 1u) >> 1); /* golomb.go:206-209 */
    t.del = t.inrun ? 0 : ((nd & 1u) ? -half : half);
    t.pos2 = s.pos + pre + k + (big ? 1u : 0u); /* prefix + 1, then k bits (v >= 2) or k - 1 */
}
template <class W>
ALAC_DEV int32_t gol_commit(W& wv, const Bits& bits, RegLane<W>& s, uint32_t size, uint32_t kb, uint32_t wb,
                            uint32_t chan_bits, uint32_t i, uint32_t ns, const GolTent& t) {
    int32_t del = t.del;
    if (wv.any(t.slow)) {
        if (t.slow) {
            del = golomb_slow(bits, s, size, kb, wb, chan_bits, i, ns);
            s.rd.reseek(wv, s.pos);
        }
    }
    const bool ok = t.dec && !t.slow;
    s.pos = ok ? t.pos2 : s.pos;
    s.mean = ok ? t.mean2 : s.mean;
    s.zmode = ok ? 0u : s.zmode;
    s.zrem = (t.on && t.inrun) ? s.zrem - 1u : s.zrem;
    s.rd.slide(wv, s.pos);
    return del;
}

/* ---- predictor step for i > na (UnpcBlock, predictor.go:99-684), chanBits <= 23 ---------------------------------
 * hb[j] = out[i-1-j] ^ BIAS (sign-biased history: |a - b| of biased values is one unsigned sad). Taps walked from
 * the highest down. The adaptation is sign-normalised: D0 = |del| shrinks by t_j = (na-j) * ((|d_j| + rnd) >>
 * denShift) tap after tap and tap j adapts while something of D0 is left. chanBits <= 23 keeps q < 2^23 and
 * t_j < 2^27: nothing wraps, which is what makes this equal to the reference's signed countdown.
 * GEN: the wave-uniform order na on NR = 16 register taps; WRAP: int16 coefficients (predictor.go:664,675). */
template <int NR, bool GEN, bool WRAP, bool CB_POS = false>
ALAC_DEV int32_t predict_narrow(int32_t (&coef)[NR], const uint32_t (&hb)[NR + 1], uint32_t na, int32_t del,
                                uint32_t den_shift, int32_t den_half, uint32_t rnd_neg, uint32_t chan_shift) {
    constexpr uint32_t BIAS = 0x80000000u;
    uint32_t topb = hb[NR];
    if (GEN) {
#pragma unroll
        for (int j = 1; j < NR; ++j)
            if (na == (uint32_t)j) ALAC_PICK(topb, hb[j]); /* scalar branch: na is wave-uniform */
    }
    /* no compares on the hot path (a v_cmp / v_cndmask pair costs a lone wave ~17 cycles, plain ALU ops ~5):
     * everything that depends on the sign of the residual is derived from its sign mask */
    const uint32_t sgnm = (uint32_t)(del >> 31); /* ~0 for del < 0 */
    const uint32_t nsg = (uint32_t)del >> 31;    /* 1 for del < 0 */
    const uint32_t rnd = rnd_neg & sgnm;
    /* D0 = |del|: what is left of it after the taps above. Signed and never wrapping: the taps take at most
     * sum(na - j) * 2^23 = 136 * 2^23 < 2^31 away from a value >= 0 */
    int32_t rem = (int32_t)(((uint32_t)del ^ sgnm) + nsg);
    /* den_half - sum coef_j * (top - h_j), as one multiply-add chain over e_j = h_j - top */
    int32_t acc = den_half;
#pragma unroll
    for (int j = NR - 1; j >= 0; --j) {
        if (GEN && (uint32_t)j >= na) continue; /* scalar branch: taps the order does not have */
        const int32_t e = (int32_t)(hb[j] - topb); /* out[i-1-j] - top; the bias cancels */
        acc = ALAC_MAD24(coef[j], e, acc);         /* uses coef[j] before its update */
        /* coefficient step sign(del) * -sign(top - h_j) = sign(del) * sign(e): (sign(e) ^ sgnm) + nsg */
        const int32_t delta = (int32_t)ALAC_XAD(ALAC_SIGN(e), sgnm, nsg);
        const uint32_t q = ALAC_SAD(topb, hb[j], rnd) >> den_shift;
        const int32_t go = ALAC_CLAMP01(rem); /* tap j adapts while the budget is not used up */
        const int32_t cj = ALAC_MAD24(delta, go, coef[j]);
        coef[j] = WRAP ? (int32_t)(int16_t)cj : cj; /* predictor.go:664,675 */
        rem = ALAC_MSUB24(rem, q, na - (uint32_t)j);
    }
    const int32_t o = del + (int32_t)(topb ^ BIAS) + (acc >> den_shift);
    /* CB_POS: the caller knows chanBits >= 1, so the shift count is <= 31 and sext_cs' guard for 32 is not needed */
    return CB_POS ? (int32_t)((uint32_t)o << chan_shift) >> chan_shift : sext_cs(o, chan_shift);
}

/* what a phase does with the reconstructed samples */
enum { OUT_UTILE = 0,  /* U of a pair: hand-off tile */
       OUT_STEREO = 1, /* V of a pair: unmix with U, PCM */
       OUT_MONO = 2,   /* single channel: PCM */
       OUT_RAW = 3,    /* int32 samples into this lane's row (split pipeline, alac_split.h) */
       OUT_NONE = 4 }; /* nothing: entropy scan only */

template <class W, int NA, int OUT, bool NARROW>
ALAC_DEV void regular_phase(W& wv, const DevCfg& cfg, const Bits& bits, RegLane<W>& s, uint32_t size, uint32_t ns,
                            uint32_t n_it, uint32_t hdr_pos, uint32_t den_shift, uint32_t chan_bits, int32_t mix_res,
                            uint32_t mix_sh, uint32_t na_rt, uint32_t shift_pos, uint32_t sb, uint32_t mode) {
    constexpr bool LAST = OUT == OUT_STEREO || OUT == OUT_MONO;
    constexpr bool CPE = OUT == OUT_STEREO;
    constexpr bool RAW = OUT == OUT_RAW;
    constexpr bool SCAN = OUT == OUT_NONE;
    /* NARROW: chanBits <= 23, so every product fits the 24-bit multipliers and nothing in the adaptation can
     * wrap; otherwise plain 32-bit arithmetic in the reference's literal form. mode != 0 (per lane): the delta
     * pre-pass of decoder.go:307-309 runs on the residual stream first.
     * shift_pos / sb: start of the shift-byte block and bits per value to merge (0 = none), LAST only.
     * NA != 0: exactly NA taps, int32 coefficients (unpcBlock4/5/6/8). NA == 0: the general form for the
     * wave-uniform order na_rt (0..16, 31) on NR = 16 register taps, coefficients wrapped to int16. */
    constexpr bool GEN = NA == 0;
    constexpr int NR = GEN ? 16 : NA;
    /* orders other than 4/5/6/8 run unpcBlockGeneral: int16 coefficients (predictor.go:81-93) */
    constexpr bool WRAP = !(NA == 4 || NA == 5 || NA == 6 || NA == 8);
    const uint32_t na = GEN ? na_rt : (uint32_t)NA;
    constexpr uint32_t BIAS = 0x80000000u;
    const uint32_t kb = cfg.kb;
    const uint32_t wb = (1u << kb) - 1u;
    const uint32_t chan_shift = 32u - chan_bits;
    const int32_t den_half = den_shift ? (int32_t)(1u << (den_shift - 1u)) : 0;
    const uint32_t rnd_neg = (1u << den_shift) - 1u;

    int32_t coef[NR];
    uint32_t hb[NR + 1]; /* hb[j] = out[i-1-j] ^ BIAS: |a - b| of biased values is one unsigned sad */
#pragma unroll
    for (int j = 0; j < NR; ++j)
        coef[j] = (!GEN || ((uint32_t)j < na && na != 31)) ? (int32_t)(int16_t)bits.get(hdr_pos + 16u + 16u * (uint32_t)j, 16) : 0;
#pragma unroll
    for (int j = 0; j <= NR; ++j) hb[j] = BIAS;
    uint64_t pk_acc = 0; /* little-endian byte packer: whole dwords go to the stager */
    uint32_t pk_n = 0;
    const uint32_t bps = cfg.bps;
    const uint64_t pk_msk = bps == 4 ? 0xffffffffull : ((1ull << (8u * bps)) - 1ull);
    const bool merge_any = LAST && wv.any(sb != 0);
    int32_t u_next = 0;
    if (LAST && CPE) u_next = *wv.u_row(0);
    int32_t dprev = 0;
    const bool mode_any = !SCAN && wv.any(mode != 0);
    /* decoder.go:307-309: UnpcBlock(numActive 31, denShift 0) over the residuals before the coefficient pass */
    auto prepass = [&](uint32_t idx, int32_t del) -> int32_t {
        if (!mode_any) return del;
        const int32_t dd = idx == 0 ? del : sext_cs(del + dprev, chan_shift);
        dprev = mode != 0 ? dd : dprev;
        return mode != 0 ? dd : del;
    };

    /* one residual in two halves (gol_tentative / gol_commit above): the main loop puts the tentative half of
     * sample i+1 in the same basic block as the predictor taps of sample i */
    auto tentative = [&](uint32_t i, GolTent& t) { gol_tentative(s, kb, i, ns, t); };
    auto commit = [&](uint32_t i, GolTent& t) -> int32_t { return gol_commit(wv, bits, s, size, kb, wb, chan_bits, i, ns, t); };
    /* ---- predictor step for i > na (UnpcBlock, predictor.go:99-684) -------------------------------------------
     * Taps walked from the highest down. The adaptation is sign-normalised: D0 = |del| shrinks by
     * t_j = (na-j) * ((|d_j| + rnd) >> denShift) tap after tap and tap j adapts while the running total of the
     * taps above it is still below D0. */
    auto predict = [&](int32_t del) -> int32_t {
        if (!NARROW) {
            /* literal form of predictor.go:99-684 on 32-bit arithmetic */
            int32_t top = (int32_t)(hb[NR] ^ BIAS);
            if (GEN) {
#pragma unroll
                for (int j = 1; j < NR; ++j)
                    if (na == (uint32_t)j) ALAC_PICK(top, (int32_t)(hb[j] ^ BIAS));
            }
            int32_t d[NR];
            int32_t acc = den_half;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                d[j] = top - (int32_t)(hb[j] ^ BIAS);
                if (!GEN || (uint32_t)j < na) acc -= coef[j] * d[j];
            }
            const int32_t o = sext_cs(del + top + (acc >> den_shift), chan_shift);
            if (del != 0) {
                const int32_t sg = del > 0 ? 1 : -1;
                int32_t del0 = del;
                bool go = true;
#pragma unroll
                for (int j = NR - 1; j >= 0; --j) {
                    if (GEN && (uint32_t)j >= na) continue;
                    const int32_t sgn = sg > 0 ? sign_of(d[j]) : -sign_of(d[j]);
                    int32_t cj = coef[j] - sgn;
                    if (WRAP) cj = (int32_t)(int16_t)cj;
                    coef[j] = go ? cj : coef[j];
                    del0 -= go ? (int32_t)(na - (uint32_t)j) * ((sgn * d[j]) >> den_shift) : 0;
                    go = go && (sg > 0 ? del0 > 0 : del0 < 0);
                }
            }
            return o;
        }
        return predict_narrow<NR, GEN, WRAP>(coef, hb, na, del, den_shift, den_half, rnd_neg, chan_shift);
    };
    /* ---- history, hand-off / unmix / PCM of sample i ------------------------------------------------------------ */
    auto emit = [&](uint32_t i, int32_t o, bool on, int32_t u_pre, uint32_t sh_l, uint32_t sh_r) {
#pragma unroll
        for (int j = NR; j >= 1; --j) hb[j] = hb[j - 1];
        hb[0] = (uint32_t)o ^ BIAS;
        if (RAW) {
            wv.st_push_if((uint32_t)o, on); /* one int32 sample per step into the lane's row */
            return;
        }
        if (!LAST) {
#ifndef ALAC_EXP_NO_U_STORE
            *wv.u_row(i) = o; /* dead lanes write their own unused cell */
#else
            asm volatile("" ::"v"(o));
#endif
            return;
        }
        int32_t l = o, r = 0;
        if (CPE) {
            const int32_t u = u_pre, vv = o;
            if (mix_res != 0) { /* matrix.go:40-41 */
                l = u + vv - (ALAC_MUL24(mix_res, vv) >> mix_sh);
                r = l - vv;
            } else {
                l = u;
                r = vv;
            }
        }
        if (cfg.bit_depth == 20) { /* matrix.go:77-78, 237 */
            l = (int32_t)((uint32_t)l << 4);
            r = (int32_t)((uint32_t)r << 4);
        }
        if (merge_any) { /* matrix.go:129-132, 266-268: (x << 8*bytesShifted) | shift value */
            l = (int32_t)((uint32_t)l << sb) | (int32_t)sh_l;
            r = (int32_t)((uint32_t)r << sb) | (int32_t)sh_r;
        }
        if (bps == 2 && CPE) {
            wv.st_push_if(((uint32_t)l & 0xffffu) | ((uint32_t)r << 16), on);
        } else {
            /* generic widths: append bps bytes per sample, emit a dword whenever four are ready (all selects) */
            pk_acc |= ((uint64_t)(uint32_t)l & pk_msk) << (8u * pk_n);
            pk_n += on ? bps : 0u;
            bool em = pk_n >= 4u;
            wv.st_push_if((uint32_t)pk_acc, em);
            pk_acc = em ? pk_acc >> 32 : pk_acc;
            pk_n = em ? pk_n - 4u : pk_n;
            if (CPE) {
                pk_acc |= ((uint64_t)(uint32_t)r & pk_msk) << (8u * pk_n);
                pk_n += on ? bps : 0u;
                em = pk_n >= 4u;
                wv.st_push_if((uint32_t)pk_acc, em);
                pk_acc = em ? pk_acc >> 32 : pk_acc;
                pk_n = em ? pk_n - 4u : pk_n;
            }
            /* a lane that is not `on` appended nothing: clear what the OR left above its valid bytes */
            pk_acc &= pk_n ? ((1ull << (8u * pk_n)) - 1ull) : 0ull;
        }
    };
    /* per-step memory prefetches of sample i: shift values (24/32-bit) and the U hand-off, one step ahead */
    auto fetch = [&](uint32_t i, int32_t& u_pre, uint32_t& sh_l, uint32_t& sh_r) {
        sh_l = sh_r = 0;
        if (merge_any) {
            /* both shift values of the frame sit side by side (decoder.go:492-502): one window */
            const uint64_t sw = bits.window(shift_pos + i * (CPE ? 2u : 1u) * sb);
            sh_l = sb ? (uint32_t)(sw >> (64u - sb)) : 0u;
            sh_r = (CPE && sb) ? (uint32_t)((sw << sb) >> (64u - sb)) : 0u;
        }
        u_pre = 0;
        if (LAST && CPE) {
            /* row n_it <= frame_length exists: the tile ends in spare cells */
            u_pre = u_next;
#ifndef ALAC_EXP_NO_U_LOAD
            u_next = *wv.u_row(i + 1u);
#else
            u_next = (int32_t)i;
#endif
        }
    };

    /* ---- head: out[0] = pc1[0], warm-up (predictor.go:53-79); also the whole block for copy / delta mode ------ */
    const bool simple_all = SCAN || (GEN && (na == 0 || na == 31));
    const uint32_t head = simple_all ? n_it : umin(na + 1u, n_it);
    uint32_t i = 0;
    for (; i < head; ++i) {
        if ((i & 3u) == 0) s.rd.tick(wv); /* scalar test: bitstream ring refill, 4 steps ahead of need */
        int32_t u_pre;
        uint32_t sh_l, sh_r;
        fetch(i, u_pre, sh_l, sh_r);
        GolTent t;
        tentative(i, t);
        int32_t del = commit(i, t);
        if (SCAN) continue; /* entropy scan: only the position matters */
        del = prepass(i, del);
        const int32_t o = (i == 0 || (GEN && na == 0)) ? del : sext_cs(del + (int32_t)(hb[0] ^ BIAS), chan_shift);
        emit(i, o, t.on, u_pre, sh_l, sh_r);
#ifndef ALAC_EXP_NO_FLUSH
        if (LAST || RAW) wv.st_step(); /* collective */
#endif
    }
    /* ---- main loop, software-pipelined by one sample: predictor of sample i with the Golomb code of i+1 -------- */
    if (i < n_it) {
        GolTent t;
        tentative(i, t);
        int32_t del = prepass(i, commit(i, t));
        bool on = t.on;
        for (; i < n_it; ++i) {
            if ((i & 3u) == 0) s.rd.tick(wv);
            int32_t u_pre;
            uint32_t sh_l, sh_r;
            fetch(i, u_pre, sh_l, sh_r);
            GolTent tn;
            tentative(i + 1u, tn);         /* chain 1: entropy code of the next sample (i + 1 >= ns: a dead step) */
            const int32_t o = predict(del); /* chain 2: taps of this sample */
            emit(i, o, on, u_pre, sh_l, sh_r);
            del = prepass(i + 1u, commit(i + 1u, tn));
            on = tn.on;
#ifndef ALAC_EXP_NO_FLUSH
            if (LAST || RAW) wv.st_step(); /* collective */
#endif
        }
    }
    if (LAST) wv.st_tail(pk_acc, s.err == 0 ? pk_n : 0u); /* bytes of the last, incomplete dword */
}

/* the order switch is scalar: NA is wave-uniform by construction of the waves */
template <class W, int OUT, bool NARROW>
ALAC_DEV void regular_phase_na(W& wv, uint32_t na, const DevCfg& cfg, const Bits& bits, RegLane<W>& s, uint32_t size,
                               uint32_t ns, uint32_t n_it, uint32_t hdr_pos, uint32_t den_shift, uint32_t chan_bits,
                               int32_t mix_res, uint32_t mix_sh, uint32_t shift_pos, uint32_t sb, uint32_t mode) {
    /* one instantiation per order 1..16 (exact tap count, no skips); 0 (copy) and 31 (delta) share the general one */
    switch (na) {
        case 1: regular_phase<W, 1, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 2: regular_phase<W, 2, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 3: regular_phase<W, 3, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 4: regular_phase<W, 4, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 5: regular_phase<W, 5, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 6: regular_phase<W, 6, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 7: regular_phase<W, 7, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 8: regular_phase<W, 8, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 9: regular_phase<W, 9, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 10: regular_phase<W, 10, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 11: regular_phase<W, 11, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 12: regular_phase<W, 12, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 13: regular_phase<W, 13, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 14: regular_phase<W, 14, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 15: regular_phase<W, 15, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        case 16: regular_phase<W, 16, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
        default: regular_phase<W, 0, OUT, NARROW>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode); break;
    }
}

/* Entropy scan of one channel for decode_wave<..., SCAN>: the lean Golomb loop with nothing behind it. */
template <class W>
ALAC_DEV void scan_channel(W& wv, const DevCfg& cfg, const Bits& bits, const uint8_t* pkt, uint32_t size, bool go,
                           uint32_t& pos, uint32_t ns, uint32_t pb_local, uint32_t chan_bits, int32_t& err) {
    RegLane<W> s;
    s.rd.init(pkt, size);
    s.err = 0;
    s.max_pos = size * 8u;
    s.pos = go ? pos : 0u;
    s.mean = cfg.mb;
    s.zmode = 0;
    s.zrem = 0;
    s.pb = pb_local;
    const uint32_t my_ns = go ? ns : 0u;
    const uint32_t n_it = wv.max_u32(my_ns);
    s.rd.start(wv, s.pos);
    regular_phase<W, 0, OUT_NONE, true>(wv, cfg, bits, s, size, my_ns, n_it, 0u, 0u, chan_bits, 0, 0u, 0u, 0u, 0u, 0u);
    if (go) {
        pos = s.pos;
        err = s.err;
    }
}

/*
 * decode_regular: every lane of the wave holds a regular packet with the same key = numU*32 + numV (lanes
 * without a packet pass live = false). Same contract as decode_wave.
 */
template <class W>
ALAC_DEV int32_t decode_regular(W& wv, const DevCfg& cfg, uint32_t key, bool live, const uint8_t* pkt, uint32_t size,
                                uint8_t* out, uint32_t* frames_out) {
    const Bits bits{pkt, size};
    const bool cpe = cfg.num_channels == 2;
    const uint32_t na_u = key >> 5, na_v = key & 31u;

    RegLane<W> s;
    s.rd.init(pkt, size);
    s.err = 0;
    s.max_pos = size * 8u;

    /* header (accepted by classify_regular, so no error can arise here): decoder.go:213-235, 421-450 */
    uint32_t pos = 23;
    uint32_t ns = 0;
    if (live) {
        ns = cfg.frame_length;
        if (bits.get(19, 4) >> 3) {
            ns = bits.get(pos, 32);
            pos += 32;
        }
    }
    const int32_t mix_bits = (int32_t)bits.get(pos, 8);
    const int32_t mix_res = (int32_t)(int8_t)bits.get(pos + 8, 8);
    const uint32_t mix_sh = (uint32_t)mix_bits > 31u ? 31u : (uint32_t)mix_bits;
    const uint32_t hdr_u = pos + 16u;
    const uint32_t hdr_v = hdr_u + 16u + 16u * na_u;
    const uint32_t hu = bits.get(hdr_u, 16);
    const uint32_t hv = bits.get(hdr_v, 16);
    const uint32_t bs = (bits.get(19, 4) >> 1) & 3u;
    const uint32_t shift_pos = cpe ? hdr_v + 16u + 16u * na_v : hdr_v; /* decoder.go:289-293, 453-457 */
    s.pos = shift_pos + bs * 8u * (cpe ? 2u : 1u) * ns;
    const uint32_t chan_bits = cfg.bit_depth - 8u * bs + (cpe ? 1u : 0u);
    /* the 16- and 20-bit writers ignore the shift buffer (matrix.go:30,66) */
    const uint32_t sb = (cfg.bit_depth == 24 || cfg.bit_depth == 32) ? bs * 8u : 0u;
    const uint32_t n_it = wv.max_u32(ns);
    if (live) wv.st_begin(out);

    /* ---- U (or the mono channel) ---- */
    s.mean = cfg.mb;
    s.zmode = 0;
    s.zrem = 0;
    s.pb = (cfg.pb * ((hu >> 5) & 7u)) / 4u; /* decoder.go:299 */
    s.rd.start(wv, live ? s.pos : 0u);
    if (cpe) regular_phase_na<W, OUT_UTILE, true>(wv, na_u, cfg, bits, s, size, ns, n_it, hdr_u, (hu >> 8) & 0xfu, chan_bits, mix_res, mix_sh, shift_pos, 0u, 0u);
    else regular_phase_na<W, OUT_MONO, true>(wv, na_u, cfg, bits, s, size, ns, n_it, hdr_u, (hu >> 8) & 0xfu, chan_bits, 0, 0, shift_pos, sb, 0u);
    uint32_t err_chan = 0;
    /* ---- V ---- */
    if (cpe) {
        const bool u_failed = s.err != 0;
        if (!u_failed && ((s.pos >> 3) > size + 4u || (s.pos >> 3) > size)) s.err = ST_MALFORMED; /* DynDecomp entry */
        const int32_t err_u = s.err;
        s.mean = cfg.mb;
        s.zmode = 0;
        s.zrem = 0;
        s.pb = (cfg.pb * ((hv >> 5) & 7u)) / 4u;
        s.rd.start(wv, (live && s.err == 0) ? s.pos : 0u);
        regular_phase_na<W, OUT_STEREO, true>(wv, na_v, cfg, bits, s, size, ns, n_it, hdr_v, (hv >> 8) & 0xfu, chan_bits, mix_res, mix_sh, shift_pos, sb, 0u);
        if (err_u == 0 && s.err != 0) err_chan = 1;
    }
    if (!live) return 0;
    (void)wv.st_finish();
    if (s.err) {
        *frames_out = 0;
        if (s.err == ST_MALFORMED) return ST_MALFORMED;
        const uint32_t stage = cpe ? (uint32_t)(err_chan == 0 ? ALACGPU_STAGE_ENTROPY_U : ALACGPU_STAGE_ENTROPY_V)
                                   : (uint32_t)ALACGPU_STAGE_ENTROPY;
        return ALACGPU_STATUS(s.err, cpe ? ALACGPU_CTX_CPE : ALACGPU_CTX_SCE, stage);
    }
    *frames_out = ns;
    return 0;
}

} /* namespace alac */
#endif
