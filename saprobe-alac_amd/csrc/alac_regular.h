/*
 * alac_regular.h — building blocks of the lean decoders: the packet classifier, the LDS-ring bit reader, the
 * Golomb/Rice step and the adaptive predictor step.
 *
 * A packet is REGULAR when classify_regular() accepts it: mono or stereo, its first tag is the one element that
 * covers the whole frame (SCE/LFE for 1 channel, CPE for 2), compressed, any legal chanBits (1..23 on 24-bit
 * multiply-adds, 24..33 on the literal 32-bit form), mode 0 on both channels, predictor orders 0..16 or 31, header
 * inside the packet.
 * Regular packets are decoded by a pair of wavefronts per 64 packets (alac_duo.h); everything else is scanned
 * first (decode_wave<SCAN>, alac_wave.h) and finished by the split pipeline (alac_split.h) or the whole-packet
 * decoder. All of them produce identical bytes and status words; which one runs is a speed choice made per packet
 * by the classifier, and waves are built from packets with the same (numU, numV).
 *
 * What "lean" buys: with the orders wave-uniform the predictor is compiled for exactly NA taps
 * and the warm-up test is scalar; the per-sample steps are plain integer arithmetic on bit masks (no compare, no
 * select: a v_cmp / v_cndmask pair costs a lone wave ~17 cycles, an add ~5); only an escape code, the start of a
 * zero run or an error take the one slow branch; |diff| + rounding is one v_sad_u32 on a sign-biased history;
 * products are 24-bit multiply-adds, exact because chanBits <= 23.
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
#ifndef ALAC_STORE4
/* four consecutive dwords to a 16-byte aligned address: one global_store_dwordx4 on the GPU */
#define ALAC_STORE4(q, a, b, c, d) \
    do {                          \
        (q)[0] = (a);             \
        (q)[1] = (b);             \
        (q)[2] = (c);             \
        (q)[3] = (d);             \
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
#ifndef ALAC_BFI
/* bitwise select (a & m) | (b & ~m): one v_bfi_b32 on the GPU, opaque to the optimiser (it knows the masks here are
 * all-ones or zero and turns the expression back into a compare and a select) */
#define ALAC_BFI(m, a, b) ((((uint32_t)(a)) & (uint32_t)(m)) | (((uint32_t)(b)) & ~(uint32_t)(m)))
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
#ifndef ALAC_OWN_REG
/* gives a wave-uniform value a register of its own (opaque copy) on the GPU */
#define ALAC_OWN_REG(x) ((void)0)
#endif
#ifndef ALAC_MULU24
/* exact when both operands fit 24-bit unsigned: v_mul_u32_u24 / v_mad_u32_u24 on the GPU */
#define ALAC_MULU24(a, b) ((uint32_t)(a) * (uint32_t)(b))
#endif

namespace alac {

/* sort key of a regular packet: numU*32 + numV, + KEY_WIDE when chanBits > 23 (the literal 32-bit predictor) */
constexpr uint32_t KEY_WIDE = 1024;
constexpr uint32_t KEY_IRREGULAR = 2048; /* everything else: scan first (alac_wave.h) */
constexpr uint32_t NUM_KEYS = 2049;

/* orders the lean decoder runs: 4/5/6/8 on exactly NA taps, the others (general form, int16 coefficient wrap) on
 * 16 register taps with wave-uniform skips; 0 copies and 31 is delta mode. 17..30 exist only on paper. */
ALAC_DEV bool regular_order(uint32_t na) { return na <= 16 || na == 31; }

/* Sort key of a packet; no entropy decoding, reads only the element header. */
ALAC_DEV uint32_t classify_regular(const DevCfg& cfg, const uint8_t* pkt, uint32_t size, uint32_t avail) {
    if (cfg.num_channels > 2 || cfg.aligned16 == 0 || cfg.kb == 0 ||
        cfg.frame_length > 65536u || cfg.frame_length <= 32u)
        return KEY_IRREGULAR;
    const Bits bits{pkt, size, avail};
    if (size < 12) return KEY_IRREGULAR;
    const uint32_t tag = bits.get(0, 3);
    const bool cpe = cfg.num_channels == 2;
    if (cpe ? tag != 1 : !(tag == 0 || tag == 3)) return KEY_IRREGULAR;
    if (bits.get(7, 12) != 0) return KEY_IRREGULAR;
    const uint32_t hdr = bits.get(19, 4);
    if (hdr & 1u) return KEY_IRREGULAR; /* escape element */
    const uint32_t bs = (hdr >> 1) & 3u;
    if (bs == 3) return KEY_IRREGULAR;
    /* 24-bit products need chanBits <= 23 (16-bit; 20-bit; 24/32-bit with their usual shift bytes); wider channels
     * (24/32-bit without shift bytes: chanBits 24..33, decoder.go:371) take the same wave pair with the literal
     * 32-bit predictor, under their own keys */
    const uint32_t chan_bits = cfg.bit_depth - 8u * bs + (cpe ? 1u : 0u);
    if (chan_bits < 1 || chan_bits > 33 || cfg.bit_depth < 8u * bs) return KEY_IRREGULAR;
    const uint32_t wide = chan_bits > 23u ? KEY_WIDE : 0u;
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
     * (anything else is an error or panic case: decode_wave reports those); with shift bytes, eight bytes of entropy
     * stream behind them keep the lean decoder's 8-byte windows on the shift values inside the packet
     * (Bits::window_raw) */
    const uint64_t ent = (uint64_t)pos + (uint64_t)bs * 8u * (cpe ? 2u : 1u) * ns;
    if ((ent >> 3) >= size || (bs != 0 && (ent >> 3) + 8u > size)) return KEY_IRREGULAR;
    return nu * 32u + nv + wide;
}

/* ---- the lean path's bit reader: an LDS ring per lane, refilled ahead of time --------------------------------
 * w0,w1 hold stream dwords widx, widx+1 and w2 the next one, as in FastRd, but they are fed from a ring of
 * W::kRingDw (16 or 32) dwords in LDS (W::ring_*), never straight from HBM. The ring is topped up 16 bytes at a time on a wave-uniform
 * schedule (every 4th step): tick() first commits the block whose global load was issued 4 steps earlier, then
 * issues the next one. So no step ever waits on an HBM/L2 round trip: the data a step needs left memory at
 * least four steps ago, and each packet byte is fetched from L2 exactly once. A plain step consumes <= 32 bits,
 * so 4 dwords per 4 steps sustain it (reseek() covers the slow path); start() prefills 16 dwords.
 * Dense blob (see Bits): blocks that lie wholly inside the packet are loaded as they are (one global_load_dwordx4);
 * a block that reaches past the packet's last byte takes tail4(): aligned dwords that hold at least one packet byte
 * are fetched (they cannot leave the blob's pages), the neighbour's bytes in them are cleared, and dwords wholly
 * behind the packet are zeros without a fetch — the reference's zero pad (bitbuffer.go:33), as far out as anyone looks. */
template <class W>
struct RingRd {
    const uint32_t* base; /* packet start rounded down to a dword */
    uint32_t bias;        /* stream bit 0 is bit `bias` of base[0] */
    uint32_t end_b;       /* first byte, counted from base, that is not packet data */
    uint32_t full;        /* dwords [0, full) of base lie wholly inside the packet */
    uint32_t w0, w1, w2, widx;
    uint32_t fill;        /* ring holds dwords [fill - RING, fill); multiple of 4 */
    static constexpr uint32_t RING = W::kRingDw;
    uint32_t p0, p1, p2, p3;
    bool pend;

    ALAC_DEV void init(const uint8_t* pkt, uint32_t size) {
        /* pointer arithmetic, not an integer round trip: the compiler keeps the global address space */
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(pkt) & 3u);
        base = reinterpret_cast<const uint32_t*>(pkt - mis);
        bias = mis * 8u;
        end_b = size ? mis + size : 0u; /* a lane without a packet keeps nothing of what it reads */
        full = end_b >> 2;
        w0 = w1 = w2 = widx = fill = 0;
        p0 = p1 = p2 = p3 = 0;
        pend = false;
    }
    /* dword idx of a block that reaches past the packet, branch-free: fetch it (or, behind the packet, the last dword
     * that holds packet bytes: an address that is always good), keep what is packet data */
    ALAC_DEV uint32_t tail1(uint32_t idx) const {
        const uint32_t last = end_b ? (end_b - 1u) >> 2 : 0u;
        const uint32_t v = base[umin(idx, last)];
        const uint32_t lo = idx * 4u;
        const uint32_t nb = end_b > lo ? umin(end_b - lo, 4u) : 0u;
        return v & (nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u));
    }
    ALAC_DEV void load4(uint32_t at) {
        /* one 16-byte load, 4-byte aligned. The dwords stay RAW (little-endian) in p0..p3: touching them here
         * would make the wave wait for the load on the spot; commit() swaps them four steps later. */
        if (at + 4u <= full) {
            ALAC_LOAD4(base + at, p0, p1, p2, p3);
        } else { /* the last blocks of the packet, and everything behind it */
            p0 = tail1(at);
            p1 = tail1(at + 1u);
            p2 = tail1(at + 2u);
            p3 = tail1(at + 3u);
        }
    }
    ALAC_DEV void commit(W& wv) {
        wv.ring_write4(fill & (RING - 1u), __builtin_bswap32(p0), __builtin_bswap32(p1), __builtin_bswap32(p2),
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
            load4(fill);
            commit(wv);
        }
        reseek(wv, pos);
    }
    ALAC_DEV void reseek(W& wv, uint32_t pos) {
        widx = (pos + bias) >> 5;
        /* a slow-path step (escape code + zero-run code) can eat more than one dword, more than tick() puts
         * back: top the ring up on the spot whenever it runs low. Plain steps take <= 32 bits (prefix + 1 + k,
         * k <= 23), which the 4 dwords per 4 steps of tick() cover. Positions are < 2^29 bits here (a live lane
         * stays below max_pos + 66), so the loop ends. */
        while (fill < widx + 12u) {
            if (!pend) load4(fill);
            commit(wv);
        }
        w0 = wv.ring_read(widx & (RING - 1u));
        w1 = wv.ring_read((widx + 1u) & (RING - 1u));
        w2 = wv.ring_read((widx + 2u) & (RING - 1u));
    }
    ALAC_DEV uint32_t window(uint32_t pos) const {
        const uint32_t r = (pos + bias) & 31u;
        return (uint32_t)(((((uint64_t)w0) << 32) | w1) << r >> 32);
    }
    /* the cache moves by 0 or 1 dword per step (the slow path reseeks); the move is a bit mask, not a compare;
     * w2 is re-read from LDS every step */
    ALAC_DEV void slide(W& wv, uint32_t pos) {
        const uint32_t ni = (pos + bias) >> 5;
        const uint32_t cm = 0u - (ni - widx);
        w0 = (w1 & cm) | (w0 & ~cm);
        w1 = (w2 & cm) | (w1 & ~cm);
        widx = ni;
        w2 = wv.ring_read((ni + 2u) & (RING - 1u));
    }
    /* every 4th step, wave-uniform */
    ALAC_DEV void tick(W& wv) {
        if (pend) commit(wv);
        if (fill + 4u <= widx + RING) {
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
template <class W, class B>
ALAC_DEV int32_t golomb_slow(const B& bits, RegLane<W>& s, uint32_t size, uint32_t kb, uint32_t wb,
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
template <class W, class B>
ALAC_DEV int32_t gol_step(W& wv, const B& bits, RegLane<W>& s, uint32_t size, uint32_t kb, uint32_t wb,
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
    s.mean = ALAC_BFI(okm, mean2, o_mean);
    s.zmode = o_zmode & ~okm;
    s.zrem = ALAC_SUBSAT(o_zrem, 1u);
    on_mask = next_on;
    /* a plain divergent branch: one compare, one exec-mask save and a skip when no lane is in there (wrapping it
     * in a wave-wide any() first only adds scalar instructions to every step) */
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
    /* den_half - sum coef_j * (top - h_j), as one multiply-add chain over e_j = h_j - top.
     * The build runs with the pre-RA scheduler off (csrc/Makefile), so the instructions issue in THIS order: each
     * tap's nine instructions together. Measured against the alternative of one operation at a time over all taps
     * (independent instructions back to back): 2.45 ms vs 3.01 ms on the benchmark batch — with a partner wave on the
     * SIMD filling the gaps, short live ranges matter more than the distance between dependent instructions. */
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

/* The same step for chanBits > 23 (32-bit streams without shift bytes): the literal form of predictor.go:99-684 on
 * plain 32-bit arithmetic, where products and the countdown may wrap exactly as the reference's int32 do. */
template <int NR, bool GEN, bool WRAP>
ALAC_DEV int32_t predict_wide(int32_t (&coef)[NR], const uint32_t (&hb)[NR + 1], uint32_t na, int32_t del,
                              uint32_t den_shift, int32_t den_half, uint32_t chan_shift) {
    constexpr uint32_t BIAS = 0x80000000u;
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

/* what the predictor wave does with the reconstructed samples (alac_duo.h) */
enum { OUT_UTILE = 0,  /* U of a pair: hand-off tile */
       OUT_STEREO = 1, /* V of a pair: unmix with U, PCM */
       OUT_MONO = 2,   /* single channel: PCM */
       OUT_RAW = 3 };  /* int32 samples into this lane's row (split pipeline, alac_split.h) */

/* Entropy scan of one channel for decode_wave<..., SCAN>: the lean Golomb loop. With res_row (wave-uniform: all
 * lanes or none) the residuals are kept: four per 16-byte store into the lane's row, so that the split pipeline's
 * predictor pass (alac_split.h) does not have to decode the stream a second time. Rows hold frame_length + 3 samples
 * rounded up to 4, lanes write whole groups of four up to the wave's longest channel (ns <= frame_length). */
template <class W, class B>
ALAC_DEV void scan_channel(W& wv, const DevCfg& cfg, const B& bits, const uint8_t* pkt, uint32_t size, bool go,
                           uint32_t& pos, uint32_t ns, uint32_t pb_local, uint32_t chan_bits, int32_t& err,
                           int32_t* res_row) {
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
    uint32_t kb = cfg.kb;
    ALAC_OWN_REG(kb);
    const uint32_t wb = go_shl(1u, kb) - 1u; /* golomb.go:60 */
    s.rd.start(wv, s.pos);
    uint32_t ns_live = my_ns;
    uint32_t on_mask = (uint32_t)((int32_t)(0u - ns_live) >> 31);
    const bool keep = res_row != nullptr;
    /* the four residuals of a group are stored at the top of the NEXT group, right behind the ring's top-up: vector
     * memory operations retire in order (vmcnt), so a store issued just before a top-up's wait would make the entropy
     * chain wait for the store's round trip; issued right after it, it has four steps to drain */
    int32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    uint32_t i = 0;
    for (; i + 4u <= n_it; i += 4u) { /* four steps per ring top-up, straight-line */
        s.rd.tick(wv);
        if (keep && go && i != 0u) ALAC_STORE4(res_row + (i - 4u), h0, h1, h2, h3);
        h0 = gol_step(wv, bits, s, size, kb, wb, chan_bits, i, my_ns, ns_live, on_mask);
        h1 = gol_step(wv, bits, s, size, kb, wb, chan_bits, i + 1u, my_ns, ns_live, on_mask);
        h2 = gol_step(wv, bits, s, size, kb, wb, chan_bits, i + 2u, my_ns, ns_live, on_mask);
        h3 = gol_step(wv, bits, s, size, kb, wb, chan_bits, i + 3u, my_ns, ns_live, on_mask);
    }
    if (keep && go && i != 0u) ALAC_STORE4(res_row + (i - 4u), h0, h1, h2, h3);
    for (; i < n_it; ++i) {
        if ((i & 3u) == 0) s.rd.tick(wv);
        const int32_t d = gol_step(wv, bits, s, size, kb, wb, chan_bits, i, my_ns, ns_live, on_mask);
        if (keep && go) res_row[i] = d;
    }
    if (go) {
        pos = s.pos;
        err = s.err;
    }
}

} /* namespace alac */
#endif
