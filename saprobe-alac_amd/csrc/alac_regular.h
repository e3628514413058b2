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
#ifndef ALAC_UNLIKELY
#define ALAC_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif
#ifndef ALAC_HD
/* callable from the host side of alacgpu.hip too */
#define ALAC_HD ALAC_DEV
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
#ifndef ALAC_FFBH
/* leading zeros, 0xffffffff for 0 (v_ffbh_u32 as it is: no fix-up to 32) */
#define ALAC_FFBH(x) ((uint32_t)(x) ? (uint32_t)__builtin_clz((uint32_t)(x)) : 0xffffffffu)
#endif
#ifndef ALAC_SEXT_BITS
/* the low `bits` (1..31) of x, sign-extended: one v_bfe_i32 on the GPU (the compiler makes two shifts of it) */
#define ALAC_SEXT_BITS(x, bits) ((int32_t)((uint32_t)(x) << (32u - (bits))) >> (32u - (bits)))
#endif
#ifndef ALAC_MED3_0
/* 0 for x <= 0, else min(x, m): with m in {0, 1} "x is positive and m is set"; one v_med3_i32 on the GPU */
#define ALAC_MED3_0(x, m) ((x) <= 0 ? 0 : ((x) < (m) ? (x) : (m)))
#endif
#ifndef ALAC_NOT_ADD
/* ~a + c: one v_xad_u32 with the inline constant -1 on the GPU */
#define ALAC_NOT_ADD(a, c) (~(uint32_t)(a) + (uint32_t)(c))
#endif
#ifndef ALAC_ALIGNBIT
/* ({hi, lo} >> sh[4:0]) as 32 bits: v_alignbit_b32 on the GPU */
#define ALAC_ALIGNBIT(hi, lo, sh) ((uint32_t)(((((uint64_t)(uint32_t)(hi)) << 32) | (uint32_t)(lo)) >> ((sh) & 31u)))
#endif
#ifndef ALAC_MULHI
/* the high 32 bits of the 64-bit product: v_mul_hi_u32 on the GPU */
#define ALAC_MULHI(a, b) ((uint32_t)(((uint64_t)(uint32_t)(a) * (uint64_t)(uint32_t)(b)) >> 32))
#endif
#ifndef ALAC_BFE
/* (x >> off[4:0]) & ((1 << width[4:0]) - 1): v_bfe_u32 on the GPU */
#define ALAC_BFE(x, off, width) ((((uint32_t)(x)) >> ((off) & 31u)) & ((1u << ((width) & 31u)) - 1u))
#endif

namespace alac {

/* sort key of a regular packet: numU*32 + numV, + KEY_WIDE when chanBits > 23 (the literal 32-bit predictor) */
constexpr uint32_t KEY_WIDE = 1024;
constexpr uint32_t KEY_IRREGULAR = 2048; /* everything else: scan first (alac_wave.h) */
constexpr uint32_t NUM_KEYS = 2049;

/* orders the lean decoder runs: 4/5/6/8 on exactly NA taps, the others (general form, int16 coefficient wrap) on
 * 16 register taps with wave-uniform skips; 0 copies and 31 is delta mode. 17..30 exist only on paper. */
ALAC_DEV bool regular_order(uint32_t na) { return na <= 16 || na == 31; }
ALAC_DEV int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* Configurations the lean Golomb step (gol_step) covers: KB >= 1, and a PB small enough that the running mean stays
 * below 2^25 + 512 whatever the stream holds (golomb.go:215 with n <= 0xffff and pb = PB * pbFactor / 4 <= 127: the
 * update is a contraction towards 512 * n; the products pb * mean and mean << 2 (golomb.go:223) then never wrap and
 * k <= 16). Every real cookie has PB 40; anything else takes the whole-packet decoder of alac_wave.h, which follows
 * the reference's uint32 arithmetic literally. */
ALAC_HD bool lean_config(const DevCfg& cfg) { return cfg.kb != 0 && cfg.pb <= 73u; }

/* Sort key of a packet; no entropy decoding, reads only the element header. */
ALAC_DEV uint32_t classify_regular(const DevCfg& cfg, const uint8_t* pkt, uint32_t size, uint32_t avail) {
    if (cfg.num_channels > 2 || cfg.aligned16 == 0 || !lean_config(cfg) ||
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
     * (anything else is an error or panic case: decode_wave reports those); with shift bytes, TEN bytes of entropy
     * stream behind them keep the lean decoder's fetches on the shift values inside the packet: the 8-byte windows
     * (Bits::window_raw) and the 12-byte block fetch of the 3-byte writer (Bits::load12), whose last block may hold a
     * single frame's two shift bytes. (Eight until round 3: a one-frame packet with a nine-byte entropy stream had its
     * block fetch pulled back by two bytes and came out with the wrong low bytes; found by the GPU suite's 17 000-packet
     * STRESS batch.) Shorter packets take the whole-packet decoder. */
    const uint64_t ent = (uint64_t)pos + (uint64_t)bs * 8u * (cpe ? 2u : 1u) * ns;
    if ((ent >> 3) >= size || (bs != 0 && (ent >> 3) + 10u > size)) return KEY_IRREGULAR;
    return nu * 32u + nv + wide;
}

/* ---- the lean path's bit reader: an LDS ring per lane, refilled ahead of time --------------------------------
 * The stream lives in a ring of W::kRingDw (32) dwords per lane in LDS (W::ring_*), never read straight from HBM by a
 * step. The ring is topped up 32 bytes at a time on a wave-uniform schedule (every 8th step): tick() first commits the
 * block whose two global loads were issued 8 steps earlier, then issues the next ones, and each packet byte is fetched
 * from L2 exactly once. EIGHT steps, not four (round 3): a lane enters a new 128-byte line every fourth block, so of a
 * wave's 64 lanes some miss L2 at every top-up, and four steps of the entropy wave are shorter than a trip to HBM. A
 * plain step consumes <= 26 bits (<= 41 with an inline escape code, which reseeks), so 8 dwords per 8 steps sustain it
 * (reseek() covers the slow path); start() prefills 24 dwords.
 * THE WINDOW (round 4). wa, wb are two consecutive stream dwords that hold the lane's next code: gol_step() asks for
 * them (ONE ds_read2_b32: ring_read2) as soon as it knows where its code ends, and first looks at them in the NEXT step,
 * behind that step's mean / k / zero-run arithmetic: the LDS round trip hides behind a dozen instructions that do not
 * need it. Rounds 2-3 kept a cache of three dwords in registers and slid it with two v_bfi per step (+ the index
 * arithmetic and the read of the third dword): ten instructions for the window where this form has four. Slot 32 of a
 * lane's row repeats slot 0 (commit), so that the pair read at slot 31 needs no wrap.
 * Positions are BIASED: stream bit p is bit p + bias of the dword array that starts at `base` (the packet's start
 * rounded down to a dword), so that no step adds the bias again. The lane state (RegLane::pos) and start / reseek / tick
 * hold such a position MINUS ONE, P: the pair is dwords P >> 5 and the next, the code starts (P & 31) + 1 = 1..32 bits
 * into it, and its 32 bits are v_alignbit_b32(wa, wb, ~P) — a shift count of 0..31 in every case, where counting from the
 * position itself would need the count 32 (or a 64-bit shift of a register PAIR, which a ds_read2_b32 delivers the
 * wrong way round). window64 takes the position itself.
 * Dense blob (see Bits): blocks that lie wholly inside the packet are loaded as they are (one global_load_dwordx4);
 * a block that reaches past the packet's last byte takes tail1(): aligned dwords that hold at least one packet byte
 * are fetched (they cannot leave the blob's pages), the neighbour's bytes in them are cleared, and dwords wholly
 * behind the packet are zeros without a fetch — the reference's zero pad (bitbuffer.go:33), as far out as anyone looks. */
template <class W>
struct RingRd {
    const uint32_t* base; /* packet start rounded down to a dword */
    uint32_t bias;        /* stream bit 0 is bit `bias` of base[0] */
    uint32_t end_b;       /* first byte, counted from base, that is not packet data */
    uint32_t full;        /* dwords [0, full) of base lie wholly inside the packet */
    uint32_t wa, wb;      /* stream dwords P >> 5 and P >> 5 + 1 of the lane's position minus one (see above) */
    uint32_t fill;        /* ring holds dwords [fill - RING, fill); multiple of 8 */
    static constexpr uint32_t RING = W::kRingDw;
    static_assert(RING >= 32, "blocks of 8 dwords need a ring of 32");
    uint32_t p0, p1, p2, p3, p4, p5, p6, p7;
    bool pend;

    ALAC_DEV void init(const uint8_t* pkt, uint32_t size) {
        /* pointer arithmetic, not an integer round trip: the compiler keeps the global address space */
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(pkt) & 3u);
        base = reinterpret_cast<const uint32_t*>(pkt - mis);
        bias = mis * 8u;
        end_b = size ? mis + size : 0u; /* a lane without a packet keeps nothing of what it reads */
        full = end_b >> 2;
        wa = wb = fill = 0;
        p0 = p1 = p2 = p3 = p4 = p5 = p6 = p7 = 0;
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
    ALAC_DEV void load8(uint32_t at) {
        /* two 16-byte loads, 4-byte aligned. The dwords stay RAW (little-endian) in p0..p7: touching them here
         * would make the wave wait for the loads on the spot; commit() swaps them eight steps later. */
        if (at + 8u <= full) {
            ALAC_LOAD4(base + at, p0, p1, p2, p3);
            ALAC_LOAD4(base + at + 4u, p4, p5, p6, p7);
        } else { /* the last blocks of the packet, and everything behind it */
            p0 = tail1(at);
            p1 = tail1(at + 1u);
            p2 = tail1(at + 2u);
            p3 = tail1(at + 3u);
            p4 = tail1(at + 4u);
            p5 = tail1(at + 5u);
            p6 = tail1(at + 6u);
            p7 = tail1(at + 7u);
        }
    }
    ALAC_DEV void commit(W& wv) {
        const uint32_t slot = fill & (RING - 1u);
        const uint32_t d0 = __builtin_bswap32(p0);
        wv.ring_write4(slot, d0, __builtin_bswap32(p1), __builtin_bswap32(p2), __builtin_bswap32(p3));
        wv.ring_write4((fill + 4u) & (RING - 1u), __builtin_bswap32(p4), __builtin_bswap32(p5), __builtin_bswap32(p6),
                       __builtin_bswap32(p7));
        /* slot RING repeats slot 0 (a block that does not start the ring writes the spare slot behind it instead) */
        wv.ring_write1(slot == 0u ? RING : RING + 1u, d0);
        fill += 8u;
        pend = false;
    }
    /* channel start: synchronous prefill from the block holding the position (posb: minus one, as everywhere below) */
    ALAC_DEV void start(W& wv, uint32_t posb) {
        const uint32_t ni = posb >> 5;
        fill = ni & ~7u;
        pend = false;
#pragma nounroll
        for (int b = 0; b < 3; ++b) {
            load8(fill);
            commit(wv);
        }
        reseek(wv, posb);
    }
    ALAC_DEV void reseek(W& wv, uint32_t posb) {
        const uint32_t ni = posb >> 5;
        /* a slow-path step (escape code + zero-run code) can eat more than one dword, more than tick() puts
         * back: top the ring up on the spot whenever it runs low: up to seven plain steps (<= 26 bits each) may follow
         * before the next top-up, and a step reads the dword behind its position's. Positions are < 2^29 bits here (a
         * live lane stays below max_pos + 66), so the loop ends. */
        while (fill < ni + 12u) {
            if (!pend) load8(fill);
            commit(wv);
        }
        wv.ring_read2(ni & (RING - 1u), wa, wb);
    }
    /* (cold) 64 stream bits from posb, MSB first, out of the LDS ring: what the slow path looks at. It used to ask the
     * stateless reader, i.e. global memory, a trip of a microsecond or two for every rare step; streams that are all
     * rare steps (24- and 32-bit without shift bytes: incompressible low bytes, every other code an escape code) were
     * bound by that. The ring is topped up first when it does not reach three dwords behind the position. */
    ALAC_DEV uint64_t window64(W& wv, uint32_t posb) {
        const uint32_t ni = posb >> 5, r = posb & 31u;
        while (fill < ni + 4u) {
            if (!pend) load8(fill);
            commit(wv);
        }
        const uint64_t a = wv.ring_read(ni & (RING - 1u)), b = wv.ring_read((ni + 1u) & (RING - 1u));
        const uint64_t c = wv.ring_read((ni + 2u) & (RING - 1u));
        const uint64_t hi = (a << 32) | b;
        return r ? (hi << r) | (c >> (32u - r)) : hi;
    }
    /* every 8th step, wave-uniform; posb: the lane's position */
    ALAC_DEV void tick(W& wv, uint32_t posb) {
        if (pend) commit(wv);
        if (fill + 8u <= (posb >> 5) + RING) {
            load8(fill);
            pend = true;
        }
    }
};

/* zero-run countdown of a lane that decodes nothing more: it has all its samples, has failed, or never had a packet. Far
 * above any real run length (<= 65535) and any number of steps that could count it down. */
constexpr uint32_t GOL_PARK = 0x40000000u;

/* per-lane Golomb + reader state of one channel. pos and max_pos are biased (RingRd). */
template <class W>
struct RegLane {
    RingRd<W> rd;
    uint32_t pos; /* biased position of the next code MINUS ONE (RingRd: THE WINDOW); max_pos: biased, not minus one */
    uint32_t mean, zmode, pb, max_pos;
    /* zeros this lane still has to queue from a zero run (golomb.go:232-240), MINUS ONE: -1 outside a run, so that the
     * mask "not inside a run" is one arithmetic shift (gol_step) */
    uint32_t zq;
    uint32_t pbs; /* pb << 23: (pb * mean) >> 9 is the high half of mean * pbs (pb <= 127: lean_config) */
    /* the HEAD of the next step: what it needs of pos, mean and zq before it looks at the window (gol_head) — worked out
     * at the end of the step before, while the ring's answer is on its way: ~pos, 31 - k, (pb * mean) >> 9, the mask
     * "not inside a zero run" */
    uint32_t h_sh, h_ck, h_t9, h_norun;
    /* nonzero: the lane takes golomb_slow() in every step it decodes a code in. Set where the plain step's shortcuts
     * do not hold: within reach of the packet's end (overrun, golomb.go:168) or of the channel's last sample
     * (golomb.go:223: no zero run behind it; lock step: nothing at all behind it), and for the code that follows a
     * zero run (zmode = 1, golomb.go:206). Recomputed at every ring top-up (gol_near) and by the slow path. */
    uint32_t near;
    int32_t err;
    ALAC_DEV uint32_t upos() const { return pos + 1u - rd.bias; } /* the stream position as the reference counts it */
    ALAC_DEV void set_upos(uint32_t p) { pos = p + rd.bias - 1u; } /* p >= 1 */
    ALAC_DEV void set_pb(uint32_t p) {
        pb = p;
        pbs = p << 23;
    }
};

/* `near` for the steps up to the next top-up (at most 8 from step i on): a plain step takes at most 8 + 1 + 16 bits
 * (lean_config), one with an inline escape code (gol_step: any chanBits <= 32) at most 9 + 32: eight of them 328; the
 * channel's last sample is among the next eight when i + 8 >= ns_live */
constexpr uint32_t GOL_TICK = 8;  /* steps per ring top-up */
constexpr uint32_t GOL_REACH = 41u * GOL_TICK; /* bits a lane can move between two looks at `near` */
template <class W>
ALAC_DEV uint32_t gol_near(const RegLane<W>& s, uint32_t i, uint32_t ns_live) {
    return ALAC_SUBSAT(s.pos + (GOL_REACH + 1u), s.max_pos) | ALAC_SUBSAT(i + GOL_TICK + 1u, ns_live) | s.zmode;
}

/* The rare part of DynDecomp (golomb.go:167-247) for one lane: the lane's last samples and everything behind them,
 * overrun, an escape code, the start of a zero run and the code behind one. Redoes the sample from its start with the
 * stateless reader; returns n + zmode (what the plain step would have queued: the predictor wave folds the sign,
 * golomb.go:206-209). Works on local copies and writes the lane state back once (stores into the state from several
 * exits make the compiler keep it in scratch memory). ns_live: the lane's sample count, 0 once it has failed. */
template <class W, class B>
ALAC_DEV uint32_t golomb_slow(W& wv, const B& bits, RegLane<W>& s, uint32_t size, uint32_t kb, uint32_t wb,
                              uint32_t chan_bits, uint32_t i, uint32_t ns, uint32_t& ns_live) {
    if (i >= ns_live) { /* nothing left to decode: park the lane (it looks like one inside an endless zero run) */
        s.zq = GOL_PARK - 1u;
        return 0u;
    }
    const uint32_t bias = s.rd.bias;
    uint32_t pos = s.upos(), mean = s.mean, zmode = s.zmode, zrem = s.zq + 1u;
    const uint32_t max_pos = s.max_pos - bias;
    int32_t err = 0;
    uint32_t ndq = 0;
    if (pos >= max_pos) {
        err = ST_OVERRUN; /* golomb.go:168-170 */
    } else {
        uint32_t m = mean >> 9;
        const uint32_t k = umin(31u - clz32(m + 3u), kb);
        m = (1u << k) - 1u;
        const uint32_t w = (uint32_t)(s.rd.window64(wv, pos + bias) >> 32);
        uint32_t n = clz32(~w);
        if (n >= 9) { /* getStreamBits(bitPos+9, maxSize), golomb.go:184-186,86-108 */
            const uint32_t gpos = pos + 9u;
            const uint32_t gb = gpos & 7u;
            const bool five = chan_bits + gb > 32u;
            if ((gpos >> 3) > size || (five && (gpos >> 3) >= size)) err = ST_MALFORMED;
            const uint64_t w2 = s.rd.window64(wv, gpos + bias);
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
            /* golomb.go:206-209 computes (nd + 1) >> 1 in uint32: for nd = 2^32 - 1 that is 0, not 2^31. The predictor
             * wave unfolds (nd >> 1) ^ -(nd & 1), which agrees everywhere else: hand it a 0 in that one case. */
            ndq = nd == 0xffffffffu ? 0u : nd;
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
                    const uint32_t wz = (uint32_t)(s.rd.window64(wv, pos + bias) >> 32);
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
    if (err == 0) {
        s.set_upos(pos);
        s.mean = mean;
        s.zmode = zmode;
        s.zq = zrem - 1u;
        /* the steps up to the next top-up: at most seven (gol_near) */
        s.near = ALAC_SUBSAT(pos + bias + GOL_REACH, s.max_pos) | ALAC_SUBSAT(i + GOL_TICK + 1u, ns_live) | zmode;
        return ndq;
    }
    s.err = err;
    s.zq = GOL_PARK - 1u;
    ns_live = 0u;
    return 0u;
}

/* the head of a step from the lane state (RegLane::h_*): at a channel's start and behind the slow path; in the plain
 * path every step works out the next one's (gol_step, block B) */
template <class W>
ALAC_DEV void gol_head(RegLane<W>& s, uint32_t c31kb) {
    s.h_sh = ~s.pos;
    s.h_norun = (uint32_t)((int32_t)s.zq >> 31);
    s.h_t9 = ALAC_MULHI(s.mean, s.pbs); /* (pb * mean) >> 9, golomb.go:215 */
    /* k = min(lg3a(mean >> 9), KB), golomb.go:172-174, as ck = 31 - k; c31kb = 31 - KB as a signed number */
    s.h_ck = (uint32_t)imax((int32_t)ALAC_FFBH((s.mean >> 9) + 3u), (int32_t)c31kb);
}

/*
 * One residual (DynDecomp, golomb.go:167-247), the form the entropy wave of alac_duo.h and the scan run. Every
 * instruction of the step costs the same, and one that needs the result of the instruction right before it costs a lone
 * wave 8.3 cycles instead of 4.8 (profiles/microbench), so the step is written for the fewest of them AND in an order in
 * which no instruction reads its predecessor's result: plain integer arithmetic on bit masks, no select, no sign folding
 * (the predictor wave does that), one branch for everything rare (golomb_slow). What lets it be short:
 *  - k is kept as 31 - k (v_ffbh gives it that way round) and the code's value is one bit-field extract;
 *  - the 64 stream bits the code lies in were asked for by the step before (RingRd: THE WINDOW);
 *  - (pb * mean) >> 9 is one v_mul_hi_u32 against pb << 23;
 *  - a lane inside a zero run (zq >= 0; a parked lane is one too) queues a 0 and moves nothing: its n is masked, and with
 *    n = 0 and mean = 0 (where a run leaves it, golomb.go:245) the mean update yields 0 again by itself; the countdown is
 *    kept minus one, so that the mask is its sign and the step down (not below -1) one v_xad_u32;
 *  - overrun, the end of the lane's samples and zmode are not tested here at all: `near` sends the lane to the slow
 *    path while any of them is within reach (RegLane).
 * TWO BLOCKS (round 4). A: window -> prefix -> value -> new position -> the ring slot of the next window; the read of
 * that slot (the caller's W::ring_read2_at, an instruction of the compiler's own, which therefore also places the wait
 * for it: right in front of the next step's block A); B: the mean, the rare-case flags and the next step's HEAD (RegLane)
 * — a dozen instructions between the read and the first look at its answer. On the GPU each block is ONE asm statement
 * (alac_gpu.h: ALAC_GOL_BLOCK_A / _B, the statements below instruction for instruction): the compiler keeps neither the
 * order of single statements nor asm statements apart from their readers without s_nop; a block it can
 * only take whole. The C++ form is what tests/host_sim runs and what ALAC_GOL_ASM 0 builds.
 * Returns n + zmode (0 inside a run).
 */
#ifndef ALAC_GOL_ASM
#define ALAC_GOL_ASM 0
#endif
template <bool ESC, class W, class B>
ALAC_DEV uint32_t gol_step(W& wv, const B& bits, RegLane<W>& s, uint32_t size, uint32_t kb, uint32_t wb, uint32_t c31kb,
                           uint32_t chan_bits, uint32_t i, uint32_t ns, uint32_t& ns_live) {
    const uint32_t o_pos = s.pos, o_mean = s.mean, o_zq = s.zq;
    RingRd<W>& rd = s.rd;
    const uint32_t o_wa = rd.wa, o_wb = rd.wb; /* stream dwords o_pos >> 5 and the next (o_pos: minus one) */
#ifdef ALAC_PAD_A /* experiment: what an instruction more in the entropy step costs */
    {
        uint32_t pad_ = i;
#pragma unroll
        for (int t_ = 0; t_ < ALAC_PAD_A; ++t_) asm volatile("v_add_u32 %0, %0, %0" : "+v"(pad_));
    }
#endif
    uint32_t n, esc, mt, pos2, aoff, zq2, mean2, nhi;
    uint32_t rare, sh2, ck2, t92, norun2;
#if ALAC_GOL_ASM
    ALAC_GOL_BLOCK_A(o_wa, o_wb, s.h_sh, s.h_ck, s.h_t9, s.h_norun, o_pos, o_mean, o_zq, s.pb, n, esc, mt, pos2, aoff, zq2, mean2, nhi);
#else
    {
        const uint32_t ck = s.h_ck, norun = s.h_norun;
        const uint32_t w = ALAC_ALIGNBIT(o_wa, o_wb, s.h_sh);   /* the 32 stream bits at the position */
        const uint32_t k = 31u - ck; /* 1..16 */
        const uint32_t nw = ~w;
        mt = o_mean - s.h_t9;
        const uint32_t pre = ALAC_FFBH(nw); /* leading ones; all 32 read as 2^32 - 1: rare either way */
        zq2 = ALAC_NOT_ADD(norun, o_zq);    /* one down inside a run, -1 stays */
        const uint32_t off = ck - pre;
        esc = ALAC_SUBSAT(pre, 8u);         /* nine ones: an escape code (golomb.go:184) */
        const uint32_t v = ALAC_BFE(w, off, k); /* the k bits behind the prefix and its 0 (golomb.go:192) */
        uint32_t pk = pre << k;
        const uint32_t c2 = v > 1u ? 1u : 0u; /* v >= 2: value v - 1 and k bits; else 0 and k - 1 bits */
        const uint32_t vm1 = ALAC_SUBSAT(v, 1u);
        pk -= pre;                          /* pre * (2^k - 1) */
        const uint32_t cons = pre + k + c2; /* prefix + 1, then k bits (v >= 2) or k - 1 */
        n = pk + vm1;
        const uint32_t cm = cons & norun;
        n &= norun;
        pos2 = o_pos + cm;
        mean2 = ALAC_MULU24(s.pb, n) + mt;  /* golomb.go:215; n <= 0xffff or rare */
        aoff = pos2 >> 3;
        nhi = n >> 16;
        aoff &= 4u * (RingRd<W>::RING - 1u); /* byte offset of ring slot (pos2 >> 5) & 31 in the lane's row */
    }
#endif
    /* the next step's window: asked for now, looked at in the next step's block A */
    wv.ring_read2_at(aoff, rd.wa, rd.wb);
#if ALAC_GOL_ASM
    ALAC_GOL_BLOCK_B(mean2, nhi, esc, pos2, zq2, s.h_norun, s.near, s.pbs, c31kb, rare, sh2, ck2, t92, norun2);
#else
    {
        /* rare cases, as nonzero-means-true flags: escape code (golomb.go:184), n > 0xffff (:216), near (RegLane), start
         * of a zero run (:223: mean * 4 < 512; mean2 < 2^26, the shift cannot wrap) */
        const uint32_t zs = ALAC_SUBSAT(128u, mean2);
        uint32_t x = mean2 >> 9;
        sh2 = ~pos2;
        const uint32_t fl = esc | nhi | zs;
        t92 = ALAC_MULHI(mean2, s.pbs);
        x += 3u;
        norun2 = (uint32_t)((int32_t)zq2 >> 31);
        x = ALAC_FFBH(x);
        rare = (fl | s.near) & s.h_norun;
        ck2 = (uint32_t)imax((int32_t)x, (int32_t)c31kb);
    }
#endif
    s.pos = pos2;
    s.mean = mean2;
    s.zq = zq2;
    s.h_sh = sh2;
    s.h_ck = ck2;
    s.h_t9 = t92;
    s.h_norun = norun2;
    uint32_t ndq = n;
    /* ONE wave-uniform branch around everything rare, and the divergent ones inside it: the plain path holds one compare
     * and one scalar branch, and whatever the compiler needs to merge the lanes' state (copies, saved exec masks) stays
     * in there */
    if (ALAC_UNLIKELY(wv.any(rare != 0u))) {
        if (rare != 0u) {
            /* ESC: the two rare cases that need no more than the 96 stream bits from the position's dword on, for a lane
             * that is not within reach of the packet's end (near), so that nothing of the reference's bounds can fail.
             * (1) An escape code (one sample in two thousand of music has one, i.e. one step in thirty of a wave of 64;
             * every other sample of a 24-bit stream without shift bytes): the value is the chan_bits <= 32 bits behind
             * the nine ones (golomb.go:184-186, getStreamBits :86-108). (2) n > 0xffff: the mean is clamped
             * (golomb.go:216-218). Everything else, and a zero run behind either (mean * 4 < 512), takes golomb_slow from
             * the lane's old state. */
            bool done = false;
            if (ESC && s.near == 0u) {
                if (esc != 0u && chan_bits <= 32u && chan_bits != 0u) {
                    const uint64_t hi = (((uint64_t)o_wa) << 32) | o_wb;
                    const uint32_t o_wc = wv.ring_read(((o_pos >> 5) + 2u) & (RingRd<W>::RING - 1u));
                    const uint32_t r1 = (o_pos & 31u) + 1u; /* 1..32: where the code starts in the pair */
                    const uint64_t w64 = r1 < 32u ? (hi << r1) | ((uint64_t)o_wc >> (32u - r1))
                                                  : ((((uint64_t)o_wb) << 32) | o_wc); /* 64 stream bits from the position */
                    const uint32_t n2 = (uint32_t)((w64 << 9) >> (64u - chan_bits));
                    const uint32_t m2 = n2 > 0xffffu ? 0xffffu : s.pb * n2 + mt;
                    if (m2 >= 128u) {
                        s.pos = o_pos + 9u + chan_bits;
                        s.mean = m2;
                        /* golomb.go:206-209 computes (nd + 1) >> 1 in uint32: see golomb_slow */
                        ndq = n2 == 0xffffffffu ? 0u : n2;
                        done = true;
                    }
                } else if (esc == 0u && (n >> 16) != 0u) {
                    s.mean = 0xffffu; /* position and countdown as the plain step left them; no zero run: 0xffff * 4 >= 512 */
                    done = true;
                }
            }
            if (!done) {
                s.pos = o_pos;
                s.mean = o_mean;
                s.zq = o_zq;
                ndq = golomb_slow(wv, bits, s, size, kb, wb, chan_bits, i, ns, ns_live);
            }
            rd.reseek(wv, s.pos); /* the window afresh */
            gol_head(s, c31kb);   /* and the head of the lane's next step */
        }
    }
    return ndq;
}

/* what the entropy wave queues -> the residual (golomb.go:206-209: del = ((nd + 1) >> 1) * (-(nd & 1) | 1)) */
ALAC_DEV int32_t gol_unfold(uint32_t nd) { return (int32_t)((nd >> 1) ^ (0u - (nd & 1u))); }

/* ---- predictor step for i > na (UnpcBlock, predictor.go:99-684), chanBits <= 23 ---------------------------------
 * hb[j] = out[i-1-j] ^ BIAS (sign-biased history: |a - b| of biased values is one unsigned sad). Taps walked from
 * the highest down. The adaptation is sign-normalised: D0 = |del| shrinks by t_j = (na-j) * ((|d_j| + rnd) >>
 * denShift) tap after tap and tap j adapts while something of D0 is left. chanBits <= 23 keeps q < 2^23 and
 * t_j < 2^27: nothing wraps, which is what makes this equal to the reference's signed countdown.
 * GEN: the wave-uniform order na on NR = 16 register taps; WRAP: int16 coefficients (predictor.go:664,675). */
/* MID (round 3): chanBits 24 and 25 — 24-bit streams without shift bytes, 32-bit streams with one shift byte — in the
 * same mask arithmetic with 32-bit products (two instructions where the 24-bit multiply-add is one) and a sticky `go`:
 * |e| < 2^26 and q < 2^26, so a single t_j = (na-j) * q < 2^30 cannot wrap, but their sum can (136 * 2^26), and the
 * countdown must not come back up through the wrap: once a tap stopped adapting, none below it does (the reference
 * breaks out of its loop there). Until then nothing has wrapped and the normalised countdown equals the reference's
 * signed one. The literal form (predict_wide) is left for chanBits 32 and 33, whose differences wrap in int32. */
template <int NR, bool GEN, bool WRAP, bool CB_POS = false, bool MID = false>
ALAC_DEV int32_t predict_narrow_core(int32_t (&coef)[NR], const uint32_t (&hb)[NR + 1], uint32_t na, int32_t del, uint32_t sgnm,
                                     uint32_t nsg, int32_t rem, uint32_t den_shift, int32_t den_half, uint32_t rnd_neg,
                                     uint32_t chan_shift) {
    constexpr uint32_t BIAS = 0x80000000u;
    uint32_t topb = hb[NR];
    if (GEN) {
#pragma unroll
        for (int j = 1; j < NR; ++j)
            if (na == (uint32_t)j) ALAC_PICK(topb, hb[j]); /* scalar branch: na is wave-uniform */
    }
#ifdef ALAC_PAD_B /* experiment: what an instruction more in the predictor step costs */
    {
        uint32_t pad_ = na;
#pragma unroll
        for (int t_ = 0; t_ < ALAC_PAD_B; ++t_) asm volatile("v_add_u32 %0, %0, %0" : "+v"(pad_));
    }
#endif
    /* no compares on the hot path (a v_cmp / v_cndmask pair costs a lone wave ~17 cycles, plain ALU ops ~5):
     * everything that depends on the sign of the residual is derived from its sign mask
     * (sgnm: ~0 for del < 0, nsg: 1 for del < 0) */
    const uint32_t rnd = rnd_neg & sgnm;
    /* rem = D0 = |del|: what is left of it after the taps above. Signed and never wrapping: the taps take at most
     * sum(na - j) * 2^23 = 136 * 2^23 < 2^31 away from a value >= 0 */
    /* den_half - sum coef_j * (top - h_j), as one multiply-add chain over e_j = h_j - top.
     * The build runs with the pre-RA scheduler off (csrc/Makefile), so the instructions issue in THIS order: each
     * tap's nine instructions together. Measured against the alternative of one operation at a time over all taps
     * (independent instructions back to back): 2.45 ms vs 3.01 ms on the benchmark batch — with a partner wave on the
     * SIMD filling the gaps, short live ranges matter more than the distance between dependent instructions. */
    int32_t acc = den_half;
    int32_t accx = 0;                               /* sum coef_j * ex_j (ALAC_TAP_ORDER 1) */
    uint32_t ntx = 0u - (topb ^ sgnm);              /* ex_j = (h_j ^ sgnm) + ntx */
    ALAC_OWN_REG(ntx); /* opaque: the compiler would take the sum apart again (xor + sub per tap instead of one v_xad_u32) */
    int32_t gos = 1;
#pragma unroll
    for (int j = NR - 1; j >= 0; --j) {
        if (GEN && (uint32_t)j >= na) continue; /* scalar branch: taps the order does not have */
#ifndef ALAC_TAP_ORDER
#define ALAC_TAP_ORDER 2
#endif
#if ALAC_TAP_ORDER == 0
        const int32_t e = (int32_t)(hb[j] - topb); /* out[i-1-j] - top; the bias cancels */
        acc = ALAC_MAD24(coef[j], e, acc);         /* uses coef[j] before its update */
        /* coefficient step sign(del) * -sign(top - h_j) = sign(del) * sign(e): (sign(e) ^ sgnm) + nsg */
        const int32_t delta = (int32_t)ALAC_XAD(ALAC_SIGN(e), sgnm, nsg);
        const uint32_t q = ALAC_SAD(topb, hb[j], rnd) >> den_shift;
        const int32_t go = ALAC_CLAMP01(rem); /* tap j adapts while the budget is not used up */
        const int32_t cj = ALAC_MAD24(delta, go, coef[j]);
        coef[j] = WRAP ? (int32_t)(int16_t)cj : cj; /* predictor.go:664,675 */
        rem = ALAC_MSUB24(rem, q, na - (uint32_t)j);
#elif ALAC_TAP_ORDER == 1
        /* the eight-instruction tap (see below) with every instruction an asm statement, in the order of rounds 1-2: the
         * gated pair kernel (k_dec16g.hip: six pairs per CU) runs 5-8 % faster with this one than with the wait-state-free
         * order (70 000 packets 3.04 against 3.18 ms, 98 304 3.28 against 3.54), every other kernel slower */
        const int32_t ex = (int32_t)ALAC_XAD(hb[j], sgnm, ntx);
        const uint32_t ae = ALAC_SAD(topb, hb[j], rnd);       /* |e| + rounding */
        const int32_t delta = ALAC_SIGN(ex);                  /* sign(del) * -sign(top - h_j) */
        accx = MID ? (int32_t)((uint32_t)accx + (uint32_t)coef[j] * (uint32_t)ex)
                   : ALAC_MAD24(coef[j], ex, accx);           /* uses coef[j] before its update */
        const uint32_t q = ae >> den_shift;
        int32_t go = ALAC_CLAMP01(rem);                       /* tap j adapts while the budget is not used up */
        if (MID) {
            go &= gos;
            gos = go;
        }
        rem = MID ? (int32_t)((uint32_t)rem - q * (na - (uint32_t)j)) : ALAC_MSUB24(rem, q, na - (uint32_t)j);
        const int32_t cj = ALAC_MAD24(delta, go, coef[j]);
        coef[j] = WRAP ? (int32_t)(int16_t)cj : cj;           /* predictor.go:664,675 */
#else
        /* Issue order (the build keeps source order, csrc/Makefile): a wave issues in order, an instruction that needs
         * the result of the one right before it waits 8.3 cycles instead of 4.8 (profiles/microbench/valu_more: chains=1),
         * and the predictor wave's own issue rate is what a pair's step takes (round 3: ten instructions more in it cost
         * 8 %, twenty more in the entropy wave 1 %). So no instruction here uses the result of its predecessor.
         * EIGHT instructions per tap (round 3; nine before): the difference is formed with the residual's sign already
         * folded in, ex = +-(out[i-1-j] - top) = (h ^ sgnm) - (top ^ sgnm) in one v_xad_u32 against the step's ntx, so that
         * the coefficient step sign(del) * sign(e) is sign(ex) itself (no second xad per tap) and the multiply-add chain
         * sums coef * ex = +-(coef * e): one conditional negation per step (acc, below) puts that right.
         * WAIT STATES (round 3). The compiler cannot look into inline asm: it counts an asm statement as no time at all and
         * assumes its result may need a wait state before anyone reads it, so between an asm statement that writes a
         * register and the first instruction that reads it there must be an instruction of the compiler's own, or it puts
         * in an s_nop — and an s_nop costs the predictor wave two thirds of a VALU instruction (measured: 20 more per step
         * +12.7 %, 20 more v_add +19.6 %). With all eight tap instructions as asm statements the order-12 step carried 20
         * of them. Hence ex and q are plain C (the compiler's own v_xad_u32 and v_lshrrev_b32) and sit where they part
         * every asm result from its reader: ae .. q, go and delta .. cj, acc and rem .. the next tap. */
        const uint32_t ae = ALAC_SAD(topb, hb[j], rnd);       /* |e| + rounding */
        const int32_t ex = (int32_t)((hb[j] ^ sgnm) + ntx);   /* plain: the compiler's own v_xad_u32 */
        int32_t go = ALAC_CLAMP01(rem);                       /* tap j adapts while the budget is not used up */
        if (MID) {
            go &= gos;
            gos = go;
        }
        const int32_t delta = ALAC_SIGN(ex);                  /* sign(del) * -sign(top - h_j) */
        const uint32_t q = ae >> den_shift;                   /* plain */
        accx = MID ? (int32_t)((uint32_t)accx + (uint32_t)coef[j] * (uint32_t)ex)
                   : ALAC_MAD24(coef[j], ex, accx);           /* uses coef[j] before its update */
        rem = MID ? (int32_t)((uint32_t)rem - q * (na - (uint32_t)j)) : ALAC_MSUB24(rem, q, na - (uint32_t)j);
        const int32_t cj = ALAC_MAD24(delta, go, coef[j]);
        coef[j] = WRAP ? (int32_t)(int16_t)cj : cj;           /* predictor.go:664,675 */
#endif
    }
    /* den_half + sum coef_j * e_j: the chain's sum with the residual's sign taken out again ((x ^ -1) + 1 = -x) */
#if ALAC_TAP_ORDER == 1
    acc = (int32_t)ALAC_XAD(accx, sgnm, nsg + (uint32_t)den_half);
#elif ALAC_TAP_ORDER == 2
    acc = (int32_t)(((uint32_t)accx ^ sgnm) + (nsg + (uint32_t)den_half)); /* plain, like ex */
#endif
    const int32_t o = del + (int32_t)(topb ^ BIAS) + (acc >> den_shift);
    /* CB_POS: the caller knows chanBits >= 1, so the shift count is <= 31 and sext_cs' guard for 32 is not needed */
    return CB_POS ? ALAC_SEXT_BITS(o, 32u - chan_shift) : sext_cs(o, chan_shift);
}

/* the step from the residual itself ... */
template <int NR, bool GEN, bool WRAP, bool CB_POS = false, bool MID = false>
ALAC_DEV int32_t predict_narrow(int32_t (&coef)[NR], const uint32_t (&hb)[NR + 1], uint32_t na, int32_t del,
                                uint32_t den_shift, int32_t den_half, uint32_t rnd_neg, uint32_t chan_shift) {
    const uint32_t sgnm = (uint32_t)(del >> 31), nsg = (uint32_t)del >> 31;
    return predict_narrow_core<NR, GEN, WRAP, CB_POS, MID>(coef, hb, na, del, sgnm, nsg, (int32_t)(((uint32_t)del ^ sgnm) + nsg), den_shift,
                                                      den_half, rnd_neg, chan_shift);
}
/* ... and from what the entropy wave queues, nd = n + zmode (gol_step): sign, magnitude and residual all come out of
 * its lowest bit and the rest (golomb.go:206-209), cheaper here than folding there and taking the sign apart again */
template <int NR, bool GEN, bool WRAP, bool CB_POS = false>
ALAC_DEV int32_t predict_narrow_nd(int32_t (&coef)[NR], const uint32_t (&hb)[NR + 1], uint32_t na, uint32_t nd,
                                   uint32_t den_shift, int32_t den_half, uint32_t rnd_neg, uint32_t chan_shift) {
    const uint32_t nsg = nd & 1u, sgnm = 0u - nsg, hm = nd >> 1;
    return predict_narrow_core<NR, GEN, WRAP, CB_POS>(coef, hb, na, (int32_t)(hm ^ sgnm), sgnm, nsg, (int32_t)(hm + nsg), den_shift,
                                                      den_half, rnd_neg, chan_shift);
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
 * lanes or none) what the steps yield (n + zmode, the sign not folded yet: gol_unfold) is kept: four per 16-byte store
 * into the lane's row, so that the split pipeline's predictor pass (alac_split.h) does not have to decode the stream a
 * second time. Rows hold frame_length + 3 values rounded up to 4, lanes write whole groups of four up to the wave's
 * longest channel (ns <= frame_length). */
template <class W, class B>
ALAC_DEV void scan_channel(W& wv, const DevCfg& cfg, const B& bits, const uint8_t* pkt, uint32_t size, bool go,
                           uint32_t& pos, uint32_t ns, uint32_t pb_local, uint32_t chan_bits, int32_t& err,
                           int32_t* res_row) {
    RegLane<W> s;
    s.rd.init(pkt, size);
    s.err = 0;
    s.max_pos = size * 8u + s.rd.bias;
    s.set_upos(go ? pos : 1u);
    s.mean = cfg.mb;
    s.zmode = 0;
    s.zq = 0xffffffffu;
    s.near = 0;
    s.set_pb(pb_local);
    const uint32_t my_ns = go ? ns : 0u;
    const uint32_t n_it = wv.max_u32(my_ns);
    uint32_t kb = cfg.kb;
    ALAC_OWN_REG(kb);
    const uint32_t wb = go_shl(1u, kb) - 1u; /* golomb.go:60 */
    const uint32_t c31kb = 31u - kb;
    s.rd.start(wv, s.pos);
    gol_head(s, c31kb);
    uint32_t ns_live = my_ns;
    const bool keep = res_row != nullptr;
    /* the four residuals of a group are stored at the top of the NEXT group, right behind the ring's top-up: vector
     * memory operations retire in order (vmcnt), so a store issued just before a top-up's wait would make the entropy
     * chain wait for the store's round trip; issued right after it, it has four steps to drain */
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    uint32_t i = 0;
    auto four = [&]() { /* four steps, straight-line; the four before them go to the row first (see above) */
        if (keep && go && i != 0u) ALAC_STORE4(res_row + (i - 4u), (int32_t)h0, (int32_t)h1, (int32_t)h2, (int32_t)h3);
        h0 = gol_step<true>(wv, bits, s, size, kb, wb, c31kb, chan_bits, i, my_ns, ns_live);
        h1 = gol_step<true>(wv, bits, s, size, kb, wb, c31kb, chan_bits, i + 1u, my_ns, ns_live);
        h2 = gol_step<true>(wv, bits, s, size, kb, wb, c31kb, chan_bits, i + 2u, my_ns, ns_live);
        h3 = gol_step<true>(wv, bits, s, size, kb, wb, c31kb, chan_bits, i + 3u, my_ns, ns_live);
        i += 4u;
    };
    while (i + GOL_TICK <= n_it) { /* eight steps per ring top-up */
        s.rd.tick(wv, s.pos);
        s.near = gol_near(s, i, ns_live);
#pragma nounroll
        for (uint32_t h = 0; h < GOL_TICK; h += 4u) four();
    }
    if (i + 4u <= n_it) {
        s.rd.tick(wv, s.pos);
        s.near = gol_near(s, i, ns_live);
        four();
    }
    if (keep && go && i != 0u) ALAC_STORE4(res_row + (i - 4u), (int32_t)h0, (int32_t)h1, (int32_t)h2, (int32_t)h3);
    for (; i < n_it; ++i) {
        if ((i & (GOL_TICK - 1u)) == 0) {
            s.rd.tick(wv, s.pos);
            s.near = gol_near(s, i, ns_live);
        }
        const uint32_t d = gol_step<true>(wv, bits, s, size, kb, wb, c31kb, chan_bits, i, my_ns, ns_live);
        if (keep && go) res_row[i] = (int32_t)d;
    }
    if (go) {
        pos = s.upos();
        err = s.err;
    }
}

} /* namespace alac */
#endif
