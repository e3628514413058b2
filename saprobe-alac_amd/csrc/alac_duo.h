/*
 * alac_duo.h — regular packets decoded by a PAIR of wavefronts per 64 packets (role specialisation).
 *
 * Why. The per-sample step of alac_regular.h is two dependency chains — the Golomb code of sample i+1 and the
 * predictor taps of sample i — that one wavefront cannot overlap: a lone gfx950 wave issues a dependent VALU
 * instruction every ~8.3 cycles and an independent one every ~4.8, and the compiler's schedule keeps each chain
 * contiguous. With ONE wave per SIMD (a 65 536-packet batch is exactly 1 024 full waves) the SIMD idles between
 * dependent instructions; two waves on a SIMD interleave in hardware (measured: 1.62x the throughput of one).
 * A batch has no second wave's worth of packets to give, so the step itself is cut in two:
 *
 *   wave A (entropy)     Golomb/Rice decode of residual chunk c (golomb.go:148-253): a serial chain, ~70
 *                        dependent VALU instructions per sample, nothing else;
 *   wave B (predictor)   adaptive FIR reconstruction (predictor.go:45-684) of chunk c-1, unmix (matrix.go:40-41),
 *                        shift-byte merge, PCM packing and the LDS stager: many short independent chains.
 *
 * The two waves of a workgroup hold the SAME 64 packets (lane = packet in both). Residuals go A -> B through a
 * double-buffered LDS queue of DUO_CHUNK steps x 64 lanes; one s_barrier per chunk is the only synchronisation.
 * Iteration c: A writes R[c & 1], B reads R[(c-1) & 1]; the barrier at the end of the iteration publishes the
 * chunk. U of a pair still goes through the HBM hand-off tile (B stores it in the U phase and loads it in the V
 * phase): V's first bit is known only when U is fully parsed. Status and frame count come from A (it sees the
 * entropy errors), PCM from B.
 *
 * ROLE_BOTH runs the same code with one caller playing both roles in turn (tests/host_sim: one lane, no
 * barriers) so the logic is checked against the oracle in the CPU suite.
 */
#ifndef ALAC_DUO_H
#define ALAC_DUO_H

#include <type_traits>

#include "alac_regular.h"

#ifndef ALAC_DUO_UN8_MAX
#define ALAC_DUO_UN8_MAX 12 /* longest predictor whose steady-state groups are 8 steps (else 4) */
#endif
#ifndef ALAC_DUO_STAMP
/* profiling build only (-DALAC_DUO_PROF in alacgpu.hip): time stamps around the parts of an iteration */
#define ALAC_DUO_STAMP(k)
#endif

namespace alac {

#ifndef ALAC_DUO_CHUNK
#define ALAC_DUO_CHUNK 16
#endif
constexpr uint32_t DUO_CHUNK = ALAC_DUO_CHUNK;
 /* steps per queue buffer (a multiple of 8) */
enum { ROLE_A = 0, ROLE_B = 1, ROLE_BOTH = 2, ROLE_C = 3 };

/*
 * One channel of a regular element. s: the lane's Golomb + reader state (role A). NA: this channel's predictor
 * order (role B; wave-uniform), 0 = the general form for na_rt in {0..16, 31} on 16 register taps.
 * F16: 16-bit PCM of a pair (one dword per frame): the writer is compiled without the other widths' selects.
 * NARROW: chanBits <= 23 (predict_narrow), else the 32-bit literal form (predict_wide).
 * mode (per lane) != 0: the delta pre-pass of decoder.go:307-309 runs on the residual stream first (split
 * pipeline only: regular packets have mode 0).
 */
template <class W, class B, int NA, int OUT, int ROLE, bool F16, bool NARROW, bool EA, bool UN8W = false, bool EC = false, int WMODE = 0>
ALAC_DEV void duo_phase(W& wv, const DevCfg& cfg, const B& bits, RegLane<W>& s, uint32_t size, uint32_t ns,
                        uint32_t n_it, uint32_t hdr_pos, uint32_t den_shift, uint32_t chan_bits, int32_t mix_res,
                        uint32_t mix_sh, uint32_t na_rt, uint32_t shift_pos, uint32_t sb, uint32_t mode) {
    constexpr bool DO_A = ROLE == ROLE_A || ROLE == ROLE_BOTH, DO_B = ROLE == ROLE_B || ROLE == ROLE_BOTH;
    constexpr bool RAW = OUT == OUT_RAW;
    constexpr bool LAST = OUT == OUT_STEREO || OUT == OUT_MONO || RAW; /* B stages what it reconstructs */
    constexpr bool CPE = OUT == OUT_STEREO;
    /* what role B reads: residuals through the LDS queue where the entropy wave has slack (it folds the sign,
     * golomb.go:206-209); n + zmode where it has not: as the scan left it in the rows of the split pipeline (a lone chain:
     * every instruction counts there), and in workgroups with a writer wave, where the entropy wave sets the pace */
    constexpr bool QND = W::kResMem || EC; /* (EC is declared with the template: with a writer wave the entropy wave is the longest) */
    constexpr bool GEN = NA == 0;
    /* GEN only ever serves orders 0 (copy) and 31 (delta): every order with taps has its own instantiation */
    constexpr int NR = GEN ? 1 : NA;
    constexpr bool WRAP = !(NA == 4 || NA == 5 || NA == 6 || NA == 8); /* predictor.go:81-93 */
    constexpr uint32_t BIAS = 0x80000000u;
    /* EA: the PCM writer runs in wave A (long predictors: B is the longer of the two). Samples then go B -> A
     * through the second half of each queue buffer, in chunks of half the size, and A writes them two chunks late. */
    /* EC (round 3): a THIRD wave writes the PCM (role C: unmix, packing, stager, flush; it reads the U tile itself).
     * What a pair's step takes is the issue time of its longest wave, and since the entropy step lost a third of its
     * instructions that is the predictor wave in the last phase (predictor + writer: 90 issue slots against 66 in the
     * U phase): with the writer in a wave of its own all three stay near 66. Same queue protocol as EA. */
    static_assert(!(EA && EC), "one writer");
    constexpr bool DO_C = (ROLE == ROLE_C || ROLE == ROLE_BOTH) && EC;
    constexpr bool EMIT_A = (EA || EC) && LAST && !RAW; /* the samples leave wave B through the queue */
    constexpr bool DO_EMIT = EMIT_A ? (EC ? DO_C : DO_A) : DO_B;
    /* FWD: with a writer wave, what the PCM of a pair needs from memory — the U samples, the shift bytes of the 3-byte
     * block writer — is still fetched by the predictor wave and handed on through the queue (rows 2 CH .. and 3 CH ..).
     * A wave's memory counter retires in issue order, loads and stores alike: a writer that fetched them itself would
     * wait, at every load, for the PCM stores it issued before (24-bit pairs: 820 ticks per step in the writer against
     * 430 in the other two). The predictor wave stores nothing in this phase. */
#ifndef ALAC_FWD
#define ALAC_FWD 0 /* measured (round 3): 16-bit pairs 2.36 -> 2.45 ms, 24-bit pairs 4.04 -> 3.78 (two waves: 3.52): not kept */
#endif
    constexpr bool FWD = ALAC_FWD != 0 && EMIT_A && EC && OUT == OUT_STEREO;
    constexpr uint32_t CH = EMIT_A ? DUO_CHUNK / 2u : DUO_CHUNK;
    const uint32_t na = GEN ? na_rt : (uint32_t)NA;
    uint32_t kb = cfg.kb;
    ALAC_OWN_REG(kb); /* its own register: cfg is an 8-dword kernel-argument tuple that would otherwise be pulled out
                         of its spill slot, whole, in every step of the entropy loop (8 v_readlane per step) */
    const uint32_t wb = go_shl(1u, kb) - 1u; /* SetAGParams golomb.go:60: KB >= 32 gives all ones (KB is a cookie byte) */
    const uint32_t c31kb = 31u - kb;         /* gol_step keeps k as 31 - k; as a signed number (KB may exceed 31) */
    const uint32_t chan_shift = 32u - chan_bits;
    const int32_t den_half = den_shift ? (int32_t)(1u << (den_shift - 1u)) : 0;
    const uint32_t rnd_neg = (1u << den_shift) - 1u;

    /* ---- role B state: coefficients, sign-biased history, byte packer ---- */
    int32_t coef[NR];
    uint32_t hb[NR + 1];
    if (DO_B) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
            coef[j] = (!GEN || ((uint32_t)j < na && na != 31)) ? (int32_t)(int16_t)bits.get(hdr_pos + 16u + 16u * (uint32_t)j, 16) : 0;
#pragma unroll
        for (int j = 0; j <= NR; ++j) hb[j] = BIAS;
    }
    uint64_t pk_acc = 0;
    uint32_t pk_n = 0;
    const uint32_t bps = cfg.bps;
    const uint64_t pk_msk = bps == 4 ? 0xffffffffull : ((1ull << (8u * bps)) - 1ull);
    const bool merge_any = DO_EMIT && LAST && !F16 && !RAW && wv.any(sb != 0);
    /* decoder.go:307-309: UnpcBlock(numActive 31, denShift 0) over the residuals before the coefficient pass */
    const bool mode_any = DO_B && wv.any(mode != 0);
    int32_t dprev = 0;
    auto prepass = [&](uint32_t idx, int32_t del) -> int32_t {
        if (!mode_any) return del;
        const int32_t dd = idx == 0 ? del : sext_cs(del + dprev, chan_shift);
        dprev = mode != 0 ? dd : dprev;
        return mode != 0 ? dd : del;
    };
    /* wrap: the int16 coefficient wrap of unpcBlockGeneral (predictor.go:664,675) is applied in this step */
    /* wide channels (chanBits > 23): 24 and 25 bits take the mask arithmetic with 32-bit products (predict_narrow_core:
     * MID), only 32 and 33 the literal form. WMODE 1: the caller's streams are 24 bits deep, their wide channels have
     * 24 or 25 bits, always; otherwise the wave looks at its lanes once per phase. */
    constexpr bool MID_ONLY = !NARROW && WMODE == 1;
    const bool mid = !NARROW && (MID_ONLY || !wv.any(ns != 0u && chan_bits > 25u));
    auto predict = [&](int32_t del, auto wrap) -> int32_t {
        constexpr bool WR = decltype(wrap)::value;
        if (NARROW) return predict_narrow<NR, GEN, WR, !RAW>(coef, hb, na, del, den_shift, den_half, rnd_neg, chan_shift);
        if (MID_ONLY || mid) return predict_narrow<NR, GEN, WR, !RAW, true>(coef, hb, na, del, den_shift, den_half, rnd_neg, chan_shift);
        if constexpr (!MID_ONLY) return predict_wide<NR, GEN, WR>(coef, hb, na, del, den_shift, den_half, chan_shift);
        return 0;
    };
    /* the same from what the queue / the row holds */
    auto predict_q = [&](int32_t x, auto wrap) -> int32_t {
        constexpr bool WR = decltype(wrap)::value;
        if (!QND) return predict(x, wrap);
        if (NARROW) return predict_narrow_nd<NR, GEN, WR, !RAW>(coef, hb, na, (uint32_t)x, den_shift, den_half, rnd_neg, chan_shift);
        return predict(gol_unfold((uint32_t)x), wrap);
    };
    using wrap_yes = std::integral_constant<bool, WRAP>;
    using wrap_no = std::integral_constant<bool, false>;
    const uint32_t nzm = mix_res != 0 ? 0xffffffffu : 0u; /* per lane: the pair is matrixed (matrix.go:34) */

    /* A: residuals of chunk c (DynDecomp, golomb.go:167-247). Whole chunks run as straight-line groups of four steps (the bitstream ring is topped
     * up once per group, 4 steps ahead of need). */
    uint32_t ns_live = s.err == 0 ? ns : 0u;
    if (DO_A) gol_head(s, c31kb); /* the head of the channel's first step (gol_step) */
    auto qval = [&](uint32_t nd) -> int32_t { return QND ? (int32_t)nd : gol_unfold(nd); };
    auto golomb_chunk = [&](uint32_t c) {
        const uint32_t buf = c & 1u;
        if ((c + 1u) * CH <= n_it) { /* CH is 8 or 16: whole top-up periods */
#pragma nounroll
            for (uint32_t g = 0; g < CH; g += GOL_TICK) {
                s.rd.tick(wv, s.pos);
                s.near = gol_near(s, c * CH + g, ns_live);
#pragma nounroll
                for (uint32_t h = g; h < g + GOL_TICK; h += 4u) {
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j)
                        wv.rq_write(buf, h + j, qval(gol_step<true>(wv, bits, s, size, kb, wb, c31kb, chan_bits, c * CH + h + j, ns, ns_live)));
                }
            }
            return;
        }
#pragma nounroll
        for (uint32_t j = 0; j < CH; ++j) {
            const uint32_t i = c * CH + j;
            if (i >= n_it) break;
            if ((i & (GOL_TICK - 1u)) == 0) {
                s.rd.tick(wv, s.pos);
                s.near = gol_near(s, i, ns_live);
            }
            wv.rq_write(buf, j, qval(gol_step<true>(wv, bits, s, size, kb, wb, c31kb, chan_bits, i, ns, ns_live)));
        }
    };
    /* PCM of frame i from its last channel's sample o (unmix / shift merge / packing / stager).
     * u: the U sample of frame i (pairs), sw: window on the frame's shift values (24/32-bit) */
    /* FP_OK / fp: 16-bit pairs in chunks that every lane keeps whole or not at all: the writer pushes a group's dwords at
     * fixed places and counts them once (GpuWave::st_put), instead of testing and counting per step */
    constexpr bool FP_OK = F16 && CPE && LAST && !EMIT_A && !RAW;
    uint32_t fp_base = 0, fp_inc = 0;
    using fp_yes = std::integral_constant<bool, true>;
    using fp_no = std::integral_constant<bool, false>;
    auto emit = [&](uint32_t i, int32_t o, int32_t u, uint64_t sw, uint32_t jj, auto fp) {
        constexpr bool FP = decltype(fp)::value;
        const bool on = i < ns;
        if (RAW) {
            wv.st_push_if((uint32_t)o, on); /* one int32 sample per step into the lane's row */
            return;
        }
        int32_t l = o, r = 0;
        if (CPE) {
            /* matrix.go:40-41 (mixRes != 0) and :50-51 (plain copy) in one branch-free form */
            const int32_t vv = o;
            /* mixRes * v: exact on 24 bits for chanBits <= 23, int32 wrap-around otherwise (as the reference's) */
            const int32_t mv = NARROW ? ALAC_MUL24(mix_res, vv) : (int32_t)((uint32_t)mix_res * (uint32_t)vv);
            l = u + (int32_t)((uint32_t)vv & nzm) - (mv >> mix_sh);
            r = (int32_t)((((uint32_t)(l - vv)) & nzm) | ((uint32_t)vv & ~nzm));
        }
        if (F16) {
            const uint32_t pcm = ((uint32_t)l & 0xffffu) | ((uint32_t)r << 16);
            if (FP) wv.st_put(fp_base, jj, pcm, fp_inc);
            else wv.st_push_if(pcm, on);
            return;
        }
        if (cfg.bit_depth == 20) { /* matrix.go:77-78, 237 */
            l = (int32_t)((uint32_t)l << 4);
            r = (int32_t)((uint32_t)r << 4);
        }
        if (merge_any) { /* matrix.go:129-132, 266-268: (x << 8*bytesShifted) | shift value */
            const uint32_t sh_l = sb ? (uint32_t)(sw >> (64u - sb)) : 0u;
            const uint32_t sh_r = (CPE && sb) ? (uint32_t)((sw << sb) >> (64u - sb)) : 0u;
            l = (int32_t)((uint32_t)l << sb) | (int32_t)sh_l;
            r = (int32_t)((uint32_t)r << sb) | (int32_t)sh_r;
        }
        {
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
    /* B: sample i (step j of chunk buffer buf) is reconstructed: history; then the U hand-off tile, the sample
     * queue to wave A, or the writer */
    auto put = [&](uint32_t buf, uint32_t j, uint32_t i, int32_t o, int32_t u, uint64_t sw, uint32_t jj, auto fp) {
#pragma unroll
        for (int t = NR; t >= 1; --t) hb[t] = hb[t - 1];
        hb[0] = (uint32_t)o ^ BIAS;
        if (!LAST) *wv.u_row(i) = o; /* dead lanes write their own unused cell */
        else if (EMIT_A) {
            wv.rq_write(buf, CH + j, o);
            if (FWD) wv.rq_write(buf, 2u * CH + j, u); /* the U sample rides along (see FWD) */
        }
        else emit(i, o, u, sw, jj, fp);
    };
    constexpr bool PK3 = CPE && LAST && !F16 && !RAW && !EMIT_A;      /* ... in the predictor wave */
    constexpr bool PK3C = CPE && LAST && !F16 && !RAW && EMIT_A && EC; /* ... in the writer wave */
    const bool sb8 = (PK3 || PK3C) && !wv.any(ns != 0u && sb != 8u);
    const bool sb0 = (PK3 || PK3C) && !wv.any(ns != 0u && sb != 0u);
    const bool pk3 = (PK3 || PK3C) && bps == 3u && (sb8 || sb0);
    const uint32_t sh_byte = shift_pos >> 3, sh_bit = shift_pos & 7u;
    /* four frames of a 3-byte pair from their V samples vv, U samples uu and the 12 bytes fetched around their eight
     * shift bytes: unmix (matrix.go:40-41 / :50-51, as in emit()), shift merge (:129-132) or the 20-bit shift (:77-78),
     * six dwords into the stager */
    auto pk3_block = [&](uint32_t rowq, const int32_t (&vv4)[4], const int32_t (&uu4)[4], uint32_t g0, uint32_t g1, uint32_t g2) {
        /* the eight shift bytes of the block: stream bits sh_bit .. sh_bit + 64 of the 12 bytes fetched */
        const uint32_t w0 = (uint32_t)(((((uint64_t)g0) << 32) | g1) << sh_bit >> 32);
        const uint32_t w1 = (uint32_t)(((((uint64_t)g1) << 32) | g2) << sh_bit >> 32);
        uint32_t lq[4], rq[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const int32_t vv = vv4[j];
            const int32_t mv = NARROW ? ALAC_MUL24(mix_res, vv) : (int32_t)((uint32_t)mix_res * (uint32_t)vv);
            int32_t l = uu4[j] + (int32_t)((uint32_t)vv & nzm) - (mv >> mix_sh);
            int32_t r = (int32_t)((((uint32_t)(l - vv)) & nzm) | ((uint32_t)vv & ~nzm));
            if (sb8) { /* matrix.go:129-132 */
                const uint32_t w = j < 2u ? w0 : w1;
                const uint32_t sl = (j & 1u) ? (w >> 8) & 0xffu : w >> 24;
                const uint32_t sr = (j & 1u) ? w & 0xffu : (w >> 16) & 0xffu;
                l = (int32_t)(((uint32_t)l << 8) | sl);
                r = (int32_t)(((uint32_t)r << 8) | sr);
            } else if (cfg.bit_depth == 20) { /* matrix.go:77-78 */
                l = (int32_t)((uint32_t)l << 4);
                r = (int32_t)((uint32_t)r << 4);
            }
            lq[j] = (uint32_t)l;
            rq[j] = (uint32_t)r;
        }
        /* L0 L0 L0 R0 | R0 R0 L1 L1 | L1 R1 R1 R1, twice */
        const uint32_t d0 = (lq[0] & 0xffffffu) | (rq[0] << 24);
        const uint32_t d1 = ((rq[0] >> 8) & 0xffffu) | (lq[1] << 16);
        const uint32_t d2 = ((lq[1] >> 16) & 0xffu) | (rq[1] << 8);
        const uint32_t d3 = (lq[2] & 0xffffffu) | (rq[2] << 24);
        const uint32_t d4 = ((rq[2] >> 8) & 0xffffu) | (lq[3] << 16);
        const uint32_t d5 = ((lq[3] >> 16) & 0xffu) | (rq[3] << 8);
        /* a lane whose frames end inside the block (a partial frame) keeps the whole dwords of its
         * 6, 12 or 18 bytes; an odd count leaves two bytes for st_tail, as the generic packer would */
        const uint32_t nv = umin(ALAC_SUBSAT(ns, rowq), 4u);
        const uint32_t nby = nv * 6u;
        wv.st_push6_n(d0, d1, d2, d3, d4, d5, nby >> 2);
        if (nby & 2u) {
            pk_acc = (nv == 1u ? d1 : d4) & 0xffffu;
            pk_n = 2u;
        }
        wv.st_step();
    };
    const uint32_t sstep_b = (CPE ? 2u : 1u) * sb;
    const uint32_t steady_end = (n_it / CH) * CH; /* whole chunks end here */
    /* A (EMIT_A): inputs of the chunk it writes, requested before the Golomb work of the iteration */
    int32_t sq_v[CH], u_v[CH];
    uint64_t sw_v[CH];
    const uint32_t sstep_a = (CPE ? 2u : 1u) * sb;
    uint32_t g_v[CH / 4u][3] = {}; /* 3-byte pairs: the 12 bytes around each block's shift bytes */
    auto whole_pk3 = [&](uint32_t c) { return PK3C && pk3 && (c + 1u) * CH <= n_it; }; /* wave-uniform */
    auto fetch_chunk = [&](uint32_t c) {
        const uint32_t buf = c & 1u;
#pragma unroll
        for (uint32_t j = 0; j < CH; ++j) {
            const uint32_t i = c * CH + j;
            sq_v[j] = 0;
            if (!EC) {
                u_v[j] = 0;
                sw_v[j] = 0;
            }
            if (i < n_it) { /* scalar: rows past n_it do not exist in the tile */
                sq_v[j] = wv.rq_read(buf, CH + j);
                if (FWD) u_v[j] = wv.rq_read(buf, 2u * CH + j);
                if (!EC) {
                    if (CPE) u_v[j] = *wv.u_row(i);
                    if (merge_any) sw_v[j] = bits.window_raw(shift_pos + i * sstep_a);
                }
            }
        }
        if (FWD && PK3C && whole_pk3(c) && sb8) {
#pragma unroll
            for (uint32_t q = 0; q < CH / 4u; ++q)
#pragma unroll
                for (uint32_t k = 0; k < 3u; ++k) g_v[q][k] = (uint32_t)wv.rq_read(buf, 3u * CH + 3u * q + k);
        }
    };
    /* the writer wave: what a chunk needs from memory (U samples, shift values) is asked for a whole iteration ahead
     * (unless the predictor wave hands it on: FWD) */
    int32_t u_n[CH];
    uint64_t sw_n[CH];
    uint32_t g_n[CH / 4u][3] = {};
    auto fetch_mem_ahead = [&](uint32_t c) {
        const bool blocks = whole_pk3(c);
#pragma unroll
        for (uint32_t j = 0; j < CH; ++j) {
            const uint32_t i = c * CH + j;
            u_n[j] = 0;
            sw_n[j] = 0;
            if (i < n_it) {
                if (CPE && !FWD) u_n[j] = *wv.u_row(i);
                if (merge_any && !blocks) sw_n[j] = bits.window_raw(shift_pos + i * sstep_a);
            }
        }
        if (!FWD && PK3C && blocks && sb8) {
#pragma unroll
            for (uint32_t q = 0; q < CH / 4u; ++q) bits.load12(sh_byte + 2u * (c * CH + 4u * q), g_n[q][0], g_n[q][1], g_n[q][2]);
        }
    };
    auto emit_chunk = [&](uint32_t c) {
        if (PK3C && whole_pk3(c)) { /* the writer wave, 3-byte pairs, a whole chunk: two blocks of four frames */
#pragma unroll
            for (uint32_t q = 0; q < CH / 4u; ++q) {
                int32_t vv4[4], uu4[4];
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    vv4[j] = sq_v[4u * q + j];
                    uu4[j] = u_v[4u * q + j];
                }
                pk3_block(c * CH + 4u * q, vv4, uu4, g_v[q][0], g_v[q][1], g_v[q][2]);
            }
            return;
        }
#pragma unroll
        for (uint32_t j = 0; j < CH; ++j) {
            const uint32_t i = c * CH + j;
            if (i < n_it) emit(i, sq_v[j], u_v[j], sw_v[j], 0u, fp_no{});
        }
        wv.st_step(); /* collective of wave A */
    };
    /* steady-state groups of role B: UN unrolled steps. Long predictors and the wide writers (64-bit shift windows)
     * take half groups, or registers run out. For the wide writers, what a group needs from HBM / L2 (the U samples
     * of its frames, the 8-byte windows on their shift values) is requested one group AHEAD, into upre / spre, so
     * that the load latency hides behind a whole group of taps (24-bit stereo: 4.65 -> 4.11 ms). The 16-bit writer
     * asks for its U samples at the top of its own group: one dword per frame, first needed a whole step later, and
     * the extra registers and moves of looking ahead cost it more than the wait (2.45 -> 2.62 ms). */
#ifndef ALAC_DUO_UN8_WIDE_MAX
#define ALAC_DUO_UN8_WIDE_MAX 0 /* same for the writers of the wider samples (generic: their groups also hold 64-bit shift windows) */
#endif
    /* UN8W: the caller's streams have 3-byte samples, whose writer (PK3 below) holds no 64-bit windows: groups of eight
     * for predictors of up to eight taps. What it asks for a group ahead (shift bytes, U samples: scattered 64-byte
     * reads from HBM) then has eight steps to arrive instead of four. */
    constexpr uint32_t UN = (NARROW && (((F16 || !LAST || RAW || EMIT_A) && NR <= ALAC_DUO_UN8_MAX) ||
                                        NR <= (UN8W ? 8 : ALAC_DUO_UN8_WIDE_MAX))) ? 8u : 4u;
    constexpr bool HBM_IN = (LAST && !RAW && !EMIT_A) || FWD; /* this wave reads the U tile / shift bytes (to write, or to hand on) */
    /* role B fed from memory (split pipeline's predictor pass: W::kResMem): the residuals of a group are requested one
     * group ahead, like the U samples of the wide writers */
    constexpr bool RMEM = W::kResMem && ROLE == ROLE_B;
    int32_t dpre[UN];
    uint32_t dpre_row = 0xffffffffu;
#ifndef ALAC_AHEAD_F16
#define ALAC_AHEAD_F16 1
#endif
    constexpr bool AHEAD = HBM_IN && (!F16 || ALAC_AHEAD_F16 != 0);
    int32_t upre[UN];
    uint64_t spre[UN];
    uint32_t pre_row = 0xffffffffu; /* first frame of the group upre / spre hold */
    /* 3-byte pairs (24-bit with one shift byte per sample, 20-bit): four frames are exactly six dwords, at the same
     * byte positions in every group, and the eight shift bytes of four frames are eight consecutive stream bytes. The
     * writer then works per group: one 12-byte fetch for the shift bytes (instead of four 8-byte windows), byte picks
     * instead of 64-bit shifts, three byte permutes per two frames instead of the generic packer's selects
     * (BASELINE config c). All live lanes must agree on the shift width (a wave of 24-bit packets with bytesShifted 2
     * among them takes the generic writer). */
    constexpr uint32_t NSUB = UN / 4u; /* blocks of four frames in a group */
    uint32_t gpre[NSUB][3] = {};
    auto prefetch_group = [&](uint32_t row0) {
        pre_row = row0;
#pragma unroll
        for (uint32_t j = 0; j < UN; ++j) {
            upre[j] = 0;
            spre[j] = 0;
            if (CPE) upre[j] = *wv.u_row(row0 + j);
            if (merge_any && !pk3) spre[j] = bits.window_raw(shift_pos + (row0 + j) * sstep_b);
        }
        if ((PK3 || (FWD && PK3C)) && pk3 && sb8) {
#pragma unroll
            for (uint32_t q = 0; q < NSUB; ++q) bits.load12(sh_byte + 2u * (row0 + 4u * q), gpre[q][0], gpre[q][1], gpre[q][2]);
        }
    };
    /* B: samples of chunk c (UnpcBlock, predictor.go:45-684): out[0] = residual, warm-up up to na (:53-79),
     * copy (0) / delta (31) modes, then the adaptive taps */
    auto predict_chunk = [&](uint32_t c) {
        const uint32_t buf = c & 1u;
        const bool simple = GEN && (na == 0 || na == 31);
        /* both shift values of a frame sit side by side (decoder.go:492-502): one window */
        const uint32_t sstep = (CPE ? 2u : 1u) * sb;
        if (!simple && !mode_any && c * CH > na && (c + 1u) * CH <= n_it) {
            /* steady state, a whole chunk: straight-line code; residuals (LDS), U samples and shift values
             * (HBM/L2) are all requested up front and their latency hides behind the taps of the first steps;
             * the history shift becomes register renaming across the unrolled steps */
            auto groups = [&](auto wrap, auto fp) {
                constexpr bool FP = decltype(fp)::value;
#if ALAC_AHEAD_F16 == 2
#pragma unroll
#else
#pragma nounroll
#endif
                for (uint32_t g = 0; g < CH; g += UN) {
                    const uint32_t row0 = c * CH + g;
                    if (FP) fp_base = wv.st_group_base(UN);
                    int32_t dv[UN], uv[UN];
                    uint64_t sv[UN];
                    if (AHEAD && pre_row != row0) prefetch_group(row0); /* first steady group: nothing was ahead */
                    if (RMEM && dpre_row != row0) {
#pragma unroll
                        for (uint32_t j = 0; j < UN; ++j) dpre[j] = wv.rq_read(buf, g + j);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < UN; ++j) {
                        dv[j] = RMEM ? dpre[j] : wv.rq_read(buf, g + j);
                        uv[j] = AHEAD ? upre[j] : 0;
                        sv[j] = AHEAD ? spre[j] : 0ull;
                        if (HBM_IN && !AHEAD && CPE) uv[j] = *wv.u_row(row0 + j);
                    }
                    uint32_t gq[NSUB][3];
#pragma unroll
                    for (uint32_t q = 0; q < NSUB; ++q) {
                        gq[q][0] = gpre[q][0];
                        gq[q][1] = gpre[q][1];
                        gq[q][2] = gpre[q][2];
                    }
                    if (AHEAD && row0 + 2u * UN <= steady_end) prefetch_group(row0 + UN);
                    if (RMEM && row0 + 2u * UN <= steady_end) { /* rq_read indexes from the chunk's first step */
                        dpre_row = row0 + UN;
#pragma unroll
                        for (uint32_t j = 0; j < UN; ++j) dpre[j] = wv.rq_read(buf, g + UN + j);
                    }
                    if (FWD && PK3C && pk3 && sb8) { /* the shift bytes of the group's blocks, for the writer wave */
#pragma unroll
                        for (uint32_t q = 0; q < NSUB; ++q)
#pragma unroll
                            for (uint32_t k = 0; k < 3u; ++k) wv.rq_write(buf, 3u * CH + 3u * (g / 4u + q) + k, (int32_t)gq[q][k]);
                    }
                    if (PK3 && pk3) {
#pragma unroll
                        for (uint32_t q = 0; q < NSUB; ++q) {
                            int32_t vv4[4], uu4[4];
#pragma unroll
                            for (uint32_t j = 0; j < 4u; ++j) {
                                const int32_t vv = predict_q(dv[4u * q + j], wrap);
#pragma unroll
                                for (int t = NR; t >= 1; --t) hb[t] = hb[t - 1];
                                hb[0] = (uint32_t)vv ^ BIAS;
                                vv4[j] = vv;
                                uu4[j] = uv[4u * q + j];
                            }
                            pk3_block(row0 + 4u * q, vv4, uu4, gq[q][0], gq[q][1], gq[q][2]);
                        }
                        continue;
                    }
#pragma unroll
                    for (uint32_t j = 0; j < UN; ++j)
                        put(buf, g + j, c * CH + g + j, predict_q(dv[j], wrap), uv[j], sv[j], j, fp);
                    if (FP) wv.st_advance(fp_inc);
                    /* collective of the writing wave, once per group: a lane row holds 64 dwords, a flush takes
                     * 32, and a group adds at most 8 steps x 2 dwords */
                    if (LAST && !EMIT_A) wv.st_step();
                }
            };
            /* the writer's short cut (FP_OK above) when no lane's frames end strictly inside this chunk; a lane whose
             * frames do writes its tail out right behind the chunk, for the short cut scribbles over finished rows */
            auto run_groups = [&](auto wrap) {
                if (FP_OK) {
                    const uint32_t c_end = (c + 1u) * CH;
                    const bool ends_here = ns > c * CH && ns < c_end;
                    if (!wv.any(ends_here)) {
                        fp_inc = ns >= c_end ? UN : 0u;
                        groups(wrap, fp_yes{});
                        return;
                    }
                    groups(wrap, fp_no{});
                    if (ends_here) (void)wv.st_finish();
                    return;
                }
                groups(wrap, fp_no{});
            };
            if (WRAP) {
                /* A coefficient moves by at most 1 per step, so one that is further than a chunk away from the
                 * int16 limits cannot wrap inside this chunk: test once per chunk and run the chunk without the
                 * per-tap, per-step sign extension (one instruction of ten) unless some lane is that close. */
                constexpr uint32_t T = 32767u - CH;
                uint32_t far = 0;
#pragma unroll
                for (int j = 0; j < NR; ++j) far = umax(far, (uint32_t)coef[j] + T);
                if (!wv.any(far > 2u * T)) {
                    run_groups(wrap_no{});
                    return;
                }
            }
            run_groups(wrap_yes{});
            return;
        }
        if (FWD && PK3C && whole_pk3(c) && sb8) { /* a whole chunk outside the steady state: its shift bytes, on the spot */
#pragma unroll
            for (uint32_t q = 0; q < CH / 4u; ++q) {
                uint32_t g0, g1, g2;
                bits.load12(sh_byte + 2u * (c * CH + 4u * q), g0, g1, g2);
                wv.rq_write(buf, 3u * CH + 3u * q, (int32_t)g0);
                wv.rq_write(buf, 3u * CH + 3u * q + 1u, (int32_t)g1);
                wv.rq_write(buf, 3u * CH + 3u * q + 2u, (int32_t)g2);
            }
        }
#pragma nounroll
        for (uint32_t j = 0; j < CH; ++j) {
            const uint32_t i = c * CH + j;
            if (i >= n_it) break;
            const int32_t qv = wv.rq_read(buf, j);
            const int32_t del = prepass(i, QND ? gol_unfold((uint32_t)qv) : qv);
            int32_t o;
            if (i == 0 || (GEN && na == 0)) o = del;
            else if (i <= na || (GEN && na == 31)) o = sext_cs(del + (int32_t)(hb[0] ^ BIAS), chan_shift);
            else o = predict(del, wrap_yes{});
            put(buf, j, i, o, (CPE && (!EMIT_A || FWD)) ? *wv.u_row(i) : 0,
                (!EMIT_A && merge_any) ? bits.window_raw(shift_pos + i * sstep) : 0ull, 0u, fp_no{});
            if (LAST && !EMIT_A) wv.st_step();
            if (FP_OK && i + 1u == ns) (void)wv.st_finish(); /* see run_groups */
        }
    };

    /* iteration c: A produces residual chunk c while B consumes chunk c-1 (and, EMIT_A, A writes the PCM of chunk
     * c-2 from the samples B queued in iteration c-1); the barrier publishes the buffers written in the iteration */
    const uint32_t nch = (n_it + CH - 1u) / CH;
    const uint32_t iters = nch + (EMIT_A ? 2u : 1u);
#ifdef ALAC_DUO_PROF
    constexpr uint32_t kProfPhase = LAST ? 4u : 0u; /* ALAC_DUO_STAMP: the U phase and the last phase apart */
#endif
    for (uint32_t c = 0; c < iters; ++c) {
        ALAC_DUO_STAMP(0);
        if (DO_A) {
            if (EMIT_A && !EC && c >= 2u) fetch_chunk(c - 2u);
            if (c < nch) golomb_chunk(c);
        }
        ALAC_DUO_STAMP(1);
        if (DO_B) {
            if (c >= 1u && c <= nch) predict_chunk(c - 1u);
        }
        ALAC_DUO_STAMP(2);
        if (DO_A && EMIT_A && !EC) {
            if (c >= 2u) emit_chunk(c - 2u);
        }
        if (DO_C && EMIT_A) {
            if (c >= 2u) {
#pragma unroll
                for (uint32_t j = 0; j < CH; ++j) {
                    if (!FWD) u_v[j] = u_n[j];
                    sw_v[j] = sw_n[j];
                }
                if (!FWD) {
#pragma unroll
                    for (uint32_t q = 0; q < CH / 4u; ++q) {
                        g_v[q][0] = g_n[q][0];
                        g_v[q][1] = g_n[q][1];
                        g_v[q][2] = g_n[q][2];
                    }
                }
            }
            if ((merge_any || !FWD) && c >= 1u && c <= nch) fetch_mem_ahead(c - 1u);
            if (c >= 2u) {
                fetch_chunk(c - 2u);
                emit_chunk(c - 2u);
            }
        }
        ALAC_DUO_STAMP(3);
        wv.duo_sync();
        ALAC_DUO_STAMP(4);
    }
    if (DO_EMIT && LAST) wv.st_tail(pk_acc, pk_n); /* bytes of the last, incomplete dword */
    if (!LAST) wv.duo_sync_mem();                  /* the U tile is complete before anyone loads from it */
}

/*
 * Role B on SEVERAL LANES PER PACKET: the predictor waves of the four-wave workgroups (k_dec16q.hip) in small batches. A
 * predictor step costs eight instructions per tap and a workgroup's step is its longest wave's: with twelve taps role B
 * needs 120 instructions against the entropy wave's ~45, and while every workgroup has a CU to itself nothing else fills
 * the SIMD: a lone wave issues an instruction every 4.8 cycles, so a packet's time IS its longest wave's instruction count
 * (profiles/r04_final/single_packets.txt: 1.15 / 1.38 ms for six / eight taps on one lane). So a packet's taps are spread
 * over LPP = 2 (round 3) or 4 (round 4) neighbouring lanes of a predictor wave: lane q holds taps q T .. q T + T - 1
 * (T = ceil(order / LPP)), the prediction is the sum over the group (DPP adds), what the higher lanes' taps take off the
 * adaptation countdown (alac_regular.h: predict_narrow_core) reaches the lower lanes by DPP moves — the t_j do not depend
 * on the countdown, so every lane's starting value is |del| minus the sum over the lanes above it —, the history moves
 * by renaming and one DPP move, the sample enters at lane 0. ~10 T + 30 instructions per step. The order is wave-uniform
 * (one key per workgroup), so top = out[i - 1 - order] sits at a fixed place: lane (order - 1) / T, register
 * order - that * T. A wave holds 64 / LPP packets: two predictor waves (roles 1 and 3) cover 64 packets on two lanes and 32
 * on four (batches of up to 16 383 packets: alacgpu.hip: pick_ppw).
 * Protocol, queue and hand-off rows exactly as role B of duo_phase() in a workgroup with a writer wave (EC): residuals
 * come as n + zmode, U samples go to the hand-off tile, the samples of the last channel to rows CH.. of the queue buffer
 * for the writer wave. wv.lane is the packet's COLUMN (its lane in the entropy and writer waves), q the lane's number in
 * its group. chanBits <= 23 only.
 */
/* NAC: the order as a compile-time constant (four lanes: where top sits is then one DPP pattern and one register, without
 * a branch or a select in the step), 0: the wave-uniform run-time value na (two lanes: one select) */
template <class W, class B, int T, int OUT, int LPP, int NAC = 0>
ALAC_DEV void duo_phase_lanes(W& wv, const B& bits, uint32_t q, uint32_t n_it, uint32_t hdr_pos, uint32_t den_shift,
                              uint32_t chan_bits, uint32_t na_rt) {
    static_assert(LPP == 2 || LPP == 4, "lanes per packet");
    static_assert(LPP == 2 ? NAC == 0 : (NAC >= 1 && (NAC + 3) / 4 == T), "four lanes: T = ceil(order / 4)");
    const uint32_t na = NAC != 0 ? (uint32_t)NAC : na_rt;
    constexpr bool LAST = OUT != OUT_UTILE;
    constexpr uint32_t CH = LAST ? DUO_CHUNK / 2u : DUO_CHUNK;
    constexpr uint32_t BIAS = 0x80000000u;
    const bool wraps = order_wraps16(na); /* wave-uniform */
    const uint32_t chan_shift = 32u - chan_bits;
    const int32_t den_half = den_shift ? (int32_t)(1u << (den_shift - 1u)) : 0;
    const int32_t dh0 = q == 0u ? den_half : 0; /* the rounding term enters the sum once */
    const uint32_t rnd_neg = (1u << den_shift) - 1u;
    const uint32_t q0m = q == 0u ? 0xffffffffu : 0u;
    /* (four lanes) which of the lanes above this one exist: lane q takes the sums of lanes q + 1 .. 3 */
    const uint32_t up1 = q < 3u ? 0xffffffffu : 0u, up2 = q < 2u ? 0xffffffffu : 0u, up3 = q < 1u ? 0xffffffffu : 0u;
    int32_t coef[T], wneg[T], m[T];
    uint32_t g[T + 1]; /* g[t] = out[i - 1 - (q T + t)] ^ BIAS */
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t jt = q * (uint32_t)T + (uint32_t)t;
        const bool have = jt < na;
        coef[t] = have ? (int32_t)(int16_t)bits.get(hdr_pos + 16u + 16u * jt, 16) : 0;
        wneg[t] = have ? -(int32_t)(na - jt) : 0;
        m[t] = have ? 1 : 0;
        g[t] = BIAS;
    }
    g[T] = BIAS;
    int32_t prev = 0;
    /* top = out[i - 1 - na]: g[TTOP] of lane QTOP (two lanes: lane 1's g[T], g[T - 1] when the order is odd) */
    constexpr int QTOP = NAC != 0 ? (NAC - 1) / T : 1, TTOP = NAC != 0 ? NAC - QTOP * T : T; /* TTOP = 1..T */
    const bool top_last = na == 2u * (uint32_t)T; /* (two lanes) */

    /* one step: sample i from what the entropy wave queued (nd = n + zmode, golomb.go:206-209). PLAIN: i > order. */
    auto step = [&](uint32_t i, uint32_t nd, auto plain_c, auto wrap_c) -> int32_t {
        constexpr bool PLAIN = decltype(plain_c)::value, WR = decltype(wrap_c)::value;
        uint32_t topb;
        if constexpr (LPP == 2) topb = wv.pair_hi(top_last ? g[T] : g[T - 1]);
        else topb = wv.template quad_from<QTOP>(g[TTOP]);
        const uint32_t nsg = nd & 1u, sgnm = 0u - nsg, hm = nd >> 1;
        const int32_t del = (int32_t)(hm ^ sgnm);
        int32_t rem = (int32_t)(hm + nsg); /* |del| */
        if (!PLAIN) {
            if (i <= na) rem = 0; /* nothing adapts during the warm-up (wave-uniform) */
        }
        const uint32_t rnd = rnd_neg & sgnm;
        int32_t acc = dh0, s = 0;
        int32_t e[T], sg[T];
        uint32_t qv[T];
#pragma unroll
        for (int t = T - 1; t >= 0; --t) {
            e[t] = (int32_t)(g[t] - topb);
            const uint32_t ae = ALAC_SAD(topb, g[t], rnd);
            sg[t] = ALAC_SIGN(e[t]);
            qv[t] = ae >> den_shift;
            if (t != 0) {
                acc = ALAC_MAD24(coef[t], e[t], acc);
                s = ALAC_MAD24(qv[t], wneg[t], s);
            } else { /* what the DPP moves read comes from instructions the compiler knows (it inserts their wait states) */
                acc = ALAC_MUL24(coef[t], e[t]) + acc;
                s = ALAC_MUL24((int32_t)qv[t], wneg[t]) + s;
            }
        }
        /* minus what the taps of the lanes above take away */
        if (LPP == 2) rem += (int32_t)(wv.pair_hi((uint32_t)s) & q0m);
        else rem += (int32_t)((wv.template quad_up<1>((uint32_t)s) & up1) + (wv.template quad_up<2>((uint32_t)s) & up2) +
                              (wv.template quad_up<3>((uint32_t)s) & up3));
#pragma unroll
        for (int t = T - 1; t >= 0; --t) {
            const int32_t go = ALAC_MED3_0(rem, m[t]);
            const int32_t delta = (int32_t)ALAC_XAD(sg[t], sgnm, nsg);
            rem = ALAC_MAD24(qv[t], wneg[t], rem);
            int32_t cj = ALAC_MAD24(delta, go, coef[t]);
            if (WR) cj = (int32_t)(int16_t)cj; /* predictor.go:664,675 */
            coef[t] = cj;
        }
        const int32_t at = LPP == 2 ? wv.pair_sum(acc) : wv.quad_sum(acc);
        int32_t o = ALAC_SEXT_BITS(del + (int32_t)(topb ^ BIAS) + (at >> den_shift), chan_bits);
        if (!PLAIN) {
            if (i == 0u) o = del; /* out[0] = pc1[0] */
            else if (i <= na) o = sext_cs(del + prev, chan_shift); /* predictor.go:63-79 */
        }
        /* the tap below a lane's first comes from the lane before */
        const uint32_t g0 = LPP == 2 ? wv.pair_from_below(g[T - 1], (uint32_t)o ^ BIAS, q0m) : wv.quad_from_below(g[T - 1], (uint32_t)o ^ BIAS, q0m);
#pragma unroll
        for (int t = T; t >= 1; --t) g[t] = g[t - 1];
        g[0] = g0;
        prev = o;
        return o;
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    auto put = [&](uint32_t buf, uint32_t k, uint32_t i, int32_t o) {
        if (!LAST) *wv.u_row(i) = o; /* every lane of the group stores the same value to the packet's cell */
        else wv.rq_write(buf, CH + k, o);
    };
    auto predict_chunk = [&](uint32_t cc) {
        const uint32_t i0 = cc * CH, buf = cc & 1u;
        const uint32_t nst = umin(CH, n_it - i0);
        if (i0 > na && nst == CH) {
            auto run = [&](auto wrap_c) {
                uint32_t dv[CH];
#pragma unroll
                for (uint32_t k = 0; k < CH; ++k) dv[k] = (uint32_t)wv.rq_read(buf, k);
#pragma unroll
                for (uint32_t k = 0; k < CH; ++k) put(buf, k, i0 + k, step(i0 + k, dv[k], yes{}, wrap_c));
            };
            if (wraps) {
                /* a coefficient moves by at most 1 per step: one that is further than a chunk away from the int16 limits
                 * cannot wrap inside this chunk (predict_chunk of duo_phase) */
                constexpr uint32_t TW = 32767u - CH;
                uint32_t far = 0;
#pragma unroll
                for (int t = 0; t < T; ++t) far = umax(far, (uint32_t)coef[t] + TW);
                if (wv.any(far > 2u * TW)) {
                    run(yes{});
                    return;
                }
            }
            run(no{});
            return;
        }
#pragma nounroll
        for (uint32_t k = 0; k < nst; ++k) {
            const uint32_t nd = (uint32_t)wv.rq_read(buf, k);
            const int32_t o = wraps ? step(i0 + k, nd, no{}, yes{}) : step(i0 + k, nd, no{}, no{});
            put(buf, k, i0 + k, o);
        }
    };
    const uint32_t nch = (n_it + CH - 1u) / CH;
    const uint32_t iters = nch + (LAST ? 2u : 1u);
    for (uint32_t c = 0; c < iters; ++c) {
        if (c >= 1u && c <= nch) predict_chunk(c - 1u);
        wv.duo_sync();
    }
    if (!LAST) wv.duo_sync_mem(); /* the U tile is complete before anyone loads from it */
}

/* taps per lane by the (wave-uniform) order, 3..16 (duo_lanes_key) */
template <class W, int OUT, int LPP, class B>
ALAC_DEV void duo_phase_lanes_na(W& wv, uint32_t na, const B& bits, uint32_t q, uint32_t n_it, uint32_t hdr_pos, uint32_t den_shift,
                                 uint32_t chan_bits) {
    if constexpr (LPP == 4) {
#define ALAC_L4(N) \
    case N: duo_phase_lanes<W, B, (N + 3) / 4, OUT, 4, N>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
        switch (na) { /* 3..16 (duo_lanes_key) */
            ALAC_L4(3) ALAC_L4(4) ALAC_L4(5) ALAC_L4(6) ALAC_L4(7) ALAC_L4(8) ALAC_L4(9) ALAC_L4(10) ALAC_L4(11) ALAC_L4(12)
            ALAC_L4(13) ALAC_L4(14) ALAC_L4(15)
            default: duo_phase_lanes<W, B, 4, OUT, 4, 16>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
        }
#undef ALAC_L4
    } else {
        switch ((na + 1u) / 2u) {
            case 0:
            case 1:
            case 2: duo_phase_lanes<W, B, 2, OUT, 2>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
            case 3: duo_phase_lanes<W, B, 3, OUT, 2>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
            case 4: duo_phase_lanes<W, B, 4, OUT, 2>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
            case 5: duo_phase_lanes<W, B, 5, OUT, 2>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
            case 6: duo_phase_lanes<W, B, 6, OUT, 2>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
            case 7: duo_phase_lanes<W, B, 7, OUT, 2>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
            default: duo_phase_lanes<W, B, 8, OUT, 2>(wv, bits, q, n_it, hdr_pos, den_shift, chan_bits, na); break;
        }
    }
}

/* keys the several-lane predictor waves take: every order 3..16 and the longer one at least `lanes_min` */
ALAC_DEV bool duo_lanes_key(uint32_t key, bool cpe, uint32_t lanes_min) {
    const uint32_t nu = (key >> 5) & 31u, nv = key & 31u;
    if ((key & KEY_WIDE) != 0u || nu < 3u || nu > 16u) return false;
    if (!cpe) return nu >= lanes_min;
    if (nv < 3u || nv > 16u) return false;
    return umax(nu, nv) >= lanes_min;
}

/* which wave writes the PCM of a channel with predictor order na: B's step grows by nine instructions per tap, A's
 * does not; for single channels the writer balances the pair better in A from order 5 on (mono 16-bit: 1.93 ->
 * 1.63 ms). Not for pairs: the writer then needs the U tile and the shift bytes from HBM, and in wave A every wait
 * for the bitstream ring (vmcnt counts in order) would also wait for those loads (measured: 3.15 -> 3.55 ms). */
ALAC_DEV constexpr bool duo_emit_in_a(uint32_t na, bool cpe) { return !cpe && na >= 5u && na <= 16u; }

/* the order switch is scalar: NA is wave-uniform by construction of the waves. Role A never looks at the order. */
template <class W, int OUT, int ROLE, bool F16, bool NARROW = true, bool UN8W = false, bool EC = false, int WMODE = 0, class B>
ALAC_DEV void duo_phase_na(W& wv, uint32_t na, const DevCfg& cfg, const B& bits, RegLane<W>& s, uint32_t size,
                           uint32_t ns, uint32_t n_it, uint32_t hdr_pos, uint32_t den_shift, uint32_t chan_bits,
                           int32_t mix_res, uint32_t mix_sh, uint32_t shift_pos, uint32_t sb, uint32_t mode = 0u) {
#define ALAC_DUO_CASE(N)                                                                                              \
    case N:                                                                                                           \
        duo_phase<W, B, N, OUT, ROLE, F16, NARROW, CAN_EA && duo_emit_in_a(N, OUT == OUT_STEREO), UN8W, EC, WMODE>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, \
                                                                       den_shift, chan_bits, mix_res, mix_sh, na,     \
                                                                       shift_pos, sb, mode);                          \
        break;
    /* phases that write PCM; not the wide ones; not when a third wave writes */
    constexpr bool CAN_EA = !EC && NARROW && (OUT == OUT_STEREO || OUT == OUT_MONO);
    if (ROLE == ROLE_A || ROLE == ROLE_C) { /* neither looks at the order */
        if constexpr (CAN_EA) {
            if (duo_emit_in_a(na, OUT == OUT_STEREO)) {
                duo_phase<W, B, 0, OUT, ROLE, F16, NARROW, true, false, false>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift,
                                                                            chan_bits, mix_res, mix_sh, na, shift_pos, sb, mode);
                return;
            }
        }
        duo_phase<W, B, 0, OUT, ROLE, F16, NARROW, false, false, EC, WMODE>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits,
                                                                         mix_res, mix_sh, na, shift_pos, sb, mode);
        return;
    }
    switch (na) {
        ALAC_DUO_CASE(1)
        ALAC_DUO_CASE(2)
        ALAC_DUO_CASE(3)
        ALAC_DUO_CASE(4)
        ALAC_DUO_CASE(5)
        ALAC_DUO_CASE(6)
        ALAC_DUO_CASE(7)
        ALAC_DUO_CASE(8)
        ALAC_DUO_CASE(9)
        ALAC_DUO_CASE(10)
        ALAC_DUO_CASE(11)
        ALAC_DUO_CASE(12)
        ALAC_DUO_CASE(13)
        ALAC_DUO_CASE(14)
        ALAC_DUO_CASE(15)
        ALAC_DUO_CASE(16)
        default:
            duo_phase<W, B, 0, OUT, ROLE, F16, NARROW, false, false, EC, WMODE>(wv, cfg, bits, s, size, ns, n_it, hdr_pos, den_shift, chan_bits,
                                                                             mix_res, mix_sh, na, shift_pos, sb, mode);
            break;
    }
#undef ALAC_DUO_CASE
}

/*
 * decode_regular_duo: same contract as decode_wave (alac_wave.h) for the caller playing role A (or both):
 * every lane holds a regular packet with the same key = numU*32 + numV (+ KEY_WIDE), lanes without a packet pass live = false;
 * returns the status word and sets *frames_out. The caller playing role B passes the same arguments; its return
 * value and *frames_out mean nothing.
 */
/* EC: the caller's workgroup has a third wave (ROLE_C) that writes the PCM of the narrow phases (duo_phase); its return
 * value and *frames_out mean nothing either. */
/* LANES 2 / 4: the caller is a predictor wave on that many lanes per packet (duo_phase_lanes; wv.lane = the packet's column,
 * q = the lane's number in its group); ns_other: the frame count of the packet in the same column of the workgroup's other
 * predictor wave, so that both agree with the entropy wave on the number of steps. */
template <class W, int ROLE, int WIDE_SEL = -1, int DEPTH_SEL = 0, bool EC = false, int LANES = 0>
ALAC_DEV int32_t decode_regular_duo(W& wv, const DevCfg& cfg, uint32_t key, bool live, const uint8_t* pkt, uint32_t size,
                                    uint32_t avail, uint8_t* out, uint32_t* frames_out, uint32_t q = 0u, uint32_t ns_other = 0u) {
    constexpr bool DO_A = ROLE == ROLE_A || ROLE == ROLE_BOTH, DO_B = ROLE == ROLE_B || ROLE == ROLE_BOTH;
    constexpr bool DO_C = ROLE == ROLE_C || ROLE == ROLE_BOTH;
    constexpr int WM = DEPTH_SEL == 24 ? 1 : 0; /* 24-bit streams: wide channels have 24 or 25 bits, never more (duo_phase) */
    const BitsT<false> bits{pkt, size, avail}; /* regular packets hold at least 12 bytes (classify_regular) */
    const bool cpe = cfg.num_channels == 2;
    /* chanBits > 23: predict_wide. WIDE_SEL 0 / 1: the caller only ever passes keys of that kind (the other half is
     * not instantiated: the GPU library compiles the two halves as separate kernels, in parallel) */
    const bool wide = WIDE_SEL < 0 ? (key & KEY_WIDE) != 0 : WIDE_SEL != 0;
    const uint32_t na_u = (key >> 5) & 31u, na_v = key & 31u;

    RegLane<W> s;
    s.rd.init(pkt, size);
    s.err = 0;
    s.near = 0;
    s.max_pos = size * 8u + s.rd.bias; /* positions of the lane state are biased (RingRd) */

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
    s.set_upos(shift_pos + bs * 8u * (cpe ? 2u : 1u) * ns);
    const uint32_t chan_bits = cfg.bit_depth - 8u * bs + (cpe ? 1u : 0u);
    /* the 16- and 20-bit writers ignore the shift buffer (matrix.go:30,66) */
    const uint32_t sb = (cfg.bit_depth == 24 || cfg.bit_depth == 32) ? bs * 8u : 0u;
    const uint32_t n_it = wv.max_u32(LANES != 0 ? umax(ns, ns_other) : ns);
    if constexpr (LANES != 0) {
        static_assert(ROLE == ROLE_B && EC && WIDE_SEL == 0, "a predictor wave beside an entropy and a writer wave, narrow channels");
        if (cpe) {
            duo_phase_lanes_na<W, OUT_UTILE, LANES>(wv, na_u, bits, q, n_it, hdr_u, (hu >> 8) & 0xfu, chan_bits);
            duo_phase_lanes_na<W, OUT_STEREO, LANES>(wv, na_v, bits, q, n_it, hdr_v, (hv >> 8) & 0xfu, chan_bits);
        } else {
            duo_phase_lanes_na<W, OUT_MONO, LANES>(wv, na_u, bits, q, n_it, hdr_u, (hu >> 8) & 0xfu, chan_bits);
        }
        return 0;
    }
    /* the stager belongs to the wave that writes the PCM of the last channel (duo_emit_in_a) */
    const bool emit_a = !EC && !wide && duo_emit_in_a(cpe ? na_v : na_u, cpe);
    const bool writer = (EC && !wide) ? DO_C : (emit_a ? DO_A : DO_B);
    if (writer && live) wv.st_begin(out);

    /* ---- U (or the mono channel) ---- */
    s.mean = cfg.mb;
    s.zmode = 0;
    s.zq = 0xffffffffu;
    s.set_pb((cfg.pb * ((hu >> 5) & 7u)) / 4u); /* decoder.go:299 */
    if (DO_A) s.rd.start(wv, live ? s.pos : s.rd.bias);
    if constexpr (WIDE_SEL != 1) if (!wide) {
        if (cpe) duo_phase_na<W, OUT_UTILE, ROLE, false, true, false, EC>(wv, na_u, cfg, bits, s, size, ns, n_it, hdr_u, (hu >> 8) & 0xfu, chan_bits, mix_res, mix_sh, shift_pos, 0u);
        else duo_phase_na<W, OUT_MONO, ROLE, false, true, false, EC>(wv, na_u, cfg, bits, s, size, ns, n_it, hdr_u, (hu >> 8) & 0xfu, chan_bits, 0, 0, shift_pos, sb);
    }
    if constexpr (WIDE_SEL != 0) if (wide) {
        if (cpe) duo_phase_na<W, OUT_UTILE, ROLE, false, false, false, false, WM>(wv, na_u, cfg, bits, s, size, ns, n_it, hdr_u, (hu >> 8) & 0xfu, chan_bits, mix_res, mix_sh, shift_pos, 0u);
        else duo_phase_na<W, OUT_MONO, ROLE, false, false, false, false, WM>(wv, na_u, cfg, bits, s, size, ns, n_it, hdr_u, (hu >> 8) & 0xfu, chan_bits, 0, 0, shift_pos, sb);
    }
    uint32_t err_chan = 0;
    /* ---- V ---- */
    if (cpe) {
        const bool u_failed = s.err != 0;
        if (!u_failed && ((s.upos() >> 3) > size + 4u || (s.upos() >> 3) > size)) s.err = ST_MALFORMED; /* DynDecomp entry */
        const int32_t err_u = s.err;
        s.mean = cfg.mb;
        s.zmode = 0;
        s.zq = 0xffffffffu;
        s.set_pb((cfg.pb * ((hv >> 5) & 7u)) / 4u);
        if (DO_A) s.rd.start(wv, (live && s.err == 0) ? s.pos : s.rd.bias);
        if constexpr (WIDE_SEL != 0) if (wide)
            duo_phase_na<W, OUT_STEREO, ROLE, false, false, false, false, WM>(wv, na_v, cfg, bits, s, size, ns, n_it, hdr_v, (hv >> 8) & 0xfu, chan_bits, mix_res, mix_sh, shift_pos, sb);
        /* DEPTH_SEL 16 / 24 / 32: the caller's configuration has that sample width (24 stands for both 3-byte depths) */
        if constexpr (WIDE_SEL != 1) if (!wide) {
            if constexpr (DEPTH_SEL == 0 || DEPTH_SEL == 16) if (cfg.bit_depth == 16)
                duo_phase_na<W, OUT_STEREO, ROLE, true, true, false, EC>(wv, na_v, cfg, bits, s, size, ns, n_it, hdr_v, (hv >> 8) & 0xfu, chan_bits, mix_res, mix_sh, shift_pos, sb);
            if constexpr (DEPTH_SEL != 16) if (cfg.bit_depth != 16)
                duo_phase_na<W, OUT_STEREO, ROLE, false, true, DEPTH_SEL == 24, EC>(wv, na_v, cfg, bits, s, size, ns, n_it, hdr_v, (hv >> 8) & 0xfu, chan_bits, mix_res, mix_sh, shift_pos, sb);
        }
        if (err_u == 0 && s.err != 0) err_chan = 1;
    }
    if (writer && live) (void)wv.st_finish();
    if (!DO_A || !live) return 0;
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
