/*
 * alac_split.h — the split pipeline for packets with more than two channels (BASELINE config d).
 *
 * A packet of E elements is a serial chain E times as long as a stereo one if a single lane walks it
 * (decoder.go:142-203: element k+1 starts where the entropy stream of element k ends, and nothing in the
 * bitstream says where that is). The split pipeline cuts the chain:
 *
 *   1. scan     decode_wave<..., SCAN> (alac_wave.h): one lane per packet walks the packet with the lean
 *               Golomb loop only (no predictor, no PCM), records where every channel's entropy stream starts
 *               (ChanDesc) and settles the packet's status and frame count (PktDesc) — every error the
 *               reference can raise is raised here, in stream order;
 *   2. decode   decode_channel_task: one lane per (packet, channel), wave pairs (alac_duo.h) sorted by predictor
 *               order, int32 samples to the task's row through the LDS stager (coalesced 128-B lines);
 *   3. interleave  interleave_frame: one thread per (packet, frame): unmix pairs (matrix.go:40-41), shift-byte
 *               merge (matrix.go:129-132), escape elements straight from the bitstream (decoder.go:326-345,
 *               507-535), PCM bytes in frame order (coalesced both ways), zero fill of unwritten slots.
 *
 * Packets whose orders have no lean instantiation (17..30) take decode_wave whole (ROUTE_LEGACY).
 */
#ifndef ALAC_SPLIT_H
#define ALAC_SPLIT_H

#include "alac_duo.h"

namespace alac {

/* sort key of a channel task: order | wide << 5 (wide: chanBits > 23, plain 32-bit arithmetic) */
ALAC_DEV bool chan_is_narrow(const DevCfg& cfg, uint32_t chan_bits) {
    return chan_bits <= 23u && cfg.frame_length <= 65536u;
}
ALAC_DEV uint32_t chan_task_key(const DevCfg& cfg, const ChanDesc& d) {
    const uint32_t na = (d.info >> CD_NA_SHIFT) & 31u;
    const uint32_t chan_bits = (d.info >> CD_CHANBITS_SHIFT) & 63u;
    return na | (chan_is_narrow(cfg, chan_bits) ? 0u : 32u);
}
constexpr uint32_t NUM_TASK_KEYS = 64;
constexpr uint32_t TASK_NONE = 0xffffu;

/* samples of one channel, by the wave pair of alac_duo.h (ROLE: which of the two this caller is): every lane holds
 * a task with the same key (live = false: no task) */
template <class W, int ROLE>
ALAC_DEV void decode_channel_task(W& wv, const DevCfg& cfg, uint32_t key, bool live, const uint8_t* pkt, uint32_t size,
                                  uint32_t avail, const ChanDesc& d, int32_t* row) {
    constexpr bool DO_A = ROLE == ROLE_A || ROLE == ROLE_BOTH, DO_B = ROLE == ROLE_B || ROLE == ROLE_BOTH;
    const Bits bits{pkt, size, avail};
    const uint32_t na = key & 31u;
    const bool narrow = (key & 32u) == 0;
    RegLane<W> s;
    s.rd.init(pkt, size);
    s.err = 0;
    s.near = 0;
    s.max_pos = size * 8u + s.rd.bias;
    const uint32_t h = bits.get(d.hdr_pos, 16);
    const uint32_t mode = live ? (h >> 12) : 0u;
    const uint32_t den_shift = (h >> 8) & 0xfu;
    s.set_pb((cfg.pb * ((h >> 5) & 7u)) / 4u); /* decoder.go:299 */
    s.mean = cfg.mb;
    s.zmode = 0;
    s.zq = 0xffffffffu;
    s.set_upos(live ? d.ent_pos : 1u);
    const uint32_t ns = live ? d.ns : 0u;
    const uint32_t chan_bits = live ? ((d.info >> CD_CHANBITS_SHIFT) & 63u) : 16u;
    const uint32_t n_it = wv.max_u32(ns);
    if (DO_B && live) wv.st_begin(reinterpret_cast<uint8_t*>(row));
    if (DO_A) s.rd.start(wv, s.pos);
    if (narrow)
        duo_phase_na<W, OUT_RAW, ROLE, false, true>(wv, na, cfg, bits, s, size, ns, n_it, d.hdr_pos, den_shift, chan_bits, 0, 0u,
                                                    0u, 0u, mode);
    else
        duo_phase_na<W, OUT_RAW, ROLE, false, false>(wv, na, cfg, bits, s, size, ns, n_it, d.hdr_pos, den_shift, chan_bits, 0,
                                                     0u, 0u, 0u, mode);
    if (DO_B && live) (void)wv.st_finish();
}

/* PCM of frame i of one split packet, written at `frame` (num_channels * bps bytes; the kernel stages 256 frames
 * in LDS and copies them out as whole lines). rows: the packet's sample rows, row r at rows + r*row_stride. */
ALAC_DEV void interleave_frame(const DevCfg& cfg, const uint8_t* pkt, uint32_t size, uint32_t avail, const PktDesc& pd,
                               const ChanDesc* cd, const int32_t* rows, size_t row_stride, uint32_t i, uint8_t* frame) {
    const Bits bits{pkt, size, avail};
    const uint32_t num_chan = cfg.num_channels, bps = cfg.bps, depth = cfg.bit_depth;
    for (uint32_t slot = 0; slot < pd.nslots; ++slot) {
        const ChanDesc d = cd[slot];
        if (!(d.info & CD_VALID) || (d.info & CD_SECOND) || i >= d.ns) continue;
        const bool cpe = (d.info & CD_CPE) != 0, escape = (d.info & CD_ESCAPE) != 0;
        const uint32_t nch_e = cpe ? 2u : 1u;
        const uint32_t chan_bits = (d.info >> CD_CHANBITS_SHIFT) & 63u;
        const uint32_t out_chan = (d.info >> CD_OUTCHAN_SHIFT) & 7u;
        const uint32_t sb = (d.info >> CD_SB_SHIFT) & 31u;
        int32_t a, b = 0;
        if (escape) { /* decodeSCEEscape / decodeCPEEscape, decoder.go:326-345 / 507-535 */
            const uint32_t cs = 32u - chan_bits;
            a = sext_cs((int32_t)bits.get(d.hdr_pos + (i * nch_e) * chan_bits, chan_bits), cs);
            if (cpe) b = sext_cs((int32_t)bits.get(d.hdr_pos + (i * nch_e + 1u) * chan_bits, chan_bits), cs);
        } else {
            a = rows[(size_t)slot * row_stride + i];
            if (cpe) b = rows[(size_t)(slot + 1u) * row_stride + i];
        }
        int32_t l = a, r = 0;
        if (cpe) {
            const int32_t mix_res = (int32_t)(int8_t)(d.mix & 0xff);
            const uint32_t mix_sh = ((uint32_t)d.mix >> 8) & 31u;
            if (mix_res != 0) { /* matrix.go:40-41 */
                l = a + b - ((mix_res * b) >> mix_sh);
                r = l - b;
            } else {
                r = b;
            }
        }
        if (depth == 20) { /* matrix.go:77-78, 237 */
            l = (int32_t)((uint32_t)l << 4);
            r = (int32_t)((uint32_t)r << 4);
        }
        if (sb) { /* matrix.go:129-132, 266-268 */
            const uint32_t sp = d.shift_pos + i * nch_e * sb;
            l = (int32_t)((uint32_t)l << sb) | (int32_t)bits.get(sp, sb);
            if (cpe) r = (int32_t)((uint32_t)r << sb) | (int32_t)bits.get(sp + sb, sb);
        }
        uint8_t* dst = frame + out_chan * bps;
        store_le(dst, l, bps);
        if (cpe) store_le(dst + bps, r, bps);
    }
    /* DecodePacket hands back output[:n] of a zeroed buffer (decoder.go:120,127) */
    for (uint32_t sidx = 0; sidx < num_chan; ++sidx)
        if (i >= pd.written[sidx])
            for (uint32_t k = 0; k < bps; ++k) frame[sidx * bps + k] = 0;
}

/* The same frame for the layouts whose frames are whole dwords (NC channels x BPS bytes, NC * BPS % 4 == 0), built in
 * registers: slot s of an NC-channel stream always lands in output channel layout_offset(NC, s) (decoder.go:55-64),
 * so with the slots unrolled every byte position is a compile-time constant — no byte stores, no per-byte address
 * arithmetic. f[] receives the frame as little-endian dwords; channels nobody wrote stay zero (decoder.go:120,127). */
/* what a frame of the split packets needs from memory: per bitstream channel slot the sample (or, for the first slot
 * of a pair, both) and the 8-byte window on its shift values. Kept apart from the arithmetic so that alac_interleave
 * can ask for the next slice's values before it stores the current one (k_split.hip). */
template <int NC>
struct IlLoaded {
    int32_t a[NC], b[NC];
    uint64_t w[NC];
};

template <int NC, int BPS>
ALAC_DEV void interleave_load(const DevCfg& cfg, const uint8_t* pkt, uint32_t size, uint32_t avail, const PktDesc& pd,
                              const ChanDesc* cd, const int32_t* rows, size_t row_stride, uint32_t i, IlLoaded<NC>& L) {
    const Bits bits{pkt, size, avail};
#pragma unroll
    for (int slot = 0; slot < NC; ++slot) {
        L.a[slot] = L.b[slot] = 0;
        L.w[slot] = 0;
        if ((uint32_t)slot >= pd.nslots) continue;
        const ChanDesc d = cd[slot];
        if (!(d.info & CD_VALID) || (d.info & CD_SECOND) || i >= d.ns) continue;
        const bool cpe = (d.info & CD_CPE) != 0, escape = (d.info & CD_ESCAPE) != 0;
        const uint32_t nch_e = cpe ? 2u : 1u;
        const uint32_t chan_bits = (d.info >> CD_CHANBITS_SHIFT) & 63u;
        const uint32_t sb = (d.info >> CD_SB_SHIFT) & 31u;
        if (escape) { /* decodeSCEEscape / decodeCPEEscape, decoder.go:326-345 / 507-535 */
            const uint32_t cs = 32u - chan_bits;
            L.a[slot] = sext_cs((int32_t)bits.get(d.hdr_pos + (i * nch_e) * chan_bits, chan_bits), cs);
            if (cpe) L.b[slot] = sext_cs((int32_t)bits.get(d.hdr_pos + (i * nch_e + 1u) * chan_bits, chan_bits), cs);
        } else {
            L.a[slot] = rows[(size_t)slot * row_stride + i];
            if (cpe && slot + 1 < NC) L.b[slot] = rows[(size_t)(slot + 1) * row_stride + i];
        }
        /* matrix.go:129-132, 266-268: both shift values of a frame lie side by side (decoder.go:492-502) */
        if (sb) L.w[slot] = bits.window(d.shift_pos + i * nch_e * sb);
    }
}

#ifndef ALAC_LOAD4_I32 /* four int32 from a 16-byte aligned address: one global_load_dwordx4 on the GPU (alac_gpu.h) */
#define ALAC_LOAD4_I32(q, a, b, c, d) \
    do {                              \
        (a) = (q)[0];                 \
        (b) = (q)[1];                 \
        (c) = (q)[2];                 \
        (d) = (q)[3];                 \
    } while (0)
#endif
/* The same for FOUR consecutive frames i0 .. i0 + 3 (i0 a multiple of four) — round 4: alac_interleave is bound by the number of
 * its memory requests, not by their bytes (int16 rows halved the bytes and made it slower: experiments/rows16.txt), so a lane
 * asks for four frames' samples of a row in ONE 16-byte load (rows are 16-byte aligned, their stride a multiple of four) and
 * for the shift values of as many frames as a window holds: Bits::window has at least 57 stream bits, a frame's shift values
 * are nch * sb bits side by side. Rows are readable up to their stride; frames beyond a channel's count are not used
 * (interleave_build looks at ns). */
template <int NC, int BPS>
ALAC_DEV void interleave_load4(const DevCfg& cfg, const uint8_t* pkt, uint32_t size, uint32_t avail, const PktDesc& pd,
                               const ChanDesc* cd, const int32_t* rows, size_t row_stride, uint32_t i0, IlLoaded<NC> (&L)[4]) {
    const Bits bits{pkt, size, avail};
#pragma unroll
    for (int slot = 0; slot < NC; ++slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            L[j].a[slot] = L[j].b[slot] = 0;
            L[j].w[slot] = 0;
        }
        if ((uint32_t)slot >= pd.nslots) continue;
        const ChanDesc d = cd[slot];
        if (!(d.info & CD_VALID) || (d.info & CD_SECOND) || i0 >= d.ns) continue;
        const bool cpe = (d.info & CD_CPE) != 0, escape = (d.info & CD_ESCAPE) != 0;
        const uint32_t nch_e = cpe ? 2u : 1u;
        const uint32_t chan_bits = (d.info >> CD_CHANBITS_SHIFT) & 63u;
        const uint32_t sb = (d.info >> CD_SB_SHIFT) & 31u;
        if (escape) { /* decodeSCEEscape / decodeCPEEscape, decoder.go:326-345 / 507-535 */
            const uint32_t cs = 32u - chan_bits;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t i = i0 + (uint32_t)j;
                if (i >= d.ns) continue;
                L[j].a[slot] = sext_cs((int32_t)bits.get(d.hdr_pos + (i * nch_e) * chan_bits, chan_bits), cs);
                if (cpe) L[j].b[slot] = sext_cs((int32_t)bits.get(d.hdr_pos + (i * nch_e + 1u) * chan_bits, chan_bits), cs);
            }
        } else {
            ALAC_LOAD4_I32(rows + (size_t)slot * row_stride + i0, L[0].a[slot], L[1].a[slot], L[2].a[slot], L[3].a[slot]);
            if (cpe && slot + 1 < NC)
                ALAC_LOAD4_I32(rows + (size_t)(slot + 1) * row_stride + i0, L[0].b[slot], L[1].b[slot], L[2].b[slot], L[3].b[slot]);
        }
        if (sb) { /* matrix.go:129-132, 266-268: both shift values of a frame lie side by side (decoder.go:492-502) */
            const uint32_t step = nch_e * sb;
            const uint32_t sp = d.shift_pos + i0 * step;
            if (step <= 14u) { /* four frames in one window */
                const uint64_t w = bits.window(sp);
#pragma unroll
                for (int j = 0; j < 4; ++j) L[j].w[slot] = w << ((uint32_t)j * step);
            } else if (step <= 28u) { /* two and two */
                const uint64_t w0 = bits.window(sp), w1 = bits.window(sp + 2u * step);
                L[0].w[slot] = w0;
                L[1].w[slot] = w0 << step;
                L[2].w[slot] = w1;
                L[3].w[slot] = w1 << step;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) L[j].w[slot] = bits.window(sp + (uint32_t)j * step);
            }
        }
    }
}

template <int NC, int BPS>
ALAC_DEV void interleave_build(const DevCfg& cfg, const PktDesc& pd, const ChanDesc* cd, uint32_t i, const IlLoaded<NC>& L,
                               uint32_t (&f)[NC * BPS / 4]) {
    static_assert((NC * BPS) % 4 == 0, "whole dwords only");
    constexpr uint32_t MASK = BPS == 4 ? 0xffffffffu : ((1u << (8 * (BPS & 3))) - 1u);
#pragma unroll
    for (int k = 0; k < NC * BPS / 4; ++k) f[k] = 0;
    /* later elements overwrite earlier ones where a (non-standard) element order makes them meet, as the reference's
     * sequential writes do: replace, not OR */
    auto put = [&](auto off_c, int32_t v) {
        constexpr int OFF = decltype(off_c)::value; /* byte offset inside the frame */
        constexpr int D = OFF / 4, SH = 8 * (OFF % 4);
        constexpr uint32_t M0 = MASK << SH;
        const uint32_t u = (uint32_t)v & MASK;
        f[D] = (f[D] & ~M0) | (u << SH);
        if (SH + 8 * BPS > 32) {
            constexpr uint32_t M1 = SH ? (MASK >> ((32 - SH) & 31)) : 0u;
            f[D + 1 < NC * BPS / 4 ? D + 1 : D] = (f[D + 1 < NC * BPS / 4 ? D + 1 : D] & ~M1) | (u >> ((32 - SH) & 31));
        }
    };
    auto emit_at = [&](int chan, int32_t v) { /* chan is a constant after unrolling: the switch folds */
        switch (chan) {
            case 0: put(std::integral_constant<int, 0>{}, v); break;
            case 1: put(std::integral_constant<int, (NC > 1 ? 1 : 0) * BPS>{}, v); break;
            case 2: put(std::integral_constant<int, (NC > 2 ? 2 : 0) * BPS>{}, v); break;
            case 3: put(std::integral_constant<int, (NC > 3 ? 3 : 0) * BPS>{}, v); break;
            case 4: put(std::integral_constant<int, (NC > 4 ? 4 : 0) * BPS>{}, v); break;
            case 5: put(std::integral_constant<int, (NC > 5 ? 5 : 0) * BPS>{}, v); break;
            case 6: put(std::integral_constant<int, (NC > 6 ? 6 : 0) * BPS>{}, v); break;
            case 7: put(std::integral_constant<int, (NC > 7 ? 7 : 0) * BPS>{}, v); break;
            default: break; /* a pair in the last slot: the scan calls that malformed, such packets do not get here */
        }
    };
    const uint32_t depth = cfg.bit_depth;
#pragma unroll
    for (int slot = 0; slot < NC; ++slot) {
        constexpr uint32_t tbl[8] = {0x0u, 0x10u, 0x102u, 0x3102u, 0x43102u, 0x354102u, 0x3654102u, 0x35410762u};
        const int out_chan = (int)((tbl[NC - 1] >> (4 * slot)) & 0xfu);
        if ((uint32_t)slot >= pd.nslots) continue;
        const ChanDesc d = cd[slot];
        if (!(d.info & CD_VALID) || (d.info & CD_SECOND) || i >= d.ns) continue;
        const bool cpe = (d.info & CD_CPE) != 0;
        const uint32_t sb = (d.info >> CD_SB_SHIFT) & 31u;
        const int32_t a = L.a[slot], b = L.b[slot];
        int32_t l = a, r = 0;
        if (cpe) {
            const int32_t mix_res = (int32_t)(int8_t)(d.mix & 0xff);
            const uint32_t mix_sh = ((uint32_t)d.mix >> 8) & 31u;
            if (mix_res != 0) { /* matrix.go:40-41 */
                l = a + b - ((mix_res * b) >> mix_sh);
                r = l - b;
            } else {
                r = b;
            }
        }
        if (depth == 20) { /* matrix.go:77-78, 237 */
            l = (int32_t)((uint32_t)l << 4);
            r = (int32_t)((uint32_t)r << 4);
        }
        if (sb) {
            const uint64_t w = L.w[slot];
            l = (int32_t)((uint32_t)l << sb) | (int32_t)(uint32_t)(w >> (64u - sb));
            if (cpe) r = (int32_t)((uint32_t)r << sb) | (int32_t)(uint32_t)((w << sb) >> (64u - sb));
        }
        emit_at(out_chan, l);
        if (cpe) emit_at(out_chan + 1, r); /* R goes right behind L (matrix.go:31-32), whatever the layout says about the next slot */
    }
}

/* load + build in one go (the host simulation, and whoever has no use for the split) */
template <int NC, int BPS>
ALAC_DEV void interleave_frame_packed(const DevCfg& cfg, const uint8_t* pkt, uint32_t size, uint32_t avail, const PktDesc& pd,
                                      const ChanDesc* cd, const int32_t* rows, size_t row_stride, uint32_t i,
                                      uint32_t (&f)[NC * BPS / 4]) {
    IlLoaded<NC> L;
    interleave_load<NC, BPS>(cfg, pkt, size, avail, pd, cd, rows, row_stride, i, L);
    interleave_build<NC, BPS>(cfg, pd, cd, i, L, f);
}

} /* namespace alac */
#endif
