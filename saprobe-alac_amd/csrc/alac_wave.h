/*
 * alac_wave.h — the ALAC packet decoder as executed by one wavefront: 64 lanes, one packet per lane.
 *
 * Design. Lanes run in lock step: every lane yields exactly one residual per loop
 * iteration (a zero run is a per-lane countdown, not a burst), so the whole wave sits at the same
 * sample index i. That makes the U-channel hand-off row scr[i][lane] one coalesced 256-B access, lets
 * the wave stage PCM through LDS and write it back as whole 128-B lines (W::st_*), and confines
 * divergence to rare paths (escape codes, zero-run starts, escape elements). Control flow is
 * wave-uniform (loop bounds are wave maxima, bodies are predicated per lane) because the LDS flush is a
 * collective. This file holds the whole-packet form (decode_wave): any depth, any element mix, orders 0..31 on NA
 * register taps plus a per-lane fall-back tile. Since round 1's wave pair it runs only as the SCAN pass of irregular
 * packets (entropy only: status, frame count, channel descriptors, residual rows) and as the decoder of the few
 * packets the scan routes ROUTE_LEGACY; regular packets, sorted by (numU, numV, width), take alac_duo.h.
 *
 * Nothing here is a port: the reference decodes one packet at a time with whole-block passes
 * (DynDecomp over the block, then UnpcBlock, then Write*); this fuses them per sample and keeps the
 * Golomb state, a 3-dword bitstream cache, the predictor history and coefficients in registers.
 *
 * Bit-exactness contract: identical PCM bytes, frame count and status word to the reference
 * (mycophonic/saprobe-alac) for every input, including Go's shift/wrap semantics. Reference lines are
 * cited at each step (paths relative to the reference tree).
 *
 * The file is plain C++ templated on a wave policy W: the k_*.hip units instantiate it with the gfx950
 * policy of alac_gpu.h (LDS stager, ballots, DPP reductions); tests/host_sim instantiates it with a one-lane policy
 * and g++ to check the LOGIC against the oracle where no GPU exists. It is not a CPU decode path of the
 * product: libalacgpu.so contains no host decoder.
 */
#ifndef ALAC_WAVE_H
#define ALAC_WAVE_H

#include <stdint.h>

#include "../../include/alacgpu.h"

#ifndef ALAC_DEV
#error "define ALAC_DEV before including alac_wave.h"
#endif
#ifndef ALAC_MUL24
/* exact when both operands fit 24-bit signed: v_mul_i32_i24 / v_mad_i32_i24 on the GPU */
#define ALAC_MUL24(a, b) ((int32_t)(a) * (int32_t)(b))
#endif

namespace alac {

struct DevCfg {
    uint32_t frame_length;
    uint32_t bit_depth;
    uint32_t num_channels;
    uint32_t pb, mb, kb;
    uint32_t bps;       /* BytesPerSample, internal/alac/format.go:23-34 */
    uint32_t aligned16; /* PCM base and stride are multiples of 16: the LDS stager may be used */
};

/* channelLayoutOffsets (decoder.go:55-64) packed 4 bits per entry, entry k at bits 4k */
ALAC_DEV uint32_t layout_offset(uint32_t num_chan, uint32_t chan_idx) {
    const uint32_t tbl[8] = {0x0u, 0x10u, 0x102u, 0x3102u, 0x43102u, 0x354102u, 0x3654102u, 0x35410762u};
    return (tbl[num_chan - 1] >> (4 * chan_idx)) & 0xfu;
}

/* ---- Go shift semantics (SURVEY.md §8a trap 1): counts >= 32 give 0 / sign fill ----------------------- */
ALAC_DEV uint32_t go_shl(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x << n; }
ALAC_DEV uint32_t go_shr(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x >> n; }
/* (x << chanShift) >> chanShift, predictor.go:68,78,130 */
ALAC_DEV int32_t sext_cs(int32_t x, uint32_t cs) { return cs >= 32 ? 0 : (int32_t)((uint32_t)x << cs) >> cs; }
/* signOfInt, predictor.go:35-39 */
ALAC_DEV int32_t sign_of(int32_t v) { return (v > 0) - (v < 0); }
ALAC_DEV uint32_t clz32(uint32_t x) { return x ? (uint32_t)__builtin_clz(x) : 32u; }
ALAC_DEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
ALAC_DEV uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

/* ---- stateless bit access (headers, escape elements, shift bytes, rare codes) ---------------------------
 * 64-bit big-endian window whose MSB is stream bit `pos` (>= 57 valid bits). Packets lie DENSELY in the blob (an
 * mdat as it is in the file, internal/mp4/mp4.go:382-420): what follows a packet is its neighbour, not padding. The
 * reference sees every packet followed by 4 zero bytes (bitbuffer.go:33) and panics beyond them, so bytes from `size`
 * on read as ZERO here, and nothing at or beyond `avail` (the bytes of the blob from the packet's start) is touched.
 * The byte offset is clamped to size+8: every consumer of a position that far out raises a status before the data
 * could matter. */
template <bool SMALL>
struct BitsT {
    const uint8_t* p;
    uint32_t size;
    uint32_t avail; /* readable bytes from p (>= size), capped at 2^32-1; the readers never need it: they stay inside size */

    ALAC_DEV uint64_t window(uint32_t pos) const {
        const uint32_t b = umin(pos >> 3, size + 8u);
        uint64_t raw;
        if (!SMALL || size >= 8u) { /* SMALL = false: the caller knows size >= 8 (regular packets: >= 12) */
            /* no branch on where the window lies: the load is pulled back so that it ends with the packet's last byte
             * and the bytes it was pulled back over are shifted out again — zeros come in for everything behind the
             * packet (memory order: the first byte is the low one) */
            const uint32_t bb = umin(b, size - 8u);
            const uint32_t d = b - bb; /* 0..16 */
            __builtin_memcpy(&raw, p + bb, 8);
            raw = d >= 8u ? 0ull : raw >> (8u * d);
        } else if (size != 0u) {
            /* a packet of fewer than 8 bytes lies in at most three aligned dwords, each of which holds a packet byte
             * (or is fetched as the last one that does): assemble them, drop the bytes in front, keep `size` bytes */
            const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p - mis);
            const uint32_t last = (mis + size - 1u) >> 2;
            const uint64_t lo = (uint64_t)q[0] | ((uint64_t)q[umin(1u, last)] << 32);
            const uint64_t hi = q[umin(2u, last)];
            uint64_t v = mis ? (lo >> (8u * mis)) | (hi << (64u - 8u * mis)) : lo;
            v &= (1ull << (8u * size)) - 1ull;
            raw = b >= 8u ? 0ull : v >> (8u * b);
        } else {
            raw = 0;
        }
        return __builtin_bswap64(raw) << (pos & 7u);
    }
    /* the same without the end handling, for windows that lie wholly inside the packet when the lane is still
     * decoding (the shift values of a regular packet: classify_regular keeps 10 bytes of entropy stream behind them).
     * A lane that has run out of frames keeps being asked (lock step): its offset is held inside its packet, what it
     * reads is not used. Needs size >= 8. */
    ALAC_DEV uint64_t window_raw(uint32_t pos) const {
        uint64_t raw;
        __builtin_memcpy(&raw, p + umin(pos >> 3, size - 8u), 8);
        return __builtin_bswap64(raw) << (pos & 7u);
    }
    /* 12 bytes from byte offset `off` (same rule: held inside the packet; needs size >= 12), as three big-endian
     * dwords: bits [8*off, 8*off + 96) of the stream */
    ALAC_DEV void load12(uint32_t off, uint32_t& a, uint32_t& b, uint32_t& c) const {
        uint32_t raw[3];
        __builtin_memcpy(raw, p + umin(off, size - 12u), 12);
        a = __builtin_bswap32(raw[0]);
        b = __builtin_bswap32(raw[1]);
        c = __builtin_bswap32(raw[2]);
    }
    /* n bits (0..32) at pos: BitBuffer.Read / ReadSmall / ReadOne all reduce to this (bitbuffer.go:55-96) */
    ALAC_DEV uint32_t get(uint32_t pos, uint32_t n) const {
        return n == 0 ? 0u : (uint32_t)(window(pos) >> (64u - n));
    }
    /* where the Go code panics on a slice bound (fresh decoder: len(Buf) = size+4):
     * Read: Buf[Pos:Pos+3] (bitbuffer.go:58); ReadSmall: Buf[Pos:Pos+2] (:75) */
    ALAC_DEV bool read_panics(uint32_t pos) const { return (pos >> 3) > size + 1u; }
    ALAC_DEV bool read_small_panics(uint32_t pos) const { return (pos >> 3) > size + 2u; }
    ALAC_DEV bool past_end(uint32_t pos) const { return (pos >> 3) >= size; } /* bitbuffer.go:115 */
};
using Bits = BitsT<true>;

/* Advance (bitbuffer.go:99-103): BitIdx is uint32 and wraps; positions far past the packet are clamped
 * (every later use of them errors the same way wherever they are). */
ALAC_DEV uint32_t advance(uint32_t pos, uint32_t nbits) {
    uint64_t np = (uint64_t)(pos & ~7u) + (uint64_t)(uint32_t)((pos & 7u) + nbits);
    return np > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)np;
}

/* ---- the whole-packet decoder's bit reader: three big-endian dwords of the stream cached in registers ------
 * w0,w1 hold stream dwords widx, widx+1; w2 (dword widx+2) is loaded one step ahead so its latency hides
 * behind ~3 samples of work. window() is one 64-bit funnel shift. A step consumes < 32 bits, so the
 * cache slides by at most one dword per call of slide(). Dwords are fetched whole and aligned: one that holds at
 * least one packet byte lies in the blob's pages; bytes from end_b on (the neighbour packet) are cleared, dwords
 * wholly behind the packet are not fetched at all (dense blob, see Bits). */
struct FastRd {
    const uint32_t* base; /* packet start rounded down to a dword */
    uint32_t bias;        /* stream bit 0 is bit `bias` of base[0] */
    uint32_t end_b;       /* first byte, counted from base, that is not packet data */
    uint32_t w0, w1, w2, widx;

    ALAC_DEV uint32_t ld(uint32_t idx) const {
        /* branch-free: fetch the dword (or, behind the packet, the last one that holds packet bytes), keep what is
         * packet data. end_b >= 1 for a lane with a packet; a lane without one reads base[0] and keeps nothing. */
        const uint32_t last = end_b ? (end_b - 1u) >> 2 : 0u;
        const uint32_t v = base[umin(idx, last)];
        const uint32_t lo = umin(idx, 0x3fffffffu) * 4u;
        const uint32_t nb = end_b > lo ? umin(end_b - lo, 4u) : 0u; /* packet bytes in dword idx */
        const uint32_t keep = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
        return __builtin_bswap32(v & keep);
    }
    ALAC_DEV void seek(uint32_t pos) {
        widx = (pos + bias) >> 5;
        w0 = ld(widx);
        w1 = ld(widx + 1);
        w2 = ld(widx + 2);
    }
    ALAC_DEV uint32_t window(uint32_t pos) const {
        const uint32_t r = (pos + bias) & 31u;
        return (uint32_t)(((((uint64_t)w0) << 32) | w1) << r >> 32);
    }
    ALAC_DEV void slide(uint32_t pos) {
        if (((pos + bias) >> 5) != widx) {
            w0 = w1;
            w1 = w2;
            ++widx;
            w2 = ld(widx + 2);
        }
    }
};

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

/* Orders the unrolled reference predictors handle with int32 coefficients (predictor.go:81-93); every
 * other order takes unpcBlockGeneral and wraps its coefficients to int16 at each update (:664,:675). */
ALAC_DEV bool order_wraps16(uint32_t na) { return !(na == 4 || na == 5 || na == 6 || na == 8); }

/* ---- descriptors of the split pipeline (alac_split.h): written by the scan pass, one per bitstream channel ---- */
struct ChanDesc {
    uint32_t hdr_pos;   /* compressed: bit position of the channel's predictor header; escape: first raw sample */
    uint32_t ent_pos;   /* compressed: first bit of the channel's entropy stream */
    uint32_t ns;        /* numSamples of the element */
    uint32_t shift_pos; /* first bit of the element's shift block */
    uint32_t info;      /* CD_* fields below */
    int32_t mix;        /* mixRes (low 8 bits, signed) | mixBits clamped to 31 << 8 */
    uint32_t pad0, pad1;
};
enum {
    CD_VALID = 1u << 0, CD_ESCAPE = 1u << 1, CD_CPE = 1u << 2, CD_SECOND = 1u << 3,
    CD_CHANBITS_SHIFT = 4,  /* 6 bits: 0..33 */
    CD_OUTCHAN_SHIFT = 10,  /* 3 bits: output slot of THIS channel */
    CD_SB_SHIFT = 13,       /* 5 bits: shift bits merged into the PCM (0, 8, 16) */
    CD_NA_SHIFT = 18,       /* 5 bits */
    CD_MODE = 1u << 23,
};
struct PktDesc {
    int32_t status;
    uint32_t frames;
    uint32_t nslots;     /* bitstream channels described */
    uint32_t route;      /* ROUTE_* */
    uint32_t written[8]; /* frames written per output slot */
};
enum { ROUTE_NONE = 0, ROUTE_SPLIT = 1, ROUTE_LEGACY = 2 };

/* Golomb-only pass over one channel (defined in alac_regular.h): advances pos to the end of the entropy stream */
template <class W, class B>
ALAC_DEV void scan_channel(W& wv, const DevCfg& cfg, const B& bits, const uint8_t* pkt, uint32_t size, bool go,
                           uint32_t& pos, uint32_t ns, uint32_t pb_local, uint32_t chan_bits, int32_t& err,
                           int32_t* res_row);

/* ------------------------------------------------------------------------------------------------------
 * decode_wave<W, NA, WRAP>: every lane of the wave calls this with its own packet (live = false for lanes
 * without one). W is the wave policy:
 *   bool     W::any(bool)            wave-wide OR
 *   uint32_t W::max_u32(uint32_t)    wave-wide max
 *   void     W::st_begin(out)        this lane's PCM slot for the LDS stager
 *   void     W::st_push(uint32_t)    append one little-endian dword of PCM to this lane's row
 *   void     W::st_step()            collective: write out rows that completed a 128-B chunk
 *   uint32_t W::st_finish()          per lane: write the unflushed tail, return dwords written in total
 *   int32_t* W::u_row(i)             &scratch[i][lane]: U hand-off tile
 *   int32_t* W::g_slot(k)            &fallback[k][lane]: general-predictor state (orders outside the class)
 * NA = predictor taps held in registers; WRAP = compile the per-lane int16 coefficient wrap.
 * Returns the status word; *frames_out = numSamples of the last element (decoder.go:206).
 * ------------------------------------------------------------------------------------------------------ */
template <class W, int NA, bool WRAP, bool SCAN = false>
ALAC_DEV int32_t decode_wave(W& wv, const DevCfg& cfg, bool live, const uint8_t* pkt, uint32_t size, uint32_t avail,
                             uint8_t* out, uint32_t* frames_out, ChanDesc* cd = nullptr, PktDesc* pd = nullptr,
                             int32_t* res_rows = nullptr, size_t res_stride = 0) {
    /* res_rows (SCAN, more than two channels): this packet's sample rows; the scan leaves every compressed channel's
     * residuals in the row of its slot for the predictor pass (wave-uniform: null for all lanes or for none) */
    /* SCAN: walk the packet exactly like a decode (same errors in the same order) but only find where every
     * channel's entropy stream starts and ends (scan_channel), describe the channels in cd[0..7] and the packet
     * in *pd; no prediction, no PCM. */
    bool legacy = false;
    const Bits bits{pkt, size, avail};
    const uint32_t num_chan = cfg.num_channels;
    const uint32_t bps = cfg.bps;
    const uint32_t frame_stride = num_chan * bps;
    const uint32_t depth = cfg.bit_depth;
    const uint32_t wb = go_shl(1u, cfg.kb) - 1u; /* SetAGParams golomb.go:60 */

    FastRd rd;
    {
        /* pointer arithmetic, not an integer round trip: the compiler keeps the global address space */
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(pkt) & 3u);
        rd.base = reinterpret_cast<const uint32_t*>(pkt - mis);
        rd.bias = mis * 8u;
        rd.end_b = size ? mis + size : 0u;
    }
    rd.w0 = rd.w1 = rd.w2 = rd.widx = 0;

    uint32_t pos = 0;
    uint32_t num_samples = cfg.frame_length; /* decoder.go:136 */
    uint32_t chan_idx = 0;
    uint32_t written[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* frames written per output channel slot */
    int32_t st = 0;
    bool walking = live;

    while (wv.any(walking)) {
        /* ================= phase A (per lane): walk tags to the next audio element ======================== */
        bool has = false, cpe = false, escape = false, use_shift = false, staged = false;
        uint32_t ctx = 0, nch_e = 0, out_chan = 0, ns = 0, chan_bits = 0, chan_shift = 0, shift_bits = 0;
        uint32_t hdr_pos = 0, shift_pos = 0, data_pos = 0, mix_sh = 0;
        int32_t mix_res = 0;
        while (walking && !has) {
            /* ---- element dispatch, decoder.go:142-203 */
            if (bits.past_end(pos)) {
                st = ALACGPU_STATUS(ST_OVERRUN, 0, 0);
                walking = false;
                break;
            }
            const uint32_t tag = bits.get(pos, 3);
            pos += 3;
            if (tag == 2 || tag == 5) { /* CCE / PCE, decoder.go:179-180 */
                st = ALACGPU_STATUS(ST_UNSUPPORTED, 0, 0);
                walking = false;
                break;
            }
            if (tag == 4) { /* skipDSE, decoder.go:555-574 */
                const uint32_t align = bits.get(pos + 4, 1);
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
                    walking = false;
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
                    walking = false;
                }
                continue;
            }
            if (tag == 7) { /* END, decoder.go:192-195 */
                walking = false;
                break;
            }
            cpe = tag == 1;
            if (cpe && chan_idx + 2 > num_chan) { /* decoder.go:163-165 */
                walking = false;
                break;
            }
            ctx = cpe ? ALACGPU_CTX_CPE : ALACGPU_CTX_SCE;
            nch_e = cpe ? 2u : 1u;
            out_chan = layout_offset(num_chan, chan_idx);
            int32_t e = 0;
            if (out_chan + nch_e > num_chan) {
                /* a pair that does not fit the frame: the reference writes outside the frame (and panics on
                 * a full frame); oracle and kernel both report it as malformed */
                e = ST_MALFORMED;
            } else if (bits.read_small_panics(pos) || bits.read_panics(pos + 4)) {
                e = ST_MALFORMED; /* ReadSmall(4) instance tag, Read(12) unused: decoder.go:213-216 */
            } else if (bits.get(pos + 4, 12) != 0) {
                e = ST_HEADER; /* decoder.go:217-219 / 356-359 */
            } else if (bits.read_panics(pos + 16)) {
                e = ST_MALFORMED;
            }
            uint32_t bytes_shifted = 0;
            if (!e) {
                const uint32_t hdr = bits.get(pos + 16, 4); /* decoder.go:221-229 */
                pos += 20;
                bytes_shifted = (hdr >> 1) & 3u;
                escape = (hdr & 1u) != 0;
                if (bytes_shifted == 3) {
                    e = ST_SHIFT;
                } else {
                    chan_bits = depth - bytes_shifted * 8u + (cpe ? 1u : 0u);
                    ns = num_samples;
                    bool bad = false;
                    if (hdr >> 3) { /* partial frame, decoder.go:232-235 */
                        bad = bits.read_panics(pos) || bits.read_panics(pos + 16);
                        ns = bits.get(pos, 32);
                        pos += 32;
                    }
                    if (!escape) {
                        /* decodeSCECompressed / decodeCPECompressed field walk, decoder.go:272-293 / 421-457 */
                        const int32_t mix_bits = (int32_t)bits.get(pos, 8);
                        mix_res = (int32_t)(int8_t)bits.get(pos + 8, 8);
                        mix_sh = (uint32_t)mix_bits > 31u ? 31u : (uint32_t)mix_bits; /* >> by >= 32 sign-fills */
                        hdr_pos = pos + 16;
                        uint32_t q = hdr_pos, last_read = pos + 8;
                        for (uint32_t c = 0; c < nch_e; ++c) {
                            const uint32_t num = bits.get(q + 11, 5);
                            last_read = num ? q + 16u + (num - 1u) * 16u : q + 8u;
                            q += 16u + 16u * num;
                        }
                        /* header reads are sequential and can only panic: test the last one */
                        bad = bad || bits.read_panics(last_read);
                        shift_pos = q;
                        pos = q;
                        if (bytes_shifted != 0) pos = advance(pos, bytes_shifted * 8u * nch_e * ns);
                        /* DynDecomp entry: input := Buf[Pos:] (golomb.go:149), predCoefs[:numSamples] (:155) */
                        bad = bad || (pos >> 3) > size + 4u || ns > cfg.frame_length;
                        use_shift = bytes_shifted != 0 && (depth == 24 || depth == 32);
                        shift_bits = bytes_shifted * 8u;
                    } else {
                        mix_res = 0;
                        if (cpe) chan_bits = depth; /* decoder.go:388 */
                        data_pos = pos;
                        /* mixU[:numSamples:numSamples] (decoder.go:328,509); the raw-sample Read()s walk
                         * forward, so only the last one can be the first to panic */
                        bad = bad || ns > cfg.frame_length;
                        if (!bad && ns != 0) {
                            const uint32_t last_w = chan_bits > 16 ? chan_bits - 16u : chan_bits;
                            bad = bits.read_panics(pos + nch_e * ns * chan_bits - last_w);
                        }
                        pos = advance(pos, nch_e * ns * chan_bits);
                    }
                    if (bad) e = ST_MALFORMED;
                }
            }
            if (e) {
                st = ALACGPU_STATUS(e, ctx, 0);
                walking = false;
                break;
            }
            chan_shift = 32u - chan_bits; /* wraps for chanBits 33, predictor.go:46 */
            has = true;
        }
        /* PCM of an element that covers the whole frame is one contiguous stream: stage it through LDS */
        staged = !SCAN && has && cfg.aligned16 != 0 && nch_e == num_chan;
        if (staged) wv.st_begin(out);

        /* ================= phase B (wave-uniform): channels of the element, U then V ====================== */
        int32_t err = 0;
        uint32_t err_chan = 0;
        uint64_t pk_acc = 0; /* byte packer for the staged path */
        uint32_t pk_n = 0;
        const uint32_t nch_max = wv.max_u32(has ? nch_e : 0u);
        for (uint32_t c = 0; c < nch_max; ++c) {
            const bool part = has && c < nch_e && err == 0;
            const bool last_chan = c + 1 == nch_e;
            /* ---- per-channel header, decoder.go:275-286 */
            uint32_t mode = 0, den_shift = 0, na = 0, pb_local = 0;
            int32_t coef[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) coef[j] = 0;
            bool fast = true; /* this channel's order runs on the register taps of this variant */
            if (part && !escape) {
                const uint32_t h = bits.get(hdr_pos, 16);
                mode = h >> 12;
                den_shift = (h >> 8) & 0xfu;
                pb_local = (cfg.pb * ((h >> 5) & 7u)) / 4u; /* decoder.go:299 */
                na = h & 0x1fu;
                fast = na == 0 || na == 31 || (na <= (uint32_t)NA && (WRAP || !order_wraps16(na)));
                if (SCAN) {
                    /* no lean instantiation for orders 17..30; with one or two channels only escape elements take
                     * the split pipeline (no sample rows are kept for them), compressed ones go to decode_wave */
                    legacy = legacy || (na > 16 && na != 31) || num_chan <= 2;
                } else if (fast) {
#pragma unroll
                    for (int j = 0; j < NA; ++j)
                        if ((uint32_t)j < na) coef[j] = (int32_t)(int16_t)bits.get(hdr_pos + 16u + 16u * (uint32_t)j, 16);
                } else {
                    for (uint32_t j = 0; j < na; ++j)
                        *wv.g_slot(32u + j) = (int32_t)(int16_t)bits.get(hdr_pos + 16u + 16u * j, 16);
                }
                hdr_pos += 16u + 16u * na;
                if ((pos >> 3) > size + 4u || (ns != 0 && (pos >> 3) > size)) {
                    err = ST_MALFORMED; /* Buf[Pos:] out of range, or maxPos wrapped: first read32bit panics */
                    err_chan = c;
                }
            }
            if (SCAN && part && cd) {
                ChanDesc d;
                d.hdr_pos = escape ? data_pos : hdr_pos - (16u + 16u * na);
                d.ent_pos = pos;
                d.ns = ns;
                d.shift_pos = shift_pos;
                d.info = CD_VALID | (escape ? CD_ESCAPE : 0u) | (cpe ? CD_CPE : 0u) | (c ? CD_SECOND : 0u) |
                         (chan_bits << CD_CHANBITS_SHIFT) | ((out_chan + c) << CD_OUTCHAN_SHIFT) |
                         ((use_shift ? shift_bits : 0u) << CD_SB_SHIFT) | (na << CD_NA_SHIFT) | (mode ? CD_MODE : 0u);
                d.mix = (int32_t)((uint32_t)(mix_res & 0xff) | (mix_sh << 8));
                d.pad0 = d.pad1 = 0;
                cd[chan_idx + c] = d;
            }
            const bool run = part && err == 0;
            const bool wrap16 = WRAP && order_wraps16(na);
            const int32_t den_half = den_shift ? (int32_t)(1u << (den_shift - 1u)) : 0;
            const uint32_t max_pos = size * 8u;
            /* 24-bit multiplier eligibility: |coef| < 2^17 and |diff| < 2^24 */
            const bool narrow = chan_bits <= 23u && cfg.frame_length <= 65536u;

            uint32_t mean = cfg.mb, zmode = 0, zrem = 0;
            int32_t hist[NA + 1]; /* hist[j] = out[i-1-j] */
#pragma unroll
            for (int j = 0; j <= NA; ++j) hist[j] = 0;
            int32_t dprev = 0; /* delta pre-pass state (mode != 0) */
            if (SCAN) {
                int32_t e2 = 0;
                scan_channel(wv, cfg, bits, pkt, size, run && !escape && ns != 0, pos, ns, pb_local, chan_bits, e2,
                             res_rows ? res_rows + (size_t)umin(chan_idx + c, 7u) * res_stride : nullptr);
                if (run && e2) {
                    err = e2;
                    err_chan = c;
                }
            } else if (run && !escape && ns != 0) {
                rd.seek(pos);
            }

            const uint32_t n_it = SCAN ? 0u : wv.max_u32(run ? ns : 0u);
            for (uint32_t i = 0; i < n_it; ++i) {
                const bool on = run && i < ns && err == 0;
                if (on) {
                    int32_t u_pre = 0;
                    if (cpe && last_chan) u_pre = *wv.u_row(i); /* issued early: latency hides behind the decode */
                    int32_t o = 0;
                    if (!escape) {
                        /* ---- one residual: DynDecomp, golomb.go:167-247 ------------------------------------ */
                        int32_t del = 0;
                        if (zrem != 0) {
                            --zrem; /* inside a zero run (golomb.go:236-239) */
                        } else if (pos >= max_pos) {
                            err = ST_OVERRUN; /* golomb.go:168-170 */
                            err_chan = c;
                        } else {
                            uint32_t m = mean >> 9;
                            const uint32_t k = umin(31u - clz32(m + 3u), cfg.kb);
                            m = go_shl(1u, k) - 1u;
                            /* The reference's window is read32bit(bitPos >> 3) << (bitPos & 7) (golomb.go:179-180): 32 bits
                             * from the BYTE the position lies in, so its low bitPos & 7 bits are zeros, not stream bits. A
                             * prefix and k bits reach in there once k > 16, which takes KB > 16 and a mean above 2^25: an
                             * effective pb above 127 (cookie PB > 73), where pb * mean wraps and the mean no longer
                             * contracts. (Found by the randomized sweep with PB 255, KB 32, round 4.) */
                            const uint32_t w = rd.window(pos) & (0xffffffffu << (pos & 7u));
                            uint32_t n = clz32(~w);
                            if (n >= 9) {
                                /* escape code: getStreamBits(bitPos+9, maxSize), golomb.go:184-186,86-108 */
                                const uint32_t gpos = pos + 9u;
                                const uint32_t gb = gpos & 7u;
                                const bool five = chan_bits + gb > 32u;
                                if ((gpos >> 3) > size || (five && (gpos >> 3) >= size)) {
                                    err = ST_MALFORMED; /* read32bit / input[byteOffset+4] out of range */
                                    err_chan = c;
                                } else {
                                    const uint64_t w2 = bits.window(gpos);
                                    if (chan_bits == 0) n = 0;
                                    else if (chan_bits <= 32) n = (uint32_t)(w2 >> (64u - chan_bits));
                                    else n = (uint32_t)(w2 >> 31) & ((2u << gb) - 1u); /* numBits 33: only byte 5 survives */
                                    pos += 9u + chan_bits;
                                    rd.seek(pos);
                                }
                            } else {
                                /* k == 1 needs no special case: v < 2 always, n = pre * 1, k - 1 = 0 extra bits
                                 * (golomb.go:188-201) */
                                const uint32_t v = k == 0 ? 0u : (w << (n + 1u)) >> (32u - k);
                                /* consumed: prefix + 1, then k bits (v >= 2) or k - 1 (v < 2) */
                                pos += n + 1u + k - (v >= 2 ? 0u : 1u);
                                n = v >= 2 ? n * m + v - 1u : n * m;
                                rd.slide(pos);
                            }
                            if (err == 0) {
                                const uint32_t nd = n + zmode;
                                const int32_t half = (int32_t)((nd + 1u) >> 1); /* golomb.go:206-209 */
                                del = (nd & 1u) ? -half : half;
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
                                        err_chan = c;
                                    } else {
                                        const uint32_t wz = rd.window(pos) & (0xffffffffu << (pos & 7u)); /* as above (golomb.go:115-116) */
                                        const uint32_t pre = clz32(~wz);
                                        uint32_t rl;
                                        if (pre >= 9) {
                                            rl = (wz << 9) >> 16;
                                            pos += 25u;
                                        } else {
                                            /* kz > 32 (the mean has passed 2^30 and mean << 2 wrapped: pb > 127 only): Go's
                                             * shift by 32 - kz, a huge uint32, gives 0 (golomb.go:133) */
                                            const uint32_t val = kz == 0 ? 0u : go_shr(wz << (pre + 1u), 32u - kz);
                                            pos += pre + 1u + kz;
                                            if (val < 2) {
                                                rl = pre * mz;
                                                pos -= 1u;
                                            } else {
                                                rl = pre * mz + val - 1u;
                                            }
                                        }
                                        if ((uint64_t)i + 1u + rl > ns) {
                                            err = ST_SAMPLE_OVERRUN; /* golomb.go:232-234 */
                                            err_chan = c;
                                        }
                                        zrem = rl;
                                        if (rl >= 65535u) zmode = 0;
                                        mean = 0;
                                        rd.slide(pos);
                                    }
                                }
                            }
                        }
                        /* ---- delta pre-pass when mode != 0 (decoder.go:307-309: numActive 31, denShift 0) */
                        if (mode != 0) {
                            dprev = i == 0 ? del : sext_cs(del + dprev, chan_shift);
                            del = dprev;
                        }
                        /* ---- one predictor step: UnpcBlock, predictor.go:45-94 ------------------------------ */
                        const int32_t prev = hist[0];
                        if (i == 0 || na == 0) {
                            o = del; /* out[0] = pc1[0]; numActive 0 copies */
                        } else if (na == 31 || i <= na) {
                            o = sext_cs(del + prev, chan_shift); /* delta mode / warm-up, predictor.go:63-79 */
                        } else if (fast) {
                            /* unpcBlock4/5/6/8 and the general form (predictor.go:99-684) on NA register taps;
                             * taps j >= na carry coef 0 and are masked out of the adaptation */
                            int32_t top = hist[1];
#pragma unroll
                            for (int j = 2; j <= NA; ++j) top = na == (uint32_t)j ? hist[j] : top;
                            int32_t d[NA];
                            int32_t acc = den_half;
                            if (narrow) {
#pragma unroll
                                for (int j = 0; j < NA; ++j) {
                                    d[j] = top - hist[j];
                                    acc -= ALAC_MUL24(coef[j], d[j]);
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < NA; ++j) {
                                    d[j] = top - hist[j];
                                    acc -= coef[j] * d[j];
                                }
                            }
                            o = sext_cs(del + top + (acc >> den_shift), chan_shift);
                            if (del != 0) {
                                if (narrow) {
                                    /* sign-normalised adaptation (exact while nothing can wrap: |del| < 2^23):
                                     * D0 = sign*del0 shrinks by (na-j)*q_j, q_j = (sign*|d_j|) >> denShift taken
                                     * on the positive side, until D0 <= 0 (predictor.go:134-186) */
                                    const bool neg = del < 0;
                                    const int32_t rnd = neg ? (int32_t)((1u << den_shift) - 1u) : 0;
                                    int32_t big_d0 = neg ? -del : del;
                                    bool go = true;
#pragma unroll
                                    for (int j = NA - 1; j >= 0; --j) {
                                        const bool act = go && (uint32_t)j < na;
                                        const int32_t sd = sign_of(d[j]);
                                        int32_t cj = coef[j] - (neg ? -sd : sd);
                                        if (WRAP && wrap16) cj = (int32_t)(int16_t)cj; /* predictor.go:664,675 */
                                        const int32_t ad = d[j] < 0 ? -d[j] : d[j];
                                        const int32_t q = (ad + rnd) >> den_shift;
                                        coef[j] = act ? cj : coef[j];
                                        big_d0 -= act ? ALAC_MUL24((int32_t)(na - (uint32_t)j), q) : 0;
                                        if ((uint32_t)j < na) go = act && big_d0 > 0; /* masked taps do not break the chain */
                                    }
                                } else {
                                    const int32_t sg = del > 0 ? 1 : -1;
                                    int32_t del0 = del;
                                    bool go = true;
#pragma unroll
                                    for (int j = NA - 1; j >= 0; --j) {
                                        const bool act = go && (uint32_t)j < na;
                                        const int32_t sgn = sg > 0 ? sign_of(d[j]) : -sign_of(d[j]);
                                        int32_t cj = coef[j] - sgn;
                                        if (WRAP && wrap16) cj = (int32_t)(int16_t)cj;
                                        coef[j] = act ? cj : coef[j];
                                        del0 -= act ? (int32_t)(na - (uint32_t)j) * ((sgn * d[j]) >> den_shift) : 0;
                                        if ((uint32_t)j < na) go = act && (sg > 0 ? del0 > 0 : del0 < 0);
                                    }
                                }
                            }
                        } else {
                            /* orders outside this variant's class: state lives in the fall-back tile
                             * (g_slot 0..31 history ring, 32..63 coefficients); unpcBlockGeneral or the
                             * int32-coefficient form per order (predictor.go:81-93). Rare by construction. */
                            const bool w16 = order_wraps16(na);
                            const int32_t top = *wv.g_slot((i - 1u - na) & 31u);
                            int32_t sum1 = 0;
                            for (uint32_t j = 0; j < na; ++j)
                                sum1 += *wv.g_slot(32u + j) * (*wv.g_slot((i - 1u - j) & 31u) - top);
                            o = sext_cs(del + top + ((sum1 + den_half) >> den_shift), chan_shift);
                            const int32_t sg = sign_of(del);
                            if (sg != 0) {
                                int32_t del0 = del;
                                for (int32_t j = (int32_t)na - 1; j >= 0; --j) {
                                    const int32_t dd = top - *wv.g_slot((i - 1u - (uint32_t)j) & 31u);
                                    const int32_t sgn = sg > 0 ? sign_of(dd) : -sign_of(dd);
                                    int32_t cj = *wv.g_slot(32u + (uint32_t)j) - sgn;
                                    if (w16) cj = (int32_t)(int16_t)cj;
                                    *wv.g_slot(32u + (uint32_t)j) = cj;
                                    del0 -= (int32_t)(na - (uint32_t)j) * ((sgn * dd) >> den_shift);
                                    if (sg > 0 ? del0 <= 0 : del0 >= 0) break;
                                }
                            }
                        }
#pragma unroll
                        for (int j = NA; j >= 1; --j) hist[j] = hist[j - 1];
                        hist[0] = o;
                        if (!fast) *wv.g_slot(i & 31u) = o;
                    } else {
                        /* decodeSCEEscape / decodeCPEEscape, decoder.go:326-345 / 507-535 */
                        o = sext_cs((int32_t)bits.get(data_pos + (i * nch_e + c) * chan_bits, chan_bits), chan_shift);
                    }

                    /* ---- hand-off / unmix / PCM ---------------------------------------------------------------- */
                    if (err == 0) {
                        if (!last_chan) {
                            *wv.u_row(i) = o; /* U waits for V */
                        } else {
                            int32_t l = o, r = 0;
                            if (cpe) {
                                const int32_t u = u_pre, v = o;
                                if (mix_res != 0) { /* matrix.go:40-41 */
                                    l = u + v - ((mix_res * v) >> mix_sh);
                                    r = l - v;
                                } else {
                                    l = u;
                                    r = v;
                                }
                            }
                            if (depth == 20) { /* matrix.go:77-78, 237 */
                                l = (int32_t)((uint32_t)l << 4);
                                r = (int32_t)((uint32_t)r << 4);
                            }
                            if (use_shift) { /* matrix.go:129-132, 266-268 */
                                const uint32_t sp = shift_pos + i * nch_e * shift_bits;
                                l = (int32_t)((uint32_t)l << shift_bits) | (int32_t)bits.get(sp, shift_bits);
                                if (cpe) r = (int32_t)((uint32_t)r << shift_bits) | (int32_t)bits.get(sp + shift_bits, shift_bits);
                            }
                            if (staged) {
                                if (bps == 2 && cpe) {
                                    wv.st_push(((uint32_t)l & 0xffffu) | ((uint32_t)r << 16));
                                } else {
                                    /* little-endian byte packer: whole dwords go to the stager */
                                    const uint64_t msk = bps == 4 ? 0xffffffffull : ((1ull << (8u * bps)) - 1ull);
                                    pk_acc |= ((uint64_t)(uint32_t)l & msk) << (8u * pk_n);
                                    pk_n += bps;
                                    if (pk_n >= 4) {
                                        wv.st_push((uint32_t)pk_acc);
                                        pk_acc >>= 32;
                                        pk_n -= 4;
                                    }
                                    if (cpe) {
                                        pk_acc |= ((uint64_t)(uint32_t)r & msk) << (8u * pk_n);
                                        pk_n += bps;
                                        if (pk_n >= 4) {
                                            wv.st_push((uint32_t)pk_acc);
                                            pk_acc >>= 32;
                                            pk_n -= 4;
                                        }
                                    }
                                }
                            } else {
                                uint8_t* dst = out + (uint64_t)i * frame_stride + out_chan * bps;
                                store_le(dst, l, bps);
                                if (cpe) store_le(dst + bps, r, bps);
                            }
                        }
                    }
                }
                wv.st_step(); /* collective */
            }
            /* UnpcBlock warm-up indexes 1..numActive of the frame-length buffers (predictor.go:76-79) */
            if (run && err == 0 && !escape && na != 0 && na != 31 && na >= cfg.frame_length) {
                err = ST_MALFORMED;
                err_chan = c;
            }
        }
        if (staged) {
            /* tail of the stream: dwords still in the row, then the bytes still in the packer */
            const uint32_t done = wv.st_finish();
            for (uint32_t b = 0; b < pk_n; ++b) out[(uint64_t)done * 4u + b] = (uint8_t)(pk_acc >> (8u * b));
        }
        if (has) {
            if (err) {
                const uint32_t stage = (escape || err == ST_MALFORMED) ? (uint32_t)ALACGPU_STAGE_NONE
                                       : cpe ? (uint32_t)(err_chan == 0 ? ALACGPU_STAGE_ENTROPY_U : ALACGPU_STAGE_ENTROPY_V)
                                             : (uint32_t)ALACGPU_STAGE_ENTROPY;
                st = ALACGPU_STATUS(err, ctx, stage);
                walking = false;
            } else {
#pragma unroll
                for (uint32_t s = 0; s < 8; ++s)
                    if (s >= out_chan && s < out_chan + nch_e) written[s] = umax(written[s], ns);
                num_samples = ns;
                chan_idx += nch_e;
                if (chan_idx >= num_chan) walking = false; /* decoder.go:200-202 */
            }
        }
    }

    if (!live) return 0;
    if (st) {
        /* a Go panic carries no wrapping context: report the bare code */
        if (ALACGPU_STATUS_CODE(st) == ST_MALFORMED) st = ST_MALFORMED;
        *frames_out = 0;
        if (SCAN && pd) {
            pd->status = st;
            pd->frames = 0;
            pd->nslots = 0;
            pd->route = ROUTE_NONE;
        }
        return st;
    }
    if (SCAN) {
        if (pd) {
            pd->status = 0;
            pd->frames = num_samples;
            pd->nslots = chan_idx;
            pd->route = legacy ? ROUTE_LEGACY : ROUTE_SPLIT;
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) pd->written[k] = written[k];
        }
        *frames_out = num_samples;
        return 0;
    }
    /* DecodePacket hands back output[:n] of a zeroed frame buffer (decoder.go:120,127): slots no element
     * wrote, or wrote for fewer frames than the last element, read as zero */
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
