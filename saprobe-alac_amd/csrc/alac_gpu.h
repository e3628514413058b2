/*
 * alac_gpu.h — what the kernel translation units of libalacgpu.so share: the gfx950 forms of the building-block
 * macros of alac_regular.h, the wave policies (GpuWave: LDS stager, bitstream rings, residual queue, DPP reductions;
 * GpuWaveMem: residuals from memory), the launch plan, and the kernels' declarations. The library is built from
 * several translation units (k_sort, k_scan, k_dec16q, k_dec16g, k_dec24q, k_dec32q, k_decw24, k_decw32, k_split, alacgpu)
 * so that the kernels compile in parallel; nothing crosses between them on the device side.
 */
#ifndef ALAC_GPU_H
#define ALAC_GPU_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#define ALAC_DEV __device__ __forceinline__
#define ALAC_HD __host__ __device__ __forceinline__
#define ALAC_NOINLINE
#define ALAC_MUL24(a, b) __mul24((int)(a), (int)(b))
/* |a - b| + c in one instruction. As an expression (max - min + c) the compiler shares the max / min between the
 * unrolled steps of a chunk and ends up with three or four instructions for most taps. */
__device__ __forceinline__ uint32_t alac_sad(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define ALAC_SAD(a, b, c) alac_sad((uint32_t)(a), (uint32_t)(b), (uint32_t)(c))

__device__ __forceinline__ int32_t alac_sign_med3(int32_t x) {
    int32_t r;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(x));
    return r;
}
#define ALAC_SIGN(x) alac_sign_med3(x)
__device__ __forceinline__ int32_t alac_clamp01_med3(int32_t x) {
    int32_t r;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(r) : "v"(x));
    return r;
}
#define ALAC_CLAMP01(x) alac_clamp01_med3(x)
__device__ __forceinline__ uint32_t alac_xad(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define ALAC_XAD(a, b, c) alac_xad((uint32_t)(a), (uint32_t)(b), (uint32_t)(c))
__device__ __forceinline__ uint32_t alac_not_add(uint32_t a, uint32_t c) {
    uint32_t r;
    asm("v_xad_u32 %0, %1, -1, %2" : "=v"(r) : "v"(a), "v"(c));
    return r;
}
#define ALAC_NOT_ADD(a, c) alac_not_add((uint32_t)(a), (uint32_t)(c))
__device__ __forceinline__ uint32_t alac_bfi(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
#define ALAC_BFI(m, a, b) alac_bfi((uint32_t)(m), (uint32_t)(a), (uint32_t)(b))
__device__ __forceinline__ int32_t alac_mad24(int32_t a, int32_t b, int32_t c) {
    int32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define ALAC_MAD24(a, b, c) alac_mad24((int32_t)(a), (int32_t)(b), (int32_t)(c))
__device__ __forceinline__ int32_t alac_msub24(int32_t acc, int32_t a, int32_t negc) {
    int32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(negc), "v"(acc));
    return r;
}
#define ALAC_MSUB24(acc, a, c) alac_msub24((int32_t)(acc), (int32_t)(a), -(int32_t)(c))
#define ALAC_SUBSAT(a, b) __builtin_elementwise_sub_sat((uint32_t)(a), (uint32_t)(b))
/* v_ffbh_u32 as it is: leading zeros, 2^32 - 1 for 0 */
__device__ __forceinline__ uint32_t alac_ffbh(uint32_t x) {
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
#define ALAC_FFBH(x) alac_ffbh((uint32_t)(x))
/* (x >> off[4:0]) & ((1 << width[4:0]) - 1) */
__device__ __forceinline__ uint32_t alac_bfe(uint32_t x, uint32_t off, uint32_t width) {
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(off), "v"(width));
    return r;
}
#define ALAC_BFE(x, off, width) alac_bfe((uint32_t)(x), (uint32_t)(off), (uint32_t)(width))
__device__ __forceinline__ int32_t alac_sext_bits(int32_t x, uint32_t bits) {
    int32_t r;
    asm("v_bfe_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(bits));
    return r;
}
#define ALAC_SEXT_BITS(x, bits) alac_sext_bits((int32_t)(x), (uint32_t)(bits))
#define ALAC_MULU24(a, b) __umul24((unsigned)(a), (unsigned)(b))
#define ALAC_MULHI(a, b) __umulhi((unsigned)(a), (unsigned)(b))
#define ALAC_ALIGNBIT(hi, lo, sh) __builtin_amdgcn_alignbit((unsigned)(hi), (unsigned)(lo), (unsigned)(sh))
__device__ __forceinline__ int32_t alac_med3_0(int32_t x, int32_t m) {
    int32_t r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(m));
    return r;
}
#define ALAC_MED3_0(x, m) alac_med3_0((int32_t)(x), (int32_t)(m))
/* The two blocks of the Golomb step (alac_regular.h: gol_step, whose C++ form these follow instruction for
 * instruction), each ONE asm statement: the order inside is the order of issue, and no instruction reads the result of
 * the one before it (a lone wave: 8.3 cycles instead of 4.8). v_cmp .. v_addc: two instructions apart (the compiler keeps
 * the same distance). 128 is no inline constant and VOP3 takes no literal on gfx9: it comes in an SGPR. */
#ifndef ALAC_GOL_ASM
#define ALAC_GOL_ASM 0
#endif
#define ALAC_GOL_BLOCK_A(wa, wb, sh, ck, t9, norun, pos, mean, zq, pb, n, esc, mt, pos2, aoff, zq2, mean2, nhi)            \
    do {                                                                                                                   \
        uint32_t w_, k_, t_, pre_, v_, pk_, vm1_, cons_;                                                                   \
        asm("v_alignbit_b32 %[w], %[Wa], %[Wb], %[Sh]\n\t"                                                                 \
            "v_sub_u32 %[k], 31, %[Ck]\n\t"                                                                                \
            "v_not_b32 %[t], %[w]\n\t"                                                                                     \
            "v_sub_u32 %[Mt], %[Mean], %[T9]\n\t"                                                                          \
            "v_ffbh_u32 %[pre], %[t]\n\t"                                                                                  \
            "v_xad_u32 %[Zq2], %[Norun], -1, %[Zq]\n\t"                                                                    \
            "v_sub_u32 %[t], %[Ck], %[pre]\n\t"                                                                            \
            "v_sub_u32_e64 %[Esc], %[pre], 8 clamp\n\t"                                                                    \
            "v_bfe_u32 %[v], %[w], %[t], %[k]\n\t"                                                                         \
            "v_lshlrev_b32 %[pk], %[k], %[pre]\n\t"                                                                        \
            "v_cmp_lt_u32 vcc, 1, %[v]\n\t"                                                                                \
            "v_sub_u32_e64 %[vm1], %[v], 1 clamp\n\t"                                                                      \
            "v_sub_u32 %[pk], %[pk], %[pre]\n\t"                                                                           \
            "v_addc_co_u32 %[cons], vcc, %[pre], %[k], vcc\n\t"                                                            \
            "v_add_u32 %[N], %[pk], %[vm1]\n\t"                                                                            \
            "v_and_b32 %[cons], %[cons], %[Norun]\n\t"                                                                     \
            "v_and_b32 %[N], %[N], %[Norun]\n\t"                                                                           \
            "v_add_u32 %[Pos2], %[Pos], %[cons]\n\t"                                                                       \
            "v_mad_u32_u24 %[Mean2], %[Pb], %[N], %[Mt]\n\t"                                                               \
            "v_lshrrev_b32 %[Aoff], 3, %[Pos2]\n\t"                                                                        \
            "v_lshrrev_b32 %[Nhi], 16, %[N]\n\t"                                                                           \
            "v_and_b32 %[Aoff], 0x7c, %[Aoff]"                                                                             \
            : [w] "=&v"(w_), [k] "=&v"(k_), [t] "=&v"(t_), [pre] "=&v"(pre_), [v] "=&v"(v_), [pk] "=&v"(pk_),              \
              [vm1] "=&v"(vm1_), [cons] "=&v"(cons_), [N] "=&v"(n), [Esc] "=&v"(esc), [Mt] "=&v"(mt), [Pos2] "=&v"(pos2),  \
              [Aoff] "=&v"(aoff), [Zq2] "=&v"(zq2), [Mean2] "=&v"(mean2), [Nhi] "=&v"(nhi)                                 \
            : [Wa] "v"(wa), [Wb] "v"(wb), [Sh] "v"(sh), [Ck] "v"(ck), [T9] "v"(t9), [Norun] "v"(norun), [Pos] "v"(pos),    \
              [Mean] "v"(mean), [Zq] "v"(zq), [Pb] "v"(pb)                                                                 \
            : "vcc");                                                                                                      \
    } while (0)
#define ALAC_GOL_BLOCK_B(mean2, nhi, esc, pos2, zq2, norun, near, pbs, c31kb, rare, sh2, ck2, t92, norun2)                 \
    do {                                                                                                                   \
        uint32_t zs_, x_;                                                                                                  \
        asm("v_sub_u32_e64 %[zs], %[C128], %[Mean2] clamp\n\t"                                                             \
            "v_lshrrev_b32 %[x], 9, %[Mean2]\n\t"                                                                          \
            "v_not_b32 %[Sh2], %[Pos2]\n\t"                                                                                \
            "v_or3_b32 %[zs], %[Esc], %[Nhi], %[zs]\n\t"                                                                   \
            "v_mul_hi_u32 %[T92], %[Mean2], %[Pbs]\n\t"                                                                    \
            "v_add_u32 %[x], 3, %[x]\n\t"                                                                                  \
            "v_ashrrev_i32 %[Norun2], 31, %[Zq2]\n\t"                                                                      \
            "v_ffbh_u32 %[x], %[x]\n\t"                                                                                    \
            "v_bitop3_b32 %[Rare], %[zs], %[Norun], %[Near] bitop3:0xc8\n\t"                                               \
            "v_max_i32 %[Ck2], %[x], %[C31kb]"                                                                             \
            : [zs] "=&v"(zs_), [x] "=&v"(x_), [Sh2] "=&v"(sh2), [T92] "=&v"(t92), [Norun2] "=&v"(norun2),                  \
              [Rare] "=&v"(rare), [Ck2] "=&v"(ck2)                                                                         \
            : [C128] "s"(128u), [Mean2] "v"(mean2), [Nhi] "v"(nhi), [Esc] "v"(esc), [Pos2] "v"(pos2), [Zq2] "v"(zq2),      \
              [Norun] "v"(norun), [Near] "v"(near), [Pbs] "v"(pbs), [C31kb] "v"(c31kb));                                   \
    } while (0)
#define ALAC_PICK(dst, src) asm volatile("v_mov_b32 %0, %1" : "+v"(dst) : "v"(src))
#define ALAC_OWN_REG(x) asm volatile("" : "+v"(x))
typedef uint32_t alac_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#define ALAC_LOAD4(q, a, b, c, d)                                                       \
    do {                                                                                \
        const alac_u32x4_a4 v_ = *reinterpret_cast<const alac_u32x4_a4*>(q);            \
        (a) = v_.x;                                                                     \
        (b) = v_.y;                                                                     \
        (c) = v_.z;                                                                     \
        (d) = v_.w;                                                                     \
    } while (0)
typedef int32_t alac_i32x4 __attribute__((ext_vector_type(4)));
#define ALAC_LOAD4_I32(q, a, b, c, d)                                                  \
    do {                                                                               \
        const alac_i32x4 v_ = *reinterpret_cast<const alac_i32x4*>(q); /* 16-byte aligned */ \
        (a) = v_.x;                                                                    \
        (b) = v_.y;                                                                    \
        (c) = v_.z;                                                                    \
        (d) = v_.w;                                                                    \
    } while (0)
#define ALAC_STORE4(q, a, b, c, d) (*reinterpret_cast<alac_i32x4*>(q) = alac_i32x4{(a), (b), (c), (d)})
#ifdef ALAC_DUO_PROF
/* profiling build: cycles (s_memtime) between the stamps of alac_duo.h, summed per role over all waves into
 * Plan::prof (the plan is zeroed before every decode; alacgpu_debug_prof reads the last one back) */
#define ALAC_DUO_STAMP(k)                                                   \
    do {                                                                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
        if ((k) > 0) wv.prof[kProfPhase + (k) - 1] += t_ - wv.prof_t;         \
        wv.prof_t = t_;                                                     \
    } while (0)
#endif
#include "alac_wave.h"
#include "alac_regular.h"
#include "alac_duo.h"
#include "alac_split.h"

/* s_setprio levels of the wave pair (see k_decode_body.inc: pair_item) */
#ifndef ALAC_PRIO_B_LONG
#define ALAC_PRIO_B_LONG 3  /* predictor waves, order > 8 */
#define ALAC_PRIO_B_MID 2   /* order 6..8 */
#define ALAC_PRIO_B_SHORT 1 /* order < 6 */
#define ALAC_PRIO_A 2       /* entropy waves */
#endif

namespace alack {


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kWave = 64;
constexpr uint32_t kTimingSlots = 64;
/* LDS footprint of a wave pair, set per translation unit. ALAC_LDS_ROWS: dwords of PCM a lane row holds, flushed half
 * a row at a time (64: whole 128-byte lines per packet; 32: 64-byte pieces, the two halves of a line follow each other
 * through L2). ALAC_LDS_RING: dwords of bitstream ring per lane (32 or 16). 64/32 = 34.5 KB per workgroup = four pairs
 * per CU; 32/32 = 26.1 KB = six; 32/16 = 21.9 KB = seven. */
#ifndef ALAC_LDS_ROWS
#define ALAC_LDS_ROWS 64
#endif
#ifndef ALAC_LDS_RING
#define ALAC_LDS_RING 32
#endif
#ifndef ALAC_LDS_FLUSH
#define ALAC_LDS_FLUSH (ALAC_LDS_ROWS / 2)
#endif
constexpr uint32_t kRing = ALAC_LDS_ROWS;                   /* dwords of PCM a lane row holds (two flush chunks) */
/* dwords per lane and flush. Half a row where a writer pushes groups that straddle the flush boundary (six dwords per four
 * 3-byte frames); a WHOLE row (ALAC_LDS_FLUSH = ALAC_LDS_ROWS = 32) where every push is a power of two of dwords that ends
 * on the boundary and st_step() follows every group: the row is written out the moment it is full, as whole 128-byte
 * lines, from half the LDS (k_dec16q.hip, k_dec32q.hip) */
constexpr uint32_t kFlush = ALAC_LDS_FLUSH;
static_assert(kFlush <= kRing && kRing % 4 == 0 && (kFlush == 16 || kFlush == 32), "the stager flushes 64- or 128-byte pieces of its rows");
constexpr uint32_t kRowStride = kRing + 1;                  /* odd stride = conflict-free column access */
constexpr uint32_t kFallbackSlots = 64;
constexpr uint32_t kRingDw = ALAC_LDS_RING;                 /* bitstream ring, dwords per lane */
/* Bitstream rings, SLOT-major (round 4): slot s of all 64 lanes lies side by side (s_ring[s * 64 + lane]), so that a
 * wave's access to ANY mix of slots — the lanes' positions drift apart — is free of bank conflicts (the bank is the lane's).
 * Rounds 1-3 had a row of 36 dwords per lane: lanes l and l + 8 shared a bank, and since round 4 the step's window read
 * sits in the entropy chain (alac_regular.h: RingRd). Two spare slots behind the ring: slot kRingDw repeats slot 0. */
constexpr uint32_t kRingSlots = kRingDw + 2;

/* device-side launch plan, rebuilt by every decode */
/* sort keys: 0..2047 regular packets (numU*32 + numV + KEY_WIDE, alac_regular.h); 2048 / 2049 irregular packets
 * (below). A workgroup holds packets of ONE key. */
constexpr uint32_t kKeys = alac::KEY_IRREGULAR + 2u;
constexpr uint32_t kKeyLegacy = alac::KEY_IRREGULAR;     /* decode_wave */
constexpr uint32_t kKeyScan = alac::KEY_IRREGULAR + 1u;  /* decode_wave<SCAN> + split pipeline */
struct Plan {
    uint32_t count[kKeys];     /* packets per key */
    uint32_t pkt_start[kKeys]; /* first index in perm[] */
    uint32_t cursor[kKeys];    /* scatter cursors */
    /* compact list of the non-empty keys in dispatch order (slowest first) */
    uint32_t nk;
    uint32_t list_key[kKeys];
    uint32_t list_wave0[kKeys]; /* first block id */
    uint32_t total_waves;
    uint32_t irr_waves;  /* waves of the irregular keys (>= KEY_IRREGULAR): they come first */
    uint32_t wide_waves; /* then those of the wide keys (KEY_WIDE..), then the narrow ones */
    /* the pair kernels (k_decode_body.inc; [0] narrow keys, [1] wide): workgroups admitted per CU so far and the
     * next item of the queue. Zeroed with the rest of the plan before every decode. */
    uint32_t gate[2][512];
    uint32_t balance[2][512]; /* per SIMD of the CU, a byte each: entropy waves - predictor waves placed there */
    uint32_t queue[2];
#ifdef ALAC_DUO_PROF
    unsigned long long prof[32]; /* [role A: U phase 0..3, last phase 4..7 | role C: 8..15 | role B: 16..19, 20..23] */
#endif
};

/* LDS of the decode kernel (one wave per workgroup). Referenced by name, never through a generic pointer, so
 * every access is a ds_* instruction (a pointer kept in a struct decays to flat_* loads and stores). */
static __shared__ uint32_t s_rows[kWave * kRowStride];                                  /* PCM stager rows */
static __shared__ unsigned long long s_optr[kWave];                                     /* PCM slot of each lane's packet */
static __shared__ __attribute__((aligned(16))) uint32_t s_ring[kRingSlots * kWave];     /* bitstream rings */
/* residual queue of the wave pair (alac_duo.h), A -> B, double-buffered chunks */
constexpr uint32_t kQ = alac::DUO_CHUNK;
/* rows per buffer: a chunk of residuals A -> B; where another wave writes the PCM (alac_duo.h: EMIT_A) half a chunk of
 * residuals and half a chunk of samples B -> writer; with a writer wave and ALAC_FWD (ALAC_LDS_QROWS 32) also half a
 * chunk of U samples and six rows of shift bytes (alac_duo.h: FWD) */
#ifndef ALAC_LDS_QROWS
#define ALAC_LDS_QROWS ALAC_DUO_CHUNK
#endif
constexpr uint32_t kQRows = ALAC_LDS_QROWS;
static __shared__ int32_t s_rq[2 * kQRows * kWave];

/* U hand-off tile of one wave: frame_length rows of 64 cells and one spare row (the single-wave decoders read one
 * row ahead) */
__host__ __device__ inline size_t u_tile_cells(uint32_t frame_length) { return ((size_t)frame_length + 1u) * kWave; }

/* ---- gfx950 wave policy for alac::decode_wave --------------------------------------------------------- */
struct GpuWave {
    static constexpr bool kResMem = false; /* residuals come through the LDS queue */
    static constexpr uint32_t kRingDw = alack::kRingDw;
    int32_t* u_tile;           /* HBM: the wave's U hand-off tile (wave-uniform: rows are addressed scalar base + lane) */
    int32_t* g_tile;           /* HBM: this lane's column of the wave's fall-back tile */
    uint8_t* my_out;
    uint32_t lane, wcnt, flushed;
    uint32_t ppw;              /* packets (= live lanes) per wave; also the row stride of the HBM tiles */
#ifdef ALAC_DUO_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = 0;
#endif

    ALAC_DEV bool any(bool p) const { return __ballot(p) != 0ull; }
    ALAC_DEV uint32_t max_u32(uint32_t v) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
            v = t > v ? t : v;
        }
        /* every lane holds the maximum: hand it back as a scalar, so loops bounded by it are uniform */
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    }
    /* position in the lane's stager row of the next dword pushed / the next one flushed: wcnt, flushed modulo kRing. Rows of
     * 48 dwords (k_dec24q.hip: eight six-dword groups of the 3-byte pair writer, so that no group straddles the end of the
     * row; whole 128-byte lines from 12 KB of rows) keep the two positions in registers of their own */
    static constexpr bool kRingPow2 = (kRing & (kRing - 1u)) == 0u;
    uint32_t wpos, fpos;
    ALAC_DEV static uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
    ALAC_DEV static uint32_t st_wrap(uint32_t x) { return kRingPow2 ? (x & (kRing - 1u)) : umin_(x, x - kRing); } /* x < 2 kRing */
    ALAC_DEV uint32_t st_pos() const { return kRingPow2 ? (wcnt & (kRing - 1u)) : wpos; }
    ALAC_DEV void st_adv(uint32_t n) {
        wcnt += n;
        if (!kRingPow2) wpos = st_wrap(wpos + n);
    }
    ALAC_DEV void st_begin(uint8_t* out) {
        my_out = out;
        s_optr[lane] = (unsigned long long)reinterpret_cast<uintptr_t>(out);
        wcnt = flushed = 0;
        wpos = fpos = 0;
    }
    ALAC_DEV void st_push(uint32_t v) {
        s_rows[lane * kRowStride + st_pos()] = v;
        st_adv(1u);
    }
    /* branch-free form: a lane that is not `on` rewrites its next free slot and does not advance */
    ALAC_DEV void st_push_if(uint32_t v, bool on) {
        s_rows[lane * kRowStride + st_pos()] = v;
        st_adv(on ? 1u : 0u);
    }
    /* Groups of un = 4 or 8 dwords, one per step, from lanes that either keep all of them or none (alac_duo.h: whole chunks
     * inside or behind the lane's frames): the group's place in the row is worked out once, the eight stores carry
     * immediate offsets. A lane that keeps them (`inc` = un) stands at a multiple of un (it has pushed one dword per
     * frame); one that does not has already written its tail out (st_finish) or never had anything: it scribbles over
     * eight dwords of its own row that nobody will look at. (16-bit pairs written by the predictor wave: rows of 32 / 64.) */
    ALAC_DEV uint32_t st_group_base(uint32_t un) const { return lane * kRowStride + (kRingPow2 ? (wcnt & (kRing - un)) : wpos); }
    ALAC_DEV void st_put(uint32_t base, uint32_t j, uint32_t v, uint32_t) { s_rows[base + j] = v; }
    ALAC_DEV void st_advance(uint32_t n) { st_adv(n); }
    /* six dwords at once (four 24-bit stereo frames): straight on from the lane's position; a group that crosses the end of
     * the row (rows of 32 / 64: one in five / ten, the same one for every lane that is still going; rows of 48: none while the
     * lane pushes whole groups) wraps dword by dword */
    /* (all six are stored; the lane's position moves on by `count`) */
    ALAC_DEV void st_push6_n(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t d4, uint32_t d5, uint32_t count) {
        const uint32_t pos = st_pos();
        uint32_t* row = s_rows + lane * kRowStride;
        if (pos <= kRing - 6u) {
            uint32_t* r = row + pos;
            r[0] = d0;
            r[1] = d1;
            r[2] = d2;
            r[3] = d3;
            r[4] = d4;
            r[5] = d5;
        } else {
            row[pos] = d0;
            row[st_wrap(pos + 1u)] = d1;
            row[st_wrap(pos + 2u)] = d2;
            row[st_wrap(pos + 3u)] = d3;
            row[st_wrap(pos + 4u)] = d4;
            row[st_wrap(pos + 5u)] = d5;
        }
        st_adv(count); /* the lane keeps the first `count` of them (a partial frame ends inside the group) */
    }
    /* bytes of the last, incomplete dword of the stream (after every dword pushed so far) */
    ALAC_DEV void st_tail(uint64_t acc, uint32_t nbytes) {
        for (uint32_t b = 0; b < nbytes; ++b) my_out[(size_t)wcnt * 4u + b] = (uint8_t)(acc >> (8u * b));
    }
    /* Collective. Rows that just completed a kFlush-dword chunk are written out: kFlush / 4 lanes x 16 B per packet
     * (whole 128-B lines with kFlush = 32, halves with 16), 64 / (kFlush / 4) packets per store instruction. Lock step
     * makes `flushed` identical in all full lanes. */
    ALAC_DEV void st_step() {
        constexpr uint32_t LPP = kFlush / 4u;      /* lanes per packet */
        constexpr uint32_t PPI = kWave / LPP;      /* packets per store instruction */
        const bool full = (wcnt - flushed) >= kFlush;
        const unsigned long long mask = __ballot(full);
        if (mask == 0ull) return;
        __builtin_amdgcn_wave_barrier();
        const int first = __ffsll((long long)mask) - 1;
        const uint32_t fl = (uint32_t)__shfl((int)flushed, first, 64);
        const uint32_t col0 = kRingPow2 ? (fl & (kRing - 1u)) : (uint32_t)__shfl((int)fpos, first, 64);
        const uint32_t piece = lane & (LPP - 1u);
        const uint32_t groups = (ppw + PPI - 1u) / PPI;
        const uint32_t col = st_wrap(col0 + piece * 4u); /* kRing and col0 are multiples of four: a piece never straddles the end */
        for (uint32_t k = 0; k < groups; ++k) {
            const uint32_t q = PPI * k + lane / LPP;
            if ((mask >> q) & 1ull) {
                const uint32_t* r = s_rows + q * kRowStride + col;
                const uint4 v = make_uint4(r[0], r[1], r[2], r[3]);
                uint8_t* dst = reinterpret_cast<uint8_t*>((uintptr_t)s_optr[q]) + ((size_t)fl + piece * 4u) * 4u;
                /* the address came through LDS as an integer: name the global address space, or it is a flat store */
                *reinterpret_cast<__attribute__((address_space(1))) u32x4*>((uintptr_t)dst) = u32x4{v.x, v.y, v.z, v.w};
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (full) {
            flushed += kFlush;
            if (!kRingPow2) fpos = st_wrap(fpos + kFlush);
        }
    }
    ALAC_DEV uint32_t st_finish() {
        uint32_t c = kRingPow2 ? 0u : fpos;
        for (uint32_t w = flushed; w < wcnt; ++w) {
            *reinterpret_cast<uint32_t*>(my_out + (size_t)w * 4u) = s_rows[lane * kRowStride + (kRingPow2 ? (w & (kRing - 1u)) : c)];
            if (!kRingPow2) c = st_wrap(c + 1u);
        }
        flushed = wcnt;
        if (!kRingPow2) fpos = wpos;
        return wcnt;
    }
    /* bitstream ring of the entropy wave: kRingDw dwords per lane, slot-major (kRingSlots) */
    ALAC_DEV void ring_write4(uint32_t slot, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        uint32_t* q = &s_ring[slot * kWave + lane]; /* two ds_write2st64_b32 */
        q[0] = a;
        q[kWave] = b;
        q[2 * kWave] = c;
        q[3 * kWave] = d;
    }
    ALAC_DEV uint32_t ring_read(uint32_t slot) const { return s_ring[slot * kWave + lane]; }
    /* slots `slot` and `slot + 1` in one ds_read2st64_b32 (slot <= kRingDw - 1: slot kRingDw repeats slot 0, RingRd::commit) */
    ALAC_DEV void ring_read2(uint32_t slot, uint32_t& a, uint32_t& b) const {
        const uint32_t* q = &s_ring[slot * kWave + lane];
        a = q[0];
        b = q[kWave];
    }
    /* the same by 4 * slot (what gol_step has at hand: one v_lshl_add_u32 to the address) */
    ALAC_DEV void ring_read2_at(uint32_t byte_off, uint32_t& a, uint32_t& b) const {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(&s_ring[lane]) + byte_off * kWave);
        a = q[0];
        b = q[kWave];
    }
    ALAC_DEV void ring_write1(uint32_t slot, uint32_t v) { s_ring[slot * kWave + lane] = v; }
    /* residual queue: row j of buffer buf holds step j of the chunk for all 64 lanes (conflict-free) */
    ALAC_DEV void rq_write(uint32_t buf, uint32_t j, int32_t v) { s_rq[(buf * kQRows + j) * kWave + lane] = v; }
    ALAC_DEV int32_t rq_read(uint32_t buf, uint32_t j) const { return s_rq[(buf * kQRows + j) * kWave + lane]; }
    /* two lanes per packet (alac_duo.h: duo_phase_lanes): DPP moves inside every pair of neighbouring lanes. Their source
     * must be the result of an instruction the compiler knows (not of inline asm): it inserts the wait states they need. */
    /* (mov_dpp, not update_dpp(0, ...): every lane of a quad_perm has a source, so there is no old value to keep; with one
     * the compiler writes a v_mov 0 in front of every DPP move) */
    ALAC_DEV uint32_t pair_hi(uint32_t x) const { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xF5, 0xf, 0xf, true); } /* [1,1,3,3] */
    ALAC_DEV int32_t pair_sum(int32_t x) const { return x + __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true); }            /* [1,0,3,2] */
    /* lane 1 takes lane 0's x, lane 0 takes `fresh` (q0m: all ones in lane 0) */
    ALAC_DEV uint32_t pair_from_below(uint32_t x, uint32_t fresh, uint32_t q0m) const {
        return ALAC_BFI(q0m, fresh, (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xA0, 0xf, 0xf, true)); /* [0,0,2,2] */
    }
    /* four lanes per packet: the same inside every group of four neighbouring lanes (quad_perm) */
    template <int L>
    ALAC_DEV uint32_t quad_from(uint32_t x) const { /* every lane takes lane L's x */
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, L * 0x55, 0xf, 0xf, true);
    }
    template <int D>
    ALAC_DEV uint32_t quad_up(uint32_t x) const { /* lane q takes lane min(q + D, 3)'s x (the caller masks lanes without one) */
        constexpr int c = D == 1 ? 0xF9 : D == 2 ? 0xFE : 0xFF; /* [1,2,3,3] [2,3,3,3] [3,3,3,3] */
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, c, 0xf, 0xf, true);
    }
    ALAC_DEV int32_t quad_sum(int32_t x) const {
        const int32_t y = x + __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true); /* [1,0,3,2] */
        return y + __builtin_amdgcn_mov_dpp(y, 0x4E, 0xf, 0xf, true);            /* [2,3,0,1] */
    }
    /* lane q takes lane q - 1's x, lane 0 takes `fresh` (q0m: all ones in lane 0) */
    ALAC_DEV uint32_t quad_from_below(uint32_t x, uint32_t fresh, uint32_t q0m) const {
        return ALAC_BFI(q0m, fresh, (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x90, 0xf, 0xf, true)); /* [0,0,1,2] */
    }
    /* chunk hand-over between the two waves of the workgroup: LDS traffic only, so outstanding global loads
     * (ring refills, U prefetch) and stores (U tile) are NOT waited for — __syncthreads() would drain them */
    ALAC_DEV void duo_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    /* end of the U phase: wave B's tile stores must have landed before wave A loads them */
    ALAC_DEV void duo_sync_mem() {
        __threadfence_block();
        __syncthreads();
    }
    /* rows of 64 cells whatever ppw is: a constant stride lets unrolled steps address their rows by immediate
     * offsets from one base. A lane with a packet slot owns cell `lane`; in a narrow wave (ppw < 64: small batches) the
     * lanes beyond ppw share cell ppw of the row, a dummy: they store without a branch like everyone else, but one dword
     * instead of 64 - ppw of them (round 3: the dead lanes of BASELINE config b's sixteen-lane waves wrote three quarters
     * of every row: 6.2 x the algorithmic traffic). */
    ALAC_DEV int32_t* u_row(uint32_t i) const { return u_tile + ((size_t)i * kWave + (ppw < kWave ? (lane < ppw ? lane : ppw) : lane)); }
    ALAC_DEV int32_t* g_slot(uint32_t k) const { return g_tile + (size_t)k * ppw; }
};

/* The predictor wave of the split pipeline when the residuals are already in memory (the scan left them in the
 * task's row, scan_channel): role B of alac_duo.h with the residual "queue" read straight from the row — no entropy
 * wave, no barrier. duo_sync() is called once per chunk iteration and moves the window on. */
struct GpuWaveMem : GpuWave {
    static constexpr bool kResMem = true;
    const int32_t* res; /* this lane's row: residuals in, samples out (the writer runs 32+ samples behind the reads) */
    uint32_t it, chunk0;
    ALAC_DEV int32_t rq_read(uint32_t, uint32_t j) const { return res[chunk0 + j]; }
    ALAC_DEV void rq_write(uint32_t, uint32_t, int32_t) {}
    ALAC_DEV void duo_sync() {
        ++it;
        chunk0 = (it - 1u) * kQ;
    }
    ALAC_DEV void duo_sync_mem() {}
};

/* readable bytes of the blob from a packet's start, as Bits wants them */
__device__ __forceinline__ uint32_t avail_of(uint64_t blob_bytes, uint64_t off) {
    const uint64_t left = blob_bytes - off;
    return left > 0xffffffffull ? 0xffffffffu : (uint32_t)left;
}

/* ---- kernels (one decode = these launches on the handle's stream) ---- */
__global__ void alac_classify(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                              const uint32_t* __restrict__ sizes, uint32_t n, uint16_t* __restrict__ keys, uint32_t* __restrict__ sizes_ws,
                              uint32_t* __restrict__ frames_out, int32_t* __restrict__ status, Plan* plan);
__global__ void alac_plan(Plan* plan, uint32_t ppw);
__global__ void alac_cu_census(uint32_t* __restrict__ seen, uint32_t* __restrict__ arrived, uint32_t expect);
__global__ void alac_scatter(const uint16_t* __restrict__ keys, uint32_t n, Plan* plan, uint32_t* __restrict__ perm);
__global__ void alac_task_classify(alac::DevCfg cfg, const alac::ChanDesc* __restrict__ cd, const alac::PktDesc* __restrict__ pd,
                                   const uint16_t* __restrict__ pkt_keys, uint32_t n_slots, uint16_t* __restrict__ keys, Plan* plan);
__global__ void alac_scan(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                          const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                          uint8_t* __restrict__ out, uint64_t out_stride, uint32_t* __restrict__ frames_out,
                          int32_t* __restrict__ status, int32_t* __restrict__ scratch_u, int32_t* __restrict__ scratch_g,
                          uint32_t ppw, alac::ChanDesc* __restrict__ cd, alac::PktDesc* __restrict__ pd, int32_t* __restrict__ rows,
                          uint64_t row_stride);
__global__ void alac_legacy(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                            const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                            const alac::PktDesc* __restrict__ pd, uint8_t* __restrict__ out, uint64_t out_stride,
                            uint32_t* __restrict__ frames_out, int32_t* __restrict__ status, int32_t* __restrict__ scratch_u,
                            int32_t* __restrict__ scratch_g, uint32_t ppw);
/* ---- which kernel takes the narrow regular wave slots of a batch, and in which shape (k_decode_body.inc; evaluated on the
 * device, where the number of items is known, and again on the host for alacgpu_last_dispatch) ----
 * FOUR-WAVE WORKGROUPS PER CU (round 4). A four-wave workgroup is three waves a few microseconds after it started (the spare
 * wave exits at once on a full device), so by registers a CU holds FIVE of them: four leave 3 waves x 120 registers on every
 * SIMD, a fifth workgroup's four waves find 152 free registers on each, and after its spare wave has gone the SIMDs hold
 * 4 / 4 / 4 / 3 waves and no sixth fits. What decided was LDS: 34 KB per workgroup (four per CU). With <= 30 KB of static
 * LDS (stager rows of 32 or 48 dwords) the SAME kernel is launched with a dynamic-LDS pad up to 34 KB ("fit 4") or without
 * ("fit 5"), both over the whole grid when the host's upper bound of the slot count allows five; decode_mode() — on the
 * device, from the plan's count of narrow regular wave slots — says which launch works, the others' workgroups exit at once.
 * Four per CU stay the rule: up to 4 x CUs slots the dispatcher then spreads them exactly evenly, and a CU
 * with five is 1.38 x slower than one with four (its SIMDs hold 4 / 4 / 4 / 3 waves instead of 3 each, and issue is what
 * the kernel is bound by). Five pay (profiles/r04_final/fit_sweep.txt; q = slots per CU):
 *   4 < q <= 5   a round of five instead of four and a nearly empty one (24-bit 81 920 packets 4.16 -> 3.35 ms, 32-bit
 *                7.11 -> 4.78; 16-bit 3.18 -> 2.95, where the gated pairs of k_dec16g.hip are as fast or faster and keep the batch);
 *   q > 7.5      the dispatcher refills a CU workgroup by workgroup, and with five resident the tail of one "round" overlaps
 *                the start of the next (16-bit 131 072 packets 4.55 -> 4.24 ms, 163 840 5.37 -> 5.21; 24-bit 131 072 6.42 -> 5.25).
 * Between (5 < q <= 7.5) four per CU win (16-bit 114 688 packets 3.67 against 3.89 ms, 24-bit 98 304 4.52 against 4.64,
 * 122 880 4.58 against 4.71). */
constexpr uint32_t kModeFit4 = 4u, kModeFit5 = 5u, kModeGated = 6u;
/* cap: pairs per CU the width's gated twin holds (0: it has none, or the host did not launch it for this batch); fit5: the
 * host made the "fit 5" launch (it does both for batches of more than 4 x CUs x 64 packets: alacgpu.hip: launch; a smaller batch
 * whose many keys push it over 4 x CUs slots all the same runs rounds of four); force (ALACGPU_FIT): 4 / 5 for every batch, no
 * gated twin; mono: single-channel streams — their workgroups have no U phase and no U tile, a round of five costs them nothing
 * over a round of four (16-bit mono 81 920 packets: 1.96 ms four per CU, 1.82 gated pairs, 1.57 five per CU) */
constexpr __host__ __device__ __forceinline__ uint32_t decode_mode(uint32_t items, uint32_t n_cu, uint32_t cap, uint32_t force, bool mono,
                                                                  bool fit5) {
    if (force == kModeFit4 || (force == kModeFit5 && fit5)) return force;
    if (items <= 4u * n_cu) return kModeFit4;
    if (items <= 5u * n_cu) return (cap >= 5u && !mono) ? kModeGated : fit5 ? kModeFit5 : kModeFit4;
    if (items <= 6u * n_cu && cap >= 6u) return kModeGated;
    return (2u * items <= 15u * n_cu || !fit5) ? kModeFit4 : kModeFit5;
}
/* pairs per CU the gated twin admits (k_decode_body.inc: the gate): as few as hold the batch at once */
constexpr __host__ __device__ __forceinline__ uint32_t pair_quota(uint32_t items, uint32_t n_cu, uint32_t cap) {
    const uint32_t need = (items + n_cu - 1u) / n_cu;
    return need < 5u ? 5u : need < cap ? need : cap;
}
/* The host launches the gated twin and the "fit 5" shape only for batches whose upper bound of wave slots exceeds 4 x CUs
 * (alacgpu.hip: launch): whatever is decided above, it must never hand a batch of up to 4 x CUs slots to a launch that is
 * not made for it, or that batch stays undecoded (round 3 had such a bug once, with another guess). Checked at compile time. */
constexpr bool mode_is_fit4_without_the_others(uint32_t n_cu) { /* no twin, no "fit 5" launch: rounds of four whatever the count */
    for (uint32_t items = 0u; items <= 40u * n_cu; items += (items < 8u * n_cu ? 1u : n_cu / 2u + 1u))
        if (decode_mode(items, n_cu, 0u, 0u, false, false) != kModeFit4 || decode_mode(items, n_cu, 0u, 5u, true, false) != kModeFit4) return false;
    return true;
}
constexpr bool mode_is_fit4_within_four(uint32_t n_cu) {
    for (uint32_t cap = 0u; cap <= 8u; ++cap)
        for (uint32_t items = 0u; items <= 4u * n_cu; ++items)
            if (decode_mode(items, n_cu, cap, 0u, false, true) != kModeFit4 || decode_mode(items, n_cu, cap, 0u, true, true) != kModeFit4) return false;
    return true;
}
static_assert(mode_is_fit4_without_the_others(1u) && mode_is_fit4_without_the_others(64u) && mode_is_fit4_without_the_others(256u) &&
                  mode_is_fit4_within_four(1u) && mode_is_fit4_within_four(7u) && mode_is_fit4_within_four(64u) &&
                  mode_is_fit4_within_four(256u) && mode_is_fit4_within_four(304u),
              "decode_mode would leave batches of up to 4 x CUs wave slots to a launch that is not made for them");
constexpr uint32_t kQuadLdsFit4 = 34816u; /* static + dynamic LDS of a "fit 4" launch: 4 x 34 KB <= 160 KB < 5 x 34 KB */

/* arguments of the pair kernels (k_decode_body.inc), passed as one struct */
struct PairArgs {
    alac::DevCfg cfg;
    const uint8_t* blob;
    uint64_t blob_bytes;
    const uint64_t* offsets;
    const uint32_t* sizes;
    const uint32_t* perm;
    Plan* plan;
    uint8_t* out;
    uint64_t out_stride;
    uint32_t* frames_out;
    int32_t* status;
    int32_t* scratch_u;
    const uint32_t* cu_number; /* 1 + the number of each CU (indexed like Plan::gate), 0 for CUs the census did not see */
    uint32_t* claims; /* four words per wave slot of the plan, zeroed before every decode: whoever sets the first decodes the slot */
    uint32_t ppw;
    uint32_t n_cu; /* compute units of the device */
    uint32_t cap;  /* pairs one of them holds */
    uint32_t lanes_min; /* four-wave workgroups (k_dec16q.hip): keys whose longer predictor has at least this many taps get two predictor waves */
    uint32_t fit;       /* four-wave workgroups: how many of them a CU holds with THIS launch's LDS footprint (kModeFit4 / kModeFit5) */
    uint32_t fit_force; /* experiments (ALACGPU_FIT): 4 / 5 for every batch; 0: decode_mode decides */
    uint32_t fit5;      /* 1: the "fit 5" launch is made for this batch (decode_mode) */
};
#define ALAC_DECLARE_DECODE(NAME) __global__ void NAME(PairArgs);
/* the wave pair over the regular packets, one kernel per class (sample width x channel width) */
ALAC_DECLARE_DECODE(alac_decode_16q) ALAC_DECLARE_DECODE(alac_decode_16g) ALAC_DECLARE_DECODE(alac_decode_24q) ALAC_DECLARE_DECODE(alac_decode_32q)
ALAC_DECLARE_DECODE(alac_decode_w24) ALAC_DECLARE_DECODE(alac_decode_w32)
#undef ALAC_DECLARE_DECODE
__global__ void alac_chan_predict(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                                  const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                                  const alac::ChanDesc* __restrict__ cd, int32_t* __restrict__ rows, uint64_t row_stride, uint32_t ppw);
__global__ void alac_interleave(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                                const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                                const alac::ChanDesc* __restrict__ cd, const alac::PktDesc* __restrict__ pd, const int32_t* __restrict__ rows,
                                uint64_t row_stride, uint8_t* __restrict__ out, uint64_t out_stride, uint32_t blocks_per_pkt);
__global__ void alac_interleave4(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                                const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                                const alac::ChanDesc* __restrict__ cd, const alac::PktDesc* __restrict__ pd, const int32_t* __restrict__ rows,
                                uint64_t row_stride, uint8_t* __restrict__ out, uint64_t out_stride, uint32_t blocks_per_pkt);

} /* namespace alack */
#endif
