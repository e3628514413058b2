/*
 * alacgpu.hip — gfx950 batch ALAC decode kernels + the C ABI of include/alacgpu.h.
 *
 * Replaces the reference's per-packet hot path (decoder.go:133-207 -> internal/alac golomb.go,
 * predictor.go, matrix.go) with HIP kernels over a batch of independent packets. There is no host decode
 * path in this library: every decode entry launches the kernels.
 *
 * One decode = these launches on the handle's stream:
 *   alac_classify  one thread per packet: sort key from the first element header (a few bytes read)
 *   alac_plan      one wavefront: key histogram -> packet / wave ranges (irregular keys, then the longest predictors)
 *   alac_scatter   one thread per packet: counting-sort scatter into the lane permutation
 *   alac_scan      irregular packets, one wavefront per 64: status, frame count, where each channel starts
 *   alac_decode    regular packets: a PAIR of wavefronts per 64 same-key packets (alac_duo.h): entropy wave and
 *                  predictor / PCM wave, residuals through an LDS queue; PCM staged in LDS, written as 128-B lines
 *   alac_task_classify / alac_plan / alac_scatter / alac_chan_decode   (> 2 channels) the same pair per 64
 *                  (packet, channel) tasks the scan found, int32 rows
 *   alac_interleave, alac_legacy   PCM of the scanned packets (frame order), whole-packet decoder for the rest
 * HBM traffic per packet: compressed bytes in, PCM bytes out, plus the U-channel hand-off tile of stereo pairs
 * ((frame_length + 1) x 64 x int32 per workgroup, row-coalesced, written once and read once) or the sample rows
 * of the split pipeline.
 */
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
#define ALAC_MULU24(a, b) __umul24((unsigned)(a), (unsigned)(b))
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
#ifdef ALAC_DUO_PROF
/* profiling build: cycles (s_memtime) between the stamps of alac_duo.h, summed per role over all waves */
__device__ unsigned long long g_duo_prof[16];
#define ALAC_DUO_STAMP(k)                                                   \
    do {                                                                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
        if ((k) > 0) wv.prof[(k) - 1] += t_ - wv.prof_t;                     \
        wv.prof_t = t_;                                                     \
    } while (0)
#endif
#include "alac_wave.h"
#include "alac_regular.h"
#include "alac_duo.h"
#include "alac_split.h"

/* s_setprio levels of the wave pair (see alac_decode) */
#ifndef ALAC_PRIO_B_LONG
#define ALAC_PRIO_B_LONG 3  /* predictor waves, order > 8 */
#define ALAC_PRIO_B_MID 2   /* order 6..8 */
#define ALAC_PRIO_B_SHORT 1 /* order < 6 */
#define ALAC_PRIO_A 2       /* entropy waves */
#endif

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kWave = 64;
constexpr uint32_t kTimingSlots = 64;
constexpr uint32_t kRowStride = 65; /* dwords per lane row in LDS: odd stride = conflict-free column access */
constexpr uint32_t kRing = 64;      /* dwords of PCM a lane row holds (two 128-B chunks) */
constexpr uint32_t kFallbackSlots = 64;
constexpr uint32_t kRingStride = 36; /* 32 ring dwords + 4: rows stay 16-byte aligned, lanes spread over banks */

/* device-side launch plan, rebuilt by every decode */
/* sort keys: 0..2047 regular packets (numU*32 + numV + KEY_WIDE, alac_regular.h); 2048 / 2049 irregular packets
 * (below). A workgroup holds packets of ONE key. */
constexpr uint32_t kKeys = alac::KEY_IRREGULAR + alac::NUM_CLASSES;
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
    uint32_t irr_waves; /* waves of the irregular keys (>= KEY_IRREGULAR): they come first */
};

/* LDS of the decode kernel (one wave per workgroup). Referenced by name, never through a generic pointer, so
 * every access is a ds_* instruction (a pointer kept in a struct decays to flat_* loads and stores). */
__shared__ uint32_t s_rows[kWave * kRowStride];                                  /* PCM stager rows */
__shared__ unsigned long long s_optr[kWave];                                     /* PCM slot of each lane's packet */
__shared__ __attribute__((aligned(16))) uint32_t s_ring[kWave * kRingStride];    /* bitstream rings */
/* residual queue of the wave pair (alac_duo.h), A -> B, double-buffered chunks */
constexpr uint32_t kQ = alac::DUO_CHUNK;
__shared__ int32_t s_rq[2 * kQ * kWave];

/* U hand-off tile of one wave: frame_length rows of 64 cells and one spare row (the single-wave decoders read one
 * row ahead) */
__host__ __device__ inline size_t u_tile_cells(uint32_t frame_length) { return ((size_t)frame_length + 1u) * kWave; }

/* ---- gfx950 wave policy for alac::decode_wave --------------------------------------------------------- */
struct GpuWave {
    int32_t* u_tile;           /* HBM: this lane's column of the wave's U hand-off tile */
    int32_t* g_tile;           /* HBM: this lane's column of the wave's fall-back tile */
    uint8_t* my_out;
    uint32_t lane, wcnt, flushed;
    uint32_t ppw;              /* packets (= live lanes) per wave; also the row stride of the HBM tiles */
#ifdef ALAC_DUO_PROF
    unsigned long long prof[4] = {0, 0, 0, 0}, prof_t = 0;
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
    ALAC_DEV void st_begin(uint8_t* out) {
        my_out = out;
        s_optr[lane] = (unsigned long long)reinterpret_cast<uintptr_t>(out);
        wcnt = flushed = 0;
    }
    ALAC_DEV void st_push(uint32_t v) {
        s_rows[lane * kRowStride + (wcnt & (kRing - 1u))] = v;
        ++wcnt;
    }
    /* branch-free form: a lane that is not `on` rewrites its next free slot and does not advance */
    ALAC_DEV void st_push_if(uint32_t v, bool on) {
        s_rows[lane * kRowStride + (wcnt & (kRing - 1u))] = v;
        wcnt += on ? 1u : 0u;
    }
    /* bytes of the last, incomplete dword of the stream (after every dword pushed so far) */
    ALAC_DEV void st_tail(uint64_t acc, uint32_t nbytes) {
        for (uint32_t b = 0; b < nbytes; ++b) my_out[(size_t)wcnt * 4u + b] = (uint8_t)(acc >> (8u * b));
    }
    /* Collective. Rows that just completed a 32-dword chunk are written out as 128-B lines: store
     * instruction k covers packets 8k..8k+7, eight lanes x 16 B per packet. Lock step makes `flushed`
     * identical in all full lanes. */
    ALAC_DEV void st_step() {
        const bool full = (wcnt - flushed) >= 32u;
        const unsigned long long mask = __ballot(full);
        if (mask == 0ull) return;
        __builtin_amdgcn_wave_barrier();
        const int first = __ffsll((long long)mask) - 1;
        const uint32_t fl = (uint32_t)__shfl((int)flushed, first, 64);
        const uint32_t col0 = fl & (kRing - 1u);
        const uint32_t piece = lane & 7u;
        const uint32_t groups = (ppw + 7u) >> 3;
        for (uint32_t k = 0; k < groups; ++k) {
            const uint32_t q = 8u * k + (lane >> 3);
            if ((mask >> q) & 1ull) {
                const uint32_t* r = s_rows + q * kRowStride + col0 + piece * 4u;
                const uint4 v = make_uint4(r[0], r[1], r[2], r[3]);
                uint8_t* dst = reinterpret_cast<uint8_t*>((uintptr_t)s_optr[q]) + ((size_t)fl + piece * 4u) * 4u;
                /* the address came through LDS as an integer: name the global address space, or it is a flat store */
                *reinterpret_cast<__attribute__((address_space(1))) u32x4*>((uintptr_t)dst) = u32x4{v.x, v.y, v.z, v.w};
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (full) flushed += 32u;
    }
    ALAC_DEV uint32_t st_finish() {
        for (uint32_t w = flushed; w < wcnt; ++w)
            *reinterpret_cast<uint32_t*>(my_out + (size_t)w * 4u) = s_rows[lane * kRowStride + (w & (kRing - 1u))];
        flushed = wcnt;
        return wcnt;
    }
    /* bitstream ring of the entropy wave: 32 dwords per lane, rows of kRingStride dwords (16-byte aligned) */
    ALAC_DEV void ring_write4(uint32_t slot, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        /* (slot >> 2) * 4 lets the compiler see the 16-byte alignment: one ds_write_b128 */
        *reinterpret_cast<uint4*>(&s_ring[lane * kRingStride + (slot >> 2) * 4u]) = make_uint4(a, b, c, d);
    }
    ALAC_DEV uint32_t ring_read(uint32_t slot) const { return s_ring[lane * kRingStride + slot]; }
    /* residual queue: row j of buffer buf holds step j of the chunk for all 64 lanes (conflict-free) */
    ALAC_DEV void rq_write(uint32_t buf, uint32_t j, int32_t v) { s_rq[(buf * kQ + j) * kWave + lane] = v; }
    ALAC_DEV int32_t rq_read(uint32_t buf, uint32_t j) const { return s_rq[(buf * kQ + j) * kWave + lane]; }
    /* chunk hand-over between the two waves of the workgroup: LDS traffic only, so outstanding global loads
     * (ring refills, U prefetch) and stores (U tile) are NOT waited for — __syncthreads() would drain them */
    ALAC_DEV void duo_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    /* end of the U phase: wave B's tile stores must have landed before wave A loads them */
    ALAC_DEV void duo_sync_mem() {
        __threadfence_block();
        __syncthreads();
    }
    /* rows of 64 cells whatever ppw is: a constant stride lets unrolled steps address their rows by immediate
     * offsets from one base, and every lane (with or without a packet) owns a column */
    ALAC_DEV int32_t* u_row(uint32_t i) const { return u_tile + (size_t)i * kWave; }
    ALAC_DEV int32_t* g_slot(uint32_t k) const { return g_tile + (size_t)k * ppw; }
};

/* readable bytes of the blob from a packet's start, as Bits wants them */
__device__ __forceinline__ uint32_t avail_of(uint64_t blob_bytes, uint64_t off) {
    const uint64_t left = blob_bytes - off;
    return left > 0xffffffffull ? 0xffffffffu : (uint32_t)left;
}

/* Packet descriptors are checked here, once: a packet must lie inside the blob (the caller's offsets and sizes are
 * untrusted device data). One that does not gets ALACGPU_ERR_RANGE, no sort key, and is never looked at again; the
 * sizes every later kernel uses are the checked copies in sizes_ws. d_sizes may be null: packet i is then
 * blob[offsets[i], offsets[i+1]) (the host entry's offsets[n+1]).
 * Sort-key histogram of one 256-thread block in LDS; only the keys the block saw go to the global counters
 * (a batch has a dozen distinct keys: per-packet global atomics on them serialise). */
__global__ void __launch_bounds__(256)
alac_classify(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
              const uint32_t* __restrict__ sizes, uint32_t n, uint16_t* __restrict__ keys, uint32_t* __restrict__ sizes_ws,
              uint32_t* __restrict__ frames_out, int32_t* __restrict__ status, Plan* plan) {
    __shared__ uint32_t hist[kKeys];
    for (uint32_t k = threadIdx.x; k < kKeys; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint64_t off = offsets[i];
        uint64_t sz = sizes ? (uint64_t)sizes[i] : offsets[i + 1] - off;
        const bool ok = off <= blob_bytes && sz <= blob_bytes - off && sz <= 0x0fffffffull &&
                        (sizes || offsets[i + 1] >= off);
        if (!ok) {
            keys[i] = (uint16_t)alac::TASK_NONE;
            sizes_ws[i] = 0;
            frames_out[i] = 0;
            status[i] = ALACGPU_ERR_RANGE;
        } else if (sz == 0) {
            /* an empty packet: PastEnd before the first tag (decoder.go:143-145). Settled here so that the readers
             * only ever see packets of at least one byte (their loads are anchored on the packet's last byte). */
            keys[i] = (uint16_t)alac::TASK_NONE;
            sizes_ws[i] = 0;
            frames_out[i] = 0;
            status[i] = ALACGPU_STATUS(alac::ST_OVERRUN, 0, 0);
        } else {
            sizes_ws[i] = (uint32_t)sz;
            const uint8_t* p = blob + off;
            uint32_t key = alac::classify_regular(cfg, p, (uint32_t)sz, avail_of(blob_bytes, off));
            /* not regular: scan first (with a usable KB). More than two channels: split pipeline. One or two: escape
             * elements are unpacked by alac_interleave, anything else is handed to the whole-packet decoder. */
            if (key == alac::KEY_IRREGULAR) key = cfg.kb != 0 ? kKeyScan : kKeyLegacy;
            keys[i] = (uint16_t)key;
            atomicAdd(&hist[key], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kKeys; k += blockDim.x)
        if (hist[k]) atomicAdd(&plan->count[k], hist[k]);
}

/* One wavefront: exclusive scan of the key histogram in dispatch order (highest key first: irregular packets, then
 * the longest predictors; a kernel ends when its last wave does, so the slowest waves get the lowest block ids).
 * Each lane owns a run of consecutive dispatch positions; the lane totals are scanned with shuffles. */
__global__ void __launch_bounds__(kWave) alac_plan(Plan* plan, uint32_t ppw) {
    __shared__ uint32_t cnt[kKeys];
    for (uint32_t k = threadIdx.x; k < kKeys; k += kWave) cnt[k] = plan->count[k];
    __syncthreads();
    constexpr uint32_t R = (kKeys + kWave - 1) / kWave;
    const uint32_t q0 = threadIdx.x * R; /* dispatch position q holds key kKeys - 1 - q */
    uint32_t p = 0, w = 0, z = 0, wi = 0;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t q = q0 + r;
        if (q >= kKeys) break;
        const uint32_t key = kKeys - 1u - q;
        const uint32_t c = cnt[key];
        const uint32_t cw = (c + ppw - 1) / ppw;
        p += c;
        w += cw;
        z += c ? 1u : 0u;
        wi += key >= alac::KEY_IRREGULAR ? cw : 0u;
    }
    /* inclusive scan over the 64 lanes, then make it exclusive */
    uint32_t ip = p, iw = w, iz = z, ii = wi;
#pragma unroll
    for (int o = 1; o < (int)kWave; o <<= 1) {
        const uint32_t tp = (uint32_t)__shfl_up((int)ip, o, kWave), tw = (uint32_t)__shfl_up((int)iw, o, kWave);
        const uint32_t tz = (uint32_t)__shfl_up((int)iz, o, kWave), ti = (uint32_t)__shfl_up((int)ii, o, kWave);
        if ((int)threadIdx.x >= o) {
            ip += tp;
            iw += tw;
            iz += tz;
            ii += ti;
        }
    }
    uint32_t ep = ip - p, ew = iw - w, ez = iz - z;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t q = q0 + r;
        if (q >= kKeys) break;
        const uint32_t key = kKeys - 1u - q;
        const uint32_t c = cnt[key];
        plan->pkt_start[key] = ep;
        plan->cursor[key] = 0;
        if (c) {
            plan->list_key[ez] = key;
            plan->list_wave0[ez] = ew;
            ++ez;
            ep += c;
            ew += (c + ppw - 1) / ppw;
        }
    }
    if (threadIdx.x == kWave - 1u) {
        plan->nk = iz;
        plan->total_waves = iw;
        plan->irr_waves = ii;
    }
}

/* Counting-sort scatter. A block reserves one range per key it holds with a single global atomic and hands out
 * the slots inside it from LDS. The order of packets inside a key is arbitrary (and may differ run to run);
 * results do not depend on it. */
__global__ void __launch_bounds__(256)
alac_scatter(const uint16_t* __restrict__ keys, uint32_t n, Plan* plan, uint32_t* __restrict__ perm) {
    __shared__ uint32_t hist[kKeys];
    __shared__ uint32_t base[kKeys];
    for (uint32_t k = threadIdx.x; k < kKeys; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t key = 0, local = 0;
    if (i < n) {
        key = keys[i];
        if (key != alac::TASK_NONE) local = atomicAdd(&hist[key], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kKeys; k += blockDim.x)
        if (hist[k]) base[k] = plan->pkt_start[k] + atomicAdd(&plan->cursor[k], hist[k]);
    __syncthreads();
    if (i < n && key != alac::TASK_NONE) perm[base[key] + local] = i;
}

/* Irregular packets (keys >= KEY_IRREGULAR; they own the first plan->irr_waves wave slots): one wavefront per 64 packets.
 * With a usable KB they are scanned (status, frame count, channel descriptors: split pipeline step 1, PCM comes
 * from the later kernels); with KB == 0 the whole-packet decoder takes them. */
__global__ void __launch_bounds__(kWave)
alac_scan(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
          const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
          uint8_t* __restrict__ out, uint64_t out_stride, uint32_t* __restrict__ frames_out,
          int32_t* __restrict__ status, int32_t* __restrict__ scratch_u, int32_t* __restrict__ scratch_g,
          uint32_t ppw, alac::ChanDesc* __restrict__ cd, alac::PktDesc* __restrict__ pd) {
    const uint32_t b = blockIdx.x;
    if (b >= plan->irr_waves) return;
    uint32_t e = 0;
    for (uint32_t t = 1; t < plan->nk; ++t)
        if (plan->list_wave0[t] <= b) e = t;
    const uint32_t key = plan->list_key[e];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = (b - plan->list_wave0[e]) * ppw + lane;
    const bool live = lane < ppw && idx < plan->count[key];
    const uint32_t pkt = live ? perm[plan->pkt_start[key] + idx] : 0u;

    GpuWave wv;
    wv.u_tile = scratch_u + (size_t)b * u_tile_cells(cfg.frame_length) + lane;
    wv.g_tile = scratch_g + (size_t)b * kFallbackSlots * ppw + lane;
    wv.ppw = ppw;
    wv.my_out = nullptr;
    wv.lane = lane;
    wv.wcnt = wv.flushed = 0;

    /* lanes without a packet read nothing (size 0) */
    const uint64_t off = live ? offsets[pkt] : 0ull;
    const uint8_t* p = blob + off;
    const uint32_t size = live ? sizes[pkt] : 0u;
    const uint32_t avail = avail_of(blob_bytes, off);
    uint8_t* o = out + (size_t)pkt * out_stride;
    uint32_t frames = 0;
    int32_t st;
    const uint32_t ukey = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    if (ukey == kKeyScan)
        st = alac::decode_wave<GpuWave, 16, true, true>(wv, cfg, live, p, size, avail, o, &frames, cd + (size_t)pkt * 8u, pd + pkt);
    else
        st = alac::decode_wave<GpuWave, 16, true>(wv, cfg, live, p, size, avail, o, &frames);
    if (live) {
        frames_out[pkt] = frames;
        status[pkt] = st;
    }
}

/* Regular packets: a pair of wavefronts per 64 packets with the SAME key (alac_duo.h): wave 0 = role A (entropy),
 * wave 1 = role B (predictor + PCM), lane = packet in both. Two waves per SIMD (four workgroups per CU) is what
 * the pair is built for: the register budget is capped there. */
__global__ void __launch_bounds__(2 * kWave, 2)
alac_decode(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
            const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
            uint8_t* __restrict__ out, uint64_t out_stride, uint32_t* __restrict__ frames_out,
            int32_t* __restrict__ status, int32_t* __restrict__ scratch_u, uint32_t ppw) {
    const uint32_t b = blockIdx.x + plan->irr_waves;
    if (b >= plan->total_waves) return;
    /* which key owns wave slot b: last list entry whose first wave is <= b (a handful of entries) */
    uint32_t e = 0;
    for (uint32_t t = 1; t < plan->nk; ++t)
        if (plan->list_wave0[t] <= b) e = t;
    const uint32_t key = plan->list_key[e];
    const uint32_t lane = threadIdx.x & (kWave - 1u);
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t idx = (b - plan->list_wave0[e]) * ppw + lane;
    const bool live = lane < ppw && idx < plan->count[key];
    const uint32_t pkt = live ? perm[plan->pkt_start[key] + idx] : 0u;

    GpuWave wv;
    /* U tile: a column per lane (the decoders store without a branch, so lanes without a packet need cells too) */
    wv.u_tile = scratch_u + (size_t)b * u_tile_cells(cfg.frame_length) + lane;
    wv.g_tile = nullptr;
    wv.ppw = ppw;
    wv.my_out = nullptr;
    wv.lane = lane;
    wv.wcnt = wv.flushed = 0;

    /* lanes without a packet read nothing (size 0) */
    const uint64_t off = live ? offsets[pkt] : 0ull;
    const uint8_t* p = blob + off;
    const uint32_t size = live ? sizes[pkt] : 0u;
    const uint32_t avail = avail_of(blob_bytes, off);
    uint8_t* o = out + (size_t)pkt * out_stride;
    uint32_t frames = 0;
    /* the key is wave-uniform (one key per workgroup): scalar branches pick the variant */
    const uint32_t ukey = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    if (role != 0u) {
        /* Issue priority goes to whichever wave is the longer one of its pair, and among pairs to the slowest (the
         * kernel ends with its slowest workgroup): role B grows by nine instructions per tap, role A does not, so
         * long predictors go ahead of everything, mid-length ones level with the entropy waves, short ones behind
         * (measured on the benchmark mix: 3.15 ms with the entropy waves on top, 2.73 ms this way). */
        const uint32_t na_max = max((ukey >> 5) & 31u, ukey & 31u);
        if (na_max > 8u && na_max != 31u) __builtin_amdgcn_s_setprio(ALAC_PRIO_B_LONG);
        else if (na_max >= 6u && na_max != 31u) __builtin_amdgcn_s_setprio(ALAC_PRIO_B_MID);
        else __builtin_amdgcn_s_setprio(ALAC_PRIO_B_SHORT);
        (void)alac::decode_regular_duo<GpuWave, alac::ROLE_B>(wv, cfg, ukey, live, p, size, avail, o, &frames);
#ifdef ALAC_DUO_PROF
        if (lane == 0)
            for (int k = 0; k < 4; ++k) atomicAdd(&g_duo_prof[8 + k], wv.prof[k]);
#endif
        return;
    }
    /* the entropy chain is serial: it issues whenever it can, shorter predictor waves (many independent
     * instructions) fill the slots in between */
    {
        /* where the PCM writer runs in wave A (single channels, alac_duo.h) A is the longer wave of the pair */
        const bool cpe = cfg.num_channels == 2;
        if (!(ukey & alac::KEY_WIDE) && alac::duo_emit_in_a(cpe ? (ukey & 31u) : ((ukey >> 5) & 31u), cpe)) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(ALAC_PRIO_A);
    }
    const int32_t st = alac::decode_regular_duo<GpuWave, alac::ROLE_A>(wv, cfg, ukey, live, p, size, avail, o, &frames);
#ifdef ALAC_DUO_PROF
    if (lane == 0)
        for (int k = 0; k < 4; ++k) atomicAdd(&g_duo_prof[k], wv.prof[k]);
#endif
    if (live) {
        frames_out[pkt] = frames;
        status[pkt] = st;
    }
}

/* ---- split pipeline (alac_split.h) -------------------------------------------------------------------- */
/* one thread per (packet, bitstream channel): sort key of the channel task, or TASK_NONE */
__global__ void __launch_bounds__(256)
alac_task_classify(alac::DevCfg cfg, const alac::ChanDesc* __restrict__ cd, const alac::PktDesc* __restrict__ pd,
                   const uint16_t* __restrict__ pkt_keys, uint32_t n_slots, uint16_t* __restrict__ keys, Plan* plan) {
    __shared__ uint32_t hist[alac::NUM_TASK_KEYS];
    if (threadIdx.x < alac::NUM_TASK_KEYS) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_slots) {
        const uint32_t pkt = t >> 3, slot = t & 7u;
        uint32_t key = alac::TASK_NONE;
        if (pkt_keys[pkt] == kKeyScan) {
            const alac::PktDesc q = pd[pkt];
            if (q.status == 0 && q.route == alac::ROUTE_SPLIT && slot < q.nslots) {
                const alac::ChanDesc d = cd[t];
                if ((d.info & alac::CD_VALID) && !(d.info & alac::CD_ESCAPE)) key = alac::chan_task_key(cfg, d);
            }
        }
        keys[t] = (uint16_t)key;
        if (key != alac::TASK_NONE) atomicAdd(&hist[key], 1u);
    }
    __syncthreads();
    if (threadIdx.x < alac::NUM_TASK_KEYS && hist[threadIdx.x]) atomicAdd(&plan->count[threadIdx.x], hist[threadIdx.x]);
}

/* a wave pair per 64 channel tasks with the same key: int32 samples of the channel into its row */
__global__ void __launch_bounds__(2 * kWave, 2)
alac_chan_decode(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                 const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                 const alac::ChanDesc* __restrict__ cd, int32_t* __restrict__ rows, uint64_t row_stride, uint32_t ppw) {
    const uint32_t b = blockIdx.x;
    if (b >= plan->total_waves) return;
    uint32_t e = 0;
    for (uint32_t t = 1; t < plan->nk; ++t)
        if (plan->list_wave0[t] <= b) e = t;
    const uint32_t key = plan->list_key[e];
    const uint32_t lane = threadIdx.x & (kWave - 1u);
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t idx = (b - plan->list_wave0[e]) * ppw + lane;
    const bool live = lane < ppw && idx < plan->count[key];
    const uint32_t t = live ? perm[plan->pkt_start[key] + idx] : 0u;
    const uint32_t pkt = t >> 3, slot = t & 7u;

    GpuWave wv;
    wv.u_tile = nullptr;
    wv.g_tile = nullptr;
    wv.ppw = ppw;
    wv.my_out = nullptr;
    wv.lane = lane;
    wv.wcnt = wv.flushed = 0;

    /* lanes without a packet read nothing (size 0) */
    const uint64_t off = live ? offsets[pkt] : 0ull;
    const uint8_t* p = blob + off;
    const uint32_t size = live ? sizes[pkt] : 0u;
    const uint32_t avail = avail_of(blob_bytes, off);
    alac::ChanDesc d = cd[t];
    if (!live) d.hdr_pos = d.ent_pos = d.ns = 0;
    int32_t* row = rows + ((size_t)pkt * cfg.num_channels + slot) * row_stride;
    const uint32_t ukey = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    if (role != 0u) {
        const uint32_t na = ukey & 31u; /* same priorities as alac_decode */
        if (na > 8u && na != 31u) __builtin_amdgcn_s_setprio(ALAC_PRIO_B_LONG);
        else if (na >= 6u && na != 31u) __builtin_amdgcn_s_setprio(ALAC_PRIO_B_MID);
        else __builtin_amdgcn_s_setprio(ALAC_PRIO_B_SHORT);
        alac::decode_channel_task<GpuWave, alac::ROLE_B>(wv, cfg, ukey, live, p, size, avail, d, row);
        return;
    }
    __builtin_amdgcn_s_setprio(ALAC_PRIO_A);
    alac::decode_channel_task<GpuWave, alac::ROLE_A>(wv, cfg, ukey, live, p, size, avail, d, row);
}

/* one thread per (packet, frame) of the split packets: PCM in frame order. Blocks stride over the scanned
 * packets (the tail of the permutation that belongs to kKeyScan) x 256-frame slices. A slice is assembled in LDS
 * (a frame is 1..32 bytes at a byte offset of its own) and copied out as 16-byte pieces, whole lines at a time. */
__global__ void __launch_bounds__(256)
alac_interleave(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                const alac::ChanDesc* __restrict__ cd, const alac::PktDesc* __restrict__ pd, const int32_t* __restrict__ rows,
                uint64_t row_stride, uint8_t* __restrict__ out, uint64_t out_stride, uint32_t blocks_per_pkt) {
    __shared__ __attribute__((aligned(16))) uint8_t s_slice[256 * 32];
    const uint32_t n_scan = plan->count[kKeyScan];
    const uint32_t first = plan->pkt_start[kKeyScan];
    const uint64_t items = (uint64_t)n_scan * blocks_per_pkt;
    const uint32_t fb = cfg.num_channels * cfg.bps;
    for (uint64_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint32_t pkt = perm[first + (uint32_t)(it / blocks_per_pkt)];
        const alac::PktDesc q = pd[pkt];
        if (q.status != 0 || q.route != alac::ROUTE_SPLIT) continue; /* block-uniform */
        const uint32_t f0 = (uint32_t)(it % blocks_per_pkt) * blockDim.x;
        if (f0 >= q.frames) continue;
        const uint32_t nf = min(q.frames - f0, (uint32_t)blockDim.x);
        const uint32_t f = f0 + threadIdx.x;
        if (threadIdx.x < nf)
            alac::interleave_frame(cfg, blob + offsets[pkt], sizes[pkt], avail_of(blob_bytes, offsets[pkt]), q, cd + (size_t)pkt * 8u,
                                   rows + (size_t)pkt * cfg.num_channels * row_stride, (size_t)row_stride, f,
                                   s_slice + threadIdx.x * fb);
        __syncthreads();
        uint8_t* dst = out + (size_t)pkt * out_stride + (size_t)f0 * fb; /* f0 * fb is a multiple of 256 */
        const uint32_t total = nf * fb;
        if (cfg.aligned16) {
            for (uint32_t k = threadIdx.x * 16u; k + 16u <= total; k += 256u * 16u)
                *reinterpret_cast<uint4*>(dst + k) = *reinterpret_cast<const uint4*>(s_slice + k);
            for (uint32_t k = (total & ~15u) + threadIdx.x; k < total; k += 256u) dst[k] = s_slice[k];
        } else {
            for (uint32_t k = threadIdx.x; k < total; k += 256u) dst[k] = s_slice[k];
        }
        __syncthreads();
    }
}

/* packets the scan routed to the whole-packet decoder (orders 17..30): same wave mapping as alac_decode */
__global__ void __launch_bounds__(kWave)
alac_legacy(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
            const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
            const alac::PktDesc* __restrict__ pd, uint8_t* __restrict__ out, uint64_t out_stride,
            uint32_t* __restrict__ frames_out, int32_t* __restrict__ status, int32_t* __restrict__ scratch_u,
            int32_t* __restrict__ scratch_g, uint32_t ppw) {
    const uint32_t b = blockIdx.x;
    if (b >= plan->total_waves) return;
    uint32_t e = 0;
    for (uint32_t t = 1; t < plan->nk; ++t)
        if (plan->list_wave0[t] <= b) e = t;
    const uint32_t key = plan->list_key[e];
    if (key != kKeyScan) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = (b - plan->list_wave0[e]) * ppw + lane;
    const bool in_wave = lane < ppw && idx < plan->count[key];
    const uint32_t pkt = in_wave ? perm[plan->pkt_start[key] + idx] : 0u;
    const bool live = in_wave && pd[pkt].status == 0 && pd[pkt].route == alac::ROUTE_LEGACY;
    if (__ballot(live) == 0ull) return;

    GpuWave wv;
    wv.u_tile = scratch_u + (size_t)b * u_tile_cells(cfg.frame_length) + lane;
    wv.g_tile = scratch_g + (size_t)b * kFallbackSlots * ppw + lane;
    wv.ppw = ppw;
    wv.my_out = nullptr;
    wv.lane = lane;
    wv.wcnt = wv.flushed = 0;
    /* lanes without a packet read nothing (size 0) */
    const uint64_t off = live ? offsets[pkt] : 0ull;
    const uint8_t* p = blob + off;
    const uint32_t size = live ? sizes[pkt] : 0u;
    const uint32_t avail = avail_of(blob_bytes, off);
    uint32_t frames = 0;
    const int32_t st = alac::decode_wave<GpuWave, 16, true>(wv, cfg, live, p, size, avail, out + (size_t)pkt * out_stride, &frames);
    if (live) {
        frames_out[pkt] = frames;
        status[pkt] = st;
    }
}

thread_local char g_err[512] = "";

void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ALACGPU_E_HIP;                                                             \
        }                                                                                     \
    } while (0)

int bytes_per_sample(uint8_t depth) { /* BytesPerSample, internal/alac/format.go:23-34 */
    switch (depth) {
        case 16: return 2;
        case 20:
        case 24: return 3;
        case 32: return 4;
        default: return 0;
    }
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return ALACGPU_E_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return ALACGPU_E_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct HostBuf { /* pinned staging */
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return ALACGPU_E_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return ALACGPU_E_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

} /* namespace */

/* ---- host-side copy helpers of the host entry -------------------------------------------------------------- */
namespace {

/* A few worker threads for the staging copies of alacgpu_decode_batch (pageable caller memory <-> pinned staging):
 * one thread moves ~10 GB/s, the PCIe link 50+. Created on first use, joined by alacgpu_destroy. */
class CopyPool {
public:
    explicit CopyPool(unsigned threads) {
        for (unsigned t = 0; t < threads; ++t) workers_.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    /* fn(k) for k in [0, tasks), on the workers and on the caller; returns when all are done */
    void parallel_for(size_t tasks, const std::function<void(size_t)>& fn) {
        if (tasks == 0) return;
        if (workers_.empty() || tasks == 1) {
            for (size_t k = 0; k < tasks; ++k) fn(k);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            tasks_ = tasks;
            next_ = 0;
            left_ = tasks;
            ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return left_ == 0; });
        fn_ = nullptr;
    }

private:
    void work() {
        for (;;) {
            size_t k;
            const std::function<void(size_t)>* fn;
            {
                std::lock_guard<std::mutex> g(m_);
                if (!fn_ || next_ >= tasks_) return;
                k = next_++;
                fn = fn_;
            }
            (*fn)(k);
            std::lock_guard<std::mutex> g(m_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void run() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t tasks_ = 0, next_ = 0, left_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

constexpr int kSlots = 3; /* chunks in flight in the host entry: one uploading, one decoding, one downloading */

/* one chunk of the host entry: device and pinned staging for its packets and its PCM */
struct Slot {
    DevBuf d_in;   /* [offsets (n+1) x u64 | packet bytes] */
    DevBuf d_out;  /* [PCM n x d_stride | frames n x u32 | status n x i32] */
    HostBuf h_in, h_out;
    hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
    size_t first = 0, n = 0;
    bool busy = false;
};

} /* namespace */

struct alacgpu_decoder {
    alacgpu_config cfg;
    alac::DevCfg dev_cfg;
    int device;
    size_t frame_bytes;
    hipStream_t stream;                                       /* kernels */
    hipStream_t s_in, s_out;                                  /* host entry: uploads, downloads */
    hipEvent_t ev_start[kTimingSlots], ev_stop[kTimingSlots]; /* ring of per-launch event pairs */
    uint64_t launches;                                       /* since the last timing reset */
    DevBuf scratch_u, scratch_g, plan, cls, perm, sizes_ws;  /* kernel workspace */
    DevBuf cd, pd, plan2, keys2, perm2, rows;                /* split pipeline (more than two channels) */
    Slot slots[kSlots];                                      /* host-entry staging */
    CopyPool* pool;
    size_t chunk_bytes;                                      /* host entry: target bytes (in + out) per chunk */
};

namespace {

/* upper bound on the waves of a batch: every key present may end in one partly filled wave */
size_t max_waves(size_t n, uint32_t ppw) { return (n + ppw - 1) / ppw + std::min<size_t>(n, 18 * 18 + 8); }

/* Packets per wave. A VALU instruction costs the SIMD the same whether 64 lanes or 8 are live (profiles/microbench/
 * valu_multi_mi355x.txt, "half" rows), so waves are kept full while there is at least one workgroup per CU (256); a
 * smaller batch is spread over narrower waves to use CUs that would otherwise idle — but no further: a packet is a
 * serial chain whose speed is highest when its wave pair has a CU to itself (4 096 stereo packets: 2.57 ms on 1 024
 * four-lane pairs, 2.19 ms on 256 sixteen-lane pairs; 32 768 packets: 2.63 ms on 1 024 half-full pairs, 2.43 ms on 512
 * full ones). */
uint32_t pick_ppw(size_t n) {
    if (const char* e = getenv("ALACGPU_PPW")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) return (uint32_t)v;
    }
    uint32_t ppw = kWave;
    while (ppw > 1 && n / ppw < 256) ppw >>= 1;
    return ppw;
}

size_t row_stride_of(uint32_t frame_length) { return ((size_t)frame_length + 3u) & ~(size_t)3u; } /* 16-byte rows */

int reserve_workspace(alacgpu_decoder* dec, size_t n, uint32_t ppw) {
    const size_t waves = max_waves(n, ppw);
    int rc;
    if ((rc = dec->scratch_u.ensure(waves * u_tile_cells(dec->cfg.frame_length) * sizeof(int32_t)))) return rc;
    if ((rc = dec->scratch_g.ensure(waves * kFallbackSlots * ppw * sizeof(int32_t)))) return rc;
    if ((rc = dec->plan.ensure(sizeof(Plan)))) return rc;
    if ((rc = dec->cls.ensure((n ? n : 1) * sizeof(uint16_t)))) return rc;
    if ((rc = dec->perm.ensure((n ? n : 1) * sizeof(uint32_t)))) return rc;
    if ((rc = dec->sizes_ws.ensure((n ? n : 1) * sizeof(uint32_t)))) return rc;
    const size_t ns = (n ? n : 1) * 8;
    if ((rc = dec->cd.ensure(ns * sizeof(alac::ChanDesc)))) return rc;
    if ((rc = dec->pd.ensure((n ? n : 1) * sizeof(alac::PktDesc)))) return rc;
    if (dec->cfg.num_channels > 2) {
        if ((rc = dec->plan2.ensure(sizeof(Plan)))) return rc;
        if ((rc = dec->keys2.ensure(ns * sizeof(uint16_t)))) return rc;
        if ((rc = dec->perm2.ensure(ns * sizeof(uint32_t)))) return rc;
        if ((rc = dec->rows.ensure((n ? n : 1) * dec->cfg.num_channels * row_stride_of(dec->cfg.frame_length) * sizeof(int32_t))))
            return rc;
    }
    return ALACGPU_E_OK;
}

/* All kernels of one decode, on the handle's stream. d_sizes may be null (packet i = blob[offsets[i], offsets[i+1])).
 * The event pair brackets everything the decode launches, the sort pre-pass included. */
int launch(alacgpu_decoder* dec, const uint8_t* d_blob, uint64_t blob_bytes, const uint64_t* d_offsets,
           const uint32_t* d_sizes, size_t n, uint8_t* d_out, size_t out_stride, uint32_t* d_frames, int32_t* d_status) {
    if (n == 0) return ALACGPU_E_OK;
    if (n > 0x7fffffffu) {
        set_err("batch too large");
        return ALACGPU_E_ARG;
    }
    const uint32_t ppw = pick_ppw(n);
    int rc = reserve_workspace(dec, n, ppw);
    if (rc) return rc;
    alac::DevCfg c = dec->dev_cfg;
    c.aligned16 = (out_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(d_out) % 16) == 0) ? 1u : 0u;
    Plan* plan = (Plan*)dec->plan.p;
    const uint32_t* sz = (const uint32_t*)dec->sizes_ws.p; /* the checked sizes (alac_classify) */
    const uint32_t nb = (uint32_t)((n + 255) / 256);
    const uint32_t slot = (uint32_t)(dec->launches % kTimingSlots);
    HIP_TRY(hipEventRecord(dec->ev_start[slot], dec->stream));
    HIP_TRY(hipMemsetAsync(plan, 0, sizeof(Plan), dec->stream));
    hipLaunchKernelGGL(alac_classify, dim3(nb), dim3(256), 0, dec->stream, c, d_blob, blob_bytes, d_offsets, d_sizes,
                       (uint32_t)n, (uint16_t*)dec->cls.p, (uint32_t*)dec->sizes_ws.p, d_frames, d_status, plan);
    hipLaunchKernelGGL(alac_plan, dim3(1), dim3(kWave), 0, dec->stream, plan, ppw);
    hipLaunchKernelGGL(alac_scatter, dim3(nb), dim3(256), 0, dec->stream, (const uint16_t*)dec->cls.p, (uint32_t)n, plan,
                       (uint32_t*)dec->perm.p);
    /* irregular packets first (usually a handful of waves, or none), then the wave pairs of the regular ones */
    hipLaunchKernelGGL(alac_scan, dim3((uint32_t)max_waves(n, ppw)), dim3(kWave), 0, dec->stream, c, d_blob, blob_bytes,
                       d_offsets, sz, (const uint32_t*)dec->perm.p, (const Plan*)plan, d_out, (uint64_t)out_stride, d_frames,
                       d_status, (int32_t*)dec->scratch_u.p, (int32_t*)dec->scratch_g.p, ppw, (alac::ChanDesc*)dec->cd.p,
                       (alac::PktDesc*)dec->pd.p);
    if (dec->cfg.num_channels <= 2 && dec->cfg.kb != 0)
        hipLaunchKernelGGL(alac_decode, dim3((uint32_t)max_waves(n, ppw)), dim3(2 * kWave), 0, dec->stream, c, d_blob,
                           blob_bytes, d_offsets, sz, (const uint32_t*)dec->perm.p, (const Plan*)plan, d_out,
                           (uint64_t)out_stride, d_frames, d_status, (int32_t*)dec->scratch_u.p, ppw);
    HIP_TRY(hipGetLastError());
    if (dec->cfg.kb != 0) {
        /* irregular packets were only scanned by alac_scan (status, frames, channel descriptors) */
        const uint64_t rs = row_stride_of(dec->cfg.frame_length);
        const uint32_t bpp = (dec->cfg.frame_length + 255u) / 256u;
        if (dec->cfg.num_channels > 2) {
            /* split pipeline: one lane per channel, sorted by predictor order */
            Plan* plan2 = (Plan*)dec->plan2.p;
            const size_t n_slots = n * 8;
            const uint32_t nb2 = (uint32_t)((n_slots + 255) / 256);
            const uint32_t ppw2 = pick_ppw(n * dec->cfg.num_channels);
            HIP_TRY(hipMemsetAsync(plan2, 0, sizeof(Plan), dec->stream));
            hipLaunchKernelGGL(alac_task_classify, dim3(nb2), dim3(256), 0, dec->stream, c, (const alac::ChanDesc*)dec->cd.p,
                               (const alac::PktDesc*)dec->pd.p, (const uint16_t*)dec->cls.p, (uint32_t)n_slots,
                               (uint16_t*)dec->keys2.p, plan2);
            hipLaunchKernelGGL(alac_plan, dim3(1), dim3(kWave), 0, dec->stream, plan2, ppw2);
            hipLaunchKernelGGL(alac_scatter, dim3(nb2), dim3(256), 0, dec->stream, (const uint16_t*)dec->keys2.p,
                               (uint32_t)n_slots, plan2, (uint32_t*)dec->perm2.p);
            hipLaunchKernelGGL(alac_chan_decode, dim3((uint32_t)max_waves(n_slots, ppw2)), dim3(2 * kWave), 0, dec->stream, c,
                               d_blob, blob_bytes, d_offsets, sz, (const uint32_t*)dec->perm2.p, (const Plan*)plan2,
                               (const alac::ChanDesc*)dec->cd.p, (int32_t*)dec->rows.p, rs, ppw2);
        }
        /* PCM of the split packets (with one or two channels: of the escape-only packets) */
        const uint32_t ib = (uint32_t)std::min<uint64_t>((uint64_t)n * bpp, 8192);
        hipLaunchKernelGGL(alac_interleave, dim3(ib), dim3(256), 0, dec->stream, c, d_blob, blob_bytes, d_offsets, sz,
                           (const uint32_t*)dec->perm.p, (const Plan*)plan, (const alac::ChanDesc*)dec->cd.p,
                           (const alac::PktDesc*)dec->pd.p, (const int32_t*)dec->rows.p, rs, d_out, (uint64_t)out_stride, bpp);
        hipLaunchKernelGGL(alac_legacy, dim3((uint32_t)max_waves(n, ppw)), dim3(kWave), 0, dec->stream, c, d_blob, blob_bytes,
                           d_offsets, sz, (const uint32_t*)dec->perm.p, (const Plan*)plan, (const alac::PktDesc*)dec->pd.p, d_out,
                           (uint64_t)out_stride, d_frames, d_status, (int32_t*)dec->scratch_u.p, (int32_t*)dec->scratch_g.p, ppw);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(dec->ev_stop[slot], dec->stream));
    dec->launches++;
    return ALACGPU_E_OK;
}

bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); /* an ordinary (pageable) pointer: not an error */
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

} /* namespace */

extern "C" {

int alacgpu_create(const alacgpu_config* cfg, int device, alacgpu_decoder** out) {
    if (!cfg || !out) {
        set_err("null argument");
        return ALACGPU_E_ARG;
    }
    *out = nullptr;
    const int bps = bytes_per_sample(cfg->bit_depth);
    if (bps == 0) { /* decoder.go:91-93 */
        set_err("invalid configuration: alac: unsupported bit depth: %d", (int)cfg->bit_depth);
        return ALACGPU_E_CONFIG;
    }
    if (cfg->num_channels < 1 || cfg->num_channels > 8) {
        set_err("invalid configuration: NumChannels %d outside 1..8", (int)cfg->num_channels);
        return ALACGPU_E_CONFIG;
    }
    if (cfg->frame_length == 0 || cfg->frame_length > (1u << 24)) {
        set_err("invalid configuration: FrameLength %u", cfg->frame_length);
        return ALACGPU_E_CONFIG;
    }
    HIP_TRY(hipSetDevice(device));
    alacgpu_decoder* d = new (std::nothrow) alacgpu_decoder();
    if (!d) {
        set_err("out of memory");
        return ALACGPU_E_ARG;
    }
    d->cfg = *cfg;
    d->device = device;
    d->frame_bytes = (size_t)cfg->frame_length * cfg->num_channels * (size_t)bps;
    d->dev_cfg = alac::DevCfg{cfg->frame_length, cfg->bit_depth, cfg->num_channels, cfg->pb, cfg->mb, cfg->kb,
                              (uint32_t)bps, 0u};
    d->launches = 0;
    d->pool = nullptr;
    d->chunk_bytes = (size_t)192 << 20;
    if (const char* e = getenv("ALACGPU_CHUNK_MB")) {
        const long v = atol(e);
        if (v >= 1 && v <= 65536) d->chunk_bytes = (size_t)v << 20;
    }
    d->stream = d->s_in = d->s_out = nullptr;
    for (uint32_t i = 0; i < kTimingSlots; i++) d->ev_start[i] = d->ev_stop[i] = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->s_out, hipStreamNonBlocking);
    for (uint32_t i = 0; i < kTimingSlots && e == hipSuccess; i++) {
        e = hipEventCreate(&d->ev_start[i]);
        if (e == hipSuccess) e = hipEventCreate(&d->ev_stop[i]);
    }
    for (int k = 0; k < kSlots && e == hipSuccess; k++) {
        e = hipEventCreateWithFlags(&d->slots[k].ev_in, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d->slots[k].ev_k, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d->slots[k].ev_out, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        set_err("stream/event creation failed: %s", hipGetErrorString(e));
        alacgpu_destroy(d);
        return ALACGPU_E_HIP;
    }
    *out = d;
    return ALACGPU_E_OK;
}

void alacgpu_destroy(alacgpu_decoder* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->s_in) (void)hipStreamSynchronize(d->s_in);
    if (d->s_out) (void)hipStreamSynchronize(d->s_out);
    delete d->pool;
    DevBuf* bufs[] = {&d->scratch_u, &d->scratch_g, &d->plan, &d->cls, &d->perm, &d->sizes_ws, &d->cd, &d->pd,
                      &d->plan2, &d->keys2, &d->perm2, &d->rows};
    for (DevBuf* b : bufs) b->release();
    for (int k = 0; k < kSlots; k++) {
        Slot& s = d->slots[k];
        s.d_in.release();
        s.d_out.release();
        s.h_in.release();
        s.h_out.release();
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_k) (void)hipEventDestroy(s.ev_k);
        if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    }
    for (uint32_t i = 0; i < kTimingSlots; i++) {
        if (d->ev_start[i]) (void)hipEventDestroy(d->ev_start[i]);
        if (d->ev_stop[i]) (void)hipEventDestroy(d->ev_stop[i]);
    }
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (d->s_in) (void)hipStreamDestroy(d->s_in);
    if (d->s_out) (void)hipStreamDestroy(d->s_out);
    delete d;
}

int alacgpu_get_format(const alacgpu_decoder* d, alacgpu_format* fmt) {
    if (!d || !fmt) return ALACGPU_E_ARG;
    fmt->sample_rate = (int32_t)d->cfg.sample_rate; /* decoder.go:98-102 */
    fmt->bit_depth = d->cfg.bit_depth;
    fmt->channels = d->cfg.num_channels;
    return ALACGPU_E_OK;
}

size_t alacgpu_frame_bytes(const alacgpu_decoder* d) { return d ? d->frame_bytes : 0; }

int alacgpu_reserve(alacgpu_decoder* d, size_t n) {
    if (!d) return ALACGPU_E_ARG;
    HIP_TRY(hipSetDevice(d->device));
    return reserve_workspace(d, n, pick_ppw(n));
}

int alacgpu_decode_batch_device(alacgpu_decoder* d, const uint8_t* d_blob, size_t blob_bytes, const uint64_t* d_offsets,
                                const uint32_t* d_sizes, size_t n, uint8_t* d_out, size_t out_stride,
                                uint32_t* d_frames, int32_t* d_status, int sync) {
    if (!d || (n && (!d_offsets || !d_out || !d_frames || !d_status)) || (blob_bytes && !d_blob)) {
        set_err("null argument");
        return ALACGPU_E_ARG;
    }
    if (out_stride < d->frame_bytes) {
        set_err("out_stride %zu < frame bytes %zu", out_stride, d->frame_bytes);
        return ALACGPU_E_ARG;
    }
    HIP_TRY(hipSetDevice(d->device));
    int rc = launch(d, d_blob, (uint64_t)blob_bytes, d_offsets, d_sizes, n, d_out, out_stride, d_frames, d_status);
    if (rc) return rc;
    if (sync) HIP_TRY(hipStreamSynchronize(d->stream));
    return ALACGPU_E_OK;
}

/*
 * DecodePackets from host memory: the batch is cut into chunks of whole packets; chunk c is uploaded on one stream
 * while chunk c-1 decodes on the handle's stream and chunk c-2 comes back on a third. The packets go up exactly as
 * they lie in the caller's blob (no re-pack: the kernels read dense blobs), together with their offsets, in ONE
 * transfer; PCM, frame counts and status words come back in ONE. Pageable caller memory is staged through pinned
 * buffers by a few copy threads; memory the caller pinned itself (hipHostMalloc / hipHostRegister) is used in place.
 */
int alacgpu_decode_batch(alacgpu_decoder* d, const uint8_t* blob, const uint64_t* offsets, size_t n, uint8_t* out,
                         size_t out_stride, uint32_t* frames_out, int32_t* status) {
    if (!d || (n && (!offsets || !out || !frames_out || !status))) {
        set_err("null argument");
        return ALACGPU_E_ARG;
    }
    if (out_stride < d->frame_bytes) {
        set_err("out_stride %zu < frame bytes %zu", out_stride, d->frame_bytes);
        return ALACGPU_E_ARG;
    }
    if (n == 0) return ALACGPU_E_OK;
    for (size_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0x0fffffffull) {
            set_err("bad offsets at packet %zu", i);
            return ALACGPU_E_ARG;
        }
    }
    if (offsets[n] > offsets[0] && !blob) {
        set_err("null argument");
        return ALACGPU_E_ARG;
    }
    HIP_TRY(hipSetDevice(d->device));
    const size_t fb = d->frame_bytes;
    const size_t d_stride = (fb + 15u) & ~(size_t)15u; /* 16-byte aligned device rows: the LDS-staged wide stores */
    const bool out_pinned = is_pinned(out) && is_pinned(frames_out) && is_pinned(status);
    const bool in_pinned = (offsets[n] == offsets[0]) || is_pinned(blob);
    if (!d->pool && !(out_pinned && in_pinned)) {
        unsigned t = std::thread::hardware_concurrency();
        t = t > 16 ? 8 : (t > 2 ? t / 2 : 1);
        if (const char* e = getenv("ALACGPU_COPY_THREADS")) t = (unsigned)std::max(1, atoi(e));
        d->pool = new (std::nothrow) CopyPool(t - 1); /* the calling thread copies too */
    }

    auto finish = [&](Slot& s) -> int { /* chunk is back in pinned memory (or in place): hand it to the caller */
        HIP_TRY(hipEventSynchronize(s.ev_out));
        if (!out_pinned) {
            const uint8_t* h = (const uint8_t*)s.h_out.p;
            const uint8_t* hm = h + s.n * d_stride;
            memcpy(frames_out + s.first, hm, s.n * sizeof(uint32_t));
            memcpy(status + s.first, hm + s.n * sizeof(uint32_t), s.n * sizeof(int32_t));
            const size_t pieces = std::min<size_t>(s.n, 64);
            auto body = [&](size_t k) {
                const size_t lo = s.n * k / pieces, hi = s.n * (k + 1) / pieces;
                if (out_stride == d_stride) {
                    memcpy(out + (s.first + lo) * out_stride, h + lo * d_stride, (hi - lo) * d_stride - (d_stride - fb));
                } else {
                    for (size_t i = lo; i < hi; i++) memcpy(out + (s.first + i) * out_stride, h + i * d_stride, fb);
                }
            };
            if (d->pool) d->pool->parallel_for(pieces, body);
            else for (size_t k = 0; k < pieces; k++) body(k);
        }
        s.busy = false;
        return ALACGPU_E_OK;
    };

    size_t first = 0;
    int turn = 0;
    int rc = ALACGPU_E_OK;
    while (first < n && rc == ALACGPU_E_OK) {
        /* packets of this chunk: whole packets, about chunk_bytes of traffic (packet bytes in + PCM out) */
        size_t cnt = 0;
        uint64_t bytes = 0;
        while (first + cnt < n && (cnt == 0 || bytes < d->chunk_bytes)) {
            bytes += (offsets[first + cnt + 1] - offsets[first + cnt]) + fb;
            cnt++;
        }
        Slot& s = d->slots[turn];
        turn = (turn + 1) % kSlots;
        if (s.busy && (rc = finish(s))) break;
        s.first = first;
        s.n = cnt;
        const uint64_t b0 = offsets[first], b1 = offsets[first + cnt];
        const size_t in_bytes = (size_t)(b1 - b0);
        const size_t meta = (cnt + 1) * sizeof(uint64_t);
        const size_t meta_pad = (meta + 255u) & ~(size_t)255u; /* packet bytes start 256-byte aligned on the device */
        if ((rc = s.d_in.ensure(meta_pad + in_bytes + 16))) break;
        if ((rc = s.d_out.ensure(cnt * d_stride + cnt * 8 + 16))) break;
        if ((rc = s.h_in.ensure(in_pinned ? meta_pad : meta_pad + in_bytes))) break;
        if (!out_pinned && (rc = s.h_out.ensure(cnt * d_stride + cnt * 8))) break;
        /* upload: offsets rebased to the chunk's first byte, then the bytes */
        uint64_t* h_off = (uint64_t*)s.h_in.p;
        for (size_t i = 0; i <= cnt; i++) h_off[i] = offsets[first + i] - b0;
        uint8_t* d_in = (uint8_t*)s.d_in.p;
        if (in_pinned) {
            HIP_TRY(hipMemcpyAsync(d_in, h_off, meta, hipMemcpyHostToDevice, d->s_in));
            if (in_bytes) HIP_TRY(hipMemcpyAsync(d_in + meta_pad, blob + b0, in_bytes, hipMemcpyHostToDevice, d->s_in));
        } else {
            uint8_t* hb = (uint8_t*)s.h_in.p + meta_pad;
            const size_t pieces = std::max<size_t>(1, std::min<size_t>(64, in_bytes >> 20));
            auto body = [&](size_t k) {
                const size_t lo = in_bytes * k / pieces, hi = in_bytes * (k + 1) / pieces;
                memcpy(hb + lo, blob + b0 + lo, hi - lo);
            };
            if (d->pool) d->pool->parallel_for(pieces, body);
            else for (size_t k = 0; k < pieces; k++) body(k);
            HIP_TRY(hipMemcpyAsync(d_in, s.h_in.p, meta_pad + in_bytes, hipMemcpyHostToDevice, d->s_in));
        }
        HIP_TRY(hipEventRecord(s.ev_in, d->s_in));
        /* decode */
        HIP_TRY(hipStreamWaitEvent(d->stream, s.ev_in, 0));
        uint8_t* d_pcm = (uint8_t*)s.d_out.p;
        uint32_t* d_fr = (uint32_t*)(d_pcm + cnt * d_stride);
        int32_t* d_st = (int32_t*)(d_fr + cnt);
        if ((rc = launch(d, d_in + meta_pad, in_bytes, (const uint64_t*)d_in, nullptr, cnt, d_pcm, d_stride, d_fr, d_st))) break;
        HIP_TRY(hipEventRecord(s.ev_k, d->stream));
        /* download */
        HIP_TRY(hipStreamWaitEvent(d->s_out, s.ev_k, 0));
        if (out_pinned) {
            HIP_TRY(hipMemcpy2DAsync(out + first * out_stride, out_stride, d_pcm, d_stride, fb, cnt, hipMemcpyDeviceToHost, d->s_out));
            HIP_TRY(hipMemcpyAsync(frames_out + first, d_fr, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, d->s_out));
            HIP_TRY(hipMemcpyAsync(status + first, d_st, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, d->s_out));
        } else {
            HIP_TRY(hipMemcpyAsync(s.h_out.p, d_pcm, cnt * d_stride + cnt * 8, hipMemcpyDeviceToHost, d->s_out));
        }
        HIP_TRY(hipEventRecord(s.ev_out, d->s_out));
        s.busy = true;
        first += cnt;
    }
    /* drain, oldest first */
    for (int k = 0; k < kSlots; k++) {
        Slot& s = d->slots[(turn + k) % kSlots];
        if (!s.busy) continue;
        const int r2 = finish(s);
        if (rc == ALACGPU_E_OK) rc = r2;
        s.busy = false;
    }
    return rc;
}

int alacgpu_decode_packet(alacgpu_decoder* d, const uint8_t* packet, size_t packet_len, uint8_t* out, size_t out_cap,
                          size_t* out_len, int32_t* status_out) {
    if (!d || (!packet && packet_len) || !out || !out_len) {
        set_err("null argument");
        return ALACGPU_E_ARG;
    }
    if (out_cap < d->frame_bytes) { /* decodePacketInto needs a full frame, decoder.go:131-132 */
        set_err("output capacity %zu < frame bytes %zu", out_cap, d->frame_bytes);
        return ALACGPU_E_ARG;
    }
    *out_len = 0;
    const uint64_t offs[2] = {0, packet_len};
    const uint8_t dummy = 0;
    uint32_t frames = 0;
    int32_t st = 0;
    /* a batch of one through the same entry: one upload (offsets + bytes), the kernels, one download */
    int rc = alacgpu_decode_batch(d, packet_len ? packet : &dummy, offs, 1, out, d->frame_bytes, &frames, &st);
    if (rc) return rc;
    if (status_out) *status_out = st;
    if (st != 0) {
        set_err("decode failed: status 0x%x", st);
        return ALACGPU_E_DECODE;
    }
    *out_len = (size_t)frames * d->cfg.num_channels * (size_t)d->dev_cfg.bps; /* decoder.go:206 */
    return ALACGPU_E_OK;
}

int alacgpu_timing_reset(alacgpu_decoder* d) {
    if (!d) return ALACGPU_E_ARG;
    d->launches = 0;
    return ALACGPU_E_OK;
}

int alacgpu_kernel_times(alacgpu_decoder* d, float* ms, size_t max_n, size_t* n_out) {
    if (!d || !ms || !n_out) return ALACGPU_E_ARG;
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    size_t avail = (size_t)std::min<uint64_t>(d->launches, kTimingSlots);
    size_t n = std::min(avail, max_n);
    /* the n most recent launches, oldest first */
    for (size_t i = 0; i < n; i++) {
        const uint32_t slot = (uint32_t)((d->launches - n + i) % kTimingSlots);
        HIP_TRY(hipEventElapsedTime(&ms[i], d->ev_start[slot], d->ev_stop[slot]));
    }
    *n_out = n;
    return ALACGPU_E_OK;
}

int alacgpu_last_kernel_ms(alacgpu_decoder* d, float* ms) {
    size_t got = 0;
    int rc = alacgpu_kernel_times(d, ms, 1, &got);
    if (rc) return rc;
    if (got == 0) {
        set_err("no kernel has been launched on this handle since the last timing reset");
        return ALACGPU_E_ARG;
    }
    return ALACGPU_E_OK;
}

void* alacgpu_stream(alacgpu_decoder* d) { return d ? (void*)d->stream : nullptr; }

int alacgpu_synchronize(alacgpu_decoder* d) {
    if (!d) return ALACGPU_E_ARG;
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return ALACGPU_E_OK;
}

const char* alacgpu_last_error(void) { return g_err; }

#ifdef ALAC_DUO_PROF
/* profiling build only: read and clear the stamp sums ([0..3] role A, [8..11] role B) */
int alacgpu_debug_prof(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_duo_prof), sizeof(g_duo_prof)) != hipSuccess) return ALACGPU_E_HIP;
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_duo_prof), z, sizeof(z)) != hipSuccess) return ALACGPU_E_HIP;
    return ALACGPU_E_OK;
}
#endif

const char* alacgpu_version(void) { return "alacgpu 0.3.0 gfx950"; }

} /* extern "C" */
