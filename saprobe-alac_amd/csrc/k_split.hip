/*
 * k_split.hip — split pipeline behind the scan: predictor pass over the stored residuals, interleave (one translation unit of libalacgpu.so, see alac_gpu.h).
 */
/* alac_regular.h: predict_narrow_core; measured per kernel: the predictor pass (three lone-ish waves per SIMD, residuals from
 * memory) runs 2 % faster with the all-asm tap order (8-channel 24-bit 16 384 packets: 6.87 against 7.02 ms) */
#define ALAC_TAP_ORDER 1
#include "alac_gpu.h"

namespace alack {

/* Split pipeline, predictor pass: one wavefront per 64 channel tasks with the same key; the residuals are in the
 * task's row (left there by alac_scan), the samples replace them. */
__global__ void __launch_bounds__(kWave, 3)
alac_chan_predict(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                  const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                  const alac::ChanDesc* __restrict__ cd, int32_t* __restrict__ rows, uint64_t row_stride, uint32_t ppw) {
    const uint32_t b = blockIdx.x;
    if (b >= plan->total_waves) return;
    uint32_t e = 0;
    for (uint32_t t = 1; t < plan->nk; ++t)
        if (plan->list_wave0[t] <= b) e = t;
    const uint32_t key = plan->list_key[e];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = (b - plan->list_wave0[e]) * ppw + lane;
    const bool live = lane < ppw && idx < plan->count[key];
    const uint32_t t = live ? perm[plan->pkt_start[key] + idx] : 0u;
    const uint32_t pkt = t >> 3, slot = t & 7u;

    GpuWaveMem wv;
    wv.u_tile = nullptr;
    wv.g_tile = nullptr;
    wv.ppw = ppw;
    wv.my_out = nullptr;
    wv.lane = lane;
    wv.wcnt = wv.flushed = wv.wpos = wv.fpos = 0;
    wv.it = 0;
    wv.chunk0 = 0;

    const uint64_t off = live ? offsets[pkt] : 0ull;
    const uint8_t* p = blob + off;
    const uint32_t size = live ? sizes[pkt] : 0u;
    const uint32_t avail = avail_of(blob_bytes, off);
    alac::ChanDesc d = cd[t];
    if (!live) d.hdr_pos = d.ent_pos = d.ns = 0;
    /* lanes without a task read and write row 0 of packet 0's slot... no: they get the wave's first live row and
     * neither read anything they use nor write (ns = 0) */
    int32_t* row = rows + ((size_t)pkt * cfg.num_channels + slot) * row_stride;
    wv.res = row;
    const uint32_t ukey = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    alac::decode_channel_task<GpuWaveMem, alac::ROLE_B>(wv, cfg, ukey, live, p, size, avail, d, row);
}

/* the slice buffer changes hands between the threads of a block: LDS traffic only. __syncthreads() would also wait for
 * the global stores of the slice just written, and the loads of the next slice could not start before they are done */
static __device__ __forceinline__ void il_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* LDS slice -> PCM slot: 16-byte pieces, whole lines at a time */
static __device__ __forceinline__ void il_store(uint8_t* dst, const uint8_t* slice, uint32_t total, bool aligned16) {
    if (aligned16) {
        for (uint32_t k = threadIdx.x * 16u; k + 16u <= total; k += blockDim.x * 16u)
            *reinterpret_cast<uint4*>(dst + k) = *reinterpret_cast<const uint4*>(slice + k);
        for (uint32_t k = (total & ~15u) + threadIdx.x; k < total; k += blockDim.x) dst[k] = slice[k];
    } else {
        for (uint32_t k = threadIdx.x; k < total; k += blockDim.x) dst[k] = slice[k];
    }
}

/* slices [s0, s_end) of one packet whose frames are whole dwords (compile-time byte positions). What slice s + 1 needs
 * from memory is asked for BEFORE slice s is stored: the memory counter of a wave retires in issue order, stores
 * included, so loads issued behind the stores would only be usable once those stores have reached memory. */
template <int NC, int BPS>
static __device__ __forceinline__ void il_chunk(const alac::DevCfg& cfg, const uint8_t* pk, uint32_t psz, uint32_t pav,
                                                const alac::PktDesc& q, const alac::ChanDesc* pcd, const int32_t* prow,
                                                size_t row_stride, uint8_t* dst_pkt, uint32_t s0, uint32_t s_end, uint8_t* slice) {
    constexpr uint32_t DW = NC * BPS / 4;
    alac::IlLoaded<NC> L;
    uint32_t f0 = s0 * blockDim.x;
    if (f0 + threadIdx.x < q.frames) alac::interleave_load<NC, BPS>(cfg, pk, psz, pav, q, pcd, prow, row_stride, f0 + threadIdx.x, L);
    for (uint32_t sl = s0; sl < s_end && f0 < q.frames; ++sl) {
        const uint32_t nf = min(q.frames - f0, (uint32_t)blockDim.x);
        if (threadIdx.x < nf) {
            uint32_t fr[DW];
            alac::interleave_build<NC, BPS>(cfg, q, pcd, f0 + threadIdx.x, L, fr);
            uint32_t* dw = reinterpret_cast<uint32_t*>(slice) + threadIdx.x * DW;
#pragma unroll
            for (uint32_t k = 0; k < DW; ++k) dw[k] = fr[k];
        }
        il_sync();
        const uint32_t fn = f0 + blockDim.x;
        if (sl + 1u < s_end && fn + threadIdx.x < q.frames)
            alac::interleave_load<NC, BPS>(cfg, pk, psz, pav, q, pcd, prow, row_stride, fn + threadIdx.x, L);
        il_store(dst_pkt + (size_t)f0 * (NC * BPS), slice, nf * (NC * BPS), cfg.aligned16 != 0); /* f0 is a multiple of 64: 16-byte aligned */
        il_sync();
        f0 = fn;
    }
}

/* The same with FOUR frames per lane (alac_split.h: interleave_load4; round 4): sub-slices of 4 x blockDim.x frames, a lane's four
 * frames side by side in the slice buffer (4 x blockDim.x x NC x BPS bytes of LDS). The kernel is bound by the number of its
 * memory requests: a row's samples come as 16-byte loads (1 KB per wave instead of 256 bytes), the shift values of up to four
 * frames in one window. BASELINE config d: alac_interleave 1.30 -> 1.23 ms (6.63 -> 6.49 ms in all), at 65 536 packets
 * 17.8 -> 17.1; 165 registers (three waves per SIMD) against 72: it is what the wider requests buy, not more. */
template <int NC, int BPS>
static __device__ __forceinline__ void il_chunk4(const alac::DevCfg& cfg, const uint8_t* pk, uint32_t psz, uint32_t pav,
                                                 const alac::PktDesc& q, const alac::ChanDesc* pcd, const int32_t* prow,
                                                 size_t row_stride, uint8_t* dst_pkt, uint32_t f_begin, uint32_t f_end, uint8_t* slice) {
    constexpr uint32_t DW = NC * BPS / 4;
    const uint32_t span = 4u * blockDim.x;
    alac::IlLoaded<NC> L[4];
    uint32_t f0 = f_begin;
    if (f0 + 4u * threadIdx.x < q.frames) alac::interleave_load4<NC, BPS>(cfg, pk, psz, pav, q, pcd, prow, row_stride, f0 + 4u * threadIdx.x, L);
    for (; f0 < f_end && f0 < q.frames; f0 += span) {
        const uint32_t nf = min(q.frames - f0, span);
        const uint32_t mine = f0 + 4u * threadIdx.x;
        if (mine < q.frames) {
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                if (mine + j >= q.frames) break;
                uint32_t fr[DW];
                alac::interleave_build<NC, BPS>(cfg, q, pcd, mine + j, L[j], fr);
                uint32_t* dw = reinterpret_cast<uint32_t*>(slice) + (4u * threadIdx.x + j) * DW;
#pragma unroll
                for (uint32_t k = 0; k < DW; ++k) dw[k] = fr[k];
            }
        }
        il_sync();
        const uint32_t fn = f0 + span;
        if (fn < f_end && fn + 4u * threadIdx.x < q.frames)
            alac::interleave_load4<NC, BPS>(cfg, pk, psz, pav, q, pcd, prow, row_stride, fn + 4u * threadIdx.x, L);
        il_store(dst_pkt + (size_t)f0 * (NC * BPS), slice, nf * (NC * BPS), cfg.aligned16 != 0); /* f0 is a multiple of 256: 16-byte aligned */
        il_sync();
    }
}

/* one thread per (packet, frame) of the split packets: PCM in frame order. Blocks stride over the scanned
 * packets (the tail of the permutation that belongs to kKeyScan) x chunks of eight blockDim.x-frame slices. A slice is
 * assembled in LDS (a frame is 1..32 bytes at a byte offset of its own) and copied out as 16-byte pieces, whole lines
 * at a time. */
template <bool FOUR>
static __device__ __forceinline__ void interleave_body(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
                const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                const alac::ChanDesc* __restrict__ cd, const alac::PktDesc* __restrict__ pd, const int32_t* __restrict__ rows,
                uint64_t row_stride, uint8_t* __restrict__ out, uint64_t out_stride, uint32_t blocks_per_pkt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_slice[]; /* blockDim.x frames of <= 32 bytes (launch: 32 * blockDim.x), four times that with `four` */
    const uint32_t n_scan = plan->count[kKeyScan];
    const uint32_t first = plan->pkt_start[kKeyScan];
    const uint32_t fb = cfg.num_channels * cfg.bps;
    /* a block takes kSlices consecutive slices of one packet at a time: the packet's descriptors (a chain of dependent
     * scalar loads: permutation -> descriptor -> offsets) are fetched once for them */
    constexpr uint32_t kSlices = 8;
    const uint32_t chunks_per_pkt = (blocks_per_pkt + kSlices - 1u) / kSlices;
    const uint64_t chunks = (uint64_t)n_scan * chunks_per_pkt;
    for (uint64_t ck = blockIdx.x; ck < chunks; ck += gridDim.x) {
        const uint32_t pkt = perm[first + (uint32_t)(ck / chunks_per_pkt)];
        const alac::PktDesc& q = pd[pkt]; /* read in place: a copy indexed by slot would live in scratch memory */
        if (q.status != 0 || q.route != alac::ROUTE_SPLIT) continue; /* block-uniform */
        const uint32_t s0 = (uint32_t)(ck % chunks_per_pkt) * kSlices;
        const uint8_t* pk = blob + offsets[pkt];
        const uint32_t psz = sizes[pkt], pav = avail_of(blob_bytes, offsets[pkt]);
        const alac::ChanDesc* pcd = cd + (size_t)pkt * 8u;
        const int32_t* prow = rows + (size_t)pkt * cfg.num_channels * row_stride;
        uint8_t* dst_pkt = out + (size_t)pkt * out_stride;
        const uint32_t s_end = min(s0 + kSlices, blocks_per_pkt);
#define ALAC_IL_CASE(NC_, BPS_)                                                                              \
    case (NC_) * 8 + (BPS_):                                                                                 \
        if constexpr (FOUR)                                                                                  \
            il_chunk4<NC_, BPS_>(cfg, pk, psz, pav, q, pcd, prow, (size_t)row_stride, dst_pkt, s0 * blockDim.x, s_end * blockDim.x, s_slice); \
        else                                                                                                 \
            il_chunk<NC_, BPS_>(cfg, pk, psz, pav, q, pcd, prow, (size_t)row_stride, dst_pkt, s0, s_end, s_slice); \
        break;
        switch (cfg.num_channels * 8u + cfg.bps) {
            ALAC_IL_CASE(4, 2) ALAC_IL_CASE(6, 2) ALAC_IL_CASE(8, 2) ALAC_IL_CASE(4, 3) ALAC_IL_CASE(8, 3)
            ALAC_IL_CASE(3, 4) ALAC_IL_CASE(4, 4) ALAC_IL_CASE(5, 4) ALAC_IL_CASE(6, 4) ALAC_IL_CASE(7, 4) ALAC_IL_CASE(8, 4)
            default: /* frames that are not whole dwords: byte by byte into the slice */
                for (uint32_t sl = s0; sl < s_end; ++sl) {
                    const uint32_t f0 = sl * blockDim.x;
                    if (f0 >= q.frames) break;
                    const uint32_t nf = min(q.frames - f0, (uint32_t)blockDim.x);
                    if (threadIdx.x < nf)
                        alac::interleave_frame(cfg, pk, psz, pav, q, pcd, prow, (size_t)row_stride, f0 + threadIdx.x, s_slice + threadIdx.x * fb);
                    il_sync();
                    il_store(dst_pkt + (size_t)f0 * fb, s_slice, nf * fb, cfg.aligned16 != 0);
                    il_sync();
                }
        }
#undef ALAC_IL_CASE
    }
}


#define ALAC_IL_ARGS                                                                                                                   \
    alac::DevCfg cfg, const uint8_t *__restrict__ blob, uint64_t blob_bytes, const uint64_t *__restrict__ offsets,                    \
        const uint32_t *__restrict__ sizes, const uint32_t *__restrict__ perm, const Plan *__restrict__ plan,                         \
        const alac::ChanDesc *__restrict__ cd, const alac::PktDesc *__restrict__ pd, const int32_t *__restrict__ rows, uint64_t row_stride, \
        uint8_t *__restrict__ out, uint64_t out_stride, uint32_t blocks_per_pkt
/* one frame per lane (streams of one or two channels: escape elements only, no rows; and ALACGPU_IL4=0) */
__global__ void __launch_bounds__(256) alac_interleave(ALAC_IL_ARGS) {
    interleave_body<false>(cfg, blob, blob_bytes, offsets, sizes, perm, plan, cd, pd, rows, row_stride, out, out_stride, blocks_per_pkt);
}
/* four frames per lane (more than two channels: the rows exist); a kernel of its own so that each form has its own registers */
__global__ void __launch_bounds__(256) alac_interleave4(ALAC_IL_ARGS) {
    interleave_body<true>(cfg, blob, blob_bytes, offsets, sizes, perm, plan, cd, pd, rows, row_stride, out, out_stride, blocks_per_pkt);
}
#undef ALAC_IL_ARGS

} /* namespace alack */
