/*
 * k_scan.hip — irregular packets: scan (status, frame count, channel descriptors, residual rows) and the whole-packet decoder (one translation unit of libalacgpu.so, see alac_gpu.h).
 */
#include "alac_gpu.h"

namespace alack {

/* Irregular packets (keys >= KEY_IRREGULAR; they own the first plan->irr_waves wave slots): one wavefront per 64 packets.
 * With a usable KB they are scanned (status, frame count, channel descriptors: split pipeline step 1, PCM comes
 * from the later kernels); with KB == 0 the whole-packet decoder takes them.
 * A BOUNDED grid (round 4): the host does not know how many of a batch's wave slots are irregular, and until round 3 it
 * launched one 201-register, 26-KB workgroup per wave slot of the whole batch, nearly all of which left at once — but
 * first stood in the dispatcher's line with the later rounds of the decode kernels (which is what kept these kernels off
 * the side stream for large batches). Now the grid is a few workgroups per CU (alacgpu.hip: scan_grid) and workgroup g
 * walks slots g, g + gridDim.x, ... below plan->irr_waves: an exit condition every wave reaches. */
static __device__ __forceinline__ void scan_slot(uint32_t b, const alac::DevCfg& cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes,
                                                 const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ sizes,
                                                 const uint32_t* __restrict__ perm, const Plan* __restrict__ plan, uint8_t* __restrict__ out,
                                                 uint64_t out_stride, uint32_t* __restrict__ frames_out, int32_t* __restrict__ status,
                                                 int32_t* __restrict__ scratch_u, int32_t* __restrict__ scratch_g, uint32_t ppw,
                                                 alac::ChanDesc* __restrict__ cd, alac::PktDesc* __restrict__ pd, int32_t* __restrict__ rows,
                                                 uint64_t row_stride);

__global__ void __launch_bounds__(kWave)
alac_scan(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
          const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
          uint8_t* __restrict__ out, uint64_t out_stride, uint32_t* __restrict__ frames_out,
          int32_t* __restrict__ status, int32_t* __restrict__ scratch_u, int32_t* __restrict__ scratch_g,
          uint32_t ppw, alac::ChanDesc* __restrict__ cd, alac::PktDesc* __restrict__ pd, int32_t* __restrict__ rows,
          uint64_t row_stride) {
    const uint32_t limit = plan->irr_waves;
    for (uint32_t b = blockIdx.x; b < limit; b += gridDim.x) {
        scan_slot(b, cfg, blob, blob_bytes, offsets, sizes, perm, plan, out, out_stride, frames_out, status, scratch_u, scratch_g, ppw,
                  cd, pd, rows, row_stride);
        __builtin_amdgcn_wave_barrier(); /* (one wave per workgroup: the LDS rows are the next slot's from here on) */
    }
}

static __device__ __forceinline__ void scan_slot(uint32_t b, const alac::DevCfg& cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes,
                                                 const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ sizes,
                                                 const uint32_t* __restrict__ perm, const Plan* __restrict__ plan, uint8_t* __restrict__ out,
                                                 uint64_t out_stride, uint32_t* __restrict__ frames_out, int32_t* __restrict__ status,
                                                 int32_t* __restrict__ scratch_u, int32_t* __restrict__ scratch_g, uint32_t ppw,
                                                 alac::ChanDesc* __restrict__ cd, alac::PktDesc* __restrict__ pd, int32_t* __restrict__ rows,
                                                 uint64_t row_stride) {
    uint32_t e = 0;
    for (uint32_t t = 1; t < plan->nk; ++t)
        if (plan->list_wave0[t] <= b) e = t;
    const uint32_t key = plan->list_key[e];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = (b - plan->list_wave0[e]) * ppw + lane;
    const bool live = lane < ppw && idx < plan->count[key];
    const uint32_t pkt = live ? perm[plan->pkt_start[key] + idx] : 0u;

    GpuWave wv;
    wv.u_tile = scratch_u + (size_t)b * u_tile_cells(cfg.frame_length);
    wv.g_tile = scratch_g + (size_t)b * kFallbackSlots * ppw + lane;
    wv.ppw = ppw;
    wv.my_out = nullptr;
    wv.lane = lane;
    wv.wcnt = wv.flushed = wv.wpos = wv.fpos = 0;

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
        st = alac::decode_wave<GpuWave, 16, true, true>(wv, cfg, live, p, size, avail, o, &frames, cd + (size_t)pkt * 8u, pd + pkt,
                                                        rows ? rows + (size_t)pkt * cfg.num_channels * row_stride : nullptr,
                                                        (size_t)row_stride);
    else
        st = alac::decode_wave<GpuWave, 16, true>(wv, cfg, live, p, size, avail, o, &frames);
    if (live) {
        frames_out[pkt] = frames;
        status[pkt] = st;
    }
}

/* packets the scan routed to the whole-packet decoder (orders 17..30): the scanned slots again, the same bounded grid */
static __device__ __forceinline__ void legacy_slot(uint32_t b, const alac::DevCfg& cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes,
                                                   const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ sizes,
                                                   const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                                                   const alac::PktDesc* __restrict__ pd, uint8_t* __restrict__ out, uint64_t out_stride,
                                                   uint32_t* __restrict__ frames_out, int32_t* __restrict__ status,
                                                   int32_t* __restrict__ scratch_u, int32_t* __restrict__ scratch_g, uint32_t ppw);

__global__ void __launch_bounds__(kWave)
alac_legacy(alac::DevCfg cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offsets,
            const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
            const alac::PktDesc* __restrict__ pd, uint8_t* __restrict__ out, uint64_t out_stride,
            uint32_t* __restrict__ frames_out, int32_t* __restrict__ status, int32_t* __restrict__ scratch_u,
            int32_t* __restrict__ scratch_g, uint32_t ppw) {
    const uint32_t limit = plan->irr_waves; /* kKeyScan slots are irregular ones: the first of the plan */
    for (uint32_t b = blockIdx.x; b < limit; b += gridDim.x) {
        legacy_slot(b, cfg, blob, blob_bytes, offsets, sizes, perm, plan, pd, out, out_stride, frames_out, status, scratch_u, scratch_g, ppw);
        __builtin_amdgcn_wave_barrier();
    }
}

static __device__ __forceinline__ void legacy_slot(uint32_t b, const alac::DevCfg& cfg, const uint8_t* __restrict__ blob, uint64_t blob_bytes,
                                                   const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ sizes,
                                                   const uint32_t* __restrict__ perm, const Plan* __restrict__ plan,
                                                   const alac::PktDesc* __restrict__ pd, uint8_t* __restrict__ out, uint64_t out_stride,
                                                   uint32_t* __restrict__ frames_out, int32_t* __restrict__ status,
                                                   int32_t* __restrict__ scratch_u, int32_t* __restrict__ scratch_g, uint32_t ppw) {
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
    wv.u_tile = scratch_u + (size_t)b * u_tile_cells(cfg.frame_length);
    wv.g_tile = scratch_g + (size_t)b * kFallbackSlots * ppw + lane;
    wv.ppw = ppw;
    wv.my_out = nullptr;
    wv.lane = lane;
    wv.wcnt = wv.flushed = wv.wpos = wv.fpos = 0;
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

} /* namespace alack */
