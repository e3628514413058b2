/*
 * k_sort.hip — the sort pre-pass: packet classification + descriptor checks, launch plan, counting-sort scatter, channel-task keys (one translation unit of libalacgpu.so, see alac_gpu.h).
 */
#include "alac_gpu.h"

namespace alack {

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
            /* not regular: scan first (configurations the lean Golomb step covers, alac_regular.h: lean_config). More than two channels: split pipeline. One or two: escape
             * elements are unpacked by alac_interleave, anything else is handed to the whole-packet decoder. */
            if (key == alac::KEY_IRREGULAR) key = alac::lean_config(cfg) ? kKeyScan : kKeyLegacy;
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
    uint32_t p = 0, w = 0, z = 0, wi = 0, ww = 0;
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
        ww += (key >= alac::KEY_WIDE && key < alac::KEY_IRREGULAR) ? cw : 0u;
    }
    /* inclusive scan over the 64 lanes, then make it exclusive */
    uint32_t ip = p, iw = w, iz = z, ii = wi, ix = ww;
#pragma unroll
    for (int o = 1; o < (int)kWave; o <<= 1) {
        const uint32_t tp = (uint32_t)__shfl_up((int)ip, o, kWave), tw = (uint32_t)__shfl_up((int)iw, o, kWave);
        const uint32_t tz = (uint32_t)__shfl_up((int)iz, o, kWave), ti = (uint32_t)__shfl_up((int)ii, o, kWave);
        const uint32_t tx = (uint32_t)__shfl_up((int)ix, o, kWave);
        if ((int)threadIdx.x >= o) {
            ix += tx;
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
        plan->wide_waves = ix;
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

/* Which compute units does this device have? Every workgroup marks the one it runs on (same numbering as the pair
 * kernels' gate: XCC_ID << 6 | SE << 4 | CU of HW_ID). 64 KB of LDS each = two workgroups per CU, and the grid is two
 * per CU, so the dispatcher has to use every CU; a workgroup stays until all have arrived (or, on a busy device, for a
 * bounded time: a CU that is missed only loses its fixed place in the pair kernels' item order). Run once per handle. */
__global__ void __launch_bounds__(kWave) alac_cu_census(uint32_t* __restrict__ seen, uint32_t* __restrict__ arrived, uint32_t expect) {
    __shared__ uint32_t pad[16384];
    pad[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 0) {
        uint32_t id, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        seen[((xcc & 7u) << 6) | (((id >> 13) & 3u) << 4) | ((id >> 8) & 15u)] = 1u;
        atomicAdd(arrived, 1u);
        for (int i = 0; i < 4000 && atomicAdd(arrived, 0u) < expect; ++i) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
    if (pad[(threadIdx.x * 5u) & 63u] == 0xffffffffu) seen[0] = 2u; /* keeps the LDS block allocated */
}

} /* namespace alack */
