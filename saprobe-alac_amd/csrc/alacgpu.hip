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
 *   alac_scan      irregular packets, one wavefront per 64: status, frame count, where each channel starts; with
 *                  more than two channels also every channel's residuals, into the channel's row
 *   alac_decode_{16,24,32}q   regular packets: entropy, predictor, writer and spare wave per 64 same-key packets
 *                  (alac_duo.h), residuals and samples through an LDS queue; PCM staged in LDS, written in 128-B lines
 *   alac_decode_16g, alac_decode_w{24,32}   the gated twin (16-bit batches between the rounds) and the wide keys: a PAIR
 *                  of wavefronts (entropy; predictor + PCM) per 64 same-key packets (one kernel and compilation unit per
 *                  class: sample width x chanBits)
 *   alac_task_classify / alac_plan / alac_scatter / alac_chan_predict   (> 2 channels) one wavefront per 64
 *                  (packet, channel) tasks of the same order: the predictor over the stored residuals, in place
 *   alac_interleave, alac_legacy   PCM of the scanned packets (frame order), whole-packet decoder for the rest
 * The kernels live in k_sort.hip, k_scan.hip, k_dec*.hip and k_split.hip (alac_gpu.h).
 * HBM traffic per packet: compressed bytes in, PCM bytes out, plus the U-channel hand-off tile of stereo pairs
 * ((frame_length + 1) x 64 x int32 per workgroup, row-coalesced, written once and read once) or the sample rows
 * of the split pipeline.
 */
#include "alac_gpu.h"

using namespace alack;

namespace {

thread_local char g_err[512] = "";
#ifdef ALAC_DUO_PROF
Plan* g_prof_plan = nullptr; /* profiling build: the plan of the last decode (alacgpu_debug_prof) */
#endif

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
    bool bad = false; /* the chunk holds descriptors that leave the blob (ALACGPU_ERR_RANGE, set on the way back) */
};

} /* namespace */

struct alacgpu_decoder {
    alacgpu_config cfg;
    alac::DevCfg dev_cfg;
    int device;
    size_t frame_bytes;
    hipStream_t stream;                                       /* kernels */
    hipStream_t s_in, s_out;                                  /* host entry: uploads, downloads */
    hipStream_t s_side;                                       /* launch(): the irregular packets' kernels beside the regular ones' */
    hipEvent_t ev_fork, ev_join;                              /* s_side leaves `stream` behind the sort and is back before the stop event */
    hipEvent_t ev_start[kTimingSlots], ev_stop[kTimingSlots]; /* ring of per-launch event pairs */
    uint64_t launches;                                       /* since the last timing reset */
    DevBuf scratch_u, scratch_g, plan, cls, perm, sizes_ws;  /* kernel workspace */
    DevBuf cd, pd, plan2, keys2, perm2, rows;                /* split pipeline (more than two channels) */
    Slot slots[kSlots];                                      /* host-entry staging */
    CopyPool* pool;
    uint32_t il_threads;                                     /* alac_interleave block size (64, 128 or 256) */
    uint32_t il_four;                                        /* alac_interleave: four frames per lane (ALACGPU_IL4) */
    uint32_t n_cu;                                           /* compute units of the device */
    DevBuf cu_number;                                        /* PairArgs::cu_number */
    size_t chunk_bytes;                                      /* host entry: target bytes (in + out) per chunk */
    uint32_t lanes_min;                                      /* PairArgs::lanes_min; above 16: no second predictor wave for any key */
    int side;                                                /* launch(): irregular packets on s_side (ALACGPU_SIDE: 0 never, 1 up to 6 x CUs wave slots, 2 always: the default) */
    std::thread* ahead;                                      /* alacgpu_decode_batch_start: the decode in flight (nullptr: none) */
    int ahead_rc;
    char ahead_err[512];
    size_t last_n;                                           /* the last device decode: packets, packets per wave slot, PairArgs::cap */
    uint32_t last_ppw, last_cap, last_fit5;
    uint32_t fit_force;                                      /* PairArgs::fit_force (ALACGPU_FIT) */
    uint32_t order_exp;                                      /* ALACGPU_FIRST: 4 / 5 / 6, the launch of the narrow slots that goes first (experiments) */
};

namespace {

size_t plan_claims_offset() { return (sizeof(Plan) + 255u) & ~(size_t)255u; }
size_t plan_bytes(size_t waves) { return plan_claims_offset() + waves * 4 * sizeof(uint32_t); }

/* wave pairs of a pair kernel one CU holds at a time (registers and LDS, as the runtime computes it); asked once per
 * kernel: the cache is keyed by the kernel's address (every pair kernel has the same function type, so a function-local
 * static of a template on that type would be ONE cache for all of them); every device here is the same model */
uint32_t pair_capacity(void (*kernel)(PairArgs)) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> cache;
    std::lock_guard<std::mutex> g(mu);
    for (const auto& e : cache)
        if (e.first == (const void*)kernel) return (uint32_t)e.second;
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, (int)(2 * kWave), 0) != hipSuccess || v <= 0) v = 4;
    if (const char* e = getenv("ALACGPU_PAIR_CAP")) { /* experiments: fewer pairs per CU than would fit */
        const int lim = atoi(e);
        if (lim >= 1 && lim < v) v = lim;
    }
    cache.emplace_back((const void*)kernel, v);
    return (uint32_t)v;
}

/* static LDS of a four-wave kernel (what the "fit 4" launch pads up to kQuadLdsFit4: alac_gpu.h), asked once per kernel */
uint32_t quad_static_lds(void (*kernel)(PairArgs)) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, uint32_t>> cache;
    std::lock_guard<std::mutex> g(mu);
    for (const auto& e : cache)
        if (e.first == (const void*)kernel) return e.second;
    hipFuncAttributes at;
    uint32_t v = kQuadLdsFit4; /* unknown: no pad, and no second launch */
    if (hipFuncGetAttributes(&at, (const void*)kernel) == hipSuccess) v = (uint32_t)at.sharedSizeBytes;
    else (void)hipGetLastError();
    cache.emplace_back((const void*)kernel, v);
    return v;
}

/* Numbers the device's compute units for the pair kernels (k_decode_body.inc): XCD by XCD, so that CU number c is on
 * XCD c mod 8 when there are eight of them; within an XCD in the order of the hardware ids. The table is indexed by
 * XCC_ID << 6 | SE << 4 | CU and holds number + 1. */
int cu_numbers(alacgpu_decoder* dec) {
    int rc;
    if ((rc = dec->cu_number.ensure(513 * sizeof(uint32_t)))) return rc;
    HIP_TRY(hipMemsetAsync(dec->cu_number.p, 0, 513 * sizeof(uint32_t), dec->stream));
    hipLaunchKernelGGL(alac_cu_census, dim3(2 * dec->n_cu), dim3(kWave), 0, dec->stream, (uint32_t*)dec->cu_number.p,
                       (uint32_t*)dec->cu_number.p + 512, 2 * dec->n_cu);
    HIP_TRY(hipGetLastError());
    uint32_t seen[512];
    HIP_TRY(hipMemcpyAsync(seen, dec->cu_number.p, sizeof(seen), hipMemcpyDeviceToHost, dec->stream));
    HIP_TRY(hipStreamSynchronize(dec->stream));
    uint32_t per_xcd[8] = {0}, xcds = 0, total = 0;
    for (uint32_t i = 0; i < 512; i++)
        if (seen[i]) {
            per_xcd[i >> 6]++;
            total++;
        }
    for (uint32_t x = 0; x < 8; x++) xcds += per_xcd[x] ? 1u : 0u;
    if (xcds == 0) xcds = 1;
    uint32_t next[8] = {0}, order[8] = {0}, k = 0;
    for (uint32_t x = 0; x < 8; x++)
        if (per_xcd[x]) order[x] = k++;
    for (uint32_t i = 0; i < 512; i++)
        if (seen[i]) {
            const uint32_t x = i >> 6;
            seen[i] = 1u + next[x]++ * xcds + order[x]; /* numbers above n_cu (uneven XCDs) only ever take what is left */
        }
    if (total > dec->n_cu) dec->n_cu = total;
    HIP_TRY(hipMemcpyAsync(dec->cu_number.p, seen, sizeof(seen), hipMemcpyHostToDevice, dec->stream));
    HIP_TRY(hipStreamSynchronize(dec->stream));
    return ALACGPU_E_OK;
}

/* upper bound on the waves of a batch: every key present may end in one partly filled wave. Regular keys: 18 x 18 pairs
 * of predictor orders (0..16 and 31), once more for the wide channels that only 24- and 32-bit streams can have
 * (alac_regular.h: chanBits > 23), plus the irregular keys. */
size_t max_waves(uint32_t bit_depth, size_t n, uint32_t ppw) {
    const size_t keys = (bit_depth >= 24 ? 2 : 1) * 18 * 18 + 8;
    return (n + ppw - 1) / ppw + std::min<size_t>(n, keys);
}
size_t max_waves(const alacgpu_decoder* dec, size_t n, uint32_t ppw) { return max_waves(dec->cfg.bit_depth, n, ppw); }

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
    const size_t waves = max_waves(dec, n, ppw);
    int rc;
    if ((rc = dec->scratch_u.ensure(waves * u_tile_cells(dec->cfg.frame_length) * sizeof(int32_t)))) return rc;
    if ((rc = dec->scratch_g.ensure(waves * kFallbackSlots * ppw * sizeof(int32_t)))) return rc;
    /* the plan and, behind it, the claim flags of the pair kernels (one per wave slot): zeroed together */
    if ((rc = dec->plan.ensure(plan_bytes(waves)))) return rc;
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
    dec->last_n = n;
    dec->last_ppw = ppw;
    dec->last_cap = dec->last_fit5 = 0;
    alac::DevCfg c = dec->dev_cfg;
    c.aligned16 = (out_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(d_out) % 16) == 0) ? 1u : 0u;
    Plan* plan = (Plan*)dec->plan.p;
#ifdef ALAC_DUO_PROF
    g_prof_plan = plan;
#endif
    const uint32_t* sz = (const uint32_t*)dec->sizes_ws.p; /* the checked sizes (alac_classify) */
    const uint32_t nb = (uint32_t)((n + 255) / 256);
    const uint32_t slot = (uint32_t)(dec->launches % kTimingSlots);
    HIP_TRY(hipEventRecord(dec->ev_start[slot], dec->stream));
    HIP_TRY(hipMemsetAsync(plan, 0, plan_bytes(max_waves(dec, n, ppw)), dec->stream));
    hipLaunchKernelGGL(alac_classify, dim3(nb), dim3(256), 0, dec->stream, c, d_blob, blob_bytes, d_offsets, d_sizes,
                       (uint32_t)n, (uint16_t*)dec->cls.p, (uint32_t*)dec->sizes_ws.p, d_frames, d_status, plan);
    hipLaunchKernelGGL(alac_plan, dim3(1), dim3(kWave), 0, dec->stream, plan, ppw);
    hipLaunchKernelGGL(alac_scatter, dim3(nb), dim3(256), 0, dec->stream, (const uint16_t*)dec->cls.p, (uint32_t)n, plan,
                       (uint32_t*)dec->perm.p);
    /* Packets of one or two channels: what the irregular ones need (alac_scan, then alac_interleave and alac_legacy: a
     * handful of waves, or none) depends on the sort alone and touches no wave slot, packet or descriptor of a regular
     * one, so it runs on a stream of its own BESIDE the workgroups of the regular packets instead of before and behind them
     * (65 536 stereo packets with 328 escape packets among them: 17 + 22 + 4 us and three kernel boundaries off the
     * decode). s_side leaves `stream` behind the sort (ev_fork) and is back before the stop event (ev_join): whoever waits
     * for `stream` waits for it too. With more than two channels the scan IS the decode and everything stays in line.
     * Round 3 did this only for batches of up to 6 x CUs wave slots: alac_scan and alac_legacy were launched with one
     * 201-register, 26-KB workgroup per wave slot of the whole batch and alac_interleave with up to 32 768 blocks, nearly all
     * of which exit at once — but first stand in the dispatcher's line with the later rounds of the decode kernels, on a device
     * whose register files are full (131 072 packets 4.01 -> 4.49 ms). Their grids are bounded now (scan_grid below, k_scan.hip;
     * the interleave grid), and every batch size forks (profiles/r04_final/side_stream.txt: 131 072 packets 3.85 / 3.77 ms in
     * line, 4.00 / 3.76 beside; 196 608: 5.62 -> 5.57; 24-bit 131 072: 4.68 -> 4.65). ALACGPU_SIDE=0: everything in line;
     * 1: round 3's limit of 6 x CUs wave slots. */
    const bool forked = dec->cfg.num_channels <= 2 && alac::lean_config(c) &&
                        (dec->side >= 2 || (dec->side == 1 && (n + ppw - 1) / ppw <= (size_t)6 * dec->n_cu));
    /* workgroups of alac_scan / alac_legacy: they walk the irregular wave slots (the first plan->irr_waves of the plan,
     * a number only the device knows). Beside the regular packets' kernels: two per CU, a handful of irregular slots is the
     * rule there; where the scan IS the decode (more than two channels, or a configuration the lean kernels do not take):
     * as many as the device holds (201 registers: two waves per SIMD; 26 KB of LDS: six workgroups per CU). */
    const uint32_t scan_grid = (uint32_t)std::min<size_t>(max_waves(dec, n, ppw), (size_t)(forked ? 2u : 6u) * dec->n_cu);
    hipStream_t irr = forked ? dec->s_side : dec->stream;
    if (forked) {
        HIP_TRY(hipEventRecord(dec->ev_fork, dec->stream));
        HIP_TRY(hipStreamWaitEvent(dec->s_side, dec->ev_fork, 0));
    }
    /* from here on an error must not leave s_side running behind the caller's back */
    auto rest = [&]() -> int {
    auto scan = [&]() {
        hipLaunchKernelGGL(alac_scan, dim3(scan_grid), dim3(kWave), 0, irr, c, d_blob, blob_bytes,
                           d_offsets, sz, (const uint32_t*)dec->perm.p, (const Plan*)plan, d_out, (uint64_t)out_stride, d_frames,
                           d_status, (int32_t*)dec->scratch_u.p, (int32_t*)dec->scratch_g.p, ppw, (alac::ChanDesc*)dec->cd.p,
                           (alac::PktDesc*)dec->pd.p, dec->cfg.num_channels > 2 ? (int32_t*)dec->rows.p : (int32_t*)nullptr,
                           (uint64_t)row_stride_of(dec->cfg.frame_length));
    };
    /* in line: irregular packets first. Beside: the regular packets' workgroups go first and fill the device (a scan
     * wave needs 201 registers and 26 KB of LDS: it finds room where the first of them have finished, long before the
     * slowest has) */
    if (!forked) scan();
    if (dec->cfg.num_channels <= 2 && alac::lean_config(c)) {
        /* one kernel per class of regular packets (alac_gpu.h, k_decode_body.inc): each one is launched over all the wave
         * slots and leaves the slots of the other classes alone. gated_cap: how many pairs per CU the gated twin of the
         * kernel can hold (0: it has none); which of the twins works is decided on the device. */
        PairArgs a{c, d_blob, (uint64_t)blob_bytes, d_offsets, sz, (const uint32_t*)dec->perm.p, plan, d_out, (uint64_t)out_stride,
                   d_frames, d_status, (int32_t*)dec->scratch_u.p, (const uint32_t*)dec->cu_number.p,
                   (uint32_t*)((uint8_t*)plan + plan_claims_offset()), ppw, dec->n_cu, 0u, dec->lanes_min, 4u, dec->fit_force, 0u};
        const uint32_t slots = (uint32_t)max_waves(dec, n, ppw);
        auto pairs = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(slots), dim3(2 * kWave), 0, dec->stream, a); };
        /* The narrow regular wave slots go to ONE of up to three launches, and which one is decided on the device from the
         * plan's count of them (alac_gpu.h: decode_mode; the host only knows an upper bound of ALL slots): the four-wave
         * kernel padded with dynamic LDS to 34 KB per workgroup ("fit 4": four per CU), the same kernel without the pad
         * ("fit 5"), and for 16-bit streams the gated twin of wave pairs. "Fit 4" is always launched; the other two for
         * batches of more than 4 x CUs x 64 packets (`beyond4`), and the device is TOLD whether they are (PairArgs::fit5, cap):
         * it never picks a launch that was not made. (Round 3, found by tools/gpu_fuzz.py: a host-side guess once skipped a
         * launch the device then relied on.) A smaller batch whose many keys push it over 4 x CUs slots all the same runs
         * rounds of four, as in round 3; in exchange the benchmark batch and everything below it see no empty grids at all.
         * The launches that are not the batch's exit at once — but empty grids right IN FRONT of a wave-pair kernel cost it
         * up to 13 % (16-bit 98 304 packets 3.15 -> 3.66 ms with the empty "fit 4" and "fit 5" grids in front of the gated
         * twin, 24-bit stereo without shift bytes 5.2 -> 8.4 ms with both in front of alac_decode_w24; one alone, or any
         * number behind: nothing; the four-wave launches do not care; profiles/r04_final/launch_order*.txt), so the launch
         * the host expects to work — decode_mode() of the batch as if every packet were a narrow regular one — goes first,
         * the wide keys' pairs second, the rest behind. A wrong guess costs speed, never correctness. */
        const uint32_t n_cu = dec->n_cu;
        const bool beyond4 = (n + ppw - 1) / ppw > (size_t)4 * n_cu || dec->fit_force == kModeFit5;
        const bool has_twin = dec->cfg.bit_depth == 16 && beyond4 && dec->fit_force == 0u;
        auto quad_kernel = [&]() -> void (*)(PairArgs) {
            return dec->cfg.bit_depth == 16 ? alac_decode_16q : dec->cfg.bit_depth == 32 ? alac_decode_32q : alac_decode_24q; /* 24: 20 and 24 */
        };
        {
            const uint32_t stat = quad_static_lds(quad_kernel());
            a.fit5 = (beyond4 && (size_t)((stat + 1279u) / 1280u) * 1280u * 5u <= 163840u) ? 1u : 0u; /* 1280-byte granules of 160 KB */
        }
        if (has_twin) a.cap = pair_capacity(alac_decode_16g);
        if (a.cap <= 4u) a.cap = 0u;
        dec->last_cap = a.cap;
        dec->last_fit5 = a.fit5;
        auto narrow = [&](uint32_t mode) {
            if (mode == kModeGated) {
                /* as many workgroups as the device holds at once: they share the slots out among themselves */
                if (a.cap) hipLaunchKernelGGL(alac_decode_16g, dim3(std::min<uint32_t>(a.cap * n_cu, slots)), dim3(2 * kWave), 0, dec->stream, a);
                return;
            }
            if (mode == kModeFit5 && !a.fit5) return;
            const uint32_t stat = quad_static_lds(quad_kernel());
            PairArgs q = a;
            q.fit = mode;
            hipLaunchKernelGGL(quad_kernel(), dim3(slots), dim3(4 * kWave), (mode == kModeFit4 && stat < kQuadLdsFit4) ? kQuadLdsFit4 - stat : 0u,
                               dec->stream, q);
        };
        uint32_t guess = decode_mode((uint32_t)((n + ppw - 1) / ppw), n_cu, a.cap, dec->fit_force, dec->cfg.num_channels == 1, a.fit5 != 0u);
        if (dec->order_exp) guess = dec->order_exp; /* experiments (ALACGPU_FIRST): which launch goes first */
        narrow(guess);
        /* the wide keys (chanBits > 23: 24- and 32-bit streams without their usual shift bytes): wave pairs */
        if (dec->cfg.bit_depth == 32) pairs(alac_decode_w32);
        else if (dec->cfg.bit_depth == 24) pairs(alac_decode_w24);
        for (uint32_t mode : {kModeFit4, kModeFit5, kModeGated})
            if (mode != guess) narrow(mode);
    }
    if (forked) scan();
    HIP_TRY(hipGetLastError());
    if (alac::lean_config(c)) {
        /* irregular packets were only scanned by alac_scan (status, frames, channel descriptors) */
        const uint64_t rs = row_stride_of(dec->cfg.frame_length);
        /* frames per interleave block: one wavefront's worth keeps more blocks in flight per CU (the kernel waits on
         * memory, not on arithmetic) */
        const uint32_t il_threads = dec->il_threads;
        const uint32_t bpp = (dec->cfg.frame_length + il_threads - 1u) / il_threads;
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
            hipLaunchKernelGGL(alac_chan_predict, dim3((uint32_t)max_waves(16u, n_slots, ppw2)), dim3(kWave), 0, dec->stream, c,
                               d_blob, blob_bytes, d_offsets, sz, (const uint32_t*)dec->perm2.p, (const Plan*)plan2,
                               (const alac::ChanDesc*)dec->cd.p, (int32_t*)dec->rows.p, rs, ppw2);
        }
        /* PCM of the split packets (with one or two channels: of the escape-only packets) */
        /* a block takes eight slices of a packet at a time (k_split.hip: kSlices) */
        /* (beside the regular packets' kernels: a few blocks per CU — it walks the scanned packets, a handful there) */
        const uint32_t ib = (uint32_t)std::min<uint64_t>((uint64_t)n * ((bpp + 7u) / 8u),
                                                         forked ? (uint64_t)4 * dec->n_cu : (uint64_t)8192u * (256u / il_threads));
        /* four frames per lane (k_split.hip: alac_interleave4) where the rows exist (more than two channels) and a frame is a whole
         * number of dwords (the layouts with a register-packed form: the others build their frames byte by byte and only lose
         * occupancy to the bigger kernel: 16-bit 3-channel 3.25 -> 3.53 ms); ALACGPU_IL4=0: one frame per lane everywhere */
        const uint32_t il_four = (dec->cfg.num_channels > 2 && (dec->cfg.num_channels * c.bps) % 4u == 0u && dec->il_four) ? 1u : 0u;
        hipLaunchKernelGGL(il_four ? alac_interleave4 : alac_interleave, dim3(ib), dim3(il_threads), (il_four ? 128u : 32u) * il_threads, irr, c, d_blob, blob_bytes, d_offsets, sz,
                           (const uint32_t*)dec->perm.p, (const Plan*)plan, (const alac::ChanDesc*)dec->cd.p,
                           (const alac::PktDesc*)dec->pd.p, (const int32_t*)dec->rows.p, rs, d_out, (uint64_t)out_stride, bpp);
        hipLaunchKernelGGL(alac_legacy, dim3(scan_grid), dim3(kWave), 0, irr, c, d_blob, blob_bytes,
                           d_offsets, sz, (const uint32_t*)dec->perm.p, (const Plan*)plan, (const alac::PktDesc*)dec->pd.p, d_out,
                           (uint64_t)out_stride, d_frames, d_status, (int32_t*)dec->scratch_u.p, (int32_t*)dec->scratch_g.p, ppw);
        HIP_TRY(hipGetLastError());
    }
    if (forked) {
        HIP_TRY(hipEventRecord(dec->ev_join, dec->s_side));
        HIP_TRY(hipStreamWaitEvent(dec->stream, dec->ev_join, 0));
    }
    return ALACGPU_E_OK;
    };
    rc = rest();
    if (rc != ALACGPU_E_OK) {
        if (forked) (void)hipStreamSynchronize(dec->s_side);
        return rc;
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

/* ---- handles between uses -------------------------------------------------------------------------------------
 * A file decoder makes a handle, decodes a window or two and destroys it (decode.go:50-80 / stream.py); building one
 * from nothing costs three streams, the events, the CU census with its two synchronisations, and at the first decode a
 * dozen hipMalloc and the pinned staging buffers — 15 ms, where the decode of a 10-second file takes 3. Destroyed
 * handles therefore keep their device-side belongings in a small per-process pool (a few per device, big buffers
 * dropped) and alacgpu_create takes one from there when it can. alacgpu_trim() empties the pool. */
namespace {
std::mutex g_pool_mu;
std::vector<alacgpu_decoder*> g_pool;
std::vector<std::pair<void*, size_t>> g_host_pool;  /* alacgpu_host_free: pinned buffers waiting for the next alacgpu_host_alloc */
std::vector<std::pair<void*, size_t>> g_host_sizes; /* live alacgpu_host_alloc buffers and their sizes */
constexpr size_t kPoolPerDevice = 4;
/* what a pooled handle keeps: device buffers of at most 2 GB in all, pinned host staging of at most 64 MB in all (the
 * largest go first). Device memory is what an MI355X has plenty of (288 GB), and what a file decoder's handles need most of:
 * U hand-off tiles of 1 MB per wave slot, 912 slots for a 1 024-packet window of a 24-bit stream (max_waves) — 1 GB that
 * round 4's first file benchmark allocated and freed per file, with a device-wide hipFree in the middle of the stream.
 * Four handles per device: at most 8 GB of device memory and 0.25 GB of pinned memory per device stay allocated behind
 * alacgpu_destroy() until alacgpu_trim(). */
constexpr size_t kPoolKeepDevice = (size_t)2048 << 20;
constexpr size_t kPoolKeepPinned = (size_t)64 << 20;
/* alacgpu_host_free keeps blocks of up to 64 MB, the eight newest, 192 MB in all (three windows of a file decoder, a few sizes) */
constexpr size_t kHostPoolBlock = (size_t)64 << 20, kHostPoolBytes = (size_t)192 << 20, kHostPoolCount = 8;
template <class Buf>
void keep_at_most(std::vector<Buf*> bufs, size_t limit) {
    for (;;) {
        size_t total = 0;
        Buf* big = nullptr;
        for (Buf* b : bufs) {
            total += b->cap;
            if (b->cap && (!big || b->cap > big->cap)) big = b;
        }
        if (total <= limit || !big) return;
        big->release();
    }
}

void really_destroy(alacgpu_decoder* d) {
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->s_in) (void)hipStreamSynchronize(d->s_in);
    if (d->s_out) (void)hipStreamSynchronize(d->s_out);
    if (d->s_side) (void)hipStreamSynchronize(d->s_side);
    delete d->pool;
    DevBuf* bufs[] = {&d->cu_number, &d->scratch_u, &d->scratch_g, &d->plan, &d->cls, &d->perm, &d->sizes_ws, &d->cd, &d->pd,
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
    if (d->s_side) (void)hipStreamDestroy(d->s_side);
    if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
    if (d->ev_join) (void)hipEventDestroy(d->ev_join);
    delete d;
}

/* the per-configuration part of a handle (everything else survives in the pool) */
void configure(alacgpu_decoder* d, const alacgpu_config* cfg, int bps) {
    d->cfg = *cfg;
    d->frame_bytes = (size_t)cfg->frame_length * cfg->num_channels * (size_t)bps;
    d->dev_cfg = alac::DevCfg{cfg->frame_length, cfg->bit_depth, cfg->num_channels, cfg->pb, cfg->mb, cfg->kb,
                              (uint32_t)bps, 0u};
    d->launches = 0;
    d->ahead = nullptr;
    d->ahead_rc = ALACGPU_E_OK;
    d->ahead_err[0] = 0;
    d->last_n = 0;
    d->last_ppw = d->last_cap = d->last_fit5 = 0;
    d->il_threads = 64;
    if (const char* e = getenv("ALACGPU_IL_THREADS")) {
        const int v = atoi(e);
        if (v == 64 || v == 128 || v == 256) d->il_threads = (uint32_t)v;
    }
    d->il_four = 1;
    if (const char* e = getenv("ALACGPU_IL4")) d->il_four = (uint32_t)atoi(e);
    d->lanes_min = 4;
    d->fit_force = 0;
    if (const char* e = getenv("ALACGPU_FIT")) d->fit_force = (uint32_t)atoi(e); /* experiments: 4 / 5 workgroups per CU for every batch */
    d->order_exp = 0;
    if (const char* e = getenv("ALACGPU_FIRST")) d->order_exp = (uint32_t)atoi(e);
    if (const char* e = getenv("ALACGPU_LANES_MIN")) d->lanes_min = (uint32_t)std::max(1, atoi(e)); /* experiments; 17: never */
    d->side = 2;
    if (const char* e = getenv("ALACGPU_SIDE")) d->side = atoi(e); /* experiments, tests */
    d->chunk_bytes = (size_t)192 << 20;
    if (const char* e = getenv("ALACGPU_CHUNK_MB")) {
        const long v = atol(e);
        if (v >= 1 && v <= 65536) d->chunk_bytes = (size_t)v << 20;
    }
}
} /* namespace */

static int settle_ahead(alacgpu_decoder* d);

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
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i]->device == device) {
                alacgpu_decoder* d = g_pool[i];
                g_pool.erase(g_pool.begin() + (long)i);
                configure(d, cfg, bps);
                *out = d;
                return ALACGPU_E_OK;
            }
    }
    alacgpu_decoder* d = new (std::nothrow) alacgpu_decoder();
    if (!d) {
        set_err("out of memory");
        return ALACGPU_E_ARG;
    }
    d->device = device;
    d->pool = nullptr;
    configure(d, cfg, bps);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 0;
        d->n_cu = cus > 0 ? (uint32_t)cus : 256u;
    }
    d->stream = d->s_in = d->s_out = d->s_side = nullptr;
    d->ev_fork = d->ev_join = nullptr;
    for (uint32_t i = 0; i < kTimingSlots; i++) d->ev_start[i] = d->ev_stop[i] = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->s_out, hipStreamNonBlocking);
    /* (the default priority: at the lowest one the side kernels linger until the decode's last workgroups have gone and
     * become its tail: 131 072 packets 3.85 -> 4.38 ms, 24-bit 4.68 -> 5.97; profiles/r04_final/side_stream.txt) */
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->s_side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming);
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
        really_destroy(d);
        return ALACGPU_E_HIP;
    }
    if (int rc = cu_numbers(d)) {
        really_destroy(d);
        return rc;
    }
    *out = d;
    return ALACGPU_E_OK;
}

void alacgpu_destroy(alacgpu_decoder* d) {
    if (!d) return;
    (void)settle_ahead(d);
    (void)hipSetDevice(d->device);
    /* nothing of this handle's last call may still be running when its belongings go to the next owner */
    bool ok = hipStreamSynchronize(d->stream) == hipSuccess && hipStreamSynchronize(d->s_in) == hipSuccess &&
              hipStreamSynchronize(d->s_out) == hipSuccess && hipStreamSynchronize(d->s_side) == hipSuccess;
    if (ok) {
        for (int k = 0; k < kSlots; k++) d->slots[k].busy = false;
        std::vector<DevBuf*> dev = {&d->scratch_u, &d->scratch_g, &d->plan, &d->cls, &d->perm, &d->sizes_ws, &d->cd, &d->pd,
                                    &d->plan2, &d->keys2, &d->perm2, &d->rows};
        std::vector<decltype(&d->slots[0].h_in)> pinned;
        for (int k = 0; k < kSlots; k++) {
            Slot& s = d->slots[k];
            dev.push_back(&s.d_in);
            dev.push_back(&s.d_out);
            pinned.push_back(&s.h_in);
            pinned.push_back(&s.h_out);
        }
        keep_at_most(dev, kPoolKeepDevice);
        keep_at_most(pinned, kPoolKeepPinned);
        std::lock_guard<std::mutex> g(g_pool_mu);
        size_t same = 0;
        for (alacgpu_decoder* p : g_pool) same += p->device == d->device ? 1u : 0u;
        if (same < kPoolPerDevice) {
            g_pool.push_back(d);
            return;
        }
    }
    really_destroy(d);
}

void alacgpu_trim(void) {
    std::vector<alacgpu_decoder*> all;
    std::vector<std::pair<void*, size_t>> host;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        all.swap(g_pool);
        host.swap(g_host_pool);
    }
    for (alacgpu_decoder* d : all) really_destroy(d);
    for (auto& h : host) (void)hipHostFree(h.first);
}

/* Pinned host memory for a caller's PCM (or packet) buffers: what alacgpu_decode_batch finds in pinned memory it
 * transfers in place, without the copy through its staging. Freed buffers of up to 64 MB are kept (the eight newest, 192 MB
 * in all; the smallest that fits is handed out again): a file decoder asks for the same sizes file after file, and
 * hipHostMalloc / hipHostFree cost milliseconds each and stall the device. */
void* alacgpu_host_alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < g_host_pool.size(); i++)
            if (g_host_pool[i].second >= bytes && g_host_pool[i].second <= 2 * bytes + (1u << 20) &&
                (best == SIZE_MAX || g_host_pool[i].second < g_host_pool[best].second))
                best = i;
        if (best != SIZE_MAX) {
            void* p = g_host_pool[best].first;
            g_host_sizes.emplace_back(p, g_host_pool[best].second);
            g_host_pool.erase(g_host_pool.begin() + (long)best);
            return p;
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_err("hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    std::lock_guard<std::mutex> g(g_pool_mu);
    g_host_sizes.emplace_back(p, bytes);
    return p;
}

void alacgpu_host_free(void* p) {
    if (!p) return;
    size_t bytes = 0;
    std::vector<void*> evict;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        for (size_t i = 0; i < g_host_sizes.size(); i++)
            if (g_host_sizes[i].first == p) {
                bytes = g_host_sizes[i].second;
                g_host_sizes.erase(g_host_sizes.begin() + (long)i);
                break;
            }
        if (bytes && bytes <= kHostPoolBlock) {
            /* newest last; what no longer fits goes oldest first (round 4: a pool of four blocks, first come first kept,
             * filled up with the small windows of short files and then made every long file allocate and free its own) */
            g_host_pool.emplace_back(p, bytes);
            p = nullptr;
            size_t total = 0;
            for (const auto& h : g_host_pool) total += h.second;
            while (g_host_pool.size() > kHostPoolCount || total > kHostPoolBytes) {
                evict.push_back(g_host_pool.front().first);
                total -= g_host_pool.front().second;
                g_host_pool.erase(g_host_pool.begin());
            }
        }
    }
    for (void* q : evict) (void)hipHostFree(q);
    if (p) (void)hipHostFree(p);
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

/* the decode alacgpu_decode_batch_start put on a thread of its own, if any: wait for it and take over its result */
static int settle_ahead(alacgpu_decoder* d) {
    if (!d->ahead) return ALACGPU_E_OK;
    d->ahead->join();
    delete d->ahead;
    d->ahead = nullptr;
    if (d->ahead_rc != ALACGPU_E_OK) set_err("%s", d->ahead_err);
    return d->ahead_rc;
}

int alacgpu_decode_batch_start(alacgpu_decoder* d, const uint8_t* blob, size_t blob_bytes, const uint64_t* offsets, size_t n,
                               uint8_t* out, size_t out_stride, uint32_t* frames_out, int32_t* status) {
    if (!d) return ALACGPU_E_ARG;
    if (d->ahead) {
        set_err("a decode started with alacgpu_decode_batch_start is still in flight: alacgpu_decode_batch_wait first");
        return ALACGPU_E_ARG;
    }
    d->ahead_rc = ALACGPU_E_OK;
    d->ahead = new (std::nothrow) std::thread([=] {
        d->ahead_rc = alacgpu_decode_batch(d, blob, blob_bytes, offsets, n, out, out_stride, frames_out, status);
        if (d->ahead_rc != ALACGPU_E_OK) snprintf(d->ahead_err, sizeof(d->ahead_err), "%s", alacgpu_last_error());
    });
    if (!d->ahead) {
        set_err("out of memory");
        return ALACGPU_E_ARG;
    }
    return ALACGPU_E_OK;
}

int alacgpu_decode_batch_wait(alacgpu_decoder* d) {
    if (!d) return ALACGPU_E_ARG;
    return settle_ahead(d);
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
 * they lie in the caller's blob (no re-pack: the kernels read dense blobs), together with their offsets and sizes, in
 * ONE transfer; PCM, frame counts and status words come back in ONE. Pageable caller memory is staged through pinned
 * buffers by a few copy threads; memory the caller pinned itself (hipHostMalloc / hipHostRegister) is used in place.
 * The offsets are untrusted (a sample table read from a file): a packet that does not lie inside [0, blob_bytes), or
 * whose end lies before its start, is never read; it gets ALACGPU_ERR_RANGE like in the device entry.
 */
int alacgpu_decode_batch(alacgpu_decoder* d, const uint8_t* blob, size_t blob_bytes, const uint64_t* offsets, size_t n,
                         uint8_t* out, size_t out_stride, uint32_t* frames_out, int32_t* status) {
    if (!d || (n && (!offsets || !out || !frames_out || !status)) || (blob_bytes && !blob)) {
        set_err("null argument");
        return ALACGPU_E_ARG;
    }
    if (out_stride < d->frame_bytes) {
        set_err("out_stride %zu < frame bytes %zu", out_stride, d->frame_bytes);
        return ALACGPU_E_ARG;
    }
    if (n == 0) return ALACGPU_E_OK;
    auto valid = [&](size_t i) {
        const uint64_t lo = offsets[i], hi = offsets[i + 1];
        return lo <= hi && hi <= (uint64_t)blob_bytes && hi - lo <= 0x0fffffffull;
    };
    HIP_TRY(hipSetDevice(d->device));
    const size_t fb = d->frame_bytes;
    const size_t d_stride = (fb + 15u) & ~(size_t)15u; /* 16-byte aligned device rows: the LDS-staged wide stores */
    const bool out_pinned = is_pinned(out) && is_pinned(frames_out) && is_pinned(status);
    const bool in_pinned = blob_bytes == 0 || is_pinned(blob);
    if (!d->pool && !(out_pinned && in_pinned)) {
        unsigned t = std::thread::hardware_concurrency();
        t = t > 16 ? 8 : (t > 2 ? t / 2 : 1);
        if (const char* e = getenv("ALACGPU_COPY_THREADS")) t = (unsigned)std::max(1, atoi(e));
        d->pool = new (std::nothrow) CopyPool(t - 1); /* the calling thread copies too */
    }

    /* chunk is back in pinned memory (or in place): hand it to the caller; descriptors that left the blob get their
     * status here (the device saw them as empty packets) */
    auto finish = [&](Slot& s) -> int {
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
        if (s.bad)
            for (size_t i = s.first; i < s.first + s.n; i++)
                if (!valid(i)) {
                    status[i] = ALACGPU_ERR_RANGE;
                    frames_out[i] = 0;
                }
        /* The kernels leave the bytes behind a partial frame alone, and the device-side PCM slots of a (pooled) handle
         * hold whatever an earlier batch left there: the caller gets zeros instead, as from DecodePacket's zeroed frame
         * buffer (decoder.go:120,127). One packet in a hundred has a partial frame. */
        {
            const size_t bpf = fb / d->cfg.frame_length; /* bytes per frame */
            for (size_t i = s.first; i < s.first + s.n; i++) {
                const size_t have = std::min<size_t>(frames_out[i], d->cfg.frame_length) * bpf;
                if (have < fb) memset(out + i * out_stride + have, 0, fb - have);
            }
        }
        s.busy = false;
        return ALACGPU_E_OK;
    };
    /* one chunk: upload, decode, download, all asynchronous; HIP errors come back as a code so that the caller drains */
    auto submit = [&](Slot& s, size_t first, size_t cnt, uint64_t lo, uint64_t hi) -> int {
        int rc;
        s.first = first;
        s.n = cnt;
        s.bad = false;
        const size_t in_bytes = (size_t)(hi - lo);
        const size_t meta = (cnt + 1) * sizeof(uint64_t) + cnt * sizeof(uint32_t);
        const size_t meta_pad = (meta + 255u) & ~(size_t)255u; /* packet bytes start 256-byte aligned on the device */
        if ((rc = s.d_in.ensure(meta_pad + in_bytes + 16))) return rc;
        if ((rc = s.d_out.ensure(cnt * d_stride + cnt * 8 + 16))) return rc;
        if ((rc = s.h_in.ensure(in_pinned ? meta_pad : meta_pad + in_bytes))) return rc;
        if (!out_pinned && (rc = s.h_out.ensure(cnt * d_stride + cnt * 8))) return rc;
        /* upload: offsets rebased to the chunk's first byte and sizes, then the bytes */
        uint64_t* h_off = (uint64_t*)s.h_in.p;
        uint32_t* h_sz = (uint32_t*)(h_off + cnt + 1);
        for (size_t i = 0; i < cnt; i++) {
            const bool ok = valid(first + i);
            h_off[i] = ok ? offsets[first + i] - lo : 0u;
            h_sz[i] = ok ? (uint32_t)(offsets[first + i + 1] - offsets[first + i]) : 0u;
            s.bad = s.bad || !ok;
        }
        h_off[cnt] = in_bytes;
        uint8_t* d_in = (uint8_t*)s.d_in.p;
        if (in_pinned) {
            HIP_TRY(hipMemcpyAsync(d_in, h_off, meta, hipMemcpyHostToDevice, d->s_in));
            if (in_bytes) HIP_TRY(hipMemcpyAsync(d_in + meta_pad, blob + lo, in_bytes, hipMemcpyHostToDevice, d->s_in));
        } else {
            uint8_t* hb = (uint8_t*)s.h_in.p + meta_pad;
            const size_t pieces = std::max<size_t>(1, std::min<size_t>(64, in_bytes >> 20));
            auto body = [&](size_t k) {
                const size_t a = in_bytes * k / pieces, b = in_bytes * (k + 1) / pieces;
                memcpy(hb + a, blob + lo + a, b - a);
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
        if ((rc = launch(d, d_in + meta_pad, in_bytes, (const uint64_t*)d_in, (const uint32_t*)(d_in + (cnt + 1) * sizeof(uint64_t)), cnt,
                         d_pcm, d_stride, d_fr, d_st)))
            return rc;
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
        return ALACGPU_E_OK;
    };

    size_t first = 0;
    int turn = 0;
    int rc = ALACGPU_E_OK;
    while (first < n && rc == ALACGPU_E_OK) {
        /* packets of this chunk: whole packets, about chunk_bytes of traffic (the span of the blob they cover in + PCM
         * out). With a sane table the span is offsets[first] .. offsets[first + cnt]. */
        size_t cnt = 0;
        uint64_t lo = ~0ull, hi = 0;
        while (first + cnt < n) {
            uint64_t nlo = lo, nhi = hi;
            if (valid(first + cnt)) {
                nlo = std::min(lo, offsets[first + cnt]);
                nhi = std::max(hi, offsets[first + cnt + 1]);
            }
            const uint64_t bytes = (nhi > nlo ? nhi - nlo : 0) + (uint64_t)(cnt + 1) * fb;
            if (cnt != 0 && bytes > d->chunk_bytes) break;
            lo = nlo;
            hi = nhi;
            cnt++;
        }
        if (hi < lo) lo = hi = 0; /* no valid packet in the chunk */
        Slot& s = d->slots[turn];
        turn = (turn + 1) % kSlots;
        if (s.busy && (rc = finish(s))) break;
        rc = submit(s, first, cnt, lo, hi);
        first += cnt;
    }
    if (rc != ALACGPU_E_OK) {
        /* a step of a chunk failed: nothing of this call may still be writing the caller's buffers when we return, and
         * no slot may carry a chunk of this call into the next one */
        (void)hipStreamSynchronize(d->s_in);
        (void)hipStreamSynchronize(d->stream);
        (void)hipStreamSynchronize(d->s_side);
        (void)hipStreamSynchronize(d->s_out);
        for (int k = 0; k < kSlots; k++) d->slots[k].busy = false;
        return rc;
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
    int rc = alacgpu_decode_batch(d, packet_len ? packet : &dummy, packet_len, offs, 1, out, d->frame_bytes, &frames, &st);
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

int alacgpu_pair_placement(alacgpu_decoder* d, uint32_t* tags, size_t max_n, size_t* n_out) {
    if (!d || !tags || !n_out) return ALACGPU_E_ARG;
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    *n_out = 0;
    if (!d->plan.p) return ALACGPU_E_OK;
    Plan head;
    HIP_TRY(hipMemcpy(&head, d->plan.p, sizeof(Plan), hipMemcpyDeviceToHost));
    const size_t n = std::min<size_t>(head.total_waves, max_n / 4);
    if (plan_bytes(n) > d->plan.cap) return ALACGPU_E_ARG;
    if (n) HIP_TRY(hipMemcpy(tags, (const uint8_t*)d->plan.p + plan_claims_offset(), n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n_out = n;
    return ALACGPU_E_OK;
}

int alacgpu_last_dispatch(alacgpu_decoder* d, alacgpu_dispatch* out) {
    if (!d || !out) return ALACGPU_E_ARG;
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    memset(out, 0, sizeof(*out));
    if (!d->plan.p || d->last_n == 0) return ALACGPU_E_OK;
    Plan head;
    HIP_TRY(hipMemcpy(&head, d->plan.p, sizeof(Plan), hipMemcpyDeviceToHost));
    out->packets_per_slot = d->last_ppw;
    out->slots = head.total_waves;
    out->irregular_slots = head.irr_waves;
    out->wide_slots = head.wide_waves;
    out->narrow_slots = head.total_waves - head.irr_waves - head.wide_waves;
    out->keys = head.nk;
    const bool lean = d->cfg.num_channels <= 2 && alac::lean_config(d->dev_cfg);
    /* the device's own decision (k_decode_body.inc reads the same plan and calls the same function) */
    const char* narrow = "";
    if (lean && out->narrow_slots) {
        const char* q = d->cfg.bit_depth == 16 ? "alac_decode_16q" : d->cfg.bit_depth == 32 ? "alac_decode_32q" : "alac_decode_24q";
        const uint32_t mode = decode_mode(out->narrow_slots, d->n_cu, d->last_cap, d->fit_force, d->cfg.num_channels == 1, d->last_fit5 != 0u);
        narrow = mode == kModeGated ? "alac_decode_16g" : q;
        out->gated = mode == kModeGated ? 1u : 0u;
        out->workgroups_per_cu = mode == kModeGated ? pair_quota(out->narrow_slots, d->n_cu, d->last_cap) : mode;
        /* predictor waves on several lanes per packet (k_decode_body.inc: lanes_ok) */
        if (!out->gated && out->narrow_slots <= d->n_cu + d->n_cu / 8u && d->lanes_min <= 16u) out->lanes_per_packet = d->last_ppw <= 32u ? 4u : 2u;
    }
    snprintf(out->narrow_kernel, sizeof(out->narrow_kernel), "%s", narrow);
    snprintf(out->wide_kernel, sizeof(out->wide_kernel), "%s",
             (lean && out->wide_slots) ? (d->cfg.bit_depth == 32 ? "alac_decode_w32" : "alac_decode_w24") : "");
    snprintf(out->irregular_kernels, sizeof(out->irregular_kernels), "%s",
             !out->irregular_slots ? "" : !alac::lean_config(d->dev_cfg) ? "alac_scan (whole-packet decoder)"
             : d->cfg.num_channels > 2 ? ((d->cfg.num_channels * d->dev_cfg.bps) % 4u == 0u && d->il_four
                                              ? "alac_scan + alac_chan_predict + alac_interleave4 (+ alac_legacy)"
                                              : "alac_scan + alac_chan_predict + alac_interleave (+ alac_legacy)")
                                       : "alac_scan + alac_interleave (+ alac_legacy)");
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
/* profiling build only: the stamp sums of the last decode ([0..3] role A, [8..11] role B) */
int alacgpu_debug_prof(unsigned long long* out16) {
    if (!g_prof_plan) return ALACGPU_E_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return ALACGPU_E_HIP;
    if (hipMemcpy(out16, g_prof_plan->prof, sizeof(g_prof_plan->prof), hipMemcpyDeviceToHost) != hipSuccess) return ALACGPU_E_HIP;
    return ALACGPU_E_OK;
}
#endif

const char* alacgpu_version(void) { return "alacgpu 0.5.0 gfx950"; }

} /* extern "C" */
