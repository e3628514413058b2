/*
 * lane_sim.cpp — TEST-ONLY host build of the kernel's decode logic.
 *
 * Compiles saprobe-alac_amd/csrc/alac_wave.h (the exact text the gfx950 kernel is built from) with g++
 * and a one-lane wave policy, so the decode LOGIC can be compared with the oracle in the CPU test-suite
 * (-m "not gpu"), where no GPU exists. It lives under tests/, is never linked into libalacgpu.so and is not
 * a decode path of the product.
 */
#include <sys/mman.h>
#include <unistd.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define ALAC_DEV inline
/* as narrow as the GPU's v_mul_i32_i24: a use on operands that do not fit 24 bits must show up here too */
#define ALAC_SX24(x) (((int32_t)((uint32_t)(x) << 8)) >> 8)
#define ALAC_MUL24(a, b) ((int32_t)((uint32_t)ALAC_SX24(a) * (uint32_t)ALAC_SX24(b)))
/* ... and so are the multiply-adds of alac_regular.h (v_mad_i32_i24, v_mul_u32_u24: operands cut to 24 bits) */
#define ALAC_MAD24(a, b, c) ((int32_t)((uint32_t)ALAC_SX24(a) * (uint32_t)ALAC_SX24(b) + (uint32_t)(c)))
#define ALAC_MSUB24(acc, a, c) ((int32_t)((uint32_t)(acc) - (uint32_t)ALAC_SX24(a) * (uint32_t)ALAC_SX24(c)))
#define ALAC_MULU24(a, b) (((uint32_t)(a) & 0xffffffu) * ((uint32_t)(b) & 0xffffffu))
#include "../../saprobe-alac_amd/csrc/alac_wave.h"
#include "../../saprobe-alac_amd/csrc/alac_regular.h"
#include "../../saprobe-alac_amd/csrc/alac_duo.h"
#include "../../saprobe-alac_amd/csrc/alac_split.h"

namespace {

/* one-lane wave: collectives are identities, the stager writes straight to the PCM slot */
struct HostWave {
    static constexpr bool kResMem = false;
    static constexpr uint32_t kRingDw = 32; /* alac_gpu.h: ALAC_LDS_RING */
    std::vector<int32_t> u_tile, g_tile;
    uint8_t* st_out = nullptr;
    uint32_t st_cnt = 0;
    explicit HostWave(uint32_t frame_length) : u_tile((frame_length ? frame_length : 1) + 1), g_tile(64) {}
    bool any(bool p) const { return p; }
    uint32_t max_u32(uint32_t v) const { return v; }
    void st_begin(uint8_t* out) {
        st_out = out;
        st_cnt = 0;
    }
    void st_push(uint32_t v) {
        memcpy(st_out + 4u * (size_t)st_cnt, &v, 4);
        ++st_cnt;
    }
    void st_push_if(uint32_t v, bool on) {
        if (on) st_push(v);
    }
    uint32_t st_group_base(uint32_t) const { return st_cnt; }
    void st_put(uint32_t base, uint32_t j, uint32_t v, uint32_t inc) {
        if (inc) memcpy(st_out + 4u * (size_t)(base + j), &v, 4);
    }
    void st_advance(uint32_t n) { st_cnt += n; }
    void st_push6_n(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t d4, uint32_t d5, uint32_t count) {
        const uint32_t d[6] = {d0, d1, d2, d3, d4, d5};
        for (uint32_t k = 0; k < count; ++k) st_push(d[k]);
    }
    void st_tail(uint64_t acc, uint32_t nbytes) {
        for (uint32_t b = 0; b < nbytes; ++b) st_out[4u * (size_t)st_cnt + b] = (uint8_t)(acc >> (8u * b));
    }
    void st_step() {}
    uint32_t st_finish() { return st_cnt; }
    uint32_t ring[36] = {0}; /* kRingStride: slot 32 repeats slot 0, slot 33 is spare (RingRd::commit) */
    void ring_write4(uint32_t slot, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        ring[slot] = a;
        ring[slot + 1] = b;
        ring[slot + 2] = c;
        ring[slot + 3] = d;
    }
    uint32_t ring_read(uint32_t slot) const { return ring[slot]; }
    void ring_read2(uint32_t slot, uint32_t& a, uint32_t& b) const {
        a = ring[slot];
        b = ring[slot + 1];
    }
    void ring_read2_at(uint32_t byte_off, uint32_t& a, uint32_t& b) const { ring_read2(byte_off >> 2, a, b); }
    void ring_write1(uint32_t slot, uint32_t v) { ring[slot] = v; }
    /* residual queue of alac_duo.h: one lane, both roles played by the same caller, so the barriers are no-ops */
    int32_t rq[2][2 * alac::DUO_CHUNK] = {{0}};
    void rq_write(uint32_t buf, uint32_t j, int32_t v) { rq[buf][j] = v; }
    int32_t rq_read(uint32_t buf, uint32_t j) const { return rq[buf][j]; }
    void duo_sync() {}
    void duo_sync_mem() {}
    int32_t* u_row(uint32_t i) { return &u_tile[i]; }
    int32_t* g_slot(uint32_t k) { return &g_tile[k]; }
};

/* role B alone, residuals read from the row the scan left them in (GpuWaveMem of alacgpu.hip). The host stager
 * writes a sample the moment it is pushed; the reads run ahead of the writes (sample i is written after residual i
 * was read), as on the GPU. */
struct HostWaveMem : HostWave {
    static constexpr bool kResMem = true;
    const int32_t* res = nullptr;
    uint32_t it = 0, chunk0 = 0;
    explicit HostWaveMem(uint32_t frame_length) : HostWave(frame_length) {}
    int32_t rq_read(uint32_t, uint32_t j) const { return res[chunk0 + j]; }
    void rq_write(uint32_t, uint32_t, int32_t) {}
    void duo_sync() {
        ++it;
        chunk0 = (it - 1u) * alac::DUO_CHUNK;
    }
};

}  // namespace

/* variant: 0..3 = force that class's generic variant (any class must decode any packet correctly);
 *          -1   = route like alacgpu.hip does (wave-pair decoder of alac_duo.h for regular packets, split pipeline
 *                 for > 2 channels, whole-packet decoder otherwise);
 *          -2   = split pipeline for every non-regular packet, whatever the channel count.
 * classes_out (may be null) gets the sort key / route. */
extern "C" int lane_sim_decode_batch(const alacgpu_config* cfg, const uint8_t* blob_in, size_t blob_bytes,
                                     const uint64_t* offsets, const uint32_t* sizes, size_t n, uint8_t* out,
                                     size_t out_stride, uint32_t* frames_out, int32_t* status, int poison, int variant,
                                     uint32_t* classes_out, int guard) {
    /* guard: the blob is copied so that the aligned dword holding its last byte ends at a page boundary and the next
     * page is inaccessible: a reader that looks further than the contract of Bits / RingRd allows dies here */
    const uint8_t* blob = blob_in;
    uint8_t* region = nullptr;
    size_t region_len = 0;
    if (guard) {
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        const size_t span = (blob_bytes + 3u) & ~(size_t)3u;
        region_len = ((span + page - 1) / page + 1) * page;
        region = (uint8_t*)mmap(nullptr, region_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (region == MAP_FAILED) return -1;
        uint8_t* b = region + (region_len - page) - span;
        memset(region, 0xee, region_len - page);
        memcpy(b, blob_in, blob_bytes);
        mprotect(region + region_len - page, page, PROT_NONE);
        blob = b;
    }
    alac::DevCfg dc{};
    dc.frame_length = cfg->frame_length;
    dc.bit_depth = cfg->bit_depth;
    dc.num_channels = cfg->num_channels;
    dc.pb = cfg->pb;
    dc.mb = cfg->mb;
    dc.kb = cfg->kb;
    dc.bps = cfg->bit_depth == 16 ? 2 : cfg->bit_depth == 32 ? 4 : 3;
    dc.aligned16 = (out_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(out) % 16) == 0) ? 1u : 0u;
    HostWave wv(cfg->frame_length);
    for (size_t i = 0; i < n; i++) {
        /* the kernel never relies on scratch or output contents: poison them */
        if (poison) {
            memset(wv.u_tile.data(), 0x5a, wv.u_tile.size() * sizeof(int32_t));
            memset(wv.g_tile.data(), 0x5a, wv.g_tile.size() * sizeof(int32_t));
            memset(out + i * out_stride, 0xa5, out_stride);
        }
        const uint8_t* p = blob + offsets[i];
        const uint64_t left = (uint64_t)blob_bytes - offsets[i];
        const uint32_t avail = left > 0xffffffffull ? 0xffffffffu : (uint32_t)left;
        uint8_t* o = out + i * out_stride;
        frames_out[i] = 0;
        if (variant < 0) {
            /* like the GPU pre-pass: regular packets take the lean decoder */
            const uint32_t key = alac::classify_regular(dc, p, sizes[i], avail);
            if (key != alac::KEY_IRREGULAR) {
                if (classes_out) classes_out[i] = key;
                /* the same instantiations as the GPU library's kernels, one per sample width (k_dec16 / 24 / 32.hip) */
                if (dc.bit_depth == 16)
                    status[i] = alac::decode_regular_duo<HostWave, alac::ROLE_BOTH, -1, 16, true>(wv, dc, key, true, p, sizes[i], avail, o, &frames_out[i]);
                else if (dc.bit_depth == 32)
                    status[i] = alac::decode_regular_duo<HostWave, alac::ROLE_BOTH, -1, 32, true>(wv, dc, key, true, p, sizes[i], avail, o, &frames_out[i]);
                else
                    status[i] = alac::decode_regular_duo<HostWave, alac::ROLE_BOTH, -1, 24, true>(wv, dc, key, true, p, sizes[i], avail, o, &frames_out[i]);
                continue;
            }
        }
        if (variant < 0 && alac::lean_config(dc)) {
            /* split pipeline, as alacgpu.hip runs it: scan -> one lean phase per channel -> interleave */
            alac::ChanDesc cd[8];
            memset(cd, 0, sizeof(cd));
            alac::PktDesc pd{};
            /* as on the GPU: with more than two channels the scan leaves the residuals in the rows */
            const size_t rs = (cfg->frame_length + 3u) & ~3u;
            std::vector<int32_t> rows(rs * 8, 0x5a5a5a5a);
            const bool keep_res = dc.num_channels > 2;
            status[i] = alac::decode_wave<HostWave, 16, true, true>(wv, dc, true, p, sizes[i], avail, o, &frames_out[i], cd, &pd,
                                                                    keep_res ? rows.data() : nullptr, rs);
            if (classes_out) classes_out[i] = 2048u + pd.route;
            if (status[i] != 0) continue;
            if (pd.route == alac::ROUTE_SPLIT) {
                for (uint32_t sl = 0; sl < pd.nslots; ++sl) {
                    if (!(cd[sl].info & alac::CD_VALID) || (cd[sl].info & alac::CD_ESCAPE)) continue;
                    if (keep_res) { /* predictor pass over the stored residuals (alac_chan_predict) */
                        HostWaveMem wm(cfg->frame_length);
                        wm.res = rows.data() + rs * sl;
                        alac::decode_channel_task<HostWaveMem, alac::ROLE_B>(wm, dc, alac::chan_task_key(dc, cd[sl]), true, p, sizes[i],
                                                                             avail, cd[sl], rows.data() + rs * sl);
                    } else {
                        alac::decode_channel_task<HostWave, alac::ROLE_BOTH>(wv, dc, alac::chan_task_key(dc, cd[sl]), true, p, sizes[i],
                                                                              avail, cd[sl], rows.data() + rs * sl);
                    }
                }
                /* frames of whole dwords with sample rows: four frames per "lane", as k_split.hip's il_chunk4 loads and builds them */
                bool done4 = false;
                if (keep_res) {
#define LANE_IL4_CASE(NC_, BPS_)                                                                                         \
    case (NC_) * 8 + (BPS_): {                                                                                           \
        for (uint32_t f0 = 0; f0 < pd.frames; f0 += 4) {                                                                 \
            alac::IlLoaded<NC_> L4[4];                                                                                   \
            alac::interleave_load4<NC_, BPS_>(dc, p, sizes[i], avail, pd, cd, rows.data(), rs, f0, L4);                  \
            for (uint32_t j = 0; j < 4 && f0 + j < pd.frames; ++j) {                                                     \
                uint32_t fr[(NC_) * (BPS_) / 4];                                                                         \
                alac::interleave_build<NC_, BPS_>(dc, pd, cd, f0 + j, L4[j], fr);                                        \
                memcpy(o + (size_t)(f0 + j) * dc.num_channels * dc.bps, fr, sizeof(fr));                                 \
            }                                                                                                            \
        }                                                                                                                \
        done4 = true;                                                                                                    \
        break;                                                                                                           \
    }
                    switch (dc.num_channels * 8u + dc.bps) {
                        LANE_IL4_CASE(4, 2) LANE_IL4_CASE(6, 2) LANE_IL4_CASE(8, 2) LANE_IL4_CASE(4, 3) LANE_IL4_CASE(8, 3)
                        LANE_IL4_CASE(3, 4) LANE_IL4_CASE(4, 4) LANE_IL4_CASE(5, 4) LANE_IL4_CASE(6, 4) LANE_IL4_CASE(7, 4) LANE_IL4_CASE(8, 4)
                        default: break;
                    }
#undef LANE_IL4_CASE
                }
                for (uint32_t f = 0; f < pd.frames && !done4; ++f) {
                    uint8_t* dst = o + (size_t)f * dc.num_channels * dc.bps;
                    /* the register-packed form for frames of whole dwords, as alac_interleave picks it */
#define LANE_IL_CASE(NC_, BPS_)                                                                               \
    case (NC_) * 8 + (BPS_): {                                                                                \
        uint32_t fr[(NC_) * (BPS_) / 4];                                                                      \
        alac::interleave_frame_packed<NC_, BPS_>(dc, p, sizes[i], avail, pd, cd, rows.data(), rs, f, fr);     \
        memcpy(dst, fr, sizeof(fr));                                                                          \
        break;                                                                                                \
    }
                    switch (dc.num_channels * 8u + dc.bps) {
                        LANE_IL_CASE(4, 2) LANE_IL_CASE(6, 2) LANE_IL_CASE(8, 2) LANE_IL_CASE(4, 3) LANE_IL_CASE(8, 3)
                        LANE_IL_CASE(3, 4) LANE_IL_CASE(4, 4) LANE_IL_CASE(5, 4) LANE_IL_CASE(6, 4) LANE_IL_CASE(7, 4) LANE_IL_CASE(8, 4)
                        default: alac::interleave_frame(dc, p, sizes[i], avail, pd, cd, rows.data(), rs, f, dst);
                    }
#undef LANE_IL_CASE
                }
                continue;
            }
            /* ROUTE_LEGACY: fall through to the whole-packet decoder */
        }
        const uint32_t cls = variant >= 0 ? (uint32_t)variant : 3u;
        if (classes_out && variant >= 0) classes_out[i] = 1024u + cls;
        switch (cls) { /* register-tap widths of the whole-packet decoder: 4, 6, 8, or 16 with the int16 wrap */
            case 0: status[i] = alac::decode_wave<HostWave, 4, false>(wv, dc, true, p, sizes[i], avail, o, &frames_out[i]); break;
            case 1: status[i] = alac::decode_wave<HostWave, 6, false>(wv, dc, true, p, sizes[i], avail, o, &frames_out[i]); break;
            case 2: status[i] = alac::decode_wave<HostWave, 8, false>(wv, dc, true, p, sizes[i], avail, o, &frames_out[i]); break;
            default: status[i] = alac::decode_wave<HostWave, 16, true>(wv, dc, true, p, sizes[i], avail, o, &frames_out[i]); break;
        }
    }
    if (region) munmap(region, region_len);
    return 0;
}
