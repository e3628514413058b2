/*
 * lane_sim.cpp — TEST-ONLY host build of the kernel's per-lane state machine.
 *
 * Compiles saprobe-alac_amd/csrc/alac_lane.h (the exact text the gfx950 kernel is built from) with
 * g++ and runs it one "lane" at a time, so the decode LOGIC can be compared with the oracle in the
 * CPU test-suite (-m "not gpu"), where no GPU exists. It lives under tests/, is never linked into
 * libalacgpu.so and is not a decode path of the product.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define ALAC_DEV inline
#include "../../saprobe-alac_amd/csrc/alac_lane.h"

extern "C" int lane_sim_decode_batch(const alacgpu_config* cfg, const uint8_t* blob, const uint64_t* offsets,
                                     const uint32_t* sizes, size_t n, uint8_t* out, size_t out_stride,
                                     uint32_t* frames_out, int32_t* status, int poison) {
    alac::DevCfg dc{};
    dc.frame_length = cfg->frame_length;
    dc.bit_depth = cfg->bit_depth;
    dc.num_channels = cfg->num_channels;
    dc.pb = cfg->pb;
    dc.mb = cfg->mb;
    dc.kb = cfg->kb;
    dc.bps = cfg->bit_depth == 16 ? 2 : cfg->bit_depth == 32 ? 4 : 3;
    dc.fast16s = (cfg->bit_depth == 16 && cfg->num_channels == 2 && out_stride % 16 == 0 &&
                  (reinterpret_cast<uintptr_t>(out) % 16) == 0)
                     ? 1u
                     : 0u;
    std::vector<int32_t> scr(cfg->frame_length ? cfg->frame_length : 1);
    for (size_t i = 0; i < n; i++) {
        /* the kernel never relies on scratch or output contents: poison them */
        if (poison) {
            memset(scr.data(), 0x5a, scr.size() * sizeof(int32_t));
            memset(out + i * out_stride, 0xa5, out_stride);
        }
        status[i] = alac::decode_lane<1>(dc, blob + offsets[i], sizes[i], out + i * out_stride, scr.data(),
                                         &frames_out[i]);
    }
    return 0;
}
