/*
 * alacgpu.hip — gfx950 batch ALAC decode kernel + the C ABI of include/alacgpu.h.
 *
 * Replaces the reference's per-packet hot path (decoder.go:133-207 -> internal/alac golomb.go,
 * predictor.go, matrix.go) with one HIP kernel over a batch of independent packets. There is no
 * host decode path in this library: every decode entry launches the kernel.
 *
 * Kernel shape: one 64-lane wavefront per 64 packets, one workgroup per wavefront
 * (no cross-lane traffic, so nothing to share in LDS); lanes run the state machine of alac_lane.h
 * in lock step. HBM traffic per packet: compressed bytes in, PCM bytes out, plus the U-channel
 * hand-off tile (frame_length x 64 x int32 per wave, row-coalesced, written once and read once).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#define ALAC_DEV __device__ __forceinline__
#include "alac_lane.h"

namespace {

constexpr uint32_t kWave = 64;
constexpr uint32_t kTimingSlots = 64;

__global__ void __launch_bounds__(kWave)
alac_decode_lanes(alac::DevCfg cfg, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                  const uint32_t* __restrict__ sizes, uint32_t n_packets, uint8_t* __restrict__ out,
                  uint64_t out_stride, uint32_t* __restrict__ frames_out, int32_t* __restrict__ status,
                  int32_t* __restrict__ scratch) {
    const uint32_t lane = threadIdx.x;
    const uint64_t pkt = (uint64_t)blockIdx.x * kWave + lane;
    if (pkt >= n_packets) return;
    int32_t* scr = scratch + (uint64_t)blockIdx.x * cfg.frame_length * kWave + lane;
    uint32_t frames = 0;
    const int32_t st = alac::decode_lane<kWave>(cfg, blob + offsets[pkt], sizes[pkt], out + pkt * out_stride, scr,
                                                &frames);
    frames_out[pkt] = frames;
    status[pkt] = st;
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

struct alacgpu_decoder {
    alacgpu_config cfg;
    alac::DevCfg dev_cfg;
    int device;
    size_t frame_bytes;
    hipStream_t stream;
    hipEvent_t ev_start[kTimingSlots], ev_stop[kTimingSlots]; /* ring of per-launch event pairs */
    uint64_t launches;                                       /* since the last timing reset */
    DevBuf scratch;                                  /* U hand-off tiles */
    DevBuf d_blob, d_offsets, d_sizes, d_out, d_frames, d_status; /* host-entry staging */
    HostBuf h_blob, h_meta;
};

namespace {

int launch(alacgpu_decoder* dec, const uint8_t* d_blob, const uint64_t* d_offsets, const uint32_t* d_sizes,
           size_t n, uint8_t* d_out, size_t out_stride, uint32_t* d_frames, int32_t* d_status) {
    if (n == 0) return ALACGPU_E_OK;
    if (n > 0x7fffffffu) {
        set_err("batch too large");
        return ALACGPU_E_ARG;
    }
    const uint32_t waves = (uint32_t)((n + kWave - 1) / kWave);
    int rc = dec->scratch.ensure((size_t)waves * dec->cfg.frame_length * kWave * sizeof(int32_t));
    if (rc) return rc;
    alac::DevCfg c = dec->dev_cfg;
    c.fast16s = (dec->cfg.bit_depth == 16 && dec->cfg.num_channels == 2 && out_stride % 16 == 0 &&
                 (reinterpret_cast<uintptr_t>(d_out) % 16) == 0)
                    ? 1u
                    : 0u;
    const uint32_t slot = (uint32_t)(dec->launches % kTimingSlots);
    HIP_TRY(hipEventRecord(dec->ev_start[slot], dec->stream));
    hipLaunchKernelGGL(alac_decode_lanes, dim3(waves), dim3(kWave), 0, dec->stream, c, d_blob, d_offsets, d_sizes,
                       (uint32_t)n, d_out, (uint64_t)out_stride, d_frames, d_status, (int32_t*)dec->scratch.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(dec->ev_stop[slot], dec->stream));
    dec->launches++;
    return ALACGPU_E_OK;
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
    hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    for (uint32_t i = 0; i < kTimingSlots && e == hipSuccess; i++) {
        e = hipEventCreate(&d->ev_start[i]);
        if (e == hipSuccess) e = hipEventCreate(&d->ev_stop[i]);
    }
    if (e != hipSuccess) {
        set_err("stream/event creation failed: %s", hipGetErrorString(e));
        delete d;
        return ALACGPU_E_HIP;
    }
    *out = d;
    return ALACGPU_E_OK;
}

void alacgpu_destroy(alacgpu_decoder* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipStreamSynchronize(d->stream);
    d->scratch.release();
    d->d_blob.release();
    d->d_offsets.release();
    d->d_sizes.release();
    d->d_out.release();
    d->d_frames.release();
    d->d_status.release();
    d->h_blob.release();
    d->h_meta.release();
    for (uint32_t i = 0; i < kTimingSlots; i++) {
        (void)hipEventDestroy(d->ev_start[i]);
        (void)hipEventDestroy(d->ev_stop[i]);
    }
    (void)hipStreamDestroy(d->stream);
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
    const size_t waves = (n + kWave - 1) / kWave;
    return d->scratch.ensure(waves * d->cfg.frame_length * kWave * sizeof(int32_t));
}

int alacgpu_decode_batch_device(alacgpu_decoder* d, const uint8_t* d_blob, const uint64_t* d_offsets,
                                const uint32_t* d_sizes, size_t n, uint8_t* d_out, size_t out_stride,
                                uint32_t* d_frames, int32_t* d_status, int sync) {
    if (!d || (n && (!d_blob || !d_offsets || !d_sizes || !d_out || !d_frames || !d_status))) {
        set_err("null argument");
        return ALACGPU_E_ARG;
    }
    if (out_stride < d->frame_bytes) {
        set_err("out_stride %zu < frame bytes %zu", out_stride, d->frame_bytes);
        return ALACGPU_E_ARG;
    }
    HIP_TRY(hipSetDevice(d->device));
    int rc = launch(d, d_blob, d_offsets, d_sizes, n, d_out, out_stride, d_frames, d_status);
    if (rc) return rc;
    if (sync) HIP_TRY(hipStreamSynchronize(d->stream));
    return ALACGPU_E_OK;
}

int alacgpu_decode_batch(alacgpu_decoder* d, const uint8_t* blob, const uint64_t* offsets, size_t n, uint8_t* out,
                         size_t out_stride, uint32_t* frames_out, int32_t* status) {
    if (!d || (n && (!blob || !offsets || !out || !frames_out || !status))) {
        set_err("null argument");
        return ALACGPU_E_ARG;
    }
    if (out_stride < d->frame_bytes) {
        set_err("out_stride %zu < frame bytes %zu", out_stride, d->frame_bytes);
        return ALACGPU_E_ARG;
    }
    if (n == 0) return ALACGPU_E_OK;
    HIP_TRY(hipSetDevice(d->device));

    /* re-pack into the device blob layout: 16-byte aligned packets, ALACGPU_PACKET_PAD zero bytes after each */
    int rc = d->h_meta.ensure(n * (sizeof(uint64_t) + sizeof(uint32_t)));
    if (rc) return rc;
    uint64_t* h_off = (uint64_t*)d->h_meta.p;
    uint32_t* h_sz = (uint32_t*)(h_off + n);
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0x0fffffffull) {
            set_err("bad offsets at packet %zu", i);
            return ALACGPU_E_ARG;
        }
        const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
        h_off[i] = total;
        h_sz[i] = (uint32_t)len;
        total = (total + len + ALACGPU_PACKET_PAD + 15u) & ~(size_t)15u;
    }
    total += 64;
    if ((rc = d->h_blob.ensure(total))) return rc;
    uint8_t* hb = (uint8_t*)d->h_blob.p;
    for (size_t i = 0; i < n; i++) {
        const size_t end = (i + 1 < n ? (size_t)h_off[i + 1] : total);
        memcpy(hb + h_off[i], blob + offsets[i], h_sz[i]);
        memset(hb + h_off[i] + h_sz[i], 0, end - (size_t)h_off[i] - h_sz[i]);
    }
    /* keep the device output rows 16-byte aligned so the wide-store path is taken */
    const size_t d_stride = (d->frame_bytes + 15u) & ~(size_t)15u;
    if ((rc = d->d_blob.ensure(total))) return rc;
    if ((rc = d->d_offsets.ensure(n * sizeof(uint64_t)))) return rc;
    if ((rc = d->d_sizes.ensure(n * sizeof(uint32_t)))) return rc;
    if ((rc = d->d_out.ensure(n * d_stride))) return rc;
    if ((rc = d->d_frames.ensure(n * sizeof(uint32_t)))) return rc;
    if ((rc = d->d_status.ensure(n * sizeof(int32_t)))) return rc;

    HIP_TRY(hipMemcpyAsync(d->d_blob.p, hb, total, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(d->d_offsets.p, h_off, n * sizeof(uint64_t), hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(d->d_sizes.p, h_sz, n * sizeof(uint32_t), hipMemcpyHostToDevice, d->stream));
    rc = launch(d, (const uint8_t*)d->d_blob.p, (const uint64_t*)d->d_offsets.p, (const uint32_t*)d->d_sizes.p, n,
                (uint8_t*)d->d_out.p, d_stride, (uint32_t*)d->d_frames.p, (int32_t*)d->d_status.p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(frames_out, d->d_frames.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync(status, d->d_status.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpy2DAsync(out, out_stride, d->d_out.p, d_stride, d->frame_bytes, n, hipMemcpyDeviceToHost,
                             d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return ALACGPU_E_OK;
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

void* alacgpu_stream(alacgpu_decoder* d) { return d ? (void*)d->stream : nullptr; }

int alacgpu_synchronize(alacgpu_decoder* d) {
    if (!d) return ALACGPU_E_ARG;
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return ALACGPU_E_OK;
}

const char* alacgpu_last_error(void) { return g_err; }

const char* alacgpu_version(void) { return "alacgpu 0.1.0 gfx950"; }

} /* extern "C" */
