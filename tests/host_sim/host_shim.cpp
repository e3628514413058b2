/*
 * host_shim.cpp — TEST-ONLY C entry points over the C++ host mirrors (the .hpp files of saprobe-alac_amd/host), so the Python
 * test-suite can drive them with ctypes: the MP4 demuxer (CPU) and the streaming Decoder (GPU, links libalacgpu).
 */
#include <cstdint>
#include <cstring>

#ifdef SHIM_WITH_DECODER
#include "../../saprobe-alac_amd/host/sharded_decoder.hpp"
#include "../../saprobe-alac_amd/host/stream_decoder.hpp"
#else
#include "../../saprobe-alac_amd/host/mp4_demux.hpp"
#endif

extern "C" {

/* returns the sample count, or -(1 + sentinel) */
long demux_track(const uint8_t* data, size_t len, uint8_t* cookie, size_t cookie_cap, size_t* cookie_len,
                 uint64_t* offsets, uint32_t* sizes, size_t cap) {
    try {
        const alac::mp4::Track t = alac::mp4::FindALACTrack(data, len);
        *cookie_len = t.cookie.size();
        memcpy(cookie, t.cookie.data(), t.cookie.size() < cookie_cap ? t.cookie.size() : cookie_cap);
        const size_t n = t.sizes.size() < cap ? t.sizes.size() : cap;
        memcpy(offsets, t.offsets.data(), n * sizeof(uint64_t));
        memcpy(sizes, t.sizes.data(), n * sizeof(uint32_t));
        return (long)t.sizes.size();
    } catch (const alac::mp4::Error& e) {
        return -(1 + (long)e.sentinel);
    }
}

#ifdef SHIM_WITH_DECODER
static thread_local char g_msg[256];
const char* shim_last_error() { return g_msg; }

/* kind: 1 ErrNoTrack, 2 ErrConfig, 3 ErrDecode, 4 ErrRead, 5 other */
static int fail(int kind, const char* what) {
    strncpy(g_msg, what, sizeof(g_msg) - 1);
    return -kind;
}
#define SHIM_TRY(stmt)                                                          \
    try {                                                                       \
        stmt;                                                                   \
    } catch (const alac::ErrNoTrack& e) { return fail(1, e.what());             \
    } catch (const alac::ErrConfig& e) { return fail(2, e.what());              \
    } catch (const alac::ErrDecode& e) { return fail(3, e.what());              \
    } catch (const alac::ErrRead& e) { return fail(4, e.what());                \
    } catch (const std::exception& e) { return fail(5, e.what()); }

long shim_open(const uint8_t* file, size_t len, size_t window, void** out) {
    SHIM_TRY(*out = alac::NewDecoder(file, len, 0, window).release());
    return 0;
}
void shim_close(void* d) { delete static_cast<alac::Decoder*>(d); }
long shim_read(void* d, uint8_t* p, size_t n) {
    SHIM_TRY(return (long)static_cast<alac::Decoder*>(d)->Read(p, n));
}
long long shim_seek(void* d, long long ns) { return static_cast<alac::Decoder*>(d)->Seek(ns); }
long long shim_duration(void* d) { return static_cast<alac::Decoder*>(d)->Duration(); }
long long shim_position(void* d) { return static_cast<alac::Decoder*>(d)->Position(); }

/* host/sharded_decoder.hpp: one handle + one host thread per entry of devices[] */
long shim_sharded_decode(const alacgpu_config* cfg, const int* devices, size_t n_devices, const uint8_t* blob,
                         size_t blob_bytes, const uint64_t* offsets, size_t n, uint8_t* out, size_t out_stride, uint32_t* frames,
                         int32_t* status) {
    SHIM_TRY({
        alac::ShardedDecoder dec(*cfg, std::vector<int>(devices, devices + n_devices));
        dec.DecodePackets(blob, blob_bytes, offsets, n, out, out_stride, frames, status);
    });
    return 0;
}
#endif
}
