// packet_decoder.hpp — C++ host-side mirror of the reference's packet layer over the C ABI (include/alacgpu.h).
//
// Same names, argument meaning and error behaviour as mycophonic/saprobe-alac's root package:
//   PacketConfig / PCMFormat            config.go:27-38, format.go:20-24
//   NewPacketDecoder(config)            decoder.go:90     -> throws ErrConfig
//   PacketDecoder::DecodePacket(packet) decoder.go:117    -> returns PCM bytes, throws ErrDecode
//   PacketDecoder::DecodePackets(...)   new batch entry of the north star
//   PacketDecoder::Format()             decoder.go:112
// Header-only; link with -lalacgpu. Every decode call runs the HIP kernels: there is no CPU path.
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/alacgpu.h"

namespace alac {

using PacketConfig = ::alacgpu_config;
using PCMFormat = ::alacgpu_format;

struct ErrConfig : std::runtime_error {  // errors.go:25
    using std::runtime_error::runtime_error;
};
struct ErrDecode : std::runtime_error {  // errors.go:33; `status` is the C ABI status word
    int32_t status;
    ErrDecode(int32_t st, const std::string& what) : std::runtime_error(what), status(st) {}
    int code() const { return ALACGPU_STATUS_CODE(status); }  // which internal sentinel (internal/alac/errors.go)
};

inline std::string StatusText(int32_t st) {  // the reference's error chain, decoder.go:144-189,303,468,482
    static const char* ctx[] = {nullptr, "SCE/LFE", "CPE", "DSE", "FIL"};
    static const char* stage[] = {nullptr, "entropy decode", "entropy decode U", "entropy decode V"};
    static const char* code[] = {"ok", "alac: bitstream overrun", "alac: sample count exceeds buffer",
                                 "alac: invalid frame header", "alac: invalid bytesShifted value",
                                 "alac: unsupported element type (CCE/PCE)",
                                 "alac: malformed packet (the reference panics)", "alac: packet outside the blob"};
    std::string s = "decode failed";
    const int c = ALACGPU_STATUS_CTX(st), g = ALACGPU_STATUS_STAGE(st), k = ALACGPU_STATUS_CODE(st);
    if (c > 0 && c < 5) s += std::string(": ") + ctx[c];
    if (g > 0) s += std::string(": ") + stage[g];
    s += std::string(": ") + (k < 8 ? code[k] : "alac: unknown status");
    return s;
}

class PacketDecoder {
public:
    PacketDecoder(const PacketConfig& config, int device = 0) {
        alacgpu_decoder* h = nullptr;
        const int rc = alacgpu_create(&config, device, &h);
        if (rc == ALACGPU_E_CONFIG) throw ErrConfig(alacgpu_last_error());
        if (rc != ALACGPU_E_OK) throw std::runtime_error(alacgpu_last_error());
        h_.reset(h);
    }

    PCMFormat Format() const {
        PCMFormat f{};
        alacgpu_get_format(h_.get(), &f);
        return f;
    }

    // decoder.go:117: a fresh buffer of FrameLength*ch*bps, trimmed to numSamples*ch*bps
    std::vector<uint8_t> DecodePacket(const uint8_t* packet, size_t len) {
        std::vector<uint8_t> out(alacgpu_frame_bytes(h_.get()));
        size_t n = 0;
        int32_t st = 0;
        const int rc = alacgpu_decode_packet(h_.get(), packet, len, out.data(), out.size(), &n, &st);
        if (rc == ALACGPU_E_DECODE) throw ErrDecode(st, StatusText(st));
        if (rc != ALACGPU_E_OK) throw std::runtime_error(alacgpu_last_error());
        out.resize(n);
        return out;
    }

    // Batch entry: packet i = blob[offsets[i], offsets[i+1]) of the blob_bytes readable bytes at blob; PCM i at
    // out + i*out_stride. Per-packet failures are reported in status[i] (frames[i] = 0) and do not affect the other
    // packets; a packet whose offsets leave the blob is never read (ALACGPU_ERR_RANGE).
    void DecodePackets(const uint8_t* blob, size_t blob_bytes, const uint64_t* offsets, size_t n, uint8_t* out,
                       size_t out_stride, uint32_t* frames, int32_t* status) {
        if (alacgpu_decode_batch(h_.get(), blob, blob_bytes, offsets, n, out, out_stride, frames, status) != ALACGPU_E_OK)
            throw std::runtime_error(alacgpu_last_error());
    }

    alacgpu_decoder* handle() const { return h_.get(); }
    // alacgpu_trim(): destroyed decoders leave streams, events and small buffers (<= 2 GB of device memory and
    // 64 MB of pinned memory each, four per device) in a per-process pool for the next one; this frees them
    static void Trim() { alacgpu_trim(); }

private:
    struct Del {
        void operator()(alacgpu_decoder* d) const { alacgpu_destroy(d); }
    };
    std::unique_ptr<alacgpu_decoder, Del> h_;
};

inline std::unique_ptr<PacketDecoder> NewPacketDecoder(const PacketConfig& config, int device = 0) {
    return std::make_unique<PacketDecoder>(config, device);
}

}  // namespace alac
