// sharded_decoder.hpp — DecodePackets over several GPUs of one node from one process (north star: "hipSetDevice per
// slice, no collectives"): a PacketDecoder per device, a host thread per slice, packets split by contiguous index
// range (GPU g of G gets [g*ceil(n/G), min(n, (g+1)*ceil(n/G))) — the rule of parallel.py's shard_range and of
// bench.py's ranks). Nothing is exchanged between devices: packets are independent (decoder.go:283,298-300,433-465)
// and PCM slot i is out + i*out_stride.
//
// Every handle owns its device, its three streams, its workspace and its staging (include/alacgpu.h: handles are
// re-entrant against each other; the error text of a failed call is thread-local), so the slices run concurrently.
// The same device may appear more than once in `devices` (two handles on one GPU): that is how the threading is
// tested on a one-GPU box. Header-only; link with -lalacgpu.
#pragma once

#include <algorithm>
#include <exception>
#include <thread>
#include <utility>

#include "packet_decoder.hpp"

namespace alac {

// packets [lo, hi) of slice g of G
inline std::pair<size_t, size_t> ShardRange(size_t n, size_t world, size_t rank) {
    if (world == 0 || rank >= world) throw std::invalid_argument("bad world/rank");
    const size_t per = (n + world - 1) / world;
    const size_t lo = std::min(n, rank * per);
    return {lo, std::min(n, lo + per)};
}

class ShardedDecoder {
public:
    ShardedDecoder(const PacketConfig& config, const std::vector<int>& devices) {
        if (devices.empty()) throw std::invalid_argument("no devices");
        for (int d : devices) decoders_.push_back(NewPacketDecoder(config, d));
    }

    size_t world() const { return decoders_.size(); }
    PCMFormat Format() const { return decoders_[0]->Format(); }

    // Same contract as PacketDecoder::DecodePackets; slice g is decoded by decoder g on its own thread.
    void DecodePackets(const uint8_t* blob, size_t blob_bytes, const uint64_t* offsets, size_t n, uint8_t* out,
                       size_t out_stride, uint32_t* frames, int32_t* status) {
        const size_t G = decoders_.size();
        std::vector<std::exception_ptr> errs(G);
        std::vector<std::thread> threads;
        for (size_t g = 0; g < G; ++g) {
            threads.emplace_back([&, g] {
                try {
                    const auto r = ShardRange(n, G, g);
                    if (r.first == r.second) return;
                    // every slice names its packets in the caller's blob (the host entry uploads only the span of the
                    // blob its packets cover, and checks every descriptor against blob_bytes)
                    decoders_[g]->DecodePackets(blob, blob_bytes, offsets + r.first, r.second - r.first, out + r.first * out_stride,
                                                out_stride, frames + r.first, status + r.first);
                } catch (...) {
                    errs[g] = std::current_exception();
                }
            });
        }
        for (auto& t : threads) t.join();
        for (auto& e : errs)
            if (e) std::rethrow_exception(e);
    }

    // the decoders' shard-sized workspaces are nothing a later small decoder would want to inherit from the pool
    ~ShardedDecoder() {
        decoders_.clear();
        PacketDecoder::Trim();
    }

private:
    std::vector<std::unique_ptr<PacketDecoder>> decoders_;
};

}  // namespace alac
