// stream_decoder.hpp — C++ host-side streaming façade over the batch path: the reference's
// `Decoder` (decode.go:32-190) and ParseMagicCookie (config.go:47-81).
//
// Same surface (NewDecoder, Format, Duration, Position, Seek, Read) and the same observable behaviour: PCM bytes
// in packet order, packet-aligned seeks, the error of packet k when the reader reaches packet k and again on every
// later Read. The PCM is made differently: a window of packets goes through ONE batch decode on the GPU
// (alacgpu_decode_batch) and Read / Seek are served from the decoded window — and while the caller drains window k, a
// worker thread has window k + 1 decoded into a second buffer (round 4: a 300-second file spent as long in Read's copies
// as in the decode; now the two overlap). Window PCM lives in pinned memory (alacgpu_host_alloc): the device writes it
// there directly. Header-only; link with -lalacgpu -pthread.
#pragma once

#include <algorithm>
#include <cstring>
#include <future>

#include "mp4_demux.hpp"
#include "packet_decoder.hpp"

namespace alac {

struct ErrNoTrack : std::runtime_error {  // errors.go:29; `sentinel` is the internal/mp4 one
    mp4::Sentinel sentinel;
    explicit ErrNoTrack(const mp4::Error& e) : std::runtime_error(std::string("no track found: ") + e.what()), sentinel(e.sentinel) {}
};
struct ErrRead : std::runtime_error {  // decode.go:163-169: a sample that lies outside the file
    using std::runtime_error::runtime_error;
};

// ParseMagicCookie (config.go:47-81): 24-byte ALACSpecificConfig, optional 'frma' / 'alac' wrappers.
inline PacketConfig ParseMagicCookie(const uint8_t* d, size_t n) {
    if (n >= 12 && memcmp(d + 4, "frma", 4) == 0) d += 12, n -= 12;
    if (n >= 12 && memcmp(d + 4, "alac", 4) == 0) d += 12, n -= 12;
    if (n < 24) throw ErrConfig("invalid configuration: alac: invalid magic cookie");
    if (d[4] > 0) throw ErrConfig("invalid configuration: alac: unsupported version: " + std::to_string(d[4]));
    auto be32 = [&](size_t at) { return (uint32_t)d[at] << 24 | (uint32_t)d[at + 1] << 16 | (uint32_t)d[at + 2] << 8 | d[at + 3]; };
    PacketConfig c{};
    c.frame_length = be32(0);
    c.bit_depth = d[5];
    c.pb = d[6];
    c.mb = d[7];
    c.kb = d[8];
    c.num_channels = d[9];
    c.max_run = (uint16_t)(d[10] << 8 | d[11]);
    c.max_frame_bytes = be32(12);
    c.avg_bit_rate = be32(16);
    c.sample_rate = be32(20);
    return c;
}

class Decoder {
public:
    // The file must stay mapped / alive for the lifetime of the decoder. window = packets per batch decode (the read-ahead
    // runs one window ahead of the reader; 1024 packets are 16 MB of CD audio).
    Decoder(const uint8_t* file, size_t len, int device = 0, size_t window = 1024) : file_(file), len_(len), window_(std::max<size_t>(1, window)) {
        try {
            track_ = mp4::FindALACTrack(file, len);
        } catch (const mp4::Error& e) {
            throw ErrNoTrack(e);  // decode.go:52-54
        }
        try {
            config_ = ParseMagicCookie(track_.cookie.data(), track_.cookie.size());
        } catch (const ErrConfig& e) {
            throw ErrConfig(std::string("parsing ALAC config: ") + e.what());  // decode.go:57-59
        }
        dec_ = NewPacketDecoder(config_, device);
        stride_ = alacgpu_frame_bytes(dec_->handle());
        const unsigned bps = config_.bit_depth == 16 ? 2 : config_.bit_depth == 32 ? 4 : 3;
        bpf_ = (size_t)config_.num_channels * bps;
    }
    ~Decoder() {
        Settle();
        for (Window& w : win_) w.Free();
    }
    Decoder(const Decoder&) = delete;
    Decoder& operator=(const Decoder&) = delete;

    PCMFormat Format() const { return dec_->Format(); }
    const PacketConfig& Config() const { return config_; }
    size_t Packets() const { return track_.sizes.size(); }
    // nanoseconds, as time.Duration (decode.go:82-98)
    int64_t Duration() const { return (int64_t)Packets() * config_.frame_length * 1000000000ll / config_.sample_rate; }
    int64_t Position() const { return (int64_t)idx_ * config_.frame_length * 1000000000ll / config_.sample_rate; }

    // decode.go:103-124: packet-aligned; returns the position reached
    int64_t Seek(int64_t ns) {
        const int64_t frame = (int64_t)((double)ns / 1e9 * (double)config_.sample_rate);
        int64_t target = frame / (int64_t)config_.frame_length;
        target = std::max<int64_t>(0, std::min<int64_t>(target, (int64_t)Packets()));
        idx_ = (size_t)target;
        buf_off_ = buf_len_ = 0;
        eof_ = idx_ >= Packets();
        return Position();
    }

    // decode.go:126-190. Returns the bytes read (0 = end of stream). An error is thrown only when nothing was read
    // before it in this call: data first, the error on the next call.
    size_t Read(uint8_t* p, size_t n) {
        size_t total = 0;
        while (total < n) {
            if (buf_off_ < buf_len_) {
                const size_t take = std::min(n - total, buf_len_ - buf_off_);
                memcpy(p + total, buf_ + buf_off_, take);
                buf_off_ += take;
                total += take;
                continue;
            }
            if (eof_ || idx_ >= Packets()) {
                eof_ = true;
                break;
            }
            try {
                NextPacket();
            } catch (...) {
                if (total) break;
                throw;
            }
        }
        return total;
    }

private:
    // one decoded window: packets [w0, w1); `lost`: the first sample behind it that lies outside the file (decode.go:163-169)
    struct Window {
        size_t w0 = 0, w1 = 0, lost = SIZE_MAX;
        // ONE pinned block (when the runtime grants it): [PCM n x stride | frames n x u32 | status n x i32]. The library
        // transfers into caller memory in place only when all three are pinned.
        uint8_t* block = nullptr;
        size_t cap = 0;
        bool pinned = false;
        uint8_t* out = nullptr;
        uint32_t* frames = nullptr;
        int32_t* status = nullptr;
        std::vector<uint64_t> starts;
        std::vector<uint8_t> gather;
        bool Holds(size_t k) const { return (w0 <= k && k < w1) || lost == k; }
        void Reserve(size_t n, size_t stride) {
            const size_t pcm = (n * stride + 15u) & ~(size_t)15u, bytes = pcm + n * 8u;
            if (bytes > cap) {
                Free();
                block = static_cast<uint8_t*>(alacgpu_host_alloc(bytes));
                pinned = block != nullptr;
                if (!block) block = static_cast<uint8_t*>(::operator new(bytes));  // pageable: the library stages the copies
                cap = bytes;
            }
            out = block;
            frames = reinterpret_cast<uint32_t*>(block + pcm);
            status = reinterpret_cast<int32_t*>(block + pcm + n * 4u);
        }
        void Free() {
            if (block && pinned) alacgpu_host_free(block);
            else if (block) ::operator delete(block);
            block = nullptr;
            cap = 0;
        }
    };

    // packets [first, first + window) into w: one batch decode (runs on the caller's thread or on the read-ahead worker:
    // never both at once, the handle is not safe for concurrent use)
    void DecodeWindow(Window& w, size_t first) {
        size_t last = std::min(first + window_, Packets());
        w.lost = SIZE_MAX;
        for (size_t k = first; k < last; ++k)
            if (track_.offsets[k] > len_ || track_.sizes[k] > len_ - track_.offsets[k]) {
                w.lost = k;
                last = k;
                break;
            }
        const size_t n = last - first;
        w.starts.assign(n + 1, 0);
        for (size_t k = 0; k < n; ++k) w.starts[k + 1] = w.starts[k] + track_.sizes[first + k];
        bool run = true;
        for (size_t k = 1; k < n && run; ++k) run = track_.offsets[first + k] == track_.offsets[first + k - 1] + track_.sizes[first + k - 1];
        const uint8_t* blob = n ? file_ + track_.offsets[first] : nullptr;  // one mdat run: no gather
        size_t blob_bytes = n ? (size_t)(len_ - track_.offsets[first]) : 0;
        if (!run) {
            w.gather.resize(w.starts[n] + 1);
            for (size_t k = 0; k < n; ++k) memcpy(w.gather.data() + w.starts[k], file_ + track_.offsets[first + k], track_.sizes[first + k]);
            blob = w.gather.data();
            blob_bytes = (size_t)w.starts[n];
        }
        if (n) {
            static const uint8_t none = 0;
            if (w.starts[n] == 0) {
                blob = &none;
                blob_bytes = 0;
            }
            w.Reserve(n, stride_);
            dec_->DecodePackets(blob, blob_bytes, w.starts.data(), n, w.out, stride_, w.frames, w.status);
        }
        w.w0 = first;
        w.w1 = last;
    }
    // waits for the read-ahead (if any); its window becomes valid or its error is dropped (the reader will meet it again)
    void Settle() {
        if (!ahead_.valid()) return;
        try {
            ahead_.get();
            ahead_ok_ = true;
        } catch (...) {
            ahead_ok_ = false;
        }
    }
    void NextPacket() {
        const size_t k = idx_;
        if (!win_[cur_].Holds(k)) {
            Settle();
            Window& other = win_[cur_ ^ 1u];
            if (ahead_ok_ && other.Holds(k)) {
                cur_ ^= 1u;
            } else {  // the first window, a seek, or a read-ahead that failed: decode here (and let its error out)
                DecodeWindow(win_[cur_], k);
            }
            ahead_ok_ = false;
            // the window behind this one, while the caller drains this one
            Window& c = win_[cur_];
            if (c.lost == SIZE_MAX && c.w1 < Packets() && c.w1 > c.w0) {
                Window* nxt = &win_[cur_ ^ 1u];
                const size_t first = c.w1;
                ahead_ = std::async(std::launch::async, [this, nxt, first] { DecodeWindow(*nxt, first); });
            }
        }
        Window& w = win_[cur_];
        if (w.lost == k) throw ErrRead("reading sample " + std::to_string(k) + ": unexpected EOF");
        const size_t j = k - w.w0;
        if (w.status[j]) throw ErrDecode(w.status[j], "decoding packet " + std::to_string(k) + ": " + StatusText(w.status[j]));
        buf_ = w.out + j * stride_;
        buf_len_ = (size_t)w.frames[j] * bpf_;
        buf_off_ = 0;
        ++idx_;
    }

    const uint8_t* file_;
    size_t len_, window_;
    mp4::Track track_;
    PacketConfig config_{};
    std::unique_ptr<PacketDecoder> dec_;
    size_t stride_ = 0, bpf_ = 0;
    size_t idx_ = 0;
    const uint8_t* buf_ = nullptr;
    size_t buf_off_ = 0, buf_len_ = 0;
    bool eof_ = false;
    Window win_[2];
    unsigned cur_ = 0;
    std::future<void> ahead_;  // the read-ahead of win_[cur_ ^ 1]
    bool ahead_ok_ = false;
};

inline std::unique_ptr<Decoder> NewDecoder(const uint8_t* file, size_t len, int device = 0, size_t window = 1024) {
    return std::make_unique<Decoder>(file, len, device, window);
}

}  // namespace alac
