// stream_decoder.hpp — C++ host-side streaming façade over the batch path: the reference's
// `Decoder` (decode.go:32-190) and ParseMagicCookie (config.go:47-81).
//
// Same surface (NewDecoder, Format, Duration, Position, Seek, Read) and the same observable behaviour: PCM bytes
// in packet order, packet-aligned seeks, the error of packet k when the reader reaches packet k and again on every
// later Read. The PCM is made differently: a window of packets goes through ONE batch decode on the GPU
// (alacgpu_decode_batch) and Read / Seek are served from the decoded window — and while the caller drains window k, worker
// threads have windows k + 1 and k + 2 decoded into buffers of their own, on TWO handles (round 4: a 300-second file spent
// as long in Read's copies as in the decode; a window's decode is a chain of staging copy, upload, a kernel that takes its
// 1.2 ms however few packets it holds, and download — two chains in flight fill each other's gaps, and the reader's copies
// run beside both). Every window has the same size, and both handles reserve their workspace for it when the file is
// opened: a buffer that grows in the middle of a stream is freed first, and hipFree waits for the whole device (a first
// window of a quarter of the size cost the windows behind it 4-5 ms each that way). Window PCM lives in pinned memory
// (alacgpu_host_alloc): the device writes it there directly. Header-only; link with -lalacgpu -pthread.
#pragma once

#include <algorithm>
#include <cstring>
#include <future>

#include "mp4_demux.hpp"
#include "packet_decoder.hpp"

#ifdef ALAC_STREAM_TRACE  // tools/r4_cppfile.py: where a file decode's time goes (stderr, microseconds since the first event)
#include <chrono>
#include <cstdio>
namespace alac { inline double TraceUs() { static const auto t0 = std::chrono::steady_clock::now(); return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); } }
#define ALAC_TRACE(...) (std::fprintf(stderr, "%10.1f ", alac::TraceUs()), std::fprintf(stderr, __VA_ARGS__), std::fputc('\n', stderr))
#else
#define ALAC_TRACE(...) ((void)0)
#endif

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
    // runs two windows ahead of the reader); 0: as many as make 48 MB of PCM (3 072 packets of CD audio, 2 048 of 96 kHz /
    // 24-bit stereo: a 300-second file of the latter takes 16.5 ms with windows of 1 024 packets and 11.6 ms with 2 048; three
    // such windows still fit the library's pool of pinned blocks, larger ones are allocated and freed per file: 86 ms).
    Decoder(const uint8_t* file, size_t len, int device = 0, size_t window = 0) : file_(file), len_(len), window_(window) {
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
        dec_[0] = NewPacketDecoder(config_, device);
        dec_[1] = NewPacketDecoder(config_, device);  // destroyed handles are pooled by the library: two per file cost nothing
        stride_ = alacgpu_frame_bytes(dec_[0]->handle());
        if (window_ == 0) window_ = std::max<size_t>(64, ((size_t)48 << 20) / std::max<size_t>(1, stride_));
        for (auto& d : dec_) (void)alacgpu_reserve(d->handle(), std::min(window_, Packets()));
        const unsigned bps = config_.bit_depth == 16 ? 2 : config_.bit_depth == 32 ? 4 : 3;
        bpf_ = (size_t)config_.num_channels * bps;
    }
    ~Decoder() {
        Drop();
        for (Window& w : win_) w.Free();
    }
    Decoder(const Decoder&) = delete;
    Decoder& operator=(const Decoder&) = delete;

    PCMFormat Format() const { return dec_[0]->Format(); }
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

    // packets [first, first + count) into w: one batch decode on handle `dec` (the caller's thread or a read-ahead worker;
    // a handle is never used by two of them at once)
    void DecodeWindow(Window& w, size_t first, size_t count, PacketDecoder& dec) {
        size_t last = std::min(first + count, Packets());
        w.lost = SIZE_MAX;
        w.w0 = w.w1 = first;
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
            ALAC_TRACE("window %zu +%zu: reserve", first, n);
            w.Reserve(n, stride_);
            ALAC_TRACE("window %zu +%zu: decode begins (pinned %d)", first, n, (int)w.pinned);
            dec.DecodePackets(blob, blob_bytes, w.starts.data(), n, w.out, stride_, w.frames, w.status);
            ALAC_TRACE("window %zu +%zu: decode done", first, n);
        }
        w.w1 = last;
    }
    // the read-aheads in flight, oldest first: window `slot` is being filled with the packets from `first` on
    struct Ahead {
        std::future<void> done;
        size_t first = 0, count = 0;
        unsigned slot = 0;
    };
    static constexpr unsigned kDepth = 2;
    // waits for the oldest read-ahead; true: its window is valid (false: its error is dropped, the reader will meet it again)
    bool Settle() {
        bool ok = true;
        try {
            ahead_[0].done.get();
        } catch (...) {
            ok = false;
        }
        for (unsigned i = 1; i < n_ahead_; ++i) ahead_[i - 1] = std::move(ahead_[i]);
        --n_ahead_;
        return ok;
    }
    void Drop() {
        while (n_ahead_) (void)Settle();
    }
    // keeps kDepth windows in flight behind the current one (consecutive windows take turns on the two handles, so the
    // two in flight never share one)
    void Schedule() {
        const Window& c = win_[cur_];
        if (c.lost != SIZE_MAX || c.w1 <= c.w0) return;  // the stream ends at the lost sample
        while (n_ahead_ < kDepth) {
            const size_t first = n_ahead_ ? ahead_[n_ahead_ - 1].first + ahead_[n_ahead_ - 1].count : c.w1;
            if (first >= Packets()) return;
            unsigned slot = 0;
            for (;; ++slot) {
                bool used = slot == cur_;
                for (unsigned i = 0; i < n_ahead_; ++i) used = used || ahead_[i].slot == slot;
                if (!used) break;
            }
            Ahead& a = ahead_[n_ahead_++];
            a.first = first;
            a.count = std::min(window_, Packets() - first);
            a.slot = slot;
            Window* w = &win_[slot];
            PacketDecoder* dec = dec_[seq_++ & 1u].get();
            const size_t count = a.count;
            a.done = std::async(std::launch::async, [this, w, first, count, dec] { DecodeWindow(*w, first, count, *dec); });
        }
    }
    void NextPacket() {
        const size_t k = idx_;
        if (!win_[cur_].Holds(k)) {
            bool have = false;
            if (n_ahead_ && ahead_[0].first == k) {  // the reader walked off the end of its window into the next one
                const unsigned slot = ahead_[0].slot;
                ALAC_TRACE("reader at %zu: waits", k);
                const bool ok = Settle();
                ALAC_TRACE("reader at %zu: has its window", k);
                if (ok && win_[slot].Holds(k)) {
                    cur_ = slot;
                    have = true;
                }
            }
            if (!have) {  // the first window, a seek, or a read-ahead that failed: decode here (and let its error out)
                Drop();
                seq_ = 1;
                DecodeWindow(win_[cur_], k, window_, *dec_[0]);
            }
            Schedule();
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
    std::unique_ptr<PacketDecoder> dec_[2];
    size_t stride_ = 0, bpf_ = 0;
    size_t idx_ = 0;
    const uint8_t* buf_ = nullptr;
    size_t buf_off_ = 0, buf_len_ = 0;
    bool eof_ = false;
    Window win_[kDepth + 1];
    unsigned cur_ = 0;
    Ahead ahead_[kDepth];
    unsigned n_ahead_ = 0, seq_ = 0;
};

inline std::unique_ptr<Decoder> NewDecoder(const uint8_t* file, size_t len, int device = 0, size_t window = 0) {
    return std::make_unique<Decoder>(file, len, device, window);
}

}  // namespace alac
