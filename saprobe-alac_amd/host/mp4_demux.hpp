// mp4_demux.hpp — C++ host-side MP4 / M4A sample table -> batch descriptor.
//
// Finds the first ALAC track of an ISO-BMFF file held in memory and returns its magic cookie and the flat sample
// table (offset, size per packet): what a batch decode needs. Same tracks, sample list and error sentinels as the
// reference's internal/mp4 (FindALACTrack mp4.go:233-298; box headers :60-112; stsd :313-378; sample table
// :382-420; stco/co64 :442-493; stsc :496-536; stsz :539-575; lookupSamplesPerChunk :579-591), written for a whole
// file in memory (a span walker instead of seek/read calls). Header-only, no GPU involved.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace alac {
namespace mp4 {

// internal/mp4/errors.go:24-34
enum class Sentinel { NoALACTrack, InvalidEntry, InvalidBoxSize, NoChunkOffset, InvalidCo64, NoStsc, InvalidStsc, NoStsz, InvalidStsz };

inline const char* SentinelText(Sentinel s) {
    static const char* t[] = {"mp4: no ALAC track found in container", "mp4: invalid ALAC sample entry",
                              "mp4: invalid box size", "mp4: no chunk offset box (stco/co64)",
                              "mp4: invalid co64 payload", "mp4: no stsc box", "mp4: invalid stsc payload",
                              "mp4: no stsz box", "mp4: invalid stsz payload"};
    return t[(int)s];
}

struct Error : std::runtime_error {
    Sentinel sentinel;
    explicit Error(Sentinel s) : std::runtime_error(SentinelText(s)), sentinel(s) {}
};

struct Track {
    std::vector<uint8_t> cookie;
    std::vector<uint64_t> offsets;  // file offset of each sample, decode order
    std::vector<uint32_t> sizes;
};

namespace detail {

struct Span {
    const uint8_t* p;
    size_t n;
    uint32_t u32(size_t at) const { return (uint32_t)p[at] << 24 | (uint32_t)p[at + 1] << 16 | (uint32_t)p[at + 2] << 8 | p[at + 3]; }
    uint64_t u64(size_t at) const { return (uint64_t)u32(at) << 32 | u32(at + 4); }
};

struct Box {
    bool found = false;
    size_t payload = 0, end = 0;
};

// First child `want` of [start, end). A header cut short by the end of the data ends the walk (mp4.go:166-172); a
// size below the header throws (mp4.go:107-109). visit == nullptr: plain search.
template <class F>
inline void children(const Span& f, size_t start, size_t end, F&& visit) {
    size_t pos = start;
    while (pos < end) {
        if (pos + 8 > f.n) return;
        uint64_t size = f.u32(pos);
        size_t header = 8;
        if (size == 0) size = f.n - pos;
        else if (size == 1) {
            if (pos + 16 > f.n) return;
            size = f.u64(pos + 8);
            header = 16;
        }
        if (size < header) throw Error(Sentinel::InvalidBoxSize);
        const uint64_t e = (uint64_t)pos + size;
        if (visit(f.p + pos + 4, pos + header, e > (uint64_t)SIZE_MAX ? (size_t)SIZE_MAX : (size_t)e)) return;
        if (e >= (uint64_t)f.n) return;
        pos = (size_t)e;
    }
}

inline Box find(const Span& f, size_t start, size_t end, const char* want) {
    Box b;
    children(f, start, end, [&](const uint8_t* cc, size_t payload, size_t e) {
        if (memcmp(cc, want, 4) != 0) return false;
        b.found = true;
        b.payload = payload;
        b.end = e;
        return true;
    });
    return b;
}

inline Box find_quiet(const Span& f, const Box& parent, const char* want) {  // mp4.go:427,433,500,543
    try {
        return find(f, parent.payload, parent.end, want);
    } catch (const Error&) {
        return Box{};
    }
}

inline bool cookie_of(const Span& f, const Box& stbl, std::vector<uint8_t>& out) {  // mp4.go:313-378
    const Box stsd = find(f, stbl.payload, stbl.end, "stsd");
    if (!stsd.found || stsd.end > f.n || stsd.end - stsd.payload < 8) return false;
    const Span d{f.p + stsd.payload, stsd.end - stsd.payload};
    const uint32_t count = d.u32(4);
    size_t pos = 8;
    for (uint32_t k = 0; k < count; ++k) {
        if (pos + 8 > d.n) break;
        const size_t size = d.u32(pos);
        if (size < 8 + 28 || pos + size > d.n || memcmp(d.p + pos + 4, "alac", 4) != 0) {
            if (size == 0) break;
            pos += size;
            continue;
        }
        const unsigned version = (unsigned)d.p[pos + 16] << 8 | d.p[pos + 17];
        const size_t skip = 8 + 28 + (version == 1 ? 16 : 0);  // QuickTime v1 sound description: 16 more bytes
        if (skip >= size) throw Error(Sentinel::InvalidEntry);
        out.assign(d.p + pos + skip, d.p + pos + size);
        return true;
    }
    return false;
}

inline void sample_table(const Span& f, const Box& stbl, Track& t) {  // mp4.go:382-420
    std::vector<uint64_t> chunk;
    Box b = find_quiet(f, stbl, "stco");
    if (b.found) {
        if (b.payload + 8 > f.n) throw Error(Sentinel::NoChunkOffset);
        const uint32_t n = f.u32(b.payload + 4);
        if (b.payload + 8 + (uint64_t)n * 4 > f.n) throw Error(Sentinel::NoChunkOffset);
        chunk.resize(n);
        for (uint32_t i = 0; i < n; ++i) chunk[i] = f.u32(b.payload + 8 + (size_t)i * 4);
    } else {
        b = find_quiet(f, stbl, "co64");
        if (!b.found) throw Error(Sentinel::NoChunkOffset);
        if (b.payload + 8 > f.n) throw Error(Sentinel::InvalidCo64);
        const uint32_t n = f.u32(b.payload + 4);
        if (b.payload + 8 + (uint64_t)n * 8 > f.n) throw Error(Sentinel::InvalidCo64);
        chunk.resize(n);
        for (uint32_t i = 0; i < n; ++i) chunk[i] = f.u64(b.payload + 8 + (size_t)i * 8);
    }
    b = find_quiet(f, stbl, "stsc");
    if (!b.found) throw Error(Sentinel::NoStsc);
    if (b.payload + 8 > f.n) throw Error(Sentinel::InvalidStsc);
    const uint32_t n_runs = f.u32(b.payload + 4);
    if (b.payload + 8 + (uint64_t)n_runs * 12 > f.n) throw Error(Sentinel::InvalidStsc);
    const size_t runs = b.payload + 8;
    b = find_quiet(f, stbl, "stsz");
    if (!b.found) throw Error(Sentinel::NoStsz);
    if (b.payload + 12 > f.n) throw Error(Sentinel::InvalidStsz);
    const uint32_t const_size = f.u32(b.payload + 4), n_samples = f.u32(b.payload + 8);
    const size_t entries = b.payload + 12;
    if (const_size == 0 && entries + (uint64_t)n_samples * 4 > f.n) throw Error(Sentinel::InvalidStsz);

    t.offsets.clear();
    t.sizes.clear();
    uint32_t s = 0, run = 0, per_chunk = 0;
    for (size_t c = 0; c < chunk.size() && s < n_samples; ++c) {
        // lookupSamplesPerChunk (mp4.go:579-591) restarts from the first run for every chunk and stops at the first
        // run beyond it; chunk numbers only grow, so resuming where the previous chunk stopped gives the same run
        while (run < n_runs && f.u32(runs + (size_t)run * 12) <= c + 1) {
            per_chunk = f.u32(runs + (size_t)run * 12 + 4);
            ++run;
        }
        uint64_t off = chunk[c];
        for (uint32_t k = 0; k < per_chunk && s < n_samples; ++k, ++s) {
            const uint32_t size = const_size ? const_size : f.u32(entries + (size_t)s * 4);
            t.offsets.push_back(off);
            t.sizes.push_back(size);
            off += size;
        }
    }
}

}  // namespace detail

// FindALACTrack (mp4.go:233-298): the first trak whose stsd holds an 'alac' entry. Throws mp4::Error.
inline Track FindALACTrack(const uint8_t* data, size_t len) {
    using namespace detail;
    const Span f{data, len};
    const Box moov = find(f, 0, len, "moov");
    if (!moov.found) throw Error(Sentinel::NoALACTrack);
    Track t;
    bool done = false;
    children(f, moov.payload, moov.end, [&](const uint8_t* cc, size_t payload, size_t end) {
        if (memcmp(cc, "trak", 4) != 0) return false;
        Box b;
        b.found = true;
        b.payload = payload;
        b.end = end;
        for (const char* name : {"mdia", "minf", "stbl"}) {
            b = find(f, b.payload, b.end, name);
            if (!b.found) return false;
        }
        bool has = false;
        try {
            has = cookie_of(f, b, t.cookie);
        } catch (const Error&) {
            has = false;  // "not an ALAC track": go on to the next trak (mp4.go:277-280)
        }
        if (!has) return false;
        sample_table(f, b, t);
        done = true;
        return true;
    });
    if (!done) throw Error(Sentinel::NoALACTrack);
    return t;
}

}  // namespace mp4
}  // namespace alac
