"""TEST HELPER: a minimal M4A (ISO-BMFF) writer, just enough container around ALAC packets for the demuxer and the
streaming façade to chew on. Layout options cover what internal/mp4 distinguishes: stco vs co64, several chunks
with stsc runs, constant sample size, QuickTime v1 sample entries, a cookie wrapped in 'frma' + 'alac' atoms,
a non-ALAC track in front, 64-bit box sizes."""
import struct


def box(fourcc, payload, large=False):
    if large:
        return struct.pack(">I4sQ", 1, fourcc, 16 + len(payload)) + payload
    return struct.pack(">I4s", 8 + len(payload), fourcc) + payload


def full(fourcc, payload, version=0, flags=0):
    return box(fourcc, struct.pack(">I", (version << 24) | flags) + payload)


def cookie_bytes(cfg, wrapped=False):
    """ALACSpecificConfig (config.go:64-79), optionally behind 'frma' and 'alac' atoms (config.go:50-58)."""
    c = struct.pack(">IBBBBBBHIII", cfg.frame_length, 0, cfg.bit_depth, cfg.pb, cfg.mb, cfg.kb, cfg.num_channels,
                    cfg.max_run, cfg.max_frame_bytes, cfg.avg_bit_rate, cfg.sample_rate)
    if wrapped:
        c = struct.pack(">I4s4s", 12, b"frma", b"alac") + struct.pack(">I4sI", 12 + len(c), b"alac", 0) + c
    return c


def sample_entry(fourcc, cookie, qt_version=0):
    body = bytes(6) + struct.pack(">H", 1)                      # reserved, data reference index
    body += struct.pack(">HHIHHHHI", qt_version, 0, 0, 2, 16, 0, 0, 44100 << 16)  # AudioSampleEntry (20 bytes)
    if qt_version == 1:
        body += bytes(16)
    return box(fourcc, body + cookie)


def trak(entry, chunk_offsets, stsc_runs, sizes, const_size=0, co64=False):
    stsd = full(b"stsd", struct.pack(">I", 1) + entry)
    stsc = full(b"stsc", struct.pack(">I", len(stsc_runs)) + b"".join(struct.pack(">III", a, b, 1) for a, b in stsc_runs))
    if const_size:
        stsz = full(b"stsz", struct.pack(">II", const_size, len(sizes)))
    else:
        stsz = full(b"stsz", struct.pack(">II", 0, len(sizes)) + b"".join(struct.pack(">I", s) for s in sizes))
    if co64:
        stco = full(b"co64", struct.pack(">I", len(chunk_offsets)) + b"".join(struct.pack(">Q", o) for o in chunk_offsets))
    else:
        stco = full(b"stco", struct.pack(">I", len(chunk_offsets)) + b"".join(struct.pack(">I", o) for o in chunk_offsets))
    stts = full(b"stts", struct.pack(">I", 0))
    stbl = box(b"stbl", stsd + stts + stsc + stsz + stco)
    minf = box(b"minf", full(b"smhd", bytes(4)) + stbl)
    mdia = box(b"mdia", full(b"mdhd", bytes(20)) + full(b"hdlr", bytes(20)) + minf)
    return box(b"trak", full(b"tkhd", bytes(80)) + mdia)


def write_m4a(cfg, packets, per_chunk=None, co64=False, wrapped=False, qt_version=0, const_size=False,
              decoy_track=False, gap=0, large_mdat=False):
    """packets: list of bytes. per_chunk: samples per chunk (list of run lengths, cycled), default all in one chunk.
    gap: junk bytes between chunks. Returns the file as bytes."""
    ftyp = box(b"ftyp", b"M4A \0\0\0\0M4A mp42isom")
    n = len(packets)
    runs = per_chunk or [max(n, 1)]
    chunks, i, r = [], 0, 0
    while i < n:
        k = min(runs[r % len(runs)], n - i)
        chunks.append(packets[i:i + k])
        i += k
        r += 1
    # stsc runs: (first_chunk, samples_per_chunk) whenever the count changes
    stsc, prev = [], None
    for ci, ch in enumerate(chunks):
        if len(ch) != prev:
            stsc.append((ci + 1, len(ch)))
            prev = len(ch)
    if not stsc:
        stsc = [(1, 1)]
    sizes = [len(p) for p in packets]
    cookie = cookie_bytes(cfg, wrapped)

    def build(offsets):
        traks = b""
        if decoy_track:
            traks += trak(sample_entry(b"mp4a", b"\0" * 8), [0], [(1, 1)], [4])
        traks += trak(sample_entry(b"alac", cookie, qt_version), offsets, stsc, sizes,
                      const_size=sizes[0] if (const_size and sizes) else 0, co64=co64)
        return box(b"moov", full(b"mvhd", bytes(96)) + traks)

    moov = build([0] * len(chunks))
    mdat_hdr = 16 if large_mdat else 8
    pos = len(ftyp) + len(moov) + mdat_hdr
    offsets, body = [], b""
    for ch in chunks:
        body += bytes([0xAB]) * gap
        offsets.append(pos + len(body))
        body += b"".join(ch)
    moov = build(offsets)
    assert len(moov) == pos - len(ftyp) - mdat_hdr
    return ftyp + moov + box(b"mdat", body, large=large_mdat)
