"""MP4 / M4A sample table -> batch descriptor.

Host-only, numpy: finds the first ALAC track of an ISO-BMFF file held in memory (bytes, mmap, numpy array) and
returns its magic cookie plus the flat sample table as two arrays — exactly the `(offsets, sizes)` a batch decode
needs, with no per-packet Python objects. Behaviour follows the reference's internal/mp4 (FindALACTrack,
internal/mp4/mp4.go:233-298; box headers :60-112; stsd walk :313-378; sample table :382-420; stco/co64 :442-493;
stsc :496-536; stsz :539-575; lookupSamplesPerChunk :579-591) — same tracks found, same sample list, same error
sentinels — but is written for whole files in memory: a generator over boxes instead of seek/read calls, and the
sample table is built with array operations instead of a loop per sample.
"""
import numpy as np

# error sentinels of internal/mp4/errors.go:24-34 (the text is the reference's)
ErrNoALACTrack = "mp4: no ALAC track found in container"
ErrInvalidEntry = "mp4: invalid ALAC sample entry"
ErrInvalidBoxSize = "mp4: invalid box size"
ErrNoChunkOffset = "mp4: no chunk offset box (stco/co64)"
ErrInvalidCo64 = "mp4: invalid co64 payload"
ErrNoStsc = "mp4: no stsc box"
ErrInvalidStsc = "mp4: invalid stsc payload"
ErrNoStsz = "mp4: no stsz box"
ErrInvalidStsz = "mp4: invalid stsz payload"


class Mp4Error(Exception):
    """A container-level failure; `.sentinel` is one of the Err* strings above (errors.Is target)."""

    def __init__(self, sentinel, detail=""):
        super().__init__(sentinel + (": " + detail if detail else ""))
        self.sentinel = sentinel


class Track:
    """cookie: raw magic cookie bytes; offsets[n] (uint64) / sizes[n] (uint32): the samples in decode order."""

    __slots__ = ("cookie", "offsets", "sizes")

    def __init__(self, cookie, offsets, sizes):
        self.cookie, self.offsets, self.sizes = cookie, offsets, sizes

    def __len__(self):
        return len(self.sizes)

    def contiguous(self):
        """True when sample i+1 starts where sample i ends (one mdat run: the blob needs no gather)."""
        if len(self.sizes) < 2:
            return True
        return bool(np.array_equal(self.offsets[1:], self.offsets[:-1] + self.sizes[:-1].astype(np.uint64)))


def _u32(buf, pos):
    return int.from_bytes(buf[pos:pos + 4], "big")


def _children(buf, start, end):
    """(fourcc, payload_start, box_end) of the boxes in buf[start:end]. A header cut short by the end of the data
    ends the walk (mp4.go:166-172); a size smaller than the header raises (mp4.go:107-109)."""
    pos = start
    n = len(buf)
    while pos < end:
        if pos + 8 > n:
            return
        size = _u32(buf, pos)
        fourcc = bytes(buf[pos + 4:pos + 8])
        header = 8
        if size == 0:  # to the end of the file
            size = n - pos
        elif size == 1:  # 64-bit size
            if pos + 16 > n:
                return
            size = int.from_bytes(buf[pos + 8:pos + 16], "big")
            header = 16
        if size < header:
            raise Mp4Error(ErrInvalidBoxSize, "size %d at offset %d" % (size, pos))
        yield fourcc, pos + header, pos + size
        pos += size


def _find(buf, start, end, fourcc):
    for cc, p0, p1 in _children(buf, start, end):
        if cc == fourcc:
            return p0, p1
    return None


def _cookie(buf, stbl):
    """The 'alac' sample entry's trailing bytes (mp4.go:313-378); None when this track has none."""
    box = _find(buf, stbl[0], stbl[1], b"stsd")
    if box is None:
        return None
    data = bytes(buf[box[0]:box[1]])
    if len(data) != box[1] - box[0] or len(data) < 8:
        return None  # truncated payload: "not an ALAC track"
    count = _u32(data, 4)
    pos = 8
    for _ in range(count):
        if pos + 8 > len(data):
            break
        size = _u32(data, pos)
        if size < 8 + 28 or pos + size > len(data) or data[pos + 4:pos + 8] != b"alac":
            if size == 0:
                break  # the reference would spin on a zero-sized entry; there is nothing behind it to find
            pos += size
            continue
        version = int.from_bytes(data[pos + 16:pos + 18], "big")  # reserved(6) dataRefIdx(2) version(2)
        skip = 8 + 28 + (16 if version == 1 else 0)  # QuickTime v1 sound description: 16 more bytes
        if pos + skip >= pos + size:
            raise Mp4Error(ErrInvalidEntry)
        return data[pos + skip:pos + size]
    return None


def _find_quiet(buf, box, fourcc):
    """_find for the table boxes of an stbl: a broken sibling reads as "not there" (mp4.go:427,433,500,543)."""
    try:
        return _find(buf, box[0], box[1], fourcc)
    except Mp4Error:
        return None


def _table(buf, stbl, fourcc, invalid):
    """(payload_start, entry_count) of a counted full box (version/flags, then a 32-bit count), or None."""
    box = _find_quiet(buf, stbl, fourcc)
    if box is None:
        return None
    p0 = box[0]
    if p0 + 8 > len(buf):
        raise Mp4Error(invalid, "unexpected EOF")
    return p0, _u32(buf, p0 + 4)


def _be_array(buf, pos, count, dtype, err):
    nbytes = count * np.dtype(dtype).itemsize
    if pos + nbytes > len(buf):
        raise Mp4Error(err, "unexpected EOF")
    return np.frombuffer(buf, dtype=dtype, count=count, offset=pos)


def _sample_table(buf, stbl):
    """mp4.go:382-420 with arrays: chunk offsets x samples-per-chunk runs x sample sizes."""
    t = _table(buf, stbl, b"stco", ErrNoChunkOffset)
    if t is not None:
        chunk_off = _be_array(buf, t[0] + 8, t[1], ">u4", ErrNoChunkOffset).astype(np.uint64)
    else:
        t = _table(buf, stbl, b"co64", ErrInvalidCo64)
        if t is None:
            raise Mp4Error(ErrNoChunkOffset)
        chunk_off = _be_array(buf, t[0] + 8, t[1], ">u8", ErrInvalidCo64).astype(np.uint64)

    t = _table(buf, stbl, b"stsc", ErrInvalidStsc)
    if t is None:
        raise Mp4Error(ErrNoStsc)
    runs = _be_array(buf, t[0] + 8, t[1] * 3, ">u4", ErrInvalidStsc).reshape(-1, 3).astype(np.int64)

    box = _find_quiet(buf, stbl, b"stsz")
    if box is None:
        raise Mp4Error(ErrNoStsz)
    if box[0] + 12 > len(buf):
        raise Mp4Error(ErrInvalidStsz, "unexpected EOF")
    const_size, n_samples = _u32(buf, box[0] + 4), _u32(buf, box[0] + 8)
    entry_sizes = None
    if const_size == 0:
        entry_sizes = _be_array(buf, box[0] + 12, n_samples, ">u4", ErrInvalidStsz).astype(np.uint32)

    n_chunks = len(chunk_off)
    if n_chunks == 0 or n_samples == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    # samples per chunk: the last run whose first_chunk <= chunk number, walking the runs in file order and
    # stopping at the first one beyond it (mp4.go:579-591) == a search in the running maximum of first_chunk
    per_chunk = np.zeros(n_chunks, np.int64)
    if len(runs):
        reach = np.maximum.accumulate(runs[:, 0])
        k = np.searchsorted(reach, np.arange(1, n_chunks + 1), side="right")
        per_chunk = np.where(k > 0, runs[np.maximum(k, 1) - 1, 1], 0)
    # a chunk takes its samples while the track still has some (mp4.go:402)
    before = np.cumsum(per_chunk) - per_chunk
    take = np.clip(n_samples - before, 0, per_chunk)
    total = int(take.sum())
    if total == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    sizes = np.full(total, const_size, np.uint32) if const_size else entry_sizes[:total]
    ends = np.cumsum(sizes.astype(np.uint64))
    starts = ends - sizes.astype(np.uint64)
    first = np.cumsum(take) - take  # index of each chunk's first sample
    chunk_of = np.repeat(np.arange(n_chunks), take)
    base = np.where(take > 0, starts[np.minimum(first, max(total - 1, 0))], 0).astype(np.uint64)
    offsets = chunk_off[chunk_of] + (starts - base[chunk_of])
    return offsets.astype(np.uint64), sizes


def find_alac_track(data):
    """FindALACTrack (mp4.go:233-298): the first trak whose stsd holds an 'alac' entry -> Track."""
    buf = memoryview(data).cast("B") if not isinstance(data, np.ndarray) else memoryview(np.ascontiguousarray(data, np.uint8))
    moov = _find(buf, 0, len(buf), b"moov")
    if moov is None:
        raise Mp4Error(ErrNoALACTrack)
    for cc, p0, p1 in _children(buf, moov[0], min(moov[1], len(buf))):
        if cc != b"trak":
            continue
        box = (p0, p1)
        for name in (b"mdia", b"minf", b"stbl"):
            box = _find(buf, box[0], min(box[1], len(buf)), name)
            if box is None:
                break
        if box is None:
            continue
        stbl = (box[0], min(box[1], len(buf)))
        try:
            cookie = _cookie(buf, stbl)
        except Mp4Error:
            cookie = None  # "not an ALAC track": go on to the next trak (mp4.go:277-280)
        if cookie is None:
            continue
        offsets, sizes = _sample_table(buf, stbl)
        return Track(cookie, offsets, sizes)
    raise Mp4Error(ErrNoALACTrack)
