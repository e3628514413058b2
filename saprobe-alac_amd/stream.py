"""Streaming façade over the batch path: the reference's `Decoder` (decode.go:32-190).

Same surface — NewDecoder, Format, Duration, Position, Seek, Read — and the same observable behaviour: PCM bytes in
packet order, packet-aligned seeks, the error for packet k raised when the reader reaches packet k and again on
every later Read. What differs is how the PCM gets made: instead of one DecodePacket per packet the decoder reads
ahead — a window of packets goes through ONE batch decode on the GPU (`alacgpu_decode_batch`: gather, H2D, the
kernels, D2H) and Read / Seek are served from the decoded window. While the caller drains window k the library has
windows k + 1 and k + 2 decoded on threads of its own, on two handles (alacgpu_decode_batch_start / _wait: a Python thread
would spend the time waiting for the interpreter lock the draining loop holds; a window's decode is a chain of staging copy,
upload, a kernel that takes its 1.2 ms however few packets it holds, and download — two chains in flight fill each other's
gaps), as host/stream_decoder.hpp does. Every window has the same size and both handles reserve their workspace for it
up front: a buffer that grows in the middle of a stream is freed first, and hipFree waits for the whole device.
"""
import os

import numpy as np

from . import AlacError, ErrConfig, NewPacketDecoder, ParseMagicCookie, bytes_per_sample, status_error
from . import mp4

ErrNoTrackText = "no track found"  # errors.go:29


class ErrNoTrack(AlacError):
    """errors.go:29 — no usable ALAC track in the container; `.sentinel` is the internal/mp4 sentinel."""


def _as_buffer(source):
    """bytes-like, numpy array, path, or binary file object -> a buffer holding the whole file (paths are mapped)."""
    if isinstance(source, (str, os.PathLike)):
        return np.memmap(source, dtype=np.uint8, mode="r")
    if hasattr(source, "read"):
        if hasattr(source, "seek"):
            source.seek(0)
        return source.read()
    return source


class Decoder:
    """Streams decoded PCM from an ALAC M4A/MP4 source (decode.go:32-45). `window` = packets per batch decode; 0: as many
    as make 48 MB of PCM (host/stream_decoder.hpp)."""

    def __init__(self, source, device=0, window=0):
        self._data = _as_buffer(source)
        self._view = memoryview(self._data).cast("B") if not isinstance(self._data, np.ndarray) else memoryview(self._data)
        try:
            track = mp4.find_alac_track(self._view)
        except mp4.Mp4Error as e:  # decode.go:52-54
            err = ErrNoTrack("%s: %s" % (ErrNoTrackText, e))
            err.sentinel = e.sentinel
            raise err from None
        try:
            self.config = ParseMagicCookie(track.cookie)
        except ErrConfig as e:  # decode.go:57-59
            err = ErrConfig("parsing ALAC config: %s" % e)
            err.sentinel = e.sentinel
            raise err from None
        self._dec = NewPacketDecoder(self.config, device)
        self._decs = (self._dec, NewPacketDecoder(self.config, device))  # destroyed handles are pooled by the library
        for d in self._decs:
            d.reserve(min(int(window) if int(window) > 0 else max(64, (48 << 20) // max(1, self._dec.frame_bytes)), max(1, len(track.sizes))))
        self._offsets, self._sizes = track.offsets, track.sizes
        self._bpf = self.config.NumChannels * bytes_per_sample(self.config.BitDepth)
        self._window = int(window) if int(window) > 0 else max(64, (48 << 20) // max(1, self._dec.frame_bytes))
        self._idx = 0                    # sampleIdx: next packet to hand out
        self._buf = b""                  # PCM of the packet being drained (decode.go:40-42)
        self._buf_off = 0
        self._eof = False
        self._w0 = self._w1 = 0          # decoded window: packets [w0, w1)
        self._w_out = self._w_frames = self._w_status = None
        self._w_read_err = None          # (packet index, message): a sample that lies outside the file
        self._aheads = []                # the windows behind the current one, being decoded: [w0, w1, planned end, read_err, token, decoder], oldest first
        self._seq = 0                    # consecutive windows take turns on the two handles

    # ---- decode.go:79-124 -------------------------------------------------------------------------------
    def Format(self):
        return self._dec.Format()

    def Duration(self):
        """Seconds; an approximation from packet count and frame length, as in the reference (decode.go:82-88)."""
        total = len(self._sizes) * int(self.config.FrameLength)
        return (total * 1_000_000_000 // int(self.config.SampleRate)) / 1e9

    def Position(self):
        cur = self._idx * int(self.config.FrameLength)
        return (cur * 1_000_000_000 // int(self.config.SampleRate)) / 1e9

    def Seek(self, seconds):
        """Packet-aligned seek; returns the position actually reached (decode.go:103-124)."""
        fl, sr = int(self.config.FrameLength), int(self.config.SampleRate)
        target = int(int(float(seconds) * sr) // fl) if seconds > 0 else 0
        target = max(0, min(target, len(self._sizes)))
        self._idx = target
        self._buf, self._buf_off = b"", 0
        self._eof = target >= len(self._sizes)
        return (self._idx * fl * 1_000_000_000 // sr) / 1e9

    # ---- the read-ahead window -------------------------------------------------------------------------------
    def _prepare(self, first, count):
        """Packets [first, first + count): -> (w0, w1, blob, starts, read_err), nothing decoded yet."""
        last = min(first + count, len(self._sizes))
        offs = self._offsets[first:last].astype(np.int64)
        sizes = self._sizes[first:last].astype(np.int64)
        n_file = len(self._view)
        read_err = None
        bad = np.nonzero(offs + sizes > n_file)[0]
        if len(bad):  # decode.go:163-169: the seek or the ReadFull of that sample fails when the reader gets there
            k = int(bad[0])
            read_err = (first + k, "reading sample %d: unexpected EOF" % (first + k))
            last = first + k
            offs, sizes = offs[:k], sizes[:k]
        n = last - first
        starts = np.zeros(n + 1, np.uint64)
        starts[1:] = np.cumsum(sizes, dtype=np.uint64)
        raw = np.frombuffer(self._view, dtype=np.uint8)
        if n and np.array_equal(offs[1:], offs[:-1] + sizes[:-1]):
            blob = raw[int(offs[0]):int(offs[0]) + int(starts[n])]  # one mdat run: no gather
        else:
            blob = np.empty(int(starts[n]) + 1, np.uint8)
            for k in range(n):
                blob[int(starts[k]):int(starts[k + 1])] = raw[int(offs[k]):int(offs[k] + sizes[k])]
        if n and blob.size == 0:
            blob = np.zeros(1, np.uint8)  # only empty packets: the entry still wants a readable pointer
        return first, last, blob, starts, read_err

    def _decode(self, first, count):
        """-> (w0, w1, out, frames, status, read_err): one batch decode, here and now."""
        w0, w1, blob, starts, read_err = self._prepare(first, count)
        out = frames = status = None
        if w1 > w0:
            out, frames, status = self._dec.decode_batch(blob, starts)
        return w0, w1, out, frames, status, read_err

    def _install(self, w):
        self._w0, self._w1, self._w_out, self._w_frames, self._w_status, self._w_read_err = w

    def _settle(self):
        """Waits for the oldest read-ahead; -> its window, or None (it failed: the reader meets the error again)."""
        w0, w1, _, read_err, token, dec = self._aheads.pop(0)
        try:
            out, frames, status = dec.decode_batch_wait(token)
        except AlacError:
            return None
        return w0, w1, out, frames, status, read_err

    def _drop(self):
        while self._aheads:
            self._settle()

    def _schedule(self):
        """Keeps two windows in flight behind the current one."""
        if self._w_read_err is not None or self._w1 <= self._w0:
            return  # the stream ends at the lost sample
        while len(self._aheads) < 2:
            first = self._aheads[-1][2] if self._aheads else self._w1
            if first >= len(self._sizes):
                return
            n0, n1, blob, starts, read_err = self._prepare(first, self._window)
            if n1 <= n0:
                return
            dec = self._decs[self._seq & 1]
            self._seq += 1
            self._aheads.append([n0, n1, min(first + self._window, len(self._sizes)), read_err, dec.decode_batch_start(blob, starts), dec])

    def _decode_window(self, first):
        w = None
        if self._aheads and self._aheads[0][0] == first:  # the reader walked off the end of its window into the next one
            w = self._settle()
            if w is not None and not (w[0] <= first < w[1] or (w[5] is not None and w[5][0] == first)):
                w = None
        if w is None:  # the first window, a seek, or a failed read-ahead (its error comes out here)
            self._drop()
            self._seq = 1
            w = self._decode(first, self._window)
        self._install(w)
        self._schedule()

    def _next_packet(self):
        """PCM of packet self._idx (decode.go:157-187); raises what the reference returns from Read."""
        k = self._idx
        lost = self._w_read_err is not None and self._w_read_err[0] == k
        if not lost and not (self._w0 <= k < self._w1):
            self._decode_window(k)
        if self._w_read_err is not None and self._w_read_err[0] == k:
            raise AlacError(self._w_read_err[1])
        j = k - self._w0
        st = int(self._w_status[j])
        if st:
            e = status_error(st)
            err = type(e)("decoding packet %d: %s" % (k, e), status=st, sentinel=e.sentinel)
            raise err
        pcm = self._w_out[j, :int(self._w_frames[j]) * self._bpf].tobytes()
        self._idx += 1
        return pcm

    # ---- decode.go:126-190 -----------------------------------------------------------------------------------
    def Read(self, n):
        """Up to n bytes of PCM; b"" at the end of the stream (io.EOF). An error is raised only when nothing was
        read before it in this call, like the reference's (total, err) returns read by io.ReadFull users: data
        first, the error on the next call."""
        out = bytearray()
        while len(out) < n:
            if self._buf_off < len(self._buf):
                take = min(n - len(out), len(self._buf) - self._buf_off)
                out += self._buf[self._buf_off:self._buf_off + take]
                self._buf_off += take
                continue
            if self._eof or self._idx >= len(self._sizes):
                self._eof = True
                break
            try:
                self._buf, self._buf_off = self._next_packet(), 0
            except AlacError:
                if out:
                    break
                raise
        return bytes(out)

    def readinto(self, b):
        data = self.Read(len(b))
        b[:len(data)] = data
        return len(data)

    def ReadAll(self):
        """Everything from the current position to the end, window by window."""
        parts = []
        while True:
            chunk = self.Read(self._window * int(self.config.FrameLength) * self._bpf)
            if not chunk:
                break
            parts.append(chunk)
        return b"".join(parts)

    def close(self):
        self._drop()
        for d in self._decs:
            d.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def NewDecoder(source, device=0, window=0):
    """NewDecoder (decode.go:50-76)."""
    return Decoder(source, device=device, window=window)
