//go:build alacgpu

// GPUDecoder: the streaming Decoder of decode.go (NewDecoder / Format / Duration / Position / Seek / Read, decode.go:32-190)
// on top of GPUPacketDecoder.DecodeSamples. Where Decoder.Read decodes one packet per loop turn (decode.go:157-186), this
// one gathers a window of packets from the sample table, decodes them in ONE batch on the GPU and serves Read from the
// decoded window; a packet is a serial chain of 2 x FrameLength steps whatever the batch size, so a window of thousands
// of packets takes about as long as one (DESIGN.md §3.7: 4 096 packets 1.5 ms, one packet 1.1 ms).
//
// Meant to sit next to decode.go in package alac of mycophonic/saprobe-alac, like alacgpu.go, and like it THIS FILE
// HAS NEVER BEEN COMPILED (no Go toolchain in the build image). It calls nothing of the C ABI itself. The same logic,
// compiled and tested against the oracle on the GPU, is saprobe-alac_amd/host/stream_decoder.hpp (C++) and
// saprobe-alac_amd/stream.py (Python): tests/test_container.py.
package alac

import (
	"fmt"
	"io"
	"time"

	alacint "github.com/mycophonic/saprobe-alac/internal/alac"
	mp4int "github.com/mycophonic/saprobe-alac/internal/mp4"
)

// DefaultGPUWindow is the number of packets decoded per batch: 4096 packets of 4096 frames are 93 s of CD audio and
// 64 MiB of 16-bit stereo PCM.
const DefaultGPUWindow = 4096

// GPUDecoder has Decoder's surface. Not safe for concurrent use.
type GPUDecoder struct {
	reader    io.ReadSeeker
	dec       *GPUPacketDecoder
	config    PacketConfig
	samples   []mp4int.SampleInfo
	sampleIdx int // next packet Read will hand out (Position, decode.go:92-98)
	window    int

	// the decoded window: packets [winFirst, winFirst+len(winFrames))
	winFirst  int
	winPCM    []byte
	winFrames []uint32
	winStatus []int32
	stride    int
	bufOff    int // bytes of packet sampleIdx already handed out
	packed    []byte

	// a sample of the last window that could not be read (-1: none): Read fails there, behind the packets in front of it
	lostIdx int
	lostErr error
}

// NewGPUDecoder mirrors NewDecoder (decode.go:50-80) on the given device; window <= 0 means DefaultGPUWindow.
func NewGPUDecoder(rs io.ReadSeeker, device, window int) (*GPUDecoder, error) {
	cookie, samples, err := mp4int.FindALACTrack(rs)
	if err != nil {
		return nil, fmt.Errorf("%w: %w", ErrNoTrack, err)
	}

	config, err := ParseMagicCookie(cookie)
	if err != nil {
		return nil, fmt.Errorf("parsing ALAC config: %w", err)
	}

	dec, err := NewGPUPacketDecoder(config, device)
	if err != nil {
		return nil, err
	}

	if window <= 0 {
		window = DefaultGPUWindow
	}

	// every window has the same size and the workspace for it exists before the first one is decoded: a buffer that grows in
	// the middle of a stream is freed first, which stalls the device (host/stream_decoder.hpp learned that in round 4)
	if err := dec.Reserve(min(window, len(samples))); err != nil {
		dec.Close()

		return nil, err
	}

	bps := alacint.BytesPerSample(config.BitDepth)

	return &GPUDecoder{
		reader: rs, dec: dec, config: config, samples: samples, window: window,
		stride: int(config.FrameLength) * int(config.NumChannels) * bps, lostIdx: -1,
	}, nil
}

// Close releases the GPU handle (it goes back to the library's pool).
func (s *GPUDecoder) Close() { s.dec.Close() }

// Format returns the PCM output format (decode.go:83).
func (s *GPUDecoder) Format() PCMFormat { return s.dec.Format() }

// Duration is Decoder.Duration (decode.go:87-93).
func (s *GPUDecoder) Duration() time.Duration {
	total := int64(len(s.samples)) * int64(s.config.FrameLength)

	return time.Duration(total * int64(time.Second) / int64(s.config.SampleRate))
}

// Position is Decoder.Position (decode.go:96-102): always a packet boundary; a packet counts from the moment Read has
// handed out its first byte (the reference advances its index when it decodes the packet, before draining it).
func (s *GPUDecoder) Position() time.Duration {
	idx := s.sampleIdx
	if s.bufOff > 0 {
		idx++
	}

	cur := int64(idx) * int64(s.config.FrameLength)

	return time.Duration(cur * int64(time.Second) / int64(s.config.SampleRate))
}

// Seek is Decoder.Seek (decode.go:108-130); the decoded window is kept when the target lies inside it.
func (s *GPUDecoder) Seek(t time.Duration) (time.Duration, error) {
	frameLength := int64(s.config.FrameLength)
	sampleRate := int64(s.config.SampleRate)
	targetFrame := int64(t.Seconds() * float64(sampleRate))
	target := int(targetFrame / frameLength)
	target = max(0, min(target, len(s.samples)))
	s.sampleIdx = target
	s.bufOff = 0

	return time.Duration(int64(target) * frameLength * int64(time.Second) / sampleRate), nil
}

// fill decodes the window that starts at packet first: the packets are read into one dense blob (a sample table need
// not be contiguous in the file) and go to the device in one call.
func (s *GPUDecoder) fill(first int) error {
	count := min(s.window, len(s.samples)-first)
	offsets := make([]uint64, count+1)

	var total uint64
	for i := 0; i < count; i++ {
		offsets[i] = total
		total += uint64(s.samples[first+i].Size)
	}

	offsets[count] = total

	if uint64(cap(s.packed)) < total {
		s.packed = make([]byte, total)
	}

	s.packed = s.packed[:total]

	// A sample that cannot be read (a truncated file, a bad stco entry) must not take the intact packets in front of it
	// with it: the reference decodes packet by packet and fails when it gets there (decode.go:157-186). The window shrinks
	// to the packets read so far, they are decoded and served, and Read returns the error when sampleIdx reaches the lost
	// sample (lostIdx / lostErr), as stream_decoder.hpp does with lost_ and stream.py with _w_read_err.
	s.lostIdx, s.lostErr = -1, nil

	for i := 0; i < count; i++ {
		sample := s.samples[first+i]

		var err error
		if _, err = s.reader.Seek(int64(sample.Offset), io.SeekStart); err != nil {
			err = fmt.Errorf("seeking to sample %d at offset %d: %w", first+i, sample.Offset, err)
		} else if _, err = io.ReadFull(s.reader, s.packed[offsets[i]:offsets[i+1]]); err != nil {
			err = fmt.Errorf("reading sample %d: %w", first+i, err)
		}

		if err != nil {
			s.lostIdx, s.lostErr = first+i, err
			count = i
			offsets = offsets[:count+1]
			s.packed = s.packed[:offsets[count]]

			break
		}
	}

	if count == 0 { // the window's first sample is the lost one: nothing to decode
		s.winFirst, s.winPCM, s.winFrames, s.winStatus = first, nil, []uint32{}, nil

		return nil
	}

	pcm, frames, status, err := s.dec.DecodeSamples(s.packed, offsets)
	if err != nil {
		return err
	}

	s.winFirst, s.winPCM, s.winFrames, s.winStatus = first, pcm, frames, status

	return nil
}

// Read is Decoder.Read (decode.go:133-190) served from the decoded window: same bytes, same io.EOF behaviour, and a
// packet that does not decode ends the stream there with the reference's error chain ("decoding packet %d: ...").
func (s *GPUDecoder) Read(p []byte) (int, error) { //nolint:varnamelen // p is idiomatic for io.Reader.Read
	total := 0
	bps := alacint.BytesPerSample(s.config.BitDepth)

	for len(p) > 0 {
		if s.sampleIdx >= len(s.samples) {
			if total > 0 {
				return total, nil
			}

			return 0, io.EOF
		}

		if s.sampleIdx == s.lostIdx {
			return total, s.lostErr
		}

		if s.winFrames == nil || s.sampleIdx < s.winFirst || s.sampleIdx >= s.winFirst+len(s.winFrames) {
			if err := s.fill(s.sampleIdx); err != nil {
				return total, err
			}

			if s.sampleIdx == s.lostIdx { // the window's first sample could not be read
				return total, s.lostErr
			}
		}

		k := s.sampleIdx - s.winFirst
		if st := s.winStatus[k]; st != 0 {
			return total, fmt.Errorf("decoding packet %d: %w", s.sampleIdx, statusErr(st))
		}

		size := int(s.winFrames[k]) * int(s.config.NumChannels) * bps // decoder.go:206
		data := s.winPCM[k*s.stride : k*s.stride+size]

		n := copy(p, data[s.bufOff:])
		s.bufOff += n
		total += n
		p = p[n:]

		if s.bufOff >= size {
			s.sampleIdx++
			s.bufOff = 0
		}
	}

	return total, nil
}
