//go:build alacgpu

// Package alac, cgo side: the GPU packet decoder behind libalacgpu.so (include/alacgpu.h).
//
// This file is meant to sit next to decoder.go in package alac of mycophonic/saprobe-alac: it gives the
// reference a GPUPacketDecoder with PacketDecoder's surface (decoder.go:79-128: NewPacketDecoder, DecodePacket,
// Format) plus the new batch entry DecodePackets, and leaves the pure-Go PacketDecoder untouched. It is only
// compiled with `-tags alacgpu` (it needs cgo, the header and the library):
//
//	CGO_CFLAGS=-I$REPO/include \
//	CGO_LDFLAGS="-L$REPO/saprobe-alac_amd/csrc -lalacgpu -Wl,-rpath,$REPO/saprobe-alac_amd/csrc" \
//	go test -tags alacgpu ./...
//
// The image this repository is built in has no Go toolchain: THIS FILE HAS NEVER BEEN COMPILED. It is kept in step with
// the header by tests/test_c_abi.py::test_go_shim_matches_the_header (every C function and constant it names must exist
// in include/alacgpu.h, and every call must pass as many arguments as the prototype has).
package alac

/*
#cgo LDFLAGS: -lalacgpu
#include <stdlib.h>
#include "alacgpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	alacint "github.com/mycophonic/saprobe-alac/internal/alac"
)

// ErrMalformed is returned (wrapped in ErrDecode) for packets on which the pure-Go path panics.
var ErrMalformed = errors.New("alac: malformed packet")

// ErrRange is returned (wrapped in ErrDecode) by the batch entries for a packet descriptor outside the blob.
var ErrRange = errors.New("alac: packet outside the blob")

// GPUPacketDecoder has PacketDecoder's surface (decoder.go:79-128) plus DecodePackets.
// Like PacketDecoder it is not safe for concurrent use; make one per goroutine (and per device).
type GPUPacketDecoder struct {
	h      *C.alacgpu_decoder
	config PacketConfig
	format PCMFormat
}

// NewGPUPacketDecoder mirrors NewPacketDecoder (decoder.go:90): same ErrConfig wrapping for a bad bit depth.
func NewGPUPacketDecoder(config PacketConfig, device int) (*GPUPacketDecoder, error) {
	cfg := C.alacgpu_config{
		frame_length: C.uint32_t(config.FrameLength), bit_depth: C.uint8_t(config.BitDepth),
		num_channels: C.uint8_t(config.NumChannels), pb: C.uint8_t(config.PB), mb: C.uint8_t(config.MB),
		kb: C.uint8_t(config.KB), max_run: C.uint16_t(config.MaxRun),
		max_frame_bytes: C.uint32_t(config.MaxFrameBytes), avg_bit_rate: C.uint32_t(config.AvgBitRate),
		sample_rate: C.uint32_t(config.SampleRate),
	}

	var h *C.alacgpu_decoder

	switch rc := C.alacgpu_create(&cfg, C.int(device), &h); rc {
	case C.ALACGPU_E_OK:
	case C.ALACGPU_E_CONFIG:
		if !isALACBitDepth(config.BitDepth) {
			return nil, fmt.Errorf("%w: %w: %d", ErrConfig, alacint.ErrBitDepth, config.BitDepth)
		}

		return nil, fmt.Errorf("%w: %s", ErrConfig, C.GoString(C.alacgpu_last_error()))
	default:
		return nil, fmt.Errorf("alacgpu: %s", C.GoString(C.alacgpu_last_error()))
	}

	d := &GPUPacketDecoder{h: h, config: config, format: PCMFormat{
		SampleRate: int(config.SampleRate), BitDepth: int(config.BitDepth), Channels: int(config.NumChannels),
	}}
	runtime.SetFinalizer(d, func(d *GPUPacketDecoder) { d.Close() })

	return d, nil
}

func isALACBitDepth(depth uint8) bool {
	return depth == 16 || depth == 20 || depth == 24 || depth == 32
}

// Close releases the device resources; the decoder must not be used afterwards.
func (d *GPUPacketDecoder) Close() {
	if d.h != nil {
		C.alacgpu_destroy(d.h)
		d.h = nil
	}
}

// Format returns the PCM output format (decoder.go:112).
func (d *GPUPacketDecoder) Format() PCMFormat { return d.format }

// Reserve sizes the handle's device workspace for batches of up to n packets. The batch entries grow it on demand, but a buffer
// that has to grow is freed first, and hipFree waits for the whole device: a caller that decodes batch after batch (a file
// decoder's windows) reserves for its largest one before the first decode (include/alacgpu.h: alacgpu_reserve).
func (d *GPUPacketDecoder) Reserve(n int) error {
	if n <= 0 {
		return nil
	}

	if rc := C.alacgpu_reserve(d.h, C.size_t(n)); rc != C.ALACGPU_E_OK {
		return fmt.Errorf("alacgpu: %s", C.GoString(C.alacgpu_last_error()))
	}

	return nil
}

// TrimGPUPool frees what closed decoders left in the library's per-process pools (handles with their streams and up to 2 GB
// of device memory each, pinned host blocks): include/alacgpu.h: alacgpu_trim.
func TrimGPUPool() { C.alacgpu_trim() }

// statusErr rebuilds the reference's error chain (decoder.go:144-189,303,468,482) from a status word.
func statusErr(st int32) error {
	var sentinel error

	switch st & 0xff {
	case C.ALACGPU_ERR_BITSTREAM_OVERRUN:
		sentinel = alacint.ErrBitstreamOverrun
	case C.ALACGPU_ERR_SAMPLE_OVERRUN:
		sentinel = alacint.ErrSampleOverrun
	case C.ALACGPU_ERR_INVALID_HEADER:
		sentinel = alacint.ErrInvalidHeader
	case C.ALACGPU_ERR_INVALID_SHIFT:
		sentinel = alacint.ErrInvalidShift
	case C.ALACGPU_ERR_UNSUPPORTED_ELEMENT:
		sentinel = alacint.ErrUnsupportedElement
	case C.ALACGPU_ERR_MALFORMED:
		sentinel = ErrMalformed
	case C.ALACGPU_ERR_RANGE:
		sentinel = ErrRange
	default:
		sentinel = fmt.Errorf("alacgpu: unknown status %#x", st)
	}

	ctx := [...]string{"", "SCE/LFE", "CPE", "DSE", "FIL"}[(st>>8)&0xf%5]
	stage := [...]string{"", "entropy decode", "entropy decode U", "entropy decode V"}[(st>>12)&0x3]

	switch {
	case ctx != "" && stage != "":
		return fmt.Errorf("%w: %s: %s: %w", ErrDecode, ctx, stage, sentinel)
	case ctx != "":
		return fmt.Errorf("%w: %s: %w", ErrDecode, ctx, sentinel)
	default:
		return fmt.Errorf("%w: %w", ErrDecode, sentinel)
	}
}

// DecodePacket mirrors (*PacketDecoder).DecodePacket (decoder.go:117): a fresh slice trimmed to n bytes.
func (d *GPUPacketDecoder) DecodePacket(packet []byte) ([]byte, error) {
	out := make([]byte, int(C.alacgpu_frame_bytes(d.h)))

	n, err := d.DecodePacketInto(packet, out)
	if err != nil {
		return nil, err
	}

	return out[:n], nil
}

// DecodePacketInto mirrors decodePacketInto (decoder.go:133), the entry Decoder.Read uses (decode.go:179): the PCM goes
// into the caller's buffer, which must hold a full frame (decoder.go:131-132); nothing is allocated. Returns the
// number of bytes written.
func (d *GPUPacketDecoder) DecodePacketInto(packet, out []byte) (int, error) {
	if len(out) < int(C.alacgpu_frame_bytes(d.h)) {
		return 0, fmt.Errorf("alacgpu: output buffer of %d bytes, a frame needs %d", len(out), int(C.alacgpu_frame_bytes(d.h)))
	}

	var (
		n  C.size_t
		st C.int32_t
		p  *C.uint8_t
	)

	if len(packet) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&packet[0]))
	}

	switch rc := C.alacgpu_decode_packet(d.h, p, C.size_t(len(packet)),
		(*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(len(out)), &n, &st); rc {
	case C.ALACGPU_E_OK:
		return int(n), nil
	case C.ALACGPU_E_DECODE:
		return 0, statusErr(int32(st))
	default:
		return 0, fmt.Errorf("alacgpu: %s", C.GoString(C.alacgpu_last_error()))
	}
}

// DecodePackets is the new batch entry: packets[i] -> pcm[i] (nil and errs[i] on a per-packet failure).
// The packets are copied back to back into one host blob (one copy of every packet on the host: callers that hold
// an mdat should use DecodeSamples, which copies nothing) and go to the device as they would lie in an mdat; the
// library needs no padding.
func (d *GPUPacketDecoder) DecodePackets(packets [][]byte) (pcm [][]byte, errs []error, err error) {
	count := len(packets)
	if count == 0 {
		return nil, nil, nil
	}

	offsets := make([]C.uint64_t, count+1)
	total := 0

	for i, p := range packets {
		offsets[i] = C.uint64_t(total)
		total += len(p)
	}

	offsets[count] = C.uint64_t(total)
	blob := make([]byte, total+1)

	for i, p := range packets {
		copy(blob[int(offsets[i]):], p)
	}

	stride := int(C.alacgpu_frame_bytes(d.h))
	out := make([]byte, count*stride)
	frames := make([]C.uint32_t, count)
	status := make([]C.int32_t, count)

	if rc := C.alacgpu_decode_batch(d.h, (*C.uint8_t)(unsafe.Pointer(&blob[0])), C.size_t(total), &offsets[0], C.size_t(count),
		(*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(stride), &frames[0], &status[0]); rc != C.ALACGPU_E_OK {
		return nil, nil, fmt.Errorf("alacgpu: %s", C.GoString(C.alacgpu_last_error()))
	}

	bpf := d.format.Channels * alacint.BytesPerSample(d.config.BitDepth)
	pcm, errs = make([][]byte, count), make([]error, count)

	for i := range packets {
		if status[i] != 0 {
			errs[i] = statusErr(int32(status[i]))

			continue
		}

		end := i*stride + int(frames[i])*bpf
		pcm[i] = out[i*stride : end : end]
	}

	return pcm, errs, nil
}

// DecodeSamples decodes a whole sample table in one call: mdat is the file region that holds the packets, packet i is
// mdat[offsets[i]:offsets[i+1]] (len(offsets) = packets + 1; internal/mp4 SampleInfo, mp4.go:29-32, rebased to mdat,
// for a contiguous run of samples). Nothing is copied on the host: the region goes to the device as it is. The table
// is not trusted: the library checks every descriptor against len(mdat) and reports ALACGPU_ERR_RANGE (ErrRange) in
// status[i] for a packet that leaves it, without reading it.
func (d *GPUPacketDecoder) DecodeSamples(mdat []byte, offsets []uint64) (pcm []byte, frames []uint32, status []int32, err error) {
	count := len(offsets) - 1
	if count <= 0 {
		return nil, nil, nil, nil
	}

	stride := int(C.alacgpu_frame_bytes(d.h))
	pcm = make([]byte, count*stride)
	frames = make([]uint32, count)
	status = make([]int32, count)

	var p *C.uint8_t
	if len(mdat) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&mdat[0]))
	}

	if rc := C.alacgpu_decode_batch(d.h, p, C.size_t(len(mdat)), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.size_t(count),
		(*C.uint8_t)(unsafe.Pointer(&pcm[0])), C.size_t(stride), (*C.uint32_t)(unsafe.Pointer(&frames[0])),
		(*C.int32_t)(unsafe.Pointer(&status[0]))); rc != C.ALACGPU_E_OK {
		return nil, nil, nil, fmt.Errorf("alacgpu: %s", C.GoString(C.alacgpu_last_error()))
	}

	return pcm, frames, status, nil
}
