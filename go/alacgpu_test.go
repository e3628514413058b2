//go:build alacgpu

package alac

import (
	"bytes"
	"encoding/hex"
	"errors"
	"testing"

	alacint "github.com/mycophonic/saprobe-alac/internal/alac"
)

// The hand-derived known-answer packets of tests/golden/kat.json (K1..K4) and kat2.json (K5, K8): the GPU decoder and
// the pure-Go decoder must both reproduce them, and each other.
var kats = []struct {
	name        string
	frameLength uint32
	depth, ch   uint8
	mb          uint8
	packet, pcm string
}{
	{"K1 mono escape", 4, 16, 1, 10, "0000020003FFFEFFFF0001C0", "0100FFFFFF7F0080"},
	{"K2 mono all-zero", 8, 16, 1, 10, "0000000000010047", "00000000000000000000000000000000"},
	{"K3 mono residuals", 4, 16, 1, 10, "0000000000010181CE", "0100FFFF02000000"},
	{"K4 stereo mix + escape code", 2, 16, 2, 10, "2000000402010001018EF7FC0017C0", "03000100FEFF0400"},
	{"K8 general order 2 with int16 wrap", 8, 16, 1, 255, "00000000001E04FFFF0001FBD7AFB380", "03000500040004000600FFFFFEFF0500"},
}

func TestGPUMatchesKnownAnswersAndPureGo(t *testing.T) {
	for _, k := range kats {
		cfg := PacketConfig{FrameLength: k.frameLength, BitDepth: k.depth, NumChannels: k.ch, PB: 40, MB: k.mb, KB: 14,
			MaxRun: 255, SampleRate: 44100}
		packet, _ := hex.DecodeString(k.packet)
		want, _ := hex.DecodeString(k.pcm)

		ref, err := NewPacketDecoder(cfg)
		if err != nil {
			t.Fatal(err)
		}

		refPCM, err := ref.DecodePacket(packet)
		if err != nil || !bytes.Equal(refPCM, want) {
			t.Fatalf("%s: pure-Go path: %v %x", k.name, err, refPCM)
		}

		gpu, err := NewGPUPacketDecoder(cfg, 0)
		if err != nil {
			t.Skipf("no GPU decoder: %v", err)
		}

		got, err := gpu.DecodePacket(packet)
		if err != nil || !bytes.Equal(got, want) {
			t.Fatalf("%s: GPU path: %v %x", k.name, err, got)
		}

		pcm, errs, err := gpu.DecodePackets([][]byte{packet, {}, packet[:len(packet)/2], packet})
		if err != nil {
			t.Fatal(err)
		}

		if !bytes.Equal(pcm[0], want) || !bytes.Equal(pcm[3], want) {
			t.Fatalf("%s: batch entry differs", k.name)
		}

		if !errors.Is(errs[1], ErrDecode) || !errors.Is(errs[1], alacint.ErrBitstreamOverrun) {
			t.Fatalf("%s: empty packet: %v", k.name, errs[1])
		}

		gpu.Close()
	}
}

func TestGPUConfigErrors(t *testing.T) {
	_, err := NewGPUPacketDecoder(PacketConfig{FrameLength: 4096, BitDepth: 13, NumChannels: 2}, 0)
	if !errors.Is(err, ErrConfig) || !errors.Is(err, alacint.ErrBitDepth) {
		t.Fatalf("bit depth 13: %v", err)
	}
}
