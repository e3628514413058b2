/*
 * alacgpu.h — C ABI of the MI355X (gfx950) batch ALAC packet decoder.
 *
 * This is the drop-in boundary for the reference's packet layer
 * (mycophonic/saprobe-alac, paths relative to the reference tree):
 *
 *   reference (Go)                                   this ABI
 *   ------------------------------------------------ ---------------------------
 *   PacketConfig            config.go:27-38          alacgpu_config
 *   PCMFormat               format.go:20-24          alacgpu_format
 *   NewPacketDecoder        decoder.go:90-109        alacgpu_create
 *   (*PacketDecoder).Format decoder.go:112-114       alacgpu_get_format
 *   (*PacketDecoder).DecodePacket  decoder.go:117-128  alacgpu_decode_packet
 *   decodePacketInto        decoder.go:133-207       alacgpu_decode_packet (caller buffer)
 *   DecodePackets (new batch entry, north star)      alacgpu_decode_batch / _device
 *   ErrDecode + internal sentinels errors.go:22-34,  alacgpu_status (per packet)
 *                           internal/alac/errors.go:24-33
 *
 * Plain pointers and sizes only; no C++ or torch types. All functions are
 * re-entrant across different handles; one handle is single-caller, like a
 * PacketDecoder (decoder.go:79-87 holds mutable scratch).
 *
 * There is NO CPU decode path behind this ABI: every decode entry runs the HIP
 * kernels on the handle's device and returns ALACGPU_E_HIP if that fails.
 */
#ifndef ALACGPU_H
#define ALACGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors PacketConfig (config.go:27-38) field for field. */
typedef struct alacgpu_config {
    uint32_t frame_length;    /* FrameLength   */
    uint8_t  bit_depth;       /* BitDepth: 16, 20, 24 or 32 */
    uint8_t  num_channels;    /* NumChannels: 1..8 */
    uint8_t  pb;              /* PB */
    uint8_t  mb;              /* MB */
    uint8_t  kb;              /* KB */
    uint8_t  reserved0;
    uint16_t max_run;         /* MaxRun (stored, never read by decode) */
    uint32_t max_frame_bytes; /* MaxFrameBytes */
    uint32_t avg_bit_rate;    /* AvgBitRate */
    uint32_t sample_rate;     /* SampleRate */
} alacgpu_config;

/* Mirrors PCMFormat (format.go:20-24). */
typedef struct alacgpu_format {
    int32_t sample_rate;
    int32_t bit_depth;
    int32_t channels;
} alacgpu_format;

/*
 * Per-packet status word (int32).
 *   bits 0..7   code      (alacgpu_code) — which internal sentinel
 *   bits 8..11  context   (alacgpu_ctx)  — the wrapping string the reference adds
 *   bits 12..13 stage     (alacgpu_stage)— "entropy decode" / "entropy decode U|V"
 * 0 means success. The Go shim rebuilds the reference's error chain from it:
 *   fmt.Errorf("%w: <ctx>: <stage>: %w", ErrDecode, <sentinel>)
 */
typedef enum alacgpu_code {
    ALACGPU_OK                     = 0,
    ALACGPU_ERR_BITSTREAM_OVERRUN  = 1, /* ErrBitstreamOverrun   internal/alac/errors.go:30 */
    ALACGPU_ERR_SAMPLE_OVERRUN     = 2, /* ErrSampleOverrun      internal/alac/errors.go:31 */
    ALACGPU_ERR_INVALID_HEADER     = 3, /* ErrInvalidHeader      internal/alac/errors.go:28 */
    ALACGPU_ERR_INVALID_SHIFT      = 4, /* ErrInvalidShift       internal/alac/errors.go:29 */
    ALACGPU_ERR_UNSUPPORTED_ELEMENT= 5, /* ErrUnsupportedElement internal/alac/errors.go:27 */
    ALACGPU_ERR_MALFORMED          = 6, /* input on which the Go reference panics (slice bounds);
                                           it has no defined result there, we return a status */
    ALACGPU_ERR_RANGE              = 7  /* batch entries only: the packet's offset / size does not lie inside the
                                           blob (no reference counterpart: a Go slice cannot be out of range) */
} alacgpu_code;

typedef enum alacgpu_ctx {
    ALACGPU_CTX_NONE = 0,
    ALACGPU_CTX_SCE  = 1, /* "SCE/LFE" decoder.go:156 */
    ALACGPU_CTX_CPE  = 2, /* "CPE"     decoder.go:173 */
    ALACGPU_CTX_DSE  = 3, /* "DSE"     decoder.go:184 */
    ALACGPU_CTX_FIL  = 4  /* "FIL"     decoder.go:189 */
} alacgpu_ctx;

typedef enum alacgpu_stage {
    ALACGPU_STAGE_NONE      = 0,
    ALACGPU_STAGE_ENTROPY   = 1, /* "entropy decode"   decoder.go:303 */
    ALACGPU_STAGE_ENTROPY_U = 2, /* "entropy decode U" decoder.go:468 */
    ALACGPU_STAGE_ENTROPY_V = 3  /* "entropy decode V" decoder.go:482 */
} alacgpu_stage;

#define ALACGPU_STATUS(code, ctx, stage) ((int32_t)((code) | ((ctx) << 8) | ((stage) << 12)))
#define ALACGPU_STATUS_CODE(s)  ((s) & 0xff)
#define ALACGPU_STATUS_CTX(s)   (((s) >> 8) & 0xf)
#define ALACGPU_STATUS_STAGE(s) (((s) >> 12) & 0x3)

/* Call-level return values (negative = the call itself failed). */
#define ALACGPU_E_OK        0
#define ALACGPU_E_CONFIG   -1  /* ErrConfig: unsupported bit depth (decoder.go:91-93) or
                                  NumChannels outside 1..8 (the reference index-panics at decoder.go:140) */
#define ALACGPU_E_ARG      -2  /* null pointer / capacity too small */
#define ALACGPU_E_HIP      -3  /* HIP runtime failure; see alacgpu_last_error */
#define ALACGPU_E_DECODE   -4  /* alacgpu_decode_packet only: packet failed, *status_out holds the word */

/* Packets may lie DENSELY in a blob, back to back and at any alignment — an mdat as it is in the file
 * (internal/mp4/mp4.go:382-420). The kernels treat every byte behind a packet's last one as zero (the reference pads
 * each packet with 4 zero bytes, bitbuffer.go:33) and never touch memory outside [blob, blob + blob_bytes) rounded out
 * to 4-byte words. Round 1 required 64 zero bytes behind every packet; padding is harmless but no longer needed. */
#define ALACGPU_PACKET_PAD 0

typedef struct alacgpu_decoder alacgpu_decoder;

/* NewPacketDecoder (decoder.go:90). device = HIP ordinal; one stream per handle. */
int alacgpu_create(const alacgpu_config* cfg, int device, alacgpu_decoder** out);
/* A destroyed handle's streams, events and small buffers are kept for the next alacgpu_create on the same device: a file
 * decoder makes and drops a handle per file (decode.go:50-80), and building one from nothing costs several times the
 * decode of a short file. Kept per handle: at most 2 GB of device memory (a handle's workspace is mostly the U hand-off
 * tiles, (frame_length + 1) x 256 bytes per wave slot: 1 GB for the 1 024-packet windows of a 24-bit file decoder, and
 * letting go of it means hipFree, which waits for the whole device) and 64 MB of pinned host memory (the largest buffers are
 * dropped first); four handles per device, so at most 8 GB of the device's 288 GB. alacgpu_trim() frees what is kept. */
void alacgpu_destroy(alacgpu_decoder* dec);
void alacgpu_trim(void);

/* Pinned (page-locked) host memory for the buffers a caller hands to alacgpu_decode_batch: what that entry finds in pinned
 * memory it transfers in place instead of through its own staging copies (a third of the time of a file decode goes into
 * those). NULL when the runtime refuses. Freed buffers of up to 64 MB are kept for the next alacgpu_host_alloc (the eight
 * newest, 192 MB in all; alacgpu_trim() frees them). */
void* alacgpu_host_alloc(size_t bytes);
void alacgpu_host_free(void* p);

/* (*PacketDecoder).Format (decoder.go:112). */
int alacgpu_get_format(const alacgpu_decoder* dec, alacgpu_format* fmt);

/* Bytes of one full decoded frame: FrameLength*NumChannels*BytesPerSample (decoder.go:120). */
size_t alacgpu_frame_bytes(const alacgpu_decoder* dec);

/*
 * DecodePacket / decodePacketInto (decoder.go:117,133): one packet, host buffers,
 * through the same HIP kernel as the batch entry (batch of 1). out_cap must be
 * >= alacgpu_frame_bytes(). *out_len = numSamples*numChan*bps (decoder.go:206).
 * Returns ALACGPU_E_DECODE and sets *status_out when the packet fails.
 */
int alacgpu_decode_packet(alacgpu_decoder* dec, const uint8_t* packet, size_t packet_len,
                          uint8_t* out, size_t out_cap, size_t* out_len, int32_t* status_out);

/*
 * DecodePackets, host buffers. Packet i is blob[offsets[i] .. offsets[i+1]) (dense, e.g. a whole mdat).
 * PCM of packet i is written at out + i*out_stride (out_stride >= frame bytes);
 * frames_out[i] = the packet's sample-frame count (0 on failure), status[i] = status word.
 * A failing packet's slot and the bytes of a slot behind a partial frame read as zero (decoder.go:120,127: DecodePacket
 * hands back a prefix of a zeroed frame buffer); a failing packet does not affect others.
 * The batch is cut into chunks that are uploaded, decoded and downloaded on three streams at once; the bytes go to
 * the device as they are. Pageable memory is staged through pinned buffers by a few copy threads
 * (ALACGPU_COPY_THREADS); blob / out / frames_out / status that the caller allocated with hipHostMalloc or registered
 * with hipHostRegister are transferred in place. Blocking: returns when everything is in the caller's buffers.
 * blob_bytes = readable bytes at blob. The offsets are untrusted (a sample table from a file, internal/mp4/mp4.go:382-420):
 * a packet that does not lie inside [0, blob_bytes), or that ends before it starts, is never read and gets status
 * ALACGPU_ERR_RANGE, like in the device entry (alacgpu 0.4: the argument is new; 0.3 read whatever the offsets named).
 */
int alacgpu_decode_batch(alacgpu_decoder* dec, const uint8_t* blob, size_t blob_bytes, const uint64_t* offsets,
                         size_t n_packets, uint8_t* out, size_t out_stride,
                         uint32_t* frames_out, int32_t* status);

/* The same decode on a thread of the library's own: _start returns at once, _wait blocks until the decode is done and
 * returns what alacgpu_decode_batch would have (its error text becomes the waiting thread's alacgpu_last_error). One
 * decode in flight per handle; nothing else may be called on the handle in between (alacgpu_destroy waits by itself), and
 * all buffers stay the caller's to keep alive and untouched until _wait returns. What a read-ahead file decoder needs:
 * window k + 1 is decoded while the caller drains window k (host/stream_decoder.hpp, stream.py, go/alacgpu_decoder.go): the
 * batch form of the reference's Read loop, which decodes the packet it is about to hand out (decode.go:157-186). */
int alacgpu_decode_batch_start(alacgpu_decoder* dec, const uint8_t* blob, size_t blob_bytes, const uint64_t* offsets,
                               size_t n_packets, uint8_t* out, size_t out_stride, uint32_t* frames_out, int32_t* status);
int alacgpu_decode_batch_wait(alacgpu_decoder* dec);

/*
 * DecodePackets, device-resident (the benchmark path). All pointers are device pointers on the handle's device.
 * Packet i is d_blob[d_offsets[i] .. +d_sizes[i]); d_sizes may be NULL, then d_offsets has n_packets+1 entries and
 * packet i ends where packet i+1 starts. blob_bytes = readable bytes at d_blob: a packet that does not lie inside
 * [0, blob_bytes) gets status ALACGPU_ERR_RANGE and is not read. Asynchronous on the handle's stream
 * (alacgpu_stream()) unless sync != 0: the inputs must be complete, or ordered on that stream, before the call, and
 * the outputs are complete after alacgpu_synchronize() (the stream is non-blocking: it does not order against the
 * legacy default stream or torch's current stream by itself). Some of a decode's kernels (those of the irregular
 * packets of 1-2 channel streams) run on a second stream inside the handle; it leaves the handle's stream after the
 * sort and joins it again before the decode's last event, so work a caller orders behind the handle's stream (an
 * event, a copy enqueued on alacgpu_stream()) is ordered behind those kernels as well.
 */
int alacgpu_decode_batch_device(alacgpu_decoder* dec, const uint8_t* d_blob, size_t blob_bytes,
                                const uint64_t* d_offsets, const uint32_t* d_sizes,
                                size_t n_packets, uint8_t* d_out, size_t out_stride,
                                uint32_t* d_frames_out, int32_t* d_status, int sync);

/* Device scratch the handle needs for a batch of n packets. The batch entries grow it on demand — and growing means
 * hipFree + hipMalloc, which wait for the DEVICE to go idle: an alacgpu_decode_batch_device(..., sync = 0) whose batch is
 * larger than any the handle has seen (or reserved) therefore BLOCKS until earlier work on the device is done, although
 * it is documented as asynchronous. Callers that rely on asynchrony reserve for their largest batch first. */
int alacgpu_reserve(alacgpu_decoder* dec, size_t n_packets);

/* Duration of the last decode on this handle in milliseconds: HIP events on the handle's stream around ALL the
 * kernels of the decode, the sort pre-pass included (valid after a sync). Used by bench.py's roofline. */
int alacgpu_last_kernel_ms(alacgpu_decoder* dec, float* ms);

/* Per-decode durations: every decode (all its kernels) is bracketed by a HIP event pair on the handle's
 * stream (ring of 64). alacgpu_kernel_times synchronizes the stream and returns the durations of the
 * min(max_n, launches since reset, 64) most recent launches, oldest first. */
int alacgpu_timing_reset(alacgpu_decoder* dec);
int alacgpu_kernel_times(alacgpu_decoder* dec, float* ms, size_t max_n, size_t* n_out);

/* Diagnostics: who decoded which wave slot of the last device decode on this handle, and when. Four words per wave
 * slot of the launch plan (irregular packets first, then wide keys, then narrow ones, slowest first; n_out counts
 * slots, max_n words): [0] is 0 for slots that no wave pair owns (the irregular ones), else bit 31 | SIMD of the
 * predictor wave << 19 | SIMD of the entropy wave << 17 | CU number << 8 | the pair's arrival number on its CU << 4 |
 * bit 0 set when the pair took the slot after finishing another; [1], [2] the low words of the wave's
 * s_memtime counter (about 2.1 GHz on MI355X) when the pair began and ended the slot; [3] unused. tests/test_gpu_parity.py checks the spread over
 * the CUs, tools/pair_placement.py prints it. */
int alacgpu_pair_placement(alacgpu_decoder* dec, uint32_t* tags, size_t max_n, size_t* n_out);
/* Diagnostics: what the last device decode on this handle dispatched, read back from the launch plan the device built
 * (the host only knows upper bounds when it launches): wave slots by class and the kernel that decoded the narrow
 * regular ones — the four-wave kernel of the handle's sample width or, for 16-bit batches between the rounds, its gated
 * twin (the device decides with the function the host repeats here on the plan's numbers). bench.py names the roofline's
 * kernel with it. Synchronizes the handle's stream. */
typedef struct alacgpu_dispatch {
    uint32_t packets_per_slot;   /* 64, less for small batches */
    uint32_t slots;              /* wave slots of the plan = irregular + wide + narrow */
    uint32_t irregular_slots, wide_slots, narrow_slots;
    uint32_t keys;               /* sort keys present */
    uint32_t gated;              /* 1: the gated twin took the narrow slots */
    uint32_t lanes_per_packet;   /* 2 / 4: predictor waves on that many lanes per packet for the long predictors; 0: none */
    char narrow_kernel[32], wide_kernel[32], irregular_kernels[96]; /* "" when the class is empty */
    uint32_t workgroups_per_cu;  /* four-wave kernels: 4 or 5 of their workgroups share a CU (the LDS footprint of the launch that worked); 0: gated twin / none */
} alacgpu_dispatch;
int alacgpu_last_dispatch(alacgpu_decoder* dec, alacgpu_dispatch* out);
/* Placement relies on the gfx950 layout of two hardware registers read with s_getreg_b32: HW_ID (SIMD [5:4], CU [11:8],
 * shader engine [14:13]) and XCC_ID ([3:0]); the index built from them stays below 512 and a CU the census of
 * alacgpu_create() missed only loses its fixed place in the item order, so a different layout costs speed, not
 * correctness. Measured and tested on an unpartitioned MI355X (SPX: 256 CUs in 8 XCDs) only; on a partitioned device
 * (CPX) the CU count per device and the XCD assumptions of the numbering (item i on XCD i mod 8) are untested. */

/* The handle's hipStream_t as an opaque pointer (for callers that enqueue copies). */
void* alacgpu_stream(alacgpu_decoder* dec);

int alacgpu_synchronize(alacgpu_decoder* dec);

/* Thread-local description of the last ALACGPU_E_HIP / E_ARG / E_CONFIG failure. */
const char* alacgpu_last_error(void);

/* "alacgpu <semver> gfx950" */
const char* alacgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ALACGPU_H */
