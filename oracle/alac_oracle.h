/*
 * alac_oracle.h — CPU restatement of the reference ALAC packet decoder (see alac_oracle.c).
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; never by the product path.
 * Shares only the POD config struct and the status-word encoding with include/alacgpu.h.
 */
#ifndef ALAC_ORACLE_H
#define ALAC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/alacgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct alac_oracle alac_oracle;

/* NewPacketDecoder (decoder.go:90). NULL on unsupported bit depth / channel count. */
alac_oracle* alac_oracle_create(const alacgpu_config* cfg);
void alac_oracle_destroy(alac_oracle* d);
size_t alac_oracle_frame_bytes(const alac_oracle* d);

/* DecodePacket (decoder.go:117): `out` must hold alac_oracle_frame_bytes(); it is zeroed first,
 * like the reference's fresh output buffer. Returns the status word; *frames_out = numSamples. */
int32_t alac_oracle_decode_packet(alac_oracle* d, const uint8_t* packet, size_t len, uint8_t* out,
                                  uint32_t* frames_out);

/* Batch helper (CPU baseline): packet i = blob[offsets[i] .. +sizes[i]) (sizes == NULL:
 * offsets has n+1 entries). out == NULL decodes into a per-thread throw-away buffer.
 * `threads` host threads, contiguous static partition, one decoder state per thread. */
int alac_oracle_decode_batch(const alacgpu_config* cfg, const uint8_t* blob, const uint64_t* offsets,
                             const uint32_t* sizes, size_t n, uint8_t* out, size_t out_stride,
                             uint32_t* frames_out, int32_t* status, int threads);

#ifdef __cplusplus
}
#endif
#endif
