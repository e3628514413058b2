/*
 * alac_synth.h — synthetic ALAC packet generator: encoder (exact inverse of the decode path)
 * + seeded signal source. Host-only tool that makes benchmark and test inputs; see alac_synth.c.
 */
#ifndef ALAC_SYNTH_H
#define ALAC_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#include "../../include/alacgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ALAC_SYNTH_COEF_WARM = 0, ALAC_SYNTH_COEF_RANDOM = 1, ALAC_SYNTH_COEF_GIVEN = 2 };
enum {
    ALAC_SYNTH_PROFILE_MUSIC = 0,  /* throughput distribution, SURVEY.md §8d */
    ALAC_SYNTH_PROFILE_NOISE = 1,  /* full-scale white noise: escapes */
    ALAC_SYNTH_PROFILE_QUIET = 2,  /* tiny residuals + silence: zero runs */
    ALAC_SYNTH_PROFILE_STRESS = 3, /* parity-only: random everything */
    ALAC_SYNTH_PROFILE_MUSIC_LE8 = 4, /* MUSIC without the 5 % order-12 packets (kernel experiments) */
    ALAC_SYNTH_PROFILE_MUSIC_NOSHIFT = 5, /* MUSIC with bytesShifted 0 at 24/32 bits: wide channels (chanBits 24..33) */
    ALAC_SYNTH_PROFILE_MUSIC_MIXED = 6    /* MUSIC with independent orders 4..6 per channel (9 sort keys per batch) */
};
enum {
    ALAC_SYNTH_FLAG_LEADING_FIL = 1, /* FIL element before the first audio element */
    ALAC_SYNTH_FLAG_MID_DSE = 2,     /* DSE element before the second audio element (first if only one) */
    ALAC_SYNTH_FLAG_NO_END = 4       /* omit the END tag (decoder.go:200-202 stops on channel count) */
};

/* Encoder settings of one SCE/LFE/CPE element. */
typedef struct alac_synth_elem {
    uint8_t order_u, order_v; /* numActive 0..31 (31 = delta mode, 0 = copy) */
    uint8_t den_shift;        /* 0..15 */
    uint8_t mode_u, mode_v;   /* 0..15; != 0 adds the delta pass (decoder.go:307-309) */
    uint8_t pb_factor;        /* 0..7 */
    uint8_t mix_bits;         /* 0..255 */
    int8_t mix_res;           /* -128..127 */
    uint8_t bytes_shifted;    /* 0..2 */
    uint8_t force_escape;
    uint8_t never_escape;
    uint8_t coef_mode;        /* ALAC_SYNTH_COEF_* */
    uint8_t partial;          /* set the partial-frame flag even for a full frame */
    uint8_t pad0[3];
    int16_t coefs_u[32], coefs_v[32]; /* ALAC_SYNTH_COEF_GIVEN */
    uint64_t seed;                    /* ALAC_SYNTH_COEF_RANDOM */
} alac_synth_elem;

int alac_synth_num_elements(int num_channels);

/* pcm: int32 [num_frames][num_channels] in OUTPUT (SMPTE) channel order, values in the PCM domain of
 * cfg->bit_depth. elems: alac_synth_num_elements() entries in bitstream order. Returns packet bytes (0 on
 * overflow / bad arguments). */
size_t alac_synth_encode_packet(const alacgpu_config* cfg, const alac_synth_elem* elems, const int32_t* pcm,
                                uint32_t num_frames, uint32_t flags, uint8_t* out, size_t out_cap);

void alac_synth_signal(const alacgpu_config* cfg, int profile, uint64_t seed, uint32_t num_frames, int32_t* pcm);
void alac_synth_params(const alacgpu_config* cfg, int profile, uint64_t seed, alac_synth_elem* elems,
                       uint32_t* num_frames, uint32_t* flags);
void alac_synth_pack_pcm(const alacgpu_config* cfg, const int32_t* pcm, uint32_t num_frames, uint8_t* out);

size_t alac_synth_slot_bytes(const alacgpu_config* cfg);

/* Packets first_index .. first_index+n-1 of the stream (base_seed, profile): packet i is written to
 * slots + i*slot_bytes with sizes[i] bytes and frames[i] sample frames; pcm_out (may be NULL) receives the
 * source PCM as the decoder must reproduce it, at pcm_out + i*pcm_stride. */
int alac_synth_gen_batch(const alacgpu_config* cfg, int profile, uint64_t base_seed, size_t first_index, size_t n,
                         uint8_t* slots, size_t slot_bytes, uint32_t* sizes, uint32_t* frames, uint8_t* pcm_out,
                         size_t pcm_stride, int threads);

/* In-place pack into the device blob layout (16-byte aligned packets, >= pad zero bytes after each). */
size_t alac_synth_compact(uint8_t* slots, size_t slot_bytes, const uint32_t* sizes, size_t n, size_t pad,
                          uint64_t* offsets);

#ifdef __cplusplus
}
#endif
#endif
