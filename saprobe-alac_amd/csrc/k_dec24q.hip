/*
 * k_dec24q.hip — the four-wave workgroups of k_dec16q.hip (entropy, predictor, writer, spare / second predictor wave) for
 * 20- and 24-bit streams (3-byte samples), mono and stereo, chanBits <= 23: every batch size (these widths have no gated twin: their writers
 * need more registers than three waves per SIMD leave).
 */
/* 30 KB of static LDS: five of these workgroups fit a CU ("fit 5"); launched with a dynamic-LDS pad for four (alac_gpu.h:
 * decode_mode). Stager rows of 48 dwords = eight six-dword groups of the 3-byte pair writer, flushed as whole 128-byte lines
 * (rows of 32 dwords flushed as 64-byte pieces: 24-bit stereo 65 536 packets 2.44 -> 2.54 ms) */
#define ALAC_LDS_ROWS 48
#define ALAC_LDS_FLUSH 32
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode_24q
#define ALAC_DECODE_WIDE 0
#define ALAC_DECODE_DEPTH 24
#define ALAC_DECODE_GATED 0
#define ALAC_DECODE_ROLES 4
#define ALAC_DECODE_WAVES 4 /* __launch_bounds__: waves per SIMD the register budget must allow */

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
