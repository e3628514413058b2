/*
 * k_dec32q.hip — the four-wave workgroups of k_dec16q.hip (entropy, predictor, writer, spare / second predictor wave) for
 * 32-bit streams with their usual shift bytes, chanBits <= 23: every batch size (these widths have no gated twin: their writers
 * need more registers than three waves per SIMD leave).
 */
/* 26 KB of static LDS: five of these workgroups fit a CU ("fit 5"); launched with a dynamic-LDS pad for four (alac_gpu.h: decode_mode) */
#define ALAC_LDS_ROWS 32
#define ALAC_LDS_FLUSH 32
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode_32q
#define ALAC_DECODE_WIDE 0
#define ALAC_DECODE_DEPTH 32
#define ALAC_DECODE_GATED 0
#define ALAC_DECODE_ROLES 4
#define ALAC_DECODE_WAVES 4 /* __launch_bounds__: waves per SIMD the register budget must allow */

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
