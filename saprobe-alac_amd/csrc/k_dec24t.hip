/*
 * k_dec24t.hip — the 24-bit (see k_dec24.hip) wave pair with a THIRD wave per workgroup that writes the PCM (alac_duo.h: EC): entropy,
 * predictor and writer waves, 4 workgroups x 3 waves per CU. What a workgroup's step takes is the issue time of its
 * longest wave; with the writer (unmix, packing, LDS stager, flush) in a wave of its own the predictor wave's last
 * phase is as short as its U phase (65 536 stereo packets: 2.43 -> 2.30 ms, mono 1.39 -> 1.20 ms). It takes the batches
 * that fit one round (up to 4 x CUs wave slots); larger ones stay with the two-wave kernels (k_dec16.hip, k_dec16g.hip):
 * in a second round the dispatcher no longer lands one wave of each role on every SIMD, and three-wave workgroups then
 * lose more than they gain (131 072 packets: 4.40 ms with two waves, 5.32 ms with three).
 */
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode_24t
#define ALAC_DECODE_WIDE 0
#define ALAC_DECODE_DEPTH 24
#define ALAC_DECODE_GATED 0
#define ALAC_DECODE_ROLES 3
#define ALAC_DECODE_WAVES 3 /* __launch_bounds__: waves per SIMD the register budget must allow */
#define ALAC_DECODE_SPLIT3 1 /* only batches of up to 4 x CUs wave slots */

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
