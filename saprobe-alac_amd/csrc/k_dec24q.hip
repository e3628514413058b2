/*
 * k_dec24q.hip — the four-wave workgroups of k_dec16q.hip (entropy, predictor, writer, spare / second predictor wave) for
 * 20- and 24-bit pairs (3-byte samples, chanBits <= 23; see k_dec24.hip): 20-bit pairs up to one round (4 x CUs wave
 * slots), 24-bit pairs with shift bytes while the device is half full at most (k_decode_body.inc: three_waves).
 */
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode_24q
#define ALAC_DECODE_WIDE 0
#define ALAC_DECODE_DEPTH 24
#define ALAC_DECODE_GATED 0
#define ALAC_DECODE_ROLES 4
#define ALAC_DECODE_WAVES 4 /* __launch_bounds__: waves per SIMD the register budget must allow */
#define ALAC_DECODE_SPLIT3 1 /* only the batches three_waves() names */

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
