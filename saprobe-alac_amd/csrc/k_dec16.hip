/*
 * k_dec16.hip — the wave pair of alac_duo.h for one class of regular packets: 16-bit samples, chanBits <= 23 (one translation
 * unit of libalacgpu.so, see alac_gpu.h; the kernel body is k_decode_body.inc). The compiler sizes a kernel by its
 * largest variant, so the sample widths are separate kernels — a handle only ever launches the ones of its own
 * width — and the units compile in parallel.
 */
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode_16
#define ALAC_DECODE_WIDE 0
#define ALAC_DECODE_DEPTH 16
#define ALAC_DECODE_GATED 0
#define ALAC_DECODE_WAVES 2 /* __launch_bounds__: waves per SIMD the register budget must allow */
#define ALAC_DECODE_SPLIT3 2 /* batches of up to 4 x CUs wave slots belong to alac_decode_16q (k_dec16q.hip) */

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
