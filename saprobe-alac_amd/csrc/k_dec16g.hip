/*
 * k_dec16g.hip — the wave pair of alac_duo.h for one class of regular packets: 16-bit samples, chanBits <= 23 (one translation
 * unit of libalacgpu.so, see alac_gpu.h; the kernel body is k_decode_body.inc). The compiler sizes a kernel by its
 * largest variant, so the sample widths are separate kernels — a handle only ever launches the ones of its own
 * width — and the units compile in parallel.
 */
/* the gated kind (k_decode_body.inc): 26 KB of LDS per pair (64-byte PCM pieces) and at most
 * 168 registers, so that six pairs share a CU */
#define ALAC_LDS_ROWS 32
#define ALAC_TAP_ORDER 1 /* alac_regular.h: predict_narrow_core; measured per kernel */
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode_16g
#define ALAC_DECODE_WIDE 0
#define ALAC_DECODE_DEPTH 16
#define ALAC_DECODE_GATED 1
#define ALAC_DECODE_WAVES 3 /* __launch_bounds__: waves per SIMD the register budget must allow */

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
