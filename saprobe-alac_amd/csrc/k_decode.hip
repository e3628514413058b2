/*
 * k_decode.hip — the wave pair of alac_duo.h over regular packets with chanBits <= 23 (one translation unit of libalacgpu.so, see
 * alac_gpu.h; the kernel body is k_decode_body.inc, compiled once per half of the keys so that the two halves build
 * in parallel).
 */
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode
#define ALAC_DECODE_WIDE 0

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
