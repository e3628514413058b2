/*
 * k_dec16q.hip — 16-bit streams, chanBits <= 23: FOUR-wave workgroups (one translation unit of libalacgpu.so, see
 * alac_gpu.h; the kernel body is k_decode_body.inc). The compiler sizes a kernel by its largest variant, so the sample
 * widths are separate kernels — a handle only ever launches the ones of its own width — and the units compile in parallel.
 *
 * Waves: entropy, predictor, writer (alac_duo.h: EC — unmix, packing, LDS stager and flush in a wave of their own: what a
 * workgroup's step takes is the issue time of its longest wave, and with the writer split off the predictor wave's last
 * phase is as short as its U phase) and a spare one. For keys whose longer predictor has PairArgs::lanes_min taps or
 * more, on a device with room (a CU per workgroup), the spare wave is a second predictor wave: each of the two holds 32
 * of the 64 packets, a packet's taps spread over two lanes (alac_duo.h: duo_phase_lanes). A predictor step costs eight
 * instructions per tap and a batch ends with its slowest workgroup: the 5 % of the benchmark's packets with twelve taps
 * cost BASELINE config b 28 % (4 096 packets: 2.17 -> 1.56 ms) and a lone packet as much. For every other key, and on a
 * fuller device, the spare wave exits at once. Which wave of a workgroup is the spare one rotates per CU
 * (k_decode_body.inc), or the same SIMD of every CU would stand empty.
 * It takes every batch the gated twin (k_dec16g.hip) does not: rounds of four workgroups per CU, as many as it takes.
 * (Until the end of round 3 wave PAIRS took everything beyond one round, and every width but 16 bits at any size: with
 * three-wave workgroups a second round lost more than it gained, 131 072 packets 4.40 ms with pairs and 5.32 with three
 * waves, and the writers of the wider samples could not keep up alone. With the spare wave rotating over the SIMDs and
 * the predictor tap free of wait states it is the other way round: 131 072 packets 4.78 -> 4.00 ms; 24-bit stereo
 * 65 536 packets 3.32 -> 2.57, 98 304 4.94 -> 4.45, 131 072 5.70 -> 4.74; 32-bit stereo 4.85 -> 4.26, 98 304 8.51 -> 7.66;
 * 24-bit mono 2.23 -> 1.91; A/B in one process. The two-wave kernels of the narrow keys are gone.)
 */
/* 26 KB of static LDS: five of these workgroups fit a CU ("fit 5"); launched with a dynamic-LDS pad for four (alac_gpu.h: decode_mode) */
#define ALAC_LDS_ROWS 32
#define ALAC_LDS_FLUSH 32
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode_16q
#define ALAC_DECODE_WIDE 0
#define ALAC_DECODE_DEPTH 16
#define ALAC_DECODE_GATED 0
#define ALAC_DECODE_ROLES 4
#define ALAC_DECODE_WAVES 4 /* __launch_bounds__: waves per SIMD the register budget must allow */

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
