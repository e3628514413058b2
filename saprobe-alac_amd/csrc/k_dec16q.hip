/*
 * k_dec16q.hip — 16-bit streams, batches of up to one round (4 x CUs wave slots): FOUR-wave workgroups (one translation
 * unit of libalacgpu.so, see alac_gpu.h; the kernel body is k_decode_body.inc).
 *
 * Waves: entropy, predictor, writer (alac_duo.h: EC — unmix, packing, LDS stager and flush in a wave of their own: what a
 * workgroup's step takes is the issue time of its longest wave, and with the writer split off the predictor wave's last
 * phase is as short as its U phase: 65 536 stereo packets 2.43 -> 2.30 ms, mono 1.39 -> 1.20 ms) and a spare one. For keys
 * whose longer predictor has PairArgs::lanes_min taps or more, on a device with room (a CU per workgroup), the spare wave
 * is a second predictor wave: each of the two holds 32 of the 64 packets, a packet's taps spread over two lanes
 * (alac_duo.h: duo_phase_lanes). A predictor step costs ten instructions per tap and a batch ends with its slowest
 * workgroup: the 5 % of the benchmark's packets with twelve taps cost BASELINE config b 28 % (4 096 packets: 2.17 -> 1.56
 * ms) and a lone packet as much. For every other key, and on a fuller device, the spare wave exits at once. Which wave of
 * a workgroup is the spare one rotates per CU (k_decode_body.inc), or the same SIMD of every CU would stand empty.
 * Larger batches stay with the two-wave kernels (k_dec16.hip, k_dec16g.hip): in a second round the dispatcher no longer
 * lands one wave of each role on every SIMD, and workgroups of three lose more than they gain (131 072 packets: 4.40 ms
 * with two waves, 5.32 ms with three).
 */
#include "alac_gpu.h"

#define ALAC_DECODE_KERNEL alac_decode_16q
#define ALAC_DECODE_WIDE 0
#define ALAC_DECODE_DEPTH 16
#define ALAC_DECODE_GATED 0
#define ALAC_DECODE_ROLES 4
#define ALAC_DECODE_WAVES 4 /* __launch_bounds__: waves per SIMD the register budget must allow */
#define ALAC_DECODE_SPLIT3 1 /* only batches of up to 4 x CUs wave slots */

namespace alack {

#include "k_decode_body.inc"

} /* namespace alack */
