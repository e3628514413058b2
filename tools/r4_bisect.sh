#!/bin/bash
# r4_bisect.sh — the benchmark batch (and two others) on the round-3 binary, this round's commits that touched the kernels,
# the final build and its variants (all-asm Golomb blocks, s_setprio levels), taking turns in one process on one device.
# usage: tools/r4_bisect.sh <out_raw.txt>   (the libraries are expected under profiles/exp_bin/)
out=$1; E=profiles/exp_bin; L=saprobe-alac_amd/csrc/libalacgpu.so
{
echo "# headline: 65536 x 16-bit stereo"; python tools/ab_bench.py --rounds 6 $E/libalacgpu_r3.so $E/libalacgpu_f7fab5e.so $E/libalacgpu_4488fd7.so $E/libalacgpu_d0d8209.so $E/libalacgpu_r4head.so $L $E/libalacgpu_golasm.so $E/libalacgpu_pa3.so $E/libalacgpu_pa3c2.so $E/libalacgpu_pa3b21.so $E/libalacgpu_pa3b210c1.so 2>/dev/null
echo "# 32768 x 16-bit stereo"; python tools/ab_bench.py --packets 32768 --rounds 4 $E/libalacgpu_r3.so $L $E/libalacgpu_golasm.so $E/libalacgpu_pa3.so $E/libalacgpu_pa3c2.so $E/libalacgpu_pa3b21.so $E/libalacgpu_pa3b210c1.so 2>/dev/null
echo "# config b: 4096 x 16-bit stereo"; python tools/ab_bench.py --packets 4096 --rounds 4 $E/libalacgpu_r3.so $L $E/libalacgpu_golasm.so $E/libalacgpu_pa3.so $E/libalacgpu_pa3c2.so 2>/dev/null
echo "# config c: 65536 x 24-bit stereo"; python tools/ab_bench.py --depth 24 --rounds 4 $E/libalacgpu_r3.so $E/libalacgpu_r4head.so $L $E/libalacgpu_golasm.so 2>/dev/null
echo "# config d: 16384 x 24-bit 8-ch"; python tools/ab_bench.py --depth 24 --channels 8 --packets 16384 --rounds 3 $L $E/libalacgpu_golasm.so 2>/dev/null
echo "# 1 packet"; python tools/ab_bench.py --packets 1 --rounds 4 $L $E/libalacgpu_golasm.so 2>/dev/null
} > $out
python tools/r4_show.py $out
