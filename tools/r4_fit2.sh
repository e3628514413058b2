#!/bin/bash
# r4_fit2.sh — four against five four-wave workgroups per CU (ALACGPU_FIT) and what decode_mode() picks, at the slot counts per CU
# the first sweep (tools/r4_fit.sh) left out. usage: tools/r4_fit2.sh <out_raw.txt> <lib.so>
out=$1; lib=$2
{
for p in 180224 196608 212992 262144; do
  echo "# 16-bit packets $p"; python tools/ab_bench.py --packets $p --rounds 2 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib 2>/dev/null
done
for p in 90000 106496 114688 122880 163840 196608; do
  echo "# 24-bit packets $p"; python tools/ab_bench.py --depth 24 --packets $p --rounds 2 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib 2>/dev/null
done
for p in 98304 131072; do
  echo "# 32-bit packets $p"; python tools/ab_bench.py --depth 32 --packets $p --rounds 2 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib 2>/dev/null
done
for p in 81920 131072; do
  echo "# 16-bit mono packets $p"; python tools/ab_bench.py --channels 1 --packets $p --rounds 2 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib 2>/dev/null
done
} > $out
python tools/r4_show.py $out
