#!/bin/bash
# r4_order.sh — what an empty grid in front of the launch that works costs (alacgpu.hip: launch, the narrow slots' launches):
# ALACGPU_FIRST = 4 / 5 / 6 puts the "fit 4" / "fit 5" / gated launch first, whichever of them the device then picks.
# usage: tools/r4_order.sh <out_raw.txt> <lib.so>
out=$1; lib=$2
{
for p in 65536 81920 98304 114688 131072; do
  echo "# 16-bit packets $p"; python tools/ab_bench.py --packets $p --rounds 3 $lib@ALACGPU_FIRST=4 $lib@ALACGPU_FIRST=5 $lib@ALACGPU_FIRST=6 $lib 2>/dev/null
done
for p in 65536 81920 98304 131072; do
  echo "# 24-bit packets $p"; python tools/ab_bench.py --depth 24 --packets $p --rounds 3 $lib@ALACGPU_FIRST=4 $lib@ALACGPU_FIRST=5 $lib 2>/dev/null
done
} > $out
python tools/r4_show.py $out
