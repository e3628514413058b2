#!/bin/bash
# r4_low5.sh — batches of 4..5 x CUs slots in the "fit 5" shape: the CU's fifth workgroup at the lowest issue priority (default)
# against equal priorities (ALACGPU_LOW5=0); 16-bit stereo also against the gated pairs. usage: tools/r4_low5.sh <out_raw.txt> <lib.so>
out=$1; lib=$2
{
for p in 65536 66000 70000 75000 81920; do
  echo "# 16-bit packets $p, no gated pairs"; ALACGPU_PAIR_CAP=4 python tools/ab_bench.py --packets $p --rounds 3 $lib@ALACGPU_LOW5=0 $lib 2>/dev/null
  echo "# 16-bit packets $p, as shipped"; python tools/ab_bench.py --packets $p --rounds 3 $lib 2>/dev/null
done
for p in 66000 75000 81920; do
  echo "# 24-bit packets $p"; python tools/ab_bench.py --depth 24 --packets $p --rounds 3 $lib@ALACGPU_LOW5=0 $lib 2>/dev/null
done
echo "# 32-bit packets 81920"; python tools/ab_bench.py --depth 32 --packets 81920 --rounds 3 $lib@ALACGPU_LOW5=0 $lib 2>/dev/null
echo "# 16-bit mono packets 81920"; python tools/ab_bench.py --channels 1 --packets 81920 --rounds 3 $lib@ALACGPU_LOW5=0 $lib 2>/dev/null
echo "# 16-bit 9 keys packets 70000"; ALACGPU_PAIR_CAP=4 python tools/ab_bench.py --profile 6 --packets 70000 --rounds 3 $lib@ALACGPU_LOW5=0 $lib 2>/dev/null
} > $out
python tools/r4_show.py $out
