#!/bin/bash
# r4_mono.sh — mono streams between the rounds: gated pairs (default for 16-bit) against four / five four-wave workgroups per CU
out=$1; lib=$2
{
for p in 70000 81920 98304 114688; do
  echo "# 16-bit mono packets $p"; python tools/ab_bench.py --channels 1 --packets $p --rounds 3 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib 2>/dev/null
done
for p in 70000 81920 98304; do
  echo "# 16-bit stereo packets $p"; python tools/ab_bench.py --packets $p --rounds 3 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib 2>/dev/null
done
} > $out
python tools/r4_show.py $out
