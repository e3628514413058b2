#!/bin/bash
# r4_fit.sh — four or five four-wave workgroups per CU (alac_gpu.h: quad_fit; ALACGPU_FIT 4 / 5 forces one) and, for 16-bit,
# the gated pairs, over the batch sizes around the rounds; the round-4 build before the 26 KB stager beside it.
# usage: tools/r4_fit.sh <out.txt> <lib.so> [<older lib.so>]
out=$1; lib=$2; old=${3:-}
{
for p in 65536 66000 70000 81920 90000 98304 114688 131072 132000 147456 163840; do
  echo "# 16-bit packets $p, no gated pairs"; ALACGPU_PAIR_CAP=4 python tools/ab_bench.py --packets $p --rounds 3 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib 2>/dev/null
  echo "# 16-bit packets $p, as shipped"; python tools/ab_bench.py --packets $p --rounds 3 $lib $old 2>/dev/null
done
for p in 65536 66000 81920 98304 131072 147456; do
  echo "# 24-bit packets $p"; python tools/ab_bench.py --depth 24 --packets $p --rounds 3 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib $old 2>/dev/null
done
for p in 65536 81920; do
  echo "# 32-bit packets $p"; python tools/ab_bench.py --depth 32 --packets $p --rounds 3 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib $old 2>/dev/null
done
echo "# 16-bit mono 81920"; python tools/ab_bench.py --channels 1 --packets 81920 --rounds 3 $lib@ALACGPU_FIT=4 $lib@ALACGPU_FIT=5 $lib $old 2>/dev/null
} > $out
python - "$out" <<'PY'
import json, sys
cur = None
for l in open(sys.argv[1]):
    if l.startswith("#"): cur = l.strip()
    elif l.startswith("{"):
        d = json.loads(l); print("%-44s %-50s median %.4f  min %.4f  %s" % (cur, d["lib"].split("/")[-1], d["median_ms"], d["min_ms"], d["bit_exact"]))
PY
