#!/bin/bash
# r4_side.sh — the irregular packets' kernels beside the decode (ALACGPU_SIDE 0 never / 2 always) at the lowest and at the
# default stream priority, one binary, one device. usage: tools/r4_side.sh <out.txt> <lib.so>
out=$1; lib=$2
{
for p in 4096 65536 98304 131072 196608; do
  echo "# 16-bit packets $p"; python tools/ab_bench.py --packets $p --rounds 3 $lib@ALACGPU_SIDE=0 $lib@ALACGPU_SIDE=2 $lib@ALACGPU_SIDE=2,ALACGPU_SIDE_PRIO=0 2>/dev/null
done
for p in 65536 131072; do
  echo "# 24-bit packets $p"; python tools/ab_bench.py --depth 24 --packets $p --rounds 3 $lib@ALACGPU_SIDE=0 $lib@ALACGPU_SIDE=2 $lib@ALACGPU_SIDE=2,ALACGPU_SIDE_PRIO=0 2>/dev/null
done
} > $out
python - "$out" <<'PY'
import json, sys
cur = None
for l in open(sys.argv[1]):
    if l.startswith("#"): cur = l.strip()
    elif l.startswith("{"):
        d = json.loads(l); print("%-24s %-50s median %.4f  min %.4f  %s" % (cur, d["lib"].split("/")[-1], d["median_ms"], d["min_ms"], d["bit_exact"]))
PY
