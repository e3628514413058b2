#!/bin/bash
# r4_lanes.sh — from how many taps on the second predictor wave (two lanes per packet, alac_duo.h: duo_phase_lanes) pays
# in small batches: ALACGPU_LANES_MIN x batch size, one device. usage: tools/r4_lanes.sh <out.txt> <lib.so>
out=$1; lib=$2
{
for lm in 9 7 5 4 3; do
  for p in 1 64 1024 4096 16384 24000; do
    echo "# lanes_min $lm packets $p"; ALACGPU_LANES_MIN=$lm python tools/ab_bench.py --packets $p --rounds 3 $lib
  done
done
} > $out 2>&1
python - "$out" <<'PY'
import json, sys
cur = None
for l in open(sys.argv[1]):
    if l.startswith("#"): cur = l.strip()
    elif l.startswith("{"):
        d = json.loads(l); print("%-34s median %.4f  min %.4f  bit_exact %s" % (cur, d["median_ms"], d["min_ms"], d["bit_exact"]))
PY
