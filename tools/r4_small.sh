#!/bin/bash
# r4_small.sh — small batches, two builds on one device: 1 .. 16384 packets (tools/ab_bench.py) and the single packets of
# tools/r4_single.py. usage: tools/r4_small.sh <outdir> <libA.so> <libB.so>
out=$1; A=$2; B=$3
mkdir -p $out
python tools/r4_single.py 0 64 > $out/single_packets.txt 2>&1
for p in 1 64 1024 4096 8192 12000 16384 24000; do
  echo "# packets $p"; python tools/ab_bench.py --packets $p --rounds 3 $A $B 2>/dev/null
done > $out/small_batches.txt
python - $out/small_batches.txt <<'PY'
import json, sys
cur = None
for l in open(sys.argv[1]):
    if l.startswith("#"): cur = l.strip()
    elif l.startswith("{"):
        d = json.loads(l); print("%-16s %-22s median %.4f  min %.4f  bit_exact %s" % (cur, d["lib"].split("/")[-1], d["median_ms"], d["min_ms"], d["bit_exact"]))
PY
