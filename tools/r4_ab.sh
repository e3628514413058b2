#!/bin/bash
# r4_ab.sh — A/B of builds of the library on one device (tools/ab_bench.py): the benchmark batch, BASELINE configs b, c
# and d, mono, a batch between the rounds, one packet. usage: tools/r4_ab.sh <out.txt> <libA.so> <libB.so> ...
out=$1; shift
{
echo "# headline: 65536 x 16-bit stereo"; python tools/ab_bench.py --rounds 5 "$@"
echo "# config b: 4096 x 16-bit stereo"; python tools/ab_bench.py --packets 4096 --rounds 5 "$@"
echo "# config c: 65536 x 24-bit stereo"; python tools/ab_bench.py --depth 24 --rounds 4 "$@"
echo "# config d: 16384 x 24-bit 8-ch"; python tools/ab_bench.py --depth 24 --channels 8 --packets 16384 --rounds 4 "$@"
echo "# 16-bit mono 65536"; python tools/ab_bench.py --channels 1 --rounds 4 "$@"
echo "# 98304 x 16-bit stereo"; python tools/ab_bench.py --packets 98304 --rounds 4 "$@"
echo "# 1 packet"; python tools/ab_bench.py --packets 1 --rounds 5 "$@"
} > $out 2>&1
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"): print(l.strip())
    elif l.startswith("{"):
        d = json.loads(l); print("   %-24s median %.4f  min %.4f  bit_exact %s" % (d["lib"].split("/")[-1], d["median_ms"], d["min_ms"], d["bit_exact"]))
PY
