#!/usr/bin/env python3
"""r4_show.py <raw.txt> — one line per build of the tools/ab_bench.py runs in the file ('# title' lines between them)."""
import json, sys
cur = None
for l in open(sys.argv[1]):
    if l.startswith("#"): cur = l.strip()
    elif l.startswith("{"):
        d = json.loads(l); print("%-44s %-50s median %.4f  min %.4f  %s" % (cur, d["lib"].split("/")[-1], d["median_ms"], d["min_ms"], d["bit_exact"]))
