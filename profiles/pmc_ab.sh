#!/bin/bash
# usage: profiles/pmc_ab.sh <outdir> <groups-file> <lib.so> [ab_bench args...] — one rocprofv3 --pmc pass per line of the
# groups file over tools/ab_bench.py with ONE build of the library (never mixed with trace domains)
set -u
OUT=$1; GROUPS_FILE=$2; LIB=$3; shift 3
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 "$GRAFT_REPO_ROOT/tools/ab_bench.py" --rounds 1 --steps 2 "$@" "$LIB" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $grp" >> "$OUT/failed.txt"
done < "$GROUPS_FILE"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as w:
    for k, d in agg.items():
        if "alac" not in k: continue
        w.write("kernel %s\n" % k)
        for c, v in sorted(d.items()):
            w.write("  %-34s n=%d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
PY
