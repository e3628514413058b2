#!/bin/bash
# usage: profiles/pmc_run.sh <outdir> <bench args...>   (run on the GPU box from the repo root)
# One rocprofv3 --pmc pass per counter group (counters in their own runs, no trace domains mixed in).
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-verify "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $grp" >> "$OUT/failed.txt"
done <<'GROUPS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU
SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
GRBM_GUI_ACTIVE TA_BUSY_avr
GROUPS
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as w:
    for k, d in agg.items():
        w.write("kernel %s\n" % k)
        for c, v in sorted(d.items()):
            w.write("  %-34s n=%d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(os.path.join(out, "summary.txt")).read())
PY
