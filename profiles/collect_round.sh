#!/bin/bash
# usage (on the GPU box, from the repo root): profiles/collect_round.sh <outdir> [bench.py workload flags, e.g. --depth 24]
# The evidence of one round for the headline workload (bench.py's default command) or for another configuration:
#   kernel_stats.csv       rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-entry`
#   bench_under_rocprof.json   the bench line of that same run
#   pmc_summary.txt        instruction / cycle counters (one --pmc pass per line of groups_duo.txt, never mixed with traces)
#   traffic.json           HBM bytes per launch: FETCH_SIZE x 2 (gfx950 counts 128-B requests as 64) + WRITE_SIZE, stamped with
#                          the sha256 of the kernel sources so that bench.py only quotes it for the binary it was taken from
set -u
OUT=$1; shift; mkdir -p "$OUT"
export TMPDIR=/tmp
WORKLOAD="$*"
CMD="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-entry $WORKLOAD"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $CMD > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
head -40 $(find "$OUT/trace" -name "*kernel_trace.csv" | head -1) > "$OUT/kernel_trace_head.csv"
rm -rf "$OUT/trace"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-entry --no-verify $WORKLOAD > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $grp" >> "$OUT/failed.txt"
done < <(cat profiles/groups_duo.txt profiles/groups_traffic.txt)
python3 - "$OUT" "$WORKLOAD" <<'PY'
import csv, glob, json, os, sys, collections, importlib
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].split("::")[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# per DECODE, not per dispatch: a kernel may be launched more than once per decode (the four-wave kernels: once per LDS footprint,
# one of the launches exits at once; round 4), so a counter's dispatches are summed and divided by the decodes of its pass
decodes = lambda c: max(1, len(agg["alac_classify"].get(c, [])))
with open(os.path.join(out, "pmc_summary.txt"), "w") as w:
    w.write("# per decode: sum over a kernel's dispatches / decodes in the pass (n = dispatches seen)\n")
    for k, d in agg.items():
        if "alac" not in k: continue
        w.write("kernel %s\n" % k)
        for c, v in sorted(d.items()):
            w.write("  %-34s n=%d mean=%.6g\n" % (c, len(v), sum(v) / (decodes(c) if "census" not in k else len(v))))
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("saprobe-alac_amd")
mean = lambda k, c: (sum(agg[k][c]) / decodes(c)) if agg[k].get(c) else 0.0
fetch = sum(mean(k, "FETCH_SIZE") for k in agg if "alac" in k) * 1024   # KB
write = sum(mean(k, "WRITE_SIZE") for k in agg if "alac" in k) * 1024
valu = sum(mean(k, "SQ_INSTS_VALU") for k in agg if "alac" in k and "census" not in k)  # wave-instructions per launch, all kernels of a decode
import json as _j
line = _j.load(open(os.path.join(out, "bench_under_rocprof.json")))
wl = sys.argv[2].split()
arg = lambda k, d: int(wl[wl.index(k) + 1]) if k in wl else d
json.dump({"workload": [arg("--depth", 16), arg("--channels", 2), arg("--frame-length", 4096), arg("--packets", 65536), arg("--profile", 0)],
           "bench_value": line.get("value"), "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"], "csrc_sha256": pkg.csrc_sha256(), "version": pkg.lib().alacgpu_version().decode(),
           "fetch_size_raw_bytes": fetch, "fetch_bytes_x2": 2 * fetch, "write_bytes": write,
           "traffic_bytes_per_launch": int(2 * fetch + write), "valu_insts_per_launch": int(valu),
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, summed over the kernels of one decode; FETCH_SIZE doubled "
                  "(MI355X_MICROARCH.md: gfx950 tallies 128-B requests as 64 B)"}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(open(os.path.join(out, "traffic.json")).read())
PY
rm -rf "$OUT"/pass*/
