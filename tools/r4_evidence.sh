#!/bin/bash
# Everything under profiles/r04_final that needs the GPU, in parts (run on the GPU box from the repo root):
#   tools/r4_evidence.sh gpurun_out/r04_final [part ...]      parts: head lines file ab configs fuzz (default: all)
# Each step appends to $OUT/progress.txt, so a long run is visibly alive.
set -u
OUT=$1; shift; PARTS=${*:-head lines file ab configs fuzz}
mkdir -p "$OUT/lines"
say() { echo "$(date +%T) $*" | tee -a "$OUT/progress.txt"; }
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has head; then
  say "collect_round (headline: kernel stats, PMC, stamped traffic)"
  bash profiles/collect_round.sh "$OUT" > "$OUT/collect.log" 2>&1 || say "collect_round failed"
  say "bench (default flags; quotes the PMC traffic and VALU count just taken when the stamp matches)"
  mkdir -p profiles/r04_final && cp "$OUT/traffic.json" profiles/r04_final/traffic.json
  python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || say "bench failed"
fi
if has lines; then
  say "lines"
  bash tools/r3_lines.sh "$OUT/lines" > "$OUT/lines.txt" 2>&1 || say "lines failed"
  B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-entry"
  for p in 90000 147456 163840; do $B --packets $p > $OUT/lines/sweep_$p.json 2>/dev/null; done
  for p in 81920 147456; do $B --depth 24 --packets $p > $OUT/lines/sweep24_$p.json 2>/dev/null; done
  $B --depth 32 --packets 81920 > $OUT/lines/sweep32_81920.json 2>/dev/null
  $B --frame-length 16384 --packets 16384 > $OUT/lines/frames_16384.json 2>/dev/null
  python - "$OUT/lines" >> "$OUT/lines.txt" <<'PY'
import json, glob, os, sys
print("# round-4 additions")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    n = os.path.basename(f)[:-5]
    if not any(k in n for k in ("90000", "147456", "163840", "81920", "frames_")): continue
    try:
        d = json.load(open(f))
        print("%-40s %9.0f Msamples/s  %8.3f ms/step  kernel %8.3f ms  %6.1f GB/s (%.4f)  bit_exact %s  %s" % (
            n, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["roofline"]["frac"], d["bit_exact"], d["roofline"]["kernel"]))
    except Exception as e:
        print(n, "FAILED", e)
PY
fi
if has file; then
  say "file bench"
  python tools/file_bench.py > "$OUT/file_bench.jsonl" 2> "$OUT/file_bench.err" || say "file_bench failed"
fi
if has ab; then
  say "A/B against the round-3 binary (profiles/exp_bin/libalacgpu_r3.so) in one process"
  if [ -f profiles/exp_bin/libalacgpu_r3.so ]; then
    bash tools/r4_ab.sh "$OUT/ab_r3_raw.txt" profiles/exp_bin/libalacgpu_r3.so saprobe-alac_amd/csrc/libalacgpu.so > "$OUT/ab_r3.txt" 2>&1 || say "ab failed"
  else say "no round-3 binary"; fi
fi
if has configs; then
  say "config c / d / b, the gated pairs (98 304 x 16-bit) and five workgroups per CU (81 920 x 24-bit): kernel stats, PMC, traffic"
  bash profiles/collect_round.sh "$OUT/config_c" --depth 24 > "$OUT/collect_c.log" 2>&1 || say "collect c failed"
  bash profiles/collect_round.sh "$OUT/config_d" --depth 24 --channels 8 --packets 16384 > "$OUT/collect_d.log" 2>&1 || say "collect d failed"
  bash profiles/collect_round.sh "$OUT/config_b" --packets 4096 > "$OUT/collect_b.log" 2>&1 || say "collect b failed"
  bash profiles/collect_round.sh "$OUT/fit5_81920" --depth 24 --packets 81920 > "$OUT/collect_f.log" 2>&1 || say "collect f failed"
  bash profiles/collect_round.sh "$OUT/gated_98304" --packets 98304 > "$OUT/collect_g.log" 2>&1 || say "collect g failed"
  say "kernel timeline of config d"
  bash tools/kernel_timeline.sh > "$OUT/timeline_config_d.txt" 2>&1 || say "timeline failed"
  say "instruction counts per sample"
  python tools/pmc_per_sample.py "$OUT" > "$OUT/pmc_per_sample.txt" 2>&1 || say "pmc_per_sample failed"
fi
if has fuzz; then
  say "GPU fuzz (randomized parity sweep, HIP path vs oracle through the C ABI)"
  timeout -k 10 ${FUZZ_SECONDS:-420} python tools/gpu_fuzz.py ${FUZZ_ROUNDS:-1500} ${FUZZ_SEED:-44} > "$OUT/gpu_fuzz.log" 2>&1 || say "gpu_fuzz failed or timed out (see gpu_fuzz.log)"
fi
say "done"
