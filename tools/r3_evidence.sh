#!/bin/bash
# Everything under profiles/r03_final that needs the GPU, in one go (run on the GPU box from the repo root):
#   tools/r3_evidence.sh gpurun_out/r03_final
# Each step appends to $OUT/progress.txt, so a long run is visibly alive.
set -u
OUT=$1; mkdir -p "$OUT/lines" "$OUT/placement"
say() { echo "$(date +%T) $*" | tee -a "$OUT/progress.txt"; }
say "collect_round"
bash profiles/collect_round.sh "$OUT" > "$OUT/collect.log" 2>&1 || say "collect_round failed"
say "bench (default flags; quotes the PMC traffic just taken: bench.py reads profiles/r*/traffic.json when its stamp matches)"
mkdir -p profiles/r03_final && cp "$OUT/traffic.json" profiles/r03_final/traffic.json
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || say "bench failed"
say "lines"
bash tools/r3_lines.sh "$OUT/lines" > "$OUT/lines.txt" 2>&1 || say "lines failed"
say "file bench"
python tools/file_bench.py > "$OUT/file_bench.jsonl" 2> "$OUT/file_bench.err" || say "file_bench failed"
say "placement"
for P in 70000 98304; do
  timeout -k 10 200 python tools/pair_placement.py --packets $P > "$OUT/placement/pairs_$P.txt" 2>/dev/null || say "placement $P failed"
done
say "A/B against the round-2 binary"
for P in 65536 70000 98304; do
  timeout -k 10 200 python tools/ab_bench.py --packets $P profiles/exp_bin/libalacgpu_r2.so saprobe-alac_amd/csrc/libalacgpu.so 2>/dev/null >> "$OUT/ab_16bit_${P}.txt" || say "ab $P failed"
done
timeout -k 10 200 python tools/ab_bench.py --depth 24 profiles/exp_bin/libalacgpu_r2.so saprobe-alac_amd/csrc/libalacgpu.so 2>/dev/null >> "$OUT/ab_24bit_65536.txt" || say "ab 24 failed"
say "config c / d / b and the gated pair kernel: kernel stats, PMC, traffic"
bash profiles/collect_round.sh "$OUT/config_c" --depth 24 > "$OUT/collect_c.log" 2>&1 || say "collect c failed"
bash profiles/collect_round.sh "$OUT/config_d" --depth 24 --channels 8 --packets 16384 > "$OUT/collect_d.log" 2>&1 || say "collect d failed"
bash profiles/collect_round.sh "$OUT/config_b" --packets 4096 > "$OUT/collect_b.log" 2>&1 || say "collect b failed"
bash profiles/collect_round.sh "$OUT/gated_98304" --packets 98304 > "$OUT/collect_g.log" 2>&1 || say "collect g failed"
say "kernel timeline of config d"
bash tools/kernel_timeline.sh > "$OUT/timeline_config_d.txt" 2>&1 || say "timeline failed"
say "instruction counts per sample"
python tools/pmc_per_sample.py "$OUT" > "$OUT/pmc_per_sample.txt" 2>&1 || say "pmc_per_sample failed"
say "GPU fuzz (randomized parity sweep, HIP path vs oracle through the C ABI)"
timeout -k 10 420 python tools/gpu_fuzz.py ${FUZZ_ROUNDS:-1500} 33 > "$OUT/gpu_fuzz.log" 2>&1 || say "gpu_fuzz failed or timed out (see gpu_fuzz.log)"
say "done"
