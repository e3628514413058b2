#!/bin/bash
# Bench lines of round 3 beyond the headline (device-resident, 1 GPU): the BASELINE configurations, SURVEY §8(f4)'s
# formats, the batch-size sweep around the occupancy edge and the 9-key mix. usage: tools/r3_lines.sh <outdir>
OUT=$1; mkdir -p "$OUT"
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-entry"
$B --packets 4096                                   > $OUT/config_b_4096pkts.json 2>/dev/null
$B --depth 24                                        > $OUT/config_c_24bit_stereo.json 2>/dev/null
$B --depth 24 --channels 8 --packets 16384           > $OUT/config_d_8ch_16384.json 2>/dev/null
$B --depth 24 --channels 8 --packets 65536 --steps 3 > $OUT/config_d_8ch_65536.json 2>/dev/null
$B --depth 20                                        > $OUT/f4_20bit_stereo.json 2>/dev/null
$B --depth 32                                        > $OUT/f4_32bit_stereo_shift2.json 2>/dev/null
$B --depth 32 --channels 1 --profile 5               > $OUT/f4_32bit_mono_shift0_wide.json 2>/dev/null
$B --depth 24 --profile 5                            > $OUT/f4_24bit_stereo_shift0_wide.json 2>/dev/null
$B --channels 1                                      > $OUT/f4_16bit_mono.json 2>/dev/null
$B --channels 6 --packets 21845                      > $OUT/f4_16bit_5_1.json 2>/dev/null
for p in 32768 66000 70000 81920 98304 114688 131072 196608; do $B --packets $p > $OUT/sweep_$p.json 2>/dev/null; done
for p in 66000 98304 131072; do $B --depth 24 --packets $p > $OUT/sweep24_$p.json 2>/dev/null; done
$B --depth 32 --packets 98304                        > $OUT/sweep32_98304.json 2>/dev/null
for p in 1024 64 1; do $B --packets $p              > $OUT/small_$p.json 2>/dev/null; done
$B --profile 4                                       > $OUT/no_order12_65536.json 2>/dev/null
$B --profile 4 --packets 4096                        > $OUT/no_order12_4096.json 2>/dev/null
$B --profile 6                                       > $OUT/mixed_orders_9keys.json 2>/dev/null
$B --profile 6 --packets 70000                       > $OUT/mixed_orders_9keys_70000.json 2>/dev/null
python - "$OUT" <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.load(open(f))
        print("%-40s %9.0f Msamples/s  %8.3f ms/step  kernel %8.3f ms  %6.1f GB/s (%.4f)  bit_exact %s" % (
            os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"],
            d["roofline"]["frac"], d["bit_exact"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
