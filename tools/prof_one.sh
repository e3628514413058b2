#!/bin/bash
# prof_one.sh LIB TAG [extra ab_bench args] — kernel stats of one build on the benchmark batch (run on the GPU box)
set -e
LIB=$1; TAG=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof1/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o out -- python3 $GRAFT_REPO_ROOT/tools/ab_bench.py --rounds 3 "$@" $GRAFT_REPO_ROOT/$LIB > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no stats file"; find /tmp/prof_$TAG | head; exit 1; }
cp "$f" $OUT/kernel_stats.csv
cut -d, -f1-5 $OUT/kernel_stats.csv | cut -c1-110 | head -14
grep median $OUT/log.txt | cut -c30-140
