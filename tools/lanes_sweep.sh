#!/bin/bash
# 16-bit stereo: the four-wave kernel with two-lane predictor waves from `lanes_min` taps on (17: never = three-wave kernel)
B="timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-entry"
for P in ${PACKETS:-65536 4096}; do
 for lm in ${LANES:-17 9 8 7 6 5}; do
  ALACGPU_LANES_MIN=$lm $B --packets $P $EXTRA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('packets $P lanes_min $lm: %.3f ms bit_exact %s' % (d['roofline']['kernel_ms'], d['bit_exact']))"
 done
done
