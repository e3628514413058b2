L=saprobe-alac_amd/csrc/libalacgpu.so; H=profiles/exp_bin/libalacgpu_r4head.so
for e in 0 1 2 3 4; do
echo "# EXP=$e stereo 98304"; ALACGPU_EXP=$e python tools/ab_bench.py --packets 98304 --rounds 4 $L $H 2>/dev/null
echo "# EXP=$e mono 81920"; ALACGPU_EXP=$e python tools/ab_bench.py --channels 1 --packets 81920 --rounds 4 $L $H 2>/dev/null
done
