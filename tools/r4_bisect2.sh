E=profiles/exp_bin; L=saprobe-alac_amd/csrc/libalacgpu.so
echo "# headline: 65536 x 16-bit stereo"; python tools/ab_bench.py --rounds 6 $E/libalacgpu_r3.so $E/libalacgpu_f7fab5e.so $E/libalacgpu_d0d8209.so $E/libalacgpu_r4head.so $L $E/libalacgpu_golasm.so 2>&1 | tail -8
echo "# headline: 65536 x 16-bit stereo, s_setprio variants"; python tools/ab_bench.py --rounds 6 $E/libalacgpu_r3.so $L $E/libalacgpu_pa3.so $E/libalacgpu_pa3c2.so $E/libalacgpu_pa3b21.so $E/libalacgpu_pa3b210c1.so 2>&1 | tail -8
