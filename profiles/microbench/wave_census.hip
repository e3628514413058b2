// wave_census.hip — where do the two waves of a 128-thread workgroup land? (alac_decode's geometry: 1024 workgroups of
// 2 waves, ~34.5 KB of LDS each = 4 workgroups per CU.) Every wave records HW_REG_HW_ID and HW_REG_XCC_ID, then spins
// long enough for the whole grid to be resident together. Prints, per SIMD, how many waves of each role (wave 0 / wave 1
// of its workgroup) it hosts. Build: hipcc --offload-arch=gfx950 -O3 wave_census.hip -o wave_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(128) census(unsigned* hwid, unsigned* xcc, unsigned long long* t, int spin) {
    __shared__ unsigned pad[34 * 256];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned w = blockIdx.x * 2 + (threadIdx.x >> 6);
    unsigned id, xc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = pad[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
    if ((threadIdx.x & 63) == 0) {
        hwid[w] = id;
        xcc[w] = xc;
        t[w] = t0;
    }
    if (x == 0x12345) pad[0] = x;
}

int main() {
    const int blocks = 1024, waves = blocks * 2;
    unsigned *hwid, *xcc;
    unsigned long long* t;
    (void)hipMalloc(&hwid, waves * 4);
    (void)hipMalloc(&xcc, waves * 4);
    (void)hipMalloc(&t, waves * 8);
    hipLaunchKernelGGL(census, dim3(blocks), dim3(128), 0, 0, hwid, xcc, t, 200000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(waves), x(waves);
    (void)hipMemcpy(h.data(), hwid, waves * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(x.data(), xcc, waves * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx940: [16:13])
    std::map<unsigned, std::vector<int>> per_simd;  // key: xcc, se, sh, cu, simd -> roles
    for (int w = 0; w < waves; ++w) {
        const unsigned id = h[w];
        const unsigned key = ((x[w] & 0xf) << 20) | (id & 0x6f30);  // xcc, se [14:13], cu [11:8], simd [5:4] (gfx950 dump)
        per_simd[key].push_back(w & 1);
    }
    int hist[3][9] = {{0}};
    int same = 0, mixed = 0;
    for (auto& kv : per_simd) {
        int a = 0, b = 0;
        for (int r : kv.second) (r ? b : a)++;
        if (a + b <= 8) hist[0][a + b]++;
        if (a && b) mixed++; else same++;
    }
    unsigned ormask = 0, andmask = 0xffffffffu;
    for (int w = 0; w < waves; ++w) { ormask |= h[w]; andmask &= h[w]; }
    printf("HW_ID bits that vary over the grid: %08x (or %08x and %08x)\n", ormask & ~andmask, ormask, andmask);
    for (int w = 0; w < 24; ++w) printf("  wave %2d (wg %d.%d): HW_ID %08x XCC_ID %08x\n", w, w / 2, w & 1, h[w], x[w]);
    printf("SIMDs seen: %zu (expected 1024)\n", per_simd.size());
    printf("waves per SIMD histogram:");
    for (int i = 0; i <= 8; ++i) printf(" %d:%d", i, hist[0][i]);
    printf("\nSIMDs hosting both roles: %d, only one role: %d\n", mixed, same);
    printf("first 16 workgroups (wave0 | wave1): xcc se/cu simd waveslot\n");
    for (int b = 0; b < 16; ++b) {
        for (int r = 0; r < 2; ++r) {
            const unsigned id = h[b * 2 + r];
            printf("  wg %2d w%d: xcc %u cu %3u simd %u slot %u |", b, r, x[b * 2 + r] & 0xf, (id >> 8) & 0x1ff, (id >> 4) & 3, id & 0xf);
        }
        printf("\n");
    }
    // how well does (wave ^ slot parity) balance the roles?
    std::map<unsigned, std::vector<int>> per2;
    for (int b = 0; b < blocks; ++b) {
        const unsigned p = (h[b * 2] >> 16) & 1;  // TG_ID bit 0 (bits [17:16] vary: the workgroup's slot on its CU)
        for (int r = 0; r < 2; ++r) {
            const unsigned id = h[b * 2 + r];
            const unsigned key = ((x[b * 2 + r] & 0xf) << 20) | (id & 0x6f30);
            per2[key].push_back(r ^ p);
        }
    }
    mixed = same = 0;
    for (auto& kv : per2) {
        int a = 0, b = 0;
        for (int r : kv.second) (r ? b : a)++;
        if (a && b) mixed++; else same++;
    }
    printf("with role = wave ^ (TG_ID & 1): SIMDs hosting both roles: %d, only one: %d\n", mixed, same);
    // what one CU looks like
    const unsigned cu0 = ((x[0] & 0xf) << 20) | (h[0] & 0x6f00);
    for (int w = 0; w < waves; ++w)
        if ((((x[w] & 0xf) << 20) | (h[w] & 0x6f00)) == cu0)
            printf("  same CU as wave 0: wg %4d wave %d simd %u wave_id %u tg %u\n", w / 2, w & 1, (h[w] >> 4) & 3, h[w] & 0xf, (h[w] >> 16) & 3);
    return 0;
}
