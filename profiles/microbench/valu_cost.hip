// valu_cost.hip — cycles per wave64 instruction on one SIMD of gfx950, for the integer ops the ALAC step uses.
// One wave per SIMD (1024 blocks of 64 threads), each op repeated in an unrolled independent/dependent chain,
// timed with s_memtime. Prints cycles per instruction. Build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>

#define REP 64
#define ITER 256

template <int OP, int CH>
__global__ void __launch_bounds__(64) k(unsigned* out, unsigned long long* cyc, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 77u;
    unsigned b = seed ^ threadIdx.x, c = seed + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            unsigned& x = a[r & (CH - 1)];   // CH independent chains
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 2) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
            if (OP == 5) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 7) asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));
            if (OP == 8) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(c));
            if (OP == 9) asm volatile("v_med3_i32 %0, %0, -1, 1" : "+v"(x));
            if (OP == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
            if (OP == 11) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(x), "v"(b) : "vcc");
            if (OP == 12) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(x));
            if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 14) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 16) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 17) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 18) asm volatile("s_add_u32 s20, s20, s21" ::: "s20");
            if (OP == 19) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(b));
            if (OP == 20) asm volatile("v_cndmask_b32 %0, %0, %1, s[30:31]" : "+v"(x) : "v"(b) : "s30", "s31");
            if (OP == 21) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(b), "v"(c) : "vcc");
            if (OP == 22) asm volatile("v_cmp_lt_u32 s[30:31], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[30:31]" : "+v"(x) : "v"(b), "v"(c) : "s30", "s31");
            if (OP == 23) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 24) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(x) : "v"(b), "v"(c) : "vcc");
            if (OP == 25) asm volatile("v_sub_u32 %0, %0, %1 clamp" : "+v"(x) : "v"(b));
            if (OP == 26) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 27) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 28) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 29) asm volatile("ds_write_b32 %1, %0" :: "v"(x), "v"(c << 2) : "memory");
            if (OP == 30) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 s[30:31], %0, %2\n\ts_and_b64 vcc, vcc, s[30:31]\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(b), "v"(c) : "vcc", "s30", "s31");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
__global__ void __launch_bounds__(64) k64(unsigned* out, unsigned long long* cyc, unsigned seed) {
    unsigned long long a[4];
    for (int i = 0; i < 4; ++i) a[i] = ((unsigned long long)seed << 20) * (threadIdx.x + 1) + i;
    unsigned c = (seed & 15) + 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            unsigned long long& x = a[r & 3];
            if (OP == 0) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(x) : "v"(c));
            if (OP == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(a[(r + 1) & 3]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < 4; ++i) s += (unsigned)a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class F>
static void run(const char* name, F launch) {
    const int blocks = getenv("MB_BLOCKS") ? atoi(getenv("MB_BLOCKS")) : 1024;
    unsigned* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 64 * 4);
    hipMalloc(&cyc, blocks * 8);
    launch(blocks, out, cyc);
    launch(blocks, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // s_memtime ticks at 100 MHz on gfx9-class parts? report raw ticks per instruction as well
    printf("%-18s median ticks/instr %.3f (min %.3f max %.3f)\n", name, (double)h[blocks / 2] / (REP * ITER),
           (double)h[0] / (REP * ITER), (double)h[blocks - 1] / (REP * ITER));
    hipFree(out);
    hipFree(cyc);
}

#define RUN(OP, NAME) run(NAME, [](int b, unsigned* o, unsigned long long* c) { hipLaunchKernelGGL((k<OP, 8>), dim3(b), dim3(64), 0, 0, o, c, 12345u); })
#define RUN64(OP, NAME) run(NAME, [](int b, unsigned* o, unsigned long long* c) { hipLaunchKernelGGL(k64<OP>, dim3(b), dim3(64), 0, 0, o, c, 12345u); })

#define RUNC(OP, CH, NAME) run(NAME, [](int b, unsigned* o, unsigned long long* c) { hipLaunchKernelGGL((k<OP, CH>), dim3(b), dim3(64), 0, 0, o, c, 12345u); })
int main() {
    RUNC(0, 1, "add dep1"); RUNC(0, 2, "add dep2"); RUNC(0, 4, "add dep4"); RUNC(3, 1, "mad24 dep1"); RUNC(3, 2, "mad24 dep2"); RUNC(5, 1, "sad dep1"); RUNC(21, 1, "cmp+cnd dep1"); RUNC(21, 4, "cmp+cnd dep4");
    RUN(0, "v_add_u32"); RUN(1, "v_mul_lo_u32"); RUN(2, "v_mad_u32_u24"); RUN(3, "v_mad_i32_i24"); RUN(4, "v_cndmask_b32");
    RUN(5, "v_sad_u32"); RUN(6, "v_perm_b32"); RUN(7, "v_ffbh_u32"); RUN(8, "v_lshlrev_b32"); RUN(9, "v_med3_i32");
    RUN(10, "v_mov_b32"); RUN(11, "v_cmp_lt_u32"); RUN(12, "v_bfe_i32"); RUN(13, "v_add3_u32"); RUN(14, "v_xad_u32");
    RUN(15, "v_alignbit_b32"); RUN(16, "v_mul_hi_u32"); RUN(17, "v_mul_u32_u24"); RUN(18, "s_add_u32"); RUN(19, "v_lshl_add_u32"); RUN(20, "cndmask_sgpr"); RUN(21, "cmp+cndmask vcc"); RUN(22, "cmp+cndmask sgpr"); RUN(23, "v_max_i32"); RUN(24, "cmp+addc"); RUN(25, "v_sub clamp"); RUN(26, "v_min_u32"); RUN(27, "v_and_or_b32"); RUN(28, "v_lshl_or_b32"); RUN(30, "2cmp+and+cndmask");
    RUN64(0, "v_lshlrev_b64"); RUN64(1, "v_lshl_add_u64");
    return 0;
}
