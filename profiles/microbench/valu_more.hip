// valu_more.hip — round 3 addendum to valu_multi.hip: opcodes the shortened Golomb step leans on (v_bitop3_b32,
// v_mad_u64_u32 vs v_mad_u32_u24, v_bfe_i32, v_or3, v_cndmask with an SGPR-pair mask, v_cmp into an SGPR pair), each
// as 8 independent chains (issue rate) and as ONE dependent chain (latency seen by a lone wave), W waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_more.hip -o valu_more
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP 64
#define ITER 1024

enum { ADDNOP, ADD, MAD24U, MADU64, BITOP3, BFEI, OR3, ADDLSHL, CNDS, CMPS, MULLO, LSHL64, BFI, FFBH, MIN, NOPS };
static const char* kNames[] = {"v_add_u32 + s_nop 0", "v_add_u32", "v_mad_u32_u24", "v_mad_u64_u32", "v_bitop3_b32", "v_bfe_i32", "v_or3_b32",
                               "v_add_lshl_u32", "v_cndmask_b32 (sgpr mask)", "v_cmp_ne_u32 -> sgpr", "v_mul_lo_u32",
                               "v_lshlrev_b64", "v_bfi_b32", "v_ffbh_u32", "v_min_u32", "-"};

template <int OP>
__device__ __forceinline__ void one(unsigned& x, unsigned b, unsigned c, unsigned long long& y, unsigned long long m) {
    if (OP == ADDNOP) asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 0" : "+v"(x) : "v"(b));
    if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == MAD24U) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == MADU64) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(y) : "v"(b), "v"(c) : "s10", "s11");
    if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8" : "+v"(x) : "v"(b), "v"(c));
    if (OP == BFEI) asm volatile("v_bfe_i32 %0, %0, 0, 1" : "+v"(x));
    if (OP == OR3) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == ADDLSHL) asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(x) : "v"(b));
    if (OP == CNDS) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(m));
    if (OP == CMPS) asm volatile("v_cmp_ne_u32 s[10:11], %0, %1" ::"v"(x), "v"(b) : "s10", "s11");
    if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == LSHL64) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(y) : "v"(c));
    if (OP == BFI) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == FFBH) asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));
    if (OP == MIN) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
}

template <int OP, int CHAINS>
__global__ void k(unsigned* out, unsigned long long* cyc, unsigned seed) {
    unsigned a[8];
    unsigned long long y[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed * (threadIdx.x + 1) + i * 77u;
        y[i] = ((unsigned long long)a[i] << 20) + i;
    }
    unsigned b = seed ^ threadIdx.x, c = (seed + 3) & 15;
    const unsigned long long m = __ballot((threadIdx.x & 1) != 0);
    const unsigned wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) one<OP>(a[r & (CHAINS - 1)], b, c, y[r & (CHAINS - 1)], m);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (unsigned)y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int OP, int CHAINS>
static void run(int w) {
    const int blocks = 256, threads = 256 * w, waves = blocks * threads / 64;
    unsigned* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, (size_t)blocks * threads * 4);
    (void)hipMalloc(&cyc, (size_t)waves * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 12345u);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), cyc, (size_t)waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[waves / 2] / (REP * ITER);
    printf("%-28s chains=%d W=%d  wave cyc/instr %.3f  SIMD cyc/instr %.3f\n", kNames[OP], CHAINS, w, per, per / w);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

template <int OP>
static void sweep() {
    run<OP, 8>(1);
    run<OP, 8>(2);
    run<OP, 8>(4);
    run<OP, 1>(1);
    run<OP, 1>(2);
}

int main() {
    sweep<ADDNOP>(); sweep<ADD>(); sweep<MAD24U>(); sweep<MADU64>(); sweep<BITOP3>(); sweep<BFEI>(); sweep<OR3>(); sweep<ADDLSHL>();
    sweep<CNDS>(); sweep<CMPS>(); sweep<MULLO>(); sweep<LSHL64>(); sweep<BFI>(); sweep<FFBH>(); sweep<MIN>();
    return 0;
}
