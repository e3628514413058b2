// valu_multi.hip — VALU issue rate of one gfx950 SIMD as a function of the number of resident waves, per opcode.
// Question: what is the ceiling the decode kernel's VALU-issue fraction should be measured against?
// A lone wave issues one VALU instruction per ~4.8-5.1 cycles; this measures what W waves on one SIMD reach together.
//
// Placement: blocks of 256*W threads, one block per CU (grid 256) -> W waves on every SIMD. Each wave runs
// REP*ITER instructions of one opcode in 8 independent chains and stamps s_memtime around them.
// Reported per (op, W): median cycles per instruction as one wave sees it, and SIMD cycles per instruction
// (= that / W: all waves of a block run concurrently).
// "mix" rows: waves 0..3 of a 512-thread block run op X, waves 4..7 op Y (one of each per SIMD).
// "half" rows: lanes 32..63 masked off.
// Build: hipcc --offload-arch=gfx950 -O3 valu_multi.hip -o valu_multi
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 64
#define ITER 2048

enum { ADD, SUB, AND, OR, XOR, LSHL, LSHR, ASHR, MIN, MAX, FFBH, BFE, MOV, NOT, MUL24, MULLO, MAD24, MADU24, SAD, MED3,
       XAD, ADD3, LSHLADD, LSHLOR, ANDOR, BFI, ALIGNBIT, PERM, SUBCLAMP, LSHL64, FMA, PKADD16, CNDMASK, CMP, ADDCO, NOPS };
static const char* kNames[] = {"v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32",
    "v_ashrrev_i32", "v_min_u32", "v_max_i32", "v_ffbh_u32", "v_bfe_u32", "v_mov_b32", "v_not_b32", "v_mul_u32_u24",
    "v_mul_lo_u32", "v_mad_i32_i24", "v_mad_u32_u24", "v_sad_u32", "v_med3_i32", "v_xad_u32", "v_add3_u32", "v_lshl_add_u32",
    "v_lshl_or_b32", "v_and_or_b32", "v_bfi_b32", "v_alignbit_b32", "v_perm_b32", "v_sub_u32 clamp", "v_lshlrev_b64",
    "v_fma_f32", "v_pk_add_u16", "v_cndmask_b32", "v_cmp_lt_u32", "v_add_co_u32", "-"};

template <int OP>
__device__ __forceinline__ void one(unsigned& x, unsigned b, unsigned c, unsigned long long& y) {
    if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(c));
    if (OP == LSHR) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(c));
    if (OP == ASHR) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(x) : "v"(c));
    if (OP == MIN) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == MAX) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == FFBH) asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));
    if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(x));
    if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
    if (OP == NOT) asm volatile("v_not_b32 %0, %0" : "+v"(x));
    if (OP == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == MAD24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == MADU24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == SAD) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == MED3) asm volatile("v_med3_i32 %0, %0, -1, 1" : "+v"(x));
    if (OP == XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(b));
    if (OP == LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == BFI) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == SUBCLAMP) asm volatile("v_sub_u32 %0, %0, %1 clamp" : "+v"(x) : "v"(b));
    if (OP == LSHL64) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(y) : "v"(c));
    if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
    if (OP == PKADD16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
    if (OP == CMP) asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(x), "v"(b) : "vcc");
    if (OP == ADDCO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(b) : "vcc");
}

template <int OPX, int OPY, int HALF>
__global__ void k(unsigned* out, unsigned long long* cyc, unsigned seed) {
    unsigned a[8];
    unsigned long long y[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed * (threadIdx.x + 1) + i * 77u;
        y[i] = ((unsigned long long)a[i] << 20) + i;
    }
    unsigned b = seed ^ threadIdx.x, c = (seed + 3) & 15;
    const unsigned wave = threadIdx.x >> 6;
    const bool second = OPY != NOPS && (wave >= blockDim.x / 128);
    unsigned long long t0 = 0, t1 = 0;
    if (!HALF || (threadIdx.x & 63) < 32) {
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (!second) {
            for (int it = 0; it < ITER; ++it) {
#pragma unroll
                for (int r = 0; r < REP; ++r) one<OPX>(a[r & 7], b, c, y[r & 7]);
            }
        } else {
            for (int it = 0; it < ITER; ++it) {
#pragma unroll
                for (int r = 0; r < REP; ++r) one<OPY>(a[r & 7], b, c, y[r & 7]);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (unsigned)y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int OPX, int OPY, int HALF>
static void run(const char* name, int w) {
    const int blocks = 256, threads = 256 * w, waves = blocks * threads / 64;
    unsigned* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, (size_t)blocks * threads * 4);
    (void)hipMalloc(&cyc, (size_t)waves * 8);
    hipLaunchKernelGGL((k<OPX, OPY, HALF>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 12345u);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<OPX, OPY, HALF>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 12345u);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), cyc, (size_t)waves * 8, hipMemcpyDeviceToHost);
    if (OPY == NOPS) {
        std::sort(h.begin(), h.end());
        const double per = (double)h[waves / 2] / (REP * ITER);
        printf("%-34s W=%d  wave cyc/instr %.3f (min %.3f max %.3f)  SIMD cyc/instr %.3f\n", name, w, per,
               (double)h[0] / (REP * ITER), (double)h[waves - 1] / (REP * ITER), per / w);
    } else {
        std::vector<unsigned long long> x, yv;
        const int wpb = threads / 64;
        for (int i = 0; i < waves; ++i) ((i % wpb) < wpb / 2 ? x : yv).push_back(h[i]);
        std::sort(x.begin(), x.end());
        std::sort(yv.begin(), yv.end());
        printf("%-34s W=%d  X wave cyc/instr %.3f  Y wave cyc/instr %.3f\n", name, w,
               (double)x[x.size() / 2] / (REP * ITER), (double)yv[yv.size() / 2] / (REP * ITER));
    }
    (void)hipFree(out);
    (void)hipFree(cyc);
}

template <int OP>
static void sweep() {
    for (int w : {1, 2, 4}) run<OP, NOPS, 0>(kNames[OP], w);
}

int main() {
    sweep<ADD>(); sweep<SUB>(); sweep<AND>(); sweep<OR>(); sweep<XOR>(); sweep<LSHL>(); sweep<LSHR>(); sweep<ASHR>();
    sweep<MIN>(); sweep<MAX>(); sweep<FFBH>(); sweep<BFE>(); sweep<MOV>(); sweep<NOT>(); sweep<MUL24>(); sweep<MULLO>();
    sweep<MAD24>(); sweep<MADU24>(); sweep<SAD>(); sweep<MED3>(); sweep<XAD>(); sweep<ADD3>(); sweep<LSHLADD>();
    sweep<LSHLOR>(); sweep<ANDOR>(); sweep<BFI>(); sweep<ALIGNBIT>(); sweep<PERM>(); sweep<SUBCLAMP>(); sweep<LSHL64>();
    sweep<FMA>(); sweep<PKADD16>(); sweep<CNDMASK>(); sweep<CMP>(); sweep<ADDCO>();
    run<ADD, MAD24, 0>("mix add | mad24", 2);
    run<ADD, ADD, 0>("mix add | add", 2);
    run<MAD24, MAD24, 0>("mix mad24 | mad24", 2);
    run<ADD, MAD24, 0>("mix add | mad24", 4);
    run<LSHL, SAD, 0>("mix lshl | sad", 2);
    run<ADD, NOPS, 1>("v_add_u32 half", 1);
    run<ADD, NOPS, 1>("v_add_u32 half", 2);
    run<ADD, NOPS, 1>("v_add_u32 half", 4);
    run<MAD24, NOPS, 1>("v_mad_i32_i24 half", 1);
    run<MAD24, NOPS, 1>("v_mad_i32_i24 half", 2);
    run<MAD24, NOPS, 1>("v_mad_i32_i24 half", 4);
    return 0;
}
