// Issue cost of VALU instruction classes on gfx950 as a function of the number of resident waves per SIMD.
// ns and shader cycles (2.39 GHz) per wave-instruction per SIMD; 8 independent destination registers per wave.
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_microbench3.hip -o tools/valu_microbench3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 1000
#define RUN8(OP) asm volatile(OP(0,1) OP(1,2) OP(2,3) OP(3,4) OP(4,5) OP(5,6) OP(6,7) OP(7,0) \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(p2) : "v"(a), "v"(b), "s"(sa), "s"(mask) : "vcc")
// %0..%7 = x0..x7, %8 = p2 (vgpr pair), %9 = a (vgpr), %10 = b (vgpr), %11 = sa (sgpr), %12 = mask (sgpr pair)
#define S(x) #x
#define OP_FMA(d,s)      "v_fma_f32 %" S(d) ", %" S(d) ", %9, %10\n"
#define OP_FMA3(d,s)     "v_fma_f32 %" S(d) ", %" S(s) ", %9, %" S(d) "\n"
#define OP_FMAC(d,s)     "v_fmac_f32 %" S(d) ", %9, %10\n"
#define OP_FMANEG(d,s)   "v_fma_f32 %" S(d) ", -%" S(d) ", %9, %10\n"
#define OP_FMAS(d,s)     "v_fma_f32 %" S(d) ", %" S(d) ", %11, %10\n"
#define OP_FMAMK(d,s)    "v_fmamk_f32 %" S(d) ", %" S(d) ", 0x3f9d70a4, %9\n"
#define OP_FMAAK(d,s)    "v_fmaak_f32 %" S(d) ", %" S(d) ", %9, 0x3f9d70a4\n"
#define OP_FMAINL(d,s)   "v_fma_f32 %" S(d) ", %" S(d) ", 2.0, %10\n"
#define OP_MUL(d,s)      "v_mul_f32 %" S(d) ", %" S(d) ", %9\n"
#define OP_MULLIT(d,s)   "v_mul_f32 %" S(d) ", 0x3f9d70a4, %" S(d) "\n"
#define OP_MULS(d,s)     "v_mul_f32 %" S(d) ", %11, %" S(d) "\n"
#define OP_ADD(d,s)      "v_add_f32 %" S(d) ", %" S(d) ", %9\n"
#define OP_SUB(d,s)      "v_sub_f32 %" S(d) ", %" S(d) ", %9\n"
#define OP_MOV(d,s)      "v_mov_b32 %" S(d) ", %" S(s) "\n"
#define OP_MOV0(d,s)     "v_mov_b32 %" S(d) ", 0\n"
#define OP_PKFMA(d,s)    "v_pk_fma_f32 %8, %8, %8, %8\n"
#define OP_PKMUL(d,s)    "v_pk_mul_f32 %8, %8, %8\n"
#define OP_CND(d,s)      "v_cndmask_b32 %" S(d) ", %" S(d) ", %" S(s) ", vcc\n"
#define OP_CMP(d,s)      "v_cmp_lt_f32 vcc, %" S(d) ", %" S(s) "\n"
#define OP_MIN(d,s)      "v_min_f32 %" S(d) ", %" S(d) ", %" S(s) "\n"
#define OP_MIN3(d,s)     "v_min3_f32 %" S(d) ", %" S(d) ", %9, %10\n"
#define OP_MED3(d,s)     "v_med3_f32 %" S(d) ", %" S(d) ", %9, %10\n"
#define OP_ANDOR(d,s)    "v_and_or_b32 %" S(d) ", %" S(d) ", -16, %11\n"
#define OP_ANDORV(d,s)   "v_and_or_b32 %" S(d) ", %" S(d) ", %9, %10\n"
#define OP_AND(d,s)      "v_and_b32 %" S(d) ", %" S(d) ", %9\n"
#define OP_BFI(d,s)      "v_bfi_b32 %" S(d) ", %9, %" S(d) ", %10\n"
#define OP_ADDU(d,s)     "v_add_u32 %" S(d) ", %" S(d) ", %9\n"
#define OP_RSQ(d,s)      "v_rsq_f32 %" S(d) ", %" S(d) "\n"
#define OP_SIN(d,s)      "v_sin_f32 %" S(d) ", %" S(d) "\n"
#define OP_DPP(d,s)      "v_mov_b32_dpp %" S(d) ", %" S(s) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define KERNEL(NAME, OP) __global__ void __launch_bounds__(64) NAME(float* out, float a, float b, float sa, unsigned long long mask) { \
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    typedef float f2 __attribute__((ext_vector_type(2))); f2 p2 = {x0, x1}; \
    for (int it = 0; it < ITER; ++it) { RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p2.x; }
#define LIST(X) X(FMA) X(FMA3) X(FMAC) X(FMANEG) X(FMAS) X(FMAMK) X(FMAAK) X(FMAINL) X(MUL) X(MULLIT) X(MULS) X(ADD) X(SUB) X(MOV) X(MOV0) \
    X(PKFMA) X(PKMUL) X(CND) X(CMP) X(MIN) X(MIN3) X(MED3) X(ANDOR) X(ANDORV) X(AND) X(BFI) X(ADDU) X(RSQ) X(SIN) X(DPP)
#define DEF(N) KERNEL(k_##N, OP_##N)
LIST(DEF)
typedef void (*KF)(float*, float, float, float, unsigned long long);
double run(KF f, float* d, int wps) {
    const int blocks = 256 * 4 * wps;                       // one-wave workgroups: wps per SIMD on 256 CUs x 4 SIMDs
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, 1.5f, 0x5555555555555555ull);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(f, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, 1.5f, 0x5555555555555555ull);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / 3 / ((double)wps * ITER * 64) * 1e9;     // ns per wave-instruction per SIMD
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    printf("%-10s %8s %8s %8s %8s   (cycles at 2.39 GHz per wave-instruction per SIMD; waves per SIMD = 1, 2, 4, 8)\n", "op", "1", "2", "4", "8");
#define ROW(N) { printf("%-10s", #N); for (int w : {1, 2, 4, 8}) printf(" %8.2f", run(k_##N, d, w) * 2.39); printf("\n"); }
    LIST(ROW)
    return 0;
}
