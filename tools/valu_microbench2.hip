// Issue-cost table for gfx950 (4 waves per SIMD, 8 independent registers per wave): ns per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 1500
#define RUN8(OP) asm volatile(OP(0,1) OP(1,2) OP(2,3) OP(3,4) OP(4,5) OP(5,6) OP(6,7) OP(7,0) \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(sa), "s"(mask) : "vcc")
// operand numbering: %0..%7 = x0..x7, %8 = a (vgpr), %9 = b (vgpr), %10 = sa (sgpr float), %11 = mask (sgpr pair)
#define S(x) #x
#define OP_FMA(d,s)      "v_fma_f32 %" S(d) ", %" S(d) ", %8, %9\n"
#define OP_FMAC(d,s)     "v_fmac_f32 %" S(d) ", %8, %9\n"
#define OP_MUL(d,s)      "v_mul_f32 %" S(d) ", %" S(d) ", %8\n"
#define OP_ADD(d,s)      "v_add_f32 %" S(d) ", %" S(d) ", %8\n"
#define OP_MULS(d,s)     "v_mul_f32 %" S(d) ", %10, %" S(d) "\n"
#define OP_FMAK(d,s)     "v_fmaak_f32 %" S(d) ", %" S(d) ", %8, 0x3f9d70a4\n"
#define OP_MOV(d,s)      "v_mov_b32 %" S(d) ", %" S(s) "\n"
#define OP_CND_VCC(d,s)  "v_cndmask_b32 %" S(d) ", %" S(d) ", %" S(s) ", vcc\n"
#define OP_CND_S(d,s)    "v_cndmask_b32 %" S(d) ", %" S(d) ", %" S(s) ", %11\n"
#define OP_CMP(d,s)      "v_cmp_lt_f32 vcc, %" S(d) ", %" S(s) "\n"
#define OP_CMPS(d,s)     "v_cmp_lt_f32 %11, %" S(d) ", %" S(s) "\n"
#define OP_MIN(d,s)      "v_min_f32 %" S(d) ", %" S(d) ", %" S(s) "\n"
#define OP_MAX(d,s)      "v_max_f32 %" S(d) ", %" S(d) ", %" S(s) "\n"
#define OP_MED3(d,s)     "v_med3_f32 %" S(d) ", %" S(d) ", %8, %9\n"
#define OP_MIN3(d,s)     "v_min3_f32 %" S(d) ", %" S(d) ", %8, %9\n"
#define OP_BFI(d,s)      "v_bfi_b32 %" S(d) ", %8, %" S(d) ", %9\n"
#define OP_AND(d,s)      "v_and_b32 %" S(d) ", %" S(d) ", %8\n"
#define OP_XOR(d,s)      "v_xor_b32 %" S(d) ", %" S(d) ", %8\n"
#define OP_ANDOR(d,s)    "v_and_or_b32 %" S(d) ", %" S(d) ", -16, %10\n"
#define OP_LSHL(d,s)     "v_lshlrev_b32 %" S(d) ", 3, %" S(d) "\n"
#define OP_ADDU(d,s)     "v_add_u32 %" S(d) ", %" S(d) ", %8\n"
#define OP_CVT(d,s)      "v_cvt_i32_f32 %" S(d) ", %" S(d) "\n"
#define OP_RNDNE(d,s)    "v_rndne_f32 %" S(d) ", %" S(d) "\n"
#define OP_SQRT(d,s)     "v_sqrt_f32 %" S(d) ", %" S(d) "\n"
#define OP_RCP(d,s)      "v_rcp_f32 %" S(d) ", %" S(d) "\n"
#define OP_RSQ(d,s)      "v_rsq_f32 %" S(d) ", %" S(d) "\n"
#define OP_CMPCND(d,s)   "v_cmp_lt_f32 vcc, %" S(d) ", %" S(s) "\n v_cndmask_b32 %" S(d) ", %" S(d) ", %" S(s) ", vcc\n"
#define OP_CMPCNDS(d,s)  "v_cmp_lt_f32 %11, %" S(d) ", %" S(s) "\n v_cndmask_b32 %" S(d) ", %" S(d) ", %" S(s) ", %11\n"
#define OP_CND64V(d,s)   "v_cndmask_b32_e64 %" S(d) ", %" S(d) ", %" S(s) ", vcc\n"
#define OP_PKFMA(d,s)    ""
#define KERNEL(NAME, OP) __global__ void NAME(float* out, float a, float b, float sa, unsigned long long mask) { \
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int it = 0; it < ITER; ++it) { RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); RUN8(OP); } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; }
KERNEL(k_fma, OP_FMA) KERNEL(k_fmac, OP_FMAC) KERNEL(k_mul, OP_MUL) KERNEL(k_add, OP_ADD) KERNEL(k_muls, OP_MULS) KERNEL(k_fmak, OP_FMAK)
KERNEL(k_mov, OP_MOV) KERNEL(k_cndv, OP_CND_VCC) KERNEL(k_cnds, OP_CND_S) KERNEL(k_cmp, OP_CMP) KERNEL(k_cmps, OP_CMPS)
KERNEL(k_min, OP_MIN) KERNEL(k_max, OP_MAX) KERNEL(k_med3, OP_MED3) KERNEL(k_min3, OP_MIN3) KERNEL(k_bfi, OP_BFI) KERNEL(k_and, OP_AND)
KERNEL(k_xor, OP_XOR) KERNEL(k_andor, OP_ANDOR) KERNEL(k_lshl, OP_LSHL) KERNEL(k_addu, OP_ADDU) KERNEL(k_cvt, OP_CVT) KERNEL(k_rndne, OP_RNDNE)
KERNEL(k_cmpcnd, OP_CMPCND) KERNEL(k_cmpcnds, OP_CMPCNDS) KERNEL(k_cnd64v, OP_CND64V) KERNEL(k_sqrt, OP_SQRT) KERNEL(k_rcp, OP_RCP) KERNEL(k_rsq, OP_RSQ)
typedef void (*KF)(float*, float, float, float, unsigned long long);
void run(const char* name, KF f, float* d) {
    const int wps = 4, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 1.5f, 0x5555555555555555ull);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 1.5f, 0x5555555555555555ull);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / 3, n = (double)wps * ITER * 64;
    printf("%-22s %.3f ns/wave-instr/SIMD  (x%.2f of v_fma)\n", name, t / n * 1e9, 0.0);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 2048 * sizeof(float));
#define R(n) run(#n, n, d)
    R(k_fma); R(k_fmac); R(k_mul); R(k_add); R(k_muls); R(k_fmak); R(k_mov); R(k_cndv); R(k_cnds); R(k_cmp); R(k_cmps); R(k_min); R(k_max);
    R(k_med3); R(k_min3); R(k_bfi); R(k_and); R(k_xor); R(k_andor); R(k_lshl); R(k_addu); R(k_cvt); R(k_rndne); R(k_cmpcnd); R(k_cmpcnds); R(k_cnd64v); R(k_sqrt); R(k_rcp); R(k_rsq);
    return 0;
}
