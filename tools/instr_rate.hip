// instr_rate.hip -- issue cost (clocks per wave instruction, independent streams) and dependent latency of the VALU /
// LDS instructions the build and match kernels lean on.  One wave per SIMD (256 threads, one workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 16
#define ITERS 512

#define KERNEL(NAME, DECL, IND, DEP)                                                                  \
    __global__ void k_ind_##NAME(long long *out, int n) {                                             \
        DECL;                                                                                         \
        long long t0 = __builtin_readcyclecounter();                                                  \
        for (int i = 0; i < n; i++) { IND }                                                           \
        long long t1 = __builtin_readcyclecounter();                                                  \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                       \
    }                                                                                                 \
    __global__ void k_dep_##NAME(long long *out, int n) {                                             \
        DECL;                                                                                         \
        long long t0 = __builtin_readcyclecounter();                                                  \
        for (int i = 0; i < n; i++) { DEP }                                                           \
        long long t1 = __builtin_readcyclecounter();                                                  \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                       \
    }

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// ---- 32-bit ops: 16 independent destination registers a[k]; dependent: one register chained
#define DECL32 unsigned a[16]; for (int k = 0; k < 16; k++) a[k] = threadIdx.x + k; unsigned b = threadIdx.x | 1u, c = 77u
#define I_MAD64(k) { unsigned long long r; asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %3" : "=v"(r) : "v"(a[k]), "v"(b), "v"((unsigned long long)c) : "s20", "s21"); a[k] = (unsigned)r; }
#define I_MAD24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_ADD32(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_CNDM(k)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));
#define I_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_FLOOR(k) asm volatile("v_floor_f32 %0, %0" : "+v"(a[k]));
#define I_CVTI(k)  asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k]));
#define I_CMP(k)   asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(a[k]), "v"(b) : "s20", "s21");
#define I_BPERM(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[k]) : "v"(b));
#define I_BPERMQ(k) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define I_DPP(k)   asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
#define D0(X) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0)
#define SINK32 if (n < 0) { unsigned s = 0; for (int k = 0; k < 16; k++) s += a[k]; out[1] = s; }
#define K32(NAME, I) KERNEL(NAME, DECL32, R16(I) SINK32, D0(I) SINK32)
K32(mad_u64_u32, I_MAD64)
K32(mad_u32_u24, I_MAD24)
K32(mul_lo_u32, I_MULLO)
K32(add_u32, I_ADD32)
K32(cndmask, I_CNDM)
K32(fma_f32, I_FMA32)
K32(floor_f32, I_FLOOR)
K32(cvt_i32_f32, I_CVTI)
K32(cmp_u32_sgpr, I_CMP)
K32(bpermute_wait, I_BPERM)
K32(bpermute_queue, I_BPERMQ)
K32(add_dpp, I_DPP)

// ---- 64-bit ops
#define DECL64 double a[16]; for (int k = 0; k < 16; k++) a[k] = 1.0 + 1e-3 * (threadIdx.x + k); double b = 1.0000001, c = 1e-9; float f = 1.5f + threadIdx.x; unsigned long long u = threadIdx.x
#define I_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_RCP64(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
#define I_CVT64(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f));
#define I_CVT64D(k) { float t; asm volatile("v_cvt_f32_f64 %0, %1\n v_cvt_f64_f32 %1, %0" : "=&v"(t), "+v"(a[k])); }
#define I_FLOOR64(k) asm volatile("v_floor_f64 %0, %0" : "+v"(a[k]));
#define I_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_MOV64(k) asm volatile("v_mov_b64 %0, %1" : "=v"(a[k]) : "v"(b));
#define I_LSHL64(k) asm volatile("v_lshlrev_b64 %0, 1, %1" : "=v"(u) : "v"(u));
#define I_EXP32(k) { unsigned t = (unsigned)k; asm volatile("v_exp_f32 %0, %0" : "+v"(t)); a[k] += t; }
#define SINK64 if (n < 0) { double s = 0; for (int k = 0; k < 16; k++) s += a[k]; out[1] = (long long)s + (long long)u; }
#define K64(NAME, I) KERNEL(NAME, DECL64, R16(I) SINK64, D0(I) SINK64)
K64(fma_f64, I_FMA64)
K64(add_f64, I_ADD64)
K64(mul_f64, I_MUL64)
K64(rcp_f64, I_RCP64)
K64(cvt_f64_f32, I_CVT64)
K64(floor_f64, I_FLOOR64)
K64(pk_fma_f32, I_PKFMA)
K64(mov_b64, I_MOV64)
K64(lshl_b64, I_LSHL64)

typedef void (*kern_t)(long long *, int);
struct T { const char *name; kern_t ind, dep; };
#define E(NAME) {#NAME, k_ind_##NAME, k_dep_##NAME}

int main()
{
    std::vector<T> tests = {E(add_u32), E(mad_u32_u24), E(mad_u64_u32), E(mul_lo_u32), E(cndmask), E(cmp_u32_sgpr), E(fma_f32),
                            E(floor_f32), E(cvt_i32_f32), E(add_dpp), E(bpermute_wait), E(bpermute_queue), E(pk_fma_f32),
                            E(fma_f64), E(add_f64), E(mul_f64), E(rcp_f64), E(cvt_f64_f32), E(floor_f64), E(mov_b64), E(lshl_b64)};
    long long *d;
    hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](kern_t k, int threads, int n) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, n);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, n);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return (double)ms;
    };
    // shader clock: s_memtime of a lone wave against the event time of a long run
    const int n = 1 << 15;
    double ms = run(k_ind_fma_f64, 256, 1 << 18);
    long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double ghz = h / (ms * 1e6);
    printf("s_memtime: %.3f GHz (event-timed)\n", ghz);
    printf("%-18s %8s %8s %8s %8s   clocks per wave instruction and SIMD, one workgroup of 1 / 2 / 4 waves per SIMD (independent), dependent chain\n",
           "instruction", "1w", "2w", "4w", "dep 1w");
    for (auto &t : tests) {
        double r[4];
        const int th[4] = {256, 512, 1024, 256};
        for (int m = 0; m < 4; m++) {
            const double base = run(m == 3 ? t.dep : t.ind, th[m], 16);
            const double full = run(m == 3 ? t.dep : t.ind, th[m], n + 16);
            r[m] = (full - base) * 1e6 * ghz / ((double)n * 16.0 * (th[m] / 256));
        }
        printf("%-18s %8.2f %8.2f %8.2f %8.2f\n", t.name, r[0], r[1], r[2], r[3]);
    }
    return 0;
}
