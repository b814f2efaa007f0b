// Micro-benchmark: issue cost / dependent latency of the fp64 VALU instructions the ZM ascent is made of (gfx950).
// One wave per block; blocks = 1 (a wave alone on a SIMD) or many.  Prints shader cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o valu_f64 valu_f64.hip && ./valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
constexpr int ITER = 512;

#define KERNEL(name, body, per_iter)                                                                     \
    __global__ void name(double *out, long long *cyc, double seed)                                        \
    {                                                                                                     \
        double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + 1.0, c = 1.0 - seed, d = seed * 0.25 + 2.0; \
        double e = 0.999999, f = 1e-9;                                                                    \
        int i1 = 3;                                                                                       \
        long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                               \
        for (int it = 0; it < ITER; ++it) {                                                               \
            body                                                                                          \
        }                                                                                                 \
        long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + i1;                                  \
        if (threadIdx.x == 0) cyc[blockIdx.x] = (t1 - t0);                                                \
    }                                                                                                     \
    static const int name##_n = per_iter;

// dependent FMA chain
KERNEL(k_fma_dep, REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(e), "v"(f));), 16)
KERNEL(k_fma_ind2, REP4(REP4(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(e), "v"(f));)), 32)
KERNEL(k_fma_ind4, REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));), 64)
KERNEL(k_add_dep, REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(f));), 16)
KERNEL(k_add_ind2, REP16(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(f));), 32)
KERNEL(k_mul_dep, REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(e));), 16)
KERNEL(k_mul_ind2, REP16(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(e));), 32)
KERNEL(k_rcp_dep, REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));), 16)
KERNEL(k_rcp_ind2, REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1" : "+v"(a), "+v"(b));), 32)
KERNEL(k_rcp_fma, REP16(asm volatile("v_rcp_f64 %0, %0\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(e), "v"(f));), 32)
KERNEL(k_rcp_3fma, REP16(asm volatile("v_rcp_f64 %0, %0\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));), 64)
KERNEL(k_ldexp_dep, REP16(asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(i1));), 16)
KERNEL(k_ldexp_ind2, REP16(asm volatile("v_ldexp_f64 %0, %0, %2\n v_ldexp_f64 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(i1));), 32)
KERNEL(k_rndne_dep, REP16(asm volatile("v_rndne_f64 %0, %0" : "+v"(a));), 16)
KERNEL(k_rndne_ind2, REP16(asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1" : "+v"(a), "+v"(b));), 32)
KERNEL(k_frexp_mant_ind2, REP16(asm volatile("v_frexp_mant_f64 %0, %0\n v_frexp_mant_f64 %1, %1" : "+v"(a), "+v"(b));), 32)
KERNEL(k_cvt_i32_f64_ind2, REP16(asm volatile("v_cvt_i32_f64 %0, %2\n v_cvt_i32_f64 %1, %3" : "+v"(i1), "+v"(i1) : "v"(a), "v"(b));), 32)
KERNEL(k_cvt_f64_i32_ind2, REP16(asm volatile("v_cvt_f64_i32 %0, %2\n v_cvt_f64_i32 %1, %2" : "+v"(a), "+v"(b) : "v"(i1));), 32)
KERNEL(k_div_scale_ind2, REP16(asm volatile("v_div_scale_f64 %0, vcc, %0, %2, %0\n v_div_scale_f64 %1, vcc, %1, %2, %1" : "+v"(a), "+v"(b) : "v"(e) : "vcc");), 32)
KERNEL(k_div_fmas_ind2, REP16(asm volatile("v_div_fmas_f64 %0, %0, %2, %3\n v_div_fmas_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(e), "v"(f) : "vcc");), 32)
KERNEL(k_div_fixup_ind2, REP16(asm volatile("v_div_fixup_f64 %0, %0, %2, %3\n v_div_fixup_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(e), "v"(d));), 32)
KERNEL(k_cmp_cnd, REP16(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(i1) : "v"(i1) : "vcc");), 32)
KERNEL(k_min_dep, REP16(asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(e));), 16)
KERNEL(k_mov64_ind2, REP16(asm volatile("v_mov_b64 %0, %2\n v_mov_b64 %1, %2" : "+v"(a), "+v"(b) : "v"(e));), 32)
KERNEL(k_mov32_dep, REP16(asm volatile("v_mov_b32 %0, %0" : "+v"(i1));), 16)
KERNEL(k_add32_dep, REP16(asm volatile("v_add_u32 %0, %0, %0" : "+v"(i1));), 16)
// the division sequence as hipcc emits it (one full fp64 divide), dependent on itself
KERNEL(k_fdiv_dep, REP4(a = b / a;), 4)
KERNEL(k_fdiv_ind2, REP4(a = e / a; b = e / b;), 8)
KERNEL(k_nop, REP16(asm volatile("s_nop 0");), 16)

struct T { const char *name; void (*fn)(double *, long long *, double); int n; };
#define ENT(k) {#k, k, k##_n}

int main(int argc, char **argv)
{
    std::vector<T> tests = {ENT(k_fma_dep), ENT(k_fma_ind2), ENT(k_fma_ind4), ENT(k_add_dep), ENT(k_add_ind2), ENT(k_mul_dep), ENT(k_mul_ind2),
                            ENT(k_rcp_dep), ENT(k_rcp_ind2), ENT(k_rcp_fma), ENT(k_rcp_3fma), ENT(k_ldexp_dep), ENT(k_ldexp_ind2), ENT(k_rndne_dep),
                            ENT(k_rndne_ind2), ENT(k_frexp_mant_ind2), ENT(k_cvt_i32_f64_ind2), ENT(k_cvt_f64_i32_ind2), ENT(k_div_scale_ind2),
                            ENT(k_div_fmas_ind2), ENT(k_div_fixup_ind2), ENT(k_cmp_cnd), ENT(k_min_dep), ENT(k_mov64_ind2), ENT(k_mov32_dep),
                            ENT(k_add32_dep), ENT(k_fdiv_dep), ENT(k_fdiv_ind2), ENT(k_nop)};
    double *out;
    long long *cyc;
    hipMalloc(&out, 1 << 22);
    hipMalloc(&cyc, 8 * 4096);
    // geometry: (blocks, threads): one wave alone; 2 and 3 waves on one SIMD are approximated by one block of 512 / 768 threads
    // (8 / 12 waves over the 4 SIMDs of a CU)
    const int geo[][2] = {{1, 64}, {1, 512}, {1, 768}, {1, 1024}};
    printf("%-22s", "cycles/instr");
    for (auto &g : geo) printf("  %4dx%-5d", g[0], g[1]);
    printf("   (per wave; throughput per SIMD = value / waves-per-SIMD)\n");
    for (auto &t : tests) {
        printf("%-22s", t.name);
        for (auto &g : geo) {
            long long h[16];
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(t.fn, dim3(g[0]), dim3(g[1]), 0, 0, out, cyc, 1.2345);
                hipDeviceSynchronize();
                hipMemcpy(h, cyc, sizeof(long long) * g[0], hipMemcpyDeviceToHost);
                double v = (double)h[0] / (ITER * (double)t.n);
                if (v < best) best = v;
            }
            printf("  %10.2f", best);
        }
        printf("\n");
    }
    return 0;
}
