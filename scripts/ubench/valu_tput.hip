// Throughput of fp64 VALU instructions with every SIMD loaded: W waves per SIMD (blocks of 256 threads, W blocks per CU),
// each wave running NCH independent chains.  Prints cycles per wave-instruction per SIMD derived from wall time and the
// nominal 2.4 GHz clock, and TFLOP/s for FMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
constexpr int ITER = 2048;
template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, double seed)
{
    double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + 1.0, c = 1.0 - seed, d = seed * 0.25 + 2.0, e = 0.999999, f = 1e-9;
    float fa = (float)a, fb = (float)b, fc = (float)c, fd = (float)d, fe = 0.99999f, ff = 1e-9f;
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (KIND == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(fe), "v"(ff));) }
        if (KIND == 2) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(f));) }
        if (KIND == 3) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 4) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 5) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(e), "v"(f));) }     // dependent chain
        if (KIND == 6) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) :: "vcc");) }
        if (KIND == 7) { REP16(asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + fa + fb + fc + fd;
}
template <int KIND>
void run(const char *name, int per_iter, double *out)
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%-14s", name);
    for (int W : {1, 2, 3, 4, 6, 8}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * W), dim3(256), 0, 0, out, 1.2345);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * W), dim3(256), 0, 0, out, 1.2345);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)W * ITER * per_iter;       // W waves per SIMD if blocks spread evenly
        const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
        printf("  W=%d: %6.2f cyc/instr", W, cyc);
    }
    printf("\n");
}
int main()
{
    double *out;
    hipMalloc(&out, 256 * 8 * 256 * 8 * 2);
    run<0>("fma_f64 x4", 64, out);
    run<5>("fma_f64 dep", 16, out);
    run<1>("fma_f32 x4", 64, out);
    run<2>("add_f64 x4", 64, out);
    run<3>("mul_f64 x4", 64, out);
    run<4>("rcp_f64 x4", 64, out);
    run<7>("ldexp_f64 x4", 64, out);
    run<6>("cndmask x4", 64, out);
    return 0;
}
