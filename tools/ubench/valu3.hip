// Does a wave64 VALU instruction get cheaper when half of the wave's lanes are switched off?
// Cycles per instruction per SIMD for a few opcodes with 64, 32 (lower half) and 16 active lanes, 1/2/4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define KERNEL(NAME, ASM)                                                                       \
    __global__ void NAME(float *out, long long *cyc, int iters, int active)                     \
    {                                                                                           \
        float a = threadIdx.x, b = 1.5f, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0; \
        unsigned u = threadIdx.x;                                                               \
        long long t0 = 0, t1 = 0;                                                               \
        if ((int)(threadIdx.x & 63) < active) {                                                 \
            t0 = __builtin_amdgcn_s_memtime();                                                  \
            for (int i = 0; i < iters; i++) {                                                   \
                asm volatile(REP8(ASM) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
                             : "v"(a), "v"(b), "v"(u));                                         \
            }                                                                                   \
            t1 = __builtin_amdgcn_s_memtime();                                                  \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;     \
        if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }
#define L8(OP, SRC) OP " %0, " SRC "\n" OP " %1, " SRC "\n" OP " %2, " SRC "\n" OP " %3, " SRC "\n" \
                    OP " %4, " SRC "\n" OP " %5, " SRC "\n" OP " %6, " SRC "\n" OP " %7, " SRC "\n"
KERNEL(k_add, L8("v_add_f32_e32", "%8, %9"))
KERNEL(k_max, L8("v_max_f32_e32", "%8, %9"))
KERNEL(k_alignbit, L8("v_alignbit_b32", "%10, %8, 31"))

typedef void (*kfn)(float *, long long *, int, int);
void run(const char *name, kfn f)
{
    float *out; long long *cyc;
    (void)hipMalloc(&out, sizeof(float) * 1024);
    (void)hipMalloc(&cyc, sizeof(long long) * 16);
    for (int active : {64, 32, 16}) {
        printf("%-10s active=%2d", name, active);
        for (int threads : {64, 512, 1024}) {
            const int iters = 4000;
            hipLaunchKernelGGL(f, dim3(1), dim3(threads), 0, 0, out, cyc, iters, active);
            hipLaunchKernelGGL(f, dim3(1), dim3(threads), 0, 0, out, cyc, iters, active);
            (void)hipDeviceSynchronize();
            std::vector<long long> h(threads / 64);
            (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            long long mx = 0;
            for (auto v : h) mx = v > mx ? v : mx;
            double wps = threads <= 256 ? 1.0 : threads / 256.0;
            printf("  %4dthr: %.2f cyc/inst/SIMD", threads, mx / (iters * 64.0) / wps);
        }
        printf("\n");
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run("add_f32", k_add); run("max_f32", k_max); run("alignbit", k_alignbit);
    return 0;
}
