// VALU issue-rate micro-benchmark: cycles per instruction for dependent / independent chains,
// for 1..N waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu.hip -o valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS, int KINDOP>
__global__ void k(float *out, long long *cyc, int iters, float seed)
{
    float v[CHAINS];
    unsigned u[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { v[c] = seed + c + threadIdx.x; u[c] = threadIdx.x + c; }
    float inc = seed * 0.5f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KINDOP == 0) v[c] = v[c] + inc;                                   // v_add_f32
                else if (KINDOP == 1) v[c] = fmaxf(v[c], inc + c);                    // v_max_f32
                else if (KINDOP == 2) u[c] = __builtin_amdgcn_alignbit(u[c], __float_as_uint(v[c]), 31);  // v_alignbit
                asm volatile("" : "+v"(v[c]), "+v"(u[c]));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc += v[c] + u[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS, int KINDOP>
void run(const char *name, int threads, int blocks)
{
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long) * (threads / 64) * blocks);
    const int iters = 2000;
    hipLaunchKernelGGL((k<CHAINS, KINDOP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    hipLaunchKernelGGL((k<CHAINS, KINDOP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h((threads / 64) * blocks);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    double ninst = (double)iters * 16 * CHAINS;
    // s_memtime ticks at 100 MHz-ish constant clock? report raw ticks per instruction
    printf("%-34s threads/block=%4d blocks=%4d  ticks/inst(wave)=%.3f\n", name, threads, blocks, mx / ninst);
    hipFree(out); hipFree(cyc);
}

int main()
{
    // waves per SIMD = threads/256 when one block per CU
    for (int threads : {64, 256, 512, 768, 1024}) {
        run<1, 0>("add dependent chain", threads, 1);
        run<8, 0>("add 8 independent chains", threads, 1);
        run<8, 1>("max 8 independent chains", threads, 1);
        run<8, 2>("alignbit 8 independent chains", threads, 1);
    }
    run<8, 0>("add 8 chains, all CUs 1w/SIMD", 256, 256);
    run<8, 0>("add 8 chains, all CUs 2w/SIMD", 512, 256);
    return 0;
}
