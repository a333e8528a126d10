// Host-to-device copy rates from page-locked memory in pieces, as a streaming block issues them:
// runtime copies (one or two streams) against a kernel that reads the host memory itself.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/h2d.hip -o tools/ubench/h2d && tools/ubench/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_pull(u4 *__restrict__ dst, const u4 *__restrict__ src, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv)
{
    // h2d [GiB of page-locked source, default 1] [1: write every byte of it first, default 0: one byte per page]
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 1) << 30;
    const bool fill = argc > 2 && atoi(argv[2]) != 0;
    char *h, *d;
    CK(hipHostMalloc((void **)&h, total));
    CK(hipMalloc((void **)&d, total));
    if (fill) for (size_t i = 0; i < total; i++) h[i] = (char)(i * 7);
    else for (size_t i = 0; i < total; i += 4096) h[i] = (char)i;
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreate(&x));
    CK(hipMemcpy(d, h, total, hipMemcpyHostToDevice));
    const size_t sizes[] = {(size_t)256 << 10, (size_t)512 << 10, (size_t)2 << 20, (size_t)8 << 20, (size_t)64 << 20};
    for (size_t sz : sizes) {
        const size_t n = total / sz;
        for (int mode = 0; mode < 5; mode++) {  // 0: memcpyAsync 1 stream, 1: 2 streams, 2: 4 streams, 3: pull 1 stream, 4: pull 2 streams
            const int ns = mode == 0 || mode == 3 ? 1 : mode == 2 ? 4 : 2;
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            double host_s = 0;
            for (size_t k = 0; k < n; k++) {
                const auto a = std::chrono::steady_clock::now();
                if (mode < 3) CK(hipMemcpyAsync(d + k * sz, h + k * sz, sz, hipMemcpyHostToDevice, s[k % ns]));
                else {
                    size_t blocks = sz / 16 / 256;
                    if (blocks > 2048) blocks = 2048;
                    hipLaunchKernelGGL(k_pull, dim3((unsigned)blocks), dim3(256), 0, s[k % ns], (u4 *)(d + k * sz), (const u4 *)(h + k * sz), sz / 16);
                }
                host_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
            }
            CK(hipDeviceSynchronize());
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("piece %6zu KiB  %-22s %6.1f GB/s   host %5.1f us per call\n", sz >> 10,
                   mode == 0 ? "memcpyAsync 1 stream" : mode == 1 ? "memcpyAsync 2 streams" : mode == 2 ? "memcpyAsync 4 streams"
                   : mode == 3 ? "pull kernel 1 stream" : "pull kernel 2 streams", total / dt / 1e9, host_s / n * 1e6);
        }
    }
    return 0;
}
