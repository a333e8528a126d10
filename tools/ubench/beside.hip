// Do uploads run beside a long kernel of another stream?  A kernel that keeps a few waves busy for some milliseconds
// is put on stream X, alone or followed by a device-to-host copy on X (as a block's results would follow its decoder);
// meanwhile 1-MiB pieces go up on two other streams, by the runtime's copy or by a kernel that reads host memory,
// each piece waited for before the next but one is sent (the way a streaming block consumes its input).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/beside.hip -o tools/ubench/beside && tools/ubench/beside
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_pull(u4 *__restrict__ dst, const u4 *__restrict__ src, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}
__global__ __launch_bounds__(256) void k_busy(float *out, long long cycles)
{
    const long long t0 = wall_clock64();
    float a = threadIdx.x;
    while (wall_clock64() - t0 < cycles) a = a * 1.0001f + 0.5f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t piece = (size_t)1 << 20, n_piece = 256;
    char *h, *d, *res_h;
    float *busy_d;
    CK(hipHostMalloc((void **)&h, piece * n_piece));
    CK(hipHostMalloc((void **)&res_h, (size_t)16 << 20));
    CK(hipMalloc((void **)&d, piece * n_piece));
    CK(hipMalloc((void **)&busy_d, (size_t)16 << 20));
    for (size_t i = 0; i < piece * n_piece; i += 4096) h[i] = (char)i;
    hipStream_t X, A, B;
    CK(hipStreamCreate(&A)); CK(hipStreamCreate(&B)); CK(hipStreamCreate(&X));
    hipEvent_t ev[n_piece];
    for (auto &evk : ev) CK(hipEventCreateWithFlags(&evk, hipEventDisableTiming));
    CK(hipMemcpy(d, h, piece * n_piece, hipMemcpyHostToDevice));
    const long long ms5 = 5 * 100000;  // wall_clock64 counts at 100 MHz
    for (int with_kernel = 0; with_kernel < 3; with_kernel++) {      // 0: nothing on X, 1: busy kernel, 2: busy kernel + D2H behind it
        for (int mode = 0; mode < 2; mode++) {                        // 0: runtime copy, 1: pull kernel
            CK(hipDeviceSynchronize());
            const double t0 = now();
            if (with_kernel) hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, X, busy_d, ms5);
            if (with_kernel == 2) CK(hipMemcpyAsync(res_h, busy_d, (size_t)16 << 20, hipMemcpyDeviceToHost, X));
            double worst = 0;
            for (size_t k = 0; k < n_piece; k++) {
                hipStream_t s = (k & 1) ? B : A;
                if (mode == 0) CK(hipMemcpyAsync(d + k * piece, h + k * piece, piece, hipMemcpyHostToDevice, s));
                else hipLaunchKernelGGL(k_pull, dim3(256), dim3(256), 0, s, (u4 *)(d + k * piece), (const u4 *)(h + k * piece), piece / 16);
                CK(hipEventRecord(ev[k], s));
                if (k >= 2) {
                    const double a = now();
                    CK(hipEventSynchronize(ev[k - 2]));
                    worst = std::max(worst, now() - a);
                }
            }
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
            const double t_up = now() - t0;
            CK(hipDeviceSynchronize());
            const double t_all = now() - t0;
            printf("%-28s %-12s uploads done after %6.2f ms (%5.1f GB/s), longest wait %5.2f ms, everything after %6.2f ms\n",
                   with_kernel == 0 ? "alone" : with_kernel == 1 ? "beside a 5-ms kernel" : "beside kernel + D2H behind it",
                   mode == 0 ? "runtime copy" : "pull kernel", t_up * 1e3, piece * n_piece / t_up / 1e9, worst * 1e3, t_all * 1e3);
        }
    }
    return 0;
}
