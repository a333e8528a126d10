// Per-instruction VALU throughput on one SIMD: cycles per instruction per SIMD with 1/2/4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define KERNEL(NAME, ASM)                                                                       \
    __global__ void NAME(float *out, long long *cyc, int iters)                                 \
    {                                                                                           \
        float a = threadIdx.x, b = 1.5f, d = 0.25f + threadIdx.x, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0; \
        unsigned u = threadIdx.x;                                                               \
        long long t0 = __builtin_amdgcn_s_memtime();                                            \
        for (int i = 0; i < iters; i++) {                                                       \
            asm volatile(REP8(ASM) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
                         : "v"(a), "v"(b), "v"(u), "v"(d) : "vcc", "s20", "s21", "s22", "s23");                                     \
        }                                                                                       \
        long long t1 = __builtin_amdgcn_s_memtime();                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;     \
        if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }

// 8 independent destinations per line, 8 lines => 64 instructions per loop iteration
#define L8(OP, SRC) OP " %0, " SRC "\n" OP " %1, " SRC "\n" OP " %2, " SRC "\n" OP " %3, " SRC "\n" \
                    OP " %4, " SRC "\n" OP " %5, " SRC "\n" OP " %6, " SRC "\n" OP " %7, " SRC "\n"
KERNEL(k_add, L8("v_add_f32_e32", "%8, %9"))
KERNEL(k_sub, L8("v_sub_f32_e32", "%8, %9"))
KERNEL(k_max, L8("v_max_f32_e32", "%8, %9"))
KERNEL(k_add64, L8("v_add_f32_e64", "%8, %9"))
KERNEL(k_alignbit, L8("v_alignbit_b32", "%10, %8, 31"))
KERNEL(k_lshlor, L8("v_lshl_or_b32", "%10, 1, %8"))
KERNEL(k_lshladd, L8("v_lshl_add_u32", "%10, 1, %8"))
KERNEL(k_bfi, L8("v_bfi_b32", "%10, %8, %9"))
KERNEL(k_andor, L8("v_and_or_b32", "%10, %8, %9"))
KERNEL(k_lshr, L8("v_lshrrev_b32_e32", "31, %8"))
KERNEL(k_cndmask, "s_mov_b64 vcc, 0x5555\ns_nop 4\n" L8("v_cndmask_b32_e32", "%8, %9, vcc"))
KERNEL(k_cndmask64, "s_mov_b64 s[20:21], 0x5555\ns_nop 4\n" L8("v_cndmask_b32_e64", "%8, %9, s[20:21]"))
#define L8N(OP, SRC) OP " " SRC "\n" OP " " SRC "\n" OP " " SRC "\n" OP " " SRC "\n" OP " " SRC "\n" OP " " SRC "\n" OP " " SRC "\n" OP " " SRC "\n"
KERNEL(k_cmp32, L8N("v_cmp_gt_f32_e32", "vcc, %8, %9"))
KERNEL(k_cmp64, L8N("v_cmp_gt_f32_e64", "s[20:21], %8, %9"))
KERNEL(k_addc32, "s_mov_b64 vcc, 0x5555\ns_nop 4\n" L8("v_addc_co_u32_e32", "vcc, %10, %10, vcc"))
KERNEL(k_addc64, "s_mov_b64 s[20:21], 0x5555\ns_nop 4\n" L8("v_addc_co_u32_e64", "s[22:23], %10, %10, s[20:21]"))
KERNEL(k_ashr, L8("v_ashrrev_i32_e32", "31, %8"))
KERNEL(k_min, L8("v_min_f32_e32", "%8, %9"))
KERNEL(k_maxu, L8("v_max_u32_e32", "%10, %10"))
KERNEL(k_maxi, L8("v_max_i32_e32", "%10, %10"))
KERNEL(k_or, L8("v_or_b32_e32", "%10, %10"))
KERNEL(k_lshlrev, L8("v_lshlrev_b32_e32", "1, %10"))
KERNEL(k_pkadd, L8("v_add_f16_e32", "%8, %9"))
KERNEL(k_mov, L8("v_mov_b32_e32", "%8"))
KERNEL(k_fma, L8("v_fma_f32", "%8, %9, %8"))
KERNEL(k_addu, L8("v_add_u32_e32", "%10, %10"))
KERNEL(k_max3, L8("v_max3_f32", "%8, %9, %8"))
KERNEL(k_sub64, L8("v_sub_f32_e64", "%8, %9"))

// Operand patterns of the two instructions whose rate disagrees with the plain add: three distinct sources, the
// two-operand multiply-accumulate form, destination among the sources, VOP3 encoding.
KERNEL(k_fma_3src, L8("v_fma_f32", "%8, %9, %11"))
KERNEL(k_fmac, L8("v_fmac_f32_e32", "%8, %9"))
KERNEL(k_mul, L8("v_mul_f32_e32", "%8, %9"))
KERNEL(k_mac_dpp, L8("v_mul_f32_e32", "%8, %8"))
KERNEL(k_max_e64, L8("v_max_f32_e64", "%8, %9"))
KERNEL(k_max_same, L8("v_max_f32_e32", "%8, %8"))
#define L8S(OP, SRC) OP " %0, %0, " SRC "\n" OP " %1, %1, " SRC "\n" OP " %2, %2, " SRC "\n" OP " %3, %3, " SRC "\n" \
                     OP " %4, %4, " SRC "\n" OP " %5, %5, " SRC "\n" OP " %6, %6, " SRC "\n" OP " %7, %7, " SRC "\n"
KERNEL(k_max_inplace, L8S("v_max_f32_e32", "%8"))
KERNEL(k_add_inplace, L8S("v_add_f32_e32", "%8"))
KERNEL(k_writelane, L8N("v_writelane_b32", "%0, s20, 3"))
KERNEL(k_mov_dpp, L8("v_mov_b32_dpp", "%8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL(k_add_dpp, L8("v_add_f32_dpp", "%8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL(k_mov_dpp_ror, L8("v_mov_b32_dpp", "%8 row_ror:8 row_mask:0xf bank_mask:0xf"))
KERNEL(k_swap32, "v_permlane32_swap_b32_e32 %0, %1\nv_permlane32_swap_b32_e32 %2, %3\nv_permlane32_swap_b32_e32 %4, %5\nv_permlane32_swap_b32_e32 %6, %7\n"
                 "v_permlane32_swap_b32_e32 %0, %1\nv_permlane32_swap_b32_e32 %2, %3\nv_permlane32_swap_b32_e32 %4, %5\nv_permlane32_swap_b32_e32 %6, %7\n")
KERNEL(k_swap16, "v_permlane16_swap_b32_e32 %0, %1\nv_permlane16_swap_b32_e32 %2, %3\nv_permlane16_swap_b32_e32 %4, %5\nv_permlane16_swap_b32_e32 %6, %7\n"
                 "v_permlane16_swap_b32_e32 %0, %1\nv_permlane16_swap_b32_e32 %2, %3\nv_permlane16_swap_b32_e32 %4, %5\nv_permlane16_swap_b32_e32 %6, %7\n")

typedef void (*kfn)(float *, long long *, int);
void run(const char *name, kfn f)
{
    float *out; long long *cyc;
    (void)hipMalloc(&out, sizeof(float) * 1024);
    (void)hipMalloc(&cyc, sizeof(long long) * 16);
    printf("%-12s", name);
    for (int threads : {64, 256, 512, 1024}) {
        const int iters = 4000;
        hipLaunchKernelGGL(f, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(f, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(threads / 64);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        long long mx = 0;
        for (auto v : h) mx = v > mx ? v : mx;
        double waves_per_simd = threads <= 256 ? 1.0 : threads / 256.0;
        // cycles per instruction as seen by the SIMD (all its waves together)
        printf("  %4dthr: %.2f cyc/inst/SIMD", threads, mx / (iters * 64.0) / waves_per_simd);
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run("add_e32", k_add); run("sub_e32", k_sub); run("max_e32", k_max); run("add_e64", k_add64); run("sub_e64", k_sub64);
    run("alignbit", k_alignbit); run("lshl_or", k_lshlor); run("lshl_add", k_lshladd); run("bfi", k_bfi);
    run("and_or", k_andor); run("lshrrev", k_lshr); run("cndmask", k_cndmask); run("mov", k_mov); run("fma", k_fma);
    run("add_u32", k_addu); run("max3", k_max3);
    run("cndmask64", k_cndmask64); run("cmp_e32", k_cmp32); run("cmp_e64", k_cmp64); run("addc_e32", k_addc32);
    run("addc_e64", k_addc64); run("ashr", k_ashr); run("min_f32", k_min); run("max_u32", k_maxu); run("max_i32", k_maxi);
    run("or", k_or); run("lshlrev", k_lshlrev);
    printf("-- operand patterns\n");
    run("fma a,b,a", k_fma); run("fma a,b,d", k_fma_3src); run("fmac a,b", k_fmac); run("mul a,b", k_mul); run("mul a,a", k_mac_dpp);
    run("max a,b e32", k_max); run("max a,b e64", k_max_e64); run("max a,a", k_max_same); run("max c,c,a", k_max_inplace);
    run("add c,c,a", k_add_inplace);
    printf("-- lane exchange (the lane = state decoder)\n");
    run("writelane", k_writelane); run("mov_dpp quad", k_mov_dpp); run("add_dpp quad", k_add_dpp); run("mov_dpp ror8", k_mov_dpp_ror);
    run("swap32", k_swap32); run("swap16", k_swap16);
    return 0;
}
