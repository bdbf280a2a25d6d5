// Issue cost of the f64 / conversion / integer VALU instructions the renderer's kernels are made
// of, measured on the device: every wave runs 8 independent chains of one instruction, 8 waves per
// SIMD, so the figure is the SIMD's throughput (cycles between issues of a wave64 instruction),
// printed relative to v_fma_f64.   hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kIters = 512;

// D = 64-bit destination chains, S = 64-bit source operands
#define KERNEL_DD(name, ASM)                                                               \
    __global__ void name(double* out, double s0, double s1) {                              \
        double a0 = s0 + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < kIters; ++i) {                                                 \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(s0), "v"(s1) : "vcc");                                      \
        }                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); \
    }

// 32-bit destination chains fed from 64-bit or 32-bit sources
#define KERNEL_WD(name, ASM)                                                               \
    __global__ void name(double* out, double s0, double s1) {                              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t w0 = threadIdx.x * 7u + 3u;                                               \
        for (int i = 0; i < kIters; ++i) {                                                 \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(s0), "v"(s1), "v"(w0) : "vcc", "s10", "s11");                             \
        }                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7))); \
    }

// 64-bit destination chains fed from a 32-bit source
#define KERNEL_DW(name, ASM)                                                               \
    __global__ void name(double* out, double s0, double s1) {                              \
        double a0 = s0, a1 = s0, a2 = s0, a3 = s0, a4 = s0, a5 = s0, a6 = s0, a7 = s0;     \
        uint32_t w0 = threadIdx.x * 7u + 3u;                                               \
        for (int i = 0; i < kIters; ++i) {                                                 \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(w0) \
                         : "v"(s0), "v"(s1) : "vcc");                                      \
        }                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); \
    }

#define A_FMA64(n) "v_fma_f64 %" #n ", %8, %9, %" #n "\n"
#define A_MUL64(n) "v_mul_f64 %" #n ", %8, %" #n "\n"
#define A_ADD64(n) "v_add_f64 %" #n ", %8, %" #n "\n"
#define A_MIN64(n) "v_min_f64 %" #n ", %8, %" #n "\n"
#define A_RCP64(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define A_RSQ64(n) "v_rsq_f64 %" #n ", %" #n "\n"
#define A_SQRT64(n) "v_sqrt_f64 %" #n ", %" #n "\n"
#define A_TRUNC64(n) "v_trunc_f64 %" #n ", %" #n "\n"
#define A_FLOOR64(n) "v_floor_f64 %" #n ", %" #n "\n"
#define A_RNDNE64(n) "v_rndne_f64 %" #n ", %" #n "\n"
#define A_LDEXP64(n) "v_ldexp_f64 %" #n ", %" #n ", 1\n"
#define A_DIVSCALE(n) "v_div_scale_f64 %" #n ", vcc, %8, %9, %8\n"
#define A_DIVFMAS(n) "v_div_fmas_f64 %" #n ", %8, %9, %" #n "\n"
#define A_DIVFIXUP(n) "v_div_fixup_f64 %" #n ", %" #n ", %8, %9\n"
#define A_CMP64(n) "v_cmp_lt_f64 vcc, %8, %9\n"
#define A_MOV64(n) "v_mov_b64 %" #n ", %8\n"

#define A_CVT_U32_F64(n) "v_cvt_u32_f64 %" #n ", %8\n"
#define A_CVT_I32_F64(n) "v_cvt_i32_f64 %" #n ", %8\n"
#define A_CVT_F32_F64(n) "v_cvt_f32_f64 %" #n ", %8\n"
#define A_ADD32(n) "v_add_u32 %" #n ", %10, %" #n "\n"
#define A_FMA32(n) "v_fma_f32 %" #n ", %10, %10, %" #n "\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %10, %" #n "\n"
#define A_MULHI(n) "v_mul_hi_u32 %" #n ", %10, %" #n "\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %10, %" #n ", vcc\n"
#define A_CNDMASK_S(n) "v_cndmask_b32_e64 %" #n ", %10, %" #n ", s[10:11]\n"
#define A_CNDMASK_NEW(n) "v_cndmask_b32 %" #n ", %10, %10, vcc\n"
#define A_CMP_CND(n) "v_cmp_lt_u32 vcc, %10, %" #n "\nv_cndmask_b32 %" #n ", %10, %" #n ", vcc\n"
#define A_CMP32(n) "v_cmp_lt_u32 vcc, %10, %" #n "\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %10, %" #n ", %10\n"
#define A_MAX32(n) "v_max_u32 %" #n ", %10, %" #n "\n"
#define A_BFE(n) "v_bfe_u32 %" #n ", %10, 8, 8\n"
#define A_AND(n) "v_and_b32 %" #n ", %10, %" #n "\n"
#define A_CVT_F32_UB(n) "v_cvt_f32_ubyte1 %" #n ", %10\n"

#define A_CVT_F64_U32(n) "v_cvt_f64_u32 %" #n ", %8\n"
#define A_CVT_F64_I32(n) "v_cvt_f64_i32 %" #n ", %8\n"
#define A_CVT_F64_F32(n) "v_cvt_f64_f32 %" #n ", %8\n"
#define A_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %8, %8, %" #n "\n"

KERNEL_DD(k_fma64, A_FMA64)
KERNEL_DD(k_mul64, A_MUL64)
KERNEL_DD(k_add64, A_ADD64)
KERNEL_DD(k_min64, A_MIN64)
KERNEL_DD(k_rcp64, A_RCP64)
KERNEL_DD(k_rsq64, A_RSQ64)
KERNEL_DD(k_sqrt64, A_SQRT64)
KERNEL_DD(k_trunc64, A_TRUNC64)
KERNEL_DD(k_floor64, A_FLOOR64)
KERNEL_DD(k_rndne64, A_RNDNE64)
KERNEL_DD(k_ldexp64, A_LDEXP64)
KERNEL_DD(k_divscale, A_DIVSCALE)
KERNEL_DD(k_divfmas, A_DIVFMAS)
KERNEL_DD(k_divfixup, A_DIVFIXUP)
KERNEL_DD(k_cmp64, A_CMP64)
KERNEL_DD(k_mov64, A_MOV64)
KERNEL_WD(k_cvt_u32_f64, A_CVT_U32_F64)
KERNEL_WD(k_cvt_i32_f64, A_CVT_I32_F64)
KERNEL_WD(k_cvt_f32_f64, A_CVT_F32_F64)
KERNEL_WD(k_add32, A_ADD32)
KERNEL_WD(k_fma32, A_FMA32)
KERNEL_WD(k_mullo, A_MULLO)
KERNEL_WD(k_mulhi, A_MULHI)
KERNEL_WD(k_cndmask, A_CNDMASK)
KERNEL_WD(k_cndmask_s, A_CNDMASK_S)
KERNEL_WD(k_cndmask_new, A_CNDMASK_NEW)
KERNEL_WD(k_cmp_cnd, A_CMP_CND)
KERNEL_WD(k_cmp32, A_CMP32)
KERNEL_WD(k_bfi, A_BFI)
KERNEL_WD(k_max32, A_MAX32)
KERNEL_WD(k_bfe, A_BFE)
KERNEL_WD(k_and, A_AND)
KERNEL_WD(k_cvt_f32_ub, A_CVT_F32_UB)
KERNEL_DW(k_cvt_f64_u32, A_CVT_F64_U32)
KERNEL_DW(k_cvt_f64_i32, A_CVT_F64_I32)
KERNEL_DW(k_cvt_f64_f32, A_CVT_F64_F32)
KERNEL_DW(k_mad64, A_MAD64)

struct Entry {
    const char* name;
    void (*fn)(double*, double, double);
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;  // kHz -> MHz
    const int blocks = cus * 8, threads = 256;    // 8 blocks of 4 waves per CU = 8 waves per SIMD
    double* out = nullptr;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * sizeof(double)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const Entry table[] = {
        {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_min_f64", k_min64},
        {"v_rcp_f64", k_rcp64}, {"v_rsq_f64", k_rsq64}, {"v_sqrt_f64", k_sqrt64}, {"v_trunc_f64", k_trunc64},
        {"v_floor_f64", k_floor64}, {"v_rndne_f64", k_rndne64}, {"v_ldexp_f64", k_ldexp64},
        {"v_div_scale_f64", k_divscale}, {"v_div_fmas_f64", k_divfmas}, {"v_div_fixup_f64", k_divfixup},
        {"v_cmp_lt_f64", k_cmp64}, {"v_mov_b64", k_mov64},
        {"v_cvt_u32_f64", k_cvt_u32_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_f64_f32", k_cvt_f64_f32},
        {"v_add_u32", k_add32}, {"v_fma_f32", k_fma32}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi},
        {"v_mad_u64_u32", k_mad64}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask (sgpr)", k_cndmask_s}, {"v_cndmask (no chain)", k_cndmask_new},
        {"v_cmp_u32+cndmask", k_cmp_cnd}, {"v_cmp_lt_u32", k_cmp32}, {"v_bfi_b32", k_bfi}, {"v_max_u32", k_max32}, {"v_bfe_u32", k_bfe}, {"v_and_b32", k_and},
        {"v_cvt_f32_ubyte1", k_cvt_f32_ub},
    };
    std::printf("device %s, %d CUs, %.0f MHz; %d waves per SIMD, %d instructions per wave\n", prop.name, cus, mhz,
                blocks * 4 / (cus * 4), kIters * 8);
    double base = 0.0;
    for (const Entry& t : table) {
        hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, out, 1.000001, 0.999999);  // warm-up
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, out, 1.000001, 0.999999);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // instructions issued per SIMD = 8 waves * kIters * 8
        const double per_simd = 8.0 * kIters * 8.0;
        const double cycles = best * 1e-3 * mhz * 1e6 / per_simd;
        if (base == 0.0) base = cycles;
        std::printf("%-18s %8.1f us  %6.2f cycles/instr (at %.0f MHz)  %5.2f x v_fma_f64\n", t.name, best * 1e3, cycles, mhz,
                    cycles / base);
    }
    CHECK(hipFree(out));
    return 0;
}
