// Micro-benchmark: ISSUE cost (cycles per wave-instruction on one SIMD) of the VALU / LDS instruction classes the filter
// kernels use, on gfx950.  Each kernel runs ITER x 8 independent copies of ONE instruction (inline asm, 8 accumulator
// chains, so dependent latency does not limit issue) with 1, 2 and 4 waves per SIMD.  Output feeds the cost-weighted
// instruction count of k_filter_step (tools/isa_cost.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048

#define BODY8(ASM, C0, C1, C2, C3, C4, C5, C6, C7) \
    asm volatile(ASM :: ); 

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t a, double d) {
    uint32_t x0 = threadIdx.x + a, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3, x4 = x0 * 11 + 4, x5 = x0 * 13 + 5, x6 = x0 * 17 + 6, x7 = x0 * 19 + 7;
    double f0 = d + threadIdx.x * 1e-3, f1 = f0 * 1.1, f2 = f0 * 1.2, f3 = f0 * 1.3, f4 = f0 * 1.4, f5 = f0 * 1.5, f6 = f0 * 1.6, f7 = f0 * 1.7;
    uint64_t u0 = x0, u1 = x1, u2 = x2, u3 = x3, u4 = x4, u5 = x5, u6 = x6, u7 = x7;
    __shared__ double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i;
    __syncthreads();
    const double c1 = 1.0000001, c2 = 0.5;
    const uint32_t m = 0xD2511F53u;
#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) {
#define S(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f##n) : "v"(c1), "v"(c2));
            R8(S)
#undef S
        } else if (OP == 1) {
#define S(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f##n) : "v"(c2));
            R8(S)
#undef S
        } else if (OP == 2) {
#define S(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f##n) : "v"(c1));
            R8(S)
#undef S
        } else if (OP == 3) {
#define S(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x##n) : "v"(m));
            R8(S)
#undef S
        } else if (OP == 4) {
#define S(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x##n) : "v"(m));
            R8(S)
#undef S
        } else if (OP == 5) {
#define S(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(u##n) : "v"(x##n), "v"(m) : "vcc");
            R8(S)
#undef S
        } else if (OP == 6) {
#define S(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x##n) : "v"(m));
            R8(S)
#undef S
        } else if (OP == 7) {
#define S(n) asm volatile("v_rsq_f64 %0, %0" : "+v"(f##n));
            R8(S)
#undef S
        } else if (OP == 8) {
#define S(n) asm volatile("v_rcp_f64 %0, %0" : "+v"(f##n));
            R8(S)
#undef S
        } else if (OP == 9) {
#define S(n) asm volatile("v_sqrt_f64 %0, %0" : "+v"(f##n));
            R8(S)
#undef S
        } else if (OP == 10) {
#define S(n) asm volatile("v_rndne_f64 %0, %0" : "+v"(f##n));
            R8(S)
#undef S
        } else if (OP == 11) {
#define S(n) asm volatile("v_ceil_f64 %0, %0" : "+v"(f##n));
            R8(S)
#undef S
        } else if (OP == 12) {
#define S(n) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(f##n) : "v"(x##n));
            R8(S)
#undef S
        } else if (OP == 13) {
#define S(n) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(f##n), "v"(c1), "v"(x##n), "v"(m) : "vcc");
            R8(S)
#undef S
        } else if (OP == 14) {
#define S(n) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x##n));
            R8(S)
#undef S
        } else if (OP == 15) {
#define S(n) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(f##n) : "v"(x##n));
            R8(S)
#undef S
        } else if (OP == 16) {
#define S(n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x##n) : "v"(m));
            R8(S)
#undef S
        } else if (OP == 17) {
#define S(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(u##n) : "v"(c1));
            R8(S)
#undef S
        } else if (OP == 18) {
#define S(n) asm volatile("v_exp_f32 %0, %0" : "+v"(x##n));
            R8(S)
#undef S
        } else if (OP == 19) {
#define S(n) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(u##n));
            R8(S)
#undef S
        } else if (OP == 20) {
#define S(n) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(x##n) : "v"(m), "v"(a) : "vcc");
            R8(S)
#undef S
        } else if (OP == 21) {   // ds_read_b64, random-ish addresses (x & 2047)*8, waits once per 8
#define S(n) asm volatile("ds_read_b64 %0, %1" : "=v"(f##n) : "v"((x##n & 2047u) * 8u));
            R8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 22) {   // ds_read_b128 random-ish (table look-up pattern)
            double2 t;
#define S(n) asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((x##n & 63u) * 16u)); f##n += t.x;
            R8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 23) {
#define S(n) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(f##n));
            R8(S)
#undef S
        } else if (OP == 24) {
#define S(n) asm volatile("v_max_f64 %0, %0, %1" : "+v"(f##n) : "v"(c1));
            R8(S)
#undef S
        } else if (OP == 25) {
#define S(n) asm volatile("v_trig_preop_f64 %0, %0, %1" : "+v"(f##n) : "v"(x##n));
            R8(S)
#undef S
        } else if (OP == 26) {
#define S(n) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(x##n) : "v"(f##n));
            R8(S)
#undef S
        } else if (OP == 27) {
#define S(n) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(f##n) : "v"(c1));
            R8(S)
#undef S
        } else if (OP == 28) {
#define S(n) asm volatile("v_fract_f64 %0, %0" : "+v"(f##n));
            R8(S)
#undef S
        } else if (OP == 29) {
#define S(n) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(x##n) : "v"(m));
            R8(S)
#undef S
        } else if (OP == 30) {
#define S(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x##n) : "v"(m));
            R8(S)
#undef S
        } else if (OP == 31) {
#define S(n) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x##n) : "v"(f##n));
            R8(S)
#undef S
        } else if (OP == 32) {
#define S(n) asm volatile("v_log_f32 %0, %0" : "+v"(x##n));
            R8(S)
#undef S
        } else if (OP == 33) {
#define S(n) asm volatile("v_readlane_b32 s4, %0, 3" : : "v"(x##n) : "s4");
            R8(S)
#undef S
        } else if (OP == 34) {
#define S(n) asm volatile("v_mov_b32 %0, %1" : "=v"(x##n) : "v"(m));
            R8(S)
#undef S
        } else if (OP == 35) {
#define S(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(u##n) : "v"(c1));
            R8(S)
#undef S
        } else if (OP == 36) {
#define S(n) asm volatile("v_sin_f32 %0, %0" : "+v"(x##n));
            R8(S)
#undef S
        } else if (OP == 37) {
#define S(n) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x##n));
            R8(S)
#undef S
        } else if (OP == 38) {
#define S(n) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(f##n) : "v"(x##n));
            R8(S)
#undef S
        }
    }
    uint64_t acc = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7;
    acc += (uint64_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
static const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_xor_b32", "v_rsq_f64", "v_rcp_f64",
    "v_sqrt_f64", "v_rndne_f64", "v_ceil_f64", "v_ldexp_f64", "v_cmp_f64+cndmask (2)", "v_mov_b32_dpp", "v_cvt_f64_u32", "v_fma_f32", "v_pk_fma_f32", "v_exp_f32",
    "v_lshlrev_b64", "add_co+addc (2)", "ds_read_b64 (+wait/8)", "ds_read_b128 (+wait/8,+add)", "v_frexp_mant_f64", "v_max_f64", "v_trig_preop_f64", "v_cvt_i32_f64",
    "v_div_fixup_f64", "v_fract_f64", "v_alignbit_b32", "v_mul_u32_u24", "v_cvt_f32_f64", "v_log_f32", "v_readlane_b32", "v_mov_b32", "v_pk_mul_f32", "v_sin_f32", "v_sqrt_f32", "v_cvt_f64_f32"};
template <int OP> void run(double ghz) {
    uint64_t* d; hipMalloc(&d, 8 * 256 * 1024 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s", names[OP]);
    for (int w : {1, 2, 4}) {
        int blocks = 256 * w;   // 256 threads = 4 waves = 1 per SIMD per block per CU
        k<OP><<<blocks, 256>>>(d, 1, 1.5); hipDeviceSynchronize();
        hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 1, 1.5); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = (double)ITER * 8 * w;
        printf("  w=%d: %6.2f", w, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    }
    printf("   cycles per wave-instr at %.2f GHz (loop overhead ~2 SALU per 8)\n", ghz);
    hipFree(d);
}
template <int OP> struct Runner { static void go(double g) { Runner<OP - 1>::go(g); run<OP>(g); } };
template <> struct Runner<-1> { static void go(double) {} };
int main() {
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk > 0 ? clk * 1e-6 : 2.4;
    printf("device clock %.3f GHz\n", ghz);
    Runner<38>::go(ghz);
    return 0;
}
