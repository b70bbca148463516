// Micro-benchmark: issue cost of v_mad_u64_u32 / v_fma_f64 / v_add_f64 / v_xor / ds_read_b64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
template <int OP>
__global__ void k(uint64_t* out, uint32_t a, double d) {
    uint32_t x0 = threadIdx.x + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
    double f0 = d + threadIdx.x, f1 = f0 * 1.1, f2 = f0 * 1.2, f3 = f0 * 1.3;
    uint64_t acc = 0;
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) {  // 4 independent mad_u64_u32
            uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x1, p2 = (uint64_t)0xD2511F53u * x2, p3 = (uint64_t)0xCD9E8D57u * x3;
            x0 = (uint32_t)(p0 >> 32) ^ (uint32_t)p1; x1 = (uint32_t)(p1 >> 32) ^ (uint32_t)p2; x2 = (uint32_t)(p2 >> 32) ^ (uint32_t)p3; x3 = (uint32_t)(p3 >> 32) ^ (uint32_t)p0;
        } else if (OP == 1) {  // 4 independent fma f64
            f0 = __builtin_fma(f0, 1.0000001, 0.5); f1 = __builtin_fma(f1, 1.0000001, 0.5); f2 = __builtin_fma(f2, 1.0000001, 0.5); f3 = __builtin_fma(f3, 1.0000001, 0.5);
        } else if (OP == 2) {  // 4 xor+add u32
            x0 = (x0 ^ x1) + 1; x1 = (x1 ^ x2) + 3; x2 = (x2 ^ x3) + 5; x3 = (x3 ^ x0) + 7;
        } else if (OP == 3) {  // 4 independent add f64
            f0 = f0 + 1.5; f1 = f1 + 2.5; f2 = f2 + 3.5; f3 = f3 + 4.5;
        } else if (OP == 4) {  // 4 independent mul_lo u32
            x0 = x0 * 0xD2511F53u + 1; x1 = x1 * 0xCD9E8D57u + 1; x2 = x2 * 0xD2511F53u + 1; x3 = x3 * 0xCD9E8D57u + 1;
        }
    }
    acc = x0 ^ x1 ^ x2 ^ x3;
    acc += (uint64_t)(f0 + f1 + f2 + f3);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int OP> void run(const char* name, int opsPerIter, int waves_per_simd) {
    uint64_t* d; hipMalloc(&d, 8 * 256 * 1024 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block per CU
    k<OP><<<blocks, 256>>>(d, 1, 1.0); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 1, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per wave-instruction per SIMD, assuming 2.4 GHz
    double instr_per_simd = (double)ITER * opsPerIter * waves_per_simd;
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instr (at 2.4 GHz; includes loop overhead)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("mad_u64+xor", 8, w); run<1>("fma_f64", 4, w); run<2>("xor+add u32", 8, w); run<3>("add_f64", 4, w); run<4>("mul_lo+add", 8, w);
    }
    return 0;
}
