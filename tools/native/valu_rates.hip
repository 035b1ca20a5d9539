// Issue cost of single VALU instructions on gfx950 (four waves per SIMD with four independent chains each: 256 CUs x 16 waves): cycles per wave-instruction from
// s_memtime around an unrolled chain of N independent instructions.  Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))
#define REP64(X) REP4(REP16(X))

template <int WHICH>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint64_t* cyc, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 7, d = b + 11;
    uint32_t e = a * 3 + 1, f = b * 5 + 1, g = c * 7 + 1, h = d * 9 + 1;
    const uint32_t k = 0x7feb352dU | (seed & 1);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
        if (WHICH == 0) {   // v_mul_lo_u32
            REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));)
        } else if (WHICH == 1) {   // v_mul_u32_u24
            REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));)
        } else if (WHICH == 2) {   // v_mad_u64_u32 (64-bit product)
            uint64_t p0 = a, p1 = b, p2 = c, p3 = d;
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %6, 0\n v_mad_u64_u32 %2, vcc, %4, %7, 0\n v_mad_u64_u32 %3, vcc, %4, %8, 0"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(k), "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");)
            a = (uint32_t)p0 ^ (uint32_t)(p0 >> 32); b = (uint32_t)p1; c = (uint32_t)p2; d = (uint32_t)(p3 >> 32);
        } else if (WHICH == 3) {   // v_xor_b32 (full rate reference)
            REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));)
        } else if (WHICH == 4) {   // v_exp_f32
            REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (WHICH == 5) {   // v_mul_hi_u32
            REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));)
        } else if (WHICH == 6) {   // v_mad_u32_u24
            REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));)
        } else if (WHICH == 7) {   // v_alignbit_b32 (rotate)
            REP16(asm volatile("v_alignbit_b32 %0, %0, %0, 13\n v_alignbit_b32 %1, %1, %1, 13\n v_alignbit_b32 %2, %2, %2, 13\n v_alignbit_b32 %3, %3, %3, 13" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (WHICH == 8) {   // v_mul_hi_u32_u24
            REP16(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));)
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W> static void run(const char* name, uint32_t* out, uint64_t* cyc) {
    const int blocks = 1024;                       // 4 waves per CU: one per SIMD
    hipLaunchKernelGGL(rate_kernel<W>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<W>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double n_instr = 4.0 * 256.0 * 64.0;      // per SIMD: 4 waves x 256 iterations x 16 x 4
    printf("%-18s kernel %.1f us  s_memtime delta %llu -> %.2f counter ticks per wave-instruction (100 MHz counter: x24 = cycles at 2.4 GHz)\n", name, ms * 1e3,
           (unsigned long long)h[0], (double)h[0] / n_instr);
    // wall-clock estimate: 1024 waves over 1024 SIMDs, each n_instr instructions
    printf("                   wall: %.2f ns per wave-instruction = %.1f cycles at 2.4 GHz\n", ms * 1e6 / n_instr, ms * 1e6 / n_instr * 2.4);
}

int main() {
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    run<3>("v_xor_b32", out, cyc);
    run<0>("v_mul_lo_u32", out, cyc);
    run<5>("v_mul_hi_u32", out, cyc);
    run<1>("v_mul_u32_u24", out, cyc);
    run<8>("v_mul_hi_u32_u24", out, cyc);
    run<6>("v_mad_u32_u24", out, cyc);
    run<2>("v_mad_u64_u32", out, cyc);
    run<7>("v_alignbit_b32", out, cyc);
    run<4>("v_exp_f32", out, cyc);
    return 0;
}
