#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t a, uint32_t b, int iters) {
    uint32_t x0 = threadIdx.x + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 + 11, x5 = x0 + 13, x6 = x0 + 17, x7 = x0 + 19;
    for (int i = 0; i < iters; ++i) {
#define STEP(x) \
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)); \
        if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b)); \
        if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b)); \
        if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)); \
        if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(x) : "v"(b)); \
        if (OP == 5) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)); \
        if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(b)); \
        if (OP == 7) asm volatile("v_bfe_u32 %0, %0, %1, 5" : "+v"(x) : "v"(b)); \
        if (OP == 8) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(b)); \
        if (OP == 9) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x6c" : "+v"(x) : "v"(b)); \
        if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b)); \
        if (OP == 11) asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc"); \
        if (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(b));
        STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
template <int OP> void run(const char* name, uint32_t* d) {
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1, 3, iters);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 1, 3, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 8;
    printf("%-18s %.3f ms  %.1f G wave-instr/s  (%.2f cycles/instr/SIMD at 2.4 GHz)\n", name, ms, winstr / ms / 1e6, ms * 1e-3 * 2.4e9 * 1024 / winstr);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<3>("v_mul_u32_u24", d); run<4>("v_alignbit_b32", d);
    run<5>("v_mul_hi_u32_u24", d); run<6>("v_mad_u32_u24", d); run<7>("v_bfe_u32", d); run<8>("v_lshrrev_b32", d); run<9>("v_bitop3_b32", d);
    run<10>("v_cndmask_b32", d); run<11>("v_cmp_le_u32", d); run<12>("v_lshl_add_u32", d);
    return 0;
}
