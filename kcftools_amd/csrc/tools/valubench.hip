// VALU issue-rate microbenchmark: time per wave64 instruction per SIMD at 8 waves/SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define BODY(name, txt) \
__global__ __launch_bounds__(256) void name(uint32_t* out, int iters) { \
    uint32_t a = threadIdx.x, b = threadIdx.x * 7 + 1, c = 3; uint64_t x = a, y = b; \
    asm volatile("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11"); \
    asm volatile("v_cmp_gt_u32 vcc, 40, %0" :: "v"(a) : "vcc"); \
    for (int i = 0; i < iters; ++i) { \
        asm volatile(R16(txt "\n") : "+v"(a), "+v"(x) : "v"(b), "v"(c), "v"(y) : "s12", "s13"); \
    } \
    out[blockIdx.x * 256 + threadIdx.x] = a + (uint32_t)x; \
}
BODY(k_add,      "v_add_u32 %0, %0, %2")
BODY(k_xor,      "v_xor_b32 %0, %0, %2")
BODY(k_lshr,     "v_lshrrev_b32 %0, 3, %0")
BODY(k_mullo,    "v_mul_lo_u32 %0, %0, %2")
BODY(k_mulhi,    "v_mul_hi_u32 %0, %0, %2")
BODY(k_mul24,    "v_mul_u32_u24 %0, %0, %2")
BODY(k_cmp64,    "v_cmp_eq_u64 s[12:13], %1, %4")
BODY(k_cmp32,    "v_cmp_eq_u32 s[12:13], %0, %2")
BODY(k_cmp32vcc, "v_cmp_eq_u32 vcc, %0, %2")
BODY(k_cnd_vcc,  "v_cndmask_b32 %0, %0, %2, vcc")
BODY(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %2, s[10:11]")
BODY(k_cnd_e64vcc, "v_cndmask_b32_e64 %0, %0, %2, vcc")
BODY(k_cnd_mix,  "v_cndmask_b32 %0, %0, %2, vcc\n v_add_u32 %0, %0, %2\n v_min_u32 %0, %0, %3\n v_xor_b32 %0, %0, %2")
BODY(k_cnd_mix2, "v_cndmask_b32_e64 %0, %0, %2, s[10:11]\n v_add_u32 %0, %0, %2\n v_min_u32 %0, %0, %3\n v_xor_b32 %0, %0, %2")
BODY(k_cmpvcc_cnd, "v_cmp_eq_u32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc")
BODY(k_nop_cnd, "s_nop 0\n v_cndmask_b32 %0, %0, %2, vcc")
BODY(k_mindpp,   "v_min_u32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_movdpp,   "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
BODY(k_align,    "v_alignbit_b32 %0, %0, %2, %3")
BODY(k_add64,    "v_lshl_add_u64 %1, %1, 0, %4")
BODY(k_shl64,    "v_lshlrev_b64 %1, 1, %1")
BODY(k_bfrev,    "v_bfrev_b32 %0, %0")
BODY(k_mbcnt,    "v_mbcnt_lo_u32_b32 %0, s10, %0")
BODY(k_bfe,      "v_bfe_u32 %0, %0, 8, 1")
BODY(k_lshladd,  "v_lshl_add_u32 %0, %0, 2, %2")
BODY(k_min,      "v_min_u32 %0, %0, %2")
BODY(k_cmpcnd,   "v_cmp_eq_u32 s[12:13], %0, %2\n v_cndmask_b32_e64 %0, %0, %2, s[12:13]")
template <typename K> void run(const char* name, K kern, uint32_t* d, int per = 16) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = 8.0 * iters * per;   // 256 CUs x 8 blocks x 4 waves = 8 waves per SIMD
    printf("%-12s %.3f ms  %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / instr_per_simd);
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, (1 << 20) * 4 + 64);
    run("add", k_add, d); run("xor", k_xor, d); run("lshr", k_lshr, d); run("min", k_min, d); run("lshl_add", k_lshladd, d); run("bfe", k_bfe, d);
    run("mul_lo", k_mullo, d); run("mul_hi", k_mulhi, d); run("mul_u24", k_mul24, d);
    run("cmp_eq_u64", k_cmp64, d); run("cmp_eq_u32", k_cmp32, d); run("cmp32->vcc", k_cmp32vcc, d);
    run("cndmask vcc", k_cnd_vcc, d); run("cndmask sgpr", k_cnd_sgpr, d); run("cmp+cndmask", k_cmpcnd, d, 32);
    run("cnd e64 vcc", k_cnd_e64vcc, d); run("cnd_vcc mix4", k_cnd_mix, d, 64); run("cnd_sgpr mix4", k_cnd_mix2, d, 64); run("cmpvcc+cnd", k_cmpvcc_cnd, d, 32); run("nop+cnd vcc", k_nop_cnd, d, 16);
    run("min_dpp", k_mindpp, d); run("mov_dpp", k_movdpp, d); run("alignbit", k_align, d);
    run("add_u64", k_add64, d); run("shl_b64", k_shl64, d); run("bfrev", k_bfrev, d); run("mbcnt", k_mbcnt, d);
    return 0;
}
