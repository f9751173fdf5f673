// kcfx_membench -- random-access microbenchmarks that size the table design (not part of the product path).
// Each lane reads BYTES contiguous bytes at a pseudo-random, BYTES-aligned (or LINE-aligned) offset of a big buffer.
//   usage: kcfx_membench [table_GiB=3] [accesses_M=64]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t h32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x;
}

// per-lane gather: each lane reads VEC x 16 bytes from its own random LINE-byte-aligned line. ILP independent probes per lane.
template <int VEC, int LINE, int ILP>
__global__ __launch_bounds__(256) void gather_lane(const uint8_t* __restrict__ buf, uint32_t n_lines, uint64_t n_acc, uint32_t* out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n_acc; i += stride * ILP) {
        uint4 v[ILP][VEC];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            const uint32_t line = __umulhi(h32((uint32_t)(i + u * stride) * 2654435761u + 12345u), n_lines);
            const uint4* p = reinterpret_cast<const uint4*>(buf + (size_t)line * LINE);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[u][j] = p[j];
        }
#pragma unroll
        for (int u = 0; u < ILP; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc += v[u][j].x ^ v[u][j].y ^ v[u][j].z ^ v[u][j].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// cooperative gather: G lanes share one random line of G*16 bytes (one dwordx4 each); 64/G lines per wave-instruction
template <int G, int ILP>
__global__ __launch_bounds__(256) void gather_coop(const uint8_t* __restrict__ buf, uint32_t n_lines, uint64_t n_acc, uint32_t* out) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t i = tid / G;
    const uint32_t sub = tid % G;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) / G;
    uint32_t acc = 0;
    for (; i < n_acc; i += stride * ILP) {
        uint4 v[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            const uint32_t line = __umulhi(h32((uint32_t)(i + u * stride) * 2654435761u + 12345u), n_lines);
            v[u] = reinterpret_cast<const uint4*>(buf + (size_t)line * (G * 16))[sub];
        }
#pragma unroll
        for (int u = 0; u < ILP; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the access pattern of screen_k31c: RUN consecutive lanes read one 8-byte cell each (consecutive cells) of the same random
// 128-byte line, LOADS of them per lane (the second 8 cells further, mod 16), ILP independent words per trip
template <int RUN, int LOADS, int ILP, bool NT = false>
__global__ __launch_bounds__(256) void gather_runs(const uint8_t* __restrict__ buf, uint32_t n_lines, uint64_t n_acc, uint32_t* out) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t i = tid;                                   // one access = one lane's cell(s)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n_acc; i += stride * ILP) {
        uint2 v[ILP][LOADS];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            const uint64_t a = i + u * stride;
            const uint32_t line = __umulhi(h32((uint32_t)(a / RUN) * 2654435761u + 12345u), n_lines);
            const uint32_t cell = (uint32_t)(a % RUN + (a / RUN) * 5) & 15u;
            const uint8_t* p = buf + (size_t)line * 128;
#pragma unroll
            for (int l = 0; l < LOADS; ++l) {
                const uint64_t* q = reinterpret_cast<const uint64_t*>(p + 8 * ((cell + 8 * l) & 15u));
                const uint64_t w = NT ? __builtin_nontemporal_load(q) : *q;      // NT: "nt" cache-policy bit on the load
                v[u][l] = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
            }
        }
#pragma unroll
        for (int u = 0; u < ILP; ++u)
#pragma unroll
            for (int l = 0; l < LOADS; ++l) acc += v[u][l].x ^ v[u][l].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// dependent-load latency: every lane chases its own chain of random 128-byte-aligned 16-byte loads
__global__ void chase(const uint8_t* __restrict__ buf, uint32_t n_lines, int steps, uint32_t* out) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 99u;
    for (int i = 0; i < steps; ++i) {
        const uint32_t line = __umulhi(h32(x + i), n_lines);
        const uint4 v = *reinterpret_cast<const uint4*>(buf + (size_t)line * 128);
        x = x * 31u + (v.x ^ v.y ^ v.z ^ v.w);
    }
    if (x == 0x12345678u) out[0] = x;
}

__global__ void stream_read(const uint4* __restrict__ buf, uint64_t n_vec, uint32_t* out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n_vec; i += stride) { uint4 v = buf[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void fill(uint4* buf, uint64_t n_vec) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_vec; i += stride) buf[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

template <typename F>
static void timeit(const char* name, double accesses, double bytes_per_access, F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-34s %8.3f ms  %8.2f G acc/s  %8.1f GB/s (requested bytes)\n", name, best, accesses / best / 1e6, accesses * bytes_per_access / best / 1e6);
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 3.0;
    const uint64_t n_acc = (uint64_t)((argc > 2 ? atof(argv[2]) : 64.0) * 1e6);
    const size_t bytes = (size_t)(gib * (1ull << 30)) / 4096 * 4096;
    uint8_t* buf; uint32_t* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("table %.2f GiB, %llu accesses per launch\n", gib, (unsigned long long)n_acc);
    const dim3 g(256 * 8), b(256);
    if (argc > 3 && !strcmp(argv[3], "box")) {     // "box": ONE number for bench.py -- this box's random-line rate for the screening
                                                    // kernel's own pattern (runs of 8 lanes reading 8-byte cells of one random
                                                    // 128-byte line, two loads per lane), as a JSON line
        // ... and the same with ONE non-temporal load per lane, the form the k = 31 kernel issues since round 3
        float best = 1e30f, best_nt = 1e30f;
        hipEvent_t a, b2;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b2));
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((gather_runs<8, 2, 1>), g, b, 0, 0, buf, (uint32_t)(bytes / 128), n_acc, out);
            CK(hipEventRecord(b2));
            CK(hipEventSynchronize(b2));
            float ms; CK(hipEventElapsedTime(&ms, a, b2));
            if (r > 0 && ms < best) best = ms;
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((gather_runs<8, 1, 1, true>), g, b, 0, 0, buf, (uint32_t)(bytes / 128), n_acc, out);
            CK(hipEventRecord(b2));
            CK(hipEventSynchronize(b2));
            CK(hipEventElapsedTime(&ms, a, b2));
            if (r > 0 && ms < best_nt) best_nt = ms;
        }
        printf("{\"pattern\": \"runs of 8 lanes x 8 B of one random 128-byte line, 2 loads per lane\", \"table_GiB\": %.1f, \"lines_per_launch\": %llu, "
               "\"ms\": %.4f, \"G_lines_per_s\": %.2f, \"ms_one_nontemporal_load\": %.4f, \"G_lines_per_s_one_nontemporal_load\": %.2f}\n", gib,
               (unsigned long long)(n_acc / 8), best, (double)(n_acc / 8) / best / 1e6, best_nt, (double)(n_acc / 8) / best_nt / 1e6);
        return 0;
    }
    if (argc > 3) {     // "runs": only the screen_k31c pattern; rates are per lane access -- lines/s = accesses/s / RUN
#define RUNS(R, L, I) timeit("runs of " #R " lanes x8B, " #L " load(s), ILP " #I, (double)n_acc, 8 * L, [&] { \
        hipLaunchKernelGGL((gather_runs<R, L, I>), g, b, 0, 0, buf, (uint32_t)(bytes / 128), n_acc, out); });
#define RUNS_NT(R, L, I) timeit("runs of " #R " lanes x8B, " #L " load(s), ILP " #I ", non-temporal", (double)n_acc, 8 * L, [&] { \
        hipLaunchKernelGGL((gather_runs<R, L, I, true>), g, b, 0, 0, buf, (uint32_t)(bytes / 128), n_acc, out); });
        RUNS_NT(8, 2, 1) RUNS_NT(8, 1, 2)
        RUNS(8, 1, 1) RUNS(8, 2, 1) RUNS(8, 1, 2) RUNS(8, 2, 2) RUNS(8, 2, 4) RUNS(4, 1, 2) RUNS(16, 1, 2) RUNS(1, 1, 1) RUNS(1, 1, 4)
        return 0;
    }
    timeit("stream read 16B/lane", (double)bytes / 16, 16, [&] { hipLaunchKernelGGL(stream_read, g, b, 0, 0, (const uint4*)buf, bytes / 16, out); });
#define LANE(V, L, I) timeit("lane " #V "x16B of " #L "B line, ILP " #I, (double)n_acc, V * 16, [&] { \
        hipLaunchKernelGGL((gather_lane<V, L, I>), g, b, 0, 0, buf, (uint32_t)(bytes / L), n_acc, out); });
    LANE(1, 16, 1) LANE(1, 32, 1) LANE(1, 64, 1) LANE(1, 128, 1)
    LANE(2, 32, 1) LANE(2, 64, 1)
    LANE(4, 64, 1) LANE(4, 64, 2) LANE(4, 64, 4)
    LANE(4, 128, 1) LANE(8, 128, 1) LANE(8, 128, 2)
    LANE(1, 64, 4) LANE(2, 32, 4)
#define COOP(G, I) timeit("coop " #G " lanes x16B, ILP " #I, (double)n_acc, G * 16, [&] { \
        hipLaunchKernelGGL((gather_coop<G, I>), g, b, 0, 0, buf, (uint32_t)(bytes / (G * 16)), n_acc, out); });
    COOP(2, 1) COOP(2, 4) COOP(4, 1) COOP(4, 2) COOP(4, 4) COOP(4, 8) COOP(8, 1) COOP(8, 4) COOP(16, 1) COOP(16, 4)
    // latency: one wave on the whole chip, then one wave per CU, then 8 waves per CU (dependent chains of 2000 loads)
    for (int blocks : {1, 256, 2048}) {
        hipEvent_t a, b2;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b2));
        const int steps = 2000;
        hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(bytes / 128), steps, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(bytes / 128), steps, out);
        CK(hipEventRecord(b2));
        CK(hipEventSynchronize(b2));
        float ms; CK(hipEventElapsedTime(&ms, a, b2));
        printf("chase: %4d waves x 64 lanes, %d dependent loads: %.1f ns per load (%.2f G loads/s)\n", blocks, steps, ms * 1e6 / steps,
               (double)blocks * 64 * steps / ms / 1e6);
    }
    return 0;
}
