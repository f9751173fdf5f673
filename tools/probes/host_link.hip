#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void copyk(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main() {
    const size_t CH = 32u << 20; const int NB = 4, REP = 64;
    uint8_t* h[NB]; uint8_t* d[NB];
    for (int i = 0; i < NB; ++i) { hipHostMalloc((void**)&h[i], CH, hipHostMallocDefault); memset(h[i], i + 1, CH); hipMalloc((void**)&d[i], CH); }
    hipStream_t s[2]; hipStreamCreate(&s[0]); hipStreamCreate(&s[1]);
    hipMemcpyAsync(d[0], h[0], CH, hipMemcpyHostToDevice, s[0]); hipDeviceSynchronize();
    double t = now();
    for (int r = 0; r < REP; ++r) hipMemcpyAsync(d[r & 1], h[r & 1], CH, hipMemcpyHostToDevice, s[0]);
    hipDeviceSynchronize();
    printf("H2D one stream      %.1f GB/s\n", REP * CH / (now() - t) / 1e9);
    t = now();
    for (int r = 0; r < REP; ++r) hipMemcpyAsync(d[r & 3], h[r & 3], CH, hipMemcpyHostToDevice, s[r & 1]);
    hipDeviceSynchronize();
    printf("H2D two streams     %.1f GB/s\n", REP * CH / (now() - t) / 1e9);
    for (int blocks : {256, 1024, 4096}) {
        copyk<<<blocks, 256, 0, s[0]>>>((const uint4*)h[0], (uint4*)d[0], CH / 16); hipDeviceSynchronize();
        t = now();
        for (int r = 0; r < REP; ++r) copyk<<<blocks, 256, 0, s[0]>>>((const uint4*)h[r & 3], (uint4*)d[r & 3], CH / 16);
        hipDeviceSynchronize();
        printf("kernel reads pinned host (%d blocks) %.1f GB/s\n", blocks, REP * CH / (now() - t) / 1e9);
    }
    // page cache -> pinned
    const char* path = "/tmp/h2d_probe.bin";
    const size_t FS = 2ull << 30;
    { int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0600); std::vector<uint8_t> z(64 << 20, 7); for (size_t o = 0; o < FS; o += z.size()) if (write(fd, z.data(), z.size()) < 0) return 1; close(fd); }
    int fd = open(path, O_RDONLY);
    for (int T : {4, 8, 16, 32}) {
        t = now();
        for (size_t off = 0; off < FS; off += CH) {
            std::vector<std::thread> th;
            uint8_t* dst = h[(off / CH) & 3];
            for (int k = 0; k < T; ++k) th.emplace_back([=]() { size_t lo = CH * k / T, hi = CH * (k + 1) / T; while (lo < hi) { ssize_t r = pread(fd, dst + lo, hi - lo, off + lo); if (r <= 0) break; lo += r; } });
            for (auto& x : th) x.join();
        }
        printf("pread page cache -> pinned, %2d threads  %.1f GB/s\n", T, FS / (now() - t) / 1e9);
    }
    // both at once: pread next chunk while H2D of the previous
    for (int T : {8, 16}) {
        t = now();
        int turn = 0;
        for (size_t off = 0; off < FS; off += CH) {
            std::vector<std::thread> th;
            uint8_t* dst = h[turn];
            hipStreamSynchronize(s[turn]);
            for (int k = 0; k < T; ++k) th.emplace_back([=]() { size_t lo = CH * k / T, hi = CH * (k + 1) / T; while (lo < hi) { ssize_t r = pread(fd, dst + lo, hi - lo, off + lo); if (r <= 0) break; lo += r; } });
            for (auto& x : th) x.join();
            hipMemcpyAsync(d[turn], h[turn], CH, hipMemcpyHostToDevice, s[turn]);
            turn ^= 1;
        }
        hipDeviceSynchronize();
        printf("pread || H2D (double buffer, two streams), %2d threads  %.1f GB/s\n", T, FS / (now() - t) / 1e9);
    }
    unlink(path);
    return 0;
}
