#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#include <thread>
#include <atomic>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void copyk(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void spin(uint32_t* p, int n) { uint32_t x = threadIdx.x; for (int i = 0; i < n; ++i) x = x * 1664525u + 1013904223u; p[blockIdx.x * 256 + threadIdx.x] = x; }
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipFree(0);
    const size_t G = 1ull << 30, CH = 32u << 20;
    uint8_t* h[2]; uint8_t* d[2]; uint32_t* w;
    for (int i = 0; i < 2; ++i) { hipHostMalloc((void**)&h[i], CH, hipHostMallocDefault); hipMalloc((void**)&d[i], CH); }
    hipMalloc((void**)&w, 256 * 2048 * 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; ++mode) {          // 0: H2D copies, 1: kernel reads pinned host, 2: compute kernels only
        std::atomic<bool> stop{false}, started{false};
        std::atomic<double> t_alloc0{0}, t_alloc1{0};
        void* big = nullptr;
        std::thread a([&]() { hipSetDevice(0); while (!started) std::this_thread::sleep_for(std::chrono::milliseconds(1)); std::this_thread::sleep_for(std::chrono::milliseconds(500));
                              t_alloc0 = now(); hipMalloc(&big, 160 * G); t_alloc1 = now(); std::this_thread::sleep_for(std::chrono::milliseconds(500)); stop = true; });
        std::vector<std::pair<double, double>> log;    // (time, GB or launches done in the slice)
        double t0 = now(); started = true;
        while (!stop) {
            double ts = now(); int n = 0;
            while (now() - ts < 0.25) {
                if (mode == 0) hipMemcpyAsync(d[n & 1], h[n & 1], CH, hipMemcpyHostToDevice, s);
                else if (mode == 1) copyk<<<256, 256, 0, s>>>((const uint4*)h[n & 1], (uint4*)d[n & 1], CH / 16);
                else spin<<<2048, 256, 0, s>>>(w, 20000);
                hipStreamSynchronize(s); ++n;
            }
            log.push_back({ts - t0, n / (now() - ts)});
        }
        a.join();
        printf("mode %d (%s): hipMalloc 160 GiB from %.2f to %.2f s;  ops/s per 0.25 s slice:", mode, mode == 0 ? "H2D 32 MB" : mode == 1 ? "kernel reads pinned 32 MB" : "compute kernel", t_alloc0.load() - t0, t_alloc1.load() - t0);
        for (auto& x : log) printf(" %.0f", x.second);
        printf("\n");
        hipFree(big);
        // make the next hipMalloc cold again? (the runtime may keep the block) -- allocate a different size each time
    }
    return 0;
}
