#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(uint4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0); }
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipFree(0);
    const size_t B = 100ull << 30;
    void* p; double t = now(); hipMalloc(&p, B); printf("hipMalloc 100 GiB %.2f s\n", now() - t);
    for (int r = 0; r < 3; ++r) { t = now(); hipMemsetAsync(p, 0, B, 0); hipDeviceSynchronize(); double a = now() - t; printf("hipMemsetAsync 100 GiB #%d: %.3f s = %.0f GB/s\n", r, a, B / a / 1e9); }
    for (int r = 0; r < 2; ++r) { t = now(); fill<<<256 * 32, 256>>>((uint4*)p, B / 16); hipDeviceSynchronize(); double a = now() - t; printf("fill kernel 100 GiB #%d: %.3f s = %.0f GB/s\n", r, a, B / a / 1e9); }
    for (int r = 0; r < 2; ++r) { t = now(); hipMemsetAsync(p, 0xFF, B, 0); hipDeviceSynchronize(); double a = now() - t; printf("hipMemsetAsync 0xFF 100 GiB #%d: %.3f s = %.0f GB/s\n", r, a, B / a / 1e9); }
    return 0;
}
