#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint64_t* p, size_t n, size_t stride) { size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = i; }
int main() {
    hipFree(0); setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t G = 1ull << 30;
    { void* p; double t = now(); hipMalloc(&p, 64 * G); double a = now() - t; t = now(); hipFree(p); printf("hipMalloc 64 GiB %.2f s (%.1f GB/s), free %.2f s\n", a, 64 * 1.0737 / a, now() - t); }
    { void* p; double t = now(); hipMalloc(&p, 64 * G); double a = now() - t; t = now(); hipFree(p); printf("again: hipMalloc 64 GiB %.2f s (%.1f GB/s), free %.2f s\n", a, 64 * 1.0737 / a, now() - t); }
    { void* p; double t = now(); hipError_t e = hipMalloc(&p, 192 * G); double a = now() - t; t = now(); hipFree(p); printf("hipMalloc 192 GiB (%d) %.2f s (%.1f GB/s), free %.2f s\n", (int)e, a, 192 * 1.0737 / a, now() - t); }
    { void* p[3]; double t = now(); std::vector<std::thread> th; for (int i = 0; i < 3; ++i) th.emplace_back([&, i]() { hipSetDevice(0); hipMalloc(&p[i], 64 * G); }); for (auto& x : th) x.join();
      double a = now() - t; t = now(); for (int i = 0; i < 3; ++i) hipFree(p[i]); printf("3 threads x hipMalloc 64 GiB %.2f s (%.1f GB/s), free %.2f s\n", a, 192 * 1.0737 / a, now() - t); }
    { void* p[12]; double t = now(); std::vector<std::thread> th; for (int i = 0; i < 12; ++i) th.emplace_back([&, i]() { hipSetDevice(0); hipMalloc(&p[i], 16 * G); }); for (auto& x : th) x.join();
      double a = now() - t; t = now(); for (int i = 0; i < 12; ++i) hipFree(p[i]); printf("12 threads x hipMalloc 16 GiB %.2f s (%.1f GB/s), free %.2f s\n", a, 192 * 1.0737 / a, now() - t); }
    // VMM: one reserved range, physical chunks created and mapped by several threads
    {
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        printf("VMM granularity %zu (%d)\n", gran, (int)e);
        for (int T : {1, 4, 12}) {
            const size_t total = 192 * G, chunk = total / 12;
            void* base = nullptr; double t = now();
            e = hipMemAddressReserve(&base, total, 0, nullptr, 0);
            if (e != hipSuccess) { printf("reserve failed %d\n", (int)e); break; }
            std::vector<hipMemGenericAllocationHandle_t> hd(12);
            std::vector<std::thread> th; std::vector<int> err(12, 0);
            for (int w = 0; w < T; ++w) th.emplace_back([&, w]() { hipSetDevice(0); for (int i = w; i < 12; i += T) { err[i] = (int)hipMemCreate(&hd[i], chunk, &prop, 0); if (!err[i]) err[i] = (int)hipMemMap((char*)base + i * chunk, chunk, 0, hd[i], 0); } });
            for (auto& x : th) x.join();
            hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(base, total, &ad, 1);
            double a = now() - t; int bad = 0; for (int x : err) bad |= x;
            hipError_t e2 = hipSuccess; if (!bad && e == hipSuccess) { e2 = hipMemsetAsync(base, 0, 1 << 20, 0); if (e2 == hipSuccess) e2 = hipMemsetAsync((char*)base + total - (1 << 20), 0, 1 << 20, 0); if (e2 == hipSuccess) e2 = hipDeviceSynchronize(); }
            double t2 = now();
            hipMemUnmap(base, total); for (auto& x : hd) hipMemRelease(x); hipMemAddressFree(base, total);
            printf("VMM 12 x 16 GiB, %2d threads: %.2f s (%.1f GB/s) err %d access %d touch %d, release %.2f s\n", T, a, 192 * 1.0737 / a, bad, (int)e, (int)e2, now() - t2);
        }
    }
    return 0;
}
