// churn <GiB> [hold_ms]: take <GiB> of device memory in 4 GiB blocks, write it, exit WITHOUT freeing (what a finished job
// leaves for the driver to clear). The next process then sees what a busy node gives it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int gib = argc > 1 ? atoi(argv[1]) : 64;
    const int hold = argc > 2 ? atoi(argv[2]) : 0;
    double t = now();
    (void)hipFree(0);
    const double t_init = now() - t;
    std::vector<void*> p;
    t = now();
    for (int i = 0; i < gib; i += 4) {
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)std::min(4, gib - i) << 30) != hipSuccess) break;
        p.push_back(q);
    }
    const double t_alloc = now() - t;
    t = now();
    for (size_t i = 0; i < p.size(); ++i) (void)hipMemsetAsync(p[i], 0x5a, (size_t)std::min(4, gib - 4 * (int)i) << 30, 0);
    (void)hipDeviceSynchronize();
    const double t_fill = now() - t;
    fprintf(stderr, "[churn] init %.3f s, %zu blocks (%d GiB) in %.3f s = %.1f GB/s, filled in %.3f s\n", t_init, p.size(), gib, t_alloc, gib * 1.0737 / t_alloc, t_fill);
    if (hold) std::this_thread::sleep_for(std::chrono::milliseconds(hold));
    _exit(0);
}
