// pin_probe.hip -- how long page-locked host memory takes to get, by method (the composed planes' host copy is 3 GB for a
// chromosome at -M 100 and 15.5 GB at -M 500: 2.5 s of hipHostMalloc on first use, DESIGN.md 7 "Round 4").
//   build: hipcc --offload-arch=gfx950 -O2 -pthread -o tools/probes/pin_probe tools/probes/pin_probe.hip ; run: tools/probes/pin_probe [GB]
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 4;
    const size_t bytes = gb << 30;
    void *dev = nullptr;
    if (hipMalloc(&dev, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(dev, 1, bytes);
    hipDeviceSynchronize();
    {   // 1. hipHostMalloc
        double t0 = now();
        void *p = nullptr;
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        double t1 = now();
        hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost);
        double t2 = now();
        printf("hipHostMalloc %zu GB: %s alloc %.3f s, D2H copy %.3f s (%.1f GB/s)\n", gb, hipGetErrorString(e), t1 - t0, t2 - t1, gb / (t2 - t1));
        hipHostFree(p);
    }
    for (int threads : {1, 8}) {   // 2. hipHostMalloc in `threads` pieces, side by side
        if (threads == 1) continue;
        double t0 = now();
        std::vector<void *> p(threads, nullptr);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back([&, t]() { hipSetDevice(0); hipHostMalloc(&p[t], bytes / threads, hipHostMallocDefault); });
        for (auto &th : pool) th.join();
        double t1 = now();
        printf("hipHostMalloc in %d pieces side by side: %.3f s\n", threads, t1 - t0);
        for (void *q : p) hipHostFree(q);
    }
    for (int huge : {0, 1}) {      // 3. mmap (+ MADV_HUGEPAGE) + parallel touch + hipHostRegister
        double t0 = now();
        void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) { printf("mmap failed\n"); continue; }
        if (huge) madvise(p, bytes, MADV_HUGEPAGE);
        const int T = 16;
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t]() { char *c = (char *)p + bytes / T * t; for (size_t i = 0; i < bytes / T; i += 4096) c[i] = 0; });
        for (auto &th : pool) th.join();
        double t1 = now();
        hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
        double t2 = now();
        hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost);
        double t3 = now();
        printf("mmap%s + touch on %d threads %.3f s, hipHostRegister %.3f s (%s), D2H copy %.3f s (%.1f GB/s)\n", huge ? " + MADV_HUGEPAGE" : "", T, t1 - t0, t2 - t1,
               hipGetErrorString(e), t3 - t2, gb / (t3 - t2));
        if (e == hipSuccess) hipHostUnregister(p);
        munmap(p, bytes);
    }
    {   // 4. pageable destination
        void *p = malloc(bytes);
        double t0 = now();
        hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost);
        double t1 = now();
        hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost);
        double t2 = now();
        printf("pageable malloc: first D2H copy %.3f s (%.1f GB/s), second %.3f s (%.1f GB/s)\n", t1 - t0, gb / (t1 - t0), t2 - t1, gb / (t2 - t1));
        free(p);
    }
    hipFree(dev);
    return 0;
}
