// How fast can 14.4 GB of ordinary host memory be page-locked (hipHostRegister), whole or in parallel chunks, and is the
// device pointer of a registered range the host pointer (so that chunk-wise registrations form one usable range)?
//   hipcc -O2 tools/ubench/hostreg.cpp -o tools/ubench/hostreg -lpthread
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
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 11.6) * (1ull << 30);
    (void)hipSetDevice(0);
    (void)hipFree(0);
    for (int nthreads : {1, 4, 8, 16}) {
        char *p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) { perror("mmap"); return 1; }
        double t0 = now();
        {   // touch the pages (a caller's array is populated), in parallel
            std::vector<std::thread> th;
            for (int k = 0; k < 16; ++k) th.emplace_back([=] { for (size_t i = bytes / 16 * k; i < bytes / 16 * (k + 1); i += 4096) p[i] = 1; });
            for (auto &t : th) t.join();
        }
        double t1 = now();
        std::vector<int> rc(nthreads, 0);
        {
            std::vector<std::thread> th;
            const size_t chunk = ((bytes / nthreads) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            for (int k = 0; k < nthreads; ++k)
                th.emplace_back([&, k] {
                    (void)hipSetDevice(0);
                    const size_t a = chunk * k, b = a + chunk < bytes ? a + chunk : bytes;
                    if (a < b) rc[k] = (int)hipHostRegister(p + a, b - a, hipHostRegisterMapped);
                });
            for (auto &t : th) t.join();
        }
        double t2 = now();
        void *d0 = nullptr, *d1 = nullptr;
        int g0 = (int)hipHostGetDevicePointer(&d0, p, 0), g1 = (int)hipHostGetDevicePointer(&d1, p + bytes - 4096, 0);
        int bad = 0;
        for (int r : rc) bad |= r;
        printf("%2d thread(s): touch %.3f s, register %.3f s (%.1f GB/s) rc %d; device pointer == host pointer: %d %d (rc %d %d)\n", nthreads,
               t1 - t0, t2 - t1, bytes / 1e9 / (t2 - t1), bad, d0 == (void *)p, d1 == (void *)(p + bytes - 4096), g0, g1);
        // a copy out of the middle of the range spanning chunk borders
        void *dev = nullptr;
        (void)hipMalloc(&dev, 64u << 20);
        double t3 = now();
        int c = (int)hipMemcpy(dev, p + bytes / 2 - (32u << 20), 64u << 20, hipMemcpyHostToDevice);
        double t4 = now();
        printf("    64 MB copy across the middle: rc %d, %.1f GB/s\n", c, (64u << 20) / 1e9 / (t4 - t3));
        (void)hipFree(dev);
        double t5 = now();
        {
            const size_t chunk = ((bytes / nthreads) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            for (int k = 0; k < nthreads; ++k) { const size_t a = chunk * k; if (a < bytes) (void)hipHostUnregister(p + a); }
        }
        double t6 = now();
        printf("    unregister %.3f s\n", t6 - t5);
        munmap(p, bytes);
    }
    return 0;
}
