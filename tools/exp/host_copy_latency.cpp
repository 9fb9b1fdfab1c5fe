#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main() {
    void* d; hipMalloc(&d, 1 << 20);
    hipStream_t st; hipStreamCreate(&st);
    for (size_t sz : {16384ul, 32768ul, 65536ul, 131072ul}) {
        double ta = 0, tc = 0, tf = 0, tp = 0;
        for (int r = 0; r < 12; r++) {
            auto t0 = std::chrono::steady_clock::now();
            void* h; hipHostMalloc(&h, sz, hipHostMallocDefault);
            auto t1 = std::chrono::steady_clock::now();
            hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
            auto t2 = std::chrono::steady_clock::now();
            hipHostFree(h);
            auto t3 = std::chrono::steady_clock::now();
            std::vector<char> v(sz);
            hipMemcpyAsync(v.data(), d, sz, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
            auto t4 = std::chrono::steady_clock::now();
            if (r >= 2) { ta += std::chrono::duration<double, std::micro>(t1 - t0).count(); tc += std::chrono::duration<double, std::micro>(t2 - t1).count();
                          tf += std::chrono::duration<double, std::micro>(t3 - t2).count(); tp += std::chrono::duration<double, std::micro>(t4 - t3).count(); }
        }
        printf("%7zu B: hipHostMalloc %.0f us, D2H pinned %.0f us, hipHostFree %.0f us, D2H pageable %.0f us\n", sz, ta / 10, tc / 10, tf / 10, tp / 10);
    }
}
