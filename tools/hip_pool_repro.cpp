// hip_pool_repro.cpp -- why include/heongpu/heongpu.hpp does NOT allocate with hipMallocAsync.
// On this image (ROCm 7.2.0, gfx950) buffers taken from the device's default stream-ordered pool
// overlap: with a 4 KiB canary on both sides of every allocation and one kernel that fills exactly
// the payload, canaries of LIVE neighbours are overwritten after a few malloc/free rounds on the
// null stream.  Output of one run: profiles/r1h_hip_pool/repro.txt.  The class layer therefore
// carries its own caching allocator over hipMalloc (MemoryPool in heongpu.hpp).
//   hipcc --offload-arch=gfx950 -O1 tools/hip_pool_repro.cpp -o /tmp/hip_pool_repro && /tmp/hip_pool_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
__global__ void fill(uint64_t* p, size_t n, uint64_t v) { size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
int main()
{
    hipMemPool_t pool; (void) hipDeviceGetDefaultMemPool(&pool, 0);
    uint64_t keep = UINT64_MAX; (void) hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    const size_t G = 4096;
    std::vector<size_t> sizes = {11010048, 22020096, 33030144, 11534336, 265289728, 524288, 262144, 57671680};
    int bad = 0;
    std::vector<std::pair<char*, size_t>> live;
    for (int it = 0; it < 200; it++) {
        size_t sz = sizes[(it * 7 + it / 3) % sizes.size()];
        char* p = nullptr;
        (void) hipMallocAsync((void**) &p, sz + 2 * G, nullptr);
        (void) hipMemsetAsync(p, 0xA5, G, nullptr);
        (void) hipMemsetAsync(p + G + sz, 0xA5, G, nullptr);
        fill<<<(sz / 8 + 255) / 256, 256, 0, nullptr>>>((uint64_t*) (p + G), sz / 8, it);
        live.push_back({p, sz});
        if (live.size() > 4) {
            int k = (it * 5) % live.size();
            (void) hipFreeAsync(live[k].first, nullptr);
            live.erase(live.begin() + k);
        }
        (void) hipDeviceSynchronize();
        for (auto& a : live) {
            std::vector<unsigned char> h(2 * G);
            (void) hipMemcpy(h.data(), a.first, G, hipMemcpyDeviceToHost);
            (void) hipMemcpy(h.data() + G, a.first + G + a.second, G, hipMemcpyDeviceToHost);
            size_t d = 0; for (auto c : h) d += c != 0xA5;
            if (d) { bad++; printf("it %d: buffer %zu at %p: %zu guard bytes differ (first %02x)\n", it, a.second, a.first, d, h[0]); }
        }
        if (bad > 5) break;
    }
    printf("done bad=%d\n", bad);
    return 0;
}
