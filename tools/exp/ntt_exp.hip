// ntt_exp.hip -- standalone timing harness for NTT kernel variants.
// Includes the product kernel source directly; NTT_EXP_MODE selects ablations:
//   0 normal, 1 memory only (no butterflies), 2 compute only (no global traffic)
#ifndef NTT_EXP_MODE
#define NTT_EXP_MODE 0
#endif
#define NTT_ABLATION_HEADER "../../tools/exp/ntt_ablation.cuh"
#include "../../heongpu_amd/csrc/ntt.hip"
#include "ntt_single_persistent.cuh"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace hegpu;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv)
{
    int n_power = argc > 1 ? atoi(argv[1]) : 16;
    int polys = argc > 2 ? atoi(argv[2]) : 17 * 512;
    int reps = argc > 3 ? atoi(argv[3]) : 10;
    int chunk = argc > 4 ? atoi(argv[4]) : 0; // polys per launch pair (0 = all)
    int single = argc > 5 ? atoi(argv[5]) : 0; // NttArgs::single_pass; 2 / 3 / 4: the persistent experiments (forward, N = 2^14)
    int all_fp = argc > 6 ? atoi(argv[6]) : 0; // every modulus below 2^50
    const int mods = 17;
    const u64 n = 1ull << n_power;
    std::vector<Mod> hm(mods);
    std::vector<ulonglong2> htw(mods * n), hn(mods);
    u64 q = (1ull << 60) - (1ull << 18) + 1; // not prime; timing only
    for (int k = 0; k < mods; k++) { hm[k] = make_mod(((k == 0 || k == mods - 1) && !all_fp) ? q - 2 * k * (1 << 17) : (1ull << 50) - (1ull << 18) * (k + 3) + 1); hn[k] = make_ulonglong2(12345, shoup_companion(12345, hm[k].q)); }
    for (u64 i = 0; i < mods * n; i++) { u64 w = (i * 0x9E3779B97F4A7C15ull) % q; htw[i] = make_ulonglong2(w, shoup_companion(w, q)); }
    if (all_fp) { // FP64 butterflies: tables of (double(w), RN(w / q)) pairs, Mod::fp set (as context.cpp builds them)
        for (int k = 0; k < mods; k++) {
            hm[k].fp = 1;
            for (u64 i = 0; i < n; i++) {
                const double w = (double) (htw[k * n + i].x % hm[k].q), wi = w / (double) hm[k].q;
                u64 a_, b_;
                memcpy(&a_, &w, 8);
                memcpy(&b_, &wi, 8);
                htw[k * n + i] = make_ulonglong2(a_, b_);
            }
        }
    }
    NttArgs a{};
    CK(hipMalloc((void**) &a.mods, mods * sizeof(Mod)));
    CK(hipMalloc((void**) &a.tw, mods * n * 16));
    CK(hipMalloc((void**) &a.itw, mods * n * 16));
    CK(hipMalloc((void**) &a.twB, mods * n * 16));
    CK(hipMalloc((void**) &a.itwB, mods * n * 16));
    CK(hipMemcpy((void*) a.twB, htw.data(), mods * n * 16, hipMemcpyHostToDevice));
    CK(hipMalloc((void**) &a.twB8, mods * n * 8));
    {
        std::vector<double> h8(mods * n);
        for (u64 i = 0; i < mods * n; i++) h8[i] = (double) (htw[i].x % hm[i / n].q);
        CK(hipMemcpy((void*) a.twB8, h8.data(), mods * n * 8, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy((void*) a.itwB, htw.data(), mods * n * 16, hipMemcpyHostToDevice));
    CK(hipMalloc((void**) &a.ninv, mods * 16));
    CK(hipMalloc((void**) &a.w1ninv, mods * 16));
    CK(hipMemcpy((void*) a.mods, hm.data(), mods * sizeof(Mod), hipMemcpyHostToDevice));
    CK(hipMemcpy((void*) a.tw, htw.data(), mods * n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy((void*) a.itw, htw.data(), mods * n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy((void*) a.ninv, hn.data(), mods * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy((void*) a.w1ninv, hn.data(), mods * 16, hipMemcpyHostToDevice));
    u64 *in, *out;
    CK(hipMalloc((void**) &in, polys * n * 8));
    CK(hipMalloc((void**) &out, polys * n * 8));
    std::vector<u64> h(n * 64);
    for (u64 i = 0; i < h.size(); i++) h[i] = (i * 0xBF58476D1CE4E5B9ull) % q;
    for (int p = 0; p < polys; p += 64) CK(hipMemcpy(in + p * n, h.data(), (size_t) std::min(64, polys - p) * n * 8, hipMemcpyHostToDevice));
    a.in = in; a.out = out; a.n_power = n_power; a.mod_count = mods; a.single_pass = single > 1 ? 1 : single; a.lazy_q_max = 1ull << 57;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int inverse = 0; inverse < 2; inverse++) {
        CK(ntt_launch(a, polys, inverse, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) {
            if (single > 1 && !inverse && n_power == 14) CK(launch_single_persistent<6>(a, polys, single, 0));
            else if (!chunk) CK(ntt_launch(a, polys, inverse, 0));
            else
                for (int p0 = 0; p0 < polys; p0 += chunk) {
                    NttArgs c = a;
                    c.in = a.in + (u64) p0 * n;
                    c.out = a.out + (u64) p0 * n;
                    CK(ntt_launch(c, std::min(chunk, polys - p0), inverse, 0));
                }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("mode %d chunk %d N=2^%d polys=%d %s: %.3f ms  %.1f GB/s (2W/limb)  %.3f M limb-NTT/s\n", NTT_EXP_MODE, chunk, n_power, polys,
               inverse ? "inv" : "fwd", ms, polys * 2.0 * n * 8 / (ms * 1e-3) / 1e9, polys / ms * 1e-3);
    }
    return 0;
}
