// ntt.hip -- batched negacyclic NTT / INTT over Z_q[X]/(X^N+1) for gfx950.
//
// Replaces gpuntt::GPU_NTT{,_Inplace}, GPU_INTT{,_Inplace},
// GPU_NTT_Modulus_Ordered_Inplace and GPU_NTT_Poly_Ordered_Inplace of the
// reference's unvendored thirdparty/GPU-NTT (call sites: reference
// src/lib/host/ckks/operator.cu:919,956,996,1011,1197; bfv/operator.cu:393,410).
// Definition: natural-order coefficients in, bit-reversed
// evaluation order out; stage with m groups uses root table[m + group],
// table[r] = psi^bitreverse(r); inverse = Gentleman-Sande + N^-1.
//
// MI355X design (NOT a translation of GPU-NTT's CUDA kernels):
//  * two passes per transform, N = 2^S1 x 256: a "column" pass does the S1
//    large-stride stages on 4096-element tiles [2^S1 rows][4096>>S1 cols],
//    a "row" pass does the 8 contiguous stages on 16 rows of 256.  Every
//    global access is a 128-byte-or-wider coalesced segment; each pass moves
//    the limb exactly once (2W bytes per pass).
//  * 256 threads x 16 coefficients per workgroup: radix-16 butterflies run
//    entirely in VGPRs (4 stages per LDS exchange), LDS (34-36 KiB per
//    workgroup, padded against bank conflicts) is only the transpose medium.
//  * Shoup/Harvey lazy butterflies: one mulhi64 + two mullo64 per butterfly
//    (10 v_mad_u64_u32/v_mul_lo_u32), values kept in [0,4q) (forward) /
//    [0,2q) (inverse) with a single exact correction at the end, so results
//    are canonical residues -- bit-identical to any exact NTT.
//  * wave-uniform twiddles (column pass, first round) are fetched through
//    the scalar cache; the rest are 16-byte (w, w') pairs read as dwordx4.
#include "ntt.hpp"

namespace hegpu {

#define NTT_THREADS 256

__device__ __forceinline__ u64 csub(u64 x, u64 m) { return (x >= m) ? x - m : x; }

// Harvey CT butterfly, x,y in [0,4q) -> [0,4q)
__device__ __forceinline__ void ct_bfly(u64& x, u64& y, ulonglong2 w, u64 q, u64 q2)
{
    u64 u = csub(x, q2);
    u64 t = mul_shoup_lazy(y, w.x, w.y, q);
    x = u + t;
    y = u - t + q2;
}

// GS butterfly, x,y in [0,2q) -> [0,2q)
__device__ __forceinline__ void gs_bfly(u64& x, u64& y, ulonglong2 w, u64 q, u64 q2)
{
    u64 s = x + y;
    u64 d = x - y + q2;
    x = csub(s, q2);
    y = mul_shoup_lazy(d, w.x, w.y, q);
}

// LOGR Cooley-Tukey stages on 2^LOGR register-resident values.  Local stage
// s, block b uses root index (root0 << s) + b.
template <int LOGR>
__device__ __forceinline__ void ct_radix(u64 (&x)[1 << LOGR], const ulonglong2* __restrict__ tw,
                                         u32 root0, u64 q, u64 q2)
{
#pragma unroll
    for (int s = 0; s < LOGR; s++) {
        const int half = (1 << LOGR) >> (s + 1);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            ulonglong2 w = tw[(root0 << s) + b];
#pragma unroll
            for (int j = 0; j < half; j++)
                ct_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, q, q2);
        }
    }
}

// LOGR Gentleman-Sande stages (reverse order of ct_radix).
template <int LOGR>
__device__ __forceinline__ void gs_radix(u64 (&x)[1 << LOGR], const ulonglong2* __restrict__ tw,
                                         u32 root0, u64 q, u64 q2)
{
#pragma unroll
    for (int s = LOGR - 1; s >= 0; s--) {
        const int half = (1 << LOGR) >> (s + 1);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            ulonglong2 w = tw[(root0 << s) + b];
#pragma unroll
            for (int j = 0; j < half; j++)
                gs_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, q, q2);
        }
    }
}

// Same, but the very last stage (s == 0, global stage 0) folds N^-1 in and
// fully reduces: x' = (x+y)*ninv, y' = (x-y)*(w1*ninv).
template <int LOGR>
__device__ __forceinline__ void gs_radix_last(u64 (&x)[1 << LOGR], const ulonglong2* __restrict__ tw,
                                              u32 root0, ulonglong2 ninv, ulonglong2 w1ninv,
                                              u64 q, u64 q2)
{
#pragma unroll
    for (int s = LOGR - 1; s >= 1; s--) {
        const int half = (1 << LOGR) >> (s + 1);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            ulonglong2 w = tw[(root0 << s) + b];
#pragma unroll
            for (int j = 0; j < half; j++)
                gs_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, q, q2);
        }
    }
    const int half = (1 << LOGR) >> 1;
#pragma unroll
    for (int j = 0; j < half; j++) {
        u64 s = x[j] + x[j + half];
        u64 d = x[j] - x[j + half] + q2;
        x[j] = mul_shoup(s, ninv.x, ninv.y, q);
        x[j + half] = mul_shoup(d, w1ninv.x, w1ninv.y, q);
    }
}

struct PolySel {
    u64 in_off, out_off; // element offsets of this polynomial
    int mod;             // modulus index into the plan
};

__device__ __forceinline__ PolySel select_poly(const NttArgs& a, int poly)
{
    PolySel s;
    int item = 0, j = poly;
    if (a.polys_per_item) {
        item = poly / a.polys_per_item;
        j = poly - item * a.polys_per_item;
    }
    int k = j % a.mod_count;
    if (a.mod_order) k = a.mod_order[k];
    s.mod = a.mod_offset + k;
    u64 slot = a.poly_order ? (u64) a.poly_order[j] : (u64) j;
    u64 in_slot = a.decomp_mods ? (u64) (j / a.decomp_mods) : slot;
    s.in_off = (u64) item * a.in_item_stride + (in_slot << a.n_power);
    s.out_off = (u64) item * a.out_item_stride + (slot << a.n_power);
    return s;
}

// column-tile LDS index (pad 16 elements per 256 against the 2-way conflict
// of the 16-column tile)
__device__ __forceinline__ int col_phys(int e) { return e + ((e >> 8) << 4); }
// row-tile LDS index (pad 2 elements per 16 so 128-byte-strided b128 reads
// spread over all banks)
__device__ __forceinline__ int row_phys(int e) { return e + ((e >> 4) << 1); }

#define COL_LDS_ELEMS (4096 + 256)
#define ROW_LDS_ELEMS (4096 + 512)

// ------------------------------------------------------------------ forward
// Column pass: stages 0..S1-1 (row stride 256).  grid = (256/CT, batch).
template <int S1, bool DECOMP>
__global__ __launch_bounds__(NTT_THREADS) void ntt_fwd_col(NttArgs a)
{
    constexpr int R = 1 << S1;
    constexpr int CT = 4096 / R;
    constexpr int NSA = S1 - 4;
    constexpr int RA = 1 << NSA;
    constexpr int G = 16 / RA;
    __shared__ u64 lds[(NSA > 0) ? COL_LDS_ELEMS : 1];

    const int t = threadIdx.x;
    const PolySel ps = select_poly(a, blockIdx.y);
    const Mod md = a.mods[ps.mod];
    const u64 q = md.q, q2 = 2 * md.q;
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    const u64* __restrict__ src = a.in + ps.in_off + blockIdx.x * CT;
    u64* __restrict__ dst = a.out + ps.out_off + blockIdx.x * CT;

    u64 x[16];
    const int col = t % CT, r1 = t / CT;
    if constexpr (NSA > 0) {
        // round A: G groups of radix RA, rows rbase + 16k
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int L = t + NTT_THREADS * g;
            const int c = L % CT, rb = L / CT;
            u64 y[RA];
#pragma unroll
            for (int k = 0; k < RA; k++) y[k] = src[(u64) (rb + 16 * k) * 256 + c];
            if (DECOMP) {
#pragma unroll
                for (int k = 0; k < RA; k++) y[k] = reduce64(y[k], md);
            }
            ct_radix<NSA>(y, tw, 1u, q, q2);
#pragma unroll
            for (int k = 0; k < RA; k++) lds[col_phys((rb + 16 * k) * CT + c)] = y[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = lds[col_phys((16 * r1 + k) * CT + col)];
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = src[(u64) k * 256 + col];
        if (DECOMP) {
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = reduce64(x[k], md);
        }
    }
    ct_radix<4>(x, tw, (u32) (RA + r1), q, q2);
#pragma unroll
    for (int k = 0; k < 16; k++) dst[(u64) (16 * r1 + k) * 256 + col] = x[k];
}

// Row pass: stages S1..S1+7 on contiguous rows of 256, 16 rows per block.
// Values arrive in [0,4q) from the column pass; the result is fully reduced.
// grid = (N/4096, batch); in place on a.out.
__global__ __launch_bounds__(NTT_THREADS) void ntt_fwd_row(NttArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 lds[ROW_LDS_ELEMS];
    const int t = threadIdx.x;
    const PolySel ps = select_poly(a, blockIdx.y);
    const Mod md = a.mods[ps.mod];
    const u64 q = md.q, q2 = 2 * md.q;
    const int s1 = a.n_power - 8;
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    u64* __restrict__ p = a.out + ps.out_off + (u64) blockIdx.x * 4096;

    const int row = t >> 4, i0 = t & 15;
    const u32 crow = blockIdx.x * 16 + row; // global row index
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = p[row * 256 + i0 + 16 * k];
    ct_radix<4>(x, tw, (1u << s1) + crow, q, q2);
#pragma unroll
    for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = x[k];
    __syncthreads();
    {
        const ulonglong2* l2 = reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0)]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ulonglong2 v = l2[k];
            x[2 * k] = v.x;
            x[2 * k + 1] = v.y;
        }
    }
    ct_radix<4>(x, tw, (((1u << s1) + crow) << 4) + (u32) i0, q, q2);
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = csub(csub(x[k], q2), q);
    __syncthreads();
    {
        ulonglong2* l2 = reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0)]);
#pragma unroll
        for (int k = 0; k < 8; k++) l2[k] = make_ulonglong2(x[2 * k], x[2 * k + 1]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) p[row * 256 + i0 + 16 * k] = lds[row_phys(row * 256 + i0 + 16 * k)];
}

// ------------------------------------------------------------------ inverse
// Row pass first (GS stages with t = 1..128), reads a.in, writes a.out.
__global__ __launch_bounds__(NTT_THREADS) void ntt_inv_row(NttArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 lds[ROW_LDS_ELEMS];
    const int t = threadIdx.x;
    const PolySel ps = select_poly(a, blockIdx.y);
    const Mod md = a.mods[ps.mod];
    const u64 q = md.q, q2 = 2 * md.q;
    const int s1 = a.n_power - 8;
    const ulonglong2* __restrict__ tw = a.itw + ((u64) ps.mod << a.n_power);
    const u64* __restrict__ src = a.in + ps.in_off + (u64) blockIdx.x * 4096;
    u64* __restrict__ dst = a.out + ps.out_off + (u64) blockIdx.x * 4096;

    const int row = t >> 4, i0 = t & 15;
    const u32 crow = blockIdx.x * 16 + row;
#pragma unroll
    for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = src[row * 256 + i0 + 16 * k];
    __syncthreads();
    u64 x[16];
    {
        const ulonglong2* l2 = reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0)]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ulonglong2 v = l2[k];
            x[2 * k] = v.x;
            x[2 * k + 1] = v.y;
        }
    }
    gs_radix<4>(x, tw, (((1u << s1) + crow) << 4) + (u32) i0, q, q2);
    __syncthreads();
    {
        ulonglong2* l2 = reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0)]);
#pragma unroll
        for (int k = 0; k < 8; k++) l2[k] = make_ulonglong2(x[2 * k], x[2 * k + 1]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = lds[row_phys(row * 256 + i0 + 16 * k)];
    gs_radix<4>(x, tw, (1u << s1) + crow, q, q2);
#pragma unroll
    for (int k = 0; k < 16; k++) dst[row * 256 + i0 + 16 * k] = x[k];
}

// Column pass last: GS stages S1-1..0, N^-1 folded into the final stage.
// In place on a.out.  grid = (256/CT, batch).
template <int S1>
__global__ __launch_bounds__(NTT_THREADS) void ntt_inv_col(NttArgs a)
{
    constexpr int R = 1 << S1;
    constexpr int CT = 4096 / R;
    constexpr int NSA = S1 - 4;
    constexpr int RA = 1 << NSA;
    constexpr int G = 16 / RA;
    __shared__ u64 lds[(NSA > 0) ? COL_LDS_ELEMS : 1];

    const int t = threadIdx.x;
    const PolySel ps = select_poly(a, blockIdx.y);
    const Mod md = a.mods[ps.mod];
    const u64 q = md.q, q2 = 2 * md.q;
    const ulonglong2* __restrict__ tw = a.itw + ((u64) ps.mod << a.n_power);
    const ulonglong2 ninv = a.ninv[ps.mod], w1ninv = a.w1ninv[ps.mod];
    u64* __restrict__ p = a.out + ps.out_off + blockIdx.x * CT;

    const int col = t % CT, r1 = t / CT;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = p[(u64) (16 * r1 + k) * 256 + col];
    if constexpr (NSA > 0) {
        gs_radix<4>(x, tw, (u32) (RA + r1), q, q2);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[col_phys((16 * r1 + k) * CT + col)] = x[k];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int L = t + NTT_THREADS * g;
            const int c = L % CT, rb = L / CT;
            u64 y[RA];
#pragma unroll
            for (int k = 0; k < RA; k++) y[k] = lds[col_phys((rb + 16 * k) * CT + c)];
            gs_radix_last<NSA>(y, tw, 1u, ninv, w1ninv, q, q2);
#pragma unroll
            for (int k = 0; k < RA; k++) p[(u64) (rb + 16 * k) * 256 + c] = y[k];
        }
    } else {
        gs_radix_last<4>(x, tw, 1u, ninv, w1ninv, q, q2);
#pragma unroll
        for (int k = 0; k < 16; k++) p[(u64) k * 256 + col] = x[k];
    }
}

// ------------------------------------------------------------------ launch
template <int S1>
static void launch_fwd(const NttArgs& a, int batch, hipStream_t st)
{
    constexpr int CT = 4096 >> S1;
    if (a.decomp_mods)
        hipLaunchKernelGGL((ntt_fwd_col<S1, true>), dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((ntt_fwd_col<S1, false>), dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, a);
    NttArgs b = a;
    b.in = a.out;
    b.in_item_stride = a.out_item_stride;
    b.decomp_mods = 0;
    hipLaunchKernelGGL(ntt_fwd_row, dim3((1u << a.n_power) / 4096, batch), dim3(NTT_THREADS), 0, st, b);
}

template <int S1>
static void launch_inv(const NttArgs& a, int batch, hipStream_t st)
{
    constexpr int CT = 4096 >> S1;
    hipLaunchKernelGGL(ntt_inv_row, dim3((1u << a.n_power) / 4096, batch), dim3(NTT_THREADS), 0, st, a);
    NttArgs b = a;
    b.in = a.out;
    b.in_item_stride = a.out_item_stride;
    hipLaunchKernelGGL(ntt_inv_col<S1>, dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, b);
}

hipError_t ntt_launch(const NttArgs& a, int batch, bool inverse, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (a.n_power < 12 || a.n_power > 16) return hipErrorInvalidValue;
    if (a.decomp_mods && (inverse || a.poly_order || !a.polys_per_item)) return hipErrorInvalidValue;
    if (batch > 65535) {
        // gridDim.y limit: split (poly_order / mod_order semantics need the
        // absolute polynomial index, so only plain batches are split)
        // split on item boundaries (or modulus-cycle boundaries for a flat batch)
        const int unit = a.polys_per_item ? a.polys_per_item : a.mod_count;
        if ((!a.polys_per_item && a.poly_order) || (batch % unit) || unit > 65535) return hipErrorInvalidValue;
        int done = 0;
        while (done < batch) {
            int chunk = batch - done;
            int maxc = (65535 / unit) * unit;
            if (chunk > maxc) chunk = maxc;
            NttArgs c = a;
            if (a.polys_per_item) {
                c.in = a.in + (u64) (done / unit) * a.in_item_stride;
                c.out = a.out + (u64) (done / unit) * a.out_item_stride;
            } else {
                c.in = a.in + ((u64) done << a.n_power);
                c.out = a.out + ((u64) done << a.n_power);
            }
            hipError_t e = ntt_launch(c, chunk, inverse, st);
            if (e != hipSuccess) return e;
            done += chunk;
        }
        return hipSuccess;
    }
    switch (a.n_power - 8) {
#define CASE(S)                                   \
    case S:                                       \
        if (inverse) launch_inv<S>(a, batch, st); \
        else launch_fwd<S>(a, batch, st);         \
        break;
        CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    return hipGetLastError();
}

} // namespace hegpu
