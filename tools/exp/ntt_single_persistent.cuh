// ntt_single_persistent.cuh -- round-5 experiments on the N = 2^14 single pass, for tools/exp/ntt_exp.hip ONLY (included
// after heongpu_amd/csrc/ntt.hip; not part of the product).  All three are bit-exact (they ran the product's parity tests
// while they were wired into ntt_launch, commit "ntt_fwd_single_pf ..."), none is faster than one workgroup per limb by
// more than 5 %: profiles/r5b_ntt14/README.md has the table.
//   ntt_fwd_single_pf   512 threads, each playing two of the 1024, next limb's 32 coefficients prefetched (217 registers)
//   ntt_fwd_single_ps   1024 threads, persistent workgroup, no prefetch
//   ntt_fwd_single_ps1  1024 threads, persistent, next limb's 16 coefficients prefetched (128 registers + 92 B scratch)
#pragma once
namespace hegpu {
// select_poly with its two order tables read through the scalar cache
__device__ __forceinline__ PolySel select_poly_const(const NttArgs& a, int poly)
{
    PolySel s;
    if (a.group_span) {
        // modulus-major walk: grid index = k * span + r  ->  r-th polynomial
        // (in item-major order) among those with modulus slot k
        const int k = udiv16(poly, a.mg_group_span), r = poly - k * a.group_span;
        if (a.polys_per_item) {
            const int per_item = udiv16(a.polys_per_item, a.mg_mod_count); // digits per item
            const int it = udiv16(r, a.mg_per_item), d = r - it * per_item;
            poly = it * a.polys_per_item + d * a.mod_count + k;
        } else {
            poly = r * a.mod_count + k;
        }
    }
    int item = 0, j = poly;
    if (a.polys_per_item) {
        item = udiv16(poly, a.mg_polys_per_item);
        j = poly - item * a.polys_per_item;
    }
    int k = j - udiv16(j, a.mg_mod_count) * a.mod_count;
    if (a.mod_order) k = ld_const_i32(a.mod_order + k);
    s.mod = a.mod_offset + k;
    u64 slot = a.poly_order ? (u64) ld_const_i32(a.poly_order + j) : (u64) j;
    s.digit = a.decomp_mods ? udiv16(j, a.mg_decomp_mods) : -1;
    u64 in_slot = a.decomp_mods ? (u64) s.digit * (a.decomp_in_mul ? a.decomp_in_mul : 1) + a.decomp_in_add : slot;
    s.in_off = (u64) item * a.in_item_stride + (in_slot << a.n_power);
    s.out_off = (u64) item * a.out_item_stride + (slot << a.n_power);
    s.item = item;
    s.j = j;
    return s;
}

#define PF_GRID 256 // one persistent workgroup per compute unit (its 128 KiB of LDS leave room for no second one)

// ------------------------------------------------------------------ single pass, the next limb on its way (round 5)
// ntt_fwd_single at N = 2^14 holds ONE workgroup of 1024 threads per CU (the limb fills 128 of the 160 KiB of LDS), so the
// load of a limb, its 14 stages and its store run one after the other: 0.553 ms of movement + 0.29 ms of arithmetic per
// GiB of limbs, not overlapped (profiles/r2c_experiments/README.md 3; a register prefetch at 1024 threads = 128 registers
// per thread spilled).  Here the SAME three phases run on HALF the threads, each thread playing two of the 1024 (t and
// t + 512, phase by phase, so the barriers stay where they were): 2 waves per SIMD = 256 registers per thread, enough to
// hold the 32 coefficients of the NEXT limb of a persistent workgroup (requested before the current limb's first
// butterfly, consumed an iteration later) next to the working set.  Loads and stores of neighbouring limbs are in flight
// while a limb is transformed; instruction-level parallelism of the two threads played replaces the second pair of waves.
template <int S1>
__device__ __forceinline__ void single_load16(const u64* __restrict__ limb_src, int t, u64 (&dv)[16])
{
    constexpr int CT = 4096 >> S1, NSA = S1 - 4, RA = 1 << NSA, G = 16 / RA;
    static_assert(NSA > 0, "N >= 2^13");
    const int g = t >> 8, tt = t & 255;
    const u64* __restrict__ src = limb_src + g * CT;
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        const int L = tt + 256 * gi;
        const int c = L % CT, rb = L / CT;
#pragma unroll
        for (int k = 0; k < RA; k++) dv[gi * RA + k] = gld(&src[(u64) (rb + 16 * k) * 256 + c]);
    }
}

// column stages 0 .. S1-5 of (virtual) thread t, its 16 source coefficients in load order -> the limb in LDS
template <int S1, bool FP, bool LAZY>
__device__ __forceinline__ void single_col_a(int t, const u64 (&dv)[16], u64* limb, ConstTw tw,
                                             const QC& qc, const FC& fc)
{
    constexpr int CT = 4096 >> S1, NSA = S1 - 4, RA = 1 << NSA, G = 16 / RA;
    const int g = t >> 8, tt = t & 255;
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        const int L = tt + 256 * gi;
        const int c = L % CT, rb = L / CT;
        if constexpr (FP) {
            double y[RA];
#pragma unroll
            for (int k = 0; k < RA; k++) y[k] = fp_from_u64(dv[gi * RA + k]);
            fp_ct_radix<NSA>(y, tw, 1u, fc, FpColSched<S1>::a_before, false);
#pragma unroll
            for (int k = 0; k < RA; k++) limb[single_pos<S1>(rb + 16 * k, g * CT + c)] = as_bits(y[k]);
        } else {
            u64 y[RA];
#pragma unroll
            for (int k = 0; k < RA; k++) y[k] = dv[gi * RA + k];
            ct_radix<NSA, LAZY>(y, tw, 1u, qc);
#pragma unroll
            for (int k = 0; k < RA; k++) limb[single_pos<S1>(rb + 16 * k, g * CT + c)] = y[k];
        }
    }
}

// column stages S1-4 .. S1-1 of (virtual) thread t, in place in LDS
template <int S1, bool FP, bool LAZY>
__device__ __forceinline__ void single_col_b(int t, u64* limb, ConstTw tw, const QC& qc, const FC& fc)
{
    constexpr int CT = 4096 >> S1, RA = 1 << (S1 - 4);
    static_assert(CT >= 64, "r1 is uniform over a wavefront");
    const int g = t >> 8, tt = t & 255;
    // r1 is the same for the 64 lanes of a wavefront: say so (the caller hides the thread index from the optimiser, and
    // with it the fact), so that the twiddles of this round come through the scalar cache -- a vector load here would put
    // an s_waitcnt vmcnt(0) in front of the butterflies, which also waits for the next limb's prefetch
    const int col = tt % CT, r1 = __builtin_amdgcn_readfirstlane(tt / CT);
    if constexpr (FP) {
        double x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = as_f64(limb[single_pos<S1>(16 * r1 + k, g * CT + col)]);
        fp_ct_radix<4>(x, tw, (u32) (RA + r1), fc, FpColSched<S1>::b_before, FpColSched<S1>::b_at_end);
#pragma unroll
        for (int k = 0; k < 16; k++) limb[single_pos<S1>(16 * r1 + k, g * CT + col)] = as_bits(x[k]);
    } else {
        u64 x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = limb[single_pos<S1>(16 * r1 + k, g * CT + col)];
        ct_radix<4, LAZY>(x, tw, (u32) (RA + r1), qc);
#pragma unroll
        for (int k = 0; k < 16; k++) limb[single_pos<S1>(16 * r1 + k, g * CT + col)] = x[k];
    }
}

// row stages S1 .. S1+7 of (virtual) thread t on row tile t / 256 and the store (as the second half of fwd_single_body)
template <int S1, bool FP, bool LAZY>
__device__ __forceinline__ void single_rows(int t, u64* limb, const NttArgs& a, const PolySel& ps, const Mod& md,
                                            const ulonglong2* __restrict__ tw, const QC& qc, const FC& fc)
{
    const int g = t >> 8, tt = t & 255;
    u64* lds = limb + g * 4096;
    const int row = tt >> 4, i0 = tt & 15;
    const u32 crow = g * 16 + row;
    const ulonglong2* __restrict__ tb = a.twB + ((u64) ps.mod * (15u << (a.n_power - 4))) + ((u64) crow * 15 * 16 + i0);
    u64 r[16];
    if constexpr (FP) {
        double x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = as_f64(lds[row_phys(row * 256 + i0 + 16 * k)]);
        fp_ct_radix<4>(x, tw, (1u << S1) + crow, fc, 0u, true);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = as_bits(x[k]);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
            x[2 * k] = as_f64(v.x);
            x[2 * k + 1] = as_f64(v.y);
        }
        fp_ct_radix16_tb(x, tb, fc);
#pragma unroll
        for (int k = 0; k < 16; k++) r[k] = fp_to_u64(x[k]);
    } else {
        u64 x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = lds[row_phys(row * 256 + i0 + 16 * k)];
        ct_radix<4, LAZY>(x, tw, (1u << S1) + crow, qc);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = x[k];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
            x[2 * k] = v.x;
            x[2 * k + 1] = v.y;
        }
        ct_radix16_tb<LAZY>(x, tb, qc);
        if (LAZY) {
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = reduce64(x[k], md);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = csub(csub(csub(x[k], qc.q4), 2 * qc.q), qc.q);
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++)
        *reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]) = make_ulonglong2(r[2 * k], r[2 * k + 1]);
    wave_lds_fence();
    row_store_all(a, ps, md, a.out + ps.out_off, (u64) g * 4096 + row * 256 + i0, lds, row, i0);
}

template <int S1, bool FP, bool LAZY>
__device__ __forceinline__ void single_pf_limb(const NttArgs& a, const PolySel& ps, const Mod& md, u64* limb,
                                               const u64 (&d0)[16], const u64 (&d1)[16])
{
    constexpr int HALF = 8 << S1; // threads of the workgroup: each plays t and t + HALF
    const QC qc = make_qc(md.q);
    const FC fc = make_fc(md.q);
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    const ConstTw ctw = const_tw(tw);
    // Two things keep the registers for the NEXT limb free (without them the compiler spills the prefetched values the
    // moment they arrive -- 724 bytes of scratch per lane and no overlap at all): a scheduling fence between the two
    // threads played (interleaving their phases doubles the live values), and a thread index the optimiser cannot see
    // through in every phase -- otherwise all ~100 LDS / global addresses of the six phase bodies, which depend on
    // nothing but the thread index, are hoisted out of the limb loop and stay live across it.
#define PF_OPAQUE_T(v) int v = threadIdx.x; asm volatile("" : "+v"(v))
    {
        PF_OPAQUE_T(t);
        single_col_a<S1, FP, LAZY>(t, d0, limb, ctw, qc, fc);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        PF_OPAQUE_T(t);
        single_col_a<S1, FP, LAZY>(t + HALF, d1, limb, ctw, qc, fc);
    }
    __syncthreads();
    {
        PF_OPAQUE_T(t);
        single_col_b<S1, FP, LAZY>(t, limb, ctw, qc, fc);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        PF_OPAQUE_T(t);
        single_col_b<S1, FP, LAZY>(t + HALF, limb, ctw, qc, fc);
    }
    __syncthreads();
    {
        PF_OPAQUE_T(t);
        single_rows<S1, FP, LAZY>(t, limb, a, ps, md, tw, qc, fc);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        PF_OPAQUE_T(t);
        single_rows<S1, FP, LAZY>(t + HALF, limb, a, ps, md, tw, qc, fc);
    }
#undef PF_OPAQUE_T
}

// grid = min(batch, resident workgroups) persistent workgroups of N / 32 threads, N * 8 bytes of dynamic LDS; workgroup b
// transforms polynomials b, b + grid, b + 2 grid, ...
template <int S1>
__global__ __launch_bounds__(8 << S1) void ntt_fwd_single_pf(NttArgs a, int batch)
{
    extern __shared__ __attribute__((aligned(16))) u64 limb[];
    constexpr int HALF = 8 << S1;
    const int t = threadIdx.x;
    int poly = blockIdx.x;
    PolySel ps = select_poly_const(a, poly);
    u64 d0[16], d1[16];
    single_load16<S1>(a.in + ps.in_off, t, d0);
    single_load16<S1>(a.in + ps.in_off, t + HALF, d1);
    // The first limb is waited for HERE, outside the loop: the vector-memory counter is in order, and with these loads
    // still pending at the loop header the compiler's wait insertion makes every use of d0 / d1 inside the loop wait for
    // the loads issued after them -- the next limb's prefetch -- as well (s_waitcnt vmcnt(29) ... vmcnt(0) across the
    // first phase: the prefetch overlapped a sixth of the arithmetic).  0x0F70 = vmcnt(0), the other counters untouched.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (;;) {
        const int next = poly + gridDim.x;
        const bool more = next < batch;
        PolySel pn = ps;
        u64 n0[16], n1[16];
        if (more) {
            pn = select_poly_const(a, next);
            int tl = threadIdx.x;
            asm volatile("" : "+v"(tl)); // (as in single_pf_limb: no addresses carried around the loop)
            single_load16<S1>(a.in + pn.in_off, tl, n0);
            single_load16<S1>(a.in + pn.in_off, tl + HALF, n1);
        }
        const Mod md = ld_const_mod(a.mods + ps.mod); // (a vector load here would wait for the prefetch just issued)
        if (md.fp) single_pf_limb<S1, true, false>(a, ps, md, limb, d0, d1);
        else if (fwd_stages_lazy(md, a.lazy_q_max)) single_pf_limb<S1, false, true>(a, ps, md, limb, d0, d1);
        else single_pf_limb<S1, false, false>(a, ps, md, limb, d0, d1);
        if (!more) break;
        __syncthreads(); // the row tiles of this limb are read until its stores are issued; the next limb overwrites them
#pragma unroll
        for (int k = 0; k < 16; k++) {
            d0[k] = n0[k];
            d1[k] = n1[k];
        }
        ps = pn;
        poly = next;
    }
}

// Experiment (single_pass == 3): the plain single pass, N / 16 threads, as a persistent workgroup -- no prefetch, only the
// stores of limb i and the loads of limb i + 1 in flight together and no workgroup turnover between limbs.
template <int S1, bool FP, bool LAZY>
__device__ __forceinline__ void single_ps_limb(const NttArgs& a, const PolySel& ps, const Mod& md, u64* limb)
{
    const QC qc = make_qc(md.q);
    const FC fc = make_fc(md.q);
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    const ConstTw ctw = const_tw(tw);
#define PF_OPAQUE_T(v) int v = threadIdx.x; asm volatile("" : "+v"(v))
    {
        PF_OPAQUE_T(t);
        u64 d[16];
        single_load16<S1>(a.in + ps.in_off, t, d);
        single_col_a<S1, FP, LAZY>(t, d, limb, ctw, qc, fc);
    }
    __syncthreads();
    {
        PF_OPAQUE_T(t);
        single_col_b<S1, FP, LAZY>(t, limb, ctw, qc, fc);
    }
    __syncthreads();
    {
        PF_OPAQUE_T(t);
        single_rows<S1, FP, LAZY>(t, limb, a, ps, md, tw, qc, fc);
    }
#undef PF_OPAQUE_T
}

// ... and with the next limb's 16 coefficients per thread requested ahead (single_pass == 4): 89 + 32 registers
template <int S1, bool FP, bool LAZY>
__device__ __forceinline__ void single_ps1_limb(const NttArgs& a, const PolySel& ps, const Mod& md, u64* limb, const u64 (&d)[16])
{
    const QC qc = make_qc(md.q);
    const FC fc = make_fc(md.q);
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    const ConstTw ctw = const_tw(tw);
#define PF_OPAQUE_T(v) int v = threadIdx.x; asm volatile("" : "+v"(v))
    {
        PF_OPAQUE_T(t);
        single_col_a<S1, FP, LAZY>(t, d, limb, ctw, qc, fc);
    }
    __syncthreads();
    {
        PF_OPAQUE_T(t);
        single_col_b<S1, FP, LAZY>(t, limb, ctw, qc, fc);
    }
    __syncthreads();
    {
        PF_OPAQUE_T(t);
        single_rows<S1, FP, LAZY>(t, limb, a, ps, md, tw, qc, fc);
    }
#undef PF_OPAQUE_T
}

template <int S1>
__global__ __launch_bounds__(16 << S1) void ntt_fwd_single_ps1(NttArgs a, int batch)
{
    extern __shared__ __attribute__((aligned(16))) u64 limb[];
    int poly = blockIdx.x;
    PolySel ps = select_poly_const(a, poly);
    u64 d[16];
    single_load16<S1>(a.in + ps.in_off, threadIdx.x, d);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (;;) {
        const int next = poly + gridDim.x;
        const bool more = next < batch;
        PolySel pn = ps;
        u64 nx[16];
        if (more) {
            pn = select_poly_const(a, next);
            int tl = threadIdx.x;
            asm volatile("" : "+v"(tl));
            single_load16<S1>(a.in + pn.in_off, tl, nx);
        }
        const Mod md = ld_const_mod(a.mods + ps.mod);
        if (md.fp) single_ps1_limb<S1, true, false>(a, ps, md, limb, d);
        else if (fwd_stages_lazy(md, a.lazy_q_max)) single_ps1_limb<S1, false, true>(a, ps, md, limb, d);
        else single_ps1_limb<S1, false, false>(a, ps, md, limb, d);
        if (!more) break;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = nx[k];
        ps = pn;
        poly = next;
    }
}

template <int S1>
__global__ __launch_bounds__(16 << S1) void ntt_fwd_single_ps(NttArgs a, int batch)
{
    extern __shared__ __attribute__((aligned(16))) u64 limb[];
    for (int poly = blockIdx.x; poly < batch; poly += gridDim.x) {
        const PolySel ps = select_poly_const(a, poly);
        const Mod md = ld_const_mod(a.mods + ps.mod);
        if (md.fp) single_ps_limb<S1, true, false>(a, ps, md, limb);
        else if (fwd_stages_lazy(md, a.lazy_q_max)) single_ps_limb<S1, false, true>(a, ps, md, limb);
        else single_ps_limb<S1, false, false>(a, ps, md, limb);
        __syncthreads();
    }
}


template <int S1>
static hipError_t launch_single_persistent(const NttArgs& a, int batch, int form, hipStream_t st)
{
    NttArgs g = a;
    g.group_span = 0;
    if (a.mod_count > 1 && batch % a.mod_count == 0) g.group_span = batch / a.mod_count;
    fill_magics(g);
    const dim3 grid(batch < PF_GRID ? batch : PF_GRID);
    const size_t lds = (size_t) 8 << (S1 + 8);
    if (form == 2) {
        (void) hipFuncSetAttribute((const void*) ntt_fwd_single_pf<S1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        hipLaunchKernelGGL((ntt_fwd_single_pf<S1>), grid, dim3(8 << S1), lds, st, g, batch);
    } else if (form == 3) {
        (void) hipFuncSetAttribute((const void*) ntt_fwd_single_ps<S1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        hipLaunchKernelGGL((ntt_fwd_single_ps<S1>), grid, dim3(16 << S1), lds, st, g, batch);
    } else {
        (void) hipFuncSetAttribute((const void*) ntt_fwd_single_ps1<S1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        hipLaunchKernelGGL((ntt_fwd_single_ps1<S1>), grid, dim3(16 << S1), lds, st, g, batch);
    }
    return hipGetLastError();
}
} // namespace hegpu
