/*
 * hegpu_bench.h -- measurement seam of libhegpu.so.  NOT part of the drop-in boundary (include/hegpu.h): no reference
 * interface corresponds to it, and a call with only some of the phases leaves the ciphertext in a partially
 * key-switched state.  Used by bench.py (per-launch-group timings with HIP events) and tools/.
 */
#ifndef HEGPU_BENCH_H
#define HEGPU_BENCH_H
#include "hegpu.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Runs only the launches
 * of hegpu_ckks_relinearize_inplace (method I) selected by `phases` -- 1 INTT of c2 (:919), 2 decomposing
 * column pass and 4 row pass + inner product (:932-988), 8 INTT of the P limbs (:996), 16 mod-down NTT
 * (:1003-1015) -- on whatever the buffers hold; results are meaningful only with all five (31). */
int hegpu_probe_ckks_relinearize(hegpu_context* ctx, uint64_t* ct, uint64_t ct_stride, const uint64_t* relin_key,
                                 int depth, int batch, void* ws, size_t ws_bytes, unsigned phases,
                                 hegpu_stream stream);
#ifdef __cplusplus
}
#endif
#endif /* HEGPU_BENCH_H */
