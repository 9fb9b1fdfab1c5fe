// ops.hpp -- batched operator sequences on a Context (internal C++).
#pragma once
#include "context.hpp"

namespace hegpu {

enum { OP_CKKS_RELIN = 1, OP_CKKS_RESCALE = 2, OP_CKKS_GALOIS = 3, OP_BFV_MULTIPLY = 4, OP_BFV_RELIN = 5,
       OP_BFV_GALOIS = 6 };

size_t ops_workspace_elems(const Context& c, int op, int depth, int batch);

hipError_t op_ckks_multiply(const Context& c, const u64* ct1, u64 s1, const u64* ct2, u64 s2, u64* out, u64 so,
                            int depth, int batch, hipStream_t st);
hipError_t op_ckks_relinearize(const Context& c, u64* ct, u64 cs, const u64* key, int depth, int batch, u64* ws,
                               hipStream_t st);
hipError_t op_ckks_rescale(const Context& c, u64* ct, u64 cs, int depth, int batch, u64* ws, hipStream_t st);
hipError_t op_ckks_apply_galois(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                int galois_elt, int depth, int batch, u64* ws, hipStream_t st);
hipError_t op_bfv_multiply(const Context& c, const u64* ct1, u64 s1, const u64* ct2, u64 s2, u64* out, u64 so,
                           int batch, u64* ws, hipStream_t st);
hipError_t op_bfv_relinearize(const Context& c, u64* ct, u64 cs, const u64* key, int batch, u64* ws,
                              hipStream_t st);
hipError_t op_bfv_apply_galois(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                               int galois_elt, int batch, u64* ws, hipStream_t st);

// key-switching method II (P_size > 1)
hipError_t op_ckks_relinearize_II(const Context& c, u64* ct, u64 cs, const u64* key, int depth, int batch, u64* ws,
                                  hipStream_t st);
hipError_t op_ckks_apply_galois_II(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                   int galois_elt, int depth, int batch, u64* ws, hipStream_t st);
hipError_t op_bfv_relinearize_II(const Context& c, u64* ct, u64 cs, const u64* key, int batch, u64* ws,
                                 hipStream_t st);
hipError_t op_bfv_apply_galois_II(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                  int galois_elt, int batch, u64* ws, hipStream_t st);

} // namespace hegpu
