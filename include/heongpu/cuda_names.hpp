// cuda_names.hpp -- NOT part of the backend: the reference's consumers (benchmark/*.cpp,
// test/*.cpp, example/*.cpp) call a handful of CUDA runtime functions themselves (events,
// streams, synchronisation).  Compiling those sources unchanged against this HIP backend needs
// the same names to exist; define HEONGPU_CUDA_NAMES before including <heongpu/heongpu.hpp>.
#pragma once
#include <hip/hip_runtime.h>

typedef hipEvent_t cudaEvent_t;
typedef hipStream_t cudaStream_t;
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
#define cudaEventCreate hipEventCreate
#define cudaEventRecord hipEventRecord
#define cudaEventSynchronize hipEventSynchronize
#define cudaEventElapsedTime hipEventElapsedTime
#define cudaEventDestroy hipEventDestroy
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaStreamCreate hipStreamCreate
#define cudaStreamSynchronize hipStreamSynchronize
#define cudaStreamDestroy hipStreamDestroy
#define cudaSetDevice hipSetDevice
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
