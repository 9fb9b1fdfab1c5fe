// consumer_compat/cuda_runtime.h -- NOT part of the backend.  Some of the reference's programs
// (benchmark/benchmark_tfhe.cpp:3) include <cuda_runtime.h> themselves; put this directory on the
// include path only when compiling such a source unchanged.  It maps the cuda* names those
// programs call to the HIP runtime.
#pragma once
#include "../cuda_names.hpp"
