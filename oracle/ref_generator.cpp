/*
 * oracle/ref_generator.cpp — TEST INFRASTRUCTURE ONLY: runs the REFERENCE's own input generator on the CPU.
 *
 * The reference's InitRandom kernels (GPUSortingCUDA/UtilityKernels.cuh:53-117) are compiled FROM WHERE THEY LIE
 * under /root/reference (nothing is copied) against the CUDA stand-in headers in oracle/shim/, and executed one
 * emulated thread at a time with the launch shape the reference uses, <<<256, 256>>>
 * (GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:100,215).  The kernels have no inter-thread communication, so the
 * sequential execution is exact.  Output: oracle/_ref/libref_generator.so (git-ignored; built only where
 * /root/reference exists).  It pins oracle/gs_oracle.cpp's restatement (tests/test_oracle.py) and produces the
 * golden vectors tests/golden/ref_init_random.npz (tests/golden/make_ref_golden.py), against which the oracle,
 * and through them the HIP generator, are checked on boxes that have no /root/reference.
 */
#include "cuda_runtime.h"
thread_local gs_shim_dim3 threadIdx, blockIdx, blockDim, gridDim;
#include "UtilityKernels.cuh"  // -I/root/reference/GPUSortingCUDA

extern "C" {

/* keys[size] (and payload[size] if not null) exactly as the reference's InitRandom<<<256,256>>> writes them */
void ref_init_random(uint32_t* keys, uint32_t* payload_or_null, uint32_t andCount, uint32_t seed, uint32_t size) {
    gridDim = {256, 1, 1};
    blockDim = {256, 1, 1};
    for (unsigned b = 0; b < 256; ++b)
        for (unsigned t = 0; t < 256; ++t) {
            blockIdx = {b, 0, 0};
            threadIdx = {t, 0, 0};
            if (payload_or_null) InitRandom(keys, payload_or_null, andCount, seed, size);
            else InitRandom(keys, andCount, seed, size);
        }
}

const char* ref_generator_source() { return "GPUSortingCUDA/UtilityKernels.cuh:53-117 executed on the CPU (oracle/shim)"; }
}
