/*
 * oracle/shim/cuda_runtime.h — TEST INFRASTRUCTURE ONLY.
 *
 * A host-side stand-in for the CUDA headers, just enough for g++ to PARSE the reference's
 * GPUSortingCUDA/UtilityKernels.cuh (and the Utils.cuh it includes) where they lie under
 * /root/reference, so that oracle/ref_generator.cpp can EXECUTE the reference's own InitRandom
 * kernels on the CPU (one emulated thread at a time; those kernels have no inter-thread
 * communication).  Everything else in those headers only has to compile: the warp/shared-memory
 * intrinsics below are declarations with placeholder bodies and are never called.
 */
#pragma once
#ifdef GS_SIMT_EMU
/* the kernels that DO communicate (OneSweep.cu) run under the SIMT emulator instead */
#include "simt_emu.h"
#else
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define __global__ static inline  /* only the kernels the driver calls are emitted: the others hold PTX asm */
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __restrict__

struct gs_shim_dim3 { unsigned x, y, z; };
typedef gs_shim_dim3 dim3;
/* the emulated thread's coordinates: set by the driver before every call of a kernel function */
extern thread_local gs_shim_dim3 threadIdx, blockIdx, blockDim, gridDim;

/* ---- never executed: present so the other kernels of the header type-check ---- */
inline void __syncthreads() {}
inline void __syncwarp(unsigned = 0xffffffffu) {}
inline unsigned __activemask() { return 0xffffffffu; }
template <class T> inline T __shfl_sync(unsigned, T v, int, int = 32) { return v; }
template <class T> inline T __shfl_up_sync(unsigned, T v, unsigned, int = 32) { return v; }
template <class T> inline T __shfl_down_sync(unsigned, T v, unsigned, int = 32) { return v; }
template <class T> inline T __shfl_xor_sync(unsigned, T v, int, int = 32) { return v; }
inline unsigned __ballot_sync(unsigned, int p) { return p ? 1u : 0u; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline void __threadfence() {}
template <class T, class U> inline T atomicAdd(T* p, U v) { T o = *p; *p = o + (T)v; return o; }
template <class T, class U, class W> inline T atomicCAS(T* p, U c, W v) { T o = *p; if (o == (T)c) *p = (T)v; return o; }
template <class T, class U> inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
#endif /* GS_SIMT_EMU */
