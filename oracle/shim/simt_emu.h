/*
 * oracle/shim/simt_emu.h — TEST INFRASTRUCTURE ONLY: a small SIMT emulator, enough to EXECUTE the reference's
 * OneSweep kernels (GPUSortingCUDA/Sort/OneSweep.cu) on the CPU from where they lie under /root/reference.
 *
 * Model: one CUDA thread = one fiber (ucontext) with its own stack; all fibers of a thread block run on ONE OS
 * thread, so shared/global "atomics" are plain operations and execution is deterministic.  A fiber runs until it
 * reaches a warp collective (__ballot_sync, __shfl*_sync, __activemask) or __syncthreads, where it parks; when no
 * fiber of the block can run, the scheduler completes every collective whose participants have all arrived
 * (explicit mask: the lanes of the mask; __activemask: the lanes of the warp parked there — exactly the lanes
 * that "are converged" at that point) and releases the block barrier once every live fiber has reached it.
 * Thread blocks run one after the other in blockIdx order, which is the order in which the reference's kernels
 * take their partition tickets, so the decoupled look-back never has to wait.  32-lane warps.
 */
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <functional>
#include <vector>

#define __global__ static inline  /* kernels are ordinary functions; only the ones the driver calls are emitted */
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static         /* one block at a time: a function-local static IS the block's shared array */
#define __restrict__

struct gs_dim3 { unsigned x, y, z; };
typedef gs_dim3 dim3;
struct uint4 { uint32_t x, y, z, w; };

namespace gs_emu {
enum Op { OP_NONE, OP_BALLOT, OP_SHFL, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_ACTIVEMASK };
enum State { RUNNABLE, WAIT_WARP, WAIT_BLOCK, DONE };
struct Fiber {
    ucontext_t ctx;
    void* stack;
    gs_dim3 tid;
    unsigned lane, warp;
    State state;
    Op op;
    uint32_t mask;
    uint64_t val, result;
    int arg;
};
extern Fiber* cur;
extern gs_dim3 bIdx, bDim, gDim;
uint64_t collective(Op op, uint32_t mask, uint64_t val, int arg);  // parks the calling fiber until resolved
void barrier();
void launch(unsigned grid, unsigned block, const std::function<void()>& body);
}  // namespace gs_emu

#define threadIdx (gs_emu::cur->tid)
#define blockIdx (gs_emu::bIdx)
#define blockDim (gs_emu::bDim)
#define gridDim (gs_emu::gDim)

inline void __syncthreads() { gs_emu::barrier(); }
inline void __threadfence() {}
inline void __syncwarp(unsigned = 0xffffffffu) {}  /* never reached by the kernels that are run */
inline unsigned __activemask() { return (unsigned)gs_emu::collective(gs_emu::OP_ACTIVEMASK, 0, 0, 0); }
inline unsigned __ballot_sync(unsigned mask, int pred) {
    return (unsigned)gs_emu::collective(gs_emu::OP_BALLOT, mask, pred ? 1u : 0u, 0);
}
template <class T> inline uint64_t gs_emu_pack(T v) { uint64_t u = 0; memcpy(&u, &v, sizeof(T)); return u; }
template <class T> inline T gs_emu_unpack(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }
template <class T> inline T __shfl_sync(unsigned mask, T v, int src, int = 32) {
    return gs_emu_unpack<T>(gs_emu::collective(gs_emu::OP_SHFL, mask, gs_emu_pack(v), src));
}
template <class T> inline T __shfl_up_sync(unsigned mask, T v, unsigned d, int = 32) {
    return gs_emu_unpack<T>(gs_emu::collective(gs_emu::OP_SHFL_UP, mask, gs_emu_pack(v), (int)d));
}
template <class T> inline T __shfl_down_sync(unsigned mask, T v, unsigned d, int = 32) {
    return gs_emu_unpack<T>(gs_emu::collective(gs_emu::OP_SHFL_DOWN, mask, gs_emu_pack(v), (int)d));
}
template <class T> inline T __shfl_xor_sync(unsigned mask, T v, int m, int = 32) {
    return gs_emu_unpack<T>(gs_emu::collective(gs_emu::OP_SHFL_XOR, mask, gs_emu_pack(v), m));
}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
/* one OS thread per block: read-modify-write is atomic by construction */
template <class T, class U> inline T atomicAdd(T* p, U v) { T o = *p; *p = o + (T)v; return o; }
template <class T, class U, class W> inline T atomicCAS(T* p, U c, W v) { T o = *p; if (o == (T)c) *p = (T)v; return o; }
template <class T, class U> inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
