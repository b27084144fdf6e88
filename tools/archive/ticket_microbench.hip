// ticket_microbench.hip — how fast can workgroups draw tickets from ONE device-scope counter?  (tuning aid)
// Every workgroup of 512 threads draws `per_wg` tickets (thread 0, returning atomicAdd, dependent chain) from counter
// blockIdx % ncounters; counters are 256 bytes apart.  Reported: ns per ticket per counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s:%d\n", __FILE__, __LINE__); return 1; } } while (0)

template <int SCOPE>
__global__ __launch_bounds__(512) void draw(uint32_t* counters, uint32_t ncounters, uint32_t per_wg, uint32_t* sink) {
    __shared__ uint32_t s_t;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_wg; ++i) {
        if (threadIdx.x == 0) s_t = __hip_atomic_fetch_add(&counters[(blockIdx.x % ncounters) * 64u], 1u, __ATOMIC_RELAXED, SCOPE);
        __syncthreads();
        acc += s_t;
        __syncthreads();
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

int main() {
    uint32_t *c, *sink;
    CK(hipMalloc(&c, 64 * 64 * 4)); CK(hipMalloc(&sink, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t grids[] = {16384, 512};
    for (uint32_t g = 0; g < 2; ++g)
        for (uint32_t nc : {1u, 2u, 4u, 16u}) {
            const uint32_t grid = grids[g], per = g == 0 ? 1u : 64u;
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                CK(hipMemset(c, 0, 64 * 64 * 4));
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(draw<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(512), 0, 0, c, nc, per, sink);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (it && ms < best) best = ms;
            }
            printf("grid=%5u tickets/wg=%2u counters=%2u: %8.1f us total, %6.1f ns per ticket per counter\n", grid, per, nc, best * 1e3,
                   best * 1e6 / ((double)grid * per / nc));
        }
    return 0;
}
