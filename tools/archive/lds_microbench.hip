// lds_microbench.hip — what an LDS atomic costs on this chip, by address pattern (tuning aid, not product).
// The GlobalHistogram kernel does 4 ds_add_u32 per key and is LDS-bound; the DigitBinningPass ranks with one
// ds_add_rtn_u32 per key.  This program measures wave-instructions per microsecond per CU for:
//   pattern 0  conflict-free        (lane l -> its own bank)
//   pattern 1  random over 256 bins of a WAVE-private table      (the ranking shape)
//   pattern 2  random over 256 bins of a workgroup-shared table  (histogram of pass 0)
//   pattern 3  random over 4096 bins of a workgroup-shared table (joint histograms of passes 1..3)
//   pattern 4  all lanes one address
// each as non-returning add, returning add, plain read and plain write.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_microbench.hip -o build/lds_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s:%d\n", __FILE__, __LINE__); return; } } while (0)

constexpr int NADDR = 16;

template <int THREADS, int OP>
__global__ __launch_bounds__(THREADS) void lds_kernel(uint32_t pattern, uint32_t iters, uint32_t* sink) {
    __shared__ uint32_t tab[16384];  // 64 KiB: two workgroups per CU like the histogram kernel
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < 16384; i += THREADS) tab[i] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * THREADS + tid) * 2654435761u + 12345u;
    uint32_t addr[NADDR];
#pragma unroll
    for (int j = 0; j < NADDR; ++j) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        uint32_t a;
        if (pattern == 0) a = (wave * 1024u + ((j & 3) * 64u) + lane) & 16383u;
        else if (pattern == 1) a = (wave * 256u + (x >> 24)) & 16383u;
        else if (pattern == 2) a = x >> 24;
        else if (pattern == 3) a = x >> 20;
        else a = 7;
        addr[j] = a;
    }
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NADDR; ++j) {
            if constexpr (OP == 0) __hip_atomic_fetch_add(&tab[addr[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (OP == 1) acc += __hip_atomic_fetch_add(&tab[addr[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (OP == 2) acc += *(volatile uint32_t*)&tab[addr[j]];
            else *(volatile uint32_t*)&tab[addr[j]] = acc + j;
        }
        if constexpr (OP == 0) asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (acc == 0x12345u || tab[tid] == 0xffffffffu) sink[0] = acc;
}

template <int THREADS, int OP>
static void run(const char* name, int blocks_per_cu) {
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const uint32_t iters = 2000, grid = 256 * blocks_per_cu;
    for (uint32_t pattern = 0; pattern < 5; ++pattern) {
        hipLaunchKernelGGL((lds_kernel<THREADS, OP>), dim3(grid), dim3(THREADS), 0, 0, pattern, 10u, sink);
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((lds_kernel<THREADS, OP>), dim3(grid), dim3(THREADS), 0, 0, pattern, iters, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        const double waveops_per_cu = (double)blocks_per_cu * (THREADS / 64) * iters * NADDR;
        printf("%-10s threads=%4d wg/cu=%d pattern=%u: %8.3f ms  %7.2f ns per wave-op per CU (%5.2f clk @2.4GHz)\n", name, THREADS,
               blocks_per_cu, pattern, ms, ms * 1e6 / waveops_per_cu, ms * 1e6 / waveops_per_cu * 2.4);
    }
    CK(hipFree(sink));
}

int main() {
    run<1024, 0>("add", 2);
    run<1024, 1>("add_rtn", 2);
    run<512, 0>("add", 2);
    run<512, 1>("add_rtn", 2);
    run<512, 2>("read", 2);
    run<512, 3>("write", 2);
    run<256, 0>("add", 4);
    return 0;
}
