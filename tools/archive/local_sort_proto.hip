// local_sort_proto.hip — standalone timing/validation of local_sort_kernel on a synthetic array ordered by its top 16
// bits (tuning aid).  Usage: local_sort_proto [log2n=28] [tile_keys=10752] [jitter=128]
#include "local_sort_proto_kernel.hpp"
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// bucket b covers [b * L + off(b), (b + 1) * L + off(b + 1)), off(b) in [-jit, jit]
__global__ void gen(uint32_t* k, uint32_t n, uint32_t L, uint32_t jit, unsigned long long* sum) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    auto off = [&](uint32_t b) -> int { return b == 0 ? 0 : (int)(mix(b * 2654435761u) % (2 * jit + 1)) - (int)jit; };
    uint32_t b = p / L;
    if ((long long)p < (long long)b * L + off(b)) --b;
    else if ((long long)p >= (long long)(b + 1) * L + off(b + 1)) ++b;
    if (b > 65535u) b = 65535u;
    const uint32_t key = (b << 16) | (mix(p + 12345u) & 0xffffu);
    k[p] = key;
    atomicAdd(sum, (unsigned long long)key * 0x9e3779b97f4a7c15ull + 1ull);
}
__global__ void check(const uint32_t* k, uint32_t n, unsigned long long* sum, uint32_t* bad) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (p && k[p - 1] > k[p]) atomicAdd(bad, 1u);
    atomicAdd(sum, (unsigned long long)k[p] * 0x9e3779b97f4a7c15ull + 1ull);
}

int main(int argc, char** argv) {
    const uint32_t lg = argc > 1 ? atoi(argv[1]) : 28, T = argc > 2 ? atoi(argv[2]) : 10752, jit = argc > 3 ? atoi(argv[3]) : 128;
    const uint32_t n = 1u << lg, L = n >> 16;
    uint32_t *k, *k0, *fail;
    unsigned long long* sums;
    CK(hipMalloc(&k, (size_t)n * 4)); CK(hipMalloc(&k0, (size_t)n * 4)); CK(hipMalloc(&fail, 8)); CK(hipMalloc(&sums, 16));
    CK(hipMemset(sums, 0, 16)); CK(hipMemset(fail, 0, 8));
    gen<<<(n + 255) / 256, 256>>>(k0, n, L ? L : 1, jit, sums);
    const uint32_t grid = (n + T - 1) / T;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        CK(hipMemcpy(k, k0, (size_t)n * 4, hipMemcpyDeviceToDevice));
        CK(hipEventRecord(a));
        gs::local_sort_kernel<0><<<grid, gs::LOC_THREADS>>>(k, n, T, fail);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it && ms < best) best = ms;
    }
    check<<<(n + 255) / 256, 256>>>(k, n, sums + 1, fail + 1);
    unsigned long long hs[2]; uint32_t hf[2];
    CK(hipMemcpy(hs, sums, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, fail, 8, hipMemcpyDeviceToHost));
    printf("n=2^%u tile=%u grid=%u jitter=%u: %.3f ms  (%.1f GB/s r+w)  overflow_tiles=%u unsorted_pairs=%u checksum=%s\n", lg, T, grid, jit,
           best, 8.0 * n / best * 1e-6, hf[0], hf[1], hs[0] == hs[1] ? "ok" : "MISMATCH");
    return 0;
}
