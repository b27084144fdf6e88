// r04_probe_gather3.hip — could the gather pass of the local-sort plan be fed faster?  (round 4; not part of the product path)
// The pass's real shape: 512 threads, 64 KiB LDS stage (two workgroups per CU), persistent, tiles by ticket in CHAIN order.
//   MODE 0  linear tile: wave-striped register loads -> stage -> sequential stores                      (the linear pass's load side)
//   MODE 1  gathered tile: per-tile run DESCRIPTOR (256 x 8 B, one coalesced dependent load) -> every wave loads whole runs
//           straight into the stage (global_load_lds, lane-contiguous) -> keys read back in tile order -> stage -> sequential stores
//   MODE 2  the same with the runs loaded through registers (ds_write instead of the direct-to-LDS load)
// Source layout as in r04_probe.hip's gather: run (d, t) = in[t * 16384 + d * 64 + a(d, t) .. t * 16384 + (d + 1) * 64 + a(d + 1, t)), a in [0, 16).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/r04_probe_gather3.hip -o build/r04_probe_gather3 ; run: ./build/r04_probe_gather3 [log2 n]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t jit(uint32_t d, uint32_t t) { return (d == 0u || d == 256u) ? 0u : (hash32(d * 0x9e3779b1u + t) & 15u); }

// virtual tile j (chain order: chain = j % 16, ticket k = j / 16; digit d = chain * 16 + k / G, source group g = k % G)
__device__ __forceinline__ void tile_of(uint32_t chain, uint32_t k, uint32_t G, uint32_t& d, uint32_t& g) { d = chain * 16u + k / G; g = k % G; }

__global__ void desc_kernel(uint2* __restrict__ desc, uint32_t tiles) {  // one workgroup of 256 per virtual tile (index = d * G + g)
    const uint32_t G = tiles / 256u, j = blockIdx.x, d = j / G, g = j % G, r = threadIdx.x;
    __shared__ uint32_t s_len[256];
    const uint32_t t = g * 256u + r, a0 = jit(d, t), a1 = jit(d + 1u, t);
    s_len[r] = 64u + a1 - a0;
    __syncthreads();
    uint32_t start = 0;
    for (uint32_t i = 0; i < r; ++i) start += s_len[i];
    desc[(size_t)j * 256u + r] = uint2{t * 16384u + d * 64u + a0, start | (s_len[r] << 16)};
}

template <int MODE>
__global__ __launch_bounds__(512, 4) void pass_shape_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint2* __restrict__ desc,
                                                            uint32_t* __restrict__ counters, uint32_t tiles) {
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[16384 + 256];
    __shared__ uint2 s_desc[256];
    __shared__ uint32_t s_ticket;
    const uint32_t G = tiles / 256u, per_chain = tiles / 16u;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, chain = blockIdx.x & 15u;
#pragma unroll 1
    for (;;) {
        __syncthreads();
        if (tid == 0) s_ticket = atomicAdd(&counters[chain * 32u], 1u);
        __syncthreads();
        const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket);
        if (k >= per_chain) break;
        uint32_t d, g;
        tile_of(chain, k, G, d, g);
        const uint32_t j = d * G + g;
        uint32_t key[32];
        uint32_t total = 16384u;
        if constexpr (MODE == 0) {
            const uint32_t* p = in + (size_t)j * 16384u + wave * 2048u + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) key[i] = __builtin_nontemporal_load(p + i * 64);
        } else {
            if (tid < 256) s_desc[tid] = desc[(size_t)j * 256u + tid];
            __syncthreads();
            // wave w loads runs w, w + 8, ...: lane-contiguous pieces of <= 64 keys, straight into the stage at the run's tile position
#pragma unroll 4
            for (uint32_t r = wave; r < 256u; r += 8u) {
                const uint2 e = s_desc[r];
                const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
                const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
                const uint32_t start = sl & 0xffffu, len = sl >> 16;
                if constexpr (MODE == 1) {
                    if (lane < len)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + src + lane),
                                                         (__attribute__((address_space(3))) void*)(s_stage + start), 4, 0, 0);
                    if (len > 64u && lane + 64u < len)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + src + 64u + lane),
                                                         (__attribute__((address_space(3))) void*)(s_stage + start + 64u), 4, 0, 0);
                } else {
                    uint32_t v0 = 0, v1 = 0;
                    if (lane < len) v0 = in[src + lane];
                    if (len > 64u && lane + 64u < len) v1 = in[src + 64u + lane];
                    if (lane < len) s_stage[start + lane] = v0;
                    if (len > 64u && lane + 64u < len) s_stage[start + 64u + lane] = v1;
                }
            }
            if (tid == 255) { const uint2 e = s_desc[255]; s_desc[0].x = (e.y & 0xffffu) + (e.y >> 16); }  // (total)
            __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the direct loads have landed
            __syncthreads();
            total = s_desc[0].x;
#pragma unroll
            for (int i = 0; i < 32; ++i) key[i] = s_stage[wave * 2048u + i * 64 + lane];
            __syncthreads();
        }
        // the pass would rank here; stage at "sorted" positions (identity) and write the tile out sequentially
#pragma unroll
        for (int i = 0; i < 32; ++i) s_stage[wave * 2048u + i * 64 + lane] = key[i];
        __syncthreads();
        uint32_t* o = out + (size_t)j * 16384u;
#pragma unroll
        for (int i = 0; i < 32; ++i) { const uint32_t p = tid + i * 512u; if (p < total) o[p] = s_stage[p]; }
    }
}

static float time_ms(hipStream_t s, int reps, const std::function<void()>& f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, s)); f(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    CK(hipGetLastError());
    return best;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const size_t n = (size_t)1 << lg, bytes = n * 4;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t tiles = (uint32_t)(n / 16384);
    printf("# r04_probe_gather3: %s, %d CUs, n = 2^%d u32, %u tiles, persistent grid %d x 512 threads\n", prop.gcnArchName, cus, lg, tiles, 2 * cus);
    uint32_t *in, *out, *counters; uint2* desc;
    CK(hipMalloc(&in, bytes + 4096)); CK(hipMalloc(&out, bytes + (1u << 20) * 4)); CK(hipMalloc(&counters, 16 * 32 * 4));
    CK(hipMalloc(&desc, (size_t)tiles * 256 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    // in[i] = i so that the gathered output can be checked
    { std::vector<uint32_t> h(1 << 20); for (size_t off = 0; off < n; off += h.size()) { for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(off + i); CK(hipMemcpy(in + off, h.data(), h.size() * 4, hipMemcpyHostToDevice)); } }
    hipLaunchKernelGGL(desc_kernel, dim3(tiles), dim3(256), 0, s, desc, tiles);
    CK(hipStreamSynchronize(s));
    auto run = [&](auto kern) { CK(hipMemsetAsync(counters, 0, 16 * 32 * 4, s)); hipLaunchKernelGGL(kern, dim3(2 * cus), dim3(512), 0, s, in, out, desc, counters, tiles); };
    auto rep = [&](const char* name, float ms) { printf("%-86s %8.4f ms  %7.3f TB/s\n", name, ms, 2.0 * bytes / ms * 1e-9); fflush(stdout); };
    rep("MODE 0 linear tile (register loads -> stage -> sequential stores)", time_ms(s, 7, [&] { run(pass_shape_kernel<0>); }));
    rep("MODE 1 gathered tile: descriptor + run-wise DIRECT-TO-LDS loads -> regs -> stage -> stores", time_ms(s, 7, [&] { run(pass_shape_kernel<1>); }));
    // check MODE 1: out tile j position p must hold the source index the descriptor says
    {
        std::vector<uint32_t> ho(16384); std::vector<uint2> hd(256);
        uint32_t bad = 0;
        for (uint32_t j : {0u, 1u, tiles / 2 + 3, tiles - 1}) {
            CK(hipMemcpy(ho.data(), out + (size_t)j * 16384, 16384 * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hd.data(), desc + (size_t)j * 256, 256 * 8, hipMemcpyDeviceToHost));
            for (int r = 0; r < 256; ++r) { uint32_t st = hd[r].y & 0xffff, len = hd[r].y >> 16; for (uint32_t i = 0; i < len; ++i) if (st + i < 16384 && ho[st + i] != hd[r].x + i) ++bad; }
        }
        printf("# MODE 1 check: %u wrong words in 4 tiles\n", bad);
    }
    rep("MODE 2 gathered tile: descriptor + run-wise register loads + ds_write -> regs -> stage -> stores", time_ms(s, 7, [&] { run(pass_shape_kernel<2>); }));
    rep("MODE 0 again", time_ms(s, 7, [&] { run(pass_shape_kernel<0>); }));
    printf("# done\n");
    return 0;
}
