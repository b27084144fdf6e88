// r04_probe.hip — what this box streams, by access shape (round 4; not part of the product path).
//   floors  : hipMemcpyAsync D2D, read sweeps and copies with 1..8 16-byte loads in flight per thread at 1/2/4/8
//             workgroups per CU, plain vs non-temporal
//   gather  : the read side of a pass whose input is TILE-LOCALLY sorted by the previous digit: every 16 384-key
//             output tile reads 256 runs of ~64 keys at 4-byte-aligned offsets of 256 consecutive source tiles,
//             writes sequentially
//   scatter : the write side of a pass with 2^b bins and exact prefixes (no look-back): every tile writes
//             2^b runs of TILE / 2^b keys at 4-byte-aligned offsets, reads sequentially
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/r04_probe.hip -o build/r04_probe ; run: ./build/r04_probe [log2 n]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t v4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const v4* __restrict__ in, size_t n16, uint32_t* sink) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    v4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n16; i += stride) {
        v4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= t[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const v4* __restrict__ in, v4* __restrict__ out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n16; i += stride) {
        v4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = NTL ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(t[u], out + i + u * 256);
            else out[i + u * 256] = t[u];
        }
    }
}


// cache-resident working sets: the same small buffers copied / read REPS times in one launch (does the 256 MiB memory-side cache
// or the L2 lift the rate above the HBM floor?)
template <int U>
__global__ __launch_bounds__(256) void copy_rep_kernel(const v4* __restrict__ in, v4* __restrict__ out, size_t n16, int reps) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n16; i += stride) {
            v4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) t[u] = in[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) out[i + u * 256] = t[u];
        }
}
template <int U>
__global__ __launch_bounds__(256) void read_rep_kernel(const v4* __restrict__ in, size_t n16, int reps, uint32_t* sink) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    v4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n16; i += stride) {
            v4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) t[u] = in[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= t[u];
        }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// the pass's own shape: 512 threads, one 16 384-key tile per workgroup, 32 dword loads per lane (wave-striped), dword stores
template <bool NTL>
__global__ __launch_bounds__(512, 4) void tile_copy_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    const uint32_t base = blockIdx.x * 16384u + (threadIdx.x >> 6) * 2048u + (threadIdx.x & 63u);
    uint32_t k[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) k[i] = NTL ? __builtin_nontemporal_load(in + base + i * 64) : in[base + i * 64];
#pragma unroll
    for (int i = 0; i < 32; ++i) out[base + i * 64] = k[i];
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// gather: output tile j = (digit d = j / G, source group g = j % G) with G = tiles / 256; run r of it comes from source tile
// g * 256 + r, offset d * 64 + jitter(d, tile) where jitter in [0, 8) keys keeps every run inside its own 72-key slot
// (source slots of 72 keys would overrun: instead the run length is 56 + jitter' so that runs tile [d*64, d*64+64) loosely).
// Simplest faithful form: run (d, t) = source keys [t*16384 + d*64 + a(d,t), t*16384 + (d+1)*64 + a(d+1,t)) with a(0,t) = 0,
// a(256,t) = 0, a(d,t) in [0, 16): lengths 49..79, arbitrary 4-byte alignment, every source key read exactly once.
__device__ __forceinline__ uint32_t jit(uint32_t d, uint32_t t) { return (d == 0u || d == 256u) ? 0u : (hash32(d * 0x9e3779b1u + t) & 15u); }

template <bool STAGE>
__global__ __launch_bounds__(512, 4) void gather_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t tiles,
                                                        uint32_t* __restrict__ meta /*[256][tiles]: run start in the OUTPUT of digit d*/) {
    __shared__ uint32_t s_start[257];   // start of run r inside this output tile
    __shared__ uint32_t s_src[256];     // source index of run r's first key
    __shared__ uint32_t s_stage[STAGE ? 16384 + 128 : 1];
    const uint32_t G = tiles / 256u;
    const uint32_t j = blockIdx.x, d = j / G, g = j % G, tid = threadIdx.x;
    if (tid < 256) {
        const uint32_t t = g * 256u + tid;
        const uint32_t a0 = jit(d, t), a1 = jit(d + 1u, t);
        s_src[tid] = t * 16384u + d * 64u + a0;
        s_start[tid] = 64u + a1 - a0;  // length for now
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan of 256 lengths by one wave (4 per lane)
        uint32_t l0 = s_start[tid * 4], l1 = s_start[tid * 4 + 1], l2 = s_start[tid * 4 + 2], l3 = s_start[tid * 4 + 3];
        uint32_t s = l0 + l1 + l2 + l3, inc = s;
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(inc, o, 64); if ((int)tid >= o) inc += v; }
        uint32_t ex = inc - s;
        s_start[tid * 4] = ex; s_start[tid * 4 + 1] = ex + l0; s_start[tid * 4 + 2] = ex + l0 + l1; s_start[tid * 4 + 3] = ex + l0 + l1 + l2;
        if (tid == 63) s_start[256] = inc;
    }
    __syncthreads();
    const uint32_t total = s_start[256];
    // every lane finds its run by position: p -> run = p / 64 +- a few (lengths 49..79): linear probe from the guess
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    uint32_t k[32];
    uint32_t pos[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const uint32_t p = wave * 2048u + i * 64u + lane;
        uint32_t r = p >> 6;
        r = r > 255u ? 255u : r;
        while (r > 0u && s_start[r] > p) --r;
        while (r < 255u && s_start[r + 1] <= p) ++r;
        pos[i] = p;
        const uint32_t src = s_src[r] + (p - s_start[r]);
        k[i] = p < total ? __builtin_nontemporal_load(in + src) : 0u;
    }
    // output: the tile's keys go out sequentially at meta-free position j * 16384 (lengths differ by tile: use a 16 384 slot, masked)
    uint32_t* o = out + (size_t)j * 16384u;
    if (STAGE) {
#pragma unroll
        for (int i = 0; i < 32; ++i) s_stage[pos[i]] = k[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; ++i) { const uint32_t p = tid + i * 512u; if (p < total) o[p] = s_stage[p]; }
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) if (pos[i] < total) o[pos[i]] = k[i];
    }
    (void)meta;
}


// gather2: the same access shape with ARITHMETIC addresses (no LDS search), to separate the memory system from the address work.
//   MODE 0: every wave-instruction reads one whole 64-key run, 256-byte aligned          (aligned chunks of 256 source tiles)
//   MODE 1: ... at a 4-byte-aligned offset (jitter 0..15 keys)                             (misaligned runs)
//   MODE 2: ... and split at a lane: lanes >= cut read the head of the NEXT run (another source tile)   (what a pass really sees)
// ORDER 0: output tile j = d * G + g (digit-major: the virtual order of the pass); ORDER 1: j = g * 256 + d (source-group-major)
template <int MODE, int ORDER, bool NT = true>
__global__ __launch_bounds__(512, 4) void gather2_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t tiles) {
    const uint32_t G = tiles / 256u;
    const uint32_t j = blockIdx.x;
    // ORDER 2: the pass's chain order — block b works for chain b % 16 (its XCD: b % 8), ticket b / 16; a chain = 16 digits, digit-major
    uint32_t d = ORDER == 0 ? j / G : j % 256u, g = ORDER == 0 ? j % G : j / 256u;
    if (ORDER >= 2) {
        uint32_t ch = j % 16u, k = j / 16u;
        if (ORDER >= 4) { ch = (j >> 3) & 15u; k = (j & 7u) | ((j >> 7) << 3); }  // ORDER 4, 5: a chain's consecutive units go round the 8 XCDs
        d = ch * 16u + k / G; g = k % G;
    }
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    uint32_t k[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const uint32_t r = wave * 32u + i;            // run of this wave-instruction
        uint32_t t = g * 256u + r, off = 0;
        if (MODE >= 1) off = hash32(d * 0x9e3779b1u + t) & 15u;
        if (MODE == 2) {
            const uint32_t cut = hash32(j * 977u + r) & 63u;
            if (lane >= cut) { t = g * 256u + ((r + 1u) & 255u); off = (hash32(d * 0x9e3779b1u + t) & 15u); }
        }
        // ORDER 3: the source tile holds its runs in NIBBLE-SWAPPED digit order: the 16 chains, in step, read 16 neighbouring runs
        const uint32_t ds = (ORDER == 3 || ORDER == 5) ? (((d & 15u) << 4) | (d >> 4)) : d;
        const uint32_t dd = ds == 255u && MODE >= 1 ? 254u : ds;  // (jitter must not run past the source tile)
        const uint32_t* q = in + (size_t)t * 16384u + dd * 64u + off + lane;
        k[i] = NT ? __builtin_nontemporal_load(q) : *q;
    }
    uint32_t* o = out + (size_t)(d * G + g) * 16384u + wave * 2048u + lane;
#pragma unroll
    for (int i = 0; i < 32; ++i) o[i * 64] = k[i];
}

// scatter: tile t reads 16 384 keys sequentially and writes run b (RUN keys) of bin b at out[b * stride + off(b) + t * RUN + l],
// off(b) in [0, 16) keys: 4-byte aligned runs of RUN keys, every output word written exactly once (gaps of < 16 keys between bins)
template <int RUN>
__global__ __launch_bounds__(512, 4) void scatter_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t tiles, uint32_t chains) {
    constexpr uint32_t BINS = 16384u / RUN;
    // XCD-friendly: a chain's tiles follow each other in ONE L2 (block b -> XCD b % 8 and chain b % chains)
    const uint32_t per_chain = tiles / chains;
    const uint32_t t = (blockIdx.x % chains) * per_chain + blockIdx.x / chains;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t base = t * 16384u + wave * 2048u + lane;
    uint32_t k[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) k[i] = __builtin_nontemporal_load(in + base + i * 64);
    const size_t stride = (size_t)tiles * RUN + 16u;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const uint32_t p = wave * 2048u + i * 64u + lane;  // slot of the (already sorted) stage
        const uint32_t b = p / RUN, l = p % RUN;
        out[(size_t)b * stride + (hash32(b) & 15u) + (size_t)t * RUN + l] = k[i];
    }
    (void)BINS;
}

static float time_ms(hipStream_t s, int reps, const std::function<void()>& f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, s));
        f();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return best;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const size_t n = (size_t)1 << lg;
    const size_t bytes = n * 4;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# r04_probe: %s, %d CUs, n = 2^%d u32 (%.0f MiB per buffer)\n", prop.gcnArchName, cus, lg, bytes / 1048576.0);
    uint32_t *in, *out, *sink;
    CK(hipMalloc(&in, bytes + 4096)); CK(hipMalloc(&out, bytes + (1u << 20) * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 0x5a, bytes)); CK(hipMemset(out, 0, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int R = 7;
    auto rep = [&](const char* name, float ms, double moved) { printf("%-58s %8.4f ms  %7.3f TB/s\n", name, ms, moved / ms * 1e-9); fflush(stdout); };

    rep("hipMemcpyAsync D2D (r+w)", time_ms(s, R, [&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, s)); }), 2.0 * bytes);
    const size_t n16 = bytes / 16;
    char nm[128];
#define RD(U, NT) for (int wpc : {1, 2, 4, 8}) { snprintf(nm, sizeof nm, "read  U=%d %s wg/cu=%d", U, NT ? "nt" : "  ", wpc); \
        rep(nm, time_ms(s, R, [&] { hipLaunchKernelGGL((read_kernel<U, NT>), dim3(cus * wpc), dim3(256), 0, s, (const v4*)in, n16, sink); }), 1.0 * bytes); }
    RD(1, false) RD(2, false) RD(4, false) RD(8, false) RD(4, true) RD(8, true)
#define CP(U, NTL, NTS) for (int wpc : {1, 2, 4, 8}) { snprintf(nm, sizeof nm, "copy  U=%d ld%s st%s wg/cu=%d", U, NTL ? "nt" : "  ", NTS ? "nt" : "  ", wpc); \
        rep(nm, time_ms(s, R, [&] { hipLaunchKernelGGL((copy_kernel<U, NTL, NTS>), dim3(cus * wpc), dim3(256), 0, s, (const v4*)in, (v4*)out, n16); }), 2.0 * bytes); }
    CP(1, false, false) CP(2, false, false) CP(4, false, false) CP(8, false, false) CP(4, true, false) CP(4, true, true) CP(8, true, false) CP(4, false, true)
    // one-shot grids (no grid-stride loop): every workgroup moves 256 x U x 16 bytes
    for (int big : {1, 2}) {
        const uint32_t grid = (uint32_t)(n16 / (256 * 4)) / big;
        snprintf(nm, sizeof nm, "copy  U=4 ldnt one-shot grid=%u", grid);
        rep(nm, time_ms(s, R, [&] { hipLaunchKernelGGL((copy_kernel<4, true, false>), dim3(grid), dim3(256), 0, s, (const v4*)in, (v4*)out, n16); }), 2.0 * bytes);
    }

    for (int mb : {8, 16, 32, 64, 128, 256, 512}) {
        const size_t b16 = (size_t)mb * 1048576 / 16;
        const int reps = (int)(1024 / mb);
        snprintf(nm, sizeof nm, "copy %4d MiB -> %4d MiB, %3d times in one launch", mb, mb, reps);
        rep(nm, time_ms(s, 5, [&] { hipLaunchKernelGGL((copy_rep_kernel<4>), dim3(cus * 4), dim3(256), 0, s, (const v4*)in, (v4*)out, b16, reps); }), 2.0 * mb * 1048576.0 * reps);
        snprintf(nm, sizeof nm, "read %4d MiB, %3d times in one launch", mb, reps);
        rep(nm, time_ms(s, 5, [&] { hipLaunchKernelGGL((read_rep_kernel<4>), dim3(cus * 4), dim3(256), 0, s, (const v4*)in, b16, reps, sink); }), 1.0 * mb * 1048576.0 * reps);
    }
    const uint32_t tiles = (uint32_t)(n / 16384);
    rep("tile copy 512x32 dword (pass shape) plain loads", time_ms(s, R, [&] { hipLaunchKernelGGL((tile_copy_kernel<false>), dim3(tiles), dim3(512), 0, s, in, out); }), 2.0 * bytes);
    rep("tile copy 512x32 dword (pass shape) nt loads", time_ms(s, R, [&] { hipLaunchKernelGGL((tile_copy_kernel<true>), dim3(tiles), dim3(512), 0, s, in, out); }), 2.0 * bytes);

    if (tiles >= 256 && tiles % 256 == 0) {
        rep("gather 256 runs x ~64 keys -> sequential (direct)", time_ms(s, R, [&] { hipLaunchKernelGGL((gather_kernel<false>), dim3(tiles), dim3(512), 0, s, in, out, tiles, (uint32_t*)nullptr); }), 2.0 * bytes);
        rep("gather 256 runs x ~64 keys -> LDS stage -> sequential", time_ms(s, R, [&] { hipLaunchKernelGGL((gather_kernel<true>), dim3(tiles), dim3(512), 0, s, in, out, tiles, (uint32_t*)nullptr); }), 2.0 * bytes);
    }

    if (tiles >= 256 && tiles % 256 == 0) {
#define G2(M, O, label) rep(label, time_ms(s, R, [&] { hipLaunchKernelGGL((gather2_kernel<M, O>), dim3(tiles), dim3(512), 0, s, in, out, tiles); }), 2.0 * bytes);
        G2(0, 0, "gather2 aligned runs, digit-major tiles")
        G2(1, 0, "gather2 misaligned runs, digit-major tiles")
        G2(2, 0, "gather2 misaligned + split runs, digit-major tiles")
#define G2P(M, O, label) rep(label, time_ms(s, R, [&] { hipLaunchKernelGGL((gather2_kernel<M, O, false>), dim3(tiles), dim3(512), 0, s, in, out, tiles); }), 2.0 * bytes);
        G2P(0, 0, "gather2 aligned runs, digit-major, PLAIN loads")
        G2P(1, 0, "gather2 misaligned runs, digit-major, PLAIN loads")
        G2P(2, 0, "gather2 misaligned + split, digit-major, PLAIN loads")
        G2P(2, 1, "gather2 misaligned + split, group-major, PLAIN loads")
        G2P(2, 2, "gather2 misaligned + split, CHAIN order, PLAIN loads")
        G2(2, 2, "gather2 misaligned + split, CHAIN order, nt loads")
        G2P(0, 2, "gather2 aligned, CHAIN order, PLAIN loads")
        G2P(2, 3, "gather2 misaligned + split, CHAIN order, nibble-swapped runs, PLAIN")
        G2P(0, 3, "gather2 aligned, CHAIN order, nibble-swapped runs, PLAIN")
        G2P(0, 4, "gather2 aligned, CHAIN order spread over XCDs, PLAIN")
        G2P(2, 4, "gather2 misaligned + split, CHAIN order spread over XCDs, PLAIN")
        G2P(0, 5, "gather2 aligned, CHAIN order spread over XCDs, nibble-swapped, PLAIN")
        G2P(2, 5, "gather2 misaligned + split, CHAIN order spread, nibble-swapped, PLAIN")
        G2(0, 1, "gather2 aligned runs, group-major tiles")
        G2(1, 1, "gather2 misaligned runs, group-major tiles")
        G2(2, 1, "gather2 misaligned + split runs, group-major tiles")
    }
#define SC(RUN) for (uint32_t ch : {1u, 16u}) { snprintf(nm, sizeof nm, "scatter runs of %d keys (%d bins), %u chains", RUN, 16384 / RUN, ch); \
        rep(nm, time_ms(s, R, [&] { hipLaunchKernelGGL((scatter_kernel<RUN>), dim3(tiles), dim3(512), 0, s, in, out, tiles, ch); }), 2.0 * bytes); }
    SC(1024) SC(256) SC(128) SC(64) SC(32) SC(16) SC(8) SC(4)
    printf("# done\n");
    return 0;
}
