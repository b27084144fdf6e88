// onesweep_kernels.hpp — gfx950 (CDNA4, wave64) device code of the OneSweep
// 8-bit LSD radix sort.  Written for MI355X only: 64-lane ballots, LDS-staged
// tiles, chained-scan decoupled look-back on agent-scope relaxed atomics (the
// descriptor word carries flag+count, so the data IS the flag — no fences).
//
// Behavioural spec (what, not how): reference b0nes164/GPUSorting
//   GlobalHistogram      GPUSortingCUDA/Sort/OneSweep.cu:44-123
//   Scan                 GPUSortingCUDA/Sort/OneSweep.cu:125-162
//   DigitBinningPass*    GPUSortingCUDA/Sort/OneSweep.cu:164-344, 346-600
//   key transforms       GPUSortingD3D12/Shaders/SortCommon.hlsl:134-154
//   descending rule      GPUSortingD3D12/Shaders/SortCommon.hlsl:594-597,645-656
//   InitRandom/Validate  GPUSortingCUDA/UtilityKernels.cuh:53-117, 402-479
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gs {

constexpr uint32_t RADIX = 256;
constexpr uint32_t FLAG_NOT_READY = 0;  // tile has published nothing yet
constexpr uint32_t FLAG_REDUCTION = 1;  // count<<2 = this tile's digit count
constexpr uint32_t FLAG_INCLUSIVE = 2;  // count<<2 = count of this and all earlier tiles (+ global digit start)
constexpr uint32_t FLAG_MASK = 3;

constexpr uint32_t STATUS_OK = 0;
constexpr uint32_t STATUS_TIMEOUT = 4;  // == GS_ERR_TIMEOUT

// Bound for every look-back spin (polls, each >= ~0.5 us with the sleep): ~1 s.
constexpr uint32_t SPIN_LIMIT = 1u << 21;

enum : int { KEY_U32 = 0, KEY_I32 = 1, KEY_F32 = 2 };

template <int KT>
__device__ __forceinline__ uint32_t to_bits(uint32_t u) {
    if constexpr (KT == KEY_I32) return u ^ 0x80000000u;
    if constexpr (KT == KEY_F32) return u ^ ((uint32_t)(-(int32_t)(u >> 31)) | 0x80000000u);
    return u;
}
template <int KT>
__device__ __forceinline__ uint32_t from_bits(uint32_t u) {
    if constexpr (KT == KEY_I32) return u ^ 0x80000000u;
    if constexpr (KT == KEY_F32) return u ^ (((u >> 31) - 1u) | 0x80000000u);
    return u;
}

__device__ __forceinline__ uint32_t ld_agent(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

// ---------------------------------------------------------------------------
// GlobalHistogram: one sweep over the keys, four 256-bin digit histograms.
// 16-byte loads, per-wave LDS histograms (ds_add_u32), one global atomic per
// non-empty bin per block.
// ---------------------------------------------------------------------------
constexpr int GHIST_THREADS = 256;
constexpr int GHIST_WAVES = GHIST_THREADS / 64;

template <int KT>
__global__ __launch_bounds__(GHIST_THREADS) void global_histogram_kernel(
    const uint32_t* __restrict__ keys, uint32_t* ghist, uint32_t n) {
    __shared__ uint32_t s_h[GHIST_WAVES][4 * RADIX];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < GHIST_WAVES * 4 * RADIX; i += GHIST_THREADS) (&s_h[0][0])[i] = 0;
    __syncthreads();

    uint32_t* h = s_h[tid >> 6];
    const uint32_t nvec = n >> 2;
    const uint4* kv = reinterpret_cast<const uint4*>(keys);
    const uint32_t stride = gridDim.x * GHIST_THREADS;
    for (uint32_t i = blockIdx.x * GHIST_THREADS + tid; i < nvec; i += stride) {
        const uint4 t = kv[i];
        const uint32_t k[4] = {to_bits<KT>(t.x), to_bits<KT>(t.y), to_bits<KT>(t.z), to_bits<KT>(t.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&h[k[j] & 255u], 1u);
            atomicAdd(&h[256u + ((k[j] >> 8) & 255u)], 1u);
            atomicAdd(&h[512u + ((k[j] >> 16) & 255u)], 1u);
            atomicAdd(&h[768u + (k[j] >> 24)], 1u);
        }
    }
    // scalar tail (n not a multiple of 4): handled by block 0
    if (blockIdx.x == 0) {
        for (uint32_t i = (nvec << 2) + tid; i < n; i += GHIST_THREADS) {
            const uint32_t k = to_bits<KT>(keys[i]);
            atomicAdd(&h[k & 255u], 1u);
            atomicAdd(&h[256u + ((k >> 8) & 255u)], 1u);
            atomicAdd(&h[512u + ((k >> 16) & 255u)], 1u);
            atomicAdd(&h[768u + (k >> 24)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < 4 * RADIX; b += GHIST_THREADS) {
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < GHIST_WAVES; ++w) s += s_h[w][b];
        if (s) atomicAdd(&ghist[b], s);
    }
}

// ---------------------------------------------------------------------------
// Scan: exclusive prefix of each 256-bin row; seeds descriptor row 0 of pass p
// as INCLUSIVE (so every look-back terminates at row 0 at worst).
// grid = 4 (one block per pass), block = 256.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_kernel(const uint32_t* ghist, uint32_t* desc,
                                                    uint32_t desc_stride /*words per pass*/) {
    __shared__ uint32_t s_wtot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t c = ghist[blockIdx.x * RADIX + tid];
    const uint32_t incl = wave_inclusive_scan(c, lane);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += s_wtot[w];
    desc[(size_t)blockIdx.x * desc_stride + tid] = ((base + incl - c) << 2) | FLAG_INCLUSIVE;
}

// ---------------------------------------------------------------------------
// DigitBinningPass: one stable 8-bit partition pass with chained-scan
// decoupled look-back.  One tile of THREADS*KPT keys per workgroup.
//   VB  = value bytes (0 keys-only, 4, 8)
//   KT  = key type
// Tile-local order == array order (keys are loaded wave-striped: wave w owns
// 64*KPT consecutive keys, round i = 64 consecutive keys), which is what makes
// each pass stable.
// ---------------------------------------------------------------------------
template <int VB>
struct ValT { using type = uint32_t; };
template <>
struct ValT<8> { using type = uint64_t; };

template <int THREADS, int KPT, int VB>
struct BinCfg {
    static constexpr int WAVES = THREADS / 64;
    static constexpr int TILE = THREADS * KPT;
    static constexpr int STAGE_BYTES = TILE * (VB == 8 ? 8 : 4);
    static constexpr int LDS_BYTES = STAGE_BYTES + WAVES * RADIX * 4 + 2 * RADIX * 4 + 64;
};

template <int THREADS, int KPT, int VB, int KT>
__global__ __launch_bounds__(THREADS) void digit_binning_kernel(
    const uint32_t* keys_in, uint32_t* keys_out, const void* vals_in_, void* vals_out_,
    uint32_t* desc,          // this pass: (tiles+1) rows of 256 descriptor words
    uint32_t* tile_counter,  // this pass's ticket counter
    uint32_t* status, uint32_t n, uint32_t shift, uint32_t reverse) {
    using Cfg = BinCfg<THREADS, KPT, VB>;
    using V = typename ValT<VB>::type;
    constexpr int WAVES = Cfg::WAVES;
    constexpr uint32_t TILE = Cfg::TILE;
    static_assert(THREADS >= 256 && THREADS % 64 == 0, "need >= 256 threads");

    __shared__ __attribute__((aligned(16))) unsigned char s_raw[Cfg::LDS_BYTES];
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(s_raw);
    uint32_t* s_whist = reinterpret_cast<uint32_t*>(s_raw + Cfg::STAGE_BYTES);
    uint32_t* s_dpre = s_whist + WAVES * RADIX;  // tile-local exclusive digit prefix
    uint32_t* s_gbase = s_dpre + RADIX;          // global base of digit run minus s_dpre
    uint32_t* s_misc = s_gbase + RADIX;          // [0] tile id, [4..7] wave totals of the digit scan

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
    if (tid == 0) s_misc[0] = atomicAdd(tile_counter, 1u);  // dynamic tile id: predecessors have started
    __syncthreads();
    const uint32_t tile = s_misc[0];
    const uint32_t tile_base = tile * TILE;
    const uint32_t count = (n - tile_base < TILE) ? (n - tile_base) : TILE;
    const bool full = (count == TILE);

    // ---- load (wave-striped, coalesced 256 B per wave-instruction) ----
    uint32_t key[KPT];
    const uint32_t my_base = tile_base + wave * (64u * KPT) + lane;
    if (full) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) key[i] = to_bits<KT>(keys_in[my_base + i * 64u]);
    } else {
        // dummy keys with the highest digit sort last inside the tile and are never written
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = my_base + i * 64u;
            key[i] = idx < n ? to_bits<KT>(keys_in[idx]) : 0xffffffffu;
        }
    }

    // ---- wave-level multi-split ranking (64-lane ballots) ----
    // For every round (64 consecutive keys, one per lane) each lane finds its
    // peers (lanes holding the same digit) with 8 ballots, ranks itself among
    // them with mbcnt, and the LAST peer bumps the wave's private LDS counter.
    // LDS operations of one wave execute in issue order, so the plain read of
    // round i+1 sees the write of round i; the asm memory clobber only stops the
    // compiler from reordering them.
    uint32_t* whist = s_whist + wave * RADIX;
    uint32_t off[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = (key[i] >> shift) & 255u;
        uint32_t acc_lo = 0, acc_hi = 0;  // bit l set <=> lane l's digit differs from mine
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t B = (uint32_t)__builtin_amdgcn_sbfe((int32_t)key[i], shift + k, 1);  // 0 or ~0
            const unsigned long long b = __builtin_amdgcn_ballot_w64(B != 0u);
            acc_lo = __builtin_amdgcn_bitop3_b32(acc_lo, (uint32_t)b, B, 0xF6);          // acc | (b ^ B)
            acc_hi = __builtin_amdgcn_bitop3_b32(acc_hi, (uint32_t)(b >> 32), B, 0xF6);
        }
        const uint32_t plo = ~acc_lo, phi = ~acc_hi;  // peers: lanes with my digit
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const uint32_t total = __popc(plo) + __popc(phi);
        const uint32_t pre = whist[d];            // same value for all peers (LDS broadcast)
        if (below == total - 1u) whist[d] = pre + total;  // last peer bumps the wave's counter
        asm volatile("" ::: "memory");
        off[i] = pre + below;                     // rank among this wave's keys of digit d
    }
    __syncthreads();

    // ---- per-digit: exclusive prefix over waves, tile total, publish, digit scan ----
    uint32_t tile_total = 0, scan_incl = 0;
    if (tid < RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = s_whist[w * RADIX + tid];
            s_whist[w * RADIX + tid] = run;
            run += c;
        }
        tile_total = run;
        st_agent(&desc[(size_t)(tile + 1u) * RADIX + tid], (run << 2) | FLAG_REDUCTION);
        scan_incl = wave_inclusive_scan(run, lane);
        if (lane == 63) s_misc[4 + wave] = scan_incl;
    }
    __syncthreads();
    uint32_t dpre = 0;
    if (tid < RADIX) {
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += s_misc[4 + w];
        dpre = wbase + scan_incl - tile_total;
        s_dpre[tid] = dpre;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += dpre;
    }
    __syncthreads();

    // ---- stage keys in LDS, sorted by digit (stable) ----
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = (key[i] >> shift) & 255u;
        off[i] += s_whist[wave * RADIX + d];
        s_stage[off[i]] = key[i];
    }

    // ---- decoupled look-back: one digit per thread ----
    if (tid < RADIX) {
        uint32_t prev = 0;
        uint32_t k = tile;  // row k holds tile k-1's descriptor; row 0 = global digit start (INCLUSIVE)
        uint32_t spins = 0;
        while (true) {
            const uint32_t v = ld_agent(&desc[(size_t)k * RADIX + tid]);
            const uint32_t f = v & FLAG_MASK;
            if (f == FLAG_INCLUSIVE) { prev += v >> 2; break; }
            if (f == FLAG_REDUCTION) { prev += v >> 2; --k; continue; }
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(status) != STATUS_OK)) {
                st_agent(status, STATUS_TIMEOUT);  // give up: result is invalid, but nothing hangs
                break;
            }
        }
        st_agent(&desc[(size_t)(tile + 1u) * RADIX + tid], ((prev + tile_total) << 2) | FLAG_INCLUSIVE);
        s_gbase[tid] = prev - dpre;
    }
    __syncthreads();

    // ---- scatter runs to global memory ----
    uint32_t dst[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t i = tid + j * THREADS;
        const uint32_t kb = s_stage[i];
        uint32_t o = s_gbase[(kb >> shift) & 255u] + i;
        if (reverse) o = n - 1u - o;
        if (full || i < count) keys_out[o] = from_bits<KT>(kb);
        else o = 0xffffffffu;
        dst[j] = o;
    }

    if constexpr (VB != 0) {
        const V* vals_in = reinterpret_cast<const V*>(vals_in_);
        V* vals_out = reinterpret_cast<V*>(vals_out_);
        V* s_vstage = reinterpret_cast<V*>(s_raw);
        V val[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = my_base + i * 64u;
            val[i] = (full || idx < n) ? vals_in[idx] : V(0);
        }
        __syncthreads();  // everyone is done reading the key stage
#pragma unroll
        for (int i = 0; i < KPT; ++i) s_vstage[off[i]] = val[i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t i = tid + j * THREADS;
            if (dst[j] != 0xffffffffu) vals_out[dst[j]] = s_vstage[i];
        }
    }
}

// ---------------------------------------------------------------------------
// Fixtures: InitRandom and Validate.
// ---------------------------------------------------------------------------
// Same 65536-virtual-thread structure as the reference's <<<256,256>>> launch so
// the generated array is defined element-for-element (UtilityKernels.cuh:53-117).
template <int VB>
__global__ __launch_bounds__(256) void init_random_kernel(uint32_t* keys, void* vals_, uint32_t and_count,
                                                           uint32_t seed, uint32_t n) {
    const uint32_t idx = threadIdx.x + blockDim.x * blockIdx.x;
    uint32_t z1 = (idx << 2) * seed;
    uint32_t z2 = ((idx << 2) + 1u) * seed;
    uint32_t z3 = ((idx << 2) + 2u) * seed;
    uint32_t z4 = ((idx << 2) + 3u) * seed;
    auto step = [&]() {
        z1 = ((z1 & 4294967294u) << 12) ^ (((z1 << 13) ^ z1) >> 19);
        z2 = ((z2 & 4294967288u) << 4) ^ (((z2 << 2) ^ z2) >> 25);
        z3 = ((z3 & 4294967280u) << 17) ^ (((z3 << 3) ^ z3) >> 11);
        z4 = z4 * 1664525u + 1013904223u;
    };
    step();
    for (uint64_t i = idx; i < n; i += 65536u) {
        uint32_t t = 0xffffffffu;
        for (uint32_t k = 0; k <= and_count; ++k) {
            step();
            t &= z1 ^ z2 ^ z3 ^ z4;
        }
        keys[i] = t;
        if constexpr (VB == 4) reinterpret_cast<uint32_t*>(vals_)[i] = t;
        if constexpr (VB == 8) reinterpret_cast<uint64_t*>(vals_)[i] = t;
    }
}

// Counts adjacent inversions in keys (and values), order/type aware.
template <int VB>
__global__ __launch_bounds__(256) void validate_kernel(const uint32_t* keys, const void* vals_, uint32_t n,
                                                        int key_type, int descending, uint32_t* err) {
    auto bits = [&](uint32_t u) {
        return key_type == KEY_I32 ? to_bits<KEY_I32>(u) : key_type == KEY_F32 ? to_bits<KEY_F32>(u) : u;
    };
    uint32_t bad = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += stride) {
        const uint32_t a = bits(keys[i]), b = bits(keys[i + 1]);
        bad += descending ? (a < b) : (a > b);
        if constexpr (VB == 4) {
            const uint32_t* v = reinterpret_cast<const uint32_t*>(vals_);
            const uint32_t x = bits(v[i]), y = bits(v[i + 1]);
            bad += descending ? (x < y) : (x > y);
        }
        if constexpr (VB == 8) {
            const uint64_t* v = reinterpret_cast<const uint64_t*>(vals_);
            bad += descending ? (v[i] < v[i + 1]) : (v[i] > v[i + 1]);
        }
    }
    bad = wave_reduce_sum(bad);
    if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(err, bad);
}

}  // namespace gs
