// mid_kernels.hpp — the TWO-launch sort of mid-size inputs (single-tile limit < n <= 2^20; keys-only and 4-byte
// values up to 2^21, keys-only up to 2^22, with larger tiles).
//
// SURVEY.md 8f N1 / the reference's BenchmarkOneSweep size sweep (GPUSortingD3D12/Tests.h:392-393,415-416): between
// 2^14 and 2^20 keys the six-launch pipeline is a ~50 us plateau of launch and tile latencies (profiles/
// r01_size_and_entropy_sweep_v3.txt) — four global passes of >= 9 us each whatever the size.  At these sizes the keys
// of one top-byte value fit ONE workgroup's LDS, so the sort is done as
//   K1 mid_msd_kernel     one MSD pass: every workgroup (one tile each — 8192 keys, 16 384 above 2^20, 32 768 above 2^21 —
//                         at most 128, all resident) ranks its
//                         tile by the TOP byte, publishes its 256 counts, meets the others at ONE grid barrier, derives
//                         every bucket's start and its own offsets from the count table, and scatters its tile into the
//                         alt buffer (stable);
//   K2 bucket_sort_kernel one workgroup per top-byte bucket sorts it on the remaining 24 bits entirely in LDS (three
//                         stable passes, as the single-tile sort) and writes it to its final place in the key buffer.
// Two global passes instead of four (20 B/key instead of 36), two launches instead of six.  Same result as the LSD
// sort: (stable by top byte) o (stable sort of each bucket by the low 24 bits) == stable sort by the whole key.
// If a bucket would not fit a workgroup (skewed top byte) every workgroup sees that in the SAME count table and K1
// runs the four LSD passes itself, with grid barriers between them (about the cost of the six-launch path, no
// extra launch, no host decision); K2 then finds the route flag and exits.
// Cross-workgroup data inside K1: the count table only — write-through (sc1) stores, drained, then the arrival
// atomic; readers poll with sc1 loads and read the table with sc1 loads (MI355X_MICROARCH.md, valid forms).  On the
// LSD route the keys themselves cross workgroups between passes: written and read with sc1 accesses as well.
#pragma once
#include "onesweep_kernels.hpp"

namespace gs {

// Three tile shapes, always at most 128 tiles; a top-byte bucket must fit ONE tile (K2 sorts it in LDS):
//   512 x 16 =  8 192 keys  n <= 2^20   every value width
//   512 x 32 = 16 384 keys  n <= 2^21   keys-only and 4-byte values (stage 64 + 64 KiB)
//  1024 x 32 = 32 768 keys  n <= 2^22   keys-only (stage 128 KiB)
constexpr uint32_t MID_THREADS = 512, MID_KPT = 16, MID_TILE = MID_THREADS * MID_KPT;  // the smallest shape
constexpr uint32_t MID_MAX_TILES = 128;
constexpr uint32_t MID_MAX_KEYS = MID_MAX_TILES * MID_TILE;                            // 2^20: limit of the smallest shape
// scratch words in the handle's slab (SLAB_MID: its own region, zero whenever no mid-size sort is in flight)
constexpr uint32_t MID_ARRIVE = 0;                  // barrier counter: counts up during K1, K2 puts it back to zero
constexpr uint32_t MID_ROUTE = 32;                  // 0 = MSD route (K2 sorts the buckets), 1 = K1 did the LSD passes
constexpr uint32_t MID_BSTART = 64;                 // bucket starts [256]
constexpr uint32_t MID_BCOUNT = MID_BSTART + RADIX; // bucket counts [256]
constexpr uint32_t MID_TABLE = MID_BCOUNT + RADIX;  // two count tables [2][MID_MAX_TILES][256]
constexpr uint32_t MID_WORDS = MID_TABLE + 2 * MID_MAX_TILES * RADIX;
static_assert(MID_WORDS <= SLAB_MID_WORDS, "SLAB_MID is too small");

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) {
    return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
    return __hip_atomic_load(const_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int VB>
struct SC1T { using type = uint32_t; };
template <>
struct SC1T<8> { using type = unsigned long long; };

// Rank the tile's keys among the keys of their digit inside their wave (see digit_binning_kernel): off = the ranks (below
// 64 * KPT <= 2048: two per register, which keeps the 32-keys-per-thread shapes out of scratch), the per-wave counters keep the counts.  RANK 1: slots >= count take no part; RANK 0 (ballots): the all-ones dummy
// keys behind `count` rank last in digit 255.
template <int RANK, int KPT>
__device__ __forceinline__ void mid_rank(const uint32_t (&key)[KPT], uint32_t shift, uint32_t my_base, uint32_t count,
                                         uint32_t* whist, uint32_t (&off)[KPT / 2]) {
#pragma unroll
    for (int i = 0; i < KPT / 2; ++i) off[i] = 0;
    if constexpr (RANK == 0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            uint32_t acc_lo = 0, acc_hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t B = (uint32_t)__builtin_amdgcn_sbfe((int32_t)key[i], shift + k, 1);
                const unsigned long long b = __builtin_amdgcn_ballot_w64(B != 0u);
                acc_lo = __builtin_amdgcn_bitop3_b32(acc_lo, (uint32_t)b, B, 0xF6);
                acc_hi = __builtin_amdgcn_bitop3_b32(acc_hi, (uint32_t)(b >> 32), B, 0xF6);
            }
            const uint32_t plo = ~acc_lo, phi = ~acc_hi;
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t total = __popc(plo) + __popc(phi);
            const uint32_t pre = whist[d];
            if (below == total - 1u) whist[d] = pre + total;
            asm volatile("" ::: "memory");
            off[i >> 1] |= (pre + below) << (16 * (i & 1));
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            if (my_base + i * 64u < count)
                off[i >> 1] |= __hip_atomic_fetch_add(&whist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) << (16 * (i & 1));
        }
    }
}

// Grid barrier on a monotonic counter: arrive, then wait until `target` arrivals are in (wrap-safe).  One lane polls
// with sc1 loads; bounded: a workgroup that was never dispatched (foreign load on the device) must not hang the rest.
__device__ __forceinline__ bool mid_barrier(uint32_t* arrive, uint32_t target, uint32_t* status, uint32_t tid) {
    __shared__ uint32_t s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores are out before anyone is told
    __syncthreads();
    if (tid == 0) {
        atomicAdd(arrive, 1u);
        uint32_t spins = 0;
        bool ok = true;
        while ((int32_t)(ld_agent(arrive) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { ok = false; break; }
        }
        if (!ok) st_agent(status, STATUS_TIMEOUT);
        s_ok = ok ? 1u : 0u;
    }
    __syncthreads();
    return s_ok != 0u;
}

// ---------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------
template <int VB, int KT, int RANK, int THREADS_ = (int)MID_THREADS, int KPT_ = (int)MID_KPT>
__global__ __launch_bounds__(THREADS_) void mid_msd_kernel(uint32_t* keys, uint32_t* alt, void* vals_, void* valt_, uint32_t* scratch,
                                                           uint32_t* status, uint32_t n, uint32_t descending) {
    using V = typename ValT<VB>::type;
    using VA = typename SC1T<VB>::type;
    constexpr int KPT = KPT_, WAVES = THREADS_ / 64;
    constexpr uint32_t THREADS = THREADS_, TILE = THREADS_ * KPT_;
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[TILE];
    __shared__ __attribute__((aligned(16))) V s_vstage[VB != 0 ? TILE : 1];
    __shared__ uint32_t s_whist[WAVES * RADIX];
    __shared__ uint32_t s_dpre[RADIX], s_gbase[RADIX], s_wtot[4], s_max;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tile = blockIdx.x, tiles = gridDim.x;
    const uint32_t tile_base = tile * TILE;
    const uint32_t count = n - tile_base < TILE ? n - tile_base : TILE;  // valid slots [0, count)
    const uint32_t my_base = wave * (64u * KPT) + lane;
    uint32_t* whist = s_whist + wave * RADIX;
    uint32_t* arrive = scratch + MID_ARRIVE;
    uint32_t barrier_no = 0;

    uint32_t key[KPT];
    V val[VB != 0 ? KPT : 1];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = my_base + i * 64u;
        const uint32_t ci = tile_base + (slot < count ? slot : count - 1u);
        key[i] = keys[ci];
        if constexpr (VB != 0) val[i] = reinterpret_cast<const V*>(vals_)[ci];
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) key[i] = my_base + i * 64u < count ? to_bits<KT>(key[i]) : 0xffffffffu;

    // One partition step of the tile on the digit at `shift`: rank, tile counts -> table row, grid barrier, bases from
    // the table (digit starts + the tiles in front), stage.  Leaves s_gbase[d] = global position of stage slot 0 of
    // digit d's run minus its stage offset, and G (all tiles' count of this thread's digit) in the return value.
    uint32_t off[KPT / 2];
    auto partition_step = [&](uint32_t shift, uint32_t* table, bool& alive) -> uint32_t {
        for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
        if (tid == 0) s_max = 0;
        __syncthreads();
        mid_rank<RANK, KPT>(key, shift, my_base, count, whist, off);
        __syncthreads();
        uint32_t run = 0, scan_incl = 0, mine = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = run;
                run += c;
            }
            mine = run - ((RANK == 0 && tid == RADIX - 1) ? TILE - count : 0u);  // real keys only
            st_sc1(&table[tile * RADIX + tid], mine);
            scan_incl = wave_inclusive_scan_dpp(run);
            if (lane == 63) s_wtot[wave] = scan_incl;
        }
        alive = mid_barrier(arrive, (++barrier_no) * tiles, status, tid);
        uint32_t G = 0;
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_wtot[w];
            const uint32_t dpre = wbase + scan_incl - run;  // stage offset of the digit's run (dummies included)
            s_dpre[tid] = dpre;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += dpre;
            // every tile's count of this digit: all of them for the digit's start, the tiles in front for this tile's offset
            uint32_t front = 0;
            for (uint32_t t0 = 0; t0 < tiles; t0 += 8) {
                uint32_t c[8];
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) {
                    const uint32_t t = t0 + j < tiles ? t0 + j : tiles - 1u;
                    c[j] = ld_sc1(&table[t * RADIX + tid]);
                }
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j)
                    if (t0 + j < tiles) {
                        G += c[j];
                        if (t0 + j < tile) front += c[j];
                    }
            }
            atomicMax(&s_max, G);
            scan_incl = wave_inclusive_scan_dpp(G);
            s_gbase[tid] = front - dpre;  // + digit start, below
        }
        __syncthreads();
        if (tid < RADIX && lane == 63) s_wtot[wave] = scan_incl;
        __syncthreads();
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_wtot[w];
            s_gbase[tid] += wbase + scan_incl - G;  // digit start = keys with a smaller digit, in all tiles
        }
        __syncthreads();
        return G;
    };
    auto stage = [&](uint32_t shift) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t lpos = ((off[i >> 1] >> (16 * (i & 1))) & 0xffffu) + s_whist[wave * RADIX + ((key[i] >> shift) & 255u)];
            if (RANK == 0 || my_base + i * 64u < count) {
                s_stage[lpos] = key[i];
                if constexpr (VB != 0) s_vstage[lpos] = val[i];
            }
        }
        __syncthreads();
    };

    // ---- MSD step on the top byte ----
    bool alive = true;
    uint32_t* table0 = scratch + MID_TABLE;
    uint32_t* table1 = table0 + MID_MAX_TILES * RADIX;
    const uint32_t G = partition_step(24, table0, alive);
    if (!alive) return;
    const bool lsd_route = s_max > TILE;  // a bucket K2 could not hold; the same table everywhere: the same decision everywhere
    __syncthreads();  // everybody has read s_max before the next partition step resets it
    if (tile == 0 && tid < RADIX) {
        scratch[MID_BSTART + tid] = s_gbase[tid] + s_dpre[tid];  // tile 0 has no tile in front: its base IS the digit start
        scratch[MID_BCOUNT + tid] = G;
        if (tid == 0) scratch[MID_ROUTE] = lsd_route ? 1u : 0u;
    }
    if (!lsd_route) {
        stage(24);
        // stable scatter into the alt buffer: stage slot j of digit d's run -> s_gbase[d] + j
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t i = tid + j * THREADS;
            if (i < count) {
                const uint32_t kb = s_stage[i];
                const uint32_t o = s_gbase[kb >> 24] + i;
                alt[o] = from_bits<KT>(kb);
                if constexpr (VB != 0) reinterpret_cast<V*>(valt_)[o] = s_vstage[i];
            }
        }
        return;
    }

    // ---- LSD route: the four passes here, grid barriers between them; keys cross workgroups through sc1 accesses ----
    uint32_t* kbuf[2] = {keys, alt};
    void* vbuf[2] = {vals_, valt_};
#pragma unroll 1
    for (uint32_t p = 0; p < 4; ++p) {
        const uint32_t shift = p * 8u;
        if (p != 0) {  // reload this tile from where the previous pass wrote it
            const uint32_t* kin = kbuf[p & 1u];
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t slot = my_base + i * 64u;
                const uint32_t ci = tile_base + (slot < count ? slot : count - 1u);
                key[i] = ld_sc1(&kin[ci]);
                if constexpr (VB != 0) val[i] = (V)ld_sc1(&reinterpret_cast<const VA*>(vbuf[p & 1u])[ci]);
            }
#pragma unroll
            for (int i = 0; i < KPT; ++i) key[i] = my_base + i * 64u < count ? to_bits<KT>(key[i]) : 0xffffffffu;
        }
        partition_step(shift, (p & 1u) ? table0 : table1, alive);
        if (!alive) return;
        stage(shift);
        uint32_t* kout = kbuf[(p + 1u) & 1u];
        const bool reverse = descending && p == 3;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t i = tid + j * THREADS;
            if (i < count) {
                const uint32_t kb = s_stage[i];
                uint32_t o = s_gbase[(kb >> shift) & 255u] + i;
                if (reverse) o = n - 1u - o;
                st_sc1(&kout[o], from_bits<KT>(kb));
                if constexpr (VB != 0) st_sc1(&reinterpret_cast<VA*>(vbuf[(p + 1u) & 1u])[o], (VA)s_vstage[i]);
            }
        }
        if (p != 3) {  // everybody has written pass p's output before anybody reads it
            if (!mid_barrier(arrive, (++barrier_no) * tiles, status, tid)) return;
        }
    }
}

// ---------------------------------------------------------------------------
// K2: one workgroup per top-byte bucket; the low 24 bits in three stable LDS passes
// ---------------------------------------------------------------------------
template <int VB, int KT, int RANK, int THREADS_ = (int)MID_THREADS, int KPT_ = (int)MID_KPT>
__global__ __launch_bounds__(THREADS_) void bucket_sort_kernel(uint32_t* keys, const uint32_t* alt, void* vals_, const void* valt_,
                                                               uint32_t* scratch, uint32_t n, uint32_t descending) {
    using V = typename ValT<VB>::type;
    constexpr int KPT = KPT_, WAVES = THREADS_ / 64;
    constexpr uint32_t THREADS = THREADS_, TILE = THREADS_ * KPT_;
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[TILE];
    __shared__ __attribute__((aligned(16))) V s_vstage[VB != 0 ? TILE : 1];
    __shared__ uint32_t s_whist[WAVES * RADIX];
    __shared__ uint32_t s_wtot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (blockIdx.x == 0 && tid == 0) scratch[MID_ARRIVE] = 0u;  // K1 is over: the barrier counter is idle again
    if (scratch[MID_ROUTE] != 0u) return;  // K1 ran the LSD passes itself
    const uint32_t start = scratch[MID_BSTART + blockIdx.x], count = scratch[MID_BCOUNT + blockIdx.x];
    if (count == 0u) return;
    const uint32_t my_base = wave * (64u * KPT) + lane;
    uint32_t* whist = s_whist + wave * RADIX;
    uint32_t key[KPT];
    V val[VB != 0 ? KPT : 1];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = my_base + i * 64u;
        const uint32_t ci = start + (slot < count ? slot : count - 1u);
        key[i] = alt[ci];
        if constexpr (VB != 0) val[i] = reinterpret_cast<const V*>(valt_)[ci];
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) key[i] = my_base + i * 64u < count ? to_bits<KT>(key[i]) : 0xffffffffu;
    // a bucket of 64 keys or fewer could stop earlier; the three passes on LDS cost a few microseconds at any size
#pragma unroll 1
    for (uint32_t shift = 0; shift < 24; shift += 8) {
        for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
        __syncthreads();
        uint32_t off[KPT / 2];
        mid_rank<RANK, KPT>(key, shift, my_base, count, whist, off);
        __syncthreads();
        uint32_t run = 0, scan_incl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = run;
                run += c;
            }
            scan_incl = wave_inclusive_scan_dpp(run);
            if (lane == 63) s_wtot[wave] = scan_incl;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_wtot[w];
            const uint32_t dpre = wbase + scan_incl - run;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += dpre;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t lpos = ((off[i >> 1] >> (16 * (i & 1))) & 0xffffu) + s_whist[wave * RADIX + ((key[i] >> shift) & 255u)];
            if (RANK == 0 || my_base + i * 64u < count) {
                s_stage[lpos] = key[i];
                if constexpr (VB != 0) s_vstage[lpos] = val[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            key[i] = s_stage[my_base + i * 64u];
            if constexpr (VB != 0) val[i] = s_vstage[my_base + i * 64u];
        }
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = my_base + i * 64u;
        if (slot < count) {
            const uint32_t idx = start + slot;
            const uint32_t o = descending ? n - 1u - idx : idx;
            keys[o] = from_bits<KT>(key[i]);
            if constexpr (VB != 0) reinterpret_cast<V*>(vals_)[o] = val[i];
        }
    }
}

}  // namespace gs
