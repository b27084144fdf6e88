// ls_kernels.hpp — the LOCAL-SORT plan of the 8-bit LSD sort (gfx950, wave64): 32 bytes per key instead of 36.
//
// Reference behaviour reproduced (what, not how): b0nes164/GPUSorting GPUSortingCUDA/Sort/OneSweep.cu:44-123
// (GlobalHistogram), :125-162 (Scan), :164-344 (DigitBinningPassKeysOnly): four stable 8-bit partition passes, least
// significant digit first.  The result is bit-identical; the work is cut differently:
//
//   ls_first_kernel   replaces GlobalHistogram + Scan + the first DigitBinningPass.  Every 16 384-key tile is ranked by
//                     digit 0 and written back TILE-LOCALLY sorted to the same place of the other buffer — sequential
//                     16-byte stores, no descriptors, no look-back — together with a run table R[digit][tile] =
//                     (start of the digit's run inside the tile, its length).  The array in the virtual order
//                     (digit 0, tile, position in the run) IS the output of a stable first pass; it is never materialised.
//   ls_plan_kernel    transposes the run table to R[digit][tile], sums the workgroups' table slices into CNEXT[1] and the digit-0
//                     totals, and writes the plan words: chain geometry of the gather pass, whether the two upper passes are identity
//                     permutations (bytes 2 and 3 constant: dropped as a pair).
//   ls_runscan_kernel E[digit][tile] = exclusive prefix of a digit's run lengths over the tiles (positions in the virtual order),
//                     S = the first run every 16 384-key VIRTUAL tile of every digit touches.
//   ls_pass_kernel    GATHER form (second pass): tile j of digit d = virtual positions [16 384 j, 16 384 (j + 1)) of that digit's keys
//                     (exact tiles, one partial tile per digit), i.e. ~256 runs of ~64 keys at 4-byte-aligned offsets of consecutive
//                     source tiles, found through S / E / R and loaded run by run straight into the LDS stage; tiles in
//                     (digit, tile) order are the virtual order, so the chained scan over them (16 chains = 16 groups of digit-0
//                     values) is the stable second pass.  LINEAR form (third and fourth pass): the usual tile over the pass's input.
//   no histogram sweep: every pass counts the NEXT pass's joint table H[next digit][group of this digit] while its keys are in
//                     registers (16-bit packed LDS counters, flushed into the workgroup's own SLICE in global memory — no atomics —
//                     when it runs out of tiles or a counter nears overflow); ls_reduce_kernel (ls_plan_kernel behind the first
//                     kernel) sums the slices, and the workgroups of the next pass derive digit starts, chain geometry and chain
//                     seeds from the table.
// Status (profiles/r04_ls_plan_status.txt): bit-exact and SLOWER than the default pipeline (the gather pass pays three dependent
// global round trips per tile); opt-in through gs_onesweep_options::plan.
//
// Descriptors, flags, bounded spins, POISON, fallback recount: as in onesweep_kernels.hpp (same words, same meaning).
#pragma once
#include "onesweep_kernels.hpp"

#ifndef GS_LS_CLOCK
#define GS_LS_CLOCK 0  // 1 (tuning builds, tools/r04_ls_clock.py): per-phase shader clocks of one workgroup — costs registers, never in the product
#endif

namespace gs {

constexpr uint32_t LS_THREADS = 512, LS_KPT = 32, LS_WAVES = LS_THREADS / 64, LS_TILE = LS_THREADS * LS_KPT;  // 16 384 keys
constexpr uint32_t LS_WH_WORDS = LS_WAVES * (RADIX / 2);  // per-wave digit counters, two 16-bit counters per word (a wave holds 2048 keys)
constexpr uint32_t LS_TAB_WORDS = NCH * RADIX / 2;        // next-digit joint table, two 16-bit counters per word
static_assert(NCH == 16 && LS_TILE == 16384, "the local-sort plan is written for 16 chains and 16 384-key tiles");

// words of the SLAB_LS region
constexpr uint32_t LS_TOT0 = 0;     // [256] keys per digit-0 value
constexpr uint32_t LS_OR = 256;     // OR of all keys (sortable form)
constexpr uint32_t LS_NAND = 257;   // OR of all complemented keys
constexpr uint32_t LS_DONE = 258;   // summing workgroups of ls_plan_kernel that are through (zeroed by the first kernel)
constexpr uint32_t LS_TICKET = 259;  // tile tickets of the first kernel (zero between calls: ls_plan_kernel hands it back zeroed)
constexpr uint32_t LS_FLAGS = 260;  // [4] PF_* of passes 0..3 (written by the first kernel's last workgroup)
constexpr uint32_t LS_SB = 264;     // [256] gather pass: first word of digit d in the S table (one word per tile of the digit + 1)
constexpr uint32_t LS_UU0 = 520;    // [256] tiles of d's chain in front of digit d
constexpr uint32_t LS_CU = 776;     // [16] tiles per chain
static_assert(LS_CU + NCH <= 896, "SLAB_LS layout (words 900.. : tuning aids)");
// descriptor rows the gather pass can need: sum over digits of ceil(tot_d / TILE) <= nt + 256, + row 0 of every chain
__host__ __device__ __forceinline__ uint32_t ls_gather_rows(uint32_t nt) { return nt + RADIX + NCH + 8u; }
// words of the S table: one per tile of every digit + one closing word per digit
__host__ __device__ __forceinline__ uint32_t ls_stab_words(uint32_t nt) { return nt + 2u * RADIX + 8u; }
__host__ __device__ __forceinline__ uint32_t ls_linear_rows(uint32_t nt) { return nt + 2u * NCH + 8u; }

__device__ __forceinline__ uint32_t pk16(uint32_t word, uint32_t d) { return (word >> ((d & 1u) << 4)) & 0xffffu; }

// ---------------------------------------------------------------------------------------------------------------------
// Shared tile machinery (512 threads, 32 keys per thread, wave-striped: wave w item i lane l <-> tile position w * 2048 +
// i * 64 + l, which keeps tile order == array order, i.e. the pass stable).
// ---------------------------------------------------------------------------------------------------------------------
struct LsTile {
    uint32_t* s_stage;  // [LS_TILE]
    uint32_t* s_whist;  // [LS_WH_WORDS]
    uint32_t* s_tab;    // [LS_TAB_WORDS] (counting kernels)
    uint32_t* s_misc;   // [64]
};

// rank the valid keys of the tile by digit (key >> shift) & 255: offp[i >> 1] gets the 16-bit rank of key i among the keys of
// its digit in this wave; COUNT: also adds the key to the next-digit joint table.  vlo / vhi: valid tile positions [vlo, vhi).
template <bool COUNT>
__device__ __forceinline__ void ls_rank(const LsTile& T, const uint32_t (&key)[LS_KPT], uint32_t (&offp)[LS_KPT / 2], uint32_t shift,
                                        bool full, uint32_t vlo, uint32_t vhi, uint32_t wave, uint32_t lane) {
    uint32_t* whist = T.s_whist + wave * (RADIX / 2);
#pragma unroll
    for (int i = 0; i < (int)LS_KPT / 2; ++i) offp[i] = 0;
    uint32_t p0 = wave * (64u * LS_KPT) + lane;
    asm volatile("" : "+v"(p0));  // (not hoisted out of the caller's persistent loop as 32 registers)
    // (wave-uniform: a wave whose 2048 positions are all valid ranks without predicates — the units of the gather pass are ~97 % full)
    {   // (made scalar explicitly: the compiler cannot know that wave and the bounds are the same for all lanes)
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave * (64u * LS_KPT)));
        full = full || (w0 >= (uint32_t)__builtin_amdgcn_readfirstlane((int)vlo) && w0 + 64u * LS_KPT <= (uint32_t)__builtin_amdgcn_readfirstlane((int)vhi));
    }
    if (GS_LIKELY(full)) {
#pragma unroll
        for (int i = 0; i < (int)LS_KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            const uint32_t sh = (d & 1u) << 4;
            const uint32_t r = __hip_atomic_fetch_add(&whist[d >> 1], 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            offp[i >> 1] |= ((r >> sh) & 0xffffu) << (16 * (i & 1));
            if constexpr (COUNT) {
                const uint32_t b = (key[i] >> (shift + 4u)) & 0xfffu;  // (next digit << 4) | group of this digit
                __hip_atomic_fetch_add(&T.s_tab[b >> 1], 1u << ((b & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < (int)LS_KPT; ++i) {
            const uint32_t p = p0 + i * 64u;
            if (p >= vlo && p < vhi) {
                const uint32_t d = (key[i] >> shift) & 255u;
                const uint32_t sh = (d & 1u) << 4;
                const uint32_t r = __hip_atomic_fetch_add(&whist[d >> 1], 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                offp[i >> 1] |= ((r >> sh) & 0xffffu) << (16 * (i & 1));
                if constexpr (COUNT) {
                    const uint32_t b = (key[i] >> (shift + 4u)) & 0xfffu;
                    __hip_atomic_fetch_add(&T.s_tab[b >> 1], 1u << ((b & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
}

// adds the tile's valid keys to the next-digit joint table (a loop of its own behind the staging: inside the ranking loop the two
// atomics per key kept 48 registers of keys and ranks live next to each other's temporaries, and the gather pass spilled ranks)
__device__ __forceinline__ void ls_count(const LsTile& T, const uint32_t (&key)[LS_KPT], uint32_t shift, bool full, uint32_t vlo, uint32_t vhi,
                                         uint32_t wave, uint32_t lane) {
    uint32_t p0 = wave * (64u * LS_KPT) + lane;
    asm volatile("" : "+v"(p0));
    {
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave * (64u * LS_KPT)));
        full = full || (w0 >= (uint32_t)__builtin_amdgcn_readfirstlane((int)vlo) && w0 + 64u * LS_KPT <= (uint32_t)__builtin_amdgcn_readfirstlane((int)vhi));
    }
    if (GS_LIKELY(full)) {
#pragma unroll
        for (int i = 0; i < (int)LS_KPT; ++i) {
            const uint32_t b = (key[i] >> (shift + 4u)) & 0xfffu;  // (next digit << 4) | group of this digit
            __hip_atomic_fetch_add(&T.s_tab[b >> 1], 1u << ((b & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
#pragma unroll
        for (int i = 0; i < (int)LS_KPT; ++i) {
            const uint32_t p = p0 + i * 64u;
            if (p >= vlo && p < vhi) {
                const uint32_t b = (key[i] >> (shift + 4u)) & 0xfffu;
                __hip_atomic_fetch_add(&T.s_tab[b >> 1], 1u << ((b & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// After the ranking barrier: threads k < 128 own the digit pair (2k, 2k + 1).  Turns the per-wave counters into (exclusive prefix
// over the waves + tile-local start of the digit's run), returns the pair's counts and run starts.
// Contains two barriers; every thread of the workgroup must call it.
// rot != 0 (uniform; the first kernel): the runs are laid out in the ROTATED digit order rot, rot + 1, .., 255, 0, .., rot - 1 inside the
// tile (total = the tile's keys) — one more barrier.
__device__ __forceinline__ void ls_digit_scan(const LsTile& T, uint32_t tid, uint32_t& c0, uint32_t& c1, uint32_t& dpre0, uint32_t& dpre1,
                                              uint32_t rot = 0u, uint32_t total = 0u) {
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t run = 0, incl = 0;
    c0 = c1 = dpre0 = dpre1 = 0;
    if (tid < RADIX / 2) {
#pragma unroll
        for (uint32_t w = 0; w < LS_WAVES; ++w) {  // packed: both halves stay below 2^15 (a tile holds 16 384 keys)
            const uint32_t c = T.s_whist[w * (RADIX / 2) + tid];
            T.s_whist[w * (RADIX / 2) + tid] = run;
            run += c;
        }
        c0 = run & 0xffffu;
        c1 = run >> 16;
        incl = wave_inclusive_scan_dpp(c0 + c1);
        if (lane == 63) T.s_misc[4 + wave] = incl;
    }
    __syncthreads();
    if (tid < RADIX / 2) {
        const uint32_t wbase = wave ? T.s_misc[4] : 0u;
        dpre0 = wbase + incl - (c0 + c1);
        dpre1 = dpre0 + c0;
        if (rot != 0u && (rot >> 1) == tid) T.s_misc[10] = (rot & 1u) ? dpre1 : dpre0;
    }
    if (rot != 0u) {  // uniform
        __syncthreads();
        if (tid < RADIX / 2) {
            const uint32_t first = T.s_misc[10];  // natural start of digit rot's run = keys of the digits below rot
            dpre0 = 2u * tid >= rot ? dpre0 - first : dpre0 + (total - first);
            dpre1 = 2u * tid + 1u >= rot ? dpre1 - first : dpre1 + (total - first);
        }
    }
    if (tid < RADIX / 2) {
        const uint32_t add = dpre0 | (dpre1 << 16);  // (start + prefix <= 16 384: no carry between the halves)
#pragma unroll
        for (uint32_t w = 0; w < LS_WAVES; ++w) T.s_whist[w * (RADIX / 2) + tid] += add;
    }
    __syncthreads();
}

// stage the valid keys at their tile-local sorted positions
__device__ __forceinline__ void ls_stage(const LsTile& T, const uint32_t (&key)[LS_KPT], const uint32_t (&offp)[LS_KPT / 2], uint32_t shift,
                                         bool full, uint32_t vlo, uint32_t vhi, uint32_t wave, uint32_t lane) {
    const uint32_t* whist = T.s_whist + wave * (RADIX / 2);
    uint32_t p0 = wave * (64u * LS_KPT) + lane;
    asm volatile("" : "+v"(p0));
    {
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave * (64u * LS_KPT)));
        full = full || (w0 >= (uint32_t)__builtin_amdgcn_readfirstlane((int)vlo) && w0 + 64u * LS_KPT <= (uint32_t)__builtin_amdgcn_readfirstlane((int)vhi));
    }
    if (GS_LIKELY(full)) {
#pragma unroll
        for (int i = 0; i < (int)LS_KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            const uint32_t lpos = ((offp[i >> 1] >> (16 * (i & 1))) & 0xffffu) + pk16(whist[d >> 1], d);
            T.s_stage[lpos] = key[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < (int)LS_KPT; ++i) {
            const uint32_t p = p0 + i * 64u;
            if (p >= vlo && p < vhi) {
                const uint32_t d = (key[i] >> shift) & 255u;
                const uint32_t lpos = ((offp[i >> 1] >> (16 * (i & 1))) & 0xffffu) + pk16(whist[d >> 1], d);
                T.s_stage[lpos] = key[i];
            }
        }
    }
}

// next-digit table: is a 16-bit counter past 2^15 (one more tile could overflow it)?  Thread tid looks at its four words.
__device__ __forceinline__ bool ls_tab_near_overflow(const uint32_t* s_tab, uint32_t tid) {
    const uint4 v = reinterpret_cast<const uint4*>(s_tab)[tid];
    return ((v.x | v.y | v.z | v.w) & 0x80008000u) != 0u;
}
// adds the workgroup's table to its SLICE (the workgroup's own 4096 words in global memory: plain read-modify-write, summed over
// the workgroups by ls_reduce_kernel / ls_plan_kernel) and clears it; thread tid owns words 4 tid .. 4 tid + 3 = bins 8 tid .. 8 tid + 7.
// (Round 4, first form: one global atomic per non-empty bin on CNEXT itself — 512 workgroups x 4096 atomics on the same 4096 words
//  serialise: the first kernel's last workgroup was through 220 us after its first, profiles/r04_ls_first_kernel_timeline.txt.)
__device__ __forceinline__ void ls_tab_flush(uint32_t* s_tab, uint32_t* slice, uint32_t tid) {
    const uint4 v = reinterpret_cast<const uint4*>(s_tab)[tid];
    uint4 a = reinterpret_cast<const uint4*>(slice)[2u * tid], b = reinterpret_cast<const uint4*>(slice)[2u * tid + 1u];
    a.x += v.x & 0xffffu; a.y += v.x >> 16; a.z += v.y & 0xffffu; a.w += v.y >> 16;
    b.x += v.z & 0xffffu; b.y += v.z >> 16; b.z += v.w & 0xffffu; b.w += v.w >> 16;
    reinterpret_cast<uint4*>(slice)[2u * tid] = a;
    reinterpret_cast<uint4*>(slice)[2u * tid + 1u] = b;
    reinterpret_cast<uint4*>(s_tab)[tid] = uint4{0u, 0u, 0u, 0u};
}
// a workgroup's slice: [0, 4096) next-digit table, [4096, 4352) digit-0 totals, [4352] OR of the keys, [4353] OR of their complements (first kernel)
constexpr uint32_t LS_SLICE_TOT = NCH * RADIX, LS_SLICE_OR = LS_SLICE_TOT + RADIX, LS_SLICE_WORDS = LS_SLICE_OR + 256;  // 4608: 72 x 64
static_assert(LS_SLICE_WORDS % 64 == 0, "the reduce kernels take 64 words per workgroup");

// ---------------------------------------------------------------------------------------------------------------------
// First kernel: tile-local sort by digit 0, run table, next-digit table, digit-0 totals, OR / AND of the keys, the plan.
// Persistent: workgroup b sorts tiles b, b + grid, ... (no tickets: tiles do not depend on each other).  Also the sort's
// CLEAR (reference: ClearMemory, OneSweepDispatcher.cuh:301-309): ticket counters, status, CNEXT[2..3], descriptor rows.
// ---------------------------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(LS_THREADS, 4) void ls_first_kernel(const uint32_t* __restrict__ keys_in, uint32_t* __restrict__ keys_out,
                                                                  uint32_t* __restrict__ runs_t /*[tile][256]: transposed by ls_plan_kernel*/,
                                                                  uint32_t* __restrict__ slices /*[grid][LS_SLICE_WORDS]*/,
                                                                  uint32_t* slab, size_t zero_end_words, uint32_t n,
                                                                  uint32_t plan /*bit1: identity passes may be dropped*/) {
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[LS_TILE];
    __shared__ __attribute__((aligned(16))) uint32_t s_whist[LS_WH_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[LS_TAB_WORDS];
    __shared__ uint32_t s_misc[64];
    const LsTile T{s_stage, s_whist, s_tab, s_misc};
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* ls = slab + SLAB_LS;
    uint32_t* slice = slices + (size_t)blockIdx.x * LS_SLICE_WORDS;  // the gather pass's table [digit 1][group of digit 0], as counted here
    const uint32_t nt = (n + LS_TILE - 1u) / LS_TILE;
    const uint32_t p0 = wave * (64u * LS_KPT) + lane;
    const unsigned long long wall_entry = wall_clock64();
    // the keys of tile t as they lie in memory; a partial tile reads its last key again behind its end (unconditional loads on a
    // clamped index: guarded loads are issued one at a time)
    uint32_t raw[LS_KPT];
    auto load_tile = [&](uint32_t t) {
        const uint32_t base = t * LS_TILE;
        uint32_t p0 = wave * (64u * LS_KPT) + lane;
        asm volatile("" : "+v"(p0));  // (not hoisted out of the persistent loop as 32 registers)
        if (GS_LIKELY(n - base >= LS_TILE)) {
#pragma unroll
            for (int i = 0; i < (int)LS_KPT; ++i) raw[i] = __builtin_nontemporal_load(keys_in + base + p0 + i * 64u);
        } else {
            const uint32_t count = n - base;
#pragma unroll
            for (int i = 0; i < (int)LS_KPT; ++i) {
                const uint32_t p = p0 + i * 64u;
                raw[i] = keys_in[base + (p < count ? p : count - 1u)];
            }
        }
    };
    if (blockIdx.x < nt) load_tile(blockIdx.x);
    {   // the clear: [0, SLAB_HIST) counters, status, info; CNEXT[2], CNEXT[3]; the descriptor rows
        uint4* a = reinterpret_cast<uint4*>(slab);
        const uint4 z = {0u, 0u, 0u, 0u};
        const size_t stride = (size_t)gridDim.x * LS_THREADS, first = (size_t)blockIdx.x * LS_THREADS + tid;
        for (size_t i = first; i < SLAB_HIST / 4; i += stride) a[i] = z;
        for (size_t i = (SLAB_HSUB + 2u * HSUB_STRIDE) / 4 + first; i < (SLAB_HSUB + 4u * HSUB_STRIDE) / 4; i += stride) a[i] = z;
        for (size_t i = SLAB_DESC / 4 + first; i < zero_end_words / 4; i += stride) a[i] = z;
        for (uint32_t i = tid; i < LS_SLICE_WORDS / 4; i += LS_THREADS) reinterpret_cast<uint4*>(slice)[i] = z;
        if (blockIdx.x == 0 && tid == 0) ls[LS_DONE] = 0u;  // (the arrival counter of ls_plan_kernel)
    }
    reinterpret_cast<uint4*>(s_tab)[tid] = uint4{0u, 0u, 0u, 0u};
    uint32_t tot0 = 0, tot1 = 0;        // threads < 128: keys of digits 2 tid, 2 tid + 1 in this workgroup's tiles
    uint32_t kor = 0, knand = 0;        // OR of the keys / of their complements
#if GS_LS_CLOCK  // tuning build: shader clocks per phase of workgroup 7's tiles, summed by its thread 0 into ls[900..]
    const bool clk_on = blockIdx.x == 7u && tid == 0u;
    uint32_t clk_last = 0;
    auto stamp = [&](int k) { if (clk_on) { const uint32_t c = (uint32_t)__builtin_readcyclecounter(); s_misc[48 + k] += c - clk_last; clk_last = c; } };
    unsigned long long wall0 = 0, cyc0 = 0;
    if (clk_on) { for (int k = 0; k < 12; ++k) s_misc[48 + k] = 0; cyc0 = __builtin_readcyclecounter(); clk_last = (uint32_t)cyc0; wall0 = wall_clock64(); }
#else
    auto stamp = [](int) {};
#endif
    // Tiles: the first one is blockIdx.x, every further one a ticket — the two workgroups of a CU do not run at the same speed (the
    // second-dispatched one loses the issue arbitration: with tiles b, b + grid, ... half of the workgroups were through after 331 ..
    // 375 us, the other half after 403 .. 475, profiles/r04_ls_first_kernel_timeline.txt).  The next tile's ticket is drawn at the top of a
    // tile, so that it is known when its loads are issued (behind the staging).
    uint32_t t = blockIdx.x, t_next = 0xffffffffu;
#pragma unroll 1
    for (; t < nt; t = t_next) {
        if (tid == 0) s_misc[1] = gridDim.x + atomicAdd(&ls[LS_TICKET], 1u);
        __syncthreads();  // the last tile's readers of the stage and of the counters are through
        stamp(0);
        s_whist[tid] = 0;
        s_whist[tid + LS_THREADS] = 0;
        const uint32_t base = t * LS_TILE;
        const uint32_t count = n - base < LS_TILE ? n - base : LS_TILE;
        const bool full = count == LS_TILE;
        uint32_t key[LS_KPT];
#pragma unroll
        for (int i = 0; i < (int)LS_KPT; ++i) key[i] = to_bits<KT>(raw[i]);
        if (GS_LIKELY(full)) {
#pragma unroll
            for (int i = 0; i < (int)LS_KPT; ++i) { kor |= key[i]; knand |= ~key[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < (int)LS_KPT; ++i)
                if (p0 + i * 64u < count) { kor |= key[i]; knand |= ~key[i]; }
        }
        __syncthreads();  // counters are zero
        stamp(1);  // waited for the keys
        uint32_t offp[LS_KPT / 2];
        ls_rank<true>(T, key, offp, 0u, full, 0u, count, wave, lane);
        stamp(2);
        __syncthreads();
        stamp(3);
        uint32_t c0, c1, dpre0, dpre1;
        // The tile's runs are laid out in a digit order rotated by the tile's index: the same digit's runs of consecutive tiles then sit
        // 65 536 - 256 bytes apart (uniform keys) instead of 65 536, and the ~256 run reads of one tile of the gather pass spread over the
        // memory channels instead of queueing on one or two of them (profiles/r04_ls_rotated_runs.txt).  plan bits 8..15: the step.
        const uint32_t rot = (t * ((plan >> 8) & 255u)) & 255u;
        ls_digit_scan(T, tid, c0, c1, dpre0, dpre1, rot, count);
        stamp(4);
        if (tid < RADIX / 2) {  // the run table, one coalesced row per tile: (start inside the tile) << 16 | length
            reinterpret_cast<uint2*>(runs_t + (size_t)t * RADIX)[tid] = uint2{(dpre0 << 16) | c0, (dpre1 << 16) | c1};
            tot0 += c0;
            tot1 += c1;
        }
        ls_stage(T, key, offp, 0u, full, 0u, count, wave, lane);
        // the key registers are dead: the next tile's loads fly while this one is written out (the workgroup's per-tile chain —
        // load latency, rank, scan, stage, store — is what bounds the kernel at two workgroups per CU, not the memory system:
        // 0.36 ms for the 0.15 ms read with every store removed, profiles/r04_ls_first_kernel_ablation.txt)
        stamp(5);
        t_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_misc[1]);  // (written before this tile's first barrier)
        if (t_next < nt) load_tile(t_next);
        const bool near = ls_tab_near_overflow(s_tab, tid);
        const int flush = __syncthreads_or(near ? 1 : 0);  // (also: the stage is complete)
        stamp(6);
        if (GS_LIKELY(full)) {
#pragma unroll
            for (int j = 0; j < (int)LS_KPT / 4; ++j) {
                uint4 v = reinterpret_cast<const uint4*>(s_stage)[tid + j * LS_THREADS];
                v.x = from_bits<KT>(v.x); v.y = from_bits<KT>(v.y); v.z = from_bits<KT>(v.z); v.w = from_bits<KT>(v.w);
                reinterpret_cast<uint4*>(keys_out + base)[tid + j * LS_THREADS] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < (int)LS_KPT; ++j) {
                const uint32_t i = tid + j * LS_THREADS;
                if (i < count) keys_out[base + i] = from_bits<KT>(s_stage[i]);
            }
        }
        if (GS_UNLIKELY(flush)) ls_tab_flush(s_tab, slice, tid);  // (nobody adds before the next tile's barriers)
        stamp(7);
#if GS_LS_CLOCK
        if (clk_on) ++s_misc[58];
#endif
    }
#if GS_LS_CLOCK
    if (clk_on) {
        for (int k = 0; k < 8; ++k) ls[900 + k] = s_misc[48 + k] / (s_misc[58] ? s_misc[58] : 1u);
        ls[908] = s_misc[58];
        ls[909] = (uint32_t)(__builtin_readcyclecounter() - cyc0);   // shader clocks ...
        ls[910] = (uint32_t)(wall_clock64() - wall0);                // ... per 100 MHz ticks of the same interval
    }
#endif
    // ---- hand over: the workgroup's slice (table, digit totals, OR / AND); ls_plan_kernel sums the slices ----
    __syncthreads();
    ls_tab_flush(s_tab, slice, tid);
    if (tid < RADIX / 2) reinterpret_cast<uint2*>(slice + LS_SLICE_TOT)[tid] = uint2{tot0, tot1};
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) {
        kor |= __shfl_xor(kor, dd, 64);
        knand |= __shfl_xor(knand, dd, 64);
    }
    if (lane == 0) { s_misc[16 + wave] = kor; s_misc[24 + wave] = knand; }
    __syncthreads();
    if (tid == 0) {
        uint32_t o = 0, a = 0;
        for (uint32_t w = 0; w < LS_WAVES; ++w) { o |= s_misc[16 + w]; a |= s_misc[24 + w]; }
        slice[LS_SLICE_OR] = o;
        slice[LS_SLICE_OR + 1] = a;
    }
    if ((plan & 0x10000u) && tid == 0u) {  // tuning aid: when this workgroup started and ended (100 MHz wall clock), in its slice
        slice[LS_SLICE_OR + 8] = (uint32_t)wall_entry;
        slice[LS_SLICE_OR + 9] = (uint32_t)wall_clock64();
    }
}

// Sum of the workgroups' slices, 64 words per workgroup of 1024 threads: thread (g, b) = (tid / 64, tid % 64) sums every 16th slice of
// word b with all its loads in flight at once (512 slices: 32 loads, ONE round trip; four groups of 128 slices each took 16 dependent
// rounds, 80 us for the launch, profiles/r04_ls_first_kernel_timeline.txt); the 16 partial sums meet in LDS.
constexpr uint32_t LS_RED_THREADS = 1024;
__device__ __forceinline__ uint32_t ls_slice_sum(const uint32_t* __restrict__ slices, uint32_t nslices, uint32_t word, bool use_or, uint32_t* s_part /*[1024]*/) {
    const uint32_t b = threadIdx.x & 63u, g = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (uint32_t w0 = g; w0 < nslices; w0 += 16u * 32u) {
        uint32_t v[32];
#pragma unroll
        for (uint32_t j = 0; j < 32; ++j) {
            const uint32_t w = w0 + 16u * j;
            v[j] = w < nslices ? slices[(size_t)w * LS_SLICE_WORDS + word] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < 32; ++j) acc = use_or ? (acc | v[j]) : (acc + v[j]);
    }
    s_part[g * 64u + b] = acc;
    __syncthreads();
    uint32_t r = 0;
    if (threadIdx.x < 64u)
        for (uint32_t k = 0; k < 16u; ++k) r = use_or ? (r | s_part[k * 64u + b]) : (r + s_part[k * 64u + b]);
    return r;
}
// behind a counting pass: CNEXT[pass + 1] = sum of the slices' tables (64 workgroups)
__global__ __launch_bounds__(LS_RED_THREADS) void ls_reduce_kernel(const uint32_t* __restrict__ slices, uint32_t nslices, uint32_t* __restrict__ cnext) {
    __shared__ uint32_t s_part[LS_RED_THREADS];
    const uint32_t word = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t v = ls_slice_sum(slices, nslices, word, false, s_part);
    if (threadIdx.x < 64u) cnext[word] = v;
}

// Behind the first kernel, ONE launch: workgroups [0, tblocks) transpose the run table — R[digit][tile] (a unit of the gather pass =
// consecutive tiles of one digit = contiguous words) from the rows the first kernel wrote per tile, 64 tiles x 256 words through LDS
// (row stride 257: conflict-free both ways) — the other 72 sum the slices: the gather pass's table -> CNEXT[1], digit-0 totals, OR /
// AND of the keys -> the LS words; the last of those to finish plans the passes.
__global__ __launch_bounds__(LS_RED_THREADS) void ls_plan_kernel(const uint32_t* __restrict__ runs_t, uint32_t* __restrict__ runs, uint32_t nt, uint32_t nt_pad,
                                                       uint32_t tblocks, const uint32_t* __restrict__ slices, uint32_t nslices, uint32_t* slab,
                                                       uint32_t plan /*bit1: identity passes may be dropped*/) {
    __shared__ uint32_t s[64 * 257];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* ls = slab + SLAB_LS;
    if (blockIdx.x < tblocks) {
        const uint32_t t0 = blockIdx.x * 64u;
        const uint32_t rows = nt - t0 < 64u ? nt - t0 : 64u;
        {   // 16 rows per thread, all loads in flight (a counted loop waits for each load before the next one is issued)
            uint32_t v[16];
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                const uint32_t r = (tid >> 8) + 4u * k;
                v[k] = runs_t[(size_t)(t0 + (r < rows ? r : rows - 1u)) * RADIX + (tid & 255u)];
            }
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) s[((tid >> 8) + 4u * k) * 257u + (tid & 255u)] = v[k];
        }
        __syncthreads();
        for (uint32_t d = wave; d < RADIX; d += LS_RED_THREADS / 64u)
            if (lane < rows) runs[(size_t)d * nt_pad + t0 + lane] = s[lane * 257u + d];
        return;
    }
    const uint32_t rb = blockIdx.x - tblocks;  // 0 .. 71
    const uint32_t word = rb * 64u + lane;
    const bool is_or = word >= LS_SLICE_OR;
    const uint32_t v = ls_slice_sum(slices, nslices, word, is_or, s);
    if (tid < 64u) {
        if (word < LS_SLICE_TOT) slab[SLAB_HSUB + HSUB_STRIDE + word] = v;      // CNEXT[1]
        else if (word < LS_SLICE_OR) ls[LS_TOT0 + (word - LS_SLICE_TOT)] = v;
        else if (word < LS_SLICE_OR + 2u) ls[LS_OR + (word - LS_SLICE_OR)] = v;
    }
    __syncthreads();
    if (tid == 0) {  // (one lane's release behind the barrier: a fence by every thread writes the XCD's L2 back once per wave)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s[0] = atomicAdd(&ls[LS_DONE], 1u);
    }
    __syncthreads();
    if (s[0] != LS_SLICE_WORDS / 64u - 1u) return;
    // ---- the last summing workgroup plans the passes (everything above is complete: stores, release, the arrival counter;
    // it reads the sums with agent-scope loads) ----
    const uint32_t varying = ld_agent(&ls[LS_OR]) & ld_agent(&ls[LS_NAND]);  // bit set: some key has it, some key does not
    const bool drop23 = (plan & 2u) && ((varying >> 16) & 0xffffu) == 0u;    // bytes 2 and 3 constant: passes 2, 3 are identities
    if (tid == 0) ls[LS_TICKET] = 0u;
    if (tid < 4) ls[LS_FLAGS + tid] = tid == 1u ? (drop23 ? PF_LAST : 0u) : tid == 2u ? (drop23 ? PF_SKIP : 0u) : tid == 3u ? (drop23 ? PF_SKIP : PF_LAST) : 0u;
    // tile geometry of the gather pass: digit d's keys, in (source tile, position) order, are cut into tiles of 16 384
    if (tid < RADIX) s[1024 + tid] = (ld_agent(&ls[LS_TOT0 + tid]) + LS_TILE - 1u) / LS_TILE;
    __syncthreads();
    if (tid < RADIX) {
        uint32_t before = 0, sb = 0;
        for (uint32_t d = tid & ~15u; d < tid; ++d) before += s[1024 + d];
        for (uint32_t d = 0; d < tid; ++d) sb += s[1024 + d] + 1u;
        ls[LS_UU0 + tid] = before;
        ls[LS_SB + tid] = sb;
        if ((tid & 15u) == 15u) ls[LS_CU + (tid >> 4)] = before + s[1024 + tid];
    }
}

// Behind ls_plan_kernel, one workgroup of 1024 threads per digit-0 value d: E[d][t] = exclusive prefix over the source tiles of the
// lengths of d's runs (the virtual position, among d's keys, of the first key of run (d, t)), and for every tile j of the digit
// S[SB[d] + j] = the run that holds virtual position j * 16 384 (the last run starting at or before it: empty runs in front of it
// share its E and are skipped by the search); S[SB[d] + tiles] = the row's last run, so that tile j's runs are S[j] .. S[j + 1].
__global__ __launch_bounds__(1024) void ls_runscan_kernel(const uint32_t* __restrict__ runs, uint32_t* __restrict__ eprefix, uint32_t* __restrict__ stab,
                                                           uint32_t nt, uint32_t nt_pad, const uint32_t* slab) {
    constexpr uint32_t PER = 16, CHUNK = 1024 * PER;  // entries per thread / per round
    __shared__ __attribute__((aligned(16))) uint32_t s_e[CHUNK + CHUNK / 32];  // (one pad word per 32: the per-thread runs of 16 words stay off each other's banks)
    __shared__ uint32_t s_w[20];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, d = blockIdx.x;
    const uint32_t* row = runs + (size_t)d * nt_pad;
    uint32_t* erow = eprefix + (size_t)d * nt_pad;
    auto at = [](uint32_t i) { return i + (i >> 5); };
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nt; base += CHUNK) {
        // coalesced dword loads, all in flight; the lengths go to LDS, every thread then scans 16 consecutive ones
        uint32_t c[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = base + tid + 1024u * k;
            c[k] = row[i < nt ? i : nt - 1u];
        }
        __syncthreads();  // (the previous round's s_e and s_w are read)
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) s_e[at(tid + 1024u * k)] = base + tid + 1024u * k < nt ? c[k] & 0xffffu : 0u;
        __syncthreads();
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { c[k] = sum; sum += s_e[at(tid * PER + k)]; }  // exclusive inside the thread
        const uint32_t incl = wave_inclusive_scan_dpp(sum);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t wbase = carry, total = 0;
        for (uint32_t w = 0; w < 16u; ++w) { const uint32_t x = s_w[w]; if (w < wave) wbase += x; total += x; }
        const uint32_t ex = wbase + incl - sum;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) s_e[at(tid * PER + k)] = ex + c[k];
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = base + tid + 1024u * k;
            if (i < nt) erow[i] = s_e[at(tid + 1024u * k)];
        }
        carry += total;
    }
    const bool in_lds = nt <= CHUNK;  // (2^28 keys: the whole row is still in LDS; longer rows are searched in global memory)
    __syncthreads();
    if (!in_lds) {  // the row as this workgroup wrote it, for its own agent-scope loads below (ONE lane's release: a fence by every
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // thread writes the XCD's L2 back sixteen times over)
        __syncthreads();
    }
    const uint32_t tot = carry, tiles = (tot + LS_TILE - 1u) / LS_TILE;
    const uint32_t sb = slab[SLAB_LS + LS_SB + d];
    for (uint32_t j = tid; j <= tiles; j += 1024u) {
        uint32_t r = nt - 1u;
        if (j < tiles) {
            const uint32_t v = j * LS_TILE;
            uint32_t lo = 0, hi = nt;  // first entry with E > v
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t e = in_lds ? s_e[at(mid)] : __hip_atomic_load(&erow[mid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (e <= v) lo = mid + 1u; else hi = mid;
            }
            r = lo - 1u;
        }
        stab[sb + j] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pass kernel (passes 1..3 of the plan).  Persistent workgroups, two per CU; chains = the 16 groups of the previous digit's
// values; per chain one ticket counter and one run of descriptor rows (row 0 = the chain's base, seeded by whoever draws the
// chain's ticket 0).
// GATHER (pass 1): the input is the first kernel's tile-locally sorted output in its VIRTUAL order (digit 0, source tile, position in
//   the run).  Tile j of digit d = virtual positions [j T, (j + 1) T) of that digit's keys — full tiles, one partial tile per digit —
//   found through ls_runscan_kernel's tables: S (first run the tile touches) and E (exclusive prefix of the run lengths).  Every
//   thread turns one (E, R) pair into a list entry {source index, tile position | length << 16}; wave w then loads runs w, w + 8, ...
//   lane-contiguously straight into the stage (direct-to-LDS loads) and the keys are read back in tile order.
// COUNT: the pass counts the next pass's table.   mode: bit0 descending (applies on the pass flagged PF_LAST)
// ---------------------------------------------------------------------------------------------------------------------
template <int KT, bool GATHER, bool COUNT>
__global__ __launch_bounds__(LS_THREADS, 4) void ls_pass_kernel(const uint32_t* __restrict__ keys_in, uint32_t* __restrict__ keys_out,
                                                                 const uint32_t* __restrict__ runs, const uint32_t* __restrict__ eprefix,
                                                                 const uint32_t* __restrict__ stab, uint32_t nt_pad, uint32_t* slab,
                                                                 uint32_t* __restrict__ slices /*COUNT: [grid][LS_SLICE_WORDS]*/,
                                                                 uint32_t desc_off /*words: this pass's descriptor rows*/, uint32_t n,
                                                                 uint32_t pass /*1..3*/, uint32_t mode) {
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[LS_TILE];   // GATHER, while loading: the tile's keys in tile order
    __shared__ __attribute__((aligned(16))) uint32_t s_whist[LS_WH_WORDS];  // GATHER, while loading: the tile's run list (512 entries of 8 bytes)
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[COUNT ? LS_TAB_WORDS : 4];
    __shared__ uint32_t s_gbase[RADIX];
    __shared__ uint32_t s_misc[64];
    __shared__ uint32_t s_ct[NCH], s_cstart[NCH], s_cend[NCH], s_crow[NCH + 1], s_ctot[NCH];
    __shared__ uint32_t s_unit[3 * NCH];  // GATHER, the 16 digits of the claimed chain: first S word, tiles of the chain in front, keys
    const LsTile T{s_stage, s_whist, s_tab, s_misc};
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t shift = pass * 8u;
    uint32_t* ls = slab + SLAB_LS;
    uint32_t* status = slab + SLAB_STATUS;
    uint32_t* counters = slab + SLAB_COUNTERS + pass * COUNTERS_PER_PASS * COUNTER_STRIDE;
    uint32_t* desc = slab + desc_off;
    const uint32_t* cn_in = slab + SLAB_HSUB + pass * HSUB_STRIDE;     // [this digit][group of the previous digit]
    uint32_t* slice = slices + (size_t)blockIdx.x * LS_SLICE_WORDS;   // COUNT: the next pass's table as this workgroup counts it
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t nt = (n + LS_TILE - 1u) / LS_TILE;  // tiles of the first kernel

    const uint32_t pflags = uni(ls[LS_FLAGS + pass]);
    if (pflags & PF_SKIP) return;  // identity pass
    const bool reverse = (mode & 1u) && (pflags & PF_LAST);
    const uint32_t rev_xor = reverse ? 0xffffffffu : 0u, rev_add = reverse ? n : 0u;

    // ---- what a workgroup derives once: digit starts, chain geometry, first descriptor row of every chain ----
    uint32_t my_dstart = 0;  // threads < 256: global start of digit tid's keys in this pass's output
    {
        uint32_t v[NCH], G = 0, incl = 0;
        if (tid < NCH) s_ctot[tid] = 0;
        __syncthreads();
        if (tid < RADIX) {
#pragma unroll
            for (uint32_t q = 0; q < NCH / 4; ++q) {
                const uint4 x = reinterpret_cast<const uint4*>(cn_in + tid * NCH)[q];
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            }
#pragma unroll
            for (uint32_t x = 0; x < NCH; ++x) G += v[x];
            incl = wave_inclusive_scan_dpp(G);
            if (lane == 63) s_misc[4 + wave] = incl;
#pragma unroll
            for (uint32_t x = 0; x < NCH; ++x) {
                const uint32_t r = wave_reduce_sum(v[x]);
                if (lane == 0 && r) atomicAdd(&s_ctot[x], r);
            }
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_misc[4 + w];
            my_dstart = wbase + incl - G;
        }
        if (tid < NCH) {
            uint32_t st = 0;
            for (uint32_t x = 0; x < tid; ++x) st += s_ctot[x];
            s_cstart[tid] = st;
            s_cend[tid] = st + s_ctot[tid];
            s_ct[tid] = GATHER ? ls[LS_CU + tid] : chain_tiles(st, st + s_ctot[tid], LS_TILE);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t r = 0;
            for (uint32_t x = 0; x < NCH; ++x) { s_crow[x] = r; r += s_ct[x] + 1u; }
            s_crow[NCH] = r;
        }
        if constexpr (COUNT) {
            reinterpret_cast<uint4*>(s_tab)[tid] = uint4{0u, 0u, 0u, 0u};
            for (uint32_t i = tid; i < NCH * RADIX / 4; i += LS_THREADS) reinterpret_cast<uint4*>(slice)[i] = uint4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
    }
#if GS_LS_CLOCK  // tuning build: shader clocks per phase of workgroup 7's tiles, summed by its thread 0 into ls[920 + 16 (pass - 1) ..]
    const bool clk_on = blockIdx.x == 7u && tid == 0u;
    uint32_t clk_last = 0;
    auto stamp = [&](int k) { if (clk_on) { const uint32_t c = (uint32_t)__builtin_readcyclecounter(); s_misc[48 + k] += c - clk_last; clk_last = c; } };
    const unsigned long long wall0 = wall_clock64();
    if (clk_on) { for (int k = 0; k < 12; ++k) s_misc[48 + k] = 0; clk_last = (uint32_t)__builtin_readcyclecounter(); }
#else
    auto stamp = [](int) {};
#endif
    // (Measured and not kept, profiles/r04_ls_plan_status.txt: the loop software-pipelined by one tile — the next tile's ticket drawn at
    //  the top of a tile, its table words / keys requested while the tile waits in its look-back.  A tile claimed a tile time before
    //  it publishes its counts stalls the look-back of every successor (0.47 -> 0.55 ms per plain pass), and the 32 prefetched key
    //  registers spill, each scratch reload draining vmcnt, i.e. waiting for the prefetch itself.)
    uint32_t unit_chain = 0xffffffffu;  // GATHER: the chain whose digit tables s_unit holds
#pragma unroll 1
    for (;;) {  // one tile after the other until every chain is claimed
        __syncthreads();
        // ---- claim: the ticket counter of chain blockIdx % 16 (one XCD's L2 sees a chain's neighbouring tiles); when that chain
        // is used up, wave 0 looks at all chains in one round trip ----
        uint32_t chain = blockIdx.x & (NCH - 1u);
        if (tid == 0) {
            s_misc[2] = 0u;  // set when the look-back gives up
            s_misc[8] = 0u;  // row a stuck look-back asks the workgroup to recount
            s_misc[1] = atomicAdd(&counters[chain * COUNTER_STRIDE], 1u);
        }
        if (tid == 64) s_misc[3] = ld_agent(status);
        __syncthreads();
        if (uni(s_misc[3]) != STATUS_OK) break;  // an earlier pass gave up: its output is incomplete
        uint32_t tile = uni(s_misc[1]);
        if (GS_UNLIKELY(tile >= uni(s_ct[chain]))) {
            __syncthreads();
            if (wave == 0) {
                uint32_t tiles_x = 0;
                bool open = false;
                if (lane < NCH) {
                    tiles_x = s_ct[lane];
                    open = ld_agent(&counters[lane * COUNTER_STRIDE]) < tiles_x;
                }
                unsigned long long m = __builtin_amdgcn_ballot_w64(open);
                uint32_t got_x = 0, got_t = 0xffffffffu;
                while (m) {  // wave-uniform: the open chains one by one, starting behind our own
                    const unsigned long long above = m & ~((2ull << chain) - 1ull);
                    const uint32_t x = (uint32_t)__builtin_ctzll(above ? above : m);
                    uint32_t t = 0;
                    if (lane == 0) t = atomicAdd(&counters[x * COUNTER_STRIDE], 1u);
                    t = __builtin_amdgcn_readfirstlane(t);
                    if (t < (uint32_t)__builtin_amdgcn_readlane((int)tiles_x, (int)x)) { got_x = x; got_t = t; break; }
                    m &= ~(1ull << x);
                }
                if (lane == 0) { s_misc[0] = got_x; s_misc[1] = got_t; }
            }
            __syncthreads();
            chain = uni(s_misc[0]);
            tile = uni(s_misc[1]);
            if (tile == 0xffffffffu) break;  // every chain is fully claimed
        }
        stamp(0);  // claimed
        uint32_t* cdesc = desc + (size_t)uni(s_crow[chain]) * RADIX;  // row 0 of this chain
        if (GS_UNLIKELY(tile == 0u && tid < RADIX)) {  // the chain's base: the digit's start plus the chains in front
            uint32_t seed = my_dstart;
            for (uint32_t x = 0; x < chain; ++x) seed += cn_in[tid * NCH + x];
            st_agent(&cdesc[tid], (seed << 2) | FLAG_INCLUSIVE);
        }

        // ---- geometry of the tile ----
        // LINEAR: the chain's tile grid starts at its start rounded down to 64 keys (256-byte aligned wave loads); positions
        // [vlo, vhi) of the tile are its keys.  GATHER: tile j of digit gd, runs [ge0, ge1] of that digit's row, virtual start gv0.
        uint32_t tile_base = 0, vlo = 0, vhi = 0;
        uint32_t gd = 0, ge0 = 0, ge1 = 0, gv0 = 0;
        if constexpr (!GATHER) {
            const uint32_t seg_start = uni(s_cstart[chain]), seg_end = uni(s_cend[chain]);
            tile_base = (seg_start & ~63u) + tile * LS_TILE;
            const uint32_t lo = tile_base > seg_start ? tile_base : seg_start;
            const uint32_t hi = (seg_end - tile_base < LS_TILE) ? seg_end : tile_base + LS_TILE;
            vlo = lo - tile_base;
            vhi = hi - tile_base;
        } else {
            if (GS_UNLIKELY(chain != unit_chain)) {  // (uniform) another chain's tickets: its digit tables
                __syncthreads();
                if (tid < 3 * NCH) s_unit[tid] = ls[(tid < NCH ? LS_SB : tid < 2 * NCH ? LS_UU0 - NCH : LS_TOT0 - 2 * NCH) + chain * NCH + tid];
                __syncthreads();
                unit_chain = chain;
            }
            uint32_t i = 0;
            while (i + 1u < NCH && uni(s_unit[NCH + i + 1u]) <= tile) ++i;
            gd = chain * NCH + i;
            const uint32_t j = tile - uni(s_unit[NCH + i]), sidx = uni(s_unit[i]) + j, tot = uni(s_unit[2 * NCH + i]);
            ge0 = uni(stab[sidx]);
            ge1 = uni(stab[sidx + 1u]);
            gv0 = j * LS_TILE;
            vlo = 0;
            vhi = tot - gv0 < LS_TILE ? tot - gv0 : LS_TILE;
        }
        const uint32_t cnt = uni(vhi - vlo);
        const bool full = cnt == LS_TILE;
        stamp(1);  // geometry

        // ---- load ----
        uint32_t key[LS_KPT];
        // (opaque per tile: left to itself the compiler hoists all 32 positions p0 + 64 i out of the persistent loop, spills some of
        //  them, and every scratch reload waits for vmcnt(0) — the key loads then go out one at a time: 59 000 clocks per tile in this
        //  phase, profiles/r04_ls_pass_phase_clocks.txt)
        uint32_t p0 = wave * (64u * LS_KPT) + lane;
        asm volatile("" : "+v"(p0));
        if constexpr (GATHER) {
            // The tile's runs go STRAIGHT INTO THE STAGE, run by run: every thread turns one (E, R) table pair into a list entry
            // {source index, tile position | length << 16} (the counters' words hold 512 of them), then wave w issues runs w, w + 8, ...
            // as lane-contiguous direct-to-LDS loads (no address work per key, no registers held by loads in flight), and the keys are
            // read back in tile order.  (The first form — a run-start bitmap, a rank per key and 32 gathered register loads per thread —
            // spent 10 700 clocks issuing its loads against 2 000 in the linear pass, profiles/r04_ls_pass_phase_clocks.txt; the probe of
            // this form: profiles/r04_probe_gather3.txt.)
            uint2* s_runs = reinterpret_cast<uint2*>(s_whist);
            static_assert(LS_WH_WORDS >= 2 * LS_THREADS, "the run list lives in the counters' words");
            const uint32_t* erow = eprefix + (size_t)gd * nt_pad;
            const uint32_t* rrow = runs + (size_t)gd * nt_pad;
#pragma unroll 1
            for (uint32_t e = ge0; e <= ge1; e += LS_THREADS) {  // (uniform; one round unless the runs are short: skewed keys)
                const uint32_t i = e + tid;
                uint2 ent = {0u, 0u};
                if (i <= ge1) {
                    const uint32_t a = erow[i], w = rrow[i];
                    const uint32_t b = a + (w & 0xffffu);                   // the run's virtual range [a, b)
                    const uint32_t lo = a > gv0 ? a : gv0, hi = b < gv0 + cnt ? b : gv0 + cnt;
                    if (hi > lo) ent = uint2{i * LS_TILE + (w >> 16) + (lo - a), (lo - gv0) | ((hi - lo) << 16)};
                }
                if (e != ge0) __syncthreads();  // the last round's entries are read
                s_runs[tid] = ent;
                __syncthreads();
                const uint32_t nr = ge1 - e + 1u < LS_THREADS ? ge1 - e + 1u : LS_THREADS;
#pragma unroll 2
                for (uint32_t r = wave; r < nr; r += LS_WAVES) {
                    const uint2 en = s_runs[r];
                    const uint32_t src = uni(en.x), sl = uni(en.y);
                    const uint32_t s0 = sl & 0xffffu, len = sl >> 16;
#pragma unroll 1
                    for (uint32_t off = 0; off < len; off += 64u) {
                        if (off + lane < len)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(keys_in + src + off + lane),
                                                             (__attribute__((address_space(3))) void*)(s_stage + s0 + off), 4, 0, 0);
                    }
                }
            }
            stamp(2);  // run list, loads issued
            __builtin_amdgcn_s_waitcnt(0);  // (vmcnt(0): this wave's direct loads have landed)
            __syncthreads();
            if (GS_LIKELY(cnt != 0u)) {
                const uint32_t last = cnt - 1u;
#pragma unroll
                for (int i = 0; i < (int)LS_KPT; ++i) {
                    const uint32_t p = p0 + i * 64u;
                    key[i] = to_bits<KT>(s_stage[p < cnt ? p : last]);  // (positions behind a partial tile hold its last key again)
                }
            } else {
#pragma unroll
                for (int i = 0; i < (int)LS_KPT; ++i) key[i] = 0xffffffffu;
            }
            __syncthreads();  // the stage and the list are read: stage and counters may be reused
        } else if (GS_LIKELY(full)) {
#pragma unroll
            for (int i = 0; i < (int)LS_KPT; ++i) key[i] = to_bits<KT>(__builtin_nontemporal_load(keys_in + tile_base + p0 + i * 64u));
        } else {
#pragma unroll
            for (int i = 0; i < (int)LS_KPT; ++i) {
                const uint32_t p = p0 + i * 64u;
                const uint32_t pc = p < vlo ? vlo : (p >= vhi ? vhi - 1u : p);
                key[i] = to_bits<KT>(keys_in[tile_base + pc]);
            }
        }
        stamp(3);  // loads issued (GATHER: and the list is read)

        s_whist[tid] = 0;
        s_whist[tid + LS_THREADS] = 0;
        __syncthreads();
        uint32_t offp[LS_KPT / 2];
        ls_rank<false>(T, key, offp, shift, full, vlo, vhi, wave, lane);
        stamp(4);  // waited for the keys, ranked
        __syncthreads();
        uint32_t c0, c1, dpre0, dpre1;
        ls_digit_scan(T, tid, c0, c1, dpre0, dpre1);
        stamp(5);  // barrier + digit scan
        if (tid < RADIX / 2 && !GS_FAULT_TILE(chain, tile))  // publish the tile's counts (the pair in one 8-byte store)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(&cdesc[(size_t)(tile + 1u) * RADIX + 2u * tid]),
                               ((unsigned long long)((c1 << 2) | FLAG_REDUCTION) << 32) | ((c0 << 2) | FLAG_REDUCTION),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ls_stage(T, key, offp, shift, full, vlo, vhi, wave, lane);
        if constexpr (COUNT) ls_count(T, key, shift, full, vlo, vhi, wave, lane);
        stamp(6);  // staged (and counted)

        // ---- decoupled look-back inside the chain: threads < 128 walk for two digits each ----
        uint32_t prev0 = 0, prev1 = 0;  // keys of digits 2 tid, 2 tid + 1 in front of this tile in the pass's output
        {
            uint32_t spins = 0;
            int32_t k0 = (int32_t)tile, k1 = (int32_t)tile;
            bool fin0 = tid >= RADIX / 2, fin1 = fin0, poisoned = false;
            for (;;) {
                if (!(fin0 && fin1)) {
                    for (;;) {
                        int32_t stall_row = -1;
                        // both words of the pair in ONE round trip (one 8-byte load while the two walks stand on the same row — they
                        // are published together —, two loads in flight otherwise)
                        uint32_t v0 = 0, v1 = 0;
                        if (!fin0 && !fin1 && k0 == k1) {
                            const unsigned long long v = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&cdesc[(size_t)k0 * RADIX + 2u * tid]),
                                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v0 = (uint32_t)v;
                            v1 = (uint32_t)(v >> 32);
                        } else {
                            if (!fin0) v0 = ld_agent(&cdesc[(size_t)k0 * RADIX + 2u * tid]);
                            if (!fin1) v1 = ld_agent(&cdesc[(size_t)k1 * RADIX + 2u * tid + 1u]);
                        }
                        if (!fin0) {
                            const uint32_t f = v0 & FLAG_MASK;
                            if (f == FLAG_INCLUSIVE) { prev0 += v0 >> 2; fin0 = true; }
                            else if (f == FLAG_REDUCTION) { prev0 += v0 >> 2; --k0; }
                            else if (f == FLAG_POISON) { poisoned = true; fin0 = fin1 = true; }
                            else stall_row = k0;
                        }
                        if (!fin1) {
                            const uint32_t f = v1 & FLAG_MASK;
                            if (f == FLAG_INCLUSIVE) { prev1 += v1 >> 2; fin1 = true; }
                            else if (f == FLAG_REDUCTION) { prev1 += v1 >> 2; --k1; }
                            else if (f == FLAG_POISON) { poisoned = true; fin0 = fin1 = true; }
                            else stall_row = k1;
                        }
                        if (fin0 && fin1) break;
                        if (stall_row >= 0) {
                            __builtin_amdgcn_s_sleep(1);
                            ++spins;
                            if (GS_FALLBACK && stall_row > 0 && spins > FALLBACK_SPINS) {
                                atomicMax(&s_misc[8], (uint32_t)stall_row);  // ask the workgroup to recount tile stall_row - 1
                                break;
                            }
                            if (spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(status) != STATUS_OK)) {
                                poisoned = true;
                                fin0 = fin1 = true;
                                break;
                            }
                        }
                    }
                    if (fin0 && fin1 && tid < RADIX / 2) {
                        if (poisoned) {
                            st_agent(status, STATUS_TIMEOUT);
                            s_misc[2] = 1u;  // this tile must not scatter
                        }
                        if (!GS_FAULT_TILE(chain, tile)) {
                            const uint32_t w0 = poisoned ? FLAG_POISON : (((prev0 + c0) << 2) | FLAG_INCLUSIVE);
                            const uint32_t w1 = poisoned ? FLAG_POISON : (((prev1 + c1) << 2) | FLAG_INCLUSIVE);
                            __hip_atomic_store(reinterpret_cast<unsigned long long*>(&cdesc[(size_t)(tile + 1u) * RADIX + 2u * tid]),
                                               ((unsigned long long)w1 << 32) | w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                __syncthreads();
                if (!GS_FALLBACK) break;
                const uint32_t fb_row = uni(s_misc[8]);
                if (GS_LIKELY(fb_row == 0u)) break;
                // ---- fallback: a walk waited FALLBACK_SPINS polls on row fb_row.  The workgroup recounts that tile's digits from the
                // pass input (nobody writes it during the pass), offers them as REDUCTION descriptors (compare-and-swap on NOT_READY)
                // and the stuck walks go on below that row.  Scratch: s_gbase (written only after the look-back). ----
                {
                    if (tid < RADIX) s_gbase[tid] = 0;
                    __syncthreads();
                    if (tid == 0) s_misc[8] = 0u;
                    const uint32_t ft = fb_row - 1u;
                    if constexpr (!GATHER) {
                        const uint32_t seg_start = uni(s_cstart[chain]), seg_end = uni(s_cend[chain]);
                        const uint32_t fbase = (seg_start & ~63u) + ft * LS_TILE;
                        const uint32_t flo = fbase > seg_start ? fbase : seg_start;
                        const uint32_t fhi = (seg_end - fbase < LS_TILE) ? seg_end : fbase + LS_TILE;
                        for (uint32_t idx = flo + tid; idx < fhi; idx += LS_THREADS)
                            atomicAdd(&s_gbase[(to_bits<KT>(keys_in[idx]) >> shift) & 255u], 1u);
                    } else {  // tile ft of the chain: one wave per run
                        uint32_t i = 0;
                        while (i + 1u < NCH && uni(s_unit[NCH + i + 1u]) <= ft) ++i;
                        const uint32_t fd = chain * NCH + i, fj = ft - uni(s_unit[NCH + i]), fs = uni(s_unit[i]) + fj, ftot = uni(s_unit[2 * NCH + i]);
                        const uint32_t f0 = uni(stab[fs]), f1 = uni(stab[fs + 1u]), fv0 = fj * LS_TILE;
                        const uint32_t fv1 = ftot - fv0 < LS_TILE ? ftot : fv0 + LS_TILE;
                        const uint32_t* erow = eprefix + (size_t)fd * nt_pad;
                        const uint32_t* rrow = runs + (size_t)fd * nt_pad;
                        for (uint32_t e = f0 + wave; e <= f1; e += LS_WAVES) {
                            const uint32_t a = uni(erow[e]), w = uni(rrow[e]);
                            const uint32_t b = a + (w & 0xffffu), lo = a > fv0 ? a : fv0, hi = b < fv1 ? b : fv1;
                            const uint32_t src = e * LS_TILE + (w >> 16) + (lo - a);
                            for (uint32_t q = lane; lo + q < hi; q += 64u) atomicAdd(&s_gbase[(to_bits<KT>(keys_in[src + q]) >> shift) & 255u], 1u);
                        }
                    }
                    __syncthreads();
                    if (tid < RADIX / 2) {
                        if (!fin0 && k0 == (int32_t)fb_row) {
                            const uint32_t c = s_gbase[2 * tid];
                            atomicCAS(&cdesc[(size_t)fb_row * RADIX + 2u * tid], 0u, (c << 2) | FLAG_REDUCTION);
                            prev0 += c;
                            --k0;
                        }
                        if (!fin1 && k1 == (int32_t)fb_row) {
                            const uint32_t c = s_gbase[2 * tid + 1];
                            atomicCAS(&cdesc[(size_t)fb_row * RADIX + 2u * tid + 1u], 0u, (c << 2) | FLAG_REDUCTION);
                            prev1 += c;
                            --k1;
                        }
                        spins = 0;
                    }
                    __syncthreads();
                }
            }
        }
        if (uni(s_misc[2]) != 0u) break;  // timeout or poisoned predecessor: write nothing
        stamp(7);  // look-back
        if (tid < RADIX / 2) {  // stage slot i of digit d goes to s_gbase[d] + i
            s_gbase[2 * tid] = prev0 - dpre0;
            s_gbase[2 * tid + 1] = prev1 - dpre1;
        }
        bool near = false;
        if constexpr (COUNT) near = ls_tab_near_overflow(s_tab, tid);
        const int flush = __syncthreads_or(near ? 1 : 0);
        stamp(8);

        // ---- scatter: all stage reads first, then the base look-ups, then the stores ----
        if (GS_LIKELY(full)) {
            uint32_t kb[LS_KPT];
#pragma unroll
            for (int j = 0; j < (int)LS_KPT; ++j) kb[j] = s_stage[tid + j * LS_THREADS];
#pragma unroll
            for (int j = 0; j < (int)LS_KPT; ++j) {
                const uint32_t o = s_gbase[(kb[j] >> shift) & 255u] + tid + j * LS_THREADS;
                keys_out[(o ^ rev_xor) + rev_add] = from_bits<KT>(kb[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < (int)LS_KPT; ++j) {
                const uint32_t i = tid + j * LS_THREADS;
                if (i < cnt) {
                    const uint32_t kb = s_stage[i];
                    const uint32_t o = s_gbase[(kb >> shift) & 255u] + i;
                    keys_out[(o ^ rev_xor) + rev_add] = from_bits<KT>(kb);
                }
            }
        }
        if constexpr (COUNT) {
            if (GS_UNLIKELY(flush)) ls_tab_flush(s_tab, slice, tid);
        }
        stamp(9);  // scattered
#if GS_LS_CLOCK
        if (clk_on) ++s_misc[58];
#endif
    }
#if GS_LS_CLOCK
    if (clk_on) {
        uint32_t* o = ls + 920u + 16u * (pass - 1u);
        for (int k = 0; k < 10; ++k) o[k] = s_misc[48 + k] / (s_misc[58] ? s_misc[58] : 1u);
        o[10] = s_misc[58];
        o[11] = (uint32_t)(wall_clock64() - wall0);  // this workgroup's life, 100 MHz ticks
    }
#endif
    if constexpr (COUNT) {
        __syncthreads();
        ls_tab_flush(s_tab, slice, tid);
    }
}

}  // namespace gs
