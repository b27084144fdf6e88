// local_sort_proto_kernel.hpp — EXPERIMENT (not product, not built by build()): the LOCAL finish of a sort whose keys are already ordered by their top 16 bits.
//
// After the two OneSweep passes over bits 16..23 and 24..31 the array is a sequence of BUCKETS — runs of keys with
// equal top 16 bits — and what remains is to sort every bucket by its low 16 bits.  When the buckets are small
// (uniform 2^28 keys: 65 536 buckets of ~4096) that no longer needs global passes: a workgroup loads a window of
// the array into LDS, takes the buckets that START inside its nominal tile, sorts them there and writes them back
// in place — one read and one write of the array instead of the two of passes 0 and 1, sequential both ways, no
// descriptors, no look-back.  The reference has no counterpart (its OneSweep is four global passes,
// GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:301-363); the result is the same sorted array.
//
//   window  : LOC_CAP consecutive keys from the tile's first position
//   range   : [s, e) — s = first bucket boundary >= tile start, e = first bucket boundary >= tile end; the keys
//             before s belong to the previous workgroup's range, those from e on to the next one's.  Ranges of
//             different workgroups are disjoint, so the write-back is in place; a neighbour reading the window
//             while this workgroup writes sees keys with the same top 16 bits either way, which is all it looks at
//   overflow: e beyond the window (a bucket longer than LOC_CAP - tile) — the workgroup leaves its range alone and
//             counts itself in fail[0]; the caller then runs the general passes
//   sort    : LSD over the bits in which the range's keys can differ: key - (top 16 bits of the first key) needs
//             nb = 16 + log2(buckets spanned) bits, sorted in ceil(nb / 10) rounds of <= 10-bit digits; ranks from
//             returning LDS atomics on wave-private 16-bit counters (two per dword), the same mechanism and the same
//             lane-order requirement as RANK 1 of the DigitBinningPass (lds_atomic_order_probe)
#pragma once
#include "../gpusorting_amd/csrc/onesweep_kernels.hpp"

namespace gs {

#ifndef LOC_NT
#define LOC_NT 0
#endif
constexpr int LOC_THREADS = 512;
constexpr int LOC_KPT = 30;
constexpr uint32_t LOC_CAP = LOC_THREADS * LOC_KPT;  // 15 360 keys: 60 KiB stage + 16 KiB counters, two workgroups per CU
constexpr int LOC_WAVES = LOC_THREADS / 64;
constexpr uint32_t LOC_CNT_WORDS = 512;              // 1024 16-bit counters per wave

__device__ __forceinline__ uint32_t wave_reduce_min(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t t = __shfl_down(v, d, 64);
        v = t < v ? t : v;
    }
    return v;
}

template <int KT>
__global__ __launch_bounds__(LOC_THREADS, 4) void local_sort_kernel(uint32_t* keys, uint32_t n, uint32_t tile_keys, uint32_t* fail) {
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[LOC_CAP];
    __shared__ uint32_t s_cnt[LOC_WAVES * LOC_CNT_WORDS];
    __shared__ uint32_t s_ws[LOC_WAVES], s_lim[2];
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t lo = blockIdx.x * tile_keys;
    if (lo >= n) return;
    const uint32_t tile_end = (n - lo < tile_keys) ? n - lo : tile_keys;  // relative to lo
    const uint32_t wbase = wave * (64u * LOC_KPT);
    uint32_t k[LOC_KPT];
    if (tid < 2) s_lim[tid] = 0xffffffffu;
    // ---- window -> registers -> LDS, in array order
#pragma unroll
    for (int j = 0; j < LOC_KPT; ++j) {
        const uint32_t slot = wbase + j * 64u + lane;
        const uint32_t p = lo + slot;
        k[j] = ld_stream<LOC_NT != 0>(keys + (p < n ? p : n - 1u));  // (clamped: what lies behind the array is never looked at)
    }
    // the key before the window; position 0 is a boundary by construction
    const uint32_t before = lo ? keys[lo - 1] : ~keys[0];
#pragma unroll
    for (int j = 0; j < LOC_KPT; ++j) s_stage[wbase + j * 64u + lane] = k[j];
    __syncthreads();
    // ---- bucket boundaries: s = first one in the tile, e = first one at or after the tile's end
    {
        uint32_t smin = 0xffffffffu, emin = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < LOC_KPT; ++j) {
            const uint32_t slot = wbase + j * 64u + lane;
            const uint32_t prev = slot ? s_stage[slot - 1] : before;
            const uint32_t at = ((k[j] ^ prev) >> 16) != 0u ? slot : 0xffffffffu;  // (clamped loads repeat the last key: no boundary behind the array)
            smin = at < smin ? at : smin;
            const uint32_t at_e = slot >= tile_end ? at : 0xffffffffu;
            emin = at_e < emin ? at_e : emin;
        }
        if (smin >= tile_end) smin = 0xffffffffu;
        if (n - lo <= LOC_CAP && n - lo < emin) emin = n - lo;  // the end of the array closes the last bucket
        smin = wave_reduce_min(smin);
        emin = wave_reduce_min(emin);
        if (lane == 0) {
            if (smin != 0xffffffffu) atomicMin(&s_lim[0], smin);
            if (emin != 0xffffffffu) atomicMin(&s_lim[1], emin);
        }
    }
    __syncthreads();
    const uint32_t s_rel = uni(s_lim[0]), e_rel = uni(s_lim[1]);
    if (s_rel == 0xffffffffu) return;  // a bucket of an earlier tile covers this one
    if (e_rel == 0xffffffffu) {        // the last bucket runs out of the window
        if (tid == 0) atomicAdd(fail, 1u);
        return;
    }
    const uint32_t m = e_rel - s_rel;
    const uint32_t base = uni(s_stage[s_rel]) & 0xffff0000u;
    const uint32_t span = uni(s_stage[e_rel - 1u]) - base;
    const uint32_t nb = 32u - (uint32_t)__builtin_clz(span | 0xffffu);
    const uint32_t rounds = (nb + 9u) / 10u;
    const uint32_t width = (nb + rounds - 1u) / rounds;
    const uint32_t dmask = (1u << width) - 1u;
    // the range to the front of the stage order; what is not in it sorts behind everything
#pragma unroll
    for (int j = 0; j < LOC_KPT; ++j) {
        const uint32_t slot = wbase + j * 64u + lane;
        k[j] = slot < m ? s_stage[s_rel + slot] - base : 0xffffffffu;
    }
    __syncthreads();
    uint32_t* cnt = s_cnt + wave * LOC_CNT_WORDS;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t shift = r * width;
#pragma unroll
        for (int i = 0; i < (int)LOC_CNT_WORDS / 64; ++i) cnt[i * 64 + lane] = 0;
        uint32_t rk[LOC_KPT / 2];
#pragma unroll
        for (int j = 0; j < LOC_KPT; ++j) {
            const uint32_t d = (k[j] >> shift) & dmask;
            const uint32_t sh = (d & 1u) * 16u;
            const uint32_t old = __hip_atomic_fetch_add(&cnt[d >> 1], 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t mine = (old >> sh) & 0xffffu;
            if (j & 1) rk[j >> 1] |= mine << 16;
            else rk[j >> 1] = mine;
        }
        __syncthreads();
        // counters -> stage offsets: exclusive over the digits, then over the waves; thread t owns digits 2t, 2t + 1
        {
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < LOC_WAVES; ++w) tot += s_cnt[w * LOC_CNT_WORDS + tid];
            const uint32_t pair = (tot & 0xffffu) + (tot >> 16);
            const uint32_t incl = wave_inclusive_scan_dpp(pair);
            if (lane == 63) s_ws[wave] = incl;
            __syncthreads();
            uint32_t off = incl - pair;
            for (uint32_t w = 0; w < wave; ++w) off += s_ws[w];
            uint32_t acc = off | ((off + (tot & 0xffffu)) << 16);  // both halves stay below 2^16: offsets <= LOC_CAP
#pragma unroll
            for (int w = 0; w < LOC_WAVES; ++w) {
                const uint32_t c = s_cnt[w * LOC_CNT_WORDS + tid];
                s_cnt[w * LOC_CNT_WORDS + tid] = acc;
                acc += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < LOC_KPT; ++j) {
            asm volatile("" : "+v"(k[j]));  // recompute the digit here instead of carrying 30 of them across the barriers
            const uint32_t d = (k[j] >> shift) & dmask;
            const uint32_t pos = ((cnt[d >> 1] >> ((d & 1u) * 16u)) & 0xffffu) + ((rk[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
            s_stage[pos] = k[j];
        }
        __syncthreads();
        if (r + 1u < rounds) {
#pragma unroll
            for (int j = 0; j < LOC_KPT; ++j) k[j] = s_stage[wbase + j * 64u + lane];
        }
    }
    // ---- the sorted range back where it came from
    uint32_t* out = keys + lo + s_rel;
#pragma unroll
    for (int j = 0; j < LOC_KPT; ++j) {
        const uint32_t slot = wbase + j * 64u + lane;
        if (slot < m) st_stream(out + slot, from_bits<KT>(s_stage[slot] + base));
    }
}

}  // namespace gs
