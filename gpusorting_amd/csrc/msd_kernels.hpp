// msd_kernels.hpp — device side of the multi-GPU MSD split (BASELINE.json configs[3]; no reference counterpart,
// SURVEY.md 5.8 / 8e): the shard's MSD histogram folded out of the sort's own joint histograms, and the PLAN of the
// bucket exchange — splitters, per-peer send/receive counts, overflow — computed on the device from the
// all-gathered histograms, so that the only thing the host ever reads is the handful of counts the RCCL
// send/recv calls need as host integers.
#pragma once
#include "onesweep_kernels.hpp"

namespace gs {

// plan layout (uint32 words), shared by msd_plan_kernel, gs_msd_plan (host) and the callers
constexpr uint32_t PLAN_NRECV = 0;      // keys this rank receives
constexpr uint32_t PLAN_OVERFLOW = 1;   // != 0: some rank's bucket exceeds `capacity`
constexpr uint32_t PLAN_MAXBUCKET = 2;  // largest bucket (saturated to 2^32-1)
constexpr uint32_t PLAN_PEER_FAILED = 3;  // != 0: some rank gathered a poisoned histogram row (it failed before the gather)
constexpr uint32_t MSD_POISON = 0x80000000u;  // a count no shard can reach (n < 2^30): marks the row of a rank that has failed
constexpr uint32_t PLAN_HEADER = 4;     // then send_counts[world], recv_counts[world], first_bin[world + 1]
__host__ __device__ constexpr uint32_t plan_words(uint32_t world) { return PLAN_HEADER + 3 * world + 1; }
constexpr uint32_t MSD_MAX_WORLD = 256;

// Shard histogram of the MSD bins out of the HIST region the GlobalHistogram kernel just filled:
//   nbins == 256  (prologue p0 = 3, np = 1): top byte, summed over the position-segment chains
//   nbins == 4096 (prologue p0 = 2, np = 2): 12-bit prefix = (top byte, top nibble of the byte below) — the joint
//                 histogram of the last pass, bin = d3 * 16 + (d2 >> 4)
__global__ __launch_bounds__(256) void msd_fold_kernel(const uint32_t* hist, uint32_t nbins, uint32_t* out) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= nbins) return;
    if (nbins == RADIX) {
        uint32_t g = 0;
#pragma unroll
        for (uint32_t x = 0; x < NCH; ++x) g += hist[hist_index(0, b, x)];
        out[b] = g;
    } else {
        out[b] = hist[hist_index(1, b / NCH, b % NCH)];
    }
}

__device__ __forceinline__ unsigned long long wave_inclusive_scan_u64(unsigned long long v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

// One workgroup.  table[src * nbins + b] = keys of rank `src` in MSD bin b (all-gathered).  Splitters: rank r starts
// at the first bin whose exclusive prefix of the GLOBAL histogram reaches ceil(r * total / world) — equal-count
// buckets at bin granularity, the same rule as gs_msd_splitters_n.  nbins is a multiple of 256, world <= 256.
__global__ __launch_bounds__(256) void msd_plan_kernel(const uint32_t* table, uint32_t nbins, uint32_t world, uint32_t rank,
                                                        uint32_t capacity, uint32_t* plan) {
    __shared__ unsigned long long s_wave[4], s_bucket[MSD_MAX_WORLD];
    __shared__ uint32_t s_first[MSD_MAX_WORLD + 1], s_send[MSD_MAX_WORLD], s_recv[MSD_MAX_WORLD], s_cnt;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = nbins / 256u, b0 = tid * per;  // this thread's bins [b0, b0 + per)
    for (uint32_t i = tid; i < MSD_MAX_WORLD; i += 256) { s_bucket[i] = 0; s_send[i] = 0; s_recv[i] = 0; }
    // global histogram of my bins and its exclusive prefix
    unsigned long long mine = 0;
    uint32_t poisoned = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (uint32_t j = 0; j < per; ++j)
        for (uint32_t r = 0; r < world; ++r) {
            const uint32_t c = table[(size_t)r * nbins + b0 + j];
            poisoned |= c & MSD_POISON;
            mine += c;
        }
    if (poisoned) atomicOr(&s_cnt, 1u);
    __syncthreads();
    const uint32_t peer_failed = s_cnt;
    __syncthreads();
    const unsigned long long incl = wave_inclusive_scan_u64(mine, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, total = 0;
    for (uint32_t w = 0; w < 4; ++w) {
        if (w < wave) base += s_wave[w];
        total += s_wave[w];
    }
    const unsigned long long my_excl = base + incl - mine;  // keys in bins < b0
    // splitters
    if (tid == 0) { s_first[0] = 0; s_first[world] = nbins; }
    for (uint32_t r = 1; r < world; ++r) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const unsigned long long target = (total * r + world - 1) / world;
        uint32_t below = 0;  // my bins whose exclusive prefix is < target
        unsigned long long p = my_excl;
        for (uint32_t j = 0; j < per; ++j) {
            below += p < target;
            unsigned long long g = 0;
            for (uint32_t q = 0; q < world; ++q) g += table[(size_t)q * nbins + b0 + j];
            p += g;
        }
        if (below) atomicAdd(&s_cnt, below);
        __syncthreads();
        if (tid == 0) s_first[r] = s_cnt;
    }
    __syncthreads();
    // counts: every bin goes to the rank whose range holds it
    uint32_t dst = 0;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t b = b0 + j;
        while (dst + 1 < world && s_first[dst + 1] <= b) ++dst;  // bins ascend: dst only moves forward
        unsigned long long g = 0;
        for (uint32_t q = 0; q < world; ++q) {
            const uint32_t c = table[(size_t)q * nbins + b];
            g += c;
            if (dst == rank && c) atomicAdd(&s_recv[q], c);
        }
        const uint32_t mine_b = table[(size_t)rank * nbins + b];
        if (mine_b) atomicAdd(&s_send[dst], mine_b);
        if (g) atomicAdd(&s_bucket[dst], g);
    }
    __syncthreads();
    for (uint32_t i = tid; i < world; i += 256) {
        plan[PLAN_HEADER + i] = s_send[i];
        plan[PLAN_HEADER + world + i] = s_recv[i];
    }
    for (uint32_t i = tid; i <= world; i += 256) plan[PLAN_HEADER + 2 * world + i] = s_first[i];
    if (tid == 0) {
        unsigned long long mx = 0, nrecv = 0;
        for (uint32_t r = 0; r < world; ++r) {
            mx = s_bucket[r] > mx ? s_bucket[r] : mx;
            nrecv += s_recv[r];
        }
        plan[PLAN_NRECV] = (uint32_t)(nrecv > 0xffffffffull ? 0xffffffffull : nrecv);
        plan[PLAN_OVERFLOW] = mx > capacity ? 1u : 0u;
        plan[PLAN_MAXBUCKET] = (uint32_t)(mx > 0xffffffffull ? 0xffffffffull : mx);
        plan[PLAN_PEER_FAILED] = peer_failed;
    }
}

// The rank's OWN share of a bucket exchange that goes top byte by top byte: up to 256 (source offset, destination offset, count)
// segments — elements of `words_per_elem` 4-byte words — copied by ONE launch (a hipMemcpyAsync per segment is a launch per top
// byte: 32 of them at world 8, in front of the RCCL kernel on the same stream).  blockIdx.x = segment * blocks_per_seg + part.
struct MsdSegments {
    uint32_t n;
    uint32_t seg[RADIX * 3];  // {source element, destination element, elements}
};
__global__ __launch_bounds__(256) void msd_copy_segments_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const MsdSegments segs,
                                                                 uint32_t words_per_elem, uint32_t blocks_per_seg) {
    const uint32_t i = blockIdx.x / blocks_per_seg, part = blockIdx.x % blocks_per_seg;
    if (i >= segs.n) return;
    const uint32_t* s = src + (size_t)segs.seg[3 * i] * words_per_elem;
    uint32_t* d = dst + (size_t)segs.seg[3 * i + 1] * words_per_elem;
    const size_t nw = (size_t)segs.seg[3 * i + 2] * words_per_elem;
    for (size_t w = (size_t)part * 256u + threadIdx.x; w < nw; w += (size_t)blocks_per_seg * 256u) d[w] = s[w];
}

}  // namespace gs
