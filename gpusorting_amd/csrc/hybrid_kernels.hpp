// hybrid_kernels.hpp — the TWO-LEVEL plan of large sorts of 32-bit keys (round 5): 28 bytes of HBM traffic per key instead of 36.
//
// The reference's OneSweep (GPUSortingCUDA/Sort/OneSweep.cu:44-600; dispatch OneSweepDispatcher.cuh:311-363) is one histogram
// sweep + four 8-bit LSD passes: 4 + 4 x 8 = 36 B/key.  On MI355X each pass already runs at the floor of its access shape
// (DESIGN.md 3.3), so the only lever left is bytes per key.  This plan keeps the reference's kernels — ONE GlobalHistogram sweep,
// a Scan, DigitBinningPasses with chained-scan decoupled look-back — but orders the work so that the last two digits never leave
// the chip:
//   hy_histogram_kernel   one sweep over the keys (the GlobalHistogram of OneSweep.cu:44-123): the 65 536-bin histogram of the keys'
//                         TOP 16 bits, packed 2 x 16 bit in 128 KiB of LDS per workgroup — ONE LDS add per key on it (the LSD
//                         plan's joint tables cost three) — plus the digit-0 counts per position segment that the LSD plans need,
//                         should this plan not apply.  The kernel is also the sort's clear, as global_histogram_kernel is.
//   hy_reduce_kernel      sum of the workgroups' slices -> joint32[65 536], per-segment top-byte counts, per-segment digit-0 counts
//   hy_scan_kernel        (the Scan of OneSweep.cu:125-162) exclusive scan of the 65 536 bins = start of every 16-bit-prefix bucket;
//                         DECIDES on the device whether the plan is valid — no counter overflow, no bucket above what one
//                         workgroup sorts in LDS — and then writes the info blocks, chain tables and seed rows of
//   pass A                DigitBinningPass on the TOP byte   (keys -> alt; 16 position chains, as the first pass of any sort), and
//   pass B                DigitBinningPass on byte 2         (alt -> keys; 256 chains = the top-byte buckets pass A just made: every
//                         tile lies inside one bucket, its 256 digit runs are seeded with that bucket's 16-bit-prefix starts)
//                         — both are the SAME kernels as the LSD passes (binning_body), told their digit and chain count by the
//                         info block — and
//   hy_local_sort_kernel  one workgroup per 16-bit-prefix bucket (n / 65 536 keys on average): loads the bucket, sorts it on the low
//                         16 bits by two 8-bit passes in LDS, and writes it back IN PLACE — sequential reads and writes
//                         (hy_local_sort_pairs_kernel: the same with values, which move once, behind the keys).
// (stable by byte 3) o (stable by byte 2 inside each byte-3 bucket) o (stable by the low 16 bits inside each 16-bit bucket) is
// the stable sort by the whole key: the result is bit-identical to the four LSD passes, values included.
//
// If the plan is not valid — skewed keys: presets 2-5 of the reference's entropy sweep put 10 %-60 % of the keys under one
// prefix — hy_scan_kernel raises HX_SKEW instead and the ordinary scan_kernel plans the four LSD passes on position chains
// (PF_POS) from the digit-0 counts of the same sweep: no key is read twice for the decision, nothing returns to the host.
// Every launch of a sort is enqueued regardless; the ones the other plan owns exit on their flag word (PF_SKIP / the valid word).
#pragma once
#include "onesweep_kernels.hpp"
#include "mid_kernels.hpp"

namespace gs {

constexpr uint32_t HY_BINS = 65536;               // values of a key's top 16 bits
constexpr uint32_t HY_JOINT_WORDS = HY_BINS / 2;  // the workgroup's table: two 16-bit counters per word
constexpr uint32_t HY_SLICE_WORDS = HY_JOINT_WORDS + RADIX;  // what a histogram workgroup hands on: its joint table + its digit-0 counts
// words of the slab's HY region
constexpr uint32_t HY_VALID = 0;      // 1: the sort runs on this plan (hy_local_sort_kernel's licence; zero whenever a sort starts)
constexpr uint32_t HY_MAXBUCKET = 1;  // largest 16-bit-prefix bucket (diagnostics)
// the plan's tables (an allocation of their own: nothing in them has to be zero when a sort starts)
constexpr uint32_t HYT_JOINT = 0;                       // joint32[65 536]: keys per 16-bit prefix
constexpr uint32_t HYT_BASE = HY_BINS;                  // base[65 536 + 1]: start of every prefix's bucket in the sorted order
constexpr uint32_t HYT_T = HYT_BASE + HY_BINS + 64;     // T[NCH][256]: keys of position segment x whose top byte is d
constexpr uint32_t HYT_WORDS = HYT_T + NCH * RADIX;
static_assert(HYT_BASE % 4 == 0 && HYT_T % 4 == 0, "16-byte accesses");

constexpr int HY_HIST_THREADS = 1024;
constexpr uint32_t HY_FOLD_CHUNKS = 256;  // the digit-0 replicas are folded at least this often (see global_histogram_kernel)

// ---------------------------------------------------------------------------
// Histogram sweep.  Workgroup w counts ONE contiguous range of keys inside one position segment of the first pass:
// segment x = w / wg_per_seg, range j = w % wg_per_seg of per_wg keys (a multiple of HIST_CHUNK).
//   s_j   the 16-bit-prefix table, 2 x 16 bit per word.  Random prefixes: one add per key, bank = 6 address bits of the prefix.
//         A counter that wraps loses 65 536 (high half) or 65 535 (low half carries into its neighbour) from the table's SUM, never
//         adds to it — so "sum of the decoded table == keys counted" is an EXACT test for "no counter wrapped", made when the
//         table is written out.  Dominant prefixes (sorted input, constant top bytes, skew) are counted with one add of a ballot's
//         popcount per wave, as in global_histogram_kernel.
//   s_r   digit 0 on 32 lane-private 16-bit replicas (never a bank conflict), folded into s_b0.
// ---------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(HY_HIST_THREADS, HY_HIST_THREADS / 256) void hy_histogram_kernel(
    const uint32_t* __restrict__ keys, uint32_t* slab, size_t slab_used_words, uint32_t n, uint32_t seg_len0, uint32_t per_wg,
    uint32_t wg_per_seg, uint32_t* slices, uint32_t cap /*keys the bucket-local sort's workgroup holds*/) {
    static_assert(KeyWords<KT>::value == 1, "32-bit keys");
    constexpr uint32_t T = HY_HIST_THREADS;
    __shared__ __attribute__((aligned(16))) uint32_t s_j[HY_JOINT_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t s_r[RADIX / 2 * 32];
    __shared__ uint32_t s_b0[RADIX];
    __shared__ uint32_t s_red[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t* hist = slab + SLAB_HIST;
    // the sort's clear (see global_histogram_kernel): everything nobody reads before this kernel ends
    {
        uint4* a = reinterpret_cast<uint4*>(slab);
        const size_t na = SLAB_HIST / 4, b0 = SLAB_HSUB / 4, nb = slab_used_words / 4;
        const size_t stride = (size_t)gridDim.x * T;
        const uint4 z = {0u, 0u, 0u, 0u};
        for (size_t i = (size_t)blockIdx.x * T + tid; i < na; i += stride) a[i] = z;
        for (size_t i = b0 + (size_t)blockIdx.x * T + tid; i < nb; i += stride) a[i] = z;
    }
    for (uint32_t i = tid; i < HY_JOINT_WORDS / 4; i += T) reinterpret_cast<uint4*>(s_j)[i] = uint4{0u, 0u, 0u, 0u};
    for (uint32_t i = tid; i < RADIX / 2 * 32 / 4; i += T) reinterpret_cast<uint4*>(s_r)[i] = uint4{0u, 0u, 0u, 0u};
    if (tid < RADIX) s_b0[tid] = 0;
    if (tid < 4) s_red[tid] = 0;
    // eight threads share the 32 replicas of a counter pair (see global_histogram_kernel's fold)
    auto fold = [&]() {
        __syncthreads();
        for (uint32_t i = tid; i < RADIX / 2 * 8; i += T) {
            const uint4 v = reinterpret_cast<const uint4*>(s_r)[i];
            reinterpret_cast<uint4*>(s_r)[i] = uint4{0u, 0u, 0u, 0u};
            uint32_t lo = (v.x & 0xffffu) + (v.y & 0xffffu) + (v.z & 0xffffu) + (v.w & 0xffffu);
            uint32_t hi = (v.x >> 16) + (v.y >> 16) + (v.z >> 16) + (v.w >> 16);
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                lo += __shfl_xor(lo, m, 64);
                hi += __shfl_xor(hi, m, 64);
            }
            if ((i & 7u) == 0u) {
                s_b0[(i >> 3) * 2u] += lo;
                s_b0[(i >> 3) * 2u + 1u] += hi;
            }
        }
        __syncthreads();
    };
    __syncthreads();

    const uint32_t x = blockIdx.x / wg_per_seg, jr = blockIdx.x % wg_per_seg;
    const unsigned long long sb = (unsigned long long)x * seg_len0, se = sb + seg_len0;
    const uint32_t seg_begin = sb < n ? (uint32_t)sb : n, seg_end = se < n ? (uint32_t)se : n;
    const unsigned long long rb = (unsigned long long)seg_begin + (unsigned long long)jr * per_wg;
    const uint32_t begin = rb < seg_end ? (uint32_t)rb : seg_end;
    const uint32_t end = (unsigned long long)begin + per_wg < seg_end ? begin + per_wg : seg_end;

    uint32_t k_or = 0, k_nand = 0;
    bool skew = false;               // wave-uniform: a dominant prefix was seen; cleared when it fades
    uint32_t sticky = 0xffffffffu;   // wave-uniform guess of it
    uint32_t since_fold = 0;
    // joint_off (workgroup-uniform, re-read once per work item): THIS workgroup alone has counted more than `cap` keys under one
    // prefix — that bucket cannot be sorted by one workgroup, the plan is void whatever the other workgroups see (an exact
    // verdict, not a sample: those keys exist).  It says so (HX_HY_BAD) and stops counting prefixes: skewed keys (presets 3-5 of the
    // entropy sweep: 12 % .. 60 % under prefix 0) then cost this kernel one add per key, like the LSD plan's histogram kernel.
    bool joint_off = false;
    typedef uint32_t hv4 __attribute__((ext_vector_type(4)));
    auto ld16 = [](const uint32_t* q) -> uint4 {
        const hv4 v = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(q));
        return uint4{v.x, v.y, v.z, v.w};
    };
    auto add_joint = [&](uint32_t p, uint32_t c) { atomicAdd(&s_j[p >> 1], c << ((p & 1u) << 4)); };
    // four keys of this thread (radix-sortable form); probe: look for a dominant prefix first (once per work item)
    auto process = [&](const uint4 t, const bool probe) {
        const uint32_t b[4] = {t.x, t.y, t.z, t.w};
        k_or |= t.x | t.y | t.z | t.w;
        k_nand |= ~(t.x & t.y & t.z & t.w);
        if (since_fold >= HY_FOLD_CHUNKS) {  // uniform
            fold();
            since_fold = 0;
        }
        ++since_fold;
#ifndef GS_HY_ABL_NO_REPLICA  // (ablation: what the digit-0 counts cost)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t d = b[j] & 255u;
            atomicAdd(&s_r[(d >> 1) * 32u + (lane & 31u)], 1u << ((d & 1u) * 16u));
        }
#endif
        if (joint_off) return;  // uniform
        if (probe) {
            const uint32_t p0 = b[0] >> 16;
            const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)p0);
            const uint32_t pc = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(p0 == f));
            if (pc >= GS_HIST_SKEW_LANES) {
                skew = true;
                if (pc >= 24 || sticky == 0xffffffffu) sticky = f;
            }
        }
        if (!skew) {  // uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) add_joint(b[j] >> 16, 1u);
            return;
        }
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(b[0] >> 16));
        const uint32_t pc = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((b[0] >> 16) == f));
        // Relearn BEFORE the adds: on sorted input every chunk of 4096 keys lies under a new prefix — relearning behind the adds put
        // each chunk's 64 lanes x 4 keys on ONE counter individually first (0.71 ms for the sweep on sorted keys, profiles/
        // r05_sorted_inputs.txt).
        if (pc >= 24 && f != sticky) sticky = f;
        // every lane's four keys under the remembered prefix (sorted / constant input): one add of 256 and done
        if (__builtin_amdgcn_ballot_w64((b[0] >> 16) == sticky && (b[1] >> 16) == sticky && (b[2] >> 16) == sticky && (b[3] >> 16) == sticky) == ~0ull) {
            if (lane == 0) add_joint(sticky, 256u);
            return;
        }
        uint32_t hit = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = b[j] >> 16;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(p == sticky);
            hit += (uint32_t)__popcll(m);
            if (p != sticky) add_joint(p, 1u);
            else if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u)
                add_joint(sticky, (uint32_t)__popcll(m));
        }
        if (hit < 32) {  // the guess covers < 1/8 of the lanes: relearn, or leave skew mode
            sticky = f;
            if (pc < 8) skew = false;
        }
    };
    // One work item = 4 consecutive chunks; all their 16-byte loads are issued before the first is consumed.
#ifndef GS_HY_HIST_UNROLL
#define GS_HY_HIST_UNROLL 4
#endif
    constexpr uint32_t UNROLL = GS_HY_HIST_UNROLL;
    for (uint32_t c0 = begin; c0 < end; c0 += UNROLL * HIST_CHUNK) {
        if ((unsigned long long)c0 + UNROLL * HIST_CHUNK <= end) {
            uint4 t[UNROLL];
#pragma unroll
            for (uint32_t u = 0; u < UNROLL; ++u) t[u] = ld16(keys + c0 + u * HIST_CHUNK + tid * 4u);
            if (!joint_off) {
                // the wave's dominant prefix against the bucket limit (one LDS read per wave and work item; a 16-bit counter is
                // read long before it could wrap: cap < 2^15)
                const uint32_t pf = to_bits<KT>((uint32_t)__builtin_amdgcn_readfirstlane((int)t[0].x)) >> 16;  // the prefix of the wave's first key: a hot prefix is drawn often
                if (lane == 0 && (((s_j[pf >> 1] >> ((pf & 1u) << 4)) & 0xffffu) > cap ||
                                  (skew && ((s_j[sticky >> 1] >> ((sticky & 1u) << 4)) & 0xffffu) > cap))) s_red[3] = 1u;
                joint_off = __builtin_amdgcn_readfirstlane((int)s_red[3]) != 0;  // (a benign race: a wave that misses the flag this time sees it next time)
            }
#pragma unroll
            for (uint32_t u = 0; u < UNROLL; ++u)
                process(uint4{to_bits<KT>(t[u].x), to_bits<KT>(t[u].y), to_bits<KT>(t[u].z), to_bits<KT>(t[u].w)}, u == 0);
            continue;
        }
        for (uint32_t c = c0; c < end; c += HIST_CHUNK) {  // the range's last, partial work item
            if ((unsigned long long)c + HIST_CHUNK <= end) {
                const uint4 t = ld16(keys + c + tid * 4u);
                process(uint4{to_bits<KT>(t.x), to_bits<KT>(t.y), to_bits<KT>(t.z), to_bits<KT>(t.w)}, true);
            } else {
                for (uint32_t i = c + tid; i < end; i += T) {
                    const uint32_t kb = to_bits<KT>(keys[i]);
                    k_or |= kb;
                    k_nand |= ~kb;
                    add_joint(kb >> 16, 1u);
                    atomicAdd(&s_b0[kb & 255u], 1u);
                }
            }
        }
    }
    fold();
    // the workgroup's slice: the packed table as it is (coalesced 16-byte stores) + the digit-0 counts; the overflow test on the way
    uint32_t* mine = slices + (size_t)blockIdx.x * HY_SLICE_WORDS;
    uint32_t sum = 0;
    for (uint32_t i = tid; i < HY_JOINT_WORDS / 4; i += T) {
        const uint4 v = reinterpret_cast<const uint4*>(s_j)[i];
        reinterpret_cast<uint4*>(mine)[i] = v;
        sum += (v.x & 0xffffu) + (v.x >> 16) + (v.y & 0xffffu) + (v.y >> 16) + (v.z & 0xffffu) + (v.z >> 16) + (v.w & 0xffffu) + (v.w >> 16);
    }
    if (tid < RADIX) mine[HY_JOINT_WORDS + tid] = s_b0[tid];
    sum = wave_reduce_sum(sum);
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) {
        k_or |= __shfl_xor(k_or, dd, 64);
        k_nand |= __shfl_xor(k_nand, dd, 64);
    }
    if (lane == 0) {
        atomicAdd(&s_red[0], sum);
        atomicOr(&s_red[1], k_or);
        atomicOr(&s_red[2], k_nand);
    }
    __syncthreads();
    if (tid == 0) {
        if (s_red[0] != end - begin || s_red[3] != 0u) atomicOr(&hist[HIST_TABLE_WORDS + HX_HY_BAD], 1u);
        // the OR / AND of all keys: how a sort planned on position chains finds its constant bytes (one atomic pair per workgroup)
        atomicOr(&hist[HIST_TABLE_WORDS + HX_OR], s_red[1]);
        atomicOr(&hist[HIST_TABLE_WORDS + HX_NAND], s_red[2]);
    }
}

// ---------------------------------------------------------------------------
// Sum of the slices.  Workgroup b < 256: top byte b — the 128 packed words of its 256 prefixes over all G slices (32 loads in
// flight per thread; two thread groups take alternate slices) -> joint32[b][0..255], and, slice ranges being position
// segments, T[x][b].  Workgroups 256 .. 256 + NCH - 1: digit-0 counts of position segment x -> the HIST region's table 0.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hy_reduce_kernel(const uint32_t* __restrict__ slices, uint32_t G, uint32_t wg_per_seg,
                                                         uint32_t* __restrict__ tab, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_acc[RADIX], s_T[NCH];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x >= RADIX) {
        const uint32_t x = blockIdx.x - RADIX;
        uint32_t acc = 0;
        for (uint32_t w = x * wg_per_seg; w < (x + 1u) * wg_per_seg && w < G; ++w) acc += slices[(size_t)w * HY_SLICE_WORDS + HY_JOINT_WORDS + tid];
        hist[hist_index(0, tid, x)] = acc;
        return;
    }
    // thread (grp, w4): 16-byte word w4 of the top byte's 128 packed words, in slices grp, grp + 8, ...: U loads in flight
    const uint32_t b3 = blockIdx.x, w4 = tid & 31u, grp = tid >> 5;
    s_acc[tid] = 0;
    if (tid < NCH) s_T[tid] = 0;
    __syncthreads();
    uint32_t a[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, cur_x = 0xffffffffu, acc_x = 0;
    constexpr uint32_t U = 16;
    for (uint32_t s0 = grp; s0 < G; s0 += 8u * U) {
        uint4 v[U];
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) {
            const uint32_t sl = s0 + 8u * k;  // (unconditional on a clamped index, masked below)
            v[k] = *reinterpret_cast<const uint4*>(slices + (size_t)(sl < G ? sl : G - 1u) * HY_SLICE_WORDS + b3 * 128u + w4 * 4u);
        }
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) {
            const uint32_t sl = s0 + 8u * k;
            if (sl < G) {
                const uint32_t xs = sl / wg_per_seg;
                if (xs != cur_x) {
                    if (acc_x) atomicAdd(&s_T[cur_x], acc_x);
                    cur_x = xs;
                    acc_x = 0;
                }
                const uint32_t h[8] = {v[k].x & 0xffffu, v[k].x >> 16, v[k].y & 0xffffu, v[k].y >> 16, v[k].z & 0xffffu, v[k].z >> 16, v[k].w & 0xffffu, v[k].w >> 16};
#pragma unroll
                for (int j = 0; j < 8; ++j) { a[j] += h[j]; acc_x += h[j]; }
            }
        }
    }
    if (acc_x) atomicAdd(&s_T[cur_x], acc_x);
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&s_acc[w4 * 8u + j], a[j]);  // word w4 * 4 + j / 2, half j & 1 = prefix (b3, w4 * 8 + j)
    __syncthreads();
    tab[HYT_JOINT + b3 * RADIX + tid] = s_acc[tid];
    if (tid < NCH) tab[HYT_T + tid * RADIX + b3] = s_T[tid];
}

// ---------------------------------------------------------------------------
// Scan + plan: one workgroup of 1024 threads, 64 prefixes per thread.
//   cap       keys hy_local_sort_kernel's workgroup holds
//   tile      keys per tile of passes A and B
//   plan_bits bit 0: descending (pass B applies the reversal: it is the plan's last DigitBinningPass; hy_local_sort_kernel mirrors)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hy_scan_kernel(uint32_t* slab, uint32_t* tab, uint32_t n, uint32_t seg_len0, uint32_t desc_stride,
                                                        uint32_t cap, uint32_t tile, uint32_t pregrouped /*the input is already grouped by
                                                        its top byte and lies in the ALTERNATE buffer (multi-GPU: the bucket exchange landed it
                                                        there bin by bin): pass A has nothing to do*/) {
    // Row i (16 rows) = prefixes [4096 i, 4096 (i + 1)); thread t holds prefixes 4096 i + 4 t .. + 3 of every row: all global
    // accesses are coalesced 16-byte ones (64 consecutive prefixes per thread, the first form, made every access of a wave touch 64
    // cache lines: 30 us for 256 KiB).  Wave w of row i = the 256 prefixes of top byte 16 i + w.
    __shared__ uint32_t s_wt[RADIX];  // [row][wave]: the wave's sum, then the exclusive prefix of that chunk (chunks in prefix order = flat index)
    __shared__ uint32_t s_wmax[16], s_c4[4], s_rw[4];
    __shared__ uint32_t s_bstart[RADIX + 1], s_rowbase[RADIX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* hist = slab + SLAB_HIST;
    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = reinterpret_cast<const uint4*>(tab + HYT_JOINT)[i * 1024 + tid];
    uint32_t incl[16], mx = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t s4 = v[i].x + v[i].y + v[i].z + v[i].w;
        const uint32_t m01 = v[i].x > v[i].y ? v[i].x : v[i].y, m23 = v[i].z > v[i].w ? v[i].z : v[i].w, m = m01 > m23 ? m01 : m23;
        mx = m > mx ? m : mx;
        incl[i] = wave_inclusive_scan_dpp(s4);
        if (lane == 63) s_wt[i * 16 + wave] = incl[i];
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)mx, dd, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) s_wmax[wave] = mx;
    __syncthreads();
    uint32_t cv = 0, cincl = 0;
    if (tid < RADIX) {
        cv = s_wt[tid];
        cincl = wave_inclusive_scan_dpp(cv);
        if (lane == 63) s_c4[wave] = cincl;
    }
    __syncthreads();
    uint32_t total = 0, maxb = 0;
    for (uint32_t w = 0; w < 4; ++w) total += s_c4[w];
    for (uint32_t w = 0; w < 16; ++w) maxb = s_wmax[w] > maxb ? s_wmax[w] : maxb;
    if (tid < RADIX) {
        uint32_t b = cincl - cv;
        for (uint32_t w = 0; w < wave; ++w) b += s_c4[w];
        s_wt[tid] = b;
    }
    const bool valid = hist[HIST_TABLE_WORDS + HX_HY_BAD] == 0u && maxb <= cap && total == n;  // uniform
    if (tid == 0) slab[SLAB_HY + HY_MAXBUCKET] = maxb;
    if (!valid) {
        // not this plan: the ordinary Scan kernel plans the LSD passes on position chains from the digit-0 counts of the same sweep
        if (tid == 0) hist[HIST_TABLE_WORDS + HX_SKEW] = 1u;
        return;
    }
    __syncthreads();
    // (thread d < 256: keys of position segment x whose top byte is d, for pass A's seed rows — requested here, used at the end)
    uint32_t tseg[NCH];
#pragma unroll
    for (uint32_t x = 0; x < NCH; ++x) tseg[x] = tid < RADIX ? tab[HYT_T + x * RADIX + tid] : 0u;
    // bucket starts (exclusive prefix over the 16-bit prefixes); v[] becomes the prefix
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint4 c = v[i];
        const uint32_t e = s_wt[i * 16 + wave] + incl[i] - (c.x + c.y + c.z + c.w);
        v[i] = uint4{e, e + c.x, e + c.x + c.y, e + c.x + c.y + c.z};
        reinterpret_cast<uint4*>(tab + HYT_BASE)[i * 1024 + tid] = v[i];
        if (lane == 0) s_bstart[i * 16 + wave] = e;  // start of top byte 16 i + wave
    }
    if (tid == 0) {
        s_bstart[RADIX] = n;
        tab[HYT_BASE + HY_BINS] = n;
    }
    __syncthreads();
    uint32_t* info0 = slab + SLAB_INFO;
    uint32_t* info1 = info0 + INFO_STRIDE;
    uint32_t* desc0 = slab + SLAB_DESC;
    uint32_t* desc1 = desc0 + desc_stride;
    // ---- pass B: chain c = top-byte bucket c; a chain owns tiles + 1 descriptor rows
    uint32_t rows = 0, rincl = 0;
    if (tid < RADIX) {
        rows = chain_tiles(s_bstart[tid], s_bstart[tid + 1], tile) + 1u;
        rincl = wave_inclusive_scan_dpp(rows);
        if (lane == 63) s_rw[wave] = rincl;
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t rb = rincl - rows;
        for (uint32_t w = 0; w < wave; ++w) rb += s_rw[w];
        s_rowbase[tid] = rb;
        info1[I_START + tid] = s_bstart[tid];
        info1[I_END + tid] = s_bstart[tid + 1];
        info1[I_ROW + tid] = rb;
    }
    __syncthreads();
    // seed rows of pass B: digit d of chain c starts at the start of prefix (c, d); a wave writes the row of chain 16 i + wave
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint4* row = reinterpret_cast<uint4*>(desc1 + (size_t)s_rowbase[i * 16 + wave] * RADIX);
        row[lane] = uint4{(v[i].x << 2) | FLAG_INCLUSIVE, (v[i].y << 2) | FLAG_INCLUSIVE, (v[i].z << 2) | FLAG_INCLUSIVE, (v[i].w << 2) | FLAG_INCLUSIVE};
    }
    // ---- pass A: the top byte over NCH position segments of the unsorted input (as the first pass of any sort)
    if (tid < RADIX) {
        uint32_t run = s_bstart[tid], rowa = 0;
#pragma unroll
        for (uint32_t x = 0; x < NCH; ++x) {
            const unsigned long long a = (unsigned long long)x * seg_len0, b = a + seg_len0;
            const uint32_t s0 = a < n ? (uint32_t)a : n, s1 = b < n ? (uint32_t)b : n;
            desc0[(size_t)rowa * RADIX + tid] = (run << 2) | FLAG_INCLUSIVE;
            run += tseg[x];
            if (tid == 0) {
                info0[I_START + x] = s0;
                info0[I_END + x] = s1;
                info0[I_ROW + x] = rowa;
            }
            rowa += chain_tiles(s0, s1, tile) + 1u;
        }
    }
    if (tid == 0) {
        info0[PASS_FLAGS] = pregrouped ? PF_SKIP : 0u;  // (pregrouped: pass B reads the alternate buffer, where the grouped input already is)
        info0[I_NCH] = NCH;
        info0[I_NEXT_SHIFT] = 0xffffffffu;
        info0[I_SHIFT] = 24u;
        info0[I_MODE] = 0xffffffffu;
        info1[PASS_FLAGS] = PF_SRC_ALT | PF_LAST;
        info1[I_NCH] = CHMAX;
        info1[I_NEXT_SHIFT] = 0xffffffffu;
        info1[I_SHIFT] = 16u;
        info1[I_MODE] = 0xffffffffu;
        info0[2 * INFO_STRIDE + PASS_FLAGS] = PF_SKIP;  // the launches of LSD passes 2 and 3 have nothing to do
        info0[3 * INFO_STRIDE + PASS_FLAGS] = PF_SKIP;
        hist[HIST_TABLE_WORDS + HX_HY] = 1u;
        slab[SLAB_HY + HY_VALID] = 1u;
    }
}
static_assert(CHMAX == RADIX, "pass B: one chain per top-byte value");

// ---------------------------------------------------------------------------
// One workgroup per 16-bit prefix: the bucket's keys (all share their top 16 bits) sorted on the low 16 bits by two stable
// 8-bit passes in LDS (the machinery of bucket_sort_kernel / small_sort_kernel), IN PLACE: the bucket is in registers before
// anything is written, and no other workgroup touches its range.  Descending: pass B wrote to mirrored positions (index
// n - 1 - o), so the bucket lies at [n - start - count, n - start) and is written back in reverse (keys only: equal keys
// are indistinguishable, the reverse of the stable ascending order is any descending order).
// ---------------------------------------------------------------------------
template <int KT, int THREADS_, int KPT_>
__global__ __launch_bounds__(THREADS_) void hy_local_sort_kernel(uint32_t* keys, const uint32_t* __restrict__ tab, uint32_t* __restrict__ slab,
                                                                 uint32_t n, uint32_t descending) {
    constexpr int KPT = KPT_, WAVES = THREADS_ / 64;
    constexpr uint32_t THREADS = THREADS_, TILE = THREADS_ * KPT_;
    static_assert(THREADS >= RADIX && KPT % 2 == 0 && WAVES % 2 == 0 && TILE < 65536, "one digit per thread in the scans; ranks and counters are packed two per word");
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[TILE];
    __shared__ uint32_t s_cnt[WAVES / 2 * RADIX];  // pass 1: [0 .. 255] one table for the workgroup; pass 2: per-wave counters, waves 2k and 2k + 1 in one word
    __shared__ uint32_t s_wtot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // (the plan's valid word and the bucket's geometry are requested together: one scalar round trip in front of the key loads.
    //  Measured and not kept: fewer workgroups that loop over 2 / 4 / 8 buckets each, the next bucket's geometry requested ahead —
    //  0.544 ms against 0.482 for one workgroup per bucket, profiles/r05_local_sort_variants.txt: the kernel lives on many
    //  independent workgroups per CU; the price is 65 536 workgroups that exit at once when the sort runs on the LSD passes, 0.03 ms.)
    const uint32_t valid = slab[SLAB_HY + HY_VALID], b_lo = tab[HYT_BASE + blockIdx.x], b_hi = tab[HYT_BASE + blockIdx.x + 1u];
    if (__builtin_amdgcn_readfirstlane((int)valid) == 0) return;  // the sort runs on the LSD passes
    const uint32_t start = b_lo, count = b_hi - b_lo;
    if (count == 0u || count > TILE || start > n || count > n - start) return;  // (the last three cannot happen with a valid plan)
    const uint32_t at = descending ? n - start - count : start;
    const uint32_t kpt = (uint32_t)__builtin_amdgcn_readfirstlane((int)((count + THREADS - 1u) / THREADS));  // uniform, 1 .. KPT
    const uint32_t my_base = wave * (64u * kpt) + lane;
    uint32_t key[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;  // uniform
        const uint32_t slot = my_base + i * 64u;
        key[i] = keys[at + (slot < count ? slot : count - 1u)];
    }
    if (tid < RADIX) s_cnt[tid] = 0;
#pragma unroll
    for (int i = 0; i < KPT; ++i)
        if ((uint32_t)i < kpt) key[i] = to_bits<KT>(key[i]);
    __syncthreads();
    // The kernel is bound by its LDS instructions (two passes x rank / base / stage / read back, profiles/r05_*): whatever is not
    // per key is kept small.
    // ---- pass 1, byte 0: NOT stable — the keys carry nothing along, so any order among equal low bytes will do — which lets the
    // whole workgroup rank on ONE counter table (arrival order) instead of one per wave
    uint32_t off[KPT / 2];
#pragma unroll
    for (int i = 0; i < KPT / 2; ++i) off[i] = 0;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;
        if (my_base + i * 64u < count)
            off[i >> 1] |= __hip_atomic_fetch_add(&s_cnt[key[i] & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) << (16 * (i & 1));
    }
    __syncthreads();
    uint32_t c = 0, scan_incl = 0;
    if (tid < RADIX) {
        c = s_cnt[tid];
        scan_incl = wave_inclusive_scan_dpp(c);
        if (lane == 63) s_wtot[wave] = scan_incl;
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += s_wtot[w];
        s_cnt[tid] = wbase + scan_incl - c;  // start of the digit's run
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;
        if (my_base + i * 64u < count) s_stage[((off[i >> 1] >> (16 * (i & 1))) & 0xffffu) + s_cnt[key[i] & 255u]] = key[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;
        key[i] = s_stage[my_base + i * 64u];  // (slots >= count hold nothing: masked wherever they are used)
    }
    // ---- pass 2, byte 1: stable (it must keep pass 1's order).  Per-wave counters, two waves per word (a wave counts at most
    // 64 x KPT, a stage slot is < TILE < 65 536: a half never carries into its neighbour)
    for (uint32_t i = tid; i < WAVES / 2 * RADIX; i += THREADS) s_cnt[i] = 0;  // (pass 1's readers of s_cnt are behind the barrier above)
    __syncthreads();
    const uint32_t wsh = (uint32_t)__builtin_amdgcn_readfirstlane((int)((wave & 1u) * 16u));
    uint32_t* wh = s_cnt + (wave >> 1) * RADIX;
#pragma unroll
    for (int i = 0; i < KPT / 2; ++i) off[i] = 0;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;
        if (my_base + i * 64u < count) {
            const uint32_t r = __hip_atomic_fetch_add(&wh[(key[i] >> 8) & 255u], 1u << wsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            off[i >> 1] |= ((r >> wsh) & 0xffffu) << (16 * (i & 1));
        }
    }
    __syncthreads();
    uint32_t run = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int p = 0; p < WAVES / 2; ++p) {
            const uint32_t c2 = s_cnt[p * RADIX + tid];
            s_cnt[p * RADIX + tid] = run | ((run + (c2 & 0xffffu)) << 16);
            run += (c2 & 0xffffu) + (c2 >> 16);
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
        for (int p = 0; p < WAVES / 2; ++p) s_cnt[p * RADIX + tid] += dpre * 0x10001u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;
        if (my_base + i * 64u < count)
            s_stage[((off[i >> 1] >> (16 * (i & 1))) & 0xffffu) + ((wh[(key[i] >> 8) & 255u] >> wsh) & 0xffffu)] = key[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = my_base + i * 64u;
        if ((uint32_t)i < kpt && slot < count) keys[at + (descending ? count - 1u - slot : slot)] = from_bits<KT>(s_stage[slot]);
    }
}

// ---------------------------------------------------------------------------
// The same for (key, value) pairs.  Stable in BOTH passes (values make equal keys distinguishable: the result must be the stable
// sort's, SortCommon.hlsl:594-597 for descending: its exact reverse).  The values do not travel through the two LDS passes:
// every key of the bucket shares its top 16 bits (= blockIdx.x), so those bits carry the key's ORIGINAL SLOT instead — the
// sorted element at slot j is (slot of origin << 16 | low 16 bits) — and the values move once, at the end: written to the stage
// by their owners in slot order, read back through the sorted elements' origin slots (one 4- / 8-byte LDS write and read per
// value instead of two each).  The stage of the values re-uses the keys' stage.
// ---------------------------------------------------------------------------
template <int KT, int VB, int THREADS_, int KPT_>
__global__ __launch_bounds__(THREADS_) void hy_local_sort_pairs_kernel(uint32_t* keys, void* vals_, const uint32_t* __restrict__ tab,
                                                                       const uint32_t* __restrict__ slab, uint32_t n, uint32_t descending) {
    using V = typename ValT<VB>::type;
    constexpr int KPT = KPT_, WAVES = THREADS_ / 64;
    constexpr uint32_t THREADS = THREADS_, TILE = THREADS_ * KPT_;
    static_assert(THREADS >= RADIX && KPT % 2 == 0 && WAVES % 2 == 0 && TILE < 65536, "slots of origin and counters are 16-bit fields");
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[TILE * (VB > 4 ? VB : 4)];
    __shared__ uint32_t s_cnt[WAVES / 2 * RADIX];
    __shared__ uint32_t s_wtot[4];
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(s_raw);
    V* s_vstage = reinterpret_cast<V*>(s_raw);
    V* vals = reinterpret_cast<V*>(vals_);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t valid = slab[SLAB_HY + HY_VALID], b_lo = tab[HYT_BASE + blockIdx.x], b_hi = tab[HYT_BASE + blockIdx.x + 1u];
    if (__builtin_amdgcn_readfirstlane((int)valid) == 0) return;  // the sort runs on the LSD passes
    const uint32_t start = b_lo, count = b_hi - b_lo;
    if (count == 0u || count > TILE || start > n || count > n - start) return;
    // descending: pass B wrote to mirrored positions — the bucket lies at [n - start - count, n - start) in REVERSE arrival order;
    // slot s is read from (and, sorted, written to) position count - 1 - s of that range
    const uint32_t at = descending ? n - start - count : start;
    const uint32_t kpt = (uint32_t)__builtin_amdgcn_readfirstlane((int)((count + THREADS - 1u) / THREADS));  // uniform, 1 .. KPT
    const uint32_t my_base = wave * (64u * kpt) + lane;
    uint32_t key[KPT];
    V val[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;  // uniform
        const uint32_t slot = my_base + i * 64u, cs = slot < count ? slot : count - 1u;
        const uint32_t pos = at + (descending ? count - 1u - cs : cs);
        key[i] = keys[pos];
        val[i] = vals[pos];
    }
    const uint32_t wsh = (uint32_t)__builtin_amdgcn_readfirstlane((int)((wave & 1u) * 16u));
    uint32_t* wh = s_cnt + (wave >> 1) * RADIX;
    // element = slot of origin << 16 | low 16 bits of the (radix-sortable) key
#pragma unroll
    for (int i = 0; i < KPT; ++i)
        if ((uint32_t)i < kpt) key[i] = (to_bits<KT>(key[i]) & 0xffffu) | ((my_base + i * 64u) << 16);
    uint32_t off[KPT / 2];
#pragma unroll 1
    for (uint32_t shift = 0; shift < 16; shift += 8) {
        for (uint32_t i = tid; i < WAVES / 2 * RADIX; i += THREADS) s_cnt[i] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT / 2; ++i) off[i] = 0;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if ((uint32_t)i >= kpt) continue;
            if (my_base + i * 64u < count) {
                const uint32_t r = __hip_atomic_fetch_add(&wh[(key[i] >> shift) & 255u], 1u << wsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                off[i >> 1] |= ((r >> wsh) & 0xffffu) << (16 * (i & 1));
            }
        }
        __syncthreads();
        uint32_t run = 0, scan_incl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int p = 0; p < WAVES / 2; ++p) {
                const uint32_t c2 = s_cnt[p * RADIX + tid];
                s_cnt[p * RADIX + tid] = run | ((run + (c2 & 0xffffu)) << 16);
                run += (c2 & 0xffffu) + (c2 >> 16);
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
            for (int p = 0; p < WAVES / 2; ++p) s_cnt[p * RADIX + tid] += dpre * 0x10001u;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if ((uint32_t)i >= kpt) continue;
            if (my_base + i * 64u < count)
                s_stage[((off[i >> 1] >> (16 * (i & 1))) & 0xffffu) + ((wh[(key[i] >> shift) & 255u] >> wsh) & 0xffffu)] = key[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if ((uint32_t)i >= kpt) continue;
            key[i] = s_stage[my_base + i * 64u];  // (slots >= count: masked wherever they are used)
        }
        // (the next pass zeroes the counters and crosses a barrier before anything writes the stage again)
    }
    // key[i] is now the sorted element of slot my_base + i * 64: its key goes out; its value is the one of its slot of origin
    __syncthreads();  // everybody has read the stage: it becomes the values' stage
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;
        if (my_base + i * 64u < count) s_vstage[my_base + i * 64u] = val[i];
    }
    __syncthreads();
    const uint32_t top = blockIdx.x << 16;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = my_base + i * 64u;
        if ((uint32_t)i < kpt && slot < count) {
            const uint32_t pos = at + (descending ? count - 1u - slot : slot);
            keys[pos] = from_bits<KT>(top | (key[i] & 0xffffu));
            vals[pos] = s_vstage[key[i] >> 16];
        }
    }
}

// ---------------------------------------------------------------------------
// A sort whose input arrived pre-grouped in the ALTERNATE buffer (hy_scan_kernel's `pregrouped`) and that runs on the LSD passes
// after all (the plan is void: skewed keys): the LSD plan reads the caller's buffer, so the input is copied there first.  Launched
// with every pregrouped sort; exits at once when the two-level plan is valid (16-byte words; `words16` of them, the tail by thread 0).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hy_void_copy_kernel(const uint32_t* __restrict__ slab, const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                            size_t words16, const uint32_t* __restrict__ src_tail, uint32_t* __restrict__ dst_tail,
                                                            uint32_t tail_words) {
    if (__builtin_amdgcn_readfirstlane((int)slab[SLAB_HY + HY_VALID]) != 0) return;
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < words16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail_words) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

}  // namespace gs
