/*
 * gs_oracle.cpp — CPU restatement of the reference's OneSweep path.
 * TEST INFRASTRUCTURE ONLY — see gs_oracle.h for the rules and the parity
 * pinning status (pinned against the reference's own generator and OneSweep
 * kernels executed on the CPU: oracle/ref_generator.cpp, oracle/ref_onesweep.cpp).
 * Citations are relative to /root/reference.
 */
#include "gs_oracle.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

namespace {

/* Hybrid Tausworthe/LCG step, GPUSortingCUDA/UtilityKernels.cuh:29-33.
 * The reference macros are only ever used as `z = MACRO;` so operator
 * precedence is the C one of the expanded text; restated with explicit
 * parentheses that reproduce that parse. */
struct Prng {
    uint32_t z1, z2, z3, z4;
    inline void step() {
        z1 = ((z1 & 4294967294U) << 12) ^ (((z1 << 13) ^ z1) >> 19);
        z2 = ((z2 & 4294967288U) << 4) ^ (((z2 << 2) ^ z2) >> 25);
        z3 = ((z3 & 4294967280U) << 17) ^ (((z3 << 3) ^ z3) >> 11);
        z4 = z4 * 1664525U + 1013904223U;
    }
    inline uint32_t value() const { return z1 ^ z2 ^ z3 ^ z4; }
};

inline uint32_t digit_of(uint32_t native, int key_type, uint32_t shift) {
    return (gso_key_to_bits(native, key_type) >> shift) & 255u;
}

template <class V>
void pass_impl(const uint32_t* kin, uint32_t* kout, const V* vin, V* vout, uint32_t n,
               uint32_t shift, int key_type, int reverse_index) {
    /* stable counting sort on one digit == what a DigitBinningPass computes
     * (GPUSortingCUDA/Sort/OneSweep.cu:164-344): every key goes to
     * globalExclusivePrefix[digit] + (number of earlier keys with that digit). */
    uint32_t count[256] = {0};
    for (uint32_t i = 0; i < n; ++i) count[digit_of(kin[i], key_type, shift)]++;
    uint32_t pos[256];
    uint32_t run = 0;
    for (int d = 0; d < 256; ++d) { pos[d] = run; run += count[d]; }
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t d = digit_of(kin[i], key_type, shift);
        uint32_t dst = pos[d]++;
        if (reverse_index) dst = n - 1 - dst; /* SortCommon.hlsl:594-597 */
        kout[dst] = kin[i];
        if (vin) vout[dst] = vin[i];
    }
}

template <class V>
void std_sort_pairs(uint32_t* keys, V* vals, uint32_t n, int key_type, int order) {
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;
    std::vector<uint32_t> bits(n);
    for (uint32_t i = 0; i < n; ++i) bits[i] = gso_key_to_bits(keys[i], key_type);
    std::stable_sort(idx.begin(), idx.end(),
                     [&](uint32_t a, uint32_t b) { return bits[a] < bits[b]; });
    if (order == GSO_DESCENDING) std::reverse(idx.begin(), idx.end());
    std::vector<uint32_t> k2(n);
    std::vector<V> v2(n);
    for (uint32_t i = 0; i < n; ++i) { k2[i] = keys[idx[i]]; v2[i] = vals[idx[i]]; }
    std::memcpy(keys, k2.data(), sizeof(uint32_t) * n);
    std::memcpy(vals, v2.data(), sizeof(V) * n);
}

}  // namespace

extern "C" {

void gso_init_random(uint32_t* keys, void* vals, uint32_t value_bytes, uint32_t and_count,
                     uint32_t seed, uint32_t n) {
    const uint32_t stride = 256u * 256u; /* <<<256,256>>>: OneSweepDispatcher.cuh:100,215 */
    const uint32_t nthreads = n < stride ? n : stride;
    for (uint32_t idx = 0; idx < nthreads; ++idx) {
        /* UtilityKernels.cuh:59-68: seeding + ONE discarded step */
        Prng p;
        p.z1 = (idx << 2) * seed;
        p.z2 = ((idx << 2) + 1) * seed;
        p.z3 = ((idx << 2) + 2) * seed;
        p.z4 = ((idx << 2) + 3) * seed;
        p.step();
        /* UtilityKernels.cuh:70-82 grid-stride loop; i += 65536 wraps in uint32 in
         * the reference too, but only for n > 2^32-65536 which the API rejects. */
        for (uint64_t i = idx; i < n; i += stride) {
            uint32_t t = 0xffffffffu;
            for (uint32_t k = 0; k <= and_count; ++k) {
                p.step();
                t &= p.value();
            }
            keys[i] = t;
            if (vals) {
                if (value_bytes == 4) ((uint32_t*)vals)[i] = t;       /* :114-115 */
                else if (value_bytes == 8) ((uint64_t*)vals)[i] = t;  /* zero-extended, cf. :157-168 */
            }
        }
    }
}

void gso_init_descending(uint32_t* keys, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) keys[i] = n - i; /* UtilityKernels.cuh:36-40 */
}

uint32_t gso_key_to_bits(uint32_t u, int key_type) {
    switch (key_type) {
        case GSO_KEY_I32: return u ^ 0x80000000u; /* SortCommon.hlsl:146-149 */
        case GSO_KEY_F32: {                        /* SortCommon.hlsl:134-138 (Herf) */
            uint32_t mask = (uint32_t)(-(int32_t)(u >> 31)) | 0x80000000u;
            return u ^ mask;
        }
        default: return u;
    }
}

uint32_t gso_bits_to_key(uint32_t u, int key_type) {
    switch (key_type) {
        case GSO_KEY_I32: return u ^ 0x80000000u; /* SortCommon.hlsl:151-154 */
        case GSO_KEY_F32: {                        /* SortCommon.hlsl:140-144 */
            uint32_t mask = ((u >> 31) - 1u) | 0x80000000u;
            return u ^ mask;
        }
        default: return u;
    }
}

void gso_global_histogram(const uint32_t* keys, uint32_t n, int key_type, uint32_t hist[1024]) {
    std::memset(hist, 0, 1024 * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t b = gso_key_to_bits(keys[i], key_type);
        hist[b & 255u]++;
        hist[256 + ((b >> 8) & 255u)]++;
        hist[512 + ((b >> 16) & 255u)]++;
        hist[768 + (b >> 24)]++;
    }
}

void gso_scan(const uint32_t hist[1024], uint32_t excl[1024]) {
    for (int p = 0; p < 4; ++p) {
        uint32_t run = 0;
        for (int d = 0; d < 256; ++d) { excl[p * 256 + d] = run; run += hist[p * 256 + d]; }
    }
}

void gso_digit_binning_pass(const uint32_t* kin, uint32_t* kout, const void* vin, void* vout,
                            uint32_t value_bytes, uint32_t n, uint32_t shift, int key_type,
                            int reverse_index) {
    if (!vin || value_bytes == 0)
        pass_impl<uint32_t>(kin, kout, nullptr, nullptr, n, shift, key_type, reverse_index);
    else if (value_bytes == 4)
        pass_impl<uint32_t>(kin, kout, (const uint32_t*)vin, (uint32_t*)vout, n, shift, key_type, reverse_index);
    else
        pass_impl<uint64_t>(kin, kout, (const uint64_t*)vin, (uint64_t*)vout, n, shift, key_type, reverse_index);
}

void gso_onesweep_sort(uint32_t* keys, uint32_t* alt_keys, void* vals, void* alt_vals,
                       uint32_t value_bytes, uint32_t n, int key_type, int order) {
    /* OneSweepDispatcher.cuh:325-335: 4 passes, shift 0/8/16/24, ping-pong */
    uint32_t* k[2] = {keys, alt_keys};
    void* v[2] = {vals, alt_vals};
    for (uint32_t p = 0; p < 4; ++p) {
        int rev = (order == GSO_DESCENDING && p == 3);
        gso_digit_binning_pass(k[p & 1], k[(p + 1) & 1], v[p & 1], v[(p + 1) & 1], value_bytes, n,
                               p * 8, key_type, rev);
    }
}

void gso_std_sort(uint32_t* keys, void* vals, uint32_t value_bytes, uint32_t n, int key_type,
                  int order) {
    if (!vals || value_bytes == 0) {
        if (key_type == GSO_KEY_U32) {
            std::sort(keys, keys + n);
        } else {
            std::sort(keys, keys + n, [key_type](uint32_t a, uint32_t b) {
                return gso_key_to_bits(a, key_type) < gso_key_to_bits(b, key_type);
            });
        }
        if (order == GSO_DESCENDING) std::reverse(keys, keys + n);
    } else if (value_bytes == 4) {
        std_sort_pairs<uint32_t>(keys, (uint32_t*)vals, n, key_type, order);
    } else {
        std_sort_pairs<uint64_t>(keys, (uint64_t*)vals, n, key_type, order);
    }
}

uint64_t gso_key64_to_bits(uint64_t u, int key_type) {
    /* the 32-bit rules of SortCommon.hlsl:134-154 on the 64-bit pattern */
    switch (key_type) {
        case GSO_KEY_I32: return u ^ 0x8000000000000000ull;
        case GSO_KEY_F32: return u ^ ((uint64_t)(-(int64_t)(u >> 63)) | 0x8000000000000000ull);
        default: return u;
    }
}

void gso_std_sort64(uint64_t* keys, void* vals, uint32_t value_bytes, uint32_t n, int key_type, int order) {
    /* 64-bit keys (not in the reference): the same definition as gso_std_sort — stable order by the
     * radix-sortable bits, descending = exact reverse of the stable ascending result. */
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;
    std::vector<uint64_t> bits(n);
    for (uint32_t i = 0; i < n; ++i) bits[i] = gso_key64_to_bits(keys[i], key_type);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return bits[a] < bits[b]; });
    if (order == GSO_DESCENDING) std::reverse(idx.begin(), idx.end());
    std::vector<uint64_t> k2(n);
    for (uint32_t i = 0; i < n; ++i) k2[i] = keys[idx[i]];
    std::memcpy(keys, k2.data(), sizeof(uint64_t) * n);
    if (vals && value_bytes == 4) {
        std::vector<uint32_t> v2(n);
        for (uint32_t i = 0; i < n; ++i) v2[i] = ((uint32_t*)vals)[idx[i]];
        std::memcpy(vals, v2.data(), sizeof(uint32_t) * n);
    } else if (vals && value_bytes == 8) {
        std::vector<uint64_t> v2(n);
        for (uint32_t i = 0; i < n; ++i) v2[i] = ((uint64_t*)vals)[idx[i]];
        std::memcpy(vals, v2.data(), sizeof(uint64_t) * n);
    }
}

void gso_digit_binning_pass64(const uint64_t* kin, uint64_t* kout, const void* vin, void* vout, uint32_t value_bytes,
                              uint32_t n, uint32_t shift, int key_type, int reverse_index) {
    /* one stable 8-bit partition pass on bit position shift (0..56) of the sortable 64-bit pattern */
    std::vector<uint32_t> pos(257, 0);
    for (uint32_t i = 0; i < n; ++i) pos[((gso_key64_to_bits(kin[i], key_type) >> shift) & 255u) + 1]++;
    for (int d = 0; d < 256; ++d) pos[d + 1] += pos[d];
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t dst = pos[(gso_key64_to_bits(kin[i], key_type) >> shift) & 255u]++;
        if (reverse_index) dst = n - 1 - dst;
        kout[dst] = kin[i];
        if (vin && value_bytes == 4) ((uint32_t*)vout)[dst] = ((const uint32_t*)vin)[i];
        if (vin && value_bytes == 8) ((uint64_t*)vout)[dst] = ((const uint64_t*)vin)[i];
    }
}

void gso_std_sort_parallel(uint32_t* keys, uint32_t n, uint32_t threads) {
    if (threads < 2 || n < (1u << 16)) { std::sort(keys, keys + n); return; }
    /* round the chunk count down to a power of two so the merge tree is regular */
    uint32_t chunks = 1;
    while (chunks * 2 <= threads) chunks *= 2;
    std::vector<size_t> bound(chunks + 1);
    for (uint32_t c = 0; c <= chunks; ++c) bound[c] = (size_t)n * c / chunks;
    {
        std::vector<std::thread> pool;
        for (uint32_t c = 0; c < chunks; ++c)
            pool.emplace_back([&, c] { std::sort(keys + bound[c], keys + bound[c + 1]); });
        for (auto& t : pool) t.join();
    }
    for (uint32_t width = 1; width < chunks; width *= 2) {
        std::vector<std::thread> pool;
        for (uint32_t c = 0; c + width < chunks + 0u; c += 2 * width) {
            size_t lo = bound[c], mid = bound[c + width];
            size_t hi = bound[std::min(c + 2 * width, chunks)];
            pool.emplace_back([=] { std::inplace_merge(keys + lo, keys + mid, keys + hi); });
        }
        for (auto& t : pool) t.join();
    }
}

void gso_sort_permutation_parallel(const uint32_t* keys, uint32_t n, int key_type, int order, uint32_t threads,
                                   uint32_t* perm) {
    /* The defined result of a pairs sort (OneSweep.cu:346-600: every pass is a STABLE partition) is the stable
     * order by key; with payload = original index that is the order of the 64-bit composites
     * (sortable bits << 32 | index), which are all distinct — so an ordinary (unstable) comparison sort of the
     * composites IS the stable sort, and it parallelises: chunked std::sort + merge tree.  Descending = exact
     * reverse of the stable ascending result (SortCommon.hlsl:594-597,645-656). */
    std::vector<uint64_t> c(n);
    uint32_t chunks = 1;
    while (chunks * 2 <= threads && (size_t)chunks * 2 * 65536 <= n) chunks *= 2;
    std::vector<size_t> bound(chunks + 1);
    for (uint32_t i = 0; i <= chunks; ++i) bound[i] = (size_t)n * i / chunks;
    {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < chunks; ++t)
            pool.emplace_back([&, t] {
                for (size_t i = bound[t]; i < bound[t + 1]; ++i)
                    c[i] = ((uint64_t)gso_key_to_bits(keys[i], key_type) << 32) | (uint32_t)i;
                std::sort(c.begin() + bound[t], c.begin() + bound[t + 1]);
            });
        for (auto& t : pool) t.join();
    }
    for (uint32_t width = 1; width < chunks; width *= 2) {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t + width < chunks; t += 2 * width) {
            const size_t lo = bound[t], mid = bound[t + width], hi = bound[std::min(t + 2 * width, chunks)];
            pool.emplace_back([&c, lo, mid, hi] { std::inplace_merge(c.begin() + lo, c.begin() + mid, c.begin() + hi); });
        }
        for (auto& t : pool) t.join();
    }
    if (order == GSO_DESCENDING)
        for (size_t i = 0; i < n; ++i) perm[n - 1 - i] = (uint32_t)c[i];
    else
        for (size_t i = 0; i < n; ++i) perm[i] = (uint32_t)c[i];
}

uint32_t gso_validate(const uint32_t* keys, const void* vals, uint32_t value_bytes, uint32_t n,
                      int key_type, int order) {
    /* UtilityKernels.cuh:403-429: count i with a[i] > a[i+1]; order/type-aware as
     * Utility.hlsl:147-230 (descending counts a[i] < a[i+1]; payload compared as
     * the KEY's type). */
    uint32_t err = 0;
    if (n < 2) return 0;
    for (uint32_t i = 0; i + 1 < n; ++i) {
        uint32_t a = gso_key_to_bits(keys[i], key_type), b = gso_key_to_bits(keys[i + 1], key_type);
        if (order == GSO_ASCENDING ? a > b : a < b) err++;
    }
    if (vals && value_bytes == 4) {
        const uint32_t* v = (const uint32_t*)vals;
        for (uint32_t i = 0; i + 1 < n; ++i) {
            uint32_t a = gso_key_to_bits(v[i], key_type), b = gso_key_to_bits(v[i + 1], key_type);
            if (order == GSO_ASCENDING ? a > b : a < b) err++;
        }
    } else if (vals && value_bytes == 8) {
        const uint64_t* v = (const uint64_t*)vals;
        for (uint32_t i = 0; i + 1 < n; ++i)
            if (order == GSO_ASCENDING ? v[i] > v[i + 1] : v[i] < v[i + 1]) err++;
    }
    return err;
}

void gso_msd_splitters(const uint64_t hist256[256], uint32_t world, uint32_t* first_bin) {
    uint64_t total = 0;
    for (int b = 0; b < 256; ++b) total += hist256[b];
    first_bin[0] = 0;
    uint64_t excl = 0;
    uint32_t b = 0;
    for (uint32_t r = 1; r < world; ++r) {
        /* first top-byte bin whose exclusive prefix reaches r/world of the keys
         * (ceil so that uniform data gives b = r*256/world exactly) */
        uint64_t target = (total * r + world - 1) / world;
        while (b < 256 && excl < target) { excl += hist256[b]; ++b; }
        first_bin[r] = b;
    }
    first_bin[world] = 256;
}

unsigned gso_hardware_threads(void) {
    unsigned t = std::thread::hardware_concurrency();
    return t ? t : 1;
}

}  // extern "C"
