/*
 * gs_oracle.h — CPU restatement of the reference's OneSweep path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libgpusort.so, the
 * gpusorting_amd package) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the timed CPU baseline.
 *
 * PARITY PINNING STATUS — pinned against the reference's own code, executed here
 *   The reference (b0nes164/GPUSorting @ 2024_10_08) holds NO golden vector or
 *   known-answer test for this path, and no toolchain for it
 *   exists in this image (nvcc / DXC / Unity).  But the CUDA tree's path is four
 *   kernels in one file and a generator in another, and a host compiler can
 *   parse both, from where they lie under /root/reference, against stand-in
 *   headers (oracle/shim/):
 *   - INPUT GENERATOR: GPUSortingCUDA/UtilityKernels.cuh:53-117 has no
 *     inter-thread communication; oracle/ref_generator.cpp runs it one emulated
 *     thread at a time (oracle/_ref/libref_generator.so).
 *   - SORT: GPUSortingCUDA/Sort/OneSweep.cu:44-600 (GlobalHistogram, Scan,
 *     DigitBinningPassKeysOnly, DigitBinningPassPairs) runs under a small SIMT
 *     emulator — fibers, 32-lane warp collectives, block barriers, blocks in
 *     ticket order (oracle/shim/simt_emu.*, oracle/ref_onesweep.cpp ->
 *     oracle/_ref/libref_onesweep.so).  Not the reference's own code in there:
 *     the warp primitives of Utils.cuh (inline PTX; restated with the same
 *     shuffle sequences) and the host launch sequence of
 *     OneSweepDispatcher.cuh:301-363 (cudaMemset / <<< >>>; restated).
 *   tests/test_oracle.py compares THIS restatement with both libraries directly
 *   (generator bit stream; global histogram, the buffers after every pass,
 *   sorted keys, payloads) where /root/reference exists, and
 *   tests/golden/ref_init_random.npz / ref_onesweep.npz (written by those
 *   libraries, tests/golden/make_ref_golden.py) carry the reference-produced
 *   vectors to boxes without the reference tree, where the oracle AND the HIP
 *   path are checked against them.
 *   Not executable here, hence pinned only by their definition: the D3D12/Unity
 *   extras — descending order and int/float key transforms
 *   (GPUSortingD3D12/Shaders/SortCommon.hlsl:134-154,594-597,645-656); the
 *   descending result is the exact reverse of the stable ascending one, the
 *   transforms are the standard order-preserving bit flips.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* key types / order: GPUSortingD3D12/GPUSorting.h:47-60 */
enum { GSO_KEY_U32 = 0, GSO_KEY_I32 = 1, GSO_KEY_F32 = 2 };
enum { GSO_ASCENDING = 0, GSO_DESCENDING = 1 };

/* GPUSortingCUDA/UtilityKernels.cuh:29-33,53-117 launched <<<256,256>>>
 * (GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:100,215): 65536 virtual threads,
 * one discarded PRNG step, then and_count+1 AND-ed draws per element.
 * vals may be NULL; value_bytes in {0,4,8}; value = key (zero-extended for 8). */
void gso_init_random(uint32_t* keys, void* vals, uint32_t value_bytes,
                     uint32_t and_count, uint32_t seed, uint32_t n);

/* GPUSortingCUDA/UtilityKernels.cuh:36-40 */
void gso_init_descending(uint32_t* keys, uint32_t n);

/* GPUSortingD3D12/Shaders/SortCommon.hlsl:134-154: native bits -> radix-sortable uint and back */
uint32_t gso_key_to_bits(uint32_t native_bits, int key_type);
uint32_t gso_bits_to_key(uint32_t sortable_bits, int key_type);

/* GPUSortingCUDA/Sort/OneSweep.cu:44-123: four 256-bin histograms (bytes 0..3
 * of the sortable bits), hist[p*256 + d]. */
void gso_global_histogram(const uint32_t* keys, uint32_t n, int key_type, uint32_t hist[1024]);

/* GPUSortingCUDA/Sort/OneSweep.cu:125-162: exclusive prefix sum of each row. */
void gso_scan(const uint32_t hist[1024], uint32_t excl[1024]);

/* GPUSortingCUDA/Sort/OneSweep.cu:164-344 (keys) / :346-600 (pairs): ONE stable
 * partition pass on digit (bits>>shift)&255.  reverse_index != 0 writes to
 * n-1-idx (the reference's descending rule on the last pass,
 * SortCommon.hlsl:594-597).  vals_in/out may be NULL. */
void gso_digit_binning_pass(const uint32_t* keys_in, uint32_t* keys_out,
                            const void* vals_in, void* vals_out, uint32_t value_bytes,
                            uint32_t n, uint32_t shift, int key_type, int reverse_index);

/* Whole sort through the 4 structural passes above (result ends in keys/vals;
 * alt buffers are scratch of the same size). */
void gso_onesweep_sort(uint32_t* keys, uint32_t* alt_keys, void* vals, void* alt_vals,
                       uint32_t value_bytes, uint32_t n, int key_type, int order);

/* Independent comparison-sort definition of the same result: std::sort (keys)
 * / std::stable_sort by key (pairs), descending = reverse of stable ascending.
 * This is also the "host std::sort" CPU baseline of BASELINE.md §3. */
void gso_std_sort(uint32_t* keys, void* vals, uint32_t value_bytes, uint32_t n,
                  int key_type, int order);

/* 64-bit keys (the reference sorts 32-bit keys only): the same definitions on 8-byte keys.
 * key_type GSO_KEY_U32 / I32 / F32 here mean uint64 / int64 / float64. */
uint64_t gso_key64_to_bits(uint64_t native_bits, int key_type);
void gso_std_sort64(uint64_t* keys, void* vals, uint32_t value_bytes, uint32_t n, int key_type, int order);
void gso_digit_binning_pass64(const uint64_t* keys_in, uint64_t* keys_out, const void* vals_in, void* vals_out,
                              uint32_t value_bytes, uint32_t n, uint32_t shift, int key_type, int reverse_index);

/* Multi-threaded variant for the cpu_baseline leg: chunked std::sort + merge
 * tree on `threads` host threads (keys only, u32 ascending). */
void gso_std_sort_parallel(uint32_t* keys, uint32_t n, uint32_t threads);

/* The defined result of a PAIRS sort as a permutation, for full-size parity tests: perm[j] = original index of
 * the element the stable sort by key (GPUSortingCUDA/Sort/OneSweep.cu:346-600) puts at position j; descending =
 * exact reverse (SortCommon.hlsl:594-597).  Sorted keys = keys[perm]; with payload = index the sorted payload IS
 * perm.  Comparison sort of the distinct composites (bits << 32 | index) on `threads` host threads. */
void gso_sort_permutation_parallel(const uint32_t* keys, uint32_t n, int key_type, int order, uint32_t threads,
                                   uint32_t* perm);

/* GPUSortingCUDA/UtilityKernels.cuh:402-479 and the order/type-aware form
 * GPUSortingD3D12/Shaders/Utility.hlsl:147-230: number of adjacent inversions
 * in keys (and, if vals != NULL and value_bytes==4, in vals reinterpreted as the key type). */
uint32_t gso_validate(const uint32_t* keys, const void* vals, uint32_t value_bytes,
                      uint32_t n, int key_type, int order);

/* Multi-GPU MSD split (no reference counterpart): given the
 * global top-byte histogram (256 x u64) pick world-1 splitters; returns for
 * each rank r the first top-byte value it owns in first_bin[r] (first_bin[0]=0,
 * first_bin[world]=256). */
void gso_msd_splitters(const uint64_t hist256[256], uint32_t world, uint32_t* first_bin);

unsigned gso_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
