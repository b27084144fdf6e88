/*
 * gpusort.h — C-ABI of the MI355X-native OneSweep radix sort (libgpusort.so).
 *
 * This is the drop-in boundary for the reference's OneSweep path
 * (b0nes164/GPUSorting @ 2024_10_08).  The reference has no FFI: its boundary
 * is a C++ class per backend.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference root).  On top of
 * this header, include/gpusort/OneSweepDispatcher.hpp re-creates the CUDA
 * tree's `OneSweepDispatcher` class verbatim (same method names, arguments
 * and print format) so GPUSortingCUDA/GPUSortingCUDA.cu:20-23,36-39 compiles
 * unchanged against it; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no HIP/torch types: streams are passed as void* (a hipStream_t;
 *     NULL = the null stream); device pointers are void* and caller-owned.
 *   - every call returns a gs_status; nothing throws across the boundary.
 *   - all calls are asynchronous on `stream` unless stated otherwise.
 *   - key/value device buffers must be 16-byte aligned.
 *   - 1 <= n <= GS_MAX_KEYS (30-bit tile-descriptor payload, same limit as the
 *     reference: GPUSortingCUDA/SegSort/SplitSort/SplitSortLarge.cuh:795-800).
 *   - a handle serialises its sorts: one in-flight sort per handle (it owns the
 *     chained-scan state).  Use one handle per stream.
 */
#ifndef GPUSORT_H
#define GPUSORT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_MAX_KEYS ((1u << 30) - 1u)

typedef enum gs_status {
    GS_OK = 0,
    GS_ERR_ARG = 1,      /* NULL / misaligned pointer, bad enum */
    GS_ERR_SIZE = 2,     /* n == 0, n > max_keys of the handle, n > GS_MAX_KEYS */
    GS_ERR_HIP = 3,      /* a HIP runtime call failed (gs_last_hip_error() has the code) */
    GS_ERR_TIMEOUT = 4,  /* a bounded look-back spin expired on the device */
    GS_ERR_MODE = 5,     /* pairs call on a keys-only handle / value width mismatch */
    GS_ERR_NO_DEVICE = 6, /* no gfx950 device visible */
    GS_ERR_COMM = 7       /* multi-GPU: RCCL could not be loaded or a collective failed (gs_last_rccl_error()) */
} gs_status;

/* GPUSortingD3D12/GPUSorting.h:40-45 */
typedef enum gs_mode { GS_MODE_KEYS_ONLY = 0, GS_MODE_PAIRS = 1 } gs_mode;
/* GPUSortingD3D12/GPUSorting.h:47-52 */
typedef enum gs_order { GS_ORDER_ASCENDING = 0, GS_ORDER_DESCENDING = 1 } gs_order;
/* GPUSortingD3D12/GPUSorting.h:54-60 */
typedef enum gs_key_type {
    GS_KEY_UINT32 = 0, GS_KEY_INT32 = 1, GS_KEY_FLOAT32 = 2,
    /* 64-bit keys (the reference has 32-bit keys only): 8-byte elements in d_keys / d_alt, sorted by
     * eight stable passes of the same kernels, planned by ONE GlobalHistogram sweep + Scan (identity passes are dropped
     * in pairs across the whole key); values as for 32-bit keys.  Accepted by
     * gs_onesweep_sort_keys / _sort_pairs / _digit_pass (pass 0..7) and gs_validate; not by the histogram read-back,
     * the MSD split and the generator, which are 32-bit. */
    GS_KEY_UINT64 = 3, GS_KEY_INT64 = 4, GS_KEY_FLOAT64 = 5
} gs_key_type;
/* GPUSortingCUDA/UtilityKernels.cuh:16-24 (value = number of extra AND-ed draws) */
typedef enum gs_entropy_preset {
    GS_ENTROPY_PRESET_1 = 0, GS_ENTROPY_PRESET_2 = 1, GS_ENTROPY_PRESET_3 = 2,
    GS_ENTROPY_PRESET_4 = 3, GS_ENTROPY_PRESET_5 = 4
} gs_entropy_preset;

typedef struct gs_onesweep gs_onesweep; /* opaque sorter state */

const char* gs_version(void);
const char* gs_status_string(gs_status s);
int gs_last_hip_error(void); /* hipError_t of the last failing HIP call on this thread */

/* ---- sorter object -------------------------------------------------------
 * Replaces: OneSweepDispatcher::OneSweepDispatcher(bool keysOnly, uint32_t maxSize)
 * / ~OneSweepDispatcher (GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:42-83) for
 * the scan state (m_index, m_globalHistogram, m_*PassHistogram), and the temp
 * buffers of Unity's OneSweep ctor (GPUSortingUnity/Runtime/OneSweep.cs:27-81).
 * Key/value/alt buffers stay with the caller (Unity ownership model).
 * value_bytes: 0 (keys only), 4 or 8.  Synchronous (allocates). */
gs_status gs_onesweep_create(gs_onesweep** out, uint32_t max_keys, gs_mode mode, uint32_t value_bytes);
gs_status gs_onesweep_destroy(gs_onesweep* h);

/* Everything that can be chosen about a sorter, in one place (the analogue of the reference's GPUSortingConfig + DeviceInfo,
 * GPUSortingD3D12/GPUSorting.h:40-86: mode / order / key type / payload type travel with the calls here; what is left are the
 * algorithm switches below).  The library itself reads NO environment variables: gs_onesweep_create uses the defaults;
 * harnesses that want GPUSORT_* variables (gpusorting_amd/onesweep.py, tools/) translate them into this struct.
 * Fill it with gs_onesweep_options_default() first; struct_size lets the struct grow. */
typedef struct gs_onesweep_options {
    uint32_t struct_size;            /* sizeof(gs_onesweep_options) */
    uint32_t shape_threads;          /* tile shape, threads x keys per thread; 0 x 0 (default) = the library picks by size and mode */
    uint32_t shape_keys_per_thread;
    int32_t rank_mode;               /* -1 (default) probe the device: 1 if its LDS serves same-address lanes in lane order, else 0;
                                        0 = 64-lane ballot multi-split, 1 = one returning LDS atomic per key */
    int32_t small_path;              /* 1 (default): n <= 8192 .. 32 768 keys in ONE workgroup; 0: always tiled */
    int32_t mid_path;                /* 1 (default): up to 2^20 .. 2^22 keys in two launches (MSD pass + bucket sorts) */
    int32_t skip_passes;             /* 1 (default): identity passes (a constant byte) are dropped in pairs on the device */
    int32_t position_chains;         /* skewed keys: 1 (default) position-chain plan when the histogram kernel finds the digit
                                        groups uneven, 0 never, 2 always (tests, tuning) */
    uint32_t position_chains_min_log2;  /* ... from 2^this keys up (20 .. 30); the default, 25, means 2^25 + 1: up to 2^25 keys the smaller tile shape wins */
    int32_t key64_sweeps;            /* 64-bit keys: 1 (default) one histogram sweep plans all eight passes, 2 one sweep per word */
    int32_t plan;                    /* gs_onesweep_set_plan: 0 (default) the library picks — the two-level plan for large sorts whose
                                        keys turn out near-uniform, the four LSD passes otherwise; 1 LSD passes only; 2 two-level plan wherever it can run */
    int32_t first_pass_big;          /* 1 (default): keys-only mid sizes run their first pass on the 16 384-key tile */
    uint32_t hist_blocks;            /* workgroups of the GlobalHistogram kernel; 0 (default) = one per CU (tuning aid) */
    uint32_t debug_flags;            /* 0.  The product build ignores every bit.  Tuning / experiment builds (-DGS_TUNING, -DGS_EXP): bit 30 = the two-level
                                        plan stops behind its second pass (tools/hy_bringup.py; the result is NOT sorted), GS_EXP builds: bits 8-11 ablation modes */
} gs_onesweep_options;
void gs_onesweep_options_default(gs_onesweep_options* o);
/* gs_onesweep_create with explicit options (NULL = defaults).  GS_ERR_ARG for a struct_size this library does not know, a tile
 * shape it was not built with, or a value out of range. */
gs_status gs_onesweep_create_ex(gs_onesweep** out, uint32_t max_keys, gs_mode mode, uint32_t value_bytes, const gs_onesweep_options* options);

/* Bytes of device memory a handle for max_keys allocates (descriptors + histograms + the two-level plan's tables): an upper
 * bound over modes and options (default hist_blocks). */
size_t gs_onesweep_temp_bytes(uint32_t max_keys);
/* Keys per binning tile of this build (the reference's k_partitionSize = 7680,
 * OneSweepDispatcher.cuh:23; tests ladder sizes over [P, 2P]). */
uint32_t gs_onesweep_partition_size(gs_mode mode, uint32_t value_bytes);

/* ---- the hot path ---------------------------------------------------------
 * Replaces: OneSweepDispatcher::DispatchKernelsKeysOnly(uint32_t size)
 * (OneSweepDispatcher.cuh:311-336) and Unity OneSweep.Sort(...) keys overload
 * (GPUSortingUnity/Runtime/OneSweep.cs:297-323).  Sorts d_keys[0..n) in place
 * (result in d_keys, as in the reference after 4 passes); d_alt is scratch of
 * n keys.  key_type/order as D3D12 GPUSortBase (GPUSortingD3D12/OneSweep.h:16-27). */
gs_status gs_onesweep_sort_keys(gs_onesweep* h, void* d_keys, void* d_alt, uint32_t n,
                                gs_key_type key_type, gs_order order, void* stream);

/* Replaces: OneSweepDispatcher::DispatchKernelsPairs (OneSweepDispatcher.cuh:338-363)
 * and Unity OneSweep.Sort pairs overload (OneSweep.cs:358-390).  Values are
 * bit-copied (value_bytes of the handle); stable by key; descending = exact
 * reverse of the stable ascending result (SortCommon.hlsl:594-597,645-656). */
gs_status gs_onesweep_sort_pairs(gs_onesweep* h, void* d_keys, void* d_vals, void* d_alt_keys,
                                 void* d_alt_vals, uint32_t n, gs_key_type key_type,
                                 gs_order order, void* stream);

/* Synchronises `stream` and reads the device status word of the last sort:
 * GS_OK or GS_ERR_TIMEOUT.  With the look-back fallback of the default build (a tile that
 * waits too long recounts its predecessor itself) a timeout cannot occur; the word exists
 * for builds without it (-DGS_FALLBACK=0), whose bounded spins report here instead of hanging.
 * Every sort resets the word itself, whatever route it takes.
 * (The reference checks nothing; D3D12 only warns, GPUSortingD3D12/SweepBase.h:52-53.) */
gs_status gs_onesweep_check(gs_onesweep* h, void* stream);

/* ---- tuning (no reference counterpart at run time; the reference fixes its
 * tile shape with #defines, GPUSortingCUDA/Sort/OneSweep.cu:30-36, and the D3D12
 * tree picks one per device in Tuner.h) -------------------------------------
 * Select one of the compiled tile shapes (threads x keys-per-thread).  Without a
 * choice the library picks: the shape gs_onesweep_partition_size() reports for large
 * sorts (512x32 keys-only and 8-byte values, 1024x16 4-byte values) and 512x16
 * (8192-key tiles) up to 2^23 / 2^24 / 2^25 keys (keys-only / 4-byte / 8-byte values),
 * where it is faster.  512x32, 1024x16 and 512x16 exist for every key and value type
 * (the tuning build libgpusort_tuning.so adds 256x32, 256x16 and 512x20 for uint32 keys).  (gs_onesweep_options::shape_threads / shape_keys_per_thread at create.) */
gs_status gs_onesweep_set_shape(gs_onesweep* h, uint32_t threads, uint32_t keys_per_thread);
uint32_t gs_onesweep_get_partition_size(gs_onesweep* h);
/* Ranking algorithm inside a tile: 0 = 64-lane ballot multi-split (the
 * reference's WLMS, OneSweep.cu:207-253, re-derived for wave64); 1 = one
 * returning LDS atomic per key, valid only if gs_selftest_lds_atomic_order()
 * reports 0 failures on this device. */
gs_status gs_onesweep_set_rank_mode(gs_onesweep* h, int mode);
int gs_onesweep_get_rank_mode(gs_onesweep* h); /* the mode in use (after the create-time probe): 0 or 1; -1 for a null handle */
/* Small inputs are sorted by ONE workgroup in one launch (all four passes in LDS): n <= 8192 in every
 * mode, n <= 16384 for keys-only and 4-byte values, n <= 32768 for keys-only — unless this is switched off
 * (tests use 0 to push small sizes through the tiled path as well). */
gs_status gs_onesweep_set_small_path(gs_onesweep* h, int on);
/* Mid sizes (single-tile limit < n <= 2^20; 4-byte values up to 2^22 pairs, keys-only up to 2^23; 32-bit keys) are
 * sorted in TWO launches instead of seven: one MSD pass on the
 * top byte (its workgroups claim their tiles and adopt the tiles of workgroups that were never dispatched: no residency
 * requirement) and one workgroup per top-byte bucket that sorts the remaining 24
 * bits in LDS; a top byte too skewed for that (a bucket above what the class's workgroup holds: 8192 … 34 816 keys) is noticed on the device and the first kernel
 * runs the four LSD passes itself (reference size sweep GPUSortingD3D12/Tests.h:392-393,415-416).  Same
 * result either way; 0 sends these sizes through the general path.  Only used while the library picks the tile shape.
 * Default 1 (gs_onesweep_options::mid_path at create). */
gs_status gs_onesweep_set_mid_path(gs_onesweep* h, int on);
/* A pass whose digit is the same for every key (e.g. the upper bytes of 16-bit keys) is the
 * identity permutation.  The Scan kernel sees that in the histogram and drops such passes in
 * pairs, on the device, with no host round trip (the reference always runs
 * four passes).  Results are identical either way; 0 runs all four passes.  Default 1;
 * (gs_onesweep_options::skip_passes at create). */
gs_status gs_onesweep_set_skip_passes(gs_onesweep* h, int on);
/* Plan of large sorts of 32-bit keys — keys-only and pairs (library-picked shape, rank mode 1, position chains allowed).
 * The reference's pipeline is GlobalHistogram + Scan + four 8-bit LSD DigitBinningPasses (GPUSortingCUDA/Sort/OneSweep.cu:44-344,
 * dispatch OneSweepDispatcher.cuh:311-336): 36 bytes of memory traffic per key.  The TWO-LEVEL plan (hybrid_kernels.hpp) runs the same
 * kernels in another order — one histogram sweep over the keys' top 16 bits, a Scan, a DigitBinningPass on the top byte, a
 * DigitBinningPass on byte 2 inside the top-byte buckets (256 chains), then one workgroup per 16-bit-prefix bucket sorts the low
 * 16 bits in LDS, in place: 28 bytes per key, the same result bit for bit.  It needs buckets that fit a workgroup (near-uniform top
 * 16 bits); whether they do is decided ON THE DEVICE from the histogram (no host round trip): otherwise the same launches run the
 * four LSD passes on position chains.
 *   0 (default): the two-level plan is offered from 3 x 2^24 (50 M) keys up, to pairs from 2^25 + 1 (measured crossovers);  1: never (the
 *   LSD passes only);  2 (tests): offered at every size from gs_onesweep_options::position_chains_min_log2 up.  A handle that was
 *   created without the plan's tables (below the plan's default size, or with plan 1) gets them allocated by gs_onesweep_set_plan(h, 2)
 *   — call it with no sort of the handle in flight; GS_ERR_MODE for max_keys <= 2^20 (neither the plan nor its fall-back runs there).  Pairs: the values travel with their keys through
 *   both DigitBinningPasses and are moved once more by the bucket-local sort — 52 / 76 bytes per pair instead of 68 / 100.
 * gs_onesweep_last_plan (synchronous) reports what the device decided for the last sort: *plan = 1 the two-level plan ran, 0 the LSD
 * passes (or a one- / two-launch route); *largest_bucket (may be NULL) = the largest 16-bit-prefix bucket it saw (0 if not offered). */
gs_status gs_onesweep_set_plan(gs_onesweep* h, int plan);
gs_status gs_onesweep_last_plan(gs_onesweep* h, uint32_t* plan, uint32_t* largest_bucket, void* stream);
/* Device probe: do same-address lanes of one LDS atomic get their results in
 * ascending lane order?  Synchronous; *h_failures = mismatching lanes. */
gs_status gs_selftest_lds_atomic_order(uint32_t iters, uint32_t seed, uint64_t* h_failures, void* stream);
/* Self-test of the wave-level primitives (wave64 scans by shuffle and by DPP, reduction, 64-bit ballot, mbcnt lane rank, the
 * eight-ballot multi-split) — what the reference builds from PTX and 32-lane *_sync intrinsics in GPUSortingCUDA/Utils.cuh:22-126 and
 * OneSweep.cu:207-253.  Enqueues one kernel: `waves` (a multiple of 4) waves each write 8 rows of 64 words to d_out
 * (waves x 512 uint32); row layout and the input word of a lane: onesweep_kernels.hpp, wave_primitives_kernel.  The caller compares. */
gs_status gs_selftest_wave_primitives(uint32_t seed, uint32_t waves, uint32_t* d_out, void* stream);
/* Instrumented builds only (-DGS_EXP=2, tools/trace_tiles.py): device buffer of 4 passes x grid x 8 words that
 * receives per-tile phase timestamps.  A no-op in the product build. */
gs_status gs_debug_set_trace(gs_onesweep* h, void* d_buf);

/* Debug: post-call invariants of the handle's chained-scan state (the analogue of the reference's
 * ValidateInitialOneSweepState + index checks, GPUSortingCUDA/UtilityKernels.cuh:482-502, which the reference never
 * calls after a sort).  Synchronous.  After any tiled call that completed: report[0] descriptor rows that are not
 * INCLUSIVE, [1] rows whose inclusive count decreased along their chain, [2] chains whose ticket counter is below
 * their tile count, [3] non-zero words left in the histogram region (it must be zero whenever no call is in
 * flight) — all four must be 0 — and [4 + q] the keys the descriptors of the call's q-th pass account for (== n
 * for every pass that ran, 0 for a dropped identity pass).  All zero after a single-tile sort (no scan state). */
gs_status gs_debug_check_state(gs_onesweep* h, uint64_t report[8], void* stream);
/* Test hook: overwrite the device status word gs_onesweep_check() reads (e.g. GS_ERR_TIMEOUT) — every sort must reset it
 * itself, whatever route it takes.  Synchronous. */
gs_status gs_debug_poke_status(gs_onesweep* h, uint32_t word, void* stream);
/* Test / tuning hook: copies `count` words of the handle's state slab, from word `first_word`, to the host.  Synchronous. */
gs_status gs_debug_read_slab(gs_onesweep* h, uint32_t first_word, uint32_t count, uint32_t* h_out, void* stream);

/* ---- structural entry points (parity tests, MSD split) --------------------
 * GlobalHistogram + Scan only (GPUSortingCUDA/Sort/OneSweep.cu:44-162): writes
 * the four 256-bin histograms (counts, not prefixes) to h_hist[1024] on the
 * host.  Synchronous. */
gs_status gs_onesweep_global_histogram(gs_onesweep* h, const void* d_keys, uint32_t n,
                                       gs_key_type key_type, uint32_t* h_hist, void* stream);
/* GlobalHistogram + Scan, and what the Scan kernel left for the passes (GPUSortingCUDA/Sort/OneSweep.cu:125-162, the
 * store of :141-158): h_rows[q * 256 + d] = the RAW descriptor word of digit d in the first row of pass q, which is
 * (exclusive prefix of histogram q at d) << 2 | FLAG_INCLUSIVE (2) — the reference's passHistogram_q[d] before any tile has run.
 * Synchronous.  (The direct parity check of row A2 of SURVEY.md 8a: tests diff it against the CPU restatement of the reference's Scan.) */
gs_status gs_onesweep_scan(gs_onesweep* h, const void* d_keys, uint32_t n, gs_key_type key_type, uint32_t* h_rows,
                           void* stream);
/* One stable DigitBinningPass (OneSweep.cu:164-344 / :346-600) on byte `pass`
 * (0..3) from d_keys_in to d_keys_out (values optional, NULL for keys-only).
 * reverse_index != 0 applies the reference's descending rule to this pass.
 * Self-contained (clears state, histograms, scans, runs the one pass).  With
 * pass = 3 this is the multi-GPU MSD partition step. */
gs_status gs_onesweep_digit_pass(gs_onesweep* h, const void* d_keys_in, void* d_keys_out,
                                 const void* d_vals_in, void* d_vals_out, uint32_t n,
                                 uint32_t pass, gs_key_type key_type, int reverse_index,
                                 void* stream);

/* The multi-GPU MSD split in two steps that share ONE histogram + scan of the shard (no reference
 * counterpart).  prepare: top-byte histogram of d_keys[0..n) to h_hist256[256] on the
 * host (synchronous).  partition: the stable DigitBinningPass on the top byte of the SAME buffer and n;
 * must be the next call on this handle (GS_ERR_ARG otherwise). */
gs_status gs_onesweep_msd_prepare(gs_onesweep* h, const void* d_keys, uint32_t n, gs_key_type key_type,
                                  uint32_t* h_hist256, void* stream);
gs_status gs_onesweep_msd_partition(gs_onesweep* h, const void* d_keys_in, void* d_keys_out,
                                    const void* d_vals_in, void* d_vals_out, uint32_t n, void* stream);

/* ---- profiling hook --------------------------------------------------------
 * Replaces the cudaEvent pair of BatchTiming* (OneSweepDispatcher.cuh:207-229)
 * with per-kernel HIP events recorded on the sort's own stream.
 * Slots: 0 state clear (0 since the clear is folded into the GlobalHistogram kernel), 1 GlobalHistogram,
 * 2 Scan, 3..6 DigitBinningPass 0..3, 7 whole sort. */
#define GS_PROFILE_SLOTS 8
gs_status gs_onesweep_set_profiling(gs_onesweep* h, int enabled);
/* Synchronises the last profiled sort and returns milliseconds per slot. */
gs_status gs_onesweep_get_profile(gs_onesweep* h, float ms[GS_PROFILE_SLOTS]);

/* ---- fixtures exported for parity tests ------------------------------------
 * Replaces: InitRandom<<<256,256>>> keys / pairs (GPUSortingCUDA/UtilityKernels.cuh:53-117).
 * d_vals may be NULL; value_bytes 0/4/8 (value = key, zero-extended for 8). */
gs_status gs_init_random(void* d_keys, void* d_vals, uint32_t value_bytes, uint32_t and_count,
                         uint32_t seed, uint32_t n, void* stream);
/* Replaces: Validate keys / pairs (UtilityKernels.cuh:402-479) + the 4-byte
 * read-back of DispatchValidateKeys/Pairs (OneSweepDispatcher.cuh:365-391),
 * order/type-aware as GPUSortingD3D12/Shaders/Utility.hlsl:147-230.
 * Synchronous; *h_err_count = number of adjacent inversions.  64-bit key types: the keys' order only (d_vals ignored). */
gs_status gs_validate(const void* d_keys, const void* d_vals, uint32_t value_bytes, uint32_t n,
                      gs_key_type key_type, gs_order order, uint32_t* h_err_count, void* stream);

/* ---- multi-GPU MSD split helper (host only; no reference counterpart) ------
 * From the all-reduced top-byte histogram pick the first top-byte bin each of
 * `world` ranks owns: first_bin[0] = 0 ... first_bin[world] = 256. */
gs_status gs_msd_splitters(const uint64_t hist256[256], uint32_t world, uint32_t* first_bin);
/* The same over any number of bins (first_bin[world] = nbins). */
gs_status gs_msd_splitters_n(const uint64_t* hist, uint32_t nbins, uint32_t world, uint32_t* first_bin);
/* Finer split for skewed shards (top-byte buckets can overflow): the 4096-bin histogram of the
 * 12-bit key prefix, bin = top_byte * 16 + (next byte >> 4), read back to the host (synchronous).  A shard sorted
 * by its top two bytes (two gs_onesweep_digit_pass calls: pass 2, then pass 3) is contiguous in that prefix. */
gs_status gs_onesweep_msd_fine_histogram(gs_onesweep* h, const void* d_keys, uint32_t n, gs_key_type key_type,
                                         uint32_t* h_hist4096, void* stream);

/* ---- multi-GPU: one process per GPU, MSD bucket split + RCCL exchange + per-GPU OneSweep -------------------
 * BASELINE.json configs[3].  No reference counterpart (the reference is single-GPU; SURVEY.md 5.8, 8b proposed
 * gs_onesweep_sort_sharded(gs_mgpu*, ...)).  Every rank of the job owns one gs_mgpu context; all calls below that
 * say "collective" must be made by every rank, in the same order.
 *
 * Bootstrap as with NCCL: one rank calls gs_mgpu_get_unique_id and hands the 128 bytes to every rank over whatever
 * it has (MPI, torch.distributed, a file); then every rank calls gs_mgpu_create (collective: ncclCommInitRank).
 * RCCL (librccl.so.1) is loaded on first use; libgpusort.so does not depend on it otherwise. */
#define GS_MGPU_UNIQUE_ID_BYTES 128
typedef struct gs_mgpu gs_mgpu;
gs_status gs_mgpu_get_unique_id(uint8_t id[GS_MGPU_UNIQUE_ID_BYTES]);
/* shard_keys: most keys a rank passes in; capacity (>= shard_keys): most keys a rank can receive (its bucket of the
 * global result; 1.25 x shard_keys is plenty for uniform keys).  Allocates the local sorter's scan state and one
 * scratch array per key/value array (max(shard_keys, capacity) elements).  value_bytes 0 / 4 / 8 as gs_onesweep_create. */
gs_status gs_mgpu_create(gs_mgpu** out, const uint8_t id[GS_MGPU_UNIQUE_ID_BYTES], uint32_t rank, uint32_t world,
                         uint32_t shard_keys, uint32_t capacity, gs_mode mode, uint32_t value_bytes);
/* Options of a sharded-sort context (the library reads no environment variables; gs_mgpu_create uses the defaults). */
typedef struct gs_mgpu_options {
    uint32_t struct_size;   /* sizeof(gs_mgpu_options) */
    int32_t force_exchange; /* 0 (default); 1: a single rank runs partition + exchange too (tests: world == 1 normally just sorts) */
    int32_t overlap;        /* 1 (default): pairs send their values on a second stream / communicator behind the keys; 0: one group */
    int32_t alltoallv;      /* 0 (default): grouped ncclSend / ncclRecv; 1: ncclAllToAllv (also switchable later: gs_mgpu_set_alltoallv) */
    int32_t by_bin;         /* 1 (default): the grouped exchange goes one message per (peer, top byte), so that a bucket offered the two-level plan
                               is landed bin-major and its local sort starts at the plan's second pass (gs_mgpu_last_layout); 0: one message
                               per peer, source-major landing, full local sort (round 5's exchange; what ncclAllToAllv and the 12-bit split always use) */
    gs_onesweep_options sorter;  /* options of the context's local sorter (gs_mgpu_sorter); struct_size 0 = defaults */
} gs_mgpu_options;
void gs_mgpu_options_default(gs_mgpu_options* o);
gs_status gs_mgpu_create_ex(gs_mgpu** out, const uint8_t id[GS_MGPU_UNIQUE_ID_BYTES], uint32_t rank, uint32_t world,
                            uint32_t shard_keys, uint32_t capacity, gs_mode mode, uint32_t value_bytes, const gs_mgpu_options* options);
/* The bucket exchange of the RCCL transport: 0 grouped ncclSend / ncclRecv, 1 ncclAllToAllv.  Every rank must choose alike.
 * Takes effect from the next gs_onesweep_sort_sharded (a harness can time both on one context). */
gs_status gs_mgpu_set_alltoallv(gs_mgpu* ctx, int on);
gs_status gs_mgpu_destroy(gs_mgpu* ctx);
/* The sorted array is the concatenation over ranks of what each rank gets back.  Collective.  d_keys[0..n) (and
 * d_vals) is this rank's shard (n may be 0; unchanged on return); d_out_keys / d_out_vals are caller-owned buffers
 * of `capacity` elements that receive this rank's contiguous range of the global result, *out_n its length.  Stable
 * (received order = source rank, source position).  Asynchronous on `stream` except for ONE wait on a few dozen
 * words (per-peer counts: RCCL's send/recv take them as host integers).  Pairs: the values travel on a second stream and (RCCL)
 * a second communicator behind the keys, and the local sort's GlobalHistogram + Scan run on the received keys meanwhile;
 * gs_mgpu_options::overlap = 0 keeps keys and values in one group, ::alltoallv = 1 / gs_mgpu_set_alltoallv uses ncclAllToAllv
 * instead of grouped send / recv.  GS_ERR_SIZE on EVERY rank if a bucket does not fit `capacity` even at
 * 12-bit-prefix granularity; GS_ERR_COMM on every rank if some rank failed before the histogram gather (see gs_mgpu_check for
 * failures after it).  ARGUMENT errors (GS_ERR_ARG / GS_ERR_SIZE / GS_ERR_MODE: null or misaligned pointers, n > shard_keys, a
 * key type other than the three 32-bit ones) are returned BEFORE the first collective, on the calling rank only: like the
 * arguments of any collective they must be valid on every rank or on none.  64-bit keys are not accepted here: the split
 * works on the top byte of a 32-bit key and the exchange moves 4-byte keys; sort 64-bit keys per GPU (gs_onesweep_sort_keys). */
gs_status gs_onesweep_sort_sharded(gs_mgpu* ctx, const void* d_keys, const void* d_vals, uint32_t n, gs_key_type key_type,
                                   void* d_out_keys, void* d_out_vals, uint32_t* out_n, void* stream);
/* Synchronises `stream` and reports the last call's outcome: GS_ERR_COMM if SOME rank carried an error of its own through the
 * exchange (every rank's status is all-gathered at the end of the call: its peers were served, its own result is not there, and
 * the global result is incomplete), otherwise gs_onesweep_check() of the local sorter.  (A rank that fails BEFORE the histogram
 * gather poisons its row instead, and every rank's gs_onesweep_sort_sharded returns GS_ERR_COMM at once.)  If the last
 * gs_onesweep_sort_sharded itself returned an error on this rank, the gathered words on the device belong to an EARLIER call: nothing
 * is read, GS_ERR_COMM is returned and the context stays marked as failed (its teardown aborts the communicators) until a later
 * call has completed cleanly on every rank. */
gs_status gs_mgpu_check(gs_mgpu* ctx, void* stream);
/* Test hook: the next gs_onesweep_sort_sharded on this rank fails on its own — 1: before the histogram gather, 2: after the plan
 * (as if a HIP launch had failed) — to exercise the failure agreement with the peers.  0 clears it. */
gs_status gs_mgpu_debug_fail(gs_mgpu* ctx, int where);
/* Phase times of the last gs_onesweep_sort_sharded on this rank (HIP events on its stream; synchronises):
 * ms[0] split (histogram + all-gather + plan + the host wait + partition pass), ms[1] bucket exchange,
 * ms[2] local sort, ms[3] total; bytes this rank sent to / received from OTHER ranks; whether the 12-bit split ran. */
gs_status gs_mgpu_get_profile(gs_mgpu* ctx, float ms[4], uint64_t* bytes_sent, uint64_t* bytes_received, uint32_t* fine_split);
/* The exchange plan of the last call (host memory, no synchronisation): [0] keys received, [1] overflow, [2] largest bucket,
 * [3] 1 if the split ran at the 12-bit prefix, then
 * send_counts[world], recv_counts[world], first_bin[world + 1]; `words` >= 4 + 3 * world + 1. */
gs_status gs_mgpu_last_plan(gs_mgpu* ctx, uint32_t* plan, uint32_t words);
gs_onesweep* gs_mgpu_sorter(gs_mgpu* ctx);               /* the local engine (tuning switches, gs_onesweep_check) */
gs_status gs_mgpu_set_force_exchange(gs_mgpu* ctx, int on); /* run split + exchange even with one rank (tests) */
/* How the last gs_onesweep_sort_sharded landed its bucket (host state, no synchronisation): *bin_major = 1 — the bucket exchange went
 * one message per (peer, top byte) and this rank placed the segments top byte by top byte in the local sort's alternate buffer, so the
 * local sort started at the two-level plan's second pass (the sender's split WAS its top-byte partition); 0 — source by source in the
 * output buffer, full local sort (buckets below the two-level plan's size, the 12-bit split, ncclAllToAllv, world 1 without exchange). */
gs_status gs_mgpu_last_layout(gs_mgpu* ctx, uint32_t* bin_major);
int gs_last_rccl_error(void);                             /* ncclResult_t of the last failing RCCL call on this thread */

/* The transport the pipeline runs on: RCCL by default; tests run several ranks on ONE GPU over a host-staged one.
 * Both functions take device pointers and either enqueue on `stream` or complete before returning; 0 = success.
 * exchange: for every array a < n_arrays and every peer p, send_counts[p] elements of elem_bytes[a] bytes from
 * d_send[a] + send_displs[p] go to peer p, which receives them at d_recv[a] + recv_displs[self]; counts and
 * displacements are host arrays of `world` entries, shared by all arrays. */
typedef struct gs_mgpu_transport {
    void* user;
    int (*all_gather_u32)(void* user, const void* d_send, void* d_recv, size_t count, void* stream);
    int (*exchange)(void* user, uint32_t n_arrays, const void* const* d_send, void* const* d_recv, const uint32_t* elem_bytes,
                    const uint32_t* send_counts, const uint32_t* send_displs, const uint32_t* recv_counts,
                    const uint32_t* recv_displs, void* stream);
} gs_mgpu_transport;
gs_status gs_mgpu_create_with_transport_ex(gs_mgpu** out, const gs_mgpu_transport* transport, uint32_t rank, uint32_t world,
                                           uint32_t shard_keys, uint32_t capacity, gs_mode mode, uint32_t value_bytes, const gs_mgpu_options* options);
gs_status gs_mgpu_create_with_transport(gs_mgpu** out, const gs_mgpu_transport* transport, uint32_t rank, uint32_t world,
                                        uint32_t shard_keys, uint32_t capacity, gs_mode mode, uint32_t value_bytes);
/* The plan as a host function (same rule as the device kernel; CPU tests, other transports): table[src * nbins + b] =
 * keys of rank src in MSD bin b. */
gs_status gs_msd_plan(const uint32_t* table, uint32_t nbins, uint32_t world, uint32_t rank, uint32_t capacity, uint32_t* plan);
/* One round of the per-(peer, top byte) bucket exchange as gs_onesweep_sort_sharded runs it (host function, no GPU: CPU tests, other
 * transports).  table[src * 256 + b] = keys of rank src under top byte b (the gathered table of the coarse split); first_bin[world + 1] =
 * the plan's splitters.  Round `round` carries, for every peer p, byte first_bin[p] + round of p's range: send_counts / send_displs[p]
 * (elements, in this rank's shard grouped by top byte) and recv_counts / recv_displs[q] for this rank's own byte first_bin[rank] + round
 * from every source q — landed bin-major (bin_major != 0: byte by byte, sources in rank order inside a byte = the stable top-byte
 * partition of the concatenated sources) or source-major.  *rounds (may be NULL) = rounds of a call = the widest rank's byte count. */
gs_status gs_msd_exchange_round(const uint32_t* table, uint32_t world, uint32_t rank, const uint32_t* first_bin, int bin_major, uint32_t round,
                                uint32_t* send_counts, uint32_t* send_displs, uint32_t* recv_counts, uint32_t* recv_displs, uint32_t* rounds);
/* Test hook: the device plan kernel on a host table (nbins 256 or 4096); synchronous. */
gs_status gs_debug_msd_plan_device(const uint32_t* h_table, uint32_t nbins, uint32_t world, uint32_t rank, uint32_t capacity,
                                   uint32_t* h_plan, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPUSORT_H */
