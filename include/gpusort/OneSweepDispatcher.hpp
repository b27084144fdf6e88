// OneSweepDispatcher.hpp — header-only C++ host class with the CUDA tree's
// OneSweepDispatcher surface (GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:17-392)
// on top of the C-ABI (gpusort.h).  Same constructor, same four public methods,
// same argument meaning, same printed lines, so the reference's main()
// (GPUSortingCUDA/GPUSortingCUDA.cu:20-23,36-39) compiles unchanged against it.
//
// Error behaviour: the reference ignores every runtime error; here a failing
// gs_status / hipError prints one line to stderr and the test counts as failed.
// Deliberate differences:
//   * BatchTimingPairs initialises the payload (the reference times with an
//     uninitialised payload buffer, OneSweepDispatcher.cuh:269-273);
//   * the ladder of TestAll* runs over [P, 2P] with THIS build's partition size P.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>

#include "../gpusort.h"

typedef gs_entropy_preset ENTROPY_PRESET;
#define ENTROPY_PRESET_1 GS_ENTROPY_PRESET_1
#define ENTROPY_PRESET_2 GS_ENTROPY_PRESET_2
#define ENTROPY_PRESET_3 GS_ENTROPY_PRESET_3
#define ENTROPY_PRESET_4 GS_ENTROPY_PRESET_4
#define ENTROPY_PRESET_5 GS_ENTROPY_PRESET_5

class OneSweepDispatcher {
    const bool k_keysOnly;
    const uint32_t k_maxSize;
    uint32_t k_partitionSize = 0;
    gs_onesweep* m_sorter = nullptr;
    uint32_t* m_sort = nullptr;
    uint32_t* m_alt = nullptr;
    uint32_t* m_sortPayload = nullptr;
    uint32_t* m_altPayload = nullptr;
    bool m_ok = true;

    bool ok(gs_status s, const char* what) {
        if (s != GS_OK) { fprintf(stderr, "gpusort: %s: %s\n", what, gs_status_string(s)); m_ok = false; }
        return s == GS_OK;
    }
    bool ok(hipError_t e, const char* what) {
        if (e != hipSuccess) { fprintf(stderr, "gpusort: %s: %s\n", what, hipGetErrorString(e)); m_ok = false; }
        return e == hipSuccess;
    }
    void Generate(bool pairs, uint32_t preset, uint32_t seed, uint32_t size) {
        ok(gs_init_random(m_sort, pairs ? m_sortPayload : nullptr, pairs ? 4u : 0u, preset, seed, size, nullptr), "gs_init_random");
    }
    void Dispatch(bool pairs, uint32_t size) {
        if (pairs)
            ok(gs_onesweep_sort_pairs(m_sorter, m_sort, m_sortPayload, m_alt, m_altPayload, size, GS_KEY_UINT32,
                                      GS_ORDER_ASCENDING, nullptr), "gs_onesweep_sort_pairs");
        else
            ok(gs_onesweep_sort_keys(m_sorter, m_sort, m_alt, size, GS_KEY_UINT32, GS_ORDER_ASCENDING, nullptr),
               "gs_onesweep_sort_keys");
    }
    bool Validate(bool pairs, uint32_t size) {
        uint32_t err = 1;
        ok(gs_validate(m_sort, pairs ? m_sortPayload : nullptr, pairs ? 4u : 0u, size, GS_KEY_UINT32, GS_ORDER_ASCENDING,
                       &err, nullptr), "gs_validate");
        return err == 0;
    }
    void TestAll(bool pairs) {
        if (k_maxSize < (1u << 28)) {
            printf("This test requires a minimum initialized size of %u. ", 1 << 28);
            printf("Reinitialize the object to at least %u.\n", 1 << 28);
            return;
        }
        if (pairs && k_keysOnly) { printf("Error, object was intialized for keys only"); return; }
        printf("Beginning GPUSorting OneSweep %s validation test: \n", pairs ? "pairs" : "keys");
        uint32_t passed = 0;
        for (uint32_t i = k_partitionSize; i <= 2 * k_partitionSize; ++i) {
            Generate(pairs, ENTROPY_PRESET_1, i, i);
            Dispatch(pairs, i);
            if (Validate(pairs, i)) ++passed; else printf("\n Test failed at size %u \n", i);
            if (!(i & 255)) printf(".");
        }
        printf("\n");
        for (uint32_t e = 26; e <= 28; ++e) {
            Generate(pairs, ENTROPY_PRESET_1, e, 1u << e);
            Dispatch(pairs, 1u << e);
            if (Validate(pairs, 1u << e)) ++passed; else printf("\n Test failed at size %u \n", 1u << e);
        }
        if (gs_onesweep_check(m_sorter, nullptr) != GS_OK) { printf("\n Device reported a look-back timeout \n"); passed = 0; }
        const uint32_t expected = k_partitionSize + 3 + 1;
        if (passed == expected) printf("%u/%u All tests passed.\n\n", passed, passed);
        else printf("%u/%u Test failed.\n\n", passed, expected);
    }
    void BatchTiming(bool pairs, uint32_t size, uint32_t batchCount, uint32_t seed, ENTROPY_PRESET entropyPreset) {
        if (pairs && k_keysOnly) { printf("Error, object was intialized for keys only"); return; }
        if (size > k_maxSize) { printf("Error, requested test size exceeds max initialized size. \n"); return; }
        const float entLookup[5] = {1.0f, .811f, .544f, .337f, .201f};
        printf("Beginning GPUSorting OneSweep %s batch timing test at:\n", pairs ? "pairs" : "keys");
        printf("Size: %u\n", size);
        printf("Entropy: %f bits\n", entLookup[entropyPreset]);
        printf("Test size: %u\n", batchCount);
        hipEvent_t start, stop;
        ok(hipEventCreate(&start), "hipEventCreate");
        ok(hipEventCreate(&stop), "hipEventCreate");
        float totalTime = 0.0f;
        for (uint32_t i = 0; i <= batchCount; ++i) {
            Generate(pairs, entropyPreset, i + seed, size);
            ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
            ok(hipEventRecord(start, nullptr), "hipEventRecord");
            Dispatch(pairs, size);
            ok(hipEventRecord(stop, nullptr), "hipEventRecord");
            ok(hipEventSynchronize(stop), "hipEventSynchronize");
            float millis = 0.0f;
            ok(hipEventElapsedTime(&millis, start, stop), "hipEventElapsedTime");
            if (i) totalTime += millis;
            if ((i & 15) == 0) printf(". ");
        }
        printf("\n");
        totalTime /= 1000.0f;
        printf("Total time elapsed: %f\n", totalTime);
        printf("Estimated speed at %u 32-bit elements: %E keys/sec\n\n", size, size / totalTime * batchCount);
        (void)hipEventDestroy(start);
        (void)hipEventDestroy(stop);
    }

public:
    OneSweepDispatcher(bool keysOnly, uint32_t maxSize) : k_keysOnly(keysOnly), k_maxSize(maxSize) {
        ok(gs_onesweep_create(&m_sorter, maxSize, keysOnly ? GS_MODE_KEYS_ONLY : GS_MODE_PAIRS, keysOnly ? 0u : 4u),
           "gs_onesweep_create");
        k_partitionSize = m_sorter ? gs_onesweep_get_partition_size(m_sorter) : 0;
        ok(hipMalloc((void**)&m_sort, (size_t)maxSize * sizeof(uint32_t)), "hipMalloc");
        ok(hipMalloc((void**)&m_alt, (size_t)maxSize * sizeof(uint32_t)), "hipMalloc");
        if (!keysOnly) {
            ok(hipMalloc((void**)&m_sortPayload, (size_t)maxSize * sizeof(uint32_t)), "hipMalloc");
            ok(hipMalloc((void**)&m_altPayload, (size_t)maxSize * sizeof(uint32_t)), "hipMalloc");
        }
    }
    ~OneSweepDispatcher() {
        // the reference's main() calls the destructor explicitly and never deletes: be idempotent
        if (m_sorter) { gs_onesweep_destroy(m_sorter); m_sorter = nullptr; }
        uint32_t** bufs[4] = {&m_sort, &m_alt, &m_sortPayload, &m_altPayload};
        for (auto b : bufs) if (*b) { (void)hipFree(*b); *b = nullptr; }
    }
    bool Healthy() const { return m_ok; }

    void TestAllKeysOnly() { TestAll(false); }
    void TestAllPairs() { TestAll(true); }
    void BatchTimingKeysOnly(uint32_t size, uint32_t batchCount, uint32_t seed, ENTROPY_PRESET entropyPreset) {
        BatchTiming(false, size, batchCount, seed, entropyPreset);
    }
    void BatchTimingPairs(uint32_t size, uint32_t batchCount, uint32_t seed, ENTROPY_PRESET entropyPreset) {
        BatchTiming(true, size, batchCount, seed, entropyPreset);
    }
};
