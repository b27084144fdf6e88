// GPUSortBase.hpp — header-only C++ host classes with the D3D12 tree's sorter surface
// (GPUSortingD3D12/GPUSorting.h:36-90, GPUSortBase.h:162-275, OneSweep.h:16-27) on top of the C-ABI (gpusort.h):
//
//   GPUSorting::{MODE, ORDER, KEY_TYPE, PAYLOAD_TYPE, ENTROPY_PRESET, GPUSortingConfig}
//   class GPUSortBase   TestSort(testSize, seed, shouldReadBack, shouldValidate)
//                       BatchTiming(inputSize, batchSize, seed, entropyPreset)
//                       TestAll()
//   class OneSweep : GPUSortBase   OneSweep(ORDER, KEY_TYPE)                 keys only
//                                  OneSweep(ORDER, KEY_TYPE, PAYLOAD_TYPE)   pairs
//
// Same method names, argument meaning and printed lines.  What differs, because HIP is not D3D12:
//   * no winrt::com_ptr<ID3D12Device> / DeviceInfo constructor arguments — the sorter runs on the current HIP
//     device; errors are gs_status / hipError_t lines on stderr, never exceptions (D3D12: winrt::check_hresult);
//   * the object owns its buffers and re-creates them only when the size GROWS (the reference re-creates them on
//     every size change, SweepBase.h:157-167);
//   * TestAll's ladder runs over [P, 2P] with THIS build's partition size P (the reference's P comes from its
//     Tuner table), then the same three large sizes (1<<21, 1<<22, 1<<23 with seeds 5, 7, 11);
//   * PAYLOAD_TYPE only names the payload's element type: payloads are bit-copied (SortCommon.hlsl:252-261).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../gpusort.h"

namespace GPUSorting {
typedef enum MODE { MODE_KEYS_ONLY = 0, MODE_PAIRS = 1 } MODE;
typedef enum ORDER { ORDER_ASCENDING = 0, ORDER_DESCENDING = 1 } ORDER;
typedef enum KEY_TYPE { KEY_UINT32 = 0, KEY_INT32 = 1, KEY_FLOAT32 = 2 } KEY_TYPE;
typedef enum PAYLOAD_TYPE { PAYLOAD_UINT32 = 0, PAYLOAD_INT32 = 1, PAYLOAD_FLOAT32 = 2 } PAYLOAD_TYPE;
struct GPUSortingConfig {
    MODE sortingMode;
    ORDER sortingOrder;
    KEY_TYPE sortingKeyType;
    PAYLOAD_TYPE sortingPayloadType;
};
typedef enum ENTROPY_PRESET {
    ENTROPY_PRESET_1 = 0, ENTROPY_PRESET_2 = 1, ENTROPY_PRESET_3 = 2, ENTROPY_PRESET_4 = 3, ENTROPY_PRESET_5 = 4,
} ENTROPY_PRESET;
}  // namespace GPUSorting

class GPUSortBase {
protected:
    const char* k_sortName;
    const GPUSorting::GPUSortingConfig k_sortingConfig;
    const uint32_t k_maxReadBack;
    uint32_t m_numKeys = 0;
    uint32_t m_capacity = 0;
    uint32_t m_partitionSize = 0;
    gs_onesweep* m_sorter = nullptr;
    uint32_t* m_sortBuffer = nullptr;
    uint32_t* m_altBuffer = nullptr;
    uint32_t* m_sortPayloadBuffer = nullptr;
    uint32_t* m_altPayloadBuffer = nullptr;
    hipEvent_t m_start = nullptr, m_stop = nullptr;
    bool m_ok = true;

    bool pairs() const { return k_sortingConfig.sortingMode == GPUSorting::MODE_PAIRS; }
    bool ok(gs_status s, const char* what) {
        if (s != GS_OK) { fprintf(stderr, "gpusort: %s: %s\n", what, gs_status_string(s)); m_ok = false; }
        return s == GS_OK;
    }
    bool ok(hipError_t e, const char* what) {
        if (e != hipSuccess) { fprintf(stderr, "gpusort: %s: %s\n", what, hipGetErrorString(e)); m_ok = false; }
        return e == hipSuccess;
    }
    void Release() {
        if (m_sorter) gs_onesweep_destroy(m_sorter);
        m_sorter = nullptr;
        for (uint32_t** p : {&m_sortBuffer, &m_altBuffer, &m_sortPayloadBuffer, &m_altPayloadBuffer}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
    }
    // SweepBase.h:157-167: buffers follow the size (here they only ever grow: the ladder of TestAll would
    // otherwise re-create the sorter and four device buffers 16 385 times)
    void UpdateSize(uint32_t size) {
        m_numKeys = size;
        if (size <= m_capacity && m_sorter) return;
        Release();
        m_capacity = size;
        ok(gs_onesweep_create(&m_sorter, size, pairs() ? GS_MODE_PAIRS : GS_MODE_KEYS_ONLY, pairs() ? 4u : 0u), "gs_onesweep_create");
        ok(hipMalloc(&m_sortBuffer, (size_t)size * 4), "hipMalloc");
        ok(hipMalloc(&m_altBuffer, (size_t)size * 4), "hipMalloc");
        if (pairs()) {
            ok(hipMalloc(&m_sortPayloadBuffer, (size_t)size * 4), "hipMalloc");
            ok(hipMalloc(&m_altPayloadBuffer, (size_t)size * 4), "hipMalloc");
        }
    }
    void CreateTestInput(uint32_t seed, uint32_t preset = 0) {
        ok(gs_init_random(m_sortBuffer, pairs() ? m_sortPayloadBuffer : nullptr, pairs() ? 4u : 0u, preset, seed, m_numKeys, nullptr),
           "gs_init_random");
    }
    void Sort() {
        const gs_key_type kt = (gs_key_type)k_sortingConfig.sortingKeyType;
        const gs_order order = (gs_order)k_sortingConfig.sortingOrder;
        if (pairs())
            ok(gs_onesweep_sort_pairs(m_sorter, m_sortBuffer, m_sortPayloadBuffer, m_altBuffer, m_altPayloadBuffer, m_numKeys, kt,
                                      order, nullptr), "gs_onesweep_sort_pairs");
        else
            ok(gs_onesweep_sort_keys(m_sorter, m_sortBuffer, m_altBuffer, m_numKeys, kt, order, nullptr), "gs_onesweep_sort_keys");
    }
    // GPUSortBase.h:494-515: error count of the (order- and type-aware) validation kernel
    bool ValidateOutput(bool shouldPrint) {
        uint32_t err = 1;
        ok(gs_validate(m_sortBuffer, pairs() ? m_sortPayloadBuffer : nullptr, pairs() ? 4u : 0u, m_numKeys,
                       (gs_key_type)k_sortingConfig.sortingKeyType, (gs_order)k_sortingConfig.sortingOrder, &err, nullptr),
           "gs_validate");
        ok(gs_onesweep_check(m_sorter, nullptr), "gs_onesweep_check");
        if (shouldPrint) {
            printf("%s", k_sortName);
            PrintSortingConfig(k_sortingConfig);
            if (err) printf("failed at size %u with %u errors. \n", m_numKeys, err);
            else printf("passed at size %u. \n", m_numKeys);
        }
        return err == 0 && m_ok;
    }
    bool ValidateSort(uint32_t size, uint32_t seed) {
        UpdateSize(size);
        CreateTestInput(seed);
        Sort();
        return ValidateOutput(false);
    }
    double TimeSort(uint32_t seed, GPUSorting::ENTROPY_PRESET entropyPreset) {
        CreateTestInput(seed, (uint32_t)entropyPreset);
        ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
        ok(hipEventRecord(m_start, nullptr), "hipEventRecord");
        Sort();
        ok(hipEventRecord(m_stop, nullptr), "hipEventRecord");
        ok(hipEventSynchronize(m_stop), "hipEventSynchronize");
        float ms = 0.f;
        ok(hipEventElapsedTime(&ms, m_start, m_stop), "hipEventElapsedTime");
        return ms * 1e-3;
    }

    void Banner(const char* what) {  // "Beginning <sort><config><what>"
        printf("Beginning %s", k_sortName);
        PrintSortingConfig(k_sortingConfig);
        printf("%s", what);
    }

    GPUSortBase(const char* sortName, GPUSorting::GPUSortingConfig cfg, uint32_t maxReadBack)
        : k_sortName(sortName), k_sortingConfig(cfg), k_maxReadBack(maxReadBack) {
        m_partitionSize = gs_onesweep_partition_size(pairs() ? GS_MODE_PAIRS : GS_MODE_KEYS_ONLY, pairs() ? 4u : 0u);
        ok(hipEventCreate(&m_start), "hipEventCreate");
        ok(hipEventCreate(&m_stop), "hipEventCreate");
    }

public:
    virtual ~GPUSortBase() {
        Release();
        if (m_start) (void)hipEventDestroy(m_start);
        if (m_stop) (void)hipEventDestroy(m_stop);
    }

    // GPUSortBase.h:162-204
    void TestSort(uint32_t testSize, uint32_t seed, bool shouldReadBack, bool shouldValidate) {
        UpdateSize(testSize);
        CreateTestInput(seed);
        Sort();
        if (shouldValidate) ValidateOutput(true);
        if (shouldReadBack) {
            const uint32_t readBackSize = m_numKeys < k_maxReadBack ? m_numKeys : k_maxReadBack;
            std::vector<uint32_t> vecOut(readBackSize);
            ok(hipMemcpy(vecOut.data(), m_sortBuffer, (size_t)readBackSize * 4, hipMemcpyDeviceToHost), "hipMemcpy");
            printf("---------------KEYS---------------\n");
            for (uint32_t i = 0; i < readBackSize; ++i) printf("%u %u \n", i, vecOut[i]);
            if (pairs()) {
                ok(hipMemcpy(vecOut.data(), m_sortPayloadBuffer, (size_t)readBackSize * 4, hipMemcpyDeviceToHost), "hipMemcpy");
                printf("\n \n \n");
                printf("---------------PAYLOADS---------------\n");
                for (uint32_t i = 0; i < readBackSize; ++i) printf("%u %u \n", i, vecOut[i]);
            }
        }
    }

    // GPUSortBase.h:206-235
    void BatchTiming(uint32_t inputSize, uint32_t batchSize, uint32_t seed, GPUSorting::ENTROPY_PRESET entropyPreset) {
        static const float bitsOfEntropy[5] = {1.0f, .811f, .544f, .337f, .201f};  // Thearling-Smith presets
        UpdateSize(inputSize);
        Banner("batch timing test at:\n");
        printf("Size: %u\nEntropy: %f bits\nTest size: %u\n", inputSize, bitsOfEntropy[entropyPreset], batchSize);
        double totalTime = 0.0;
        (void)TimeSort(seed, entropyPreset);  // iteration 0 warms up and is not counted
        printf(".");
        for (uint32_t i = 1; i <= batchSize; ++i) {
            totalTime += TimeSort(i + seed, entropyPreset);
            if ((i & 7) == 0) printf(".");
        }
        printf("\nTotal time elapsed: %f\n", totalTime);
        printf("Estimated speed at %u 32-bit elements: %E keys/sec\n\n", inputSize, inputSize / totalTime * batchSize);
    }

    // GPUSortBase.h:237-275
    virtual bool TestAll() {
        Banner("test all. \n");
        // every remainder class of a partition: sizes P .. 2P with seed = size ...
        uint32_t passed = 0;
        for (uint32_t size = m_partitionSize; size <= 2 * m_partitionSize; ++size) {
            passed += ValidateSort(size, size) ? 1u : 0u;
            if ((size & 127) == 0) printf(".");
        }
        printf("\n%u / %u passed. \n", passed, m_partitionSize + 1);
        // ... then three large sizes with the reference's seeds
        printf("Beginning large size tests\n");
        static const uint32_t large[3][2] = {{1u << 21, 5}, {1u << 22, 7}, {1u << 23, 11}};
        for (const auto& t : large) passed += ValidateSort(t[0], t[1]) ? 1u : 0u;
        const uint32_t expected = m_partitionSize + 1 + 3;
        const bool all = passed == expected;
        printf("%u / %u  %s \n\n", all ? expected : passed, expected, all ? "All tests passed." : "Test failed.");
        return all;
    }

    // same words as GPUSortBase.h:547-583: "keys <type> [payload <type> ]<order> "
    static void PrintSortingConfig(const GPUSorting::GPUSortingConfig& cfg) {
        static const char* const typeName[3] = {"uint32", "int32", "float32"};
        printf("keys %s ", typeName[cfg.sortingKeyType]);
        if (cfg.sortingMode == GPUSorting::MODE_PAIRS) printf("payload %s ", typeName[cfg.sortingPayloadType]);
        printf("%s ", cfg.sortingOrder == GPUSorting::ORDER_ASCENDING ? "ascending" : "descending");
    }
};

// GPUSortingD3D12/OneSweep.h:16-27 (minus the device arguments)
class OneSweep : public GPUSortBase {
public:
    OneSweep(GPUSorting::ORDER sortingOrder, GPUSorting::KEY_TYPE keyType)
        : GPUSortBase("OneSweep ", {GPUSorting::MODE_KEYS_ONLY, sortingOrder, keyType, GPUSorting::PAYLOAD_UINT32}, 8192) {}
    OneSweep(GPUSorting::ORDER sortingOrder, GPUSorting::KEY_TYPE keyType, GPUSorting::PAYLOAD_TYPE payloadType)
        : GPUSortBase("OneSweep ", {GPUSorting::MODE_PAIRS, sortingOrder, keyType, payloadType}, 8192) {}
};
