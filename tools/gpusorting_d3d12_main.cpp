// gpusorting_d3d12_main — the OneSweep part of the reference's D3D12 program against the MI355X-native library:
//   (default)   GPUSortingD3D12.cpp:134-142   OneSweep(asc, uint32, payload uint32): TestAll + BatchTiming(1<<28, 100, 10, preset 1)
//   supertest   Tests.h:6-186    SuperTestOneSweep: {asc, desc} x {uint32, int32, float32 keys} x {uint32, int32, float32 payloads}
//   benchmark   Tests.h:370-418  BenchmarkOneSweep: entropy sweep at 2^28 and size sweep 2^10..2^27, keys then pairs
// Usage: gpusorting_d3d12_main [default|supertest|benchmark] [log2_size=28] [batch=100]
#include <stdlib.h>
#include <string.h>

#include "gpusort/GPUSortBase.hpp"

static void SuperTestOneSweep() {
    const uint32_t testsExpected = 18;
    uint32_t testsPassed = 0;
    const GPUSorting::KEY_TYPE keys[3] = {GPUSorting::KEY_UINT32, GPUSorting::KEY_INT32, GPUSorting::KEY_FLOAT32};
    const GPUSorting::PAYLOAD_TYPE payloads[3] = {GPUSorting::PAYLOAD_UINT32, GPUSorting::PAYLOAD_FLOAT32, GPUSorting::PAYLOAD_INT32};
    for (GPUSorting::KEY_TYPE k : keys)
        for (GPUSorting::PAYLOAD_TYPE p : payloads)
            for (GPUSorting::ORDER o : {GPUSorting::ORDER_ASCENDING, GPUSorting::ORDER_DESCENDING}) {
                OneSweep* oneSweep = new OneSweep(o, k, p);
                testsPassed += oneSweep->TestAll();
                delete oneSweep;
            }
    printf("\n");
    printf("\n---------------------------------------------------------");
    printf("\n-------------------ONESWEEP SUPER TEST-------------------");
    printf("\n---------------------------------------------------------\n");
    if (testsPassed == testsExpected) printf("%u / %u ONESWEEP SUPER TEST PASSED!\n", testsPassed, testsExpected);
    else printf("%u / %u ONESWEEP SUPER TEST FAILED!\n", testsPassed, testsExpected);
}

static void BenchmarkOneSweep(uint32_t lg, uint32_t batch) {
    for (int pairs = 0; pairs < 2; ++pairs) {
        printf("---------------------------------------------------------");
        printf("\n--------------ONESWEEP %s ENTROPY SWEEP---------------", pairs ? "PAIRS" : "KEYS ");
        printf("\n---------------------------------------------------------\n");
        OneSweep* s = pairs ? new OneSweep(GPUSorting::ORDER_ASCENDING, GPUSorting::KEY_UINT32, GPUSorting::PAYLOAD_UINT32)
                            : new OneSweep(GPUSorting::ORDER_ASCENDING, GPUSorting::KEY_UINT32);
        s->TestAll();
        for (int e = 0; e < 5; ++e) s->BatchTiming(1u << lg, batch, 10, (GPUSorting::ENTROPY_PRESET)e);
        printf("\n---------------------------------------------------------");
        printf("\n----------------ONESWEEP %s SIZE SWEEP----------------", pairs ? "PAIRS" : "KEYS ");
        printf("\n---------------------------------------------------------\n");
        for (uint32_t i = 10; i < lg; ++i) s->BatchTiming(1u << i, batch, 10, GPUSorting::ENTROPY_PRESET_1);
        delete s;
    }
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "default";
    const uint32_t lg = argc > 2 ? (uint32_t)atoi(argv[2]) : 28u;
    const uint32_t batch = argc > 3 ? (uint32_t)atoi(argv[3]) : 100u;
    if (!strcmp(what, "supertest")) {
        SuperTestOneSweep();
    } else if (!strcmp(what, "benchmark")) {
        BenchmarkOneSweep(lg, batch);
    } else {
        OneSweep* oneSweep = new OneSweep(GPUSorting::ORDER_ASCENDING, GPUSorting::KEY_UINT32, GPUSorting::PAYLOAD_UINT32);
        const bool pass = oneSweep->TestAll();
        oneSweep->BatchTiming(1u << lg, batch, 10, GPUSorting::ENTROPY_PRESET_1);
        delete oneSweep;
        return pass ? 0 : 1;
    }
    return 0;
}
