// linear_bf16_micro.cpp -- times frcnn_linear_bf16 (the GEMM + its split-K reduction) on the RCNN head's shapes without torch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "frcnn_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main() {
    struct { const char *name; int M, N, K; } shapes[] = {{"fc6", 300, 4096, 25088}, {"fc7", 300, 4096, 4096}, {"fc6 (ResNet)", 300, 4096, 100352}};
    hipStream_t s; CK(hipStreamCreate(&s));
    for (auto &L : shapes) {
        uint16_t *x, *w; float *b; uint16_t *y[3]; void *ws;
        CK(hipMalloc(&x, (size_t)L.M * L.K * 2)); CK(hipMalloc(&w, (size_t)L.N * L.K * 2)); CK(hipMalloc(&b, L.N * 4));
        CK(hipMemset(x, 0x3c, (size_t)L.M * L.K * 2)); CK(hipMemset(w, 0x3c, (size_t)L.N * L.K * 2)); CK(hipMemset(b, 0, L.N * 4));   // 0x3c3c: a small positive bf16
        for (auto &p : y) CK(hipMalloc(&p, (size_t)L.M * L.N * 2));
        const size_t wsb = frcnn_linear_bf16_workspace_bytes(L.M, L.N, L.K);
        CK(hipMalloc(&ws, wsb));
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        bool ok = true;
        for (int i = 0; i < 4; ++i) ok = ok && frcnn_linear_bf16(x, w, b, y[i % 3], L.M, L.N, L.K, 1, 1, ws, wsb, s) == 0;
        CK(hipStreamEndCapture(s, &gr));
        if (!ok) { printf("%s: refused\n", L.name); continue; }
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<float> us;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0, s));
            for (int q = 0; q < 5; ++q) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); us.push_back(ms * 1000.f / 20);
        }
        std::sort(us.begin(), us.end());
        const double gf = 2.0 * L.M * L.N * L.K / 1e9, med = us[us.size() / 2];
        printf("%-14s %4d x %6d x %5d  %7.1f us  %6.0f TFLOP/s  weights %.0f MB at %.2f TB/s\n", L.name, L.M, L.K, L.N, med, gf / med * 1e3, (double)L.N * L.K * 2 / 1e6,
               (double)L.N * L.K * 2 / med / 1e6);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(b)); CK(hipFree(ws)); for (auto &p : y) CK(hipFree(p));
    }
    return 0;
}
