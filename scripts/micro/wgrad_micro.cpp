// wgrad_micro.cpp -- times frcnn_conv_wgrad_f32 (libfrcnn_hip.so: the weight-gradient kernel + its slab reduction) on the VGG-16 layer
// shapes at 600 x 1000 without torch: per layer a captured graph of 5 back-to-back calls, bursts of graph launches between two events.
// Usage: wgrad_micro [--f32s] [layer ...]; settings come from the environment (FRCNN_WGRAD_DB, FRCNN_WGRAD_ABL in the
// ablation build ...).  Values: uniform fp32 in [-1, 1); results are not checked here (tests/ do that).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <random>
#include <string>
#include <vector>
#include "frcnn_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Layer { const char *name; int ci, co, h, w, times; };
static const Layer kLayers[] = {
    {"conv1_1", 3, 64, 600, 1000, 1}, {"conv1_2", 64, 64, 600, 1000, 1}, {"conv2_1", 64, 128, 300, 500, 1}, {"conv2_2", 128, 128, 300, 500, 1},
    {"conv3_1", 128, 256, 150, 250, 1}, {"conv3_2", 256, 256, 150, 250, 2}, {"conv4_1", 256, 512, 75, 125, 1}, {"conv4_2", 512, 512, 75, 125, 2},
    {"conv5_1", 512, 512, 38, 63, 4}};                               // conv5_1's shape: conv5_1..3 and rpn_conv_3x3

int main(int argc, char **argv) {
    std::vector<std::string> want;
    bool f32s = false;
    for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "--f32s")) f32s = true; else want.push_back(argv[i]); }
    hipStream_t s; CK(hipStreamCreate(&s));
    const int burst = getenv("WGRAD_MICRO_BURST") ? atoi(getenv("WGRAD_MICRO_BURST")) : 6;
    std::mt19937 g(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    double total = 0, total_gflop = 0;
    for (const Layer &L : kLayers) {
        if (!want.empty() && std::find(want.begin(), want.end(), std::string(L.name)) == want.end()) continue;
        if (f32s && L.ci < 16) continue;
        const size_t nx = (size_t)L.ci * L.h * L.w, ny = (size_t)L.co * L.h * L.w, nw = (size_t)9 * L.ci * L.co;
        std::vector<float> hx(nx), hy(ny);
        for (auto &e : hx) e = u(g);
        for (auto &e : hy) e = u(g);
        float *dx, *dy, *dw; void *ws;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&dw, nw * 4));
        const size_t wsb = frcnn_conv_wgrad_workspace_bytes(L.ci, L.co, L.h, L.w, 3);
        CK(hipMalloc(&ws, wsb));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hy.data(), ny * 4, hipMemcpyHostToDevice));
        const double gflop = 2.0 * L.h * L.w * L.co * L.ci * 9 / 1e9;
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        bool ok = true;
        for (int i = 0; i < 5; ++i)
            ok = ok && (f32s ? frcnn_conv_wgrad_f32s(dx, dy, dw, L.ci, L.co, L.h, L.w, ws, wsb, s)
                             : frcnn_conv_wgrad_f32(dx, dy, dw, L.ci, L.co, L.h, L.w, 3, ws, wsb, s)) == 0;
        CK(hipStreamEndCapture(s, &gr));
        if (!ok) { printf("%s: launch refused\n", L.name); return 1; }
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<float> us;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, s));
            for (int b = 0; b < burst; ++b) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); us.push_back(ms * 200.f / burst);
        }
        std::sort(us.begin(), us.end());
        const double med = us[us.size() / 2];
        printf("%-8s %3d->%3d %4dx%-4d %6.1f GFLOP  %7.1f us  %6.1f TFLOP/s  (workspace %.1f MB)\n", L.name, L.ci, L.co, L.h, L.w, gflop, med, gflop / med * 1e3,
               wsb / 1e6);
        total += med * L.times; total_gflop += gflop * L.times;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(dw)); CK(hipFree(ws));
    }
    printf("chain (conv3_2 x 2, conv4_2 x 2, conv5_1 x 4): %.1f us, %.1f TFLOP/s\n", total, total_gflop / total * 1e3);
    return 0;
}
