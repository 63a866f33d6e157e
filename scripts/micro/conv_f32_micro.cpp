// conv_f32_micro.cpp -- times frcnn_conv_f32_ex (libfrcnn_hip.so: the fp32 MFMA convolution, inference forms: ReLU, or ReLU + fused 2x2
// max-pool where VGG-16 pools) on the VGG-16 layer shapes at 600 x 1000 without torch: per layer a captured graph of 5 back-to-back
// launches (outputs rotating over 3 buffers), bursts of graph launches between two events.  Usage: conv_f32_micro [layer ...].
// Values: uniform fp32 in [-1, 1) (weights x 0.05); results are not checked here (tests/ do that).
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

struct Layer { const char *name; int ci, co, h, w, pool, times; };
static const Layer kLayers[] = {
    {"conv1_2", 64, 64, 600, 1000, 1, 1}, {"conv2_1", 64, 128, 300, 500, 0, 1}, {"conv2_2", 128, 128, 300, 500, 1, 1}, {"conv3_1", 128, 256, 150, 250, 0, 1},
    {"conv3_2", 256, 256, 150, 250, 0, 1}, {"conv3_3", 256, 256, 150, 250, 1, 1}, {"conv4_1", 256, 512, 75, 125, 0, 1}, {"conv4_2", 512, 512, 75, 125, 0, 1},
    {"conv4_3", 512, 512, 75, 125, 1, 1}, {"conv5_1", 512, 512, 38, 63, 0, 4}};

int main(int argc, char **argv) {
    std::vector<std::string> want;
    for (int i = 1; i < argc; ++i) want.push_back(argv[i]);
    hipStream_t s; CK(hipStreamCreate(&s));
    const int burst = getenv("CONV_MICRO_BURST") ? atoi(getenv("CONV_MICRO_BURST")) : 8;
    std::mt19937 g(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    double total = 0, total_gflop = 0;
    for (const Layer &L : kLayers) {
        if (!want.empty() && std::find(want.begin(), want.end(), std::string(L.name)) == want.end()) continue;
        const size_t nx = (size_t)L.ci * L.h * L.w, nw = (size_t)9 * L.co * L.ci, ny = (size_t)L.co * L.h * L.w;
        std::vector<float> hx(nx), hw(nw);
        for (auto &e : hx) e = u(g);
        for (auto &e : hw) e = 0.05f * u(g);
        float *dx, *dw, *dy[3], *db; void *ws;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, L.co * 4)); CK(hipMemset(db, 0, L.co * 4));
        for (auto &p : dy) CK(hipMalloc(&p, ny * 4));
        const size_t wsb = frcnn_conv3x3_workspace_bytes(L.ci, L.co, L.h, L.w);
        CK(hipMalloc(&ws, wsb));
        if (frcnn_conv3x3_workspace_init(ws, wsb, s) != 0) { printf("workspace init failed\n"); return 1; }
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        const double gflop = 2.0 * L.h * L.w * L.co * L.ci * 9 / 1e9;
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        bool ok = true;
        const char *cfg_env = getenv("CONV_MICRO_CFG");               // force a decomposition (frcnn_conv3x3_f32_cfg: plain ReLU epilogue)
        for (int i = 0; i < 5; ++i)
            ok = ok && (cfg_env ? frcnn_conv3x3_f32_cfg(dx, dw, db, dy[i % 3], L.ci, L.co, L.h, L.w, 1, atoi(cfg_env), ws, wsb, s)
                                : frcnn_conv_f32_ex(dx, dw, db, nullptr, dy[i % 3], L.ci, L.co, L.h, L.w, 3, L.pool ? 4 : 1, ws, wsb, s)) == 0;
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
        printf("%-8s %3d->%3d %4dx%-4d %s %6.1f GFLOP  %7.1f us  %6.1f TFLOP/s\n", L.name, L.ci, L.co, L.h, L.w, L.pool ? "relu+pool" : "relu     ", gflop, med, gflop / med * 1e3);
        total += med * L.times; total_gflop += gflop * L.times;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(ws)); for (auto &p : dy) CK(hipFree(p));
    }
    printf("chain (conv5_1 x 4; conv1_1 not included): %.1f us, %.1f TFLOP/s\n", total, total_gflop / total * 1e3);
    return 0;
}
