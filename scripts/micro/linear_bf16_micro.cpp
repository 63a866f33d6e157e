// linear_bf16_micro.cpp -- times the bf16 FC layer of the RCNN head without torch, on RANDOM operands (constant operands flatter the matrix pipes:
// DESIGN 3.8e): frcnn_linear_bf16 (round-5 kernel, row-major weights) next to frcnn_linear_bf16_tiled (round 6: the weight stream on pre-tiled
// weights, csrc/linear_bf16.hip), each as GEMM + split-K reduction, and the largest difference between their outputs.
// Usage: linear_bf16_micro [old|new|both]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "frcnn_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

template <typename F>
static double time_graph(hipStream_t s, F &&enqueue, int per_graph) {
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < per_graph; ++i) enqueue(i);
    CK(hipStreamEndCapture(s, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    std::vector<float> us;
    for (int r = 0; r < 9; ++r) {
        CK(hipEventRecord(e0, s));
        for (int q = 0; q < 5; ++q) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); us.push_back(ms * 1000.f / (5 * per_graph));
    }
    std::sort(us.begin(), us.end());
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
    return us[us.size() / 2];
}

int main(int argc, char **argv) {
    const char *which = argc > 1 ? argv[1] : "both";
#ifdef LINEAR_MICRO_NEW_ONLY          // linked against liblinear_abl.so (csrc/linear_bf16.hip alone, timing ablations): the round-5 kernel is not in it
    const bool do_old = false, do_new = true;
    (void)which;
#define frcnn_linear_bf16(...) 0
#define frcnn_linear_bf16_workspace_bytes(...) 256
#else
    const bool do_old = strcmp(which, "new") != 0, do_new = strcmp(which, "old") != 0;
#endif
    const int nshapes_env = getenv("LINEAR_MICRO_SHAPES") ? atoi(getenv("LINEAR_MICRO_SHAPES")) : 4;
    struct { const char *name; int M, N, K; } shapes[] = {{"fc6", 300, 4096, 25088}, {"fc7", 300, 4096, 4096}, {"cls||bbox", 300, 116, 4096}, {"fc6 (ResNet)", 300, 4096, 100352},
                                                          {"fc6 K+32", 300, 4096, 25120}, {"fc6 K+64", 300, 4096, 25152}, {"fc6 K=24576", 300, 4096, 24576}, {"fc6 M=160", 160, 4096, 25088}};
    hipStream_t s; CK(hipStreamCreate(&s));
    int shape_i = 0;
    for (auto &L : shapes) {
        if (shape_i++ >= nshapes_env) break;
        const size_t nx = (size_t)L.M * L.K, nw = (size_t)L.N * L.K;
        std::vector<uint16_t> hx(nx), hw(nw);
        uint32_t seed = 12345u;
        for (auto &v : hx) v = bf16_of(rnd(seed) * 2.0f);
        for (auto &v : hw) v = bf16_of(rnd(seed) * 0.05f);
        uint16_t *x, *w, *wt; float *b; uint16_t *y[3]; float *yf[2]; void *ws, *ws2;
        CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&b, L.N * 4));
        CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice)); CK(hipMemset(b, 0, L.N * 4));
        for (auto &p : y) CK(hipMalloc(&p, (size_t)L.M * L.N * 2));
        for (auto &p : yf) CK(hipMalloc(&p, (size_t)L.M * L.N * 4));
        const size_t wsb = frcnn_linear_bf16_workspace_bytes(L.M, L.N, L.K), wsb2 = frcnn_linear_bf16_tiled_workspace_bytes(L.M, L.N, L.K);
        const size_t tb = frcnn_linear_bf16_tiled_bytes(L.N, L.K);
        CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&ws2, wsb2)); CK(hipMalloc(&wt, tb));
        if (frcnn_linear_bf16_tile_w(w, L.N, L.K, wt, s) != 0) { printf("%s: tile_w refused\n", L.name); continue; }
        CK(hipStreamSynchronize(s));
        const double gf = 2.0 * L.M * L.N * L.K / 1e9, mb = (double)nw * 2 / 1e6;
        if (do_old) {
            const double us = time_graph(s, [&](int i) { if (frcnn_linear_bf16(x, w, b, y[i % 3], L.M, L.N, L.K, 1, 1, ws, wsb, s) != 0) { printf("refused\n"); exit(1); } }, 4);
            printf("[old ] %-14s %4d x %6d x %5d  %7.1f us  %6.0f TFLOP/s  weights %.0f MB at %.2f TB/s\n", L.name, L.M, L.K, L.N, us, gf / us * 1e3, mb, mb / us);
        }
        if (do_new) {
            const double us = time_graph(s, [&](int i) { if (frcnn_linear_bf16_tiled(x, wt, b, y[i % 3], L.M, L.N, L.K, 1, 1, ws2, wsb2, s) != 0) { printf("refused\n"); exit(1); } }, 4);
            printf("[ring] %-14s %4d x %6d x %5d  %7.1f us  %6.0f TFLOP/s  weights %.0f MB at %.2f TB/s\n", L.name, L.M, L.K, L.N, us, gf / us * 1e3, mb, mb / us);
        }
        if (do_new && getenv("LINEAR_MICRO_REALLOC")) {
            // which buffer's PLACEMENT decides the kernel's two speeds (round 6: fc6 runs at ~75 or ~84 us from process to process)?  Re-allocate one of
            // x / tiled weights / slabs behind a pad of a varying size and time again: LINEAR_MICRO_REALLOC = "x" | "w" | "s" | "a" (all three), eight rounds
            const char *what = getenv("LINEAR_MICRO_REALLOC");
            std::vector<void *> pads;
            for (int round = 0; round < 8; ++round) {
                void *pad; CK(hipMalloc(&pad, (size_t)(round + 1) * 3 * 1024 * 1024 + 4096 * round)); pads.push_back(pad);
                if (strchr(what, 'x') || strchr(what, 'a')) { uint16_t *x2; CK(hipMalloc(&x2, nx * 2)); CK(hipMemcpy(x2, x, nx * 2, hipMemcpyDeviceToDevice)); CK(hipFree(x)); x = x2; }
                if (strchr(what, 'w') || strchr(what, 'a')) { uint16_t *w2; CK(hipMalloc(&w2, tb)); CK(hipMemcpy(w2, wt, tb, hipMemcpyDeviceToDevice)); CK(hipFree(wt)); wt = w2; }
                if (strchr(what, 's') || strchr(what, 'a')) { void *s2; CK(hipMalloc(&s2, wsb2)); CK(hipFree(ws2)); ws2 = s2; }
                const double us = time_graph(s, [&](int i) { if (frcnn_linear_bf16_tiled(x, wt, b, y[i % 3], L.M, L.N, L.K, 1, 1, ws2, wsb2, s) != 0) { printf("refused\n"); exit(1); } }, 4);
                printf("[ring realloc %s #%d] %-8s %7.1f us   x %p wt %p slabs %p\n", what, round, L.name, us, (void *)x, (void *)wt, ws2);
            }
            for (void *q : pads) CK(hipFree(q));
        }
        if (do_old && do_new) {                                  // fp32 outputs of both (no ReLU), compared
            frcnn_linear_bf16(x, w, b, yf[0], L.M, L.N, L.K, 0, 0, ws, wsb, s);
            frcnn_linear_bf16_tiled(x, wt, b, yf[1], L.M, L.N, L.K, 0, 0, ws2, wsb2, s);
            CK(hipStreamSynchronize(s));
            std::vector<float> a((size_t)L.M * L.N), c((size_t)L.M * L.N);
            CK(hipMemcpy(a.data(), yf[0], a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), yf[1], c.size() * 4, hipMemcpyDeviceToHost));
            double md = 0, mx = 0;
            for (size_t i = 0; i < a.size(); ++i) { md = std::max(md, (double)fabsf(a[i] - c[i])); mx = std::max(mx, (double)fabsf(a[i])); }
            printf("       %-14s max |old - ring| = %.3g of max |y| = %.3g (%.2g relative)\n", L.name, md, mx, md / mx);
        }
        fflush(stdout);
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(wt)); CK(hipFree(b)); CK(hipFree(ws)); CK(hipFree(ws2)); for (auto &p : y) CK(hipFree(p)); for (auto &p : yf) CK(hipFree(p));
    }
    return 0;
}
