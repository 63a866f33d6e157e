// conv_pair_micro.cpp -- frcnn_conv1_pair_bf16 (conv1_1 + conv1_2 + pool1 as one launch, csrc/conv_bf16_pair.hip) against the two launches it
// replaces (frcnn_conv1_bf16, frcnn_conv_bf16_ws with out_mode 2) at 600 x 1000, without torch: both forms as captured graphs of 10 repetitions
// between two events, and the outputs compared word by word.  FRCNN_BF16_PAIR_RW (4 | 6) selects the tile height.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <vector>
#include "frcnn_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class F>
static double time_graph(hipStream_t s, F body, int reps, int burst) {
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) body(i);
    CK(hipStreamEndCapture(s, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int t = 0; t < 5; ++t) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < burst; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e3 / (burst * reps));
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
    return best;
}

int main(int argc, char **argv) {
    const int H = argc > 2 ? atoi(argv[1]) : 600, W = argc > 2 ? atoi(argv[2]) : 1000, Cin = 3;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    hipStream_t s; CK(hipStreamCreate(&s));
    std::mt19937 g(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> hx((size_t)Cin * H * W), hw1(64 * Cin * 9), hb1(64), hw2(64 * 64 * 9), hb2(64);
    for (auto &e : hx) e = 120.f * u(g);
    for (auto &e : hw1) e = 0.01f * u(g);
    for (auto &e : hb1) e = 0.1f * u(g);
    for (auto &e : hw2) e = 0.05f * u(g);
    for (auto &e : hb2) e = 0.1f * u(g);
    float *dx, *dw1, *db1, *dw2, *db2; uint16_t *dw2p, *dmid, *dy[3], *dy2; void *ws;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw1, hw1.size() * 4)); CK(hipMalloc(&db1, 256)); CK(hipMalloc(&dw2, hw2.size() * 4)); CK(hipMalloc(&db2, 256));
    CK(hipMalloc(&dw2p, 4 * 9 * 64 * 16 * 2)); CK(hipMalloc(&dmid, (size_t)64 * H * W * 2));
    for (auto &p : dy) CK(hipMalloc(&p, (size_t)64 * OH * OW * 2));
    CK(hipMalloc(&dy2, (size_t)64 * OH * OW * 2));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db1, hb1.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dw2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, hb2.data(), 256, hipMemcpyHostToDevice));
    if (frcnn_bf16_pack_conv_w(dw2, 64, 64, 3, dw2p, s) != 0) { printf("pack failed\n"); return 1; }
    const size_t wsb = frcnn_conv_bf16_workspace_bytes(64, 64, H, W);
    CK(hipMalloc(&ws, wsb));
    if (frcnn_conv_bf16_workspace_init(ws, wsb, s) != 0) { printf("workspace init failed\n"); return 1; }
    CK(hipStreamSynchronize(s));
    // word-by-word comparison
    CK(hipMemsetAsync(dy[0], 0xff, (size_t)64 * OH * OW * 2, s)); CK(hipMemsetAsync(dy2, 0xee, (size_t)64 * OH * OW * 2, s));
    if (frcnn_conv1_pair_bf16(dx, dw1, db1, dw2p, db2, dy[0], Cin, H, W, s) != 0) { printf("pair launch refused\n"); return 1; }
    if (frcnn_conv1_bf16(dx, dw1, db1, dmid, Cin, 64, H, W, 1, s) != 0 || frcnn_conv_bf16_ws(dmid, dw2p, db2, dy2, 64, 64, H, W, 3, 1, 2, ws, wsb, s) != 0) { printf("two-launch chain refused\n"); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<uint16_t> a((size_t)64 * OH * OW), b(a.size());
    CK(hipMemcpy(a.data(), dy[0], a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dy2, b.size() * 2, hipMemcpyDeviceToHost));
    size_t nd = 0, nz = 0;
    for (size_t i = 0; i < a.size(); ++i) { nd += a[i] != b[i]; nz += a[i] != 0; }
    printf("conv1 pair %dx%d: %zu of %zu bf16 words differ from the two-launch chain (%zu non-zero)\n", H, W, nd, a.size(), nz);
    const double t_pair = time_graph(s, [&](int i) { frcnn_conv1_pair_bf16(dx, dw1, db1, dw2p, db2, dy[i % 3], Cin, H, W, s); }, 10, 20);
    const double t_c11 = time_graph(s, [&](int i) { (void)i; frcnn_conv1_bf16(dx, dw1, db1, dmid, Cin, 64, H, W, 1, s); }, 10, 20);
    const double t_c12 = time_graph(s, [&](int i) { frcnn_conv_bf16_ws(dmid, dw2p, db2, dy[i % 3], 64, 64, H, W, 3, 1, 2, ws, wsb, s); }, 10, 20);
    const double gflop = 2.0 * H * W * 64 * (Cin * 9 + 64 * 9) / 1e9;
    printf("pair %.1f us (%.0f TFLOP/s algorithmic)   two launches: conv1_1 %.1f + conv1_2 %.1f = %.1f us\n", t_pair, gflop / t_pair * 1e3,
           t_c11, t_c12, t_c11 + t_c12);
    return nd == 0 ? 0 : 2;
}
