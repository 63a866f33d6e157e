// conv_bf16_micro.cpp -- times frcnn_conv_bf16_ws (libfrcnn_hip.so) on the VGG-16 layer shapes at 600 x 1000 without torch: per layer a
// captured graph of 10 back-to-back launches (outputs rotating over 3 buffers), bursts of graph launches between two events.
// Usage: conv_bf16_micro [layer ...] with settings from the environment (FRCNN_BF16_DMA, FRCNN_BF16_DMA_DEFAULT, FRCNN_BF16_SPLIT ...)
//        or conv_bf16_micro --modes "141 231 611" [layer ...] to sweep FRCNN_BF16_DMA per layer.
//        --check: every mode's output is also compared with the default pick's (bf16 words that differ, largest difference) -- a
//        hardware sanity check for a new staging form in seconds, without torch; the parity tests proper are tests/.
// Values: uniform bf16 in [-1, 1) (weights x 0.05).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <random>
#include <sstream>
#include <string>
#include <vector>
#include "frcnn_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Layer { const char *name; int ci, co, h, w, pool; };
static const Layer kLayers[] = {
    {"conv1_2", 64, 64, 600, 1000, 1}, {"conv2_1", 64, 128, 300, 500, 0}, {"conv2_2", 128, 128, 300, 500, 1}, {"conv3_1", 128, 256, 150, 250, 0},
    {"conv3_2", 256, 256, 150, 250, 0}, {"conv3_3", 256, 256, 150, 250, 1}, {"conv4_1", 256, 512, 75, 125, 0}, {"conv4_2", 512, 512, 75, 125, 0},
    {"conv4_3", 512, 512, 75, 125, 1}, {"conv5_1", 512, 512, 38, 63, 0}};

static uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char **argv) {
    std::vector<std::string> modes, want;
    bool check = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--check")) check = true;
        else if (!strcmp(argv[i], "--modes") && i + 1 < argc) { std::istringstream ss(argv[++i]); std::string m; while (ss >> m) modes.push_back(m); }
        else want.push_back(argv[i]);
    }
    if (modes.empty()) modes.push_back("");
    hipStream_t s; CK(hipStreamCreate(&s));
    const int burst = getenv("CONV_MICRO_BURST") ? atoi(getenv("CONV_MICRO_BURST")) : 20;
    std::mt19937 g(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    double total_us_best = 0;
    for (const Layer &L : kLayers) {
        if (!want.empty() && std::find(want.begin(), want.end(), std::string(L.name)) == want.end()) continue;
        const int cip = frcnn_bf16_padded_channels(L.ci), cop = frcnn_bf16_padded_channels(L.co);
        const size_t nx = (size_t)cip * L.h * L.w, nw = (size_t)9 * cop * cip, ny = (size_t)cop * L.h * L.w;
        std::vector<uint16_t> hx(nx), hw(nw);
        for (auto &e : hx) e = bf16_of(u(g));
        for (auto &e : hw) e = bf16_of(0.05f * u(g));
        if (getenv("CONV_MICRO_ZERO")) {                                   // all-zero operands: the same instruction stream with no bits toggling in the
            std::fill(hx.begin(), hx.end(), (uint16_t)0);                  // matrix datapath -- separates what the data costs (power -> clocks) from what
            std::fill(hw.begin(), hw.end(), (uint16_t)0);                  // the instruction stream costs
        }
        uint16_t *dx, *dw, *dy[3]; float *db; void *ws;
        // CONV_MICRO_COLD = w | x | wx: every launch of the timed graph reads its weights / its input from a DIFFERENT copy, enough copies to exceed the 256 MB
        // Infinity Cache -- what a layer sees inside the forward (weights last touched one image ago, behind 270 MB of other weights) instead of the harness's
        // ten launches on one resident set
        const char *cold = getenv("CONV_MICRO_COLD");
        const bool cold_w = cold && strchr(cold, 'w'), cold_x = cold && strchr(cold, 'x');
        const int n_launch = (cold_w || cold_x) ? 40 : 10;
        const size_t wcopies = cold_w ? (size_t)(300e6 / (nw * 2)) + 2 : 1, xcopies = cold_x ? (size_t)(300e6 / (nx * 2)) + 2 : 1;
        std::vector<uint16_t *> dws(wcopies), dxs(xcopies);
        for (auto &p : dws) CK(hipMalloc(&p, nw * 2));
        for (auto &p : dxs) CK(hipMalloc(&p, nx * 2));
        dx = dxs[0]; dw = dws[0];
        CK(hipMalloc(&db, cop * 4)); CK(hipMemset(db, 0, cop * 4));
        for (auto &p : dy) CK(hipMalloc(&p, ny * 2));
        const size_t wsb = frcnn_conv_bf16_workspace_bytes(L.ci, L.co, L.h, L.w);
        CK(hipMalloc(&ws, wsb));
        if (frcnn_conv_bf16_workspace_init(ws, wsb, s) != 0) { printf("workspace init failed\n"); return 1; }
        for (auto &p : dxs) CK(hipMemcpy(p, hx.data(), nx * 2, hipMemcpyHostToDevice));
        for (auto &p : dws) CK(hipMemcpy(p, hw.data(), nw * 2, hipMemcpyHostToDevice));
        const double gflop = 2.0 * L.h * L.w * L.co * L.ci * 9 / 1e9;
        printf("%-8s %3d->%3d %4dx%-4d %6.1f GFLOP:", L.name, L.ci, L.co, L.h, L.w, gflop);
        const size_t nout = L.pool ? (size_t)cop * ((L.h + 1) / 2) * ((L.w + 1) / 2) : ny;
        std::vector<uint16_t> ref, got;
        if (check) {                                                       // the default pick's output
            frcnn_set_tuning("FRCNN_BF16_DMA", nullptr);
            CK(hipMemsetAsync(dy[0], 0xff, ny * 2, s));
            if (frcnn_conv_bf16_ws(dx, dw, db, dy[0], L.ci, L.co, L.h, L.w, 3, 1, L.pool ? 2 : 0, ws, wsb, s) != 0) { printf(" default launch refused\n"); return 1; }
            ref.resize(nout); got.resize(nout);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(ref.data(), dy[0], nout * 2, hipMemcpyDeviceToHost));
        }
        double best = 1e30;
        for (const std::string &m : modes) {
            frcnn_set_tuning("FRCNN_BF16_STRIP", nullptr);
            if (m == "def") frcnn_set_tuning("FRCNN_BF16_DMA", nullptr);                 // "def" in a --modes list = the default pick
            else if (m == "old") { frcnn_set_tuning("FRCNN_BF16_DMA", nullptr); frcnn_set_tuning("FRCNN_BF16_STRIP", "0"); }   // "old" = conv_dma_bf16_kernel's picks (no strip rule)
            else if (!m.empty()) frcnn_set_tuning("FRCNN_BF16_DMA", m.c_str());
            hipGraph_t gr; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            bool ok = true;
            static size_t rot = 0;
            for (int i = 0; i < n_launch; ++i, ++rot)
                ok = ok && frcnn_conv_bf16_ws(dxs[rot % xcopies], dws[rot % wcopies], db, dy[i % 3], L.ci, L.co, L.h, L.w, 3, 1, L.pool ? 2 : 0, ws, wsb, s) == 0;
            CK(hipStreamEndCapture(s, &gr));
            if (!ok) { printf("  %s: launch refused", m.c_str()); CK(hipGraphDestroy(gr)); continue; }
            if (check) {
                CK(hipMemsetAsync(dy[0], 0xff, ny * 2, s));
                if (frcnn_conv_bf16_ws(dx, dw, db, dy[0], L.ci, L.co, L.h, L.w, 3, 1, L.pool ? 2 : 0, ws, wsb, s) != 0) { printf(" launch refused\n"); return 1; }
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(got.data(), dy[0], nout * 2, hipMemcpyDeviceToHost));
                size_t ndiff = 0; double maxd = 0, maxv = 0;
                for (size_t i = 0; i < nout; ++i) {
                    uint32_t a = (uint32_t)got[i] << 16, b = (uint32_t)ref[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
                    if (got[i] != ref[i]) { ++ndiff; if (!(fabs((double)fa - fb) <= maxd)) maxd = fabs((double)fa - fb); }
                    if (fabs((double)fb) > maxv) maxv = fabs((double)fb);
                }
                if (ndiff == 0) printf("  [%s = default]", m.empty() ? "default" : m.c_str());
                else printf("  [%s: %zu of %zu words differ, max |d| %.3g of %.3g]", m.c_str(), ndiff, nout, maxd, maxv);
            }
            CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            std::vector<float> us;
            for (int r = 0; r < 7; ++r) {
                CK(hipEventRecord(e0, s));
                for (int b = 0; b < burst; ++b) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); us.push_back(ms * 1000.f / n_launch / burst);
            }
            std::sort(us.begin(), us.end());
            const double med = us[us.size() / 2];
            best = std::min(best, med);
            printf("  %s %6.1f us %5.0f TF", m.empty() ? "default" : m.c_str(), med, gflop / med * 1e3);      // GFLOP / us = PFLOP/s
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        }
        printf("\n");
        total_us_best += best * (strcmp(L.name, "conv5_1") == 0 ? 4 : 1);      // conv5_1's shape runs four times in the chain (conv5_1..3, rpn_conv_3x3)
        frcnn_set_tuning("FRCNN_BF16_DMA", nullptr);
        for (auto &p : dxs) CK(hipFree(p));
        for (auto &p : dws) CK(hipFree(p));
        CK(hipFree(db)); CK(hipFree(ws)); for (auto &p : dy) CK(hipFree(p));
    }
    printf("sum of the best per layer (conv5_1 x 4, conv1_1 not included): %.1f us\n", total_us_best);
    return 0;
}
