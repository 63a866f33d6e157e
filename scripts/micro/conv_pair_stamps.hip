// conv_pair_stamps.hip -- the conv1 pair kernel compiled with s_memtime stamps at its phase boundaries (first workgroup, first 16 tiles, every wave):
// where a tile's time goes.  Tuning tool; the stamps themselves cost a few per cent.
#define FRCNN_PAIR_STAMPS 1
#include "../../chainer-faster-rcnn_amd/csrc/conv_bf16_pair.hip"
#include <stdio.h>
#include <vector>
#include <string.h>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main() {
    const int H = 600, W = 1000, Cin = 3, OH = 300, OW = 500;
    std::mt19937 g(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> hx((size_t)Cin * H * W), hw1(64 * 27), hb(64, 0.1f);
    std::vector<uint16_t> hw2(4 * 9 * 64 * 16);
    for (auto &e : hx) e = 120.f * u(g);
    for (auto &e : hw1) e = 0.01f * u(g);
    for (auto &e : hw2) { float f = 0.05f * u(g); uint32_t b; memcpy(&b, &f, 4); e = (uint16_t)(b >> 16); }
    float *dx, *dw1, *db; uint16_t *dw2, *dy;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw1, hw1.size() * 4)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dw2, hw2.size() * 2)); CK(hipMalloc(&dy, (size_t)64 * OH * OW * 2));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dw2, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) if (frcnn_conv1_pair_bf16(dx, dw1, db, dw2, db, dy, Cin, H, W, nullptr) != 0) { printf("refused\n"); return 1; }
    CK(hipDeviceSynchronize());
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        double best = 1e30;
        for (int t = 0; t < 5; ++t) {
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < 50; ++i) frcnn_conv1_pair_bf16(dx, dw1, db, dw2, db, dy, Cin, H, W, nullptr);
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms * 20.0 < best) best = ms * 20.0;
        }
        printf("kernel (stream of 50 launches, stamps compiled in): %.1f us per launch\n", best);
        if (getenv("PAIR_TIME_ONLY")) return 0;
    }
    std::vector<unsigned long long> st(8 * 16 * 8);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(frcnn_pair_stamps), st.size() * 8));
    const bool form1 = getenv("FRCNN_BF16_PAIR_FORM") && atoi(getenv("FRCNN_BF16_PAIR_FORM")) == 1;
    const char *names1[] = {"store_patch+issue", "wait B1", "conv1_1 units", "wait B2", "main loop", "epilogue"};
    const char *namesC[] = {"main loop", "wait barrier", "epilogue"}, *namesP[] = {"conv1_1 units", "wait barrier"};
    for (int w = 0; w < (form1 ? 4 : 8); ++w) {
        printf("wave %d%s:\n", w, form1 ? "" : (w < 4 ? " (consumer)" : " (producer)"));
        const int nph = form1 ? 6 : (w < 4 ? 3 : 2);
        const char **names = form1 ? names1 : (w < 4 ? namesC : namesP);
        for (int it = 0; it < 12; ++it) {
            const unsigned long long *p = &st[(w * 16 + it) * 8];
            if (!p[0]) break;
            printf("  it %d:", it);
            for (int i = 0; i < nph; ++i) printf(" %s %llu |", names[i], p[i + 1] - p[i]);
            if (it + 1 < 16 && st[(w * 16 + it + 1) * 8]) printf(" total %llu", st[(w * 16 + it + 1) * 8] - p[0]);
            if (!form1 && w >= 4 && p[6]) printf("   first unit: patch store (incl. wait for its loads) %llu | issue next loads %llu | im2col + MFMA + map write %llu", p[4] - p[3], p[5] - p[4], p[6] - p[5]);
            printf("\n");
        }
    }
    return 0;
}
