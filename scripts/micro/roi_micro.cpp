// roi_micro.cpp -- times frcnn_roi_pool_fwd_chw (libfrcnn_hip.so) on the benchmark's own RoIs without torch: a captured graph of
// 10 back-to-back launches into 10 rotating 30 MB outputs, per environment setting given on the command line
// (e.g.  roi_micro FRCNN_ROI_ST=0 FRCNN_ROI_ST=1 "FRCNN_ROI_KERNEL=planes": applied through frcnn_set_tuning).  Each result is compared bit for bit with the
// channel-last gather kernel (frcnn_roi_pool_fwd_hwc after frcnn_chw_to_hwc: an independent implementation in the same library).
// Inputs: scripts/_data/bench_rois.npy (300 x 4 f32, written once from the oracle's proposals for synthetic.image(seed 0)) and
// optionally bench_feat.npy (512 x 38 x 63); without it the map is |N(0,1)|.
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

static std::vector<float> load_npy_f32(const char *path, size_t expect) {
    std::vector<float> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    unsigned char hdr[10];
    if (fread(hdr, 1, 10, f) != 10) { fclose(f); return v; }
    const size_t hlen = hdr[8] | (hdr[9] << 8);
    fseek(f, (long)(10 + hlen), SEEK_SET);
    v.resize(expect);
    if (fread(v.data(), 4, expect, f) != expect) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    const int C = 512, H = 38, W = 63, R = 300;
    std::string dir = "scripts/_data/";
    std::vector<float> rois = load_npy_f32((dir + "bench_rois.npy").c_str(), (size_t)R * 4);
    if (rois.empty()) { printf("missing %sbench_rois.npy\n", dir.c_str()); return 1; }
    std::vector<float> feat = load_npy_f32((dir + "bench_feat.npy").c_str(), (size_t)C * H * W);
    if (feat.empty()) {
        feat.resize((size_t)C * H * W);
        std::mt19937 g(1); std::normal_distribution<float> n(0.f, 1.f);
        for (auto &e : feat) e = fabsf(n(g));
        printf("(random map)\n");
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    float *dx, *drois, *dxt, *yref, *ys[10];
    const size_t ybytes = (size_t)R * C * 49 * 4;
    CK(hipMalloc(&dx, feat.size() * 4)); CK(hipMalloc(&dxt, feat.size() * 4)); CK(hipMalloc(&drois, rois.size() * 4)); CK(hipMalloc(&yref, ybytes));
    for (auto &p : ys) CK(hipMalloc(&p, ybytes));
    CK(hipMemcpy(dx, feat.data(), feat.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(drois, rois.data(), rois.size() * 4, hipMemcpyHostToDevice));
    if (frcnn_chw_to_hwc(dx, C, H, W, dxt, s) != 0 || frcnn_roi_pool_fwd_hwc(dxt, C, H, W, drois, R, 4, 7, 7, 1.f / 16, yref, nullptr, s) != 0) { printf("reference launch failed\n"); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<float> href((size_t)R * C * 49), hy((size_t)R * C * 49);
    CK(hipMemcpy(href.data(), yref, ybytes, hipMemcpyDeviceToHost));
    // clock ramp
    for (int i = 0; i < 3000; ++i) frcnn_roi_pool_fwd_chw(dx, C, H, W, drois, R, 4, 7, 7, 1.f / 16, ys[i % 10], nullptr, nullptr, 0, s);
    CK(hipStreamSynchronize(s));
    // the training forms: forward with arg-max (65.1 MB algorithmic) and backward (65.1 MB), same graph-of-10 timing
    {
        int32_t *am[10]; float *dxs[10];
        for (auto &q : am) CK(hipMalloc(&q, ybytes));
        for (auto &q : dxs) CK(hipMalloc(&q, feat.size() * 4));
        int32_t *amref; CK(hipMalloc(&amref, ybytes));
        if (frcnn_roi_pool_fwd_hwc(dxt, C, H, W, drois, R, 4, 7, 7, 1.f / 16, yref, amref, s) != 0) { printf("reference launch failed\n"); return 1; }
        std::vector<int32_t> ham((size_t)R * C * 49), hamref((size_t)R * C * 49);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hamref.data(), amref, ybytes, hipMemcpyDeviceToHost));
        const int burst = getenv("ROI_MICRO_BURST") ? atoi(getenv("ROI_MICRO_BURST")) : 1;
        for (const char *which : {"fwd+argmax", "fwd+argmax FRCNN_ROI_KERNEL=planes", "bwd", "bwd form=n1", "bwd form=n4", "bwd FRCNN_ROI_BWD=atomic"}) {
            if (const char *f = strstr(which, "form=")) { char form[32]; sscanf(f + 5, "%31s", form); frcnn_set_tuning("FRCNN_ROI_BWD", form); }
            if (strstr(which, "dbg8")) frcnn_set_tuning("FRCNN_ROI_BWD_DBG", "8");
            if (strstr(which, "dbg1")) frcnn_set_tuning("FRCNN_ROI_BWD_DBG", "1");
            if (strstr(which, "dbg2")) frcnn_set_tuning("FRCNN_ROI_BWD_DBG", "2");
            if (strstr(which, "dbg4")) frcnn_set_tuning("FRCNN_ROI_BWD_DBG", "4");
            const bool is_bwd = which[0] == 'b';
            if (strstr(which, "planes")) frcnn_set_tuning("FRCNN_ROI_KERNEL", "planes");
            if (strstr(which, "atomic")) frcnn_set_tuning("FRCNN_ROI_BWD", "atomic");
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 10; ++i) {
                const int st = is_bwd ? frcnn_roi_pool_bwd(ys[i], am[i], R, C, H, W, 7, 7, dxs[i], s)
                                      : frcnn_roi_pool_fwd_chw(dx, C, H, W, drois, R, 4, 7, 7, 1.f / 16, ys[i], am[i], nullptr, 0, s);
                if (st != 0) { printf("launch failed\n"); return 1; }
            }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            std::vector<float> us;
            for (int r = 0; r < 40; ++r) {
                CK(hipEventRecord(e0, s));
                for (int b = 0; b < burst; ++b) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); us.push_back(ms * 100.f / burst);
            }
            std::sort(us.begin(), us.end());
            const char *verdict = "";
            if (!is_bwd) {
                CK(hipMemcpy(hy.data(), ys[3], ybytes, hipMemcpyDeviceToHost));
                CK(hipMemcpy(ham.data(), am[3], ybytes, hipMemcpyDeviceToHost));
                verdict = (memcmp(hy.data(), href.data(), ybytes) == 0 && memcmp(ham.data(), hamref.data(), ybytes) == 0) ? "bit-exact (values and indices)" : "MISMATCH";
            }
            printf("%-40s best %6.2f us  median %6.2f us  (%.3f of 8 TB/s)  %s\n", which, us[0], us[us.size() / 2], 65.1e6 / (us[us.size() / 2] * 1e-6) / 8e12, verdict);
            frcnn_set_tuning("FRCNN_ROI_KERNEL", nullptr); frcnn_set_tuning("FRCNN_ROI_BWD", nullptr); frcnn_set_tuning("FRCNN_ROI_BWD_DBG", nullptr);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    std::vector<std::string> settings;
    for (int i = 1; i < argc; ++i) settings.push_back(argv[i]);
    if (settings.empty()) settings.push_back("DEFAULT=1");
    for (int rep = 0; rep < 2; ++rep)
    for (const auto &st : settings) {
        // a setting is "A=1,B=2": applied for the capture (the library reads its hooks at launch time), removed afterwards
        std::vector<std::string> names;
        size_t pos = 0;
        while (pos < st.size()) {
            size_t e = st.find(',', pos); if (e == std::string::npos) e = st.size();
            const std::string kv = st.substr(pos, e - pos); const size_t eq = kv.find('=');
            if (eq != std::string::npos && kv.compare(0, 6, "FRCNN_") == 0) { frcnn_set_tuning(kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str()); names.push_back(kv.substr(0, eq)); }
            pos = e + 1;
        }
        CK(hipMemsetAsync(ys[3], 0xff, ybytes, s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 10; ++i)
            if (frcnn_roi_pool_fwd_chw(dx, C, H, W, drois, R, 4, 7, 7, 1.f / 16, ys[i], nullptr, nullptr, 0, s) != 0) { printf("launch failed\n"); return 1; }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<float> us;
        // ROI_MICRO_BURST graph launches per sample without a host sync in between (default 1): a busy GPU holds its clock
        const int burst = getenv("ROI_MICRO_BURST") ? atoi(getenv("ROI_MICRO_BURST")) : 1;
        for (int r = 0; r < 40; ++r) {
            CK(hipEventRecord(e0, s));
            for (int b = 0; b < burst; ++b) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); us.push_back(ms * 100.f / burst);
        }
        std::sort(us.begin(), us.end());
        CK(hipMemcpy(hy.data(), ys[3], ybytes, hipMemcpyDeviceToHost));
        const bool same = memcmp(hy.data(), href.data(), ybytes) == 0;
        printf("%-40s best %6.2f us  median %6.2f us  (%.3f of 8 TB/s)  %s\n", st.c_str(), us[0], us[us.size() / 2], 35.01e6 / (us[us.size() / 2] * 1e-6) / 8e12,
               same ? "bit-exact" : "MISMATCH");
        if (st.find("STAMPS=1") != std::string::npos) {
            // per-wave s_memtime stamps (ablation build, FRCNN_ROI_DBG & 128): [wg][wave][32]; print a few waves as deltas from the first
            const uint32_t *u = reinterpret_cast<const uint32_t *>(hy.data());
            for (int wg : {0, 1, 100, 255}) for (int wv : {0, 5, 13, 15}) {
                const uint32_t *q = u + ((size_t)wg * 16 + wv) * 32;
                printf("  wg %3d wave %2d:", wg, wv);
                for (int k = 1; k < 16; ++k) printf(" %6u", q[k] - q[0]);
                printf("\n");
            }
        }
        for (const auto &n : names) frcnn_set_tuning(n.c_str(), nullptr);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
