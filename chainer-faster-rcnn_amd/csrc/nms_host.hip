// nms_host.hip -- `_nms`, the reference's one C FFI on this path, with its exact signature
// (/root/reference/models/gpu_nms.hpp:9-10; bound by models/gpu_nms.pyx:16-31), over the device NMS of detect.hip.
//
// Host pointers in and out, synchronous, device memory allocated and freed per call -- the conventions of the entry point it
// replaces (nms_kernel.cu:100-109,142-143), NOT those of the frcnn_* entry points (device pointers, caller-owned workspace, a
// stream).  Differences from the original, all deliberate:
//   * errors are REPORTED: *num_out = -1 (the original printed and carried on, nms_kernel.cu:12-19);
//   * `device_id` is honoured for the duration of the call and the caller's current device is restored afterwards (the
//     original left hipSetDevice's side effect behind, nms_kernel.cu:80-89);
//   * the suppression rule is cpu_nms.pyx's -- `(double)iou >= thresh` -- because the reference's CPU path is the oracle
//     (nms_kernel.cu:71 tests `>` in fp32 and is dead code).  The Cython wrapper narrows the Python threshold to a C float on
//     the way in; `_nms` recovers the double the caller wrote as the SHORTEST decimal that rounds to that float (0.7f -> 0.7,
//     0.3f -> 0.3), so gpu_nms(dets, t) returns exactly cpu_nms(dets, t) for every threshold a person would type.
// boxes_host: boxes_num rows of boxes_dim >= 5 floats [x1,y1,x2,y2,score,...], sorted by descending score by the wrapper
// (gpu_nms.pyx:25-28); keep_out (capacity boxes_num) receives indices into that array.
#include "frcnn_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {

double shortest_double_of_float(float f) {
    if (!(f == f) || f == 0.0f) return (double)f;
    char buf[64];
    for (int prec = 1; prec <= 9; ++prec) {
        snprintf(buf, sizeof(buf), "%.*g", prec, (double)f);
        if (strtof(buf, nullptr) == f) return strtod(buf, nullptr);
    }
    return (double)f;
}

struct DeviceScope {      // hipSetDevice for the call only
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); return; }
        ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
        if (!ok) {
            prev = -1;
            (void)hipGetLastError();          // a refused device id must not linger as the "last error" of the caller's next launch
        }
    }
    ~DeviceScope() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

struct DeviceBuf {
    void *p = nullptr;
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess; }
    ~DeviceBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh,
                     int device_id) {
    if (!num_out) return;
    *num_out = -1;
    if (boxes_num < 0 || boxes_dim < 5 || (boxes_num > 0 && (!keep_out || !boxes_host))) return;
    if (boxes_num == 0) { *num_out = 0; return; }
    (void)hipGetLastError();                  // start from a clean slate: frcnn_nms reports hipGetLastError() after its launches
    DeviceScope scope(device_id);
    if (!scope.ok) return;
    const size_t n = (size_t)boxes_num;
    std::vector<float> packed;
    const float *src = boxes_host;
    if (boxes_dim != 5) {                                   // the kernel reads rows of five floats
        packed.resize(n * 5);
        for (size_t i = 0; i < n; ++i)
            for (int c = 0; c < 5; ++c) packed[i * 5 + c] = boxes_host[i * (size_t)boxes_dim + c];
        src = packed.data();
    }
    const size_t wsb = frcnn_nms_workspace_bytes(boxes_num);
    DeviceBuf dets, keep, nkeep, ws;
    if (!dets.alloc(n * 5 * sizeof(float)) || !keep.alloc(n * sizeof(int32_t)) || !nkeep.alloc(sizeof(int32_t)) || !ws.alloc(wsb)) return;
    if (hipMemcpy(dets.p, src, n * 5 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return;
    const double thresh = shortest_double_of_float(nms_overlap_thresh);
    if (frcnn_nms((const float *)dets.p, boxes_num, thresh, 0, (int32_t *)keep.p, (int32_t *)nkeep.p, ws.p, wsb, nullptr) != FRCNN_OK) return;
    int32_t k = -1;
    if (hipMemcpy(&k, nkeep.p, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return;     // synchronises with the null stream
    if (k < 0 || k > boxes_num) return;
    static_assert(sizeof(int) == sizeof(int32_t), "gpu_nms.pyx:12 asserts the same");
    if (k > 0 && hipMemcpy(keep_out, keep.p, (size_t)k * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return;
    *num_out = k;
}
