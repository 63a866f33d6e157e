// store_micro.hip -- round-3 micro-measurements behind the RoI-pooling redesign.  Standalone: hipcc only, no torch.
//   * what a launch costs in a chain of graph nodes (empty kernel, the RoI kernel's grid)
//   * what 30.1 MB of output costs with plain / nt / sc1 / sc0+sc1 stores (is the dirty-L2 write-back at the end of a kernel
//     what made "the parts add up" in round 2's ablation?)
//   * what the map prologue costs: 8- or 4-channel cell images into LDS, then exit
// Build + run: scripts/micro/run_micro.sh (GPU only).  Prints one line per variant: us per launch from hipEvents around a
// captured graph of kLaunches back-to-back launches, best and median of kReps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __amdgpu_buffer_rsrc_t buf_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(1024) empty_kernel(float *y) { if (y == nullptr && threadIdx.x == 12345) y[0] = 1.f; }

template <int AUX>
__global__ void __launch_bounds__(1024) fill_kernel(float *y, uint32_t n16) {   // n16 = number of 16-byte pieces
    const buf_t b = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)(n16 * 16u), 0x00020000);
    u32x4 v; v.x = threadIdx.x; v.y = blockIdx.x; v.z = 3; v.w = 4;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x)
        __builtin_amdgcn_raw_buffer_store_b128(v, b, (int)(i * 16u), 0, AUX);
}

// the direct-store pattern of the lane = (ph, pw) design: a wave's pass = (RoI r, channel quad q) writes four 196-byte runs
// (one per channel: 49 consecutive floats) with buffer_store_dword, lanes 8 ph + pw with ph, pw < 7 active; 38400 passes = 30.1 MB.
// XCDMAP: quad index derived so that the 16 quads (64 channels) of a 12.5 KB stretch belong to ONE XCD (workgroup id % 8)
template <int AUX, bool XCDMAP>
__global__ void __launch_bounds__(1024) pattern_kernel(float *y, int R, int Q) {
    const buf_t b = __builtin_amdgcn_make_buffer_rsrc(y, 0, R * Q * 784, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ph = lane >> 3, pw = lane & 7;
    const uint32_t voff = (ph < 7 && pw < 7) ? (uint32_t)((ph * 7 + pw) * 4) : 0x80000000u;
    const int i = blockIdx.x;                           // 0 .. Q-1
    const int q = XCDMAP ? (i & 7) * (Q / 8) + (i >> 3) : i;
    const uint32_t v = lane + wave;
    for (int r = blockIdx.y * 16 + wave; r < R; r += gridDim.y * 16) {
        const uint32_t soff = (uint32_t)((r * Q + q) * 784);
        __builtin_amdgcn_raw_buffer_store_b32(v, b, (int)voff, (int)soff, AUX);
        __builtin_amdgcn_raw_buffer_store_b32(v, b, (int)(voff + 196), (int)soff, AUX);
        __builtin_amdgcn_raw_buffer_store_b32(v, b, (int)(voff + 392), (int)soff, AUX);
        __builtin_amdgcn_raw_buffer_store_b32(v, b, (int)(voff + 588), (int)soff, AUX);
    }
}
// same bytes as 16-byte-per-lane stores of the 784-byte run (49 lanes): what a staged copy-out would issue
template <int AUX>
__global__ void __launch_bounds__(1024) pattern16_kernel(float *y, int R, int Q) {
    const buf_t b = __builtin_amdgcn_make_buffer_rsrc(y, 0, R * Q * 784, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t voff = lane < 49 ? (uint32_t)(lane * 16) : 0x80000000u;
    u32x4 v; v.x = lane; v.y = wave; v.z = 3; v.w = 4;
    const int q = blockIdx.x;
    for (int r = blockIdx.y * 16 + wave; r < R; r += gridDim.y * 16)
        __builtin_amdgcn_raw_buffer_store_b128(v, b, (int)voff, (int)((r * Q + q) * 784), AUX);
}

// runs of RUN16 16-byte pieces (784 B x channel quads per pass): one wave writes a run with ceil(RUN16/64) instructions
template <int AUX, int RUN16, bool XCDMAP>
__global__ void __launch_bounds__(1024) run_kernel(float *y, int R) {
    constexpr int Q = 6272 / RUN16;                    // runs per RoI (100352 B)
    const buf_t b = __builtin_amdgcn_make_buffer_rsrc(y, 0, R * 100352, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 v; v.x = lane; v.y = wave; v.z = 3; v.w = 4;
    const int nwg_r = gridDim.x / Q;                   // workgroups sharing one run index
    const int i = blockIdx.x % Q, g = blockIdx.x / Q;
    const int q = XCDMAP ? (i & 7) * (Q / 8) + (i >> 3) : i;
    for (int r = g * 16 + wave; r < R; r += nwg_r * 16) {
        const uint32_t soff = (uint32_t)(r * 100352 + q * RUN16 * 16);
#pragma unroll
        for (int k = 0; k < (RUN16 + 63) / 64; ++k) {
            const int e = k * 64 + lane;
            __builtin_amdgcn_raw_buffer_store_b128(v, b, (int)(e < RUN16 ? (uint32_t)(e * 16) : 0x80000000u), (int)soff, AUX);
        }
    }
}

// hybrid policy: the 16-byte pieces of a run that lie in 64-byte blocks the run covers completely go out write-through (sc1); the
// one or two ragged 64-byte blocks at its ends (shared with the neighbouring run) are plain stores that the L2 merges and writes
// back at the end of the kernel (a few % of the bytes)
template <int RUN16, bool XCDMAP>
__global__ void __launch_bounds__(1024) hybrid_kernel(float *y, int R) {
    constexpr int Q = 6272 / RUN16;
    const buf_t b = __builtin_amdgcn_make_buffer_rsrc(y, 0, R * 100352, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 v; v.x = lane; v.y = wave; v.z = 3; v.w = 4;
    const int nwg_r = gridDim.x / Q;
    const int i = blockIdx.x % Q, g = blockIdx.x / Q;
    const int q = XCDMAP ? (i & 7) * (Q / 8) + (i >> 3) : i;
    const int s16 = (q * RUN16) & 3;                                   // run start in 16-byte units mod 4 (100352 % 64 == 0)
    const int head = (4 - s16) & 3;                                    // pieces before the first aligned block
    const int tail = (s16 + RUN16) & 3;                                // pieces after the last aligned block
    for (int r = g * 16 + wave; r < R; r += nwg_r * 16) {
        const uint32_t soff = (uint32_t)(r * 100352 + q * RUN16 * 16);
#pragma unroll
        for (int k = 0; k < (RUN16 + 63) / 64; ++k) {
            const int e = k * 64 + lane;
            const bool inner = e >= head && e < RUN16 - tail;
            __builtin_amdgcn_raw_buffer_store_b128(v, b, (int)(inner ? (uint32_t)(e * 16) : 0x80000000u), (int)soff, 16);
        }
        // ragged pieces: lanes 0 .. head-1 and RUN16-tail .. RUN16-1 (at most 6)
        const int e2 = lane < head ? lane : RUN16 - tail + (lane - head);
        __builtin_amdgcn_raw_buffer_store_b128(v, b, (int)(lane < head + tail ? (uint32_t)(e2 * 16) : 0x80000000u), (int)soff, 0);
    }
}

// the map prologue alone: CH channels of a 38 x 63 map as CH*4-byte cells in LDS (row pitch 64 cells), then one dependent store
template <int CH, int THREADS>
__global__ void __launch_bounds__(THREADS) prologue_kernel(const float *x, float *y, int H, int W) {
    __shared__ __attribute__((aligned(16))) float4 cells[38 * 64 * (CH / 4)];
    const int HW = H * W, c0 = blockIdx.x * CH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = THREADS / 64, kRowsPerWave = (38 + kWaves - 1) / kWaves;
    const buf_t xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, 512 * 38 * 63 * 4, 0x00020000);
    float v[kRowsPerWave][CH];
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
        const int h = wave + i * kWaves;
        const uint32_t base = (h < H && lane < W) ? (uint32_t)((c0 * HW + h * W + lane) * 4) : 0x80000000u;
#pragma unroll
        for (int c = 0; c < CH; ++c) v[i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, (int)(base + (uint32_t)(c * HW * 4)), 0, 0));
    }
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
        const int h = wave + i * kWaves;
        if (h < H) {
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) cells[(h * 64 + lane) * (CH / 4) + q] = make_float4(v[i][4 * q], v[i][4 * q + 1], v[i][4 * q + 2], v[i][4 * q + 3]);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) y[blockIdx.x + gridDim.x * blockIdx.y] = cells[(blockIdx.x * 7) % (38 * 64)].x;
}

template <typename F>
static void time_graph(const char *name, F launch, hipStream_t s) {
    constexpr int kLaunches = 10, kReps = 30;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < kLaunches; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    std::vector<float> us;
    for (int r = 0; r < kReps; ++r) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); us.push_back(ms * 1000.f / kLaunches);
    }
    std::sort(us.begin(), us.end());
    printf("%-44s best %7.2f us  median %7.2f us\n", name, us[0], us[us.size() / 2]);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t out_bytes = (size_t)300 * 512 * 49 * 4;             // 30.1 MB
    const uint32_t n16 = (uint32_t)(out_bytes / 16);
    float *ys[10];
    for (int i = 0; i < 10; ++i) CK(hipMalloc(&ys[i], out_bytes));
    float *x; CK(hipMalloc(&x, 512 * 38 * 63 * 4)); CK(hipMemset(x, 0, 512 * 38 * 63 * 4));
    // clock ramp
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(HIP_KERNEL_NAME(fill_kernel<0>), dim3(256), dim3(1024), 0, s, ys[i % 10], n16);
    CK(hipStreamSynchronize(s));
    time_graph("empty 256x1024", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, s, ys[0]); }, s);
    time_graph("empty 1024x256", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, s, ys[0]); }, s);
    time_graph("empty 4096x256", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(4096), dim3(256), 0, s, ys[0]); }, s);
#define FILL(AUX, G, B, NAME) time_graph(NAME, [&](int i) { hipLaunchKernelGGL(HIP_KERNEL_NAME(fill_kernel<AUX>), dim3(G), dim3(B), 0, s, ys[i], n16); }, s)
    FILL(0, 256, 1024, "fill 30.1MB plain        256x1024");
    FILL(2, 256, 1024, "fill 30.1MB nt           256x1024");
    FILL(16, 256, 1024, "fill 30.1MB sc1          256x1024");
    FILL(17, 256, 1024, "fill 30.1MB sc0 sc1      256x1024");
    FILL(19, 256, 1024, "fill 30.1MB sc0 sc1 nt   256x1024");
    FILL(0, 2048, 256, "fill 30.1MB plain        2048x256");
    FILL(2, 2048, 256, "fill 30.1MB nt           2048x256");
    FILL(16, 2048, 256, "fill 30.1MB sc1          2048x256");
    FILL(17, 2048, 256, "fill 30.1MB sc0 sc1      2048x256");
    FILL(0, 7350, 256, "fill 30.1MB plain        7350x256 (1 store/thread)");
    FILL(2, 7350, 256, "fill 30.1MB nt           7350x256 (1 store/thread)");
    // same buffer every launch (30 MB fits the 32 MB of L2 / the 256 MB Infinity Cache)
    time_graph("fill 30.1MB plain same buffer 256x1024", [&](int) { hipLaunchKernelGGL(HIP_KERNEL_NAME(fill_kernel<0>), dim3(256), dim3(1024), 0, s, ys[0], n16); }, s);
    time_graph("fill 30.1MB nt    same buffer 256x1024", [&](int) { hipLaunchKernelGGL(HIP_KERNEL_NAME(fill_kernel<2>), dim3(256), dim3(1024), 0, s, ys[0], n16); }, s);
#define PAT(AUX, X, NAME) time_graph(NAME, [&](int i) { hipLaunchKernelGGL(HIP_KERNEL_NAME(pattern_kernel<AUX, X>), dim3(128, 2), dim3(1024), 0, s, ys[i], 300, 128); }, s)
    PAT(0, false, "pattern 4x196B dword plain");
    PAT(2, false, "pattern 4x196B dword nt");
    PAT(16, false, "pattern 4x196B dword sc1");
    PAT(17, false, "pattern 4x196B dword sc0 sc1");
    PAT(0, true, "pattern 4x196B dword plain   xcd-contiguous");
    PAT(2, true, "pattern 4x196B dword nt      xcd-contiguous");
    PAT(16, true, "pattern 4x196B dword sc1     xcd-contiguous");
#define PAT16(AUX, NAME) time_graph(NAME, [&](int i) { hipLaunchKernelGGL(HIP_KERNEL_NAME(pattern16_kernel<AUX>), dim3(128, 2), dim3(1024), 0, s, ys[i], 300, 128); }, s)
    PAT16(0, "pattern 784B x4-dword plain");
    PAT16(2, "pattern 784B x4-dword nt");
    PAT16(16, "pattern 784B x4-dword sc1");
#define RUN(AUX, RUN16, X, NAME) time_graph(NAME, [&](int i) { hipLaunchKernelGGL(HIP_KERNEL_NAME(run_kernel<AUX, RUN16, X>), dim3(256), dim3(1024), 0, s, ys[i], 300); }, s)
    RUN(0, 49, false, "run  784B plain"); RUN(16, 49, false, "run  784B sc1"); RUN(0, 49, true, "run  784B plain xcd"); RUN(16, 49, true, "run  784B sc1 xcd");
    RUN(0, 98, false, "run 1568B plain"); RUN(16, 98, false, "run 1568B sc1"); RUN(0, 98, true, "run 1568B plain xcd"); RUN(16, 98, true, "run 1568B sc1 xcd");
    RUN(2, 98, false, "run 1568B nt");
    RUN(0, 196, false, "run 3136B plain"); RUN(16, 196, false, "run 3136B sc1"); RUN(16, 196, true, "run 3136B sc1 xcd");
    RUN(0, 392, false, "run 6272B plain"); RUN(16, 392, false, "run 6272B sc1"); RUN(16, 392, true, "run 6272B sc1 xcd");
#define HYB(RUN16, X, NAME) time_graph(NAME, [&](int i) { hipLaunchKernelGGL(HIP_KERNEL_NAME(hybrid_kernel<RUN16, X>), dim3(256), dim3(1024), 0, s, ys[i], 300); }, s)
    HYB(49, false, "hybrid  784B"); HYB(49, true, "hybrid  784B xcd"); HYB(98, false, "hybrid 1568B"); HYB(98, true, "hybrid 1568B xcd"); HYB(196, false, "hybrid 3136B");
    time_graph("prologue 8ch 1024thr x 256 wg", [&](int) { hipLaunchKernelGGL(HIP_KERNEL_NAME(prologue_kernel<8, 1024>), dim3(64, 4), dim3(1024), 0, s, x, ys[0], 38, 63); }, s);
    time_graph("prologue 4ch 1024thr x 256 wg", [&](int) { hipLaunchKernelGGL(HIP_KERNEL_NAME(prologue_kernel<4, 1024>), dim3(128, 2), dim3(1024), 0, s, x, ys[0], 38, 63); }, s);
    time_graph("prologue 4ch  512thr x 512 wg", [&](int) { hipLaunchKernelGGL(HIP_KERNEL_NAME(prologue_kernel<4, 512>), dim3(128, 4), dim3(512), 0, s, x, ys[0], 38, 63); }, s);
    time_graph("prologue 4ch  256thr x 768 wg", [&](int) { hipLaunchKernelGGL(HIP_KERNEL_NAME(prologue_kernel<4, 256>), dim3(128, 6), dim3(256), 0, s, x, ys[0], 38, 63); }, s);
    time_graph("prologue 4ch  256thr x 1536 wg", [&](int) { hipLaunchKernelGGL(HIP_KERNEL_NAME(prologue_kernel<4, 256>), dim3(128, 12), dim3(256), 0, s, x, ys[0], 38, 63); }, s);
    return 0;
}
