// gemm.hip -- fully connected head for gfx950: y(M,N) = act(x(M,K) @ W(N,K)^T + b) in exact fp32 on MFMA.
//
// Replaces L.Linear + F.relu of the RCNN head (/root/reference/models/faster_rcnn.py:33-36,127-134):
// fc6 (300 x 25088 -> 4096; 61.7 GFLOP, 411 MB of weights), fc7, cls_score, bbox_pred.
//
// v_mfma_f32_32x32x2_f32 with A = x rows (lane l -> x[m0 + (l&31)][k + (l>>5)]) and B = W rows
// (lane l -> W[n0 + (l&31)][k + (l>>5)]): D[m][n] has n on the lanes, so y leaves in 128-B runs.
// Both operands are K-contiguous in memory; 32-wide K panels are staged through LDS with an odd pitch (33)
// so the row-strided fragment reads (and the scalar staging writes) are bank-conflict free.
// M = 300 gives only 2 x 32 workgroup tiles, so K is split across workgroups (split-K partial slabs in the
// caller's workspace, combined by linear_reduce_kernel together with bias + ReLU): the decomposition
// yields 2048 equal wave-level work units for fc6 = exactly two per SIMD.
#include "frcnn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kBK = 32, kPitch = 33;

// block = 4 waves along N; wave tile = (32*AM) x 32; block tile = (32*AM) x 128
template <int AM>
__global__ void __launch_bounds__(256)
linear_mfma_f32_kernel(const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ part, int M, int N, int K,
                       int k_per_split) {
    constexpr int BM = 32 * AM, BN = 128, NT = 256;
    constexpr int XV = BM * kBK / 4, WV = BN * kBK / 4;
    constexpr int XIT = (XV + NT - 1) / NT, WIT = (WV + NT - 1) / NT;
    __shared__ float xs[2][BM][kPitch];
    __shared__ float ws[2][BN][kPitch];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);
    const int nchunks = (k_end - k_begin + kBK - 1) / kBK;

    float4 xreg[XIT], wreg[WIT];
    auto fetch = [&](int chunk) {
        const int kb = k_begin + chunk * kBK;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int v = tid + it * NT;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < XV) {
                const int row = v / (kBK / 4), k4 = v % (kBK / 4);
                const int m = m0 + row, k = kb + k4 * 4;
                if (m < M && k < k_end) q = *reinterpret_cast<const float4 *>(x + (size_t)m * K + k);
            }
            xreg[it] = q;
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int v = tid + it * NT;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < WV) {
                const int row = v / (kBK / 4), k4 = v % (kBK / 4);
                const int n = n0 + row, k = kb + k4 * 4;
                if (n < N && k < k_end) q = *reinterpret_cast<const float4 *>(w + (size_t)n * K + k);
            }
            wreg[it] = q;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int v = tid + it * NT;
            if (v < XV) {
                float *d = &xs[buf][v / (kBK / 4)][(v % (kBK / 4)) * 4];
                d[0] = xreg[it].x; d[1] = xreg[it].y; d[2] = xreg[it].z; d[3] = xreg[it].w;
            }
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int v = tid + it * NT;
            if (v < WV) {
                float *d = &ws[buf][v / (kBK / 4)][(v % (kBK / 4)) * 4];
                d[0] = wreg[it].x; d[1] = wreg[it].y; d[2] = wreg[it].z; d[3] = wreg[it].w;
            }
        }
    };

    f32x16 acc[AM];
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    if (nchunks > 0) {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    const int l31 = lane & 31, khalf = lane >> 5;
    int cur = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) fetch(chunk + 1);
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            const float b = ws[cur][wave * 32 + l31][kk + khalf];
            float a[AM];
#pragma unroll
            for (int i = 0; i < AM; ++i) a[i] = xs[cur][32 * i + l31][kk + khalf];
#pragma unroll
            for (int i = 0; i < AM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b, acc[i], 0, 0, 0);
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // partial slab of this K split: part[z][m][n]
    float *out = part + (size_t)blockIdx.z * M * N;
    const int n = n0 + wave * 32 + l31;
    if (n < N) {
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (m < M) out[(size_t)m * N + n] = acc[i][r];
            }
    }
}

__global__ void __launch_bounds__(256)
linear_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias, float *__restrict__ y, int M, int N, int splits,
                     int relu) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.0f;
        for (int s = 0; s < splits; ++s) v += part[(size_t)s * total + i];
        v += bias[i % N];
        if (relu) v = fmaxf(v, 0.0f);
        y[i] = v;
    }
}

struct LinearPlan { int am, mblocks, nblocks, splits, k_per_split; };

static LinearPlan plan_linear(int M, int N, int K) {
    LinearPlan p;
    p.am = (M > 96) ? 5 : (M > 32 ? 3 : 1);
    const int bm = 32 * p.am;
    p.mblocks = frcnn_cdiv(M, bm);
    p.nblocks = frcnn_cdiv(N, 128);
    const int tiles = p.mblocks * p.nblocks;
    const int kchunks = frcnn_cdiv(K, kBK);
    int splits = frcnn_cdiv(512, tiles);
    if (splits > kchunks / 4) splits = kchunks / 4;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    p.k_per_split = frcnn_cdiv(kchunks, splits) * kBK;
    p.splits = frcnn_cdiv(K, p.k_per_split);
    return p;
}

}  // namespace

extern "C" {

size_t frcnn_linear_workspace_bytes(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    const LinearPlan p = plan_linear(M, N, K);
    return frcnn_align256((size_t)p.splits * M * N * sizeof(float));
}

int frcnn_linear_f32(const float *x, const float *w, const float *bias, float *y, int M, int N, int K, int relu, void *workspace,
                     size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w || !bias || !y || M < 1 || N < 1 || K < 1 || (K % 4) != 0) return FRCNN_ERR_INVALID;
    const LinearPlan p = plan_linear(M, N, K);
    if (!workspace || workspace_bytes < (size_t)p.splits * M * N * sizeof(float)) return FRCNN_ERR_INVALID;
    float *part = (float *)workspace;
    const dim3 grid(p.nblocks, p.mblocks, p.splits);
    if (p.am == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_f32_kernel<5>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (p.am == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_f32_kernel<3>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_f32_kernel<1>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    const size_t total = (size_t)M * N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(linear_reduce_kernel, dim3(blocks), dim3(256), 0, stream, part, bias, y, M, N, p.splits, relu);
    return frcnn_launch_status();
}

}  // extern "C"
