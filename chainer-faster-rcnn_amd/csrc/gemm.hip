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
#include <stdlib.h>
#include <frcnn_buffer.h>   // angle brackets: shadowed by the test emulator
#include "frcnn_reduce.h"

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

// LDS-DMA form (K a multiple of 32).  A 32-k panel row is 128 B = eight 16-byte groups; pieces of 64 groups (8 rows) go from L2
// straight into LDS (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass -- the register-staged kernel above spends
// 36 ds_write_b32 per thread per panel on its odd-pitch image).  The image is lane-linear with an XOR swizzle on both sides:
// group g of row r sits in slot 8r + (g ^ ((r >> 1) & 7)), which puts the ds_read_b128 of any 16-lane group (16 different rows, one
// g) on 16 distinct bank slots.  A lane reads ONE float4 per operand for FOUR MFMA k-steps: lanes 0-31 take group 2j, lanes 32-63
// group 2j+1, and k-step t of the quad contracts element t of both -- i.e. k = 8j+t and 8j+4+t; A and B use the same map, so the
// product is the same sum in a different (fixed) order.  4x fewer LDS instructions per MFMA than the b32 fragment reads.
// TRN (round 6; the default where N % 4 == 0, FRCNN_LINEAR_F32_TRN=0 switches it off): the MFMA's operands swapped -- A = the weight value, B = the x value -- so the
// accumulator tile is the transpose (register r of lane l = output COLUMN (r & 3) + 8 (r >> 2) + 4 khalf of output ROW l31: the same two products per
// element and k-step, summed by the same instruction) and four consecutive registers are four consecutive floats of one slab row: 16-byte slab stores
// (csrc/linear_bf16.hip's form; bit-identical results).
template <int AM, bool TRN = false>
__global__ void __launch_bounds__(256, 2)
linear_dma_f32_kernel(const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ part, int M, int N, int K,
                      int k_per_split) {
    constexpr int BM = 32 * AM, BN = 128;
    constexpr int XP = BM / 8, WP = BN / 8;                  // 1 KB pieces per panel (8 rows each)
    constexpr int PPW = (XP + WP) / 4;                       // pieces per wave: (4*AM + 16) / 4
    constexpr int STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);
    const int nchunks = (k_end - k_begin) / kBK;             // whole panels only: the host guarantees K % 32 == 0
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)M * K * sizeof(float)));
    const frcnn_buf_t wbuf = frcnn_make_buf(w, (uint32_t)((size_t)N * K * sizeof(float)));

    // source offset (panel 0 of this split) of the 16 bytes this lane contributes to each of its wave's pieces
    uint32_t poff[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pid = wave + 4 * q;                        // pieces 0 .. XP-1: x rows, then W rows
        const bool isx = pid < XP;
        const int sl = (isx ? pid : pid - XP) * 64 + lane, row = sl >> 3, g = (sl & 7) ^ ((row >> 1) & 7);
        const int gr = (isx ? m0 : n0) + row;
        poff[q] = gr < (isx ? M : N) ? (uint32_t)(((size_t)gr * K + k_begin + 4 * g) * sizeof(float)) : kBufOob;
    }
    auto issue = [&](int chunk, int stage) {
        unsigned char *dst = lds + stage * STAGE + wave * 1024;
        const uint32_t so = (uint32_t)chunk * (kBK * sizeof(float));
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            if (4 * q + 3 < XP) frcnn_buf_load_lds_b128(xbuf, dst + q * 4096, poff[q], so);
            else if (4 * q >= XP) frcnn_buf_load_lds_b128(wbuf, dst + q * 4096, poff[q], so);
            else frcnn_buf_load_lds_b128(wave + 4 * q < XP ? xbuf : wbuf, dst + q * 4096, poff[q], so);
        }
    };

    f32x16 acc[AM];
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int l31 = lane & 31, khalf = lane >> 5;
    // byte offset of (row, group 2j + khalf) inside a stage: row * 128 + ((2j + khalf) ^ ((row >> 1) & 7)) * 16
    auto frag_off = [&](int row, int j) { return (uint32_t)(row * 128 + (((2 * j + khalf) ^ ((row >> 1) & 7)) << 4)); };

    if (nchunks > 0) issue(0, 0);
    frcnn_wait_vmcnt<0>();
    frcnn_barrier_nofence();
    int cur = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 < nchunks) issue(chunk + 1, cur ^ 1);
        const unsigned char *xs = lds + cur * STAGE, *wsm = xs + BM * 128;
        float4 a[2][AM], b[2];
        auto frag = [&](int j, float4 (&aa)[AM], float4 &bb) {
            bb = *reinterpret_cast<const float4 *>(wsm + frag_off(wave * 32 + l31, j));
#pragma unroll
            for (int i = 0; i < AM; ++i) aa[i] = *reinterpret_cast<const float4 *>(xs + frag_off(32 * i + l31, j));
        };
        frag(0, a[0], b[0]);
#pragma unroll
        for (int j = 0; j < kBK / 8; ++j) {
            if (j + 1 < kBK / 8) frag(j + 1, a[(j + 1) & 1], b[(j + 1) & 1]);       // next quad's fragments land under this quad's MFMAs
            const float4 bq = b[j & 1];
            const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < AM; ++i) {
                    const float4 aq = a[j & 1][i];
                    const float av = t == 0 ? aq.x : (t == 1 ? aq.y : (t == 2 ? aq.z : aq.w));
                    acc[i] = TRN ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t], av, acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[i], 0, 0, 0);
                }
        }
        frcnn_wait_vmcnt<0>();
        frcnn_barrier_nofence();
        cur ^= 1;
    }
    float *out = part + (size_t)blockIdx.z * M * N;
    if constexpr (TRN) {
        // lane = slab row m, registers 4 g .. 4 g + 3 = columns n0 + 32 wave + 8 g + 4 khalf .. + 3 (N % 4 == 0: a quad lies inside N or outside)
        const frcnn_buf_t pbuf = frcnn_make_buf(out, (uint32_t)((size_t)M * N * 4));
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            const int m = m0 + 32 * i + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nn = n0 + wave * 32 + 8 * g + 4 * khalf;
                const uint32_t off = (m < M && nn < N) ? (uint32_t)(m * N + nn) * 4u : kBufOob;
                frcnn_buf_store_f32x4_soff<0>(pbuf, off, 0u, make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]));
            }
        }
        return;
    }
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
        const float b = bias ? bias[i % N] : 0.0f;           // issued ahead of the slab loads: it is used last
        float v = frcnn_sum_splits(part, total, i, splits);
        v += b;
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
    const int forced = frcnn_tune_int("FRCNN_LINEAR_F32_SPLITS", 0);        // A/B hook (scripts/fc_bench.py --splits)
    if (forced > 0) splits = forced;
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
    if (!x || !w || !y || M < 1 || N < 1 || K < 1 || (K % 4) != 0) return FRCNN_ERR_INVALID;
    const LinearPlan p = plan_linear(M, N, K);
    // one K slab, no bias, no activation: the slab IS the result -- the GEMM writes y and the combine pass (for fc6's weight gradient: 411 MB
    // written, read and written again) is not launched
    const bool direct = p.splits == 1 && !bias && !relu;
    if (!direct && (!workspace || workspace_bytes < (size_t)p.splits * M * N * sizeof(float))) return FRCNN_ERR_INVALID;
    float *part = direct ? y : (float *)workspace;
    const dim3 grid(p.nblocks, p.mblocks, p.splits);
    const bool dma = (K % kBK) == 0 && (size_t)M * K * 4 < (1ull << 31) && (size_t)N * K * 4 < (1ull << 31) && !frcnn_tune("FRCNN_LINEAR_NODMA");
    const bool trn = dma && (N & 3) == 0 && (size_t)M * N * 4 < (1ull << 31) && frcnn_tune_int("FRCNN_LINEAR_F32_TRN", 1) != 0;      // (A/B hook)
    if (trn && p.am == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_f32_kernel<5, true>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (trn && p.am == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_f32_kernel<3, true>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (trn) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_f32_kernel<1, true>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (dma && p.am == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_f32_kernel<5>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (dma && p.am == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_f32_kernel<3>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (dma) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_f32_kernel<1>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (p.am == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_f32_kernel<5>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (p.am == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_f32_kernel<3>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_f32_kernel<1>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    if (direct) return frcnn_launch_status();
    const size_t total = (size_t)M * N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(linear_reduce_kernel, dim3(blocks), dim3(256), 0, stream, part, bias, y, M, N, p.splits, relu);
    return frcnn_launch_status();
}

}  // extern "C"
