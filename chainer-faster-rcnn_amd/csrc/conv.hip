// conv.hip -- the batch-1 NCHW convolution stack for gfx950 in exact fp32 on the matrix cores.
//
// Replaces the reference's L.Convolution2D(ci, co, 3, 1, 1) + F.ReLU pairs and F.MaxPooling2D(2, 2)
// (/root/reference/models/vgg16.py:39-68, region_proposal_network.py:53,117) and the RPN's two 1x1 heads +
// 18-way softmax (region_proposal_network.py:55-57,118-120).
//
// conv3x3: implicit GEMM  D[co][p] = sum_k Wp[k][co] * Xcol[k][p],  k = ci*9 + tap, on
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-identical to an fmaf chain; 157 TF/s peak).
//   * MFMA "A" = weights: lane l supplies Wp[k0 + (l>>5)][co0 + (l&31)];
//     MFMA "B" = activations: lane l supplies X[ci0 + (l>>5)][y + ky - 1][x0 + (l&31) + kx - 1].
//     With batch 1 the pixel index is the contiguous axis of NCHW, so lanes 0-31 of a B fragment read 32
//     consecutive floats and the D fragment (lane = pixel, register = cout) stores 128-B runs: NCHW is the
//     native layout of this instruction; no layout transform at the boundary.
//   * A workgroup owns BCO couts x BROWS rows x 32 pixels.  Per K-chunk of CK input channels it stages the
//     (BROWS+2) x 34 halo of each channel and the CK*9 x BCO weight panel in LDS; the 9 taps then re-read
//     the halo from LDS (9x reuse of every global read).  Fragment reads are ds_read_b32 over 32
//     consecutive dwords per half-wave: bank-conflict free by construction.
//   * Register-staged double buffering: chunk t+1 is fetched into VGPRs while chunk t feeds the MFMAs and
//     written to the other LDS buffer afterwards -- one barrier per chunk.
//   * Packed weights Wp = (Cin*9, Cout) row-major (frcnn_pack_conv3x3_w, once at load time).
// Work decomposition is chosen per layer (pick_conv_config) so that the number of wave-level work units
// comfortably exceeds the 1024 SIMDs even for the 38x63 layers.
//
// Roofline: compute-bound everywhere except conv1_1 (K = 27: 153.6 MB of output for 2 GFLOP).
#include "frcnn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// KS = 3: 3x3 / pad 1 (the VGG and RPN convs);  KS = 1: 1x1 / pad 0 (the RPN heads) -- same machinery, no halo.
template <int KS, int WCO, int WPX, int ACO, int APX, int CK>
__global__ void __launch_bounds__(64 * WCO * WPX)
conv_mfma_f32_kernel(const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
                        float *__restrict__ y, int Cin, int Cout, int H, int W, int relu) {
    constexpr int NT = 64 * WCO * WPX;
    constexpr int BCO = 32 * ACO * WCO;
    constexpr int BROWS = APX * WPX;
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    constexpr int kHaloPitch = 32 + KS - 1;          // 32 pixels + left/right halo
    constexpr int HR = BROWS + KS - 1;
    constexpr int KR = CK * TAPS;
    constexpr int WV = KR * BCO / 4;                 // float4s of weights per chunk
    constexpr int WIT = (WV + NT - 1) / NT;
    constexpr int HV = CK * HR * kHaloPitch;         // halo floats per chunk
    constexpr int HIT = (HV + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float w_lds[2][KR][BCO];
    __shared__ __attribute__((aligned(16))) float in_lds[2][CK][HR][kHaloPitch];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % WCO, wpx = wave / WCO;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * BROWS, co0 = blockIdx.z * BCO;
    const int HW = H * W;
    const int K = Cin * TAPS;
    const int nchunks = (Cin + CK - 1) / CK;

    float4 wreg[WIT];
    float hreg[HIT];

    auto fetch = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int v = tid + it * NT;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < WV) {
                const int row = v / (BCO / 4), c4 = v % (BCO / 4);
                const int grow = chunk * KR + row;
                if (grow < K) q = *reinterpret_cast<const float4 *>(wp + (size_t)grow * Cout + co0 + c4 * 4);
            }
            wreg[it] = q;
        }
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int e = tid + it * NT;
            float val = 0.0f;
            if (e < HV) {
                const int c = e / (HR * kHaloPitch), rem = e % (HR * kHaloPitch);
                const int hr = rem / kHaloPitch, hx = rem % kHaloPitch;
                const int gc = chunk * CK + c, gy = y0 - PAD + hr, gx = x0 - PAD + hx;
                if (gc < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) val = x[(size_t)gc * HW + gy * W + gx];
            }
            hreg[it] = val;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int v = tid + it * NT;
            if (v < WV) reinterpret_cast<float4 *>(&w_lds[buf][0][0])[v] = wreg[it];
        }
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int e = tid + it * NT;
            if (e < HV) (&in_lds[buf][0][0][0])[e] = hreg[it];
        }
    };

    f32x16 acc[ACO][APX];
#pragma unroll
    for (int i = 0; i < ACO; ++i)
#pragma unroll
        for (int j = 0; j < APX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    fetch(0);
    stage(0);
    __syncthreads();

    const int l31 = lane & 31, khalf = lane >> 5;
    const int a_col = wco * (32 * ACO) + l31;
    const int b_row = wpx * APX;
    int cur = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) fetch(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
#pragma unroll
            for (int cp = 0; cp < CK / 2; ++cp) {
                const int c = 2 * cp + khalf;
                float a[ACO], b[APX];
#pragma unroll
                for (int i = 0; i < ACO; ++i) a[i] = w_lds[cur][c * TAPS + tap][a_col + 32 * i];
#pragma unroll
                for (int j = 0; j < APX; ++j) b[j] = in_lds[cur][c][b_row + j + ky][l31 + kx];
#pragma unroll
                for (int i = 0; i < ACO; ++i)
#pragma unroll
                    for (int j = 0; j < APX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: D register r of lane l = cout (r&3) + 8*(r>>2) + 4*(l>>5), pixel l&31
    const int px = x0 + l31;
#pragma unroll
    for (int i = 0; i < ACO; ++i) {
#pragma unroll
        for (int j = 0; j < APX; ++j) {
            const int py = y0 + b_row + j;
            if (px < W && py < H) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco * (32 * ACO) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    float v = acc[i][j][r] + bias[co];
                    if (relu) v = fmaxf(v, 0.0f);
                    y[(size_t)co * HW + (size_t)py * W + px] = v;
                }
            }
        }
    }
}

// (Cout, Cin, 3, 3) -> (Cin*9, Cout)
__global__ void __launch_bounds__(256)
pack_conv3x3_w_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = Cout * Cin * 9;
    if (i >= total) return;
    const int co = i % Cout, k = i / Cout;            // destination index: k*Cout + co
    wp[i] = w[(size_t)co * Cin * 9 + k];
}

// F.MaxPooling2D(2, 2) with cover_all=True: output ceil(H/2) x ceil(W/2), windows clipped at the border.
__global__ void __launch_bounds__(256)
maxpool2x2_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int H, int W, int OH, int OW) {
    const size_t total = (size_t)C * OH * OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ow = (int)(i % OW), oh = (int)((i / OW) % OH), c = (int)(i / ((size_t)OW * OH));
        const float *p = x + ((size_t)c * H + 2 * oh) * W + 2 * ow;
        float m = p[0];
        const bool hasx = 2 * ow + 1 < W, hasy = 2 * oh + 1 < H;
        if (hasx) m = fmaxf(m, p[1]);
        if (hasy) { m = fmaxf(m, p[W]); if (hasx) m = fmaxf(m, p[W + 1]); }
        y[i] = m;
    }
}

// RPN heads.  rpn_cls_score (2A) and rpn_bbox_pred (4A) are ONE 1x1 convolution with the two weight matrices
// stacked and zero-padded to NP = 64 output channels (rpn_heads_pack_kernel, once at load); it runs on the
// MFMA kernel above (KS = 1) and writes raw[(NP, HW)]: rows [0,2A) are rpn_cls_score, rows [2A,6A) are
// rpn_bbox_pred -- both contiguous NCHW blocks, handed out as views.  softmax_channels_kernel then applies
// the reference's softmax over ALL 2A score channels (region_proposal_network.py:119), one thread per pixel.
__global__ void __launch_bounds__(256)
rpn_heads_pack_kernel(const float *__restrict__ w_cls, const float *__restrict__ b_cls, const float *__restrict__ w_bbox,
                      const float *__restrict__ b_bbox, int Cmid, int A, int NP, float *__restrict__ wp, float *__restrict__ bp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Cmid * NP) {
        const int o = i % NP, c = i / NP;
        float v = 0.0f;
        if (o < 2 * A) v = w_cls[(size_t)o * Cmid + c];
        else if (o < 6 * A) v = w_bbox[(size_t)(o - 2 * A) * Cmid + c];
        wp[i] = v;
    }
    if (i < NP) bp[i] = (i < 2 * A) ? b_cls[i] : (i < 6 * A ? b_bbox[i - 2 * A] : 0.0f);
}

__global__ void __launch_bounds__(256)
softmax_channels_kernel(const float *__restrict__ score, int n_ch, int HW, float *__restrict__ prob) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float m = score[p];
    for (int c = 1; c < n_ch; ++c) m = fmaxf(m, score[(size_t)c * HW + p]);
    float sum = 0.0f;
    for (int c = 0; c < n_ch; ++c) sum += expf(score[(size_t)c * HW + p] - m);
    for (int c = 0; c < n_ch; ++c) prob[(size_t)c * HW + p] = expf(score[(size_t)c * HW + p] - m) / sum;
}

// ---- per-layer work decomposition ------------------------------------------------------------------
// cfg 0: 64co x (8 rows x 32 px), 4 waves 1x4, wave 64co x 2 rows   -- Cout == 64 layers
// cfg 1: 128co x (4 rows x 32 px), 4 waves 2x2, wave 64co x 2 rows  -- large maps
// cfg 2: 64co x (4 rows x 32 px), 4 waves 2x2, wave 32co x 2 rows   -- mid maps (more, smaller units)
// cfg 3: 64co x (2 rows x 32 px), 4 waves 2x2, wave 32co x 1 row    -- 38x63 / 75x125 maps
template <int KS, int WCO, int WPX, int ACO, int APX, int CK>
static int launch_conv(const float *x, const float *wp, const float *bias, float *y, int Cin, int Cout, int H, int W, int relu,
                       hipStream_t stream) {
    constexpr int BCO = 32 * ACO * WCO, BROWS = APX * WPX;
    if (Cout % BCO != 0) return FRCNN_ERR_INVALID;
    const dim3 grid(frcnn_cdiv(W, 32), frcnn_cdiv(H, BROWS), Cout / BCO);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_f32_kernel<KS, WCO, WPX, ACO, APX, CK>), grid, dim3(64 * WCO * WPX), 0, stream, x, wp,
                       bias, y, Cin, Cout, H, W, relu);
    return frcnn_launch_status();
}

static int pick_conv_config(int Cin, int Cout, int H, int W) {
    (void)Cin;
    const long units22 = (long)frcnn_cdiv(W, 32) * frcnn_cdiv(H, 2) * (Cout / 64);   // 64co x 2rows wave units
    if (Cout % 128 == 0 && units22 >= 4096) return 1;
    if (Cout == 64 && units22 >= 4096) return 0;
    if (units22 >= 4096) return 2;
    return 3;
}

}  // namespace

extern "C" {

int frcnn_rpn_heads_padded_channels(int A) { return ((6 * A + 63) / 64) * 64; }

int frcnn_pack_conv3x3_w(const float *w, int Cout, int Cin, float *w_packed, void *stream) {
    if (!w || !w_packed || Cout < 1 || Cin < 1) return FRCNN_ERR_INVALID;
    const int total = Cout * Cin * 9;
    hipLaunchKernelGGL(pack_conv3x3_w_kernel, dim3(frcnn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, w_packed);
    return frcnn_launch_status();
}

int frcnn_conv3x3_f32_cfg(const float *x, const float *w_packed, const float *bias, float *y, int Cin, int Cout, int H, int W,
                          int relu, int cfg, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w_packed || !bias || !y || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (Cout % 4) != 0) return FRCNN_ERR_INVALID;
    if (cfg < 0) cfg = pick_conv_config(Cin, Cout, H, W);
    switch (cfg) {
        case 0: return launch_conv<3, 1, 4, 2, 2, 4>(x, w_packed, bias, y, Cin, Cout, H, W, relu, stream);
        case 1: return launch_conv<3, 2, 2, 2, 2, 4>(x, w_packed, bias, y, Cin, Cout, H, W, relu, stream);
        case 2: return launch_conv<3, 2, 2, 1, 2, 8>(x, w_packed, bias, y, Cin, Cout, H, W, relu, stream);
        case 3: return launch_conv<3, 2, 2, 1, 1, 8>(x, w_packed, bias, y, Cin, Cout, H, W, relu, stream);
        default: return FRCNN_ERR_INVALID;
    }
}

int frcnn_conv3x3_f32(const float *x, const float *w_packed, const float *bias, float *y, int Cin, int Cout, int H, int W, int relu,
                      void *stream) {
    return frcnn_conv3x3_f32_cfg(x, w_packed, bias, y, Cin, Cout, H, W, relu, -1, stream);
}

int frcnn_maxpool2x2_f32(const float *x, float *y, int C, int H, int W, void *stream) {
    if (!x || !y || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const size_t total = (size_t)C * OH * OW;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W, OH, OW);
    return frcnn_launch_status();
}

int frcnn_rpn_heads_pack(const float *w_cls, const float *b_cls, const float *w_bbox, const float *b_bbox, int Cmid, int A,
                         float *w_packed, float *b_packed, void *stream) {
    if (!w_cls || !b_cls || !w_bbox || !b_bbox || !w_packed || !b_packed || Cmid < 1 || A < 1) return FRCNN_ERR_INVALID;
    const int NP = frcnn_rpn_heads_padded_channels(A);
    hipLaunchKernelGGL(rpn_heads_pack_kernel, dim3(frcnn_cdiv(Cmid * NP, 256)), dim3(256), 0, (hipStream_t)stream, w_cls, b_cls, w_bbox,
                       b_bbox, Cmid, A, NP, w_packed, b_packed);
    return frcnn_launch_status();
}

int frcnn_rpn_heads_f32(const float *h, int Cmid, int H, int W, int A, const float *w_packed, const float *b_packed, float *raw,
                        float *cls_prob, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h || !w_packed || !b_packed || !raw || !cls_prob || Cmid < 1 || H < 1 || W < 1 || A < 1) return FRCNN_ERR_INVALID;
    const int NP = frcnn_rpn_heads_padded_channels(A);
    const int st = launch_conv<1, 2, 2, 1, 1, 8>(h, w_packed, b_packed, raw, Cmid, NP, H, W, 0, stream);
    if (st != FRCNN_OK) return st;
    hipLaunchKernelGGL(softmax_channels_kernel, dim3(frcnn_cdiv(H * W, 256)), dim3(256), 0, stream, raw, 2 * A, H * W, cls_prob);
    return frcnn_launch_status();
}

}  // extern "C"
