// conv_bf16.hip -- the convolution stack in bf16 on the matrix cores (BASELINE config 3: "bf16 convs / fp32 RoI").
//
// Same reference interface as conv.hip (L.Convolution2D(ci, co, 3, 1, 1) + F.relu, /root/reference/models/vgg16.py:39-68;
// rpn_conv_3x3 and the two 1x1 heads, region_proposal_network.py:53-57), different arithmetic: operands rounded to bf16
// (round to nearest even), products accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).
//
// Layout.  A bf16 MFMA lane supplies EIGHT consecutive k-values of one row/column, so the contraction index must be
// contiguous in memory -- and a K-chunk of 16 channels should still be whole cache lines.  Activations are therefore
// channel-BLOCKED, [C/16][H][W][16] bf16 (a chunk's halo row is one contiguous run of 34 x 32 B), and the weights are packed
// [C/16][tap][cout][16] bf16 (a chunk's panel for 64 couts is 9 contiguous runs of 2 KB), channel counts padded to a multiple
// of 16; k = (tap, cin).  [With plain channel-last tensors every 16-byte staging load touched its own 128-byte line and the
// kernel sat on the CU's global-load rate: ~10 B/clk.]  MFMA A = weights (row = cout), B = activations
// (column = pixel): lane l of a B fragment reads 16 contiguous bytes -- channels 8*(l>>5)..+7 of pixel l&31 -- and
// register r of the D fragment holds cout (r&3)+8*(r>>2)+4*(l>>5) of pixel l&31, i.e. four consecutive couts of one pixel:
// one 8-byte channel-last store per register quad.  The last layer of a bf16 chain can instead write fp32 NCHW
// (out_mode 1), which is what RoI pooling, the 18-way softmax and the proposal kernels consume.
//
// A workgroup (4 waves = 2 cout blocks x 2 row pairs) owns 64 couts x 4 rows x 32 px.  Per 16-channel K-chunk it stages the
// 6 x 34 pixel halo (32 B per pixel) and the 9 x 64 weight rows (32 B per row) in LDS -- pitch 48 B, so the
// ds_read_b128 fragment reads of any 16-lane group fall on 16 distinct 16-byte bank slots (conflict-free) -- register
// staged and double buffered exactly like the fp32 kernel, global reads through buffer descriptors (padding = out of
// range = 0).  At bf16 rates this kernel is bound by LDS and L2 traffic, not by the matrix cores: 3 fragment reads feed 2
// MFMAs (32 cycles each).
#include "frcnn_common.h"
#include <stdlib.h>
#include <frcnn_buffer.h>   // angle brackets: shadowed by the test emulator
#include <frcnn_intrin.h>

namespace {

constexpr int kCK = 16;                 // channels per K-chunk = the MFMA's k extent
constexpr int kPitchB = 48;             // LDS bytes per (pixel | weight row): 32 B of data + 16 B pad

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);        // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                  // round to nearest even
    return (uint16_t)(u >> 16);
}

template <int KS, int RP>           // RP = row pairs per workgroup: 2 -> 4 waves, 64co x 4 rows; 4 -> 8 waves, 64co x 8 rows
__global__ void __launch_bounds__(128 * RP)
conv_mfma_bf16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wp, const float *__restrict__ bias, void *__restrict__ y,
                      int CinP, int Cout, int CoutP, int H, int W, int relu, int out_mode, int xtiles, int ytiles) {
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    constexpr int BROWS = 2 * RP, BCO = 64, NT = 128 * RP;
    constexpr int HR = BROWS + KS - 1, HPX = 32 + KS - 1;
    constexpr int HALO_V = HR * HPX * 2;            // 16-byte vectors of activations per chunk (2 per pixel)
    constexpr int W_V = TAPS * BCO * 2;             // 16-byte vectors of weights per chunk (2 per row)
    constexpr int HIT = (HALO_V + NT - 1) / NT, WIT = (W_V + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned char in_lds[2][HR * HPX * kPitchB];
    constexpr int OP = BCO * 2 + 16;                                     // epilogue tile: LDS bytes per pixel (128 B + pad)
    constexpr int W_BYTES = TAPS * BCO * kPitchB > BROWS * 32 * OP / 2 ? TAPS * BCO * kPitchB : BROWS * 32 * OP / 2;
    __shared__ __attribute__((aligned(16))) unsigned char w_lds[2][W_BYTES];       // weights; re-used by the epilogue transpose
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave & 1, wrow = wave >> 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int tile = blockIdx.x;
    const int tx = tile % xtiles, ty = (tile / xtiles) % ytiles, cot = tile / (xtiles * ytiles);
    const int x0 = tx * 32, y0 = ty * BROWS, co0 = cot * BCO;
    const int nchunks = CinP / kCK;
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)H * W * CinP * 2));
    const frcnn_buf_t wbuf = frcnn_make_buf(wp, (uint32_t)((size_t)TAPS * CoutP * CinP * 2));
    const uint32_t x_chunk_bytes = (uint32_t)(H * W) * 32u, w_chunk_bytes = (uint32_t)(TAPS * CoutP) * 32u;

    // byte offsets (chunk 0) of this thread's staging vectors; chunk c adds c channel blocks
    uint32_t hoff[HIT], woff[WIT];
#pragma unroll
    for (int q = 0; q < HIT; ++q) {
        const int v = tid + q * NT;
        const int pix = v >> 1, half = v & 1;
        const int hr = pix / HPX, hx = pix - hr * HPX;
        const int gy = y0 - PAD + hr, gx = x0 - PAD + hx;
        const bool inside = v < HALO_V && gy >= 0 && gy < H && gx >= 0 && gx < W;
        hoff[q] = inside ? (uint32_t)((gy * W + gx) * 32 + half * 16) : kBufOob;
    }
#pragma unroll
    for (int q = 0; q < WIT; ++q) {
        const int v = tid + q * NT;
        const int row = v >> 1, half = v & 1;                    // row = tap * BCO + co_local
        const int tap = row / BCO, col = row - tap * BCO;
        woff[q] = (v < W_V && co0 + col < CoutP) ? (uint32_t)((tap * CoutP + co0 + col) * 32 + half * 16) : kBufOob;
    }
    // Two register sets: chunk t+2 is fetched while chunk t feeds the MFMAs and chunk t+1 waits in the other set, so a load has
    // two chunk times (plus the co-resident workgroup's) to land -- one chunk of 18 MFMAs is shorter than an L2 round trip.
    float4 hregA[HIT], wregA[WIT], hregB[HIT], wregB[WIT];
    auto fetch = [&](int chunk, float4 (&hreg)[HIT], float4 (&wreg)[WIT]) {
        const uint32_t xb = (uint32_t)chunk * x_chunk_bytes, wb = (uint32_t)chunk * w_chunk_bytes;
#pragma unroll
        for (int q = 0; q < HIT; ++q) hreg[q] = frcnn_buf_load_f32x4(xbuf, hoff[q] + xb);
#pragma unroll
        for (int q = 0; q < WIT; ++q) wreg[q] = frcnn_buf_load_f32x4(wbuf, woff[q] + wb);
    };
    auto stage = [&](int buf, const float4 (&hreg)[HIT], const float4 (&wreg)[WIT]) {
#pragma unroll
        for (int q = 0; q < HIT; ++q) {
            const int v = tid + q * NT;
            if (v < HALO_V) *reinterpret_cast<float4 *>(&in_lds[buf][(v >> 1) * kPitchB + (v & 1) * 16]) = hreg[q];
        }
#pragma unroll
        for (int q = 0; q < WIT; ++q) {
            const int v = tid + q * NT;
            if (v < W_V) *reinterpret_cast<float4 *>(&w_lds[buf][(v >> 1) * kPitchB + (v & 1) * 16]) = wreg[q];
        }
    };

    frcnn_f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // all fragments of a chunk are read before its MFMAs: one LDS round trip per chunk, and the wave's two output rows share
    // the halo rows between them (KS+1 distinct rows of B fragments instead of 2*KS)
    auto compute = [&](int buf) {
        const unsigned char *wl = &w_lds[buf][(wco * 32 + l31) * kPitchB + khalf * 16];
        const unsigned char *il = &in_lds[buf][((wrow * 2) * HPX + l31) * kPitchB + khalf * 16];
        uint4 a[TAPS], b[KS + 1][KS];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) a[tap] = *reinterpret_cast<const uint4 *>(wl + tap * BCO * kPitchB);
#pragma unroll
        for (int r = 0; r < KS + 1; ++r)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) b[r][kx] = *reinterpret_cast<const uint4 *>(il + (r * HPX + kx) * kPitchB);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            acc[0] = frcnn_mfma_32x32x16_bf16(a[tap], b[ky][kx], acc[0]);
            acc[1] = frcnn_mfma_32x32x16_bf16(a[tap], b[ky + 1][kx], acc[1]);
        }
    };

    fetch(0, hregA, wregA);
    if (nchunks > 1) fetch(1, hregB, wregB);
    stage(0, hregA, wregA);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
        // even chunk: LDS buffer 0 holds it, set B holds chunk+1
        if (chunk + 2 < nchunks) fetch(chunk + 2, hregA, wregA);
        compute(0);
        if (chunk + 1 < nchunks) stage(1, hregB, wregB);
        __syncthreads();
        if (chunk + 1 >= nchunks) break;
        // odd chunk: LDS buffer 1 holds it, set A holds chunk+2
        if (chunk + 3 < nchunks) fetch(chunk + 3, hregB, wregB);
        compute(1);
        if (chunk + 2 < nchunks) stage(0, hregA, wregA);
        __syncthreads();
    }

    // epilogue: register r of lane l = cout (r&3) + 8*(r>>2) + 4*khalf of pixel l31
    const int px = x0 + l31;
    if (out_mode == 0 || out_mode == 2) {
        // bf16 channel-blocked output: transpose through LDS so that each 16-cout block of a tile row leaves as one contiguous
        // run of 32 px x 32 B (16-byte stores, consecutive lanes consecutive addresses)
        unsigned char *ot = &w_lds[0][0];                             // the K loop is over: all waves passed its last barrier
        static_assert(BROWS * 32 * OP <= (int)sizeof(w_lds), "epilogue tile must fit in the weight buffers");
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wco * 32 + 8 * g + 4 * khalf;             // first of four consecutive couts (within the tile)
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v[t] = acc[j][4 * g + t] + (co0 + col + t < Cout ? bias[co0 + col + t] : 0.0f);
                    if (relu) v[t] = fmaxf(v[t], 0.0f);
                }
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<uint2 *>(ot + ((wrow * 2 + j) * 32 + l31) * OP + col * 2) = pk;
            }
        __syncthreads();
        if (out_mode == 2) {
            // F.MaxPooling2D(2, 2) (cover_all) fused: tiles start at even rows / columns, so every 2x2 window lies inside the tile;
            // y is [CoutP/16][ceil(H/2)][ceil(W/2)][16].  A maximum of bf16 values is a bf16 value: exact.
            const int OH = (H + 1) / 2, OW = (W + 1) / 2;
            for (int v = tid; v < (BROWS / 2) * 16 * 8; v += NT) {
                const int cbl = v / ((BROWS / 2) * 16 * 2), rem = v - cbl * ((BROWS / 2) * 16 * 2);
                const int opix = rem >> 1, half = rem & 1;
                const int orow = opix >> 4, ocol = opix & 15;
                const int py = y0 + 2 * orow, qx = x0 + 2 * ocol, co = co0 + cbl * 16;
                if (py >= H || qx >= W || co >= CoutP) continue;
                const bool hasx = qx + 1 < W, hasy = py + 1 < H;
                const unsigned char *t0 = ot + ((2 * orow) * 32 + 2 * ocol) * OP + (cbl * 2 + half) * 16;
                uint4 q[4];
                q[0] = *reinterpret_cast<const uint4 *>(t0);
                q[1] = hasx ? *reinterpret_cast<const uint4 *>(t0 + OP) : q[0];
                q[2] = hasy ? *reinterpret_cast<const uint4 *>(t0 + 32 * OP) : q[0];
                q[3] = (hasx && hasy) ? *reinterpret_cast<const uint4 *>(t0 + 33 * OP) : q[0];
                uint32_t o4[4];
                const uint32_t *w0 = reinterpret_cast<const uint32_t *>(&q[0]), *w1 = reinterpret_cast<const uint32_t *>(&q[1]);
                const uint32_t *w2 = reinterpret_cast<const uint32_t *>(&q[2]), *w3 = reinterpret_cast<const uint32_t *>(&q[3]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint32_t r = 0;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int sh = 16 * hh;
                        const float a = __uint_as_float((w0[t] >> sh) << 16), b = __uint_as_float((w1[t] >> sh) << 16);
                        const float c = __uint_as_float((w2[t] >> sh) << 16), d = __uint_as_float((w3[t] >> sh) << 16);
                        r |= ((__float_as_uint(fmaxf(fmaxf(a, b), fmaxf(c, d))) >> 16) & 0xffffu) << sh;
                    }
                    o4[t] = r;
                }
                *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(y) + (((size_t)(co >> 4) * OH + (py >> 1)) * OW + (qx >> 1)) * 16 + half * 8) =
                    make_uint4(o4[0], o4[1], o4[2], o4[3]);
            }
            return;
        }
        for (int v = tid; v < BROWS * 32 * 8; v += NT) {                  // 16-byte vectors: (cout block of 16, pixel, half)
            const int cbl = v / (BROWS * 32 * 2), rem = v - cbl * (BROWS * 32 * 2);
            const int pix = rem >> 1, half = rem & 1;
            const int py = y0 + (pix >> 5), qx = x0 + (pix & 31), co = co0 + cbl * 16;
            if (py < H && qx < W && co < CoutP)
                *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(y) + (((size_t)(co >> 4) * H + py) * W + qx) * 16 + half * 8) =
                    *reinterpret_cast<const uint4 *>(ot + pix * OP + (cbl * 2 + half) * 16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int py = y0 + wrow * 2 + j;
            if (px >= W || py >= H) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + wco * 32 + 8 * g + 4 * khalf;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (co + t < Cout) {
                        float v = acc[j][4 * g + t] + bias[co + t];
                        if (relu) v = fmaxf(v, 0.0f);
                        reinterpret_cast<float *>(y)[(size_t)(co + t) * H * W + (size_t)py * W + px] = v;      // fp32 NCHW
                    }
            }
        }
    }
}

// (Cout, Cin, k, k) fp32 -> [CinP/16][tap][CoutP][16] bf16, zero padded
__global__ void __launch_bounds__(256)
pack_w_bf16_kernel(const float *__restrict__ w, int Cout, int Cin, int taps, int CoutP, int CinP, uint16_t *__restrict__ wp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)taps * CoutP * CinP;
    if (i >= total) return;
    const int c16 = (int)(i % 16), co = (int)((i / 16) % CoutP), tap = (int)((i / (16 * (size_t)CoutP)) % taps);
    const int ci = (int)(i / (16 * (size_t)CoutP * taps)) * 16 + c16;
    wp[i] = (co < Cout && ci < Cin) ? f32_to_bf16(w[((size_t)co * Cin + ci) * taps + tap]) : (uint16_t)0;
}

// (C,H,W) fp32 -> [CP/16][H*W][16] bf16, channels C..CP-1 zero
__global__ void __launch_bounds__(256)
nchw_to_nhwc_bf16_kernel(const float *__restrict__ x, int C, int HW, int CP, uint16_t *__restrict__ y) {
    const size_t total = (size_t)HW * CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c16 = (int)(i % 16);
        const size_t p = (i / 16) % HW;
        const int c = (int)(i / (16 * (size_t)HW)) * 16 + c16;
        y[i] = c < C ? f32_to_bf16(x[(size_t)c * HW + p]) : (uint16_t)0;
    }
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// F.MaxPooling2D(2, 2), cover_all, channel-blocked bf16 [C/16][H][W][16]: one thread = one output pixel x 8 channels (16 bytes)
__global__ void __launch_bounds__(256)
maxpool2x2_bf16_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int C, int H, int W, int OH, int OW) {
    const size_t total = (size_t)(C / 16) * OH * OW * 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int half = (int)(i & 1), ow = (int)((i >> 1) % OW), oh = (int)(((i >> 1) / OW) % OH), cb = (int)((i >> 1) / ((size_t)OW * OH));
        const bool hasx = 2 * ow + 1 < W, hasy = 2 * oh + 1 < H;
        const uint16_t *p = x + ((((size_t)cb * H + 2 * oh) * W + 2 * ow) * 16 + half * 8);
        uint4 q[4];
        q[0] = *reinterpret_cast<const uint4 *>(p);
        q[1] = hasx ? *reinterpret_cast<const uint4 *>(p + 16) : q[0];
        q[2] = hasy ? *reinterpret_cast<const uint4 *>(p + (size_t)W * 16) : q[0];
        q[3] = (hasx && hasy) ? *reinterpret_cast<const uint4 *>(p + (size_t)W * 16 + 16) : q[0];
        uint32_t out[4];
        const uint32_t *w0 = reinterpret_cast<const uint32_t *>(&q[0]), *w1 = reinterpret_cast<const uint32_t *>(&q[1]);
        const uint32_t *w2 = reinterpret_cast<const uint32_t *>(&q[2]), *w3 = reinterpret_cast<const uint32_t *>(&q[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t r = 0;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int sh = 16 * hh;
                const float a = bf16_to_f32((uint16_t)(w0[t] >> sh)), b = bf16_to_f32((uint16_t)(w1[t] >> sh));
                const float c = bf16_to_f32((uint16_t)(w2[t] >> sh)), d = bf16_to_f32((uint16_t)(w3[t] >> sh));
                const float m = fmaxf(fmaxf(a, b), fmaxf(c, d));
                r |= ((__float_as_uint(m) >> 16) & 0xffffu) << sh;          // a maximum of bf16 values is a bf16 value: exact
            }
            out[t] = r;
        }
        *reinterpret_cast<uint4 *>(y + ((((size_t)cb * OH + oh) * OW + ow) * 16 + half * 8)) = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

// [CP/16][H*W][16] bf16 -> (C,H,W) fp32 through a 64x65 LDS tile
__global__ void __launch_bounds__(256)
nhwc_bf16_to_nchw_kernel(const uint16_t *__restrict__ x, int C, int HW, int CP, float *__restrict__ y) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int p = p0 + i, c = c0 + tx;
        tile[i][tx] = (p < HW && c < C) ? bf16_to_f32(x[((size_t)(c >> 4) * HW + p) * 16 + (c & 15)]) : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, p = p0 + tx;
        if (c < C && p < HW) y[(size_t)c * HW + p] = tile[tx][i];
    }
}

}  // namespace

extern "C" {

int frcnn_bf16_to_nchw_f32(const uint16_t *x, int C, int H, int W, float *y, void *stream) {
    if (!x || !y || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int CP = (C + 15) / 16 * 16;
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_kernel, dim3(frcnn_cdiv(H * W, 64), frcnn_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, C, H * W, CP, y);
    return frcnn_launch_status();
}

int frcnn_bf16_padded_channels(int c) { return (c + 15) / 16 * 16; }

int frcnn_bf16_pack_conv_w(const float *w, int Cout, int Cin, int ksize, uint16_t *w_packed, void *stream) {
    if (!w || !w_packed || Cout < 1 || Cin < 1 || (ksize != 1 && ksize != 3)) return FRCNN_ERR_INVALID;
    const int CoutP = frcnn_bf16_padded_channels(Cout), CinP = frcnn_bf16_padded_channels(Cin), taps = ksize * ksize;
    const size_t total = (size_t)taps * CoutP * CinP;
    hipLaunchKernelGGL(pack_w_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, taps, CoutP,
                       CinP, w_packed);
    return frcnn_launch_status();
}

int frcnn_bf16_from_nchw_f32(const float *x, int C, int H, int W, uint16_t *y, void *stream) {
    if (!x || !y || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int CP = frcnn_bf16_padded_channels(C);
    const size_t total = (size_t)H * W * CP;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, C, H * W, CP, y);
    return frcnn_launch_status();
}

int frcnn_conv_bf16(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H, int W, int ksize, int relu,
                    int out_mode, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w_packed || !bias || !y || Cin < 1 || Cout < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if ((ksize != 1 && ksize != 3) || out_mode < 0 || out_mode > 2 || (out_mode == 2 && !relu)) return FRCNN_ERR_INVALID;
    const int CinP = frcnn_bf16_padded_channels(Cin), CoutP = frcnn_bf16_padded_channels(Cout);
    if ((size_t)H * W * CinP * 2 >= (1ull << 31) || (size_t)ksize * ksize * CoutP * CinP * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;
    const int xtiles = frcnn_cdiv(W, 32), cotiles = frcnn_cdiv(CoutP, 64);
    // 8-row tiles (8 waves) carry 1.7x the MFMA work per staged byte but measured no faster than 4-row tiles on any VGG layer
    // (scripts/conv_bf16_sweep.py, r01); kept as a tuning hook: FRCNN_BF16_RP=4 selects them.
    const char *rp_env = getenv("FRCNN_BF16_RP");
    const bool big = rp_env && atoi(rp_env) == 4;
    const int ytiles = frcnn_cdiv(H, big ? 8 : 4);
    const dim3 grid(xtiles * ytiles * cotiles);
    if (ksize == 3 && big) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_bf16_kernel<3, 4>), grid, dim3(512), 0, stream, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles);
    else if (ksize == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_bf16_kernel<3, 2>), grid, dim3(256), 0, stream, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_bf16_kernel<1, 2>), grid, dim3(256), 0, stream, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles);
    return frcnn_launch_status();
}

int frcnn_maxpool2x2_bf16(const uint16_t *x, uint16_t *y, int C, int H, int W, void *stream) {
    if (!x || !y || C < 16 || (C % 16) != 0 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const size_t total = (size_t)OH * OW * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool2x2_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W, OH, OW);
    return frcnn_launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Fully connected head in bf16: y(M,N) = act(x(M,K) @ W(N,K)^T + b), operands bf16 (nearest even), fp32 accumulation.
// Replaces L.Linear + F.relu (/root/reference/models/faster_rcnn.py:33-36,127-134) on the config-3 path: fc6's 411 MB of
// fp32 weights become 205 MB and the 72 GFLOP of the head run at the bf16 MFMA rate.
// Both operands are K-contiguous, so a lane's eight consecutive k-values are one 16-byte read: A = x rows, B = W rows,
// D[m][n] with n on the lanes.  Workgroup = 4 waves along N: tile (32*AM) x 128, K panels of 64 through LDS (pitch 144 B:
// conflict-free ds_read_b128), register-staged double buffer, split-K partial slabs + the fp32 reduce/bias/ReLU pass.
namespace {

constexpr int kLBK = 64;            // k per panel
constexpr int kLPitch = 144;        // bytes per LDS row: 128 B of bf16 + 16 B pad

template <int AM>
__global__ void __launch_bounds__(256)
linear_mfma_bf16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w, float *__restrict__ part, int M, int N, int K, int k_per_split) {
    constexpr int BM = 32 * AM, BN = 128;
    constexpr int XV = BM * (kLBK / 8), WV = BN * (kLBK / 8);          // 16-byte vectors per panel
    constexpr int XIT = (XV + 255) / 256, WIT = (WV + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char xs[2][BM * kLPitch];
    __shared__ __attribute__((aligned(16))) unsigned char ws[2][BN * kLPitch];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_per_split, k_end = min(K, k_begin + k_per_split);
    const int nchunks = (k_end - k_begin + kLBK - 1) / kLBK;
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)M * K * 2));
    const frcnn_buf_t wbuf = frcnn_make_buf(w, (uint32_t)((size_t)N * K * 2));
    uint32_t xoff[XIT], woff[WIT];
#pragma unroll
    for (int q = 0; q < XIT; ++q) {
        const int v = tid + q * 256, row = v / (kLBK / 8), k8 = v % (kLBK / 8);
        xoff[q] = (v < XV && m0 + row < M) ? (uint32_t)(((size_t)(m0 + row) * K + k_begin + k8 * 8) * 2) : kBufOob;
    }
#pragma unroll
    for (int q = 0; q < WIT; ++q) {
        const int v = tid + q * 256, row = v / (kLBK / 8), k8 = v % (kLBK / 8);
        woff[q] = (v < WV && n0 + row < N) ? (uint32_t)(((size_t)(n0 + row) * K + k_begin + k8 * 8) * 2) : kBufOob;
    }
    float4 xreg[XIT], wreg[WIT];
    auto fetch = [&](int chunk) {
        // K % 8 == 0 and the split boundaries are multiples of kLBK, so a 16-byte vector is entirely inside or outside [k_begin, k_end)
        const uint32_t cb = (uint32_t)chunk * (kLBK * 2);
        const int kb = k_begin + chunk * kLBK;
#pragma unroll
        for (int q = 0; q < XIT; ++q) {
            const int k8 = (tid + q * 256) % (kLBK / 8);
            xreg[q] = frcnn_buf_load_f32x4(xbuf, (kb + k8 * 8 < k_end) ? xoff[q] + cb : kBufOob);
        }
#pragma unroll
        for (int q = 0; q < WIT; ++q) {
            const int k8 = (tid + q * 256) % (kLBK / 8);
            wreg[q] = frcnn_buf_load_f32x4(wbuf, (kb + k8 * 8 < k_end) ? woff[q] + cb : kBufOob);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < XIT; ++q) {
            const int v = tid + q * 256;
            if (v < XV) *reinterpret_cast<float4 *>(&xs[buf][(v / (kLBK / 8)) * kLPitch + (v % (kLBK / 8)) * 16]) = xreg[q];
        }
#pragma unroll
        for (int q = 0; q < WIT; ++q) {
            const int v = tid + q * 256;
            if (v < WV) *reinterpret_cast<float4 *>(&ws[buf][(v / (kLBK / 8)) * kLPitch + (v % (kLBK / 8)) * 16]) = wreg[q];
        }
    };
    frcnn_f32x16 acc[AM];
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    if (nchunks > 0) { fetch(0); stage(0); }
    __syncthreads();
    const int l31 = lane & 31, khalf = lane >> 5;
    int cur = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) fetch(chunk + 1);
        const unsigned char *wl = &ws[cur][(wave * 32 + l31) * kLPitch + khalf * 16];
        const unsigned char *xl = &xs[cur][l31 * kLPitch + khalf * 16];
#pragma unroll
        for (int ks = 0; ks < kLBK / 16; ++ks) {
            const uint4 b = *reinterpret_cast<const uint4 *>(wl + ks * 32);
#pragma unroll
            for (int i = 0; i < AM; ++i) {
                const uint4 a = *reinterpret_cast<const uint4 *>(xl + i * 32 * kLPitch + ks * 32);
                acc[i] = frcnn_mfma_32x32x16_bf16(a, b, acc[i]);
            }
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
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
linear_reduce_bf16_kernel(const float *__restrict__ part, const float *__restrict__ bias, void *__restrict__ y, int M, int N, int splits, int relu,
                          int out_bf16) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.0f;
        for (int s = 0; s < splits; ++s) v += part[(size_t)s * total + i];
        v += bias[i % N];
        if (relu) v = fmaxf(v, 0.0f);
        if (out_bf16) reinterpret_cast<uint16_t *>(y)[i] = f32_to_bf16(v);
        else reinterpret_cast<float *>(y)[i] = v;
    }
}

__global__ void __launch_bounds__(256)
f32_to_bf16_kernel(const float *__restrict__ x, size_t n, uint16_t *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = f32_to_bf16(x[i]);
}

struct LinPlan { int am, mblocks, nblocks, splits, k_per_split; };
static LinPlan plan_linear_bf16(int M, int N, int K) {
    LinPlan p;
    p.am = (M > 96) ? 5 : (M > 32 ? 3 : 1);
    p.mblocks = frcnn_cdiv(M, 32 * p.am);
    p.nblocks = frcnn_cdiv(N, 128);
    const int tiles = p.mblocks * p.nblocks, kchunks = frcnn_cdiv(K, kLBK);
    int splits = frcnn_cdiv(512, tiles);
    if (splits > kchunks / 4) splits = kchunks / 4;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    p.k_per_split = frcnn_cdiv(kchunks, splits) * kLBK;
    p.splits = frcnn_cdiv(K, p.k_per_split);
    return p;
}

}  // namespace

extern "C" {

int frcnn_f32_to_bf16(const float *x, size_t n, uint16_t *y, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!x || !y) return FRCNN_ERR_INVALID;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, y);
    return frcnn_launch_status();
}

size_t frcnn_linear_bf16_workspace_bytes(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    const LinPlan p = plan_linear_bf16(M, N, K);
    return frcnn_align256((size_t)p.splits * M * N * sizeof(float));
}

int frcnn_linear_bf16(const uint16_t *x, const uint16_t *w, const float *bias, void *y, int M, int N, int K, int relu, int out_bf16, void *workspace,
                      size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w || !bias || !y || M < 1 || N < 1 || K < 1 || (K % 8) != 0) return FRCNN_ERR_INVALID;
    if ((size_t)M * K * 2 >= (1ull << 31) || (size_t)N * K * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;
    const LinPlan p = plan_linear_bf16(M, N, K);
    if (!workspace || workspace_bytes < (size_t)p.splits * M * N * sizeof(float)) return FRCNN_ERR_INVALID;
    float *part = (float *)workspace;
    const dim3 grid(p.nblocks, p.mblocks, p.splits);
    if (p.am == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_bf16_kernel<5>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (p.am == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_bf16_kernel<3>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_bf16_kernel<1>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    const size_t total = (size_t)M * N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(linear_reduce_bf16_kernel, dim3(blocks), dim3(256), 0, stream, part, bias, y, M, N, p.splits, relu, out_bf16);
    return frcnn_launch_status();
}

}  // extern "C"
