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
#include <string.h>
#include <frcnn_buffer.h>   // angle brackets: shadowed by the test emulator
#include <frcnn_intrin.h>
#include <frcnn_sync.h>
#include "frcnn_reduce.h"

namespace {

constexpr int kCK = 16;                 // channels per K-chunk = the MFMA's k extent
constexpr int kPitchB = 48;             // LDS bytes per (pixel | weight row): 32 B of data + 16 B pad

__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(frcnn_pack_bf16x2(f, 0.0f) & 0xffffu); }   // nearest even

// maximum of four packed bf16 pairs, lane-wise per 16-bit half (a maximum of bf16 values is a bf16 value: exact)
__device__ __forceinline__ uint32_t bf16x2_max4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    uint32_t r = 0;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int sh = 16 * hh;
        const float fa = frcnn_h16_to_f32((uint16_t)(a >> sh)), fb = frcnn_h16_to_f32((uint16_t)(b >> sh));
        const float fc = frcnn_h16_to_f32((uint16_t)(c >> sh)), fd = frcnn_h16_to_f32((uint16_t)(d >> sh));
        r |= frcnn_f32_to_h16_exact(fmaxf(fmaxf(fa, fb), fmaxf(fc, fd))) << sh;
    }
    return r;
}
__device__ __forceinline__ uint4 bf16x8_max4(uint4 a, uint4 b, uint4 c, uint4 d) {
    return make_uint4(bf16x2_max4(a.x, b.x, c.x, d.x), bf16x2_max4(a.y, b.y, c.y, d.y), bf16x2_max4(a.z, b.z, c.z, d.z), bf16x2_max4(a.w, b.w, c.w, d.w));
}

// Epilogue shared by the bf16 conv kernels: bias, ReLU, and one of three output forms.  `ot` is LDS scratch of at least
// BROWS * 32 * (BCO * 2 + 16) bytes that no wave reads any more (the caller has passed a barrier after its last fragment read).
// COB = cout blocks (of 32) per wave: 1 -> wave w owns couts (w & 1) * 32 .., rows (w >> 1) * RW ..; 2 -> wave w owns all 64
// couts of rows w * RW .. (acc is indexed [cob * RW + j])
template <int BROWS, int NT, int RW, int COB = 1>       // tile rows, threads, rows per wave
__device__ __forceinline__ void conv_bf16_epilogue(frcnn_f32x16 (&acc)[RW * COB], unsigned char *ot, const float *__restrict__ bias, void *__restrict__ y,
                                                   int Cout, int CoutP, int H, int W, int relu, int out_mode, int x0, int y0, int co0) {
    constexpr int BCO = 64;
    constexpr int OP = BCO * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco0 = COB == 2 ? 0 : (wave & 1), wrow = COB == 2 ? wave : (wave >> 1);
    const int l31 = lane & 31, khalf = lane >> 5;
    // epilogue: register r of lane l = cout (r&3) + 8*(r>>2) + 4*khalf of pixel l31
    const int px = x0 + l31;
    // this lane's bias values, fetched as ONE batch before anything is stored (out-of-range couts read 0 through the buffer range
    // check): as predicated loads inside the loops below they cost a dependent memory round trip per group of four
    const frcnn_buf_t bbuf = frcnn_make_buf(bias, (uint32_t)Cout * 4u);
    float bv[COB][16];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                bv[cb][4 * g + t] = frcnn_buf_load_f32(bbuf, (uint32_t)(co0 + (wco0 + cb) * 32 + 8 * g + 4 * khalf + t) * 4u);
    if (out_mode == 0 || out_mode == 2) {
        // bf16 channel-blocked output: transpose through LDS so that each 16-cout block of a tile row leaves as one contiguous
        // run of 32 px x 32 B (16-byte stores, consecutive lanes consecutive addresses)
#pragma unroll
        for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = (wco0 + cb) * 32 + 8 * g + 4 * khalf;     // first of four consecutive couts (within the tile)
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v[t] = acc[cb * RW + j][4 * g + t] + bv[cb][4 * g + t];
                    if (relu) v[t] = fmaxf(v[t], 0.0f);
                }
                uint2 pk;
                pk.x = frcnn_pack_bf16x2(v[0], v[1]);
                pk.y = frcnn_pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(ot + ((wrow * RW + j) * 32 + l31) * OP + col * 2) = pk;
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
                const uint4 q0 = *reinterpret_cast<const uint4 *>(t0);
                const uint4 q1 = hasx ? *reinterpret_cast<const uint4 *>(t0 + OP) : q0;
                const uint4 q2 = hasy ? *reinterpret_cast<const uint4 *>(t0 + 32 * OP) : q0;
                const uint4 q3 = (hasx && hasy) ? *reinterpret_cast<const uint4 *>(t0 + 33 * OP) : q0;
                *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(y) + (((size_t)(co >> 4) * OH + (py >> 1)) * OW + (qx >> 1)) * 16 + half * 8) =
                    bf16x8_max4(q0, q1, q2, q3);
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
        // fp32 NCHW (Cout, H, W): straight-line buffer stores, lanes / couts outside the map store nothing (offset out of range)
        const frcnn_buf_t ybuf = frcnn_make_buf(y, (uint32_t)((size_t)Cout * H * W * 4));
#pragma unroll
        for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int py = y0 + wrow * RW + j;
            const bool inside = px < W && py < H;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + (wco0 + cb) * 32 + 8 * g + 4 * khalf;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v = acc[cb * RW + j][4 * g + t] + bv[cb][4 * g + t];
                    if (relu) v = fmaxf(v, 0.0f);
                    frcnn_buf_store_f32(ybuf, (inside && co + t < Cout) ? (uint32_t)(((co + t) * H + py) * W + px) * 4u : kBufOob, v);
                }
            }
        }
    }
}

// ABL = timing ablations (WRONG results; scripts/conv_bf16_sweep.py only): 1 no global loads, 2 no LDS stores, 4 no fragment reads
template <int KS, int RP, int ABL = 0>   // RP = row pairs per workgroup: 2 -> 4 waves, 64co x 4 rows; 4 -> 8 waves, 64co x 8 rows
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
    if constexpr ((ABL & 1) != 0) {
#pragma unroll
        for (int q = 0; q < HIT; ++q) hregA[q] = hregB[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < WIT; ++q) wregA[q] = wregB[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto fetch = [&](int chunk, float4 (&hreg)[HIT], float4 (&wreg)[WIT]) {
        if constexpr ((ABL & 1) != 0) return;
        const uint32_t xb = (uint32_t)chunk * x_chunk_bytes, wb = (uint32_t)chunk * w_chunk_bytes;
#pragma unroll
        for (int q = 0; q < HIT; ++q) hreg[q] = frcnn_buf_load_f32x4(xbuf, hoff[q] + xb);
#pragma unroll
        for (int q = 0; q < WIT; ++q) wreg[q] = frcnn_buf_load_f32x4(wbuf, woff[q] + wb);
    };
    auto stage = [&](int buf, const float4 (&hreg)[HIT], const float4 (&wreg)[WIT]) {
        if constexpr ((ABL & 2) != 0) {
#pragma unroll
            for (int q = 0; q < HIT; ++q) asm volatile("" ::"v"(hreg[q].x), "v"(hreg[q].y), "v"(hreg[q].z), "v"(hreg[q].w));
#pragma unroll
            for (int q = 0; q < WIT; ++q) asm volatile("" ::"v"(wreg[q].x), "v"(wreg[q].y), "v"(wreg[q].z), "v"(wreg[q].w));
            return;
        }
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
        if constexpr ((ABL & 4) != 0) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) a[tap] = make_uint4(lane + tap, buf, lane, tap);
#pragma unroll
            for (int r = 0; r < KS + 1; ++r)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) b[r][kx] = make_uint4(lane + r, kx, buf, lane);
        } else {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) a[tap] = *reinterpret_cast<const uint4 *>(wl + tap * BCO * kPitchB);
#pragma unroll
        for (int r = 0; r < KS + 1; ++r)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) b[r][kx] = *reinterpret_cast<const uint4 *>(il + (r * HPX + kx) * kPitchB);
        }
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

    static_assert(BROWS * 32 * OP <= (int)sizeof(w_lds), "epilogue tile must fit in the weight buffers");
    // the K loop is over and all waves passed its last barrier: the weight buffers are free to carry the output tile
    conv_bf16_epilogue<2 * RP, 128 * RP, 2>(acc, &w_lds[0][0], bias, y, Cout, CoutP, H, W, relu, out_mode, x0, y0, co0);
}

// The same 3x3 convolution with LDS-DMA staging (buffer_load_dwordx4 ... lds): a chunk's halo and weight panel go from L2 straight into
// LDS, 1 KB per wave-instruction -- no staging VGPRs, no ds_write_b128 (13 LDS cycles per KB through the VGPR->LDS path, which with
// the fragment reads made the register-staged kernel LDS-bound), no wait on a load result inside the K loop.  The destination of
// a piece is lane-linear, so the LDS image has pitch 32 B (no pad) and is kept conflict-free by an XOR swizzle applied on both
// sides (rule 21): 16-byte slot of (row P, half h) = 2P + (h ^ ((P >> 3) & 1)) -- every 16-lane group of a ds_read_b128 whose
// lanes read consecutive rows at any base offset then covers 16 distinct bank slots.  NS ring stages: chunk c+NS-1 is in flight
// while chunk c feeds the MFMAs; per chunk one counted s_waitcnt vmcnt(N) + one fence-less barrier.
// Tile shapes (RPW selects one; all 64 couts x 32 px wide, 4 waves):
//   RPW 1, 2  a wave owns 32 couts x 2*RPW rows (two cout blocks x two row groups across the waves): 4 / 8 tile rows
//   RPW 3, 4  a wave owns ALL 64 couts x 2 / 4 rows (four row groups across the waves): 8 / 16 tile rows.  Every B fragment feeds two
//             MFMAs and every A fragment RW of them: 0.83 / 0.5 LDS fragment reads per MFMA instead of 1.17 / 0.75, and 1.7x / 2.6x the
//             MFMA work per staged byte -- the fragment reads (84 KB per chunk for 576 MFMA cycles in the RPW 1 form) were co-limiting
//             the matrix pipe.
template <int NS, int WPS, int RPW = 1, int ABL = 0>   // ABL: 1 no DMA, 4 no compute
__global__ void __launch_bounds__(256, WPS)
conv_dma_bf16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wp, const float *__restrict__ bias, void *__restrict__ y,
                     int CinP, int Cout, int CoutP, int H, int W, int relu, int out_mode, int xtiles, int ytiles, int nsplit,
                     float *__restrict__ partial_ws, int *__restrict__ tile_counters, int xcd_cotiles) {
    constexpr int KS = 3, TAPS = 9, PAD = 1;
    constexpr int COB = RPW >= 3 ? 2 : 1;                       // cout blocks per wave
    constexpr int RW = RPW == 1 ? 2 : (RPW == 2 ? 4 : (RPW == 3 ? 2 : 4));
    constexpr int ROWG = COB == 2 ? 4 : 2;                      // row groups across the four waves
    constexpr int NACC = RW * COB;
    constexpr int BROWS = ROWG * RW, BCO = 64;
    constexpr int HR = BROWS + KS - 1, HPX = 32 + KS - 1;
    constexpr int IN_ROWS = HR * HPX;                         // 204 halo pixels, 32 B each
    constexpr int IN_PIECES = (IN_ROWS * 2 + 63) / 64;        // 1 KB pieces (64 lanes x 16 B): 7, the last one partly out of range
    constexpr int W_PIECES = TAPS * BCO * 2 / 64;             // 18
    constexpr int PIECES = IN_PIECES + W_PIECES;
    constexpr int IN_BYTES = IN_PIECES * 1024, STAGE_BYTES = PIECES * 1024;
    constexpr int PPW = (PIECES + 3) / 4;                     // pieces per wave (wave w moves pieces w, w+4, ...)
    constexpr int OP = BCO * 2 + 16;
    constexpr int RING_BYTES = NS * STAGE_BYTES > BROWS * 32 * OP ? NS * STAGE_BYTES : BROWS * 32 * OP;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[RING_BYTES];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = COB == 2 ? 0 : (wave & 1), wrow = COB == 2 ? wave : (wave >> 1);
    const int l31 = lane & 31, khalf = lane >> 5;
    // split-K for launches with fewer tiles than the chip has room for (the 38x63 maps: 160 tiles on 256 CUs): `nsplit`
    // consecutive workgroups share a tile, each takes a contiguous range of the K-chunks, the last to finish sums the
    // pieces in split order (deterministic) and runs the epilogue -- the stream-K fix-up of conv.hip, with fixed ranges
    int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    if (xcd_cotiles > 0) {
        // XCD-aware order (launches with 1, 2, 4 or 8 cout tiles): workgroups are dealt to the 8 XCDs round-robin, so XCD x works on
        // cout tile x % cotiles only -- its L2 holds ONE 64-cout weight slab instead of all of them (conv_f32s.hip measured 5-10 %)
        const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, G = 8 / xcd_cotiles;
        const int u = j * G + xcd / xcd_cotiles;                   // (pixel tile, K split) unit of this XCD's cout tile
        const int pt = u / nsplit;
        if (pt >= xtiles * ytiles) return;                       // (the grid is padded to a multiple of 8)
        split = u - pt * nsplit;
        tile = (xcd % xcd_cotiles) * xtiles * ytiles + pt;
    }
    const int tx = tile % xtiles, ty = (tile / xtiles) % ytiles, cot = tile / (xtiles * ytiles);
    const int x0 = tx * 32, y0 = ty * BROWS, co0 = cot * BCO;
    const int all_chunks = CinP / kCK;
    const int c_first = split * all_chunks / nsplit, nchunks = (split + 1) * all_chunks / nsplit - c_first;
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)H * W * CinP * 2));
    const frcnn_buf_t wbuf = frcnn_make_buf(wp, (uint32_t)((size_t)TAPS * CoutP * CinP * 2));
    const uint32_t x_chunk_bytes = (uint32_t)(H * W) * 32u, w_chunk_bytes = (uint32_t)(TAPS * CoutP) * 32u;

    // source offset (chunk 0) of the 16 bytes this lane contributes to each of its wave's pieces: slot s of a region holds
    // (row P = s >> 1, half (s & 1) ^ ((P >> 3) & 1))
    uint32_t poff[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pid = wave + 4 * q;
        if (pid < IN_PIECES) {
            const int sl = pid * 64 + lane, P = sl >> 1, half = (sl & 1) ^ ((P >> 3) & 1);
            const int hr = P / HPX, hx = P - hr * HPX;
            const int gy = y0 - PAD + hr, gx = x0 - PAD + hx;
            const bool inside = P < IN_ROWS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            poff[q] = inside ? (uint32_t)((gy * W + gx) * 32 + half * 16) : kBufOob;
        } else {
            const int sl = (pid - IN_PIECES) * 64 + lane, P = sl >> 1, half = (sl & 1) ^ ((P >> 3) & 1);
            const int tap = P / BCO, col = P - tap * BCO;
            poff[q] = (pid < PIECES && co0 + col < CoutP) ? (uint32_t)((tap * CoutP + co0 + col) * 32 + half * 16) : kBufOob;
        }
    }
    auto issue = [&](int chunk, int stage) {
        if constexpr ((ABL & 1) != 0) return;
        unsigned char *dst = ring + stage * STAGE_BYTES + wave * 1024;
        const uint32_t xs = (uint32_t)(c_first + chunk) * x_chunk_bytes, ws = (uint32_t)(c_first + chunk) * w_chunk_bytes;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            // pieces 4q .. 4q+3 belong to waves 0..3: which tensor they come from is a compile-time fact except where the halo
            // region ends inside the group (then: a wave-uniform select), and only the last group can run past the end
            if (4 * q + 3 < IN_PIECES) frcnn_buf_load_lds_b128(xbuf, dst + q * 4096, poff[q], xs);
            else if (4 * q >= IN_PIECES) {
                if (4 * q + 3 < PIECES || wave + 4 * q < PIECES) frcnn_buf_load_lds_b128(wbuf, dst + q * 4096, poff[q], ws);
            } else {
                const bool in = wave + 4 * q < IN_PIECES;
                frcnn_buf_load_lds_b128(in ? xbuf : wbuf, dst + q * 4096, poff[q], in ? xs : ws);
            }
        }
    };
    // this wave's loads per chunk (PPW, or PPW - 1 for the waves past the last piece): "at most k chunks of my loads still in flight"
    // is one counted s_waitcnt with k x that number (vmcnt holds 63: k <= 4 here)
    const bool seven = wave < PIECES - 4 * (PPW - 1);
    auto wait_allow = [&](int k) {
        static_assert(NS <= 6 && 4 * PPW <= 63, "vmcnt immediates below");
        switch (k) {
        case 0: frcnn_wait_vmcnt<0>(); break;
        case 1: if (seven) frcnn_wait_vmcnt<PPW>(); else frcnn_wait_vmcnt<PPW - 1>(); break;
        case 2: if (seven) frcnn_wait_vmcnt<2 * PPW>(); else frcnn_wait_vmcnt<2 * (PPW - 1)>(); break;
        case 3: if (seven) frcnn_wait_vmcnt<3 * PPW>(); else frcnn_wait_vmcnt<3 * (PPW - 1)>(); break;
        default: if (seven) frcnn_wait_vmcnt<4 * PPW>(); else frcnn_wait_vmcnt<4 * (PPW - 1)>(); break;
        }
    };

    // fragment byte offsets inside a stage (swizzled): A = weight row tap*64 + wco*32 + l31, B = halo pixel (RW*wrow + r)*34 + l31 + kx
    const uint32_t a_off = (uint32_t)(IN_BYTES + (wco * 32 + l31) * 32 + ((khalf ^ ((l31 >> 3) & 1)) << 4));
    uint32_t b_off[RW + KS - 1][KS];
#pragma unroll
    for (int r = 0; r < RW + KS - 1; ++r)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int P = (wrow * RW + r) * HPX + l31 + kx;
            b_off[r][kx] = (uint32_t)(P * 32 + ((khalf ^ ((P >> 3) & 1)) << 4));
        }

    frcnn_f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    auto compute = [&](int stage) {
        if constexpr ((ABL & 4) != 0) return;
        const unsigned char *st = ring + stage * STAGE_BYTES;
        if constexpr (COB == 1) {
            uint4 a[TAPS], b[RW + KS - 1][KS];
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) a[tap] = *reinterpret_cast<const uint4 *>(st + a_off + tap * BCO * 32);
#pragma unroll
            for (int r = 0; r < RW + KS - 1; ++r)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) b[r][kx] = *reinterpret_cast<const uint4 *>(st + b_off[r][kx]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int ky = tap / KS, kx = tap % KS;
#pragma unroll
                for (int j = 0; j < RW; ++j) acc[j] = frcnn_mfma_32x32x16_bf16(a[tap], b[ky + j][kx], acc[j]);
            }
        } else {
            // both cout blocks in one wave: the halo fragments are read once per chunk and stay in registers, the weight fragments
            // stream through one tap row at a time (same (chunk, tap) accumulation order per output as every other variant:
            // bit-identical results); the second cout block's weight rows sit 32 rows (1 KB, same swizzle phase) further on
            uint4 b[RW + KS - 1][KS];
#pragma unroll
            for (int r = 0; r < RW + KS - 1; ++r)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) b[r][kx] = *reinterpret_cast<const uint4 *>(st + b_off[r][kx]);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                uint4 a[2][KS];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) a[cb][kx] = *reinterpret_cast<const uint4 *>(st + a_off + ((ky * KS + kx) * BCO + cb * 32) * 32);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int j = 0; j < RW; ++j) acc[cb * RW + j] = frcnn_mfma_32x32x16_bf16(a[cb][kx], b[ky + j][kx], acc[cb * RW + j]);
            }
        }
    };

    // WPS == 1 (one workgroup per CU: launches with fewer tiles than CUs, the 38 x 63 maps): nobody else is resident to cover this
    // workgroup's latencies, so besides the deep DMA ring the FRAGMENTS are double-buffered in registers (512 of them per lane at one
    // wave per SIMD): chunk c+1's 21 ds_read_b128 are in flight while chunk c's 18 MFMAs run.  Read-then-multiply in one chunk cost
    // ~340 LDS clocks + 576 MFMA clocks per chunk in sequence; deeper rings alone changed nothing (25.4 vs 23.4 us on conv5_1).
    constexpr bool FDB = WPS == 1 && RPW == 1 && NS >= 3 && ABL == 0;
    if constexpr (FDB) {
        auto loadf = [&](uint4 (&a)[TAPS], uint4 (&b)[RW + KS - 1][KS], int stage) {
            const unsigned char *st = ring + stage * STAGE_BYTES;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) a[tap] = *reinterpret_cast<const uint4 *>(st + a_off + tap * BCO * 32);
#pragma unroll
            for (int r = 0; r < RW + KS - 1; ++r)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) b[r][kx] = *reinterpret_cast<const uint4 *>(st + b_off[r][kx]);
        };
        auto mfmas = [&](const uint4 (&a)[TAPS], const uint4 (&b)[RW + KS - 1][KS]) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int ky = tap / KS, kx = tap % KS;
#pragma unroll
                for (int j = 0; j < RW; ++j) acc[j] = frcnn_mfma_32x32x16_bf16(a[tap], b[ky + j][kx], acc[j]);
            }
        };
        // chunk s lives in stage s % NS.  Invariant after step(c): chunk c's fragments are in registers, chunks c+1 .. c+NS-1 landed or
        // in flight.  advance(c, next): wait for chunk c+1, barrier (every wave's reads of stage(c) are complete: its fragments are in
        // registers), refill stage(c) with chunk c+NS, start reading chunk c+1's fragments into `next`.
        uint4 fa0[TAPS], fb0[RW + KS - 1][KS], fa1[TAPS], fb1[RW + KS - 1][KS];
#pragma unroll
        for (int c = 0; c < NS - 1; ++c)
            if (c < nchunks) issue(c, c);
        wait_allow(min(NS - 2, nchunks - 1));
        frcnn_barrier_nofence();
        if (NS - 1 < nchunks) issue(NS - 1, NS - 1);
        loadf(fa0, fb0, 0);
        auto advance = [&](int c, uint4 (&na)[TAPS], uint4 (&nb)[RW + KS - 1][KS]) {
            wait_allow(min(c + NS - 1, nchunks - 1) - (c + 1));
            frcnn_barrier_nofence();
            if (c + NS < nchunks) issue(c + NS, c % NS);
            loadf(na, nb, (c + 1) % NS);
        };
        for (int c = 0; c < nchunks; c += 2) {
            if (c + 1 < nchunks) advance(c, fa1, fb1);
            mfmas(fa0, fb0);
            if (c + 1 >= nchunks) break;
            if (c + 2 < nchunks) advance(c + 1, fa0, fb0);
            mfmas(fa1, fb1);
        }
    } else
    if constexpr (NS == 1) {
        // single stage: no overlap inside the workgroup -- the other (up to five) workgroups of the CU run their MFMAs while this
        // one waits for its chunk; the small LDS footprint is what buys that occupancy
        for (int c = 0; c < nchunks; ++c) {
            issue(c, 0);
            frcnn_wait_vmcnt<0>();
            frcnn_barrier_nofence();
            compute(0);
            if (c + 1 < nchunks) frcnn_barrier_nofence();       // everybody is done reading before the stage is refilled
        }
    } else {
    // prologue: NS-1 chunks in flight, chunk 0 landed.  Deep rings (NS 4 .. 6, one workgroup per CU) are for launches with fewer tiles
    // than CUs (the 38 x 63 maps: 160 tiles): nobody else is resident to cover a chunk's DMA latency (~1 us for 25 KB against 0.24 us of
    // MFMAs), so the workgroup itself keeps NS-1 chunks in flight
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nchunks) issue(c, c);
    wait_allow(min(NS - 2, nchunks - 1));
    frcnn_barrier_nofence();
    int s_cur = 0, s_new = NS - 1;                               // stage of chunk c, stage chunk c+NS-1 goes to
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + NS - 1 < nchunks;
        if (more) issue(c + NS - 1, s_new);
        compute(s_cur);
        if (c + 1 < nchunks) {
            // chunk c+1 must have landed for everybody (the chunks issued after it may still be in flight), and everybody must be done
            // reading stage s_cur before it is refilled
            wait_allow(min(c + NS - 1, nchunks - 1) - (c + 1));
            frcnn_barrier_nofence();
        }
        s_cur = s_cur + 1 == NS ? 0 : s_cur + 1;
        s_new = s_new + 1 == NS ? 0 : s_new + 1;
    }
    }
    if (nsplit > 1) {
        // publish this split's accumulators (fragment-linear float4s, write-through: no release fence needed), take a ticket
        constexpr int NV = NACC * 4;                              // float4s per thread
        const size_t slot_floats = (size_t)256 * NACC * 16;
        const frcnn_buf_t pbuf = frcnn_make_buf(partial_ws + ((size_t)tile * nsplit + split) * slot_floats, (uint32_t)(slot_floats * sizeof(float)));
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                frcnn_buf_store_f32x4_wt(pbuf, (uint32_t)(((j * 4 + r4) * 256 + tid) * 16),
                                         make_float4(acc[j][4 * r4], acc[j][4 * r4 + 1], acc[j][4 * r4 + 2], acc[j][4 * r4 + 3]));
        frcnn_drain_vmem();
        __syncthreads();
        if (tid == 0) s_ticket = frcnn_ticket(&tile_counters[tile]);
        __syncthreads();
        if (s_ticket != nsplit - 1) return;                       // workgroup-uniform
        if (tid == 0) {
            frcnn_acquire_agent();
            frcnn_counter_reset(&tile_counters[tile]);            // leave the counter page zeroed for the next launch
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        for (int q = 0; q < nsplit; ++q) {
            const float4 *piece = reinterpret_cast<const float4 *>(partial_ws + ((size_t)tile * nsplit + q) * slot_floats);
            float4 v[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) v[e] = piece[(size_t)e * 256 + tid];
#pragma unroll
            for (int e = 0; e < NV; ++e) frcnn_pin(v[e]);               // one batch of loads per piece (else: load - wait - add per vector)
#pragma unroll
            for (int j = 0; j < NACC; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 t = v[j * 4 + r4];
                    acc[j][4 * r4] += t.x; acc[j][4 * r4 + 1] += t.y; acc[j][4 * r4 + 2] += t.z; acc[j][4 * r4 + 3] += t.w;
                }
        }
    }
    __syncthreads();                                            // the ring becomes the epilogue's output tile
    conv_bf16_epilogue<BROWS, 256, RW, COB>(acc, ring, bias, y, Cout, CoutP, H, W, relu, out_mode, x0, y0, co0);
}

// FRCNN_TUNING_FORMS (round 5): the forms that were measured and NOT adopted -- the resident producer / consumer kernels of
// conv_bf16_res.h, strip forms A / B / E / 907 / 908, the ring depths and tile shapes of conv_dma_bf16_kernel that no default rule picks, the 8-wave
// register-staged kernel -- are compiled only into the research builds (scripts/micro/build_micro.sh, the test emulator), not into libfrcnn_hip.so.
// In the product library a request for one of them is FRCNN_ERR_INVALID.
#ifdef FRCNN_TUNING_FORMS
#include "conv_bf16_res.h"        // conv_res_bf16_kernel: resident weight slab, producer / consumer waves (FRCNN_BF16_DMA=921 / 922; round 4)
#endif
#include "conv_bf16_strip.h"      // conv_strip_bf16_kernel: one wave per SIMD, software-pipelined ring (forms D and C are default picks; FRCNN_BF16_DMA=900..909)

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

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return frcnn_h16_to_f32(h); }

// F.MaxPooling2D(2, 2), cover_all, channel-blocked bf16 [C/16][H][W][16]: one thread = one output pixel x 8 channels (16 bytes)
__global__ void __launch_bounds__(256)
maxpool2x2_bf16_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int C, int H, int W, int OH, int OW) {
    const size_t total = (size_t)(C / 16) * OH * OW * 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int half = (int)(i & 1), ow = (int)((i >> 1) % OW), oh = (int)(((i >> 1) / OW) % OH), cb = (int)((i >> 1) / ((size_t)OW * OH));
        const bool hasx = 2 * ow + 1 < W, hasy = 2 * oh + 1 < H;
        const uint16_t *p = x + ((((size_t)cb * H + 2 * oh) * W + 2 * ow) * 16 + half * 8);
        const uint4 q0 = *reinterpret_cast<const uint4 *>(p);
        const uint4 q1 = hasx ? *reinterpret_cast<const uint4 *>(p + 16) : q0;
        const uint4 q2 = hasy ? *reinterpret_cast<const uint4 *>(p + (size_t)W * 16) : q0;
        const uint4 q3 = (hasx && hasy) ? *reinterpret_cast<const uint4 *>(p + (size_t)W * 16 + 16) : q0;
        *reinterpret_cast<uint4 *>(y + ((((size_t)cb * OH + oh) * OW + ow) * 16 + half * 8)) = bf16x8_max4(q0, q1, q2, q3);
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

// The RPN heads of the bf16 chain as ONE launch (the fp32 form: rpn_heads_fused_kernel, conv.hip): the stacked 1x1 convolution of the
// channel-blocked bf16 map + the 2A-way softmax.  On the 38x63 map the 1x1 case of conv_mfma_bf16_kernel has 20 tiles of 32 sequential
// K-chunks (22 us) and the softmax is a second launch (5 us).  Here a 512-thread workgroup owns 32 pixels and all (<= 64) stacked
// output channels, its 8 waves split the K-chunks and read their MFMA fragments straight from global memory -- both layouts are
// fragment-ready: a chunk's 32 px of h ([C/16][HW][16]) and 32 weight rows ([C/16][CoutP][16]) are contiguous 1 KB runs, 16 bytes per
// lane -- no LDS staging, no barrier inside the K range; the partial tiles meet in LDS, are added in wave order with the bias, leave as
// the fp32 (n_out, HW) map, and the first 32 threads run the reference's softmax over the 2A score rows on the tile
// (region_proposal_network.py:119).  Same operands as the two-launch form (bf16 x bf16 products are exact in fp32): the results differ
// by the order of the fp32 additions only.
constexpr int kHeadWavesB = 8, kHeadChunksB = 4;          // K-chunks (16 channels) whose loads a wave issues together
__global__ void __launch_bounds__(64 * kHeadWavesB)
rpn_heads_bf16_fused_kernel(const uint16_t *__restrict__ h, const uint16_t *__restrict__ wp, const float *__restrict__ bias, float *__restrict__ raw,
                            float *__restrict__ prob, int CinP, int CoutP, int HW, int n_out, int n_score) {
    __shared__ float part[kHeadWavesB][64][32];
    __shared__ float outt[64][33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int p0 = blockIdx.x * 32;
    const int nchunks = CinP / kCK;
    const int per_wave = (nchunks + kHeadWavesB * kHeadChunksB - 1) / (kHeadWavesB * kHeadChunksB) * kHeadChunksB;   // whole load batches
    const frcnn_buf_t hbuf = frcnn_make_buf(h, (uint32_t)((size_t)CinP * HW * 2));
    const frcnn_buf_t wbuf = frcnn_make_buf(wp, (uint32_t)((size_t)CinP * CoutP * 2));
    // per-lane byte offsets inside a chunk (out of range: pixels past the map, weight rows past CoutP -> zeros)
    const uint32_t b_off = p0 + l31 < HW ? (uint32_t)((p0 + l31) * 32 + khalf * 16) : kBufOob;
    uint32_t a_off[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) a_off[cb] = cb * 32 + l31 < CoutP ? (uint32_t)((cb * 32 + l31) * 32 + khalf * 16) : kBufOob;
    const uint32_t h_chunk = (uint32_t)HW * 32u, w_chunk = (uint32_t)CoutP * 32u;
    frcnn_f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
    for (int c0 = wave * per_wave; c0 < (wave + 1) * per_wave; c0 += kHeadChunksB) {
        float4 bq[kHeadChunksB], aq[kHeadChunksB][2];
#pragma unroll
        for (int u = 0; u < kHeadChunksB; ++u) {
            // chunks past the end of the tensors are out-of-range offsets: h_chunk * c >= the buffer size for c >= nchunks
            const uint32_t c = (uint32_t)(c0 + u);
            bq[u] = frcnn_buf_load_f32x4(hbuf, c < (uint32_t)nchunks ? b_off + c * h_chunk : kBufOob);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) aq[u][cb] = frcnn_buf_load_f32x4(wbuf, c < (uint32_t)nchunks ? a_off[cb] + c * w_chunk : kBufOob);
        }
        __builtin_amdgcn_sched_barrier(0);                             // all of the batch's loads in flight before the first MFMA waits
#pragma unroll
        for (int u = 0; u < kHeadChunksB; ++u) {
            const uint4 bv = make_uint4(__float_as_uint(bq[u].x), __float_as_uint(bq[u].y), __float_as_uint(bq[u].z), __float_as_uint(bq[u].w));
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const uint4 av = make_uint4(__float_as_uint(aq[u][cb].x), __float_as_uint(aq[u][cb].y), __float_as_uint(aq[u][cb].z), __float_as_uint(aq[u][cb].w));
                acc[cb] = frcnn_mfma_32x32x16_bf16(av, bv, acc[cb]);
            }
        }
    }
    // D register r of lane l = cout (r&3) + 8*(r>>2) + 4*(l>>5), pixel l&31
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf][l31] = acc[cb][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 64 * 32 / (64 * kHeadWavesB); ++q) {
        const int o = tid + q * 64 * kHeadWavesB, co = o >> 5, px = o & 31;
        float v = part[0][co][px];
#pragma unroll
        for (int w = 1; w < kHeadWavesB; ++w) v += part[w][co][px];
        v += co < n_out ? bias[co] : 0.0f;
        outt[co][px] = v;
        if (co < n_out && p0 + px < HW) raw[(size_t)co * HW + p0 + px] = v;
    }
    __syncthreads();
    if (tid < 32 && p0 + tid < HW) {
        float v[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = c < n_score ? outt[c][tid] : 0.0f;
        float m = v[0];
#pragma unroll
        for (int c = 1; c < 32; ++c) if (c < n_score) m = fmaxf(m, v[c]);
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < 32; ++c) if (c < n_score) { v[c] = expf(v[c] - m); sum += v[c]; }
#pragma unroll
        for (int c = 0; c < 32; ++c) if (c < n_score) prob[(size_t)c * HW + p0 + tid] = v[c] / sum;
    }
}

}  // namespace

// The strip forms of conv_bf16_strip.h (FRCNN_BF16_DMA=901 / 902 / 903 / 909 = form A / B / C / D, 900 = the cheapest applicable one of A / B / C by a
// count of MFMA rounds, 907 / 908 two measured shapes that were not adopted).  Returns 1 when the form does not exist or does not apply to the launch.
template <int COB, int RW, int RG, int CW, int KW, int NS, int ABL = 0, int WPE = 1, bool DIRECT = false, int NW = 4>
static void conv_bf16_strip_go(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int CinP, int Cout, int CoutP, int H, int W,
                               int relu, int out_mode, hipStream_t stream) {
    const int xtiles = frcnn_cdiv(W, 32), ytiles = frcnn_cdiv(H, RG * RW), cotiles = frcnn_cdiv(CoutP, 32 * COB * CW);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_strip_bf16_kernel<COB, RW, RG, CW, KW, NS, ABL, WPE, DIRECT, NW>), dim3((unsigned)((long)xtiles * ytiles * cotiles)), dim3(64 * NW), 0, stream,
                       x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles, cotiles);
}
// Which strip form a request resolves to for a launch (0 = the request does not exist / does not apply): `form` 1..10 as requested, 0 = the cheapest
// applicable one of A / B / C by a count of MFMA rounds.  ONE place for the launcher and for frcnn_conv_bf16_plan (the plan query had its own copy).
static int conv_bf16_strip_resolve(int form, int CinP, int CoutP, int H, int W, int out_mode) {
    const int chunks = CinP / kCK;
    // {couts per workgroup, tile rows, K ways, MFMAs per wave and stage}
    static const int kForm[12][4] = {{0, 0, 0, 0}, {64, 20, 1, 90}, {64, 10, 1, 45}, {32, 5, 4, 45}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {64, 10, 2, 90}, {64, 12, 1, 54}, {64, 10, 1, 45}, {64, 10, 1, 45},
                                     {64, 20, 1, 45}};
#ifdef FRCNN_TUNING_FORMS
    auto built = [](int) { return true; };
#else
    auto built = [](int f) { return f == 3 || f == 9 || f == 10; };        // the default picks: form C, form D (LDS epilogue / direct stores)
#endif
    auto applies = [&](int f) { return f >= 0 && f <= 11 && kForm[f][0] != 0 && built(f) && chunks % kForm[f][2] == 0 && !(out_mode == 2 && (kForm[f][1] & 1)); };
    if (form == 0) {
        const long cus = frcnn_cu_count() > 0 ? frcnn_cu_count() : 256;
        long best = -1;
        for (int f = 1; f <= 3; ++f) {
            if (!applies(f)) continue;
            const long wgs = (long)frcnn_cdiv(W, 32) * frcnn_cdiv(H, kForm[f][1]) * frcnn_cdiv(CoutP, kForm[f][0]);
            // MFMAs in sequence on a SIMD: rounds x (stages + two stages' worth of prologue / epilogue) x MFMAs per wave and stage
            const long cost = ((wgs + cus - 1) / cus) * (chunks / kForm[f][2] + 2) * kForm[f][3];
            if (best < 0 || cost < best) { best = cost; form = f; }
        }
    }
    return applies(form) && form != 0 ? form : 0;
}
static int conv_bf16_strip(int form, const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int CinP, int Cout, int CoutP, int H, int W,
                           int relu, int out_mode, hipStream_t stream, int abl = 0) {
    form = conv_bf16_strip_resolve(form, CinP, CoutP, H, W, out_mode);
    if (form == 0) return 1;
#ifdef FRCNN_TIMING_ABLATIONS                                                                       // WRONG results: sweeps only, never shipped
#define FRCNN_STRIP_ABL(A)                                                                                                                  \
    case A:                                                                                                                                 \
        if (form == 1) conv_bf16_strip_go<2, 5, 4, 1, 1, 3, A>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream);      \
        else if (form == 2) conv_bf16_strip_go<1, 5, 2, 2, 1, 4, A>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream); \
        else conv_bf16_strip_go<1, 5, 1, 1, 4, 2, A>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream);                \
        return 0;
    switch (abl) { FRCNN_STRIP_ABL(1) FRCNN_STRIP_ABL(2) FRCNN_STRIP_ABL(3) FRCNN_STRIP_ABL(4) FRCNN_STRIP_ABL(8) FRCNN_STRIP_ABL(11) default: break; }
#undef FRCNN_STRIP_ABL
#endif
    (void)abl;
    switch (form) {
#ifdef FRCNN_TUNING_FORMS
    case 1: conv_bf16_strip_go<2, 5, 4, 1, 1, 3>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream); break;
    case 2: conv_bf16_strip_go<1, 5, 2, 2, 1, 4>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream); break;
#endif
    case 3: conv_bf16_strip_go<1, 5, 1, 1, 4, 2>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream); break;
#ifdef FRCNN_TUNING_FORMS
    // 907: 64 couts x 10 rows with the K loop split two ways over the waves (64 couts per wave: 0.43 fragment reads per MFMA instead of form B's 0.67;
    //      40.5 vs 39.9 us on conv4_2, not adopted)
    case 7: conv_bf16_strip_go<2, 5, 2, 1, 2, 2>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream); break;
    // 908 / 909: TWO workgroups per CU (<= 256 registers, 68 KB of LDS each, two-stage rings): each covers the other's prologue and epilogue, which the
    // one-workgroup forms pay in the open on every tile of a launch of several rounds.  908: 64 couts x 12 rows (three rows per wave; 2-8 % behind 909),
    // 909 = form D: form B's waves -- the default pick of frcnn_conv_bf16_ws for launches with >= 8 K-chunks and >= one tile per CU
    case 8: conv_bf16_strip_go<2, 3, 4, 1, 1, 2, 0, 2>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream); break;
#endif
    case 9: conv_bf16_strip_go<1, 5, 2, 2, 1, 2, 0, 2>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream); break;
    // 910: form D with the bf16 output of a launch without the fused pool stored straight from the accumulators (no LDS transpose, no barrier: conv3_1 / conv3_2 /
    //      conv4_2 25.4 / 41.7 / 41.7 -> 25.1 / 40.8 / 40.9 us, bit-identical; probe 9) -- what the default rule launches; pooled and fp32-NCHW launches take 909's epilogue
    case 10:
        if (out_mode == 0 && (size_t)CoutP * H * W * 2 < (1ull << 31)) conv_bf16_strip_go<1, 5, 2, 2, 1, 2, 0, 2, true>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream);
        else conv_bf16_strip_go<1, 5, 2, 2, 1, 2, 0, 2>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream);
        break;
#ifdef FRCNN_TUNING_FORMS
    // 911 = form E (round 4): form A's tile -- 64 couts x 20 rows x 32 px -- on EIGHT of form D's waves (two per SIMD, one workgroup per CU, three 42 KB stages):
    //       the two waves of a SIMD share a stage's weight panel; 6 LDS-DMA pieces per wave and stage instead of 8.5
    case 11:
        if (out_mode == 0 && (size_t)CoutP * H * W * 2 < (1ull << 31)) conv_bf16_strip_go<1, 5, 4, 2, 1, 3, 0, 1, true, 8>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream);
        else conv_bf16_strip_go<1, 5, 4, 2, 1, 3, 0, 1, false, 8>(x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream);
        break;
#endif
    default: return 1;
    }
    return 0;
}

// The strip form the default rule picks for a 3x3 launch (0: none -- conv_dma_bf16_kernel's picks), environment hooks included; the
// measurements behind it are quoted where frcnn_conv_bf16_ws applies it.
// The resident forms (csrc/conv_bf16_res.h: weight slab in LDS, producer / consumer waves; Cin 64 -> 21 = form R, Cin 128 -> 22 = form R2) are NOT default
// picks: measured on the MI355X (profiles/r04_conv_res_micro.txt, bit-identical to the other kernels) they lose to the strip forms / conv_dma_bf16_kernel
// on all four layers they fit -- conv1_2 52.2 vs 48.5 us, conv2_1 31.1 vs 28.3, conv2_2 49.3 vs 41.7, conv3_1 31.0 vs 23.8 -- with any ring depth and
// either wave priority: with ONE multiplying wave per SIMD nothing covers that wave's own stalls at a chunk boundary (barrier, first fragments of the
// next chunk), which two symmetric workgroups per CU (strip form D) cover for each other.  FRCNN_BF16_RES=1 opts in (A/B), FRCNN_BF16_DMA=921 / 922 selects one.
static int conv_bf16_default_res_form(int CinP, int CoutP, int H, int W, int out_mode) {
#ifndef FRCNN_TUNING_FORMS
    (void)CinP; (void)CoutP; (void)H; (void)W; (void)out_mode;
    return 0;
#else
    const char *re = frcnn_tune("FRCNN_BF16_RES");
    if (!re || re[0] != '1' || frcnn_tune("FRCNN_BF16_RP") || frcnn_tune("FRCNN_BF16_DMA_DEFAULT") || frcnn_tune("FRCNN_BF16_SPLIT")) return 0;
    if (out_mode != 0 && out_mode != 2) return 0;
    const long cus = frcnn_cu_count() > 0 ? frcnn_cu_count() : 256;
    if (CinP == 64 && (long)frcnn_cdiv(W, 32) * frcnn_cdiv(H, 8) * frcnn_cdiv(CoutP, 64) >= 2 * cus) return 21;
    if (CinP == 128 && (long)frcnn_cdiv(W, 32) * frcnn_cdiv(H, 16) * frcnn_cdiv(CoutP, 32) >= 2 * cus) return 22;
    return 0;
#endif
}

static int conv_bf16_default_strip_form(int CinP, int CoutP, int H, int W, int out_mode) {
    const char *rp_env = frcnn_tune("FRCNN_BF16_RP");
    if (rp_env && atoi(rp_env) == 4) return 0;
    if (CinP / kCK < 8 || frcnn_tune("FRCNN_BF16_DMA_DEFAULT") || frcnn_tune("FRCNN_BF16_SPLIT")) return 0;
    const char *se = frcnn_tune("FRCNN_BF16_STRIP");
    if (se && se[0] == '0') return 0;
    const long cus = frcnn_cu_count() > 0 ? frcnn_cu_count() : 256;
    const long wgs_d = (long)frcnn_cdiv(W, 32) * frcnn_cdiv(H, 10) * frcnn_cdiv(CoutP, 64);
    const long wgs_c = (long)frcnn_cdiv(W, 32) * frcnn_cdiv(H, 5) * frcnn_cdiv(CoutP, 32);
    if (wgs_d >= cus) return 10;                                                      // form D (910: its un-pooled bf16 output stored straight from the accumulators); even tile rows: pool-capable
    if (CinP / kCK >= 16 && (CinP / kCK) % 4 == 0 && out_mode != 2 && 2 * wgs_c > cus && wgs_c <= cus) return 3;       // form C
    return 0;
}

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

// split-K factor for a 3x3 launch.  Measured on MI355X (r01, scripts/conv_bf16_sweep.py with FRCNN_BF16_SPLIT=1/2/4): splitting does
// not pay on any VGG layer -- conv5_x (160 tiles on 256 CUs) 27 / 26 / 30 us, conv4_2 53 / 62 / 79 us -- the per-CU LDS-DMA rate,
// not the length of a workgroup's chunk chain, is what bounds the small launches.  So the default is 1; FRCNN_BF16_SPLIT=2/4
// keeps the path reachable for other shapes (it is exercised by the emulator and GPU tests).
static int conv_bf16_pick_split(long tiles, int chunks) {
    (void)tiles;
    const char *e = frcnn_tune("FRCNN_BF16_SPLIT");
    int s = e ? atoi(e) : 1;
    if (s != 2 && s != 4) s = 1;
    while (s > 1 && chunks / s < 4) s >>= 1;                      // a split should still carry a few chunks
    return s;
}
constexpr size_t kBf16CounterPageBytes = 64 * 1024;

size_t frcnn_conv_bf16_workspace_bytes(int Cin, int Cout, int H, int W) {
    if (Cin < 1 || Cout < 1 || H < 1 || W < 1) return 0;
    const long tiles = (long)frcnn_cdiv(W, 32) * frcnn_cdiv(H, 4) * frcnn_cdiv(frcnn_bf16_padded_channels(Cout), 64);
    return kBf16CounterPageBytes + (size_t)tiles * 4 * 256 * 32 * sizeof(float);      // up to 4 splits x 32 KB of accumulators per tile
}

int frcnn_conv_bf16_workspace_init(void *workspace, size_t workspace_bytes, void *stream) {
    if (!workspace || workspace_bytes < kBf16CounterPageBytes) return FRCNN_ERR_INVALID;
    FRCNN_HIP_TRY(hipMemsetAsync(workspace, 0, kBf16CounterPageBytes, (hipStream_t)stream));
    return FRCNN_OK;
}

int frcnn_conv_bf16_ws(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H, int W, int ksize, int relu,
                       int out_mode, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w_packed || !bias || !y || Cin < 1 || Cout < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if ((ksize != 1 && ksize != 3) || out_mode < 0 || out_mode > 2 || (out_mode == 2 && !relu)) return FRCNN_ERR_INVALID;
    if (out_mode == 1 && (size_t)Cout * H * W * 4 >= (1ull << 31)) return FRCNN_ERR_INVALID;       // the fp32 output sits behind a 32-bit buffer range
    const int CinP = frcnn_bf16_padded_channels(Cin), CoutP = frcnn_bf16_padded_channels(Cout);
    if ((size_t)H * W * CinP * 2 >= (1ull << 31) || (size_t)ksize * ksize * CoutP * CinP * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;
    const int xtiles = frcnn_cdiv(W, 32), cotiles = frcnn_cdiv(CoutP, 64);
    // 8-row tiles (8 waves) carry 1.7x the MFMA work per staged byte but measured no faster than 4-row tiles on any VGG layer
    // (scripts/conv_bf16_sweep.py, r01); kept as a tuning hook: FRCNN_BF16_RP=4 selects them.
    const char *rp_env = frcnn_tune("FRCNN_BF16_RP");
#ifdef FRCNN_TUNING_FORMS
    const bool big = rp_env && atoi(rp_env) == 4;
#else
    if (rp_env && atoi(rp_env) == 4) return FRCNN_ERR_INVALID;     // the 8-wave register-staged kernel: research builds only
    constexpr bool big = false;
#endif
    const int ytiles = frcnn_cdiv(H, big ? 8 : 4);
    const dim3 grid(xtiles * ytiles * cotiles);
    // 3x3: LDS-DMA staging.  Launches with at least four tiles per CU run single-stage rings -- 25 KB of LDS per workgroup, up to six
    // workgroups per CU, the other workgroups' MFMAs cover a workgroup's wait -- smaller launches (the 75x125 and 38x63 maps: 0.6-2.4
    // tiles per CU) get the two-stage ring that overlaps the next chunk's DMA with the MFMAs inside the workgroup
    // (scripts/conv_bf16_sweep.py, r01: +20...45 % over register staging on every VGG layer).  FRCNN_BF16_DMA overrides
    // (digits = ring stages, waves/SIMD budget, row pairs per wave; 0 = the register-staged kernel; 4-digit values = timing ablations,
    // compiled only with FRCNN_TIMING_ABLATIONS).
    const char *dma_env = frcnn_tune("FRCNN_BF16_DMA");
    int mode = dma_env ? atoi(dma_env) : -1;
    // Default pick, first rule (measured on the MI355X, profiles/r03_conv_bf16_strip_micro.txt; same box, us, strip form vs conv_dma_bf16_kernel's
    // best mode): a launch with at least 8 K-chunks and at least one 64-cout x 10-row x 32-px tile per CU runs as strip form D (two workgroups
    // per CU, each one wave per SIMD with a two-stage software-pipelined ring) -- conv2_2 41.6 vs 43.5, conv3_1 24.2 vs 26.4, conv3_2/3 39.5 /
    // 39.1 vs 44.0 / 42.5, conv4_1..3 23.2 / 39.8 / 39.1 vs 26.2 / 45.7 / 45.3; a smaller launch that ONE round of form C covers (32 couts x 5
    // rows, the K loop split over the four waves: more than half the CUs, at most all) runs as form C -- the four 38 x 63 launches 15.1 vs
    // 23.1.  Short K loops (conv1_2, conv2_1: 49.6 / 30.8 vs 48.4 / 28.2) and everything else stay on conv_dma_bf16_kernel.  Form D is
    // bit-identical to it; form C sums four partial accumulators (fp32 rounding).  The one-workgroup-per-CU forms A and B (41.0 / 40.0 us on
    // conv3_2 / conv4_2) stay selectable (901, 902).  FRCNN_BF16_STRIP=0 switches the rule off (A/B measurements); the tuning hooks that
    // select a kernel family (FRCNN_BF16_RP, FRCNN_BF16_DMA_DEFAULT, FRCNN_BF16_SPLIT) keep their meaning.
    if (ksize == 3 && mode < 0) {
#ifdef FRCNN_TUNING_FORMS
        const int rform = conv_bf16_default_res_form(CinP, CoutP, H, W, out_mode);
        if (rform && conv_bf16_res(rform, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream) == 0) return frcnn_launch_status();
#endif
        const int form = conv_bf16_default_strip_form(CinP, CoutP, H, W, out_mode);
        if (form && conv_bf16_strip(form, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream) == 0) return frcnn_launch_status();
    }
    if (ksize == 3 && (mode == 921 || mode == 922)) {              // an explicitly requested resident form (its fp32-NCHW output: conv_dma_bf16_kernel's picks)
#ifdef FRCNN_TUNING_FORMS
        if (conv_bf16_res(mode - 900, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream) == 0) return frcnn_launch_status();
        if (out_mode != 1) return FRCNN_ERR_INVALID;
        mode = -1;
#else
        return FRCNN_ERR_INVALID;                                  // research builds only
#endif
    }
    if (ksize == 3 && mode >= 9010 && mode <= 9039) {             // 90<form><ablation> (FRCNN_TIMING_ABLATIONS builds; else the plain form)
        const char *ae = frcnn_tune("FRCNN_BF16_STRIP_ABL");
        return conv_bf16_strip((mode - 9000) / 10, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream, ae ? atoi(ae) : mode % 10) == 0
                   ? frcnn_launch_status() : FRCNN_ERR_INVALID;
    }
    if (ksize == 3 && mode >= 900 && mode <= 911) {
        const int rc = conv_bf16_strip(mode - 900, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, stream);
        if (rc == 0) return frcnn_launch_status();
        if (mode != 900) return FRCNN_ERR_INVALID;                // an explicitly requested form that does not apply to this launch
        mode = -1;                                                // 900: the measured default picks below
    }
    // tile rows / accumulators per thread of the DMA kernel's shapes (RPW = last digit of the mode): 64 couts x {4, 8, 8, 16} rows x 32 px
    static const int kRowsOf[5] = {0, 4, 8, 8, 16}, kAccOf[5] = {0, 2, 4, 4, 8};
    if (mode < 0) {
        const char *def_env = frcnn_tune("FRCNN_BF16_DMA_DEFAULT");       // "<big launches>,<small launches>", e.g. 224,223 (tuning hook)
        int big_mode = 141, small_mode = 231;
        if (def_env) {
            big_mode = atoi(def_env);
            const char *comma = strchr(def_env, ',');
            small_mode = comma ? atoi(comma + 1) : big_mode;
        }
        // measured picks per launch size (scripts/micro/conv_bf16_micro, r03: every mode on every VGG-16 layer shape at 600 x 1000):
        //   >= 8 four-row tiles per CU (conv1_2, conv2_x)  8-row tiles, single stage (132): 44.3 / 25.2 / 40.2 us vs 47.5 / 27.4 / 43.5 with 141
        //   4 .. 8 per CU (conv3_x)                        4-row tiles, single stage, six workgroups per CU (141)
        //   2 .. 4 per CU with 32 K-chunks (conv4_2/3)     8-row tiles (132): 45.2 vs 46.6 us
        //   fewer (conv4_1, the 38 x 63 maps)              two-stage ring (231)
        const long per_cu4 = (long)grid.x / (frcnn_cu_count() > 0 ? frcnn_cu_count() : 256);
        if (def_env) mode = (long)grid.x >= 4L * frcnn_cu_count() ? big_mode : small_mode;
        else if (per_cu4 >= 8) mode = 132;
        else if (per_cu4 >= 4) mode = 141;
        else if (per_cu4 >= 2 && CinP >= 512) mode = 132;
        else mode = 231;
    }
    const int rpw = (mode > 0 && mode < 1000) ? mode % 10 : 1;
    if (mode > 0 && (rpw < 1 || rpw > 4)) return FRCNN_ERR_INVALID;
    const int trows = kRowsOf[rpw], nacc = kAccOf[rpw];
    const int yt = frcnn_cdiv(H, trows);
    const long dma_tiles = (long)xtiles * yt * cotiles;
    // split-K needs the workspace (partial tiles + the zeroed counter page); without one every tile is whole
    int nsplit = 1;
    if (ksize == 3 && !big && mode > 0 && workspace && dma_tiles * 4 <= 16384) {
        nsplit = conv_bf16_pick_split(dma_tiles, CinP / kCK);
        if (workspace_bytes < kBf16CounterPageBytes + (size_t)dma_tiles * nsplit * 256 * nacc * 16 * sizeof(float)) nsplit = 1;
    }
    float *partials = nsplit > 1 ? (float *)((char *)workspace + kBf16CounterPageBytes) : nullptr;
    int *counters = nsplit > 1 ? (int *)workspace : nullptr;
#ifdef FRCNN_TUNING_FORMS
    if (ksize == 3 && big) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_bf16_kernel<3, 4>), grid, dim3(512), 0, stream, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles);
    else
#endif
    if (ksize == 3 && mode > 0) {
        dim3 dgrid((unsigned)(dma_tiles * nsplit));
        int xcd_cotiles = 0;
        const char *xcd_env = frcnn_tune("FRCNN_BF16_XCD");               // 1 enables: measured 1.5 % SLOWER on the VGG chain here (r02n), so off
        if ((cotiles == 1 || cotiles == 2 || cotiles == 4 || cotiles == 8) && xcd_env && xcd_env[0] == '1') {
            xcd_cotiles = cotiles;
            dgrid = dim3((unsigned)(8 * frcnn_cdiv((int)((long)xtiles * yt * nsplit), 8 / cotiles)));
        }
#define FRCNN_DMA_CASE(NS, WPS, RPW)                                                                                                     \
    case NS * 100 + WPS * 10 + RPW:                                                                                                      \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_dma_bf16_kernel<NS, WPS, RPW>), dgrid, dim3(256), 0, stream, x, w_packed, bias, y, CinP,  \
                           Cout, CoutP, H, W, relu, out_mode, xtiles, yt, nsplit, partials, counters, xcd_cotiles);                       \
        break;
#define FRCNN_DMA_ABL(NS, WPS, A)                                                                                                        \
    case NS * 1000 + WPS * 100 + 10 + A:                                                                                                 \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_dma_bf16_kernel<NS, WPS, 1, A>), grid, dim3(256), 0, stream, x, w_packed, bias, y, CinP,   \
                           Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles, 1, nullptr, nullptr, 0);                                    \
        break;
        switch (mode) {
            FRCNN_DMA_CASE(2, 3, 1) FRCNN_DMA_CASE(1, 4, 1) FRCNN_DMA_CASE(1, 3, 2)          // 231 / 141 / 132: the measured picks of the default rule below
#ifdef FRCNN_TUNING_FORMS
            FRCNN_DMA_CASE(3, 2, 1) FRCNN_DMA_CASE(2, 2, 2)
            FRCNN_DMA_CASE(3, 1, 1) FRCNN_DMA_CASE(4, 1, 1) FRCNN_DMA_CASE(5, 1, 1) FRCNN_DMA_CASE(6, 1, 1)
            FRCNN_DMA_CASE(2, 2, 3) FRCNN_DMA_CASE(2, 3, 3) FRCNN_DMA_CASE(3, 2, 3) FRCNN_DMA_CASE(2, 2, 4) FRCNN_DMA_CASE(3, 2, 4)
            FRCNN_DMA_CASE(1, 2, 4) FRCNN_DMA_CASE(1, 3, 3)
#endif
#ifdef FRCNN_TIMING_ABLATIONS                                                                       // WRONG results: sweeps only, never shipped
            FRCNN_DMA_ABL(1, 4, 1) FRCNN_DMA_ABL(1, 4, 4) FRCNN_DMA_ABL(2, 3, 1) FRCNN_DMA_ABL(2, 3, 4)
#endif
            default: return FRCNN_ERR_INVALID;
        }
#undef FRCNN_DMA_CASE
#undef FRCNN_DMA_ABL
    }
    else if (ksize == 3) {
        const char *abl_env = frcnn_tune("FRCNN_BF16_ABL");
        const int abl = abl_env ? atoi(abl_env) : 0;
#define FRCNN_ABL_CASE(A) case A: hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_bf16_kernel<3, 2, A>), grid, dim3(256), 0, stream, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles); break;
        switch (abl) {
#ifdef FRCNN_TIMING_ABLATIONS
            FRCNN_ABL_CASE(1) FRCNN_ABL_CASE(2) FRCNN_ABL_CASE(3)
#endif
            default: hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_bf16_kernel<3, 2>), grid, dim3(256), 0, stream, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles);
        }
#undef FRCNN_ABL_CASE
    }
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_bf16_kernel<1, 2>), grid, dim3(256), 0, stream, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles);
    return frcnn_launch_status();
}

int frcnn_rpn_heads_bf16(const uint16_t *h, int Cmid, int H, int W, int A, const uint16_t *w_packed, const float *bias, float *raw, float *cls_prob,
                         void *stream) {
    if (!h || !w_packed || !bias || !raw || !cls_prob || Cmid < 1 || H < 1 || W < 1 || A < 1 || 6 * A > 64 || 2 * A > 32) return FRCNN_ERR_INVALID;
    const int CinP = frcnn_bf16_padded_channels(Cmid), CoutP = frcnn_bf16_padded_channels(6 * A);
    if ((size_t)CinP * H * W * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(rpn_heads_bf16_fused_kernel, dim3(frcnn_cdiv(H * W, 32)), dim3(64 * kHeadWavesB), 0, (hipStream_t)stream, h, w_packed, bias, raw,
                       cls_prob, CinP, CoutP, H * W, 6 * A, 2 * A);
    return frcnn_launch_status();
}

int frcnn_conv_bf16(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H, int W, int ksize, int relu,
                    int out_mode, void *stream) {
    return frcnn_conv_bf16_ws(x, w_packed, bias, y, Cin, Cout, H, W, ksize, relu, out_mode, nullptr, 0, stream);
}

int frcnn_conv_bf16_plan(int Cin, int Cout, int H, int W, int ksize, int out_mode) {
    if (Cin < 1 || Cout < 1 || H < 1 || W < 1 || (ksize != 1 && ksize != 3) || out_mode < 0 || out_mode > 2) return FRCNN_ERR_INVALID;
    if (ksize != 3) return 0;
    const int CinP = frcnn_bf16_padded_channels(Cin), CoutP = frcnn_bf16_padded_channels(Cout);
    const char *dma_env = frcnn_tune("FRCNN_BF16_DMA");
    const int mode = dma_env ? atoi(dma_env) : -1;
    // the same three branches as frcnn_conv_bf16_ws, through the same resolver: what is returned is the form that would be LAUNCHED
    if (mode == 921 || mode == 922) {
#ifdef FRCNN_TUNING_FORMS
        if (conv_bf16_res_applies(mode - 900, CinP, CoutP, H, W, out_mode)) return mode;
        return out_mode == 1 ? 0 : FRCNN_ERR_INVALID;             // (its fp32-NCHW output takes conv_dma_bf16_kernel's picks; anything else it declines is refused)
#else
        return FRCNN_ERR_INVALID;                                  // research builds only
#endif
    }
    if (mode < 0) {
#ifdef FRCNN_TUNING_FORMS
        const int rform = conv_bf16_default_res_form(CinP, CoutP, H, W, out_mode);
        if (rform && conv_bf16_res_applies(rform, CinP, CoutP, H, W, out_mode)) return 900 + rform;
#endif
        const int pick = conv_bf16_default_strip_form(CinP, CoutP, H, W, out_mode);
        const int form = pick ? conv_bf16_strip_resolve(pick, CinP, CoutP, H, W, out_mode) : 0;
        return form ? 900 + form : 0;
    }
    if (mode >= 9010 && mode <= 9039) {
        const int form = conv_bf16_strip_resolve((mode - 9000) / 10, CinP, CoutP, H, W, out_mode);
        return form ? 900 + form : FRCNN_ERR_INVALID;
    }
    if (mode >= 900 && mode <= 911) {
        const int form = conv_bf16_strip_resolve(mode - 900, CinP, CoutP, H, W, out_mode);
        if (form) return 900 + form;
        return mode == 900 ? 0 : FRCNN_ERR_INVALID;               // 900 falls back to conv_dma_bf16_kernel's picks; an explicit form that does not apply is refused
    }
    return 0;
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

// LDS-DMA form of the bf16 FC kernel (K a multiple of 64): a 64-k panel row is 128 B = eight 16-byte groups (8 bf16 each), staged as
// 1 KB pieces of 8 rows into a lane-linear image with the XOR swizzle of the fp32 FC kernel (gemm.hip): group g of row r in slot
// 8r + (g ^ ((r >> 1) & 7)).  k-step ks of a panel reads group 2ks + (lane >> 5) of the lane's row -- one conflict-free ds_read_b128
// per operand per MFMA, no staging registers, no ds_write_b128 pass (this kernel is staging-bound: 460 B staged per MFMA).
template <int AM>
__global__ void __launch_bounds__(256, 2)
linear_dma_bf16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w, float *__restrict__ part, int M, int N, int K, int k_per_split) {
    constexpr int BM = 32 * AM, BN = 128;
    constexpr int XP = BM / 8, WP = BN / 8;                  // 1 KB pieces per panel (8 rows each)
    constexpr int PPW = (XP + WP) / 4;
    constexpr int STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_per_split, k_end = min(K, k_begin + k_per_split);
    const int nchunks = (k_end - k_begin) / kLBK;            // whole panels: the host guarantees K % 64 == 0
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)M * K * 2));
    const frcnn_buf_t wbuf = frcnn_make_buf(w, (uint32_t)((size_t)N * K * 2));
    uint32_t poff[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pid = wave + 4 * q;
        const bool isx = pid < XP;
        const int sl = (isx ? pid : pid - XP) * 64 + lane, row = sl >> 3, g = (sl & 7) ^ ((row >> 1) & 7);
        const int gr = (isx ? m0 : n0) + row;
        poff[q] = gr < (isx ? M : N) ? (uint32_t)(((size_t)gr * K + k_begin + 8 * g) * 2) : kBufOob;
    }
    auto issue = [&](int chunk, int stage) {
        unsigned char *dst = lds + stage * STAGE + wave * 1024;
        const uint32_t so = (uint32_t)chunk * (kLBK * 2);
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            if (4 * q + 3 < XP) frcnn_buf_load_lds_b128(xbuf, dst + q * 4096, poff[q], so);
            else if (4 * q >= XP) frcnn_buf_load_lds_b128(wbuf, dst + q * 4096, poff[q], so);
            else frcnn_buf_load_lds_b128(wave + 4 * q < XP ? xbuf : wbuf, dst + q * 4096, poff[q], so);
        }
    };
    frcnn_f32x16 acc[AM];
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int l31 = lane & 31, khalf = lane >> 5;
    auto frag_off = [&](int row, int ks) { return (uint32_t)(row * 128 + (((2 * ks + khalf) ^ ((row >> 1) & 7)) << 4)); };
    if (nchunks > 0) issue(0, 0);
    frcnn_wait_vmcnt<0>();
    frcnn_barrier_nofence();
    int cur = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 < nchunks) issue(chunk + 1, cur ^ 1);
        const unsigned char *xs = lds + cur * STAGE, *wsm = xs + BM * 128;
#pragma unroll
        for (int ks = 0; ks < kLBK / 16; ++ks) {
            const uint4 b = *reinterpret_cast<const uint4 *>(wsm + frag_off(wave * 32 + l31, ks));
#pragma unroll
            for (int i = 0; i < AM; ++i) {
                const uint4 a = *reinterpret_cast<const uint4 *>(xs + frag_off(32 * i + l31, ks));
                acc[i] = frcnn_mfma_32x32x16_bf16(a, b, acc[i]);
            }
        }
        frcnn_wait_vmcnt<0>();
        frcnn_barrier_nofence();
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
        const float b = bias[i % N];                         // issued ahead of the slab loads: it is used last
        float v = frcnn_sum_splits(part, total, i, splits);
        v += b;
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
    const bool dma = (K % kLBK) == 0 && (p.k_per_split % kLBK) == 0 && !frcnn_tune("FRCNN_LINEAR_NODMA");
    if (dma && p.am == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_bf16_kernel<5>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (dma && p.am == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_bf16_kernel<3>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (dma) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_dma_bf16_kernel<1>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (p.am == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_bf16_kernel<5>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (p.am == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_bf16_kernel<3>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_mfma_bf16_kernel<1>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    const size_t total = (size_t)M * N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(linear_reduce_bf16_kernel, dim3(blocks), dim3(256), 0, stream, part, bias, y, M, N, p.splits, relu, out_bf16);
    return frcnn_launch_status();
}

}  // extern "C"
