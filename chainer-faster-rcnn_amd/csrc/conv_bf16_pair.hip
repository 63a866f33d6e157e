// conv_bf16_pair.hip -- the first TWO layers of the bf16 chain as ONE launch: conv1_1 (Cin <= 3 -> 64) + ReLU computed on the fly into LDS,
// conv1_2 (64 -> 64) + ReLU + 2x2 ceil-mode max-pool on the matrix cores from there (/root/reference/models/vgg16.py:38-44 -- conv1_1, conv1_2,
// F.max_pooling_2d -- on the config-3 path).
//
// Why (round 4, profiles/r04_bf16_kernel_stats.csv): as two launches the pair costs 26.8 + 37-44 us of the chain's 391: conv1_1 is bound by the
// 76.8 MB of its bf16 output (2 GFLOP), conv1_2 re-reads them through four K-chunks of LDS-DMA with a prologue and an epilogue per 64 x 8-row tile.
// Fused, the 64-channel map never exists in memory: a workgroup keeps a (TR+2) x 34-pixel halo tile of it in LDS -- 476 pixels x 128 B -- built by
// conv1_1's own MFMAs (K = 27 padded to 32: two k-steps) from a 7 KB fp32 image patch, and conv1_2's weights (64 x 64 x 9 bf16 = 73.7 KB) live in
// REGISTERS for the whole persistent kernel: 36 fragments of 16 B per lane (the wave's 32 couts), so the main loop's only LDS traffic is the halo
// fragments -- 0.44 ds_read_b128 per MFMA, no DMA ring, no stage hand-over, one barrier pair per tile.  HBM traffic of the pair: 7.2 MB image +
// 19.2 MB pooled output instead of 7.2 + 76.8 + 76.8 + 19.2.
//
// Bit-compatibility: conv1_1 is conv1_f32s_kernel<2, false>'s arithmetic (csrc/conv_f32s.hip: k = ci * 9 + tap, k-step s2 holds k = 16 s2 + 8 khalf
// + e, operands rounded to bf16 nearest-even, two MFMAs per accumulator in s2 order, + bias, ReLU, one rounding to bf16); conv1_2 keeps the
// (chunk, tap) accumulation order of every conv_dma_bf16_kernel / strip form; the pool takes the maximum of the fp32 values before the one rounding
// (rounding and + bias are monotone: the same bits as the maximum of the rounded values).  So the launch equals the two-launch chain bit for bit
// (tests: emulator + GPU, word by word), and is priced against the oracle exactly like them.
//
// Shape: 256 threads = 4 waves = 2 row groups x 2 cout ways; a wave owns RW rows x 32 px x 32 couts (RW accumulators); tile = 2 RW rows x 32 px,
// one workgroup per CU striding over the tiles.
#include "frcnn_common.h"
#include <stdlib.h>
#include <frcnn_buffer.h>   // angle brackets: shadowed by the test emulator
#include <frcnn_intrin.h>

namespace {

template <int RW>
struct PairShape {
    static constexpr int TR = 2 * RW, HR = TR + 2, HPX = 34;
    static constexpr int NPIX = HR * HPX;                      // pixels of the conv1_1 halo tile
    static constexpr int NGRP = (NPIX + 31) / 32;              // conv1_1 units: 32 consecutive halo pixels x 64 couts
    static constexpr int REGION = NGRP * 1024;                 // one 16-channel chunk of the tile: 32 B per pixel, swizzled (conv_bf16.hip's LDS image)
    static constexpr int PR = TR + 4, PC = 36, PP = 37;        // fp32 image patch: rows, columns, LDS pitch (floats)
    static constexpr int PATCH = 3 * PR * PP;
    static constexpr int NPL = (3 * PR * PC + 255) / 256;      // patch elements per thread
    static_assert(RW % 2 == 0 && RW >= 2 && RW <= 8, "row pairs of the fused pool stay inside a wave");
    static_assert(4 * REGION + PATCH * 4 + 1024 <= 160 * 1024, "LDS");
};

template <int RW>
__global__ void __launch_bounds__(256, 1)
conv1_pair_bf16_kernel(const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1, const uint16_t *__restrict__ w2p,
                       const float *__restrict__ b2, uint16_t *__restrict__ y, int Cin, int H, int W, int xtiles, int ntiles) {
    using S = PairShape<RW>;
    constexpr int TR = S::TR, HPX = S::HPX, NPIX = S::NPIX, NGRP = S::NGRP, REGION = S::REGION, PR = S::PR, PC = S::PC, PP = S::PP, NPL = S::NPL;
    __shared__ __attribute__((aligned(1024))) unsigned char img[4 * REGION];
    __shared__ float patch[S::PATCH];
    __shared__ __attribute__((aligned(16))) float sb1[64], sb2[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & 1, cw = wave >> 1;                   // row group; cout way (32 couts)
    const int l31 = lane & 31, khalf = lane >> 5;
    const int K = Cin * 9, OH = (H + 1) / 2, OW = (W + 1) / 2;
    if (tid < 64) { sb1[tid] = b1[tid]; sb2[tid] = b2[tid]; }
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)Cin * H * W * 4));
    const frcnn_buf_t ybuf = frcnn_make_buf(y, (uint32_t)((size_t)64 * OH * OW * 2));

    // ---- conv1_2's weight fragments, resident in registers: [chunk][tap] x 16 B = cout cw*32 + l31, channels chunk*16 + khalf*8 .. + 7
    uint4 wa[4][9];
    {
        const frcnn_buf_t wbuf = frcnn_make_buf(w2p, 4u * 9u * 64u * 32u);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 v = frcnn_buf_load_f32x4(wbuf, (uint32_t)(((c * 9 + t) * 64 + cw * 32 + l31) * 32 + khalf * 16));
                wa[c][t] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
            }
    }
    // ---- conv1_1's weight fragments (64 couts x 32 k as bf16): lane = (cout l31 of block cb, k = 16 s2 + 8 khalf + e), k = ci * 9 + tap, from
    //      Chainer's (64, Cin, 3, 3) fp32 array; k >= K: zero
    uint4 a1[2][2];
    {
        const frcnn_buf_t w1buf = frcnn_make_buf(w1, (uint32_t)(64 * K) * 4u);
        float wv[2][2][8];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * s2 + 8 * khalf + e, co = cb * 32 + l31;
                    wv[cb][s2][e] = frcnn_buf_load_f32(w1buf, k < K ? (uint32_t)(co * K + k) * 4u : kBufOob);
                }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                a1[cb][s2] = make_uint4(frcnn_pack_bf16x2(wv[cb][s2][0], wv[cb][s2][1]), frcnn_pack_bf16x2(wv[cb][s2][2], wv[cb][s2][3]),
                                        frcnn_pack_bf16x2(wv[cb][s2][4], wv[cb][s2][5]), frcnn_pack_bf16x2(wv[cb][s2][6], wv[cb][s2][7]));
    }
    // ---- im2col: patch offsets (floats) of this lane's 16 k-values relative to (halo row, halo column) of its pixel.  A k >= K reads the pixel's own
    //      first tap (a finite value; its weight is zero) -- as conv1_f32s_kernel does
    int koff[2][8];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * s2 + 8 * khalf + e, kk = k < K ? k : 0;
            const int ci = kk / 9, tap = kk - ci * 9, ky = tap / 3, kx = tap - ky * 3;
            koff[s2][e] = (ci * PR + ky) * PP + kx;
        }
    // ---- halo fragment offsets inside a chunk image: pixel (rg * RW + r) * 34 + l31 + kx, 16 B at (khalf ^ ((P >> 3) & 1)) << 4
    uint32_t b_off[RW + 2][3];
#pragma unroll
    for (int r = 0; r < RW + 2; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = (rg * RW + r) * HPX + l31 + kx;
            b_off[r][kx] = (uint32_t)(P * 32 + ((khalf ^ ((P >> 3) & 1)) << 4));
        }

    // the fp32 image patch of a tile: rows y0 - 2 .. y0 + TR + 1, columns x0 - 2 .. x0 + 33 of every channel; zero outside the image / past Cin
    float pv[NPL];
    auto load_patch = [&](int tile) {
        const int ty = tile / xtiles, tx = tile - ty * xtiles;
        const int gy0 = ty * TR - 2, gx0 = tx * 32 - 2;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int e = tid + 256 * j;
            const int ci = e / (PR * PC), rem = e - ci * (PR * PC), pr = rem / PC, pc = rem - pr * PC;
            const int gy = gy0 + pr, gx = gx0 + pc;
            const bool ok = e < 3 * PR * PC && ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W;
            pv[j] = frcnn_buf_load_f32(xbuf, ok ? (uint32_t)((ci * H + gy) * W + gx) * 4u : kBufOob);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int e = tid + 256 * j;
            const int ci = e / (PR * PC), rem = e - ci * (PR * PC), pr = rem / PC, pc = rem - pr * PC;
            if (e < 3 * PR * PC) patch[(ci * PR + pr) * PP + pc] = pv[j];
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_patch(tile);
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
        const int ty = tile / xtiles, tx = tile - ty * xtiles;
        const int x0 = tx * 32, y0 = ty * TR;
        store_patch();                                                 // (every wave is past the previous tile's conv1_1 units: barrier B2 below)
        __syncthreads();                                               // B1: the patch is complete; every wave has left the previous tile's main loop
        // ---- conv1_1 units of this wave: halo pixels gi * 32 .. + 31 x 64 couts -> bf16 into the four chunk images
#pragma unroll 1
        for (int gi = wave; gi < NGRP; gi += 4) {
            const int P = gi * 32 + l31, Pc = P < NPIX ? P : NPIX - 1;
            const int hr = Pc / HPX, hx = Pc - hr * HPX;
            const int base = hr * PP + hx;
            uint4 bq[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = patch[base + koff[s2][e]];
                bq[s2] = make_uint4(frcnn_pack_bf16x2(v[0], v[1]), frcnn_pack_bf16x2(v[2], v[3]), frcnn_pack_bf16x2(v[4], v[5]), frcnn_pack_bf16x2(v[6], v[7]));
            }
            frcnn_f32x16 c1[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) c1[cb][r] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) c1[cb] = frcnn_mfma_32x32x16_bf16(a1[cb][s2], bq[s2], c1[cb]);
            // conv1_2 pads with zeros: a halo pixel outside the image is 0, not conv1_1 evaluated there
            const int gy = y0 - 1 + hr, gx = x0 - 1 + hx;
            const bool inside = P < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
            unsigned char *dst = img + P * 32 + 8 * khalf;
            const int sw = (P >> 3) & 1;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb * 32 + 8 * g + 4 * khalf;    // first of four consecutive channels: chunk cb * 2 + (g >> 1), half g & 1
                    const float4 bb = *reinterpret_cast<const float4 *>(&sb1[co]);
                    float v[4] = {c1[cb][4 * g] + bb.x, c1[cb][4 * g + 1] + bb.y, c1[cb][4 * g + 2] + bb.z, c1[cb][4 * g + 3] + bb.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = inside ? fmaxf(v[t], 0.0f) : 0.0f;
                    *reinterpret_cast<uint2 *>(dst + (cb * 2 + (g >> 1)) * REGION + (((g & 1) ^ sw) << 4)) =
                        make_uint2(frcnn_pack_bf16x2(v[0], v[1]), frcnn_pack_bf16x2(v[2], v[3]));
                }
        }
        if (tile + (int)gridDim.x < ntiles) load_patch(tile + gridDim.x);      // in flight under the main loop
        __syncthreads();                                               // B2: the halo tile is complete

        // ---- conv1_2: 4 chunks x 9 taps x RW MFMAs, halo fragments read one tap group ahead (at most 2 RW <= 14 reads in flight)
        frcnn_f32x16 acc[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        uint4 fb[RW + 2][3];
        auto read_group = [&](int c, int g) {
            const unsigned char *st = img + c * REGION;
            const int ky = g / 3, kx = g - ky * 3;
            if (ky == 0) {
#pragma unroll
                for (int r = 0; r < RW; ++r) fb[r][kx] = *reinterpret_cast<const uint4 *>(st + b_off[r][kx]);
            } else fb[ky + RW - 1][kx] = *reinterpret_cast<const uint4 *>(st + b_off[ky + RW - 1][kx]);
        };
        read_group(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_group(0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 9; ++g) {
                if (g >= 1 && g + 1 < 9) {
                    read_group(c, g + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (g == 8 && c + 1 < 4) {
                    read_group(c + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    read_group(c + 1, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int ky = g / 3, kx = g - ky * 3;
#pragma unroll
                for (int j = 0; j < RW; ++j) acc[j] = frcnn_mfma_32x32x16_bf16(wa[c][g], fb[ky + j][kx], acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }

        // ---- epilogue: 2x2 ceil-mode max over (row pair of this wave, lane pair) of the fp32 sums, + bias, ReLU, one rounding; register r of lane l =
        //      cout cw * 32 + (r & 3) + 8 (r >> 2) + 4 khalf of pixel l31.  Even lanes store the quads g = 0, 1, odd lanes g = 2, 3 (8 B each).
        const int px = x0 + l31, odd = l31 & 1;
        const bool own_ok = px < W, other_ok = (px ^ 1) < W;
#pragma unroll
        for (int m = 0; m < RW / 2; ++m) {
            const int py = y0 + rg * RW + 2 * m;
            const bool row2 = py + 1 < H;
            uint2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bb = *reinterpret_cast<const float4 *>(&sb2[cw * 32 + 8 * g + 4 * khalf]);
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float a = acc[2 * m][4 * g + t], b = acc[2 * m + 1][4 * g + t];
                    const float vr = row2 ? frcnn_max_f32(a, b) : a;
                    const float vo = frcnn_lane_xor1_f32(vr);
                    const float vm = !own_ok ? vo : (!other_ok ? vr : frcnn_max_f32(vr, vo));
                    v[t] = fmaxf(vm + bv[t], 0.0f);
                }
                pk[g] = make_uint2(frcnn_pack_bf16x2(v[0], v[1]), frcnn_pack_bf16x2(v[2], v[3]));
            }
            const bool st_ok = py < H && (px & ~1) < W;
            const int oy = py >> 1, ox = px >> 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = 2 * odd + h;                            // (lane-dependent: the value is selected, the registers are not indexed)
                const uint2 val = odd ? pk[2 + h] : pk[h];
                const int co = cw * 32 + 8 * g + 4 * khalf;
                frcnn_buf_store_b64(ybuf, st_ok ? (uint32_t)((((co >> 4) * OH + oy) * OW + ox) * 32 + (co & 15) * 2) : kBufOob, val);
            }
        }
    }
}

}  // namespace

extern "C" {

// conv1_1 + ReLU + conv1_2 + ReLU + 2x2 ceil-mode max-pool of the bf16 chain in one launch.  x (Cin <= 3, H, W) fp32 NCHW; w1 (64, Cin, 3, 3) fp32
// (Chainer's layout), b1 (64); w2_packed = frcnn_bf16_pack_conv_w of the (64, 64, 3, 3) weights ([4][9][64][16] bf16), b2 (64);
// y [4][ceil(H/2)][ceil(W/2)][16] bf16.  FRCNN_BF16_PAIR_RW = 4 | 6 selects the rows per wave (tile = 2 RW rows x 32 px; tuning hook).
int frcnn_conv1_pair_bf16(const float *x, const float *w1, const float *b1, const uint16_t *w2_packed, const float *b2, uint16_t *y, int Cin, int H,
                          int W, void *stream) {
    if (!x || !w1 || !b1 || !w2_packed || !b2 || !y || Cin < 1 || Cin > 3 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if ((size_t)Cin * H * W * 4 >= (1ull << 31)) return FRCNN_ERR_INVALID;
    const char *e = getenv("FRCNN_BF16_PAIR_RW");
    const int rw = e ? atoi(e) : 6;
    const int xtiles = frcnn_cdiv(W, 32);
    const int cus = frcnn_cu_count() > 0 ? frcnn_cu_count() : 256;
    if (rw == 4) {
        const int ntiles = xtiles * frcnn_cdiv(H, 8);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_pair_bf16_kernel<4>), dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(256), 0, (hipStream_t)stream, x, w1, b1,
                           w2_packed, b2, y, Cin, H, W, xtiles, ntiles);
    } else if (rw == 6) {
        const int ntiles = xtiles * frcnn_cdiv(H, 12);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_pair_bf16_kernel<6>), dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(256), 0, (hipStream_t)stream, x, w1, b1,
                           w2_packed, b2, y, Cin, H, W, xtiles, ntiles);
    } else return FRCNN_ERR_INVALID;
    return frcnn_launch_status();
}

}  // extern "C"
