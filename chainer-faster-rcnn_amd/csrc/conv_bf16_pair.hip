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
#include <type_traits>

#ifndef FRCNN_PAIR_ABL            // timing ablations (scripts/micro/conv_pair_stamps.hip builds only; WRONG results): 1 no patch loads, 2 no patch stores,
#define FRCNN_PAIR_ABL 0          // 4 no conv1_1 units, 8 no consumer epilogue, 16 no consumer MFMA loop
#endif
#ifdef FRCNN_PAIR_STAMPS          // scripts/micro/conv_pair_stamps.hip only: s_memtime at the phase boundaries of the first workgroup's tiles
__device__ unsigned long long frcnn_pair_stamps[8 * 16 * 8];
#define PAIR_STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && it < 16) frcnn_pair_stamps[(wave * 16 + it) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PAIR_STAMP(i) do { } while (0)
#endif

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
    int it = 0;
    (void)it;
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x, ++it) {
        const int ty = tile / xtiles, tx = tile - ty * xtiles;
        const int x0 = tx * 32, y0 = ty * TR;
        PAIR_STAMP(0);
        store_patch();                                                 // (every wave is past the previous tile's conv1_1 units: barrier B2 below)
        // the next tile's patch goes out NOW: a whole tile of time to arrive.  Both barriers are fence-less (LDS only, lgkmcnt): a __syncthreads()
        // would wait for these loads -- and for the previous tile's output stores -- at every barrier (first version: 82 us, a memory round trip per tile)
        if (tile + (int)gridDim.x < ntiles) load_patch(tile + gridDim.x);
        PAIR_STAMP(1);
        frcnn_barrier_nofence();                                       // B1: the patch is complete; every wave has left the previous tile's main loop
        PAIR_STAMP(2);
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
        PAIR_STAMP(3);
        frcnn_barrier_nofence();                                       // B2: the halo tile is complete
        PAIR_STAMP(4);

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

        PAIR_STAMP(5);
        // ---- epilogue: 2x2 ceil-mode max over (row pair of this wave, lane pair) of the fp32 sums, + bias, ReLU, one rounding; register r of lane l =
        //      cout cw * 32 + (r & 3) + 8 (r >> 2) + 4 khalf of pixel l31.  Even lanes store the quads g = 0, 1, odd lanes g = 2, 3 (8 B each).
        const int px = x0 + l31, odd = l31 & 1;
        const uint32_t oddm = 0u - (uint32_t)odd;
        const bool interior = y0 + TR <= H && x0 + 32 <= W;           // wave-uniform: every 2x2 window of the tile is whole
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
                    float vm;
                    if (interior) vm = frcnn_max_lane_xor1_f32(frcnn_max_f32(a, b));
                    else {
                        const float vr = row2 ? frcnn_max_f32(a, b) : a;
                        const float vo = frcnn_lane_xor1_f32(vr);
                        vm = !own_ok ? vo : (!other_ok ? vr : frcnn_max_f32(vr, vo));
                    }
                    v[t] = fmaxf(vm + bv[t], 0.0f);
                }
                pk[g] = make_uint2(frcnn_pack_bf16x2(v[0], v[1]), frcnn_pack_bf16x2(v[2], v[3]));
            }
            const bool st_ok = py < H && (px & ~1) < W;
            const int oy = py >> 1, ox = px >> 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = 2 * odd + h;                            // (lane-dependent: the value is selected, the registers are not indexed)
                const uint2 val = make_uint2((pk[2 + h].x & oddm) | (pk[h].x & ~oddm), (pk[2 + h].y & oddm) | (pk[h].y & ~oddm));     // v_bfi (a ?: on the array went through scratch)
                const int co = cw * 32 + 8 * g + 4 * khalf;
                frcnn_buf_store_b64(ybuf, st_ok ? (uint32_t)((((co >> 4) * OH + oy) * OW + ox) * 32 + (co & 15) * 2) : kBufOob, val);
            }
        }
        PAIR_STAMP(6);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Form 2 (default): producer / consumer waves.  The stamps of form 1 (scripts/micro/conv_pair_stamps.hip, one wave per SIMD) put a 12-row tile at
// 21.4 k clocks of which the MFMA loop is 7.7 k: with ONE wave on a SIMD every VALU instruction of the conv1_1 units (6.1 k) and of the epilogue
// (5.5 k, its 8-byte stores issue-bound) is paid in the open.  Here a workgroup is 8 waves, two per SIMD: waves 0..3 (consumers) run conv1_2's
// MFMA loop and its epilogue, waves 4..7 (producers) build the NEXT tile's conv1_1 map into the other half of a double-buffered LDS image
// meanwhile -- the SIMD's second wave fills the issue slots the matrix pipe leaves.  ONE fence-less barrier per tile: it says "tile k has been
// consumed, tile k + 1 is complete".  At two waves per SIMD a wave has 256 registers, so conv1_2's weights move to LDS (chunks 1..3: 55 KB;
// chunk 0 stays in registers -- all four would put the launch 3.8 KB over the 160 KB), tiles are 8 rows x 32 px (RW 4), a producer wave feeds its
// im2col from a wave-private 4-row fp32 patch (no workgroup-wide patch, no second barrier), and the pooled output leaves through a wave-private
// LDS tile as 16-byte stores, consecutive lanes consecutive addresses.  Same arithmetic, same order: bit-identical to form 1 and to the two launches.
struct Pair2 {
    static constexpr int RW = 4, TR = 8, HR = 10, HPX = 34, NPIX = HR * HPX;          // 340 halo pixels
    static constexpr int NGRP = (NPIX + 31) / 32;                                    // 11 conv1_1 units per tile
    static constexpr int REGION = 11008;                                             // bytes per 16-channel chunk image: 344 pixel slots, a multiple of 256 (bank phase)
    static constexpr int TILE = 4 * REGION;
    static constexpr int WCH = 9 * 64 * 32;                                          // one K-chunk of conv1_2's weights: 18432 B
    static constexpr int MP_ROWS = 4, MP_PITCH = 40, MP_FLOATS = 3 * MP_ROWS * MP_PITCH;       // a producer wave's fp32 patch: 3 ch x 4 rows x 40 columns (x0 - 4 .. x0 + 35:
    static constexpr int MP_LOADS = (MP_FLOATS + 63) / 64;                            // 16-byte aligned rows -> two 16-byte loads per lane; 8 4-byte loads at the image's edges)
    static constexpr int MP_V4 = MP_FLOATS / 4;                                      // 120 float4
    static constexpr int OSTAGE = 2 * 16 * 64;                                       // a consumer wave's pooled output: 2 rows x 16 px x 32 couts bf16
    static constexpr int MP_BYTES = (MP_FLOATS + 4) * 4;                               // + a 16-byte dump slot for the lanes past the real elements
    static constexpr int OFF_W = 0, OFF_IMG = 3 * WCH, OFF_MP = OFF_IMG + 2 * TILE, OFF_OS = (OFF_MP + 4 * MP_BYTES + 15) / 16 * 16, OFF_B = OFF_OS + 4 * OSTAGE;
    static constexpr int LDS = OFF_B + 512;
    static_assert(NGRP * 32 * 32 <= REGION + 512 && LDS <= 160 * 1024, "LDS");
};

__global__ void __launch_bounds__(512, 1)
conv1_pair_pc_bf16_kernel(const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1, const uint16_t *__restrict__ w2p,
                          const float *__restrict__ b2, uint16_t *__restrict__ y, int Cin, int H, int W, int xtiles, int ntiles, int prio) {
    using S = Pair2;
    constexpr int RW = S::RW, TR = S::TR, HPX = S::HPX, NPIX = S::NPIX, NGRP = S::NGRP, REGION = S::REGION;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[S::LDS];
    float *const sb1 = reinterpret_cast<float *>(lds + S::OFF_B), *const sb2 = sb1 + 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int K = Cin * 9, OH = (H + 1) / 2, OW = (W + 1) / 2;
    // ---- once per workgroup: biases, and chunks 1..3 of conv1_2's packed weights into LDS in the swizzled image the fragment reads expect
    //      (16-byte slot s = (row P = s >> 1, half (s & 1) ^ ((P >> 3) & 1)), as conv_bf16.hip's DMA pieces land)
    if (tid < 64) { sb1[tid] = b1[tid]; sb2[tid] = b2[tid]; }
    {
        const frcnn_buf_t wbuf = frcnn_make_buf(w2p, 4u * (uint32_t)S::WCH);
        for (int sidx = tid; sidx < 3 * S::WCH / 16; sidx += 512) {
            const int c = sidx / (S::WCH / 16), sl = sidx - c * (S::WCH / 16);
            const int P = sl >> 1, half = (sl & 1) ^ ((P >> 3) & 1);
            const float4 v = frcnn_buf_load_f32x4(wbuf, (uint32_t)((c + 1) * S::WCH + P * 32 + half * 16));
            *reinterpret_cast<float4 *>(lds + S::OFF_W + c * S::WCH + sl * 16) = v;
        }
    }
    const int n_my = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;   // tiles blockIdx.x + k * gridDim.x

    if (wave >= 4) {
        // =============================================================== producers: conv1_1 units of tile k + 1 while the consumers multiply tile k
        const int pw = wave - 4;
        if (prio == 2) __builtin_amdgcn_s_setprio(1);                  // the producers first (FRCNN_BF16_PAIR_PRIO: 0 none, 1 consumers, 2 producers)
        float *const mp = reinterpret_cast<float *>(lds + S::OFF_MP + pw * S::MP_BYTES);
        const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)Cin * H * W * 4));
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
        int koff[2][8];                                                // patch offsets of this lane's 16 k-values relative to (local halo row, halo column)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * s2 + 8 * khalf + e, kk = k < K ? k : 0;
                const int ci = kk / 9, tap = kk - ci * 9, ky = tap / 3, kx = tap - ky * 3;
                koff[s2][e] = (ci * S::MP_ROWS + ky) * S::MP_PITCH + kx + 2;            // (the patch starts two columns left of the halo's own)
            }
        // unit u of this wave = (tile k = u / UPT, group gi = pw + 4 * (u % UPT)); its fp32 patch: image rows y0 - 2 + hr0 .. + 3 (hr0 = first halo row of
        // the group), columns x0 - 2 .. x0 + 33, every channel -- 432 values, 7 loads per lane, fetched one unit ahead.  The kernel is bound by the
        // instructions its two waves per SIMD issue (stamps: ~13 clocks per producer instruction next to the MFMA stream), so everything that does not
        // depend on the unit is computed ONCE: element e = lane + 64 j -> (channel, patch row, patch column) -> a global byte offset relative to the
        // patch origin and an LDS byte offset; a unit whose patch lies inside the image adds one SCALAR offset in the load instruction itself.
        constexpr int UPT = (NGRP + 3) / 4;                            // unit slots per tile and wave (the last may be empty)
        // a unit's patch is fetched a whole TILE ahead (one unit ahead -- ~2 k clocks -- did not cover an L2 miss: 13 us of the launch were that wait):
        // slot q of `pv` holds the patch of the wave's q-th unit of the tile under production and is refilled for the next tile as soon as it is stored
        float pv[UPT][S::MP_LOADS];                                    // (fast path: pv[q][0..7] = two float4)
        uint32_t g_off[S::MP_LOADS], l_off[S::MP_LOADS], rc[S::MP_LOADS], g4_off[2], l4_off[2];
#pragma unroll
        for (int j = 0; j < S::MP_LOADS; ++j) {
            const int e = lane + 64 * j;
            const int ci = e / (S::MP_ROWS * S::MP_PITCH), rem = e - ci * (S::MP_ROWS * S::MP_PITCH), pr = rem / S::MP_PITCH, pc = rem - pr * S::MP_PITCH;
            const bool real = e < S::MP_FLOATS && ci < Cin;
            g_off[j] = real ? (uint32_t)((ci * H + pr) * W + pc) * 4u : kBufOob;
            l_off[j] = e < S::MP_FLOATS ? (uint32_t)e * 4u : (uint32_t)S::MP_FLOATS * 4u;          // (a dump slot past the patch)
            rc[j] = (uint32_t)pr | ((uint32_t)pc << 8);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                  // the 16-byte form: float4 e = (channel, row, 4-column group)
            const int e = lane + 64 * j;
            const int ci = e / (S::MP_ROWS * 10), rem = e - ci * (S::MP_ROWS * 10), pr = rem / 10, c4 = rem - pr * 10;
            g4_off[j] = (e < S::MP_V4 && ci < Cin) ? (uint32_t)((ci * H + pr) * W + 4 * c4) * 4u : kBufOob;
            l4_off[j] = e < S::MP_V4 ? (uint32_t)e * 16u : (uint32_t)S::MP_FLOATS * 4u;
        }
        const bool rows_aligned = (W & 3) == 0;
        auto unit_exists = [&](int u) { return u / UPT < n_my && pw + 4 * (u % UPT) < NGRP; };
        // tile coordinates without a division per use (a scalar integer division is ~40 instructions through the VALU's reciprocal and back: four of
        // them per unit were a quarter of a unit's time): the workgroup's tiles advance by (gridDim.x / xtiles, gridDim.x % xtiles), carried for the
        // tile under production (k_cur) and the one after it
        const int step_y = (int)gridDim.x / xtiles, step_x = (int)gridDim.x - step_y * xtiles;
        int k_cur = 0, cy = (int)blockIdx.x / xtiles, cx = (int)blockIdx.x - cy * xtiles;
        auto tile_xy = [&](int k, int &tx, int &ty) {                   // k == k_cur or k_cur + 1
            tx = cx; ty = cy;
            if (k != k_cur) { tx += step_x; ty += step_y; if (tx >= xtiles) { tx -= xtiles; ++ty; } }
        };
        auto unit_origin = [&](int u, int &gy0, int &gx0) {
            const int gi = pw + 4 * (u % UPT);
            int tx, ty;
            tile_xy(u / UPT, tx, ty);
            gy0 = ty * TR - 2 + (gi * 32) / HPX;
            gx0 = tx * 32 - 4;
        };
        auto unit_fast = [&](int u) {                                  // (wave-uniform) the whole patch inside the image, rows 16-byte aligned
            int gy0, gx0;
            unit_origin(u, gy0, gx0);
            return rows_aligned && gy0 >= 0 && gy0 + S::MP_ROWS <= H && gx0 >= 0 && gx0 + S::MP_PITCH <= W;
        };
        auto load_unit = [&](auto fast_tag, auto q_tag, int u) {
            constexpr int q = decltype(q_tag)::value;
            int gy0, gx0;
            unit_origin(u, gy0, gx0);
            if constexpr (decltype(fast_tag)::value) {
                const uint32_t origin = (uint32_t)(gy0 * W + gx0) * 4u;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 v = frcnn_buf_load_f32x4_soff(xbuf, g4_off[j], origin);
                    pv[q][4 * j] = v.x; pv[q][4 * j + 1] = v.y; pv[q][4 * j + 2] = v.z; pv[q][4 * j + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < S::MP_LOADS; ++j) {
                    const int gy = gy0 + (int)(rc[j] & 255u), gx = gx0 + (int)(rc[j] >> 8);
                    const bool ok = g_off[j] != kBufOob && gy >= 0 && gy < H && gx >= 0 && gx < W;
                    pv[q][j] = frcnn_buf_load_f32(xbuf, ok ? g_off[j] + (uint32_t)((gy0 * W + gx0) * 4) : kBufOob);
                }
            }
        };
        auto store_unit_patch = [&](auto fast_tag, auto q_tag) {
            constexpr int q = decltype(q_tag)::value;
            unsigned char *const mpb = reinterpret_cast<unsigned char *>(mp);
            if constexpr (decltype(fast_tag)::value) {
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<float4 *>(mpb + l4_off[j]) = make_float4(pv[q][4 * j], pv[q][4 * j + 1], pv[q][4 * j + 2], pv[q][4 * j + 3]);
            } else {
#pragma unroll
                for (int j = 0; j < S::MP_LOADS; ++j) *reinterpret_cast<float *>(mpb + l_off[j]) = pv[q][j];
            }
        };
        // patch in `mp` -> 32 halo pixels x 64 channels of tile k's image (buffer k & 1).  FAST (wave-uniform, decided by the caller): every pixel of the
        // unit lies inside the image -- no per-value select, unconditional stores, straight-line code (the first version's selects and guarded stores
        // became ~40 exec-mask branches per unit)
        auto run_unit = [&](auto fast_tag, int u) {
            constexpr bool FAST = decltype(fast_tag)::value;
            const int k = u / UPT, gi = pw + 4 * (u % UPT);
            int tx, ty;
            tile_xy(k, tx, ty);
            const int x0 = tx * 32, y0 = ty * TR;
            unsigned char *img = lds + S::OFF_IMG + (k & 1) * S::TILE;
            const int P = gi * 32 + l31, Pc = P < NPIX ? P : NPIX - 1;
            const int hr = Pc / HPX, hx = Pc - hr * HPX;
            const int base = (hr - (gi * 32) / HPX) * S::MP_PITCH + hx;
            uint4 bq[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = mp[base + koff[s2][e]];
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
            // conv1_2 pads with zeros: a halo pixel outside the image is 0 (an AND mask on the packed pair), and pixels past the tile's 340 go to a
            // dump slot (the last 8 bytes of the chunk image, which no fragment read touches) instead of behind a branch
            const int gy = y0 - 1 + hr, gx = x0 - 1 + hx;
            const uint32_t keep = FAST ? 0xffffffffu : ((P < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W) ? 0xffffffffu : 0u);
            // the two 16-byte halves of a pixel's 32-byte chunk record, swizzled: quads with (g & 1) == 0 go to half sw, the others to half 1 - sw;
            // lanes past the tile's 340 pixels share the spare record 340 (bytes 10880 .. 10911 of the 11008-byte chunk image: never read)
            const int Pw = (FAST || P < NPIX) ? P : NPIX, sw = (Pw >> 3) & 1;
            unsigned char *const dst0 = img + Pw * 32 + 8 * khalf + (sw << 4), *const dst1 = img + Pw * 32 + 8 * khalf + ((sw ^ 1) << 4);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb * 32 + 8 * g + 4 * khalf;
                    const float4 bb = *reinterpret_cast<const float4 *>(&sb1[co]);
                    const float v0 = frcnn_max_f32(c1[cb][4 * g] + bb.x, 0.0f), v1 = frcnn_max_f32(c1[cb][4 * g + 1] + bb.y, 0.0f);
                    const float v2 = frcnn_max_f32(c1[cb][4 * g + 2] + bb.z, 0.0f), v3 = frcnn_max_f32(c1[cb][4 * g + 3] + bb.w, 0.0f);
                    *reinterpret_cast<uint2 *>(((g & 1) ? dst1 : dst0) + (cb * 2 + (g >> 1)) * REGION) = make_uint2(frcnn_pack_bf16x2(v0, v1) & keep, frcnn_pack_bf16x2(v2, v3) & keep);
                }
        };
        auto unit_is_inside = [&](int u) {                             // (wave-uniform) all 32 pixels of the unit inside the image
            const int k = u / UPT, gi = pw + 4 * (u % UPT);
            int tx, ty;
            tile_xy(k, tx, ty);
            const int x0 = tx * 32, y0 = ty * TR;
            return gi * 32 + 31 < NPIX && y0 >= 1 && y0 + TR + 1 <= H && x0 >= 1 && x0 + 33 <= W;
        };
        // the unit stream: units of tile 0 before the first barrier, units of tile k + 1 between barrier k and barrier k + 1
        // unit slot q of tile T: u = T * UPT + q.  Before the first barrier: the patches of tile 0.  Iteration k (production of tile T = k + 1): slot q's
        // patch goes to LDS, the slot is refilled for tile T + 1, the unit runs.
        bool fastq[UPT];
        auto for_each_slot = [&](auto fn) { fn(std::integral_constant<int, 0>{}); fn(std::integral_constant<int, 1>{}); fn(std::integral_constant<int, 2>{}); };
        static_assert(UPT == 3, "for_each_slot");
        for_each_slot([&](auto qt) {
            constexpr int q = decltype(qt)::value;
            fastq[q] = false;
            if (unit_exists(q)) {
                fastq[q] = unit_fast(q);
                if (fastq[q]) load_unit(std::true_type{}, qt, q);
                else load_unit(std::false_type{}, qt, q);
            }
        });
        frcnn_barrier_nofence();                                       // (the consumers' weight copy; sb1 / sb2)
#pragma unroll 1
        for (int k = -1; k < n_my; ++k) {
            const int it = k + 1;
            (void)it;
            PAIR_STAMP(0);
            if (k + 1 < n_my) {
                for_each_slot([&](auto qt) {
                    constexpr int q = decltype(qt)::value;
                    const int u = (k + 1) * UPT + q;
                    if (unit_exists(u)) {                              // (wave-uniform)
                        __builtin_amdgcn_wave_barrier();               // (wave-private LDS: DS operations of one wave are in order -- a scheduling fence only)
                        if (!(FRCNN_PAIR_ABL & 2)) {
                            if (fastq[q]) store_unit_patch(std::true_type{}, qt);
                            else store_unit_patch(std::false_type{}, qt);
                        }
                        __builtin_amdgcn_wave_barrier();
                        const int un = u + UPT;                        // the same slot of the next tile
                        if (unit_exists(un) && !(FRCNN_PAIR_ABL & 1)) {
                            fastq[q] = unit_fast(un);
                            if (fastq[q]) load_unit(std::true_type{}, qt, un);
                            else load_unit(std::false_type{}, qt, un);
                        }
                        if (!(FRCNN_PAIR_ABL & 4)) {
                            if (unit_is_inside(u)) run_unit(std::true_type{}, u);
                            else run_unit(std::false_type{}, u);
                        }
                    }
                });
                ++k_cur;                                               // tile k + 1 is done: the carried coordinates move on
                cx += step_x; cy += step_y;
                if (cx >= xtiles) { cx -= xtiles; ++cy; }
            }
            PAIR_STAMP(1);
            frcnn_barrier_nofence();                                   // barrier k + 1: tile k + 1 complete (and, for k >= 0, tile k consumed)
            PAIR_STAMP(2);
        }
        return;
    }

    // =================================================================== consumers: conv1_2 on tile k, epilogue of tile k after the barrier
    const int rg = wave & 1, cw = wave >> 1;
    if (prio == 1) __builtin_amdgcn_s_setprio(1);                      // the MFMA stream first
    const frcnn_buf_t ybuf = frcnn_make_buf(y, (uint32_t)((size_t)64 * OH * OW * 2));
    uint4 wa0[9];                                                      // chunk 0 of this wave's 32 couts, in registers
    {
        const frcnn_buf_t wbuf = frcnn_make_buf(w2p, 4u * (uint32_t)S::WCH);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 v = frcnn_buf_load_f32x4(wbuf, (uint32_t)((t * 64 + cw * 32 + l31) * 32 + khalf * 16));
            wa0[t] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
        }
    }
    const uint32_t a_off = (uint32_t)((cw * 32 + l31) * 32 + ((khalf ^ ((l31 >> 3) & 1)) << 4));       // + (chunk - 1) * WCH + tap * 2048
    uint32_t b_off[RW + 2][3];
#pragma unroll
    for (int r = 0; r < RW + 2; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = (rg * RW + r) * HPX + l31 + kx;
            b_off[r][kx] = (uint32_t)(P * 32 + ((khalf ^ ((P >> 3) & 1)) << 4));
        }
    unsigned char *const ost = lds + S::OFF_OS + wave * S::OSTAGE;
    frcnn_barrier_nofence();                                           // weights in LDS (this wave's part written above), biases
    frcnn_barrier_nofence();                                           // barrier 0: tile 0 complete
    const int cstep_y = (int)gridDim.x / xtiles, cstep_x = (int)gridDim.x - cstep_y * xtiles;
    int ty = (int)blockIdx.x / xtiles, tx = (int)blockIdx.x - ty * xtiles;
#pragma unroll 1
    for (int k = 0; k < n_my; ++k) {
        const int it = k;
        (void)it;
        PAIR_STAMP(0);
        if (k > 0) { tx += cstep_x; ty += cstep_y; if (tx >= xtiles) { tx -= xtiles; ++ty; } }
        const int x0 = tx * 32, y0 = ty * TR;
        const unsigned char *img = lds + S::OFF_IMG + (k & 1) * S::TILE;
        frcnn_f32x16 acc[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        uint4 fb[RW + 2][3], fa[3];
        auto read_group = [&](int c, int g) {
            const unsigned char *st = img + c * REGION;
            const int ky = g / 3, kx = g - ky * 3;
            if (c > 0) fa[g % 3] = *reinterpret_cast<const uint4 *>(lds + S::OFF_W + (c - 1) * S::WCH + g * 2048 + a_off);
            if (ky == 0) {
#pragma unroll
                for (int r = 0; r < RW; ++r) fb[r][kx] = *reinterpret_cast<const uint4 *>(st + b_off[r][kx]);
            } else fb[ky + RW - 1][kx] = *reinterpret_cast<const uint4 *>(st + b_off[ky + RW - 1][kx]);
        };
        if (!(FRCNN_PAIR_ABL & 16)) {
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
                for (int j = 0; j < RW; ++j) acc[j] = frcnn_mfma_32x32x16_bf16(c == 0 ? wa0[g] : fa[g % 3], fb[ky + j][kx], acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PAIR_STAMP(1);
        frcnn_barrier_nofence();                                       // barrier k + 1: this tile's image may be overwritten, the next one is complete
        PAIR_STAMP(2);
        if (FRCNN_PAIR_ABL & 8) continue;

        // ---- epilogue (under the producers' next units): 2x2 ceil-mode max of the fp32 sums, + bias, ReLU, one rounding; register r of lane l = cout
        //      cw * 32 + (r & 3) + 8 (r >> 2) + 4 khalf of pixel l31.  Even lanes deposit the quads g = 0, 1 of their pooled pixel, odd lanes g = 2, 3.
        //      INTERIOR (wave-uniform, ONE branch around the whole body -- tested per value it became a scalar branch per value): every 2x2 window is whole.
        const int px = x0 + l31, odd = l31 & 1;
        const uint32_t oddm = 0u - (uint32_t)odd;
        auto pool_rows = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            const bool own_ok = px < W, other_ok = (px ^ 1) < W;
#pragma unroll
            for (int m = 0; m < RW / 2; ++m) {
                const bool row2 = y0 + rg * RW + 2 * m + 1 < H;
                uint2 pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bb = *reinterpret_cast<const float4 *>(&sb2[cw * 32 + 8 * g + 4 * khalf]);
                    const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float a = acc[2 * m][4 * g + t], b = acc[2 * m + 1][4 * g + t];
                        float vm;
                        if constexpr (INTERIOR) vm = frcnn_max_lane_xor1_f32(frcnn_max_f32(a, b));
                        else {
                            const float vr = row2 ? frcnn_max_f32(a, b) : a;
                            const float vo = frcnn_lane_xor1_f32(vr);
                            vm = !own_ok ? vo : (!other_ok ? vr : frcnn_max_f32(vr, vo));
                        }
                        v[t] = frcnn_max_f32(vm + bv[t], 0.0f);
                    }
                    pk[g] = make_uint2(frcnn_pack_bf16x2(v[0], v[1]), frcnn_pack_bf16x2(v[2], v[3]));
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint2 val = make_uint2((pk[2 + h].x & oddm) | (pk[h].x & ~oddm), (pk[2 + h].y & oddm) | (pk[h].y & ~oddm));     // v_bfi (a ?: on the array went through scratch)
                    const int col = 8 * (2 * odd + h) + 4 * khalf;     // cout within the wave's 32
                    *reinterpret_cast<uint2 *>(ost + (m * 16 + (l31 >> 1)) * 64 + col * 2) = val;
                }
            }
        };
        if (y0 + TR <= H && x0 + 32 <= W) pool_rows(std::true_type{});
        else pool_rows(std::false_type{});
        __builtin_amdgcn_wave_barrier();                               // the wave's own LDS writes, read by other lanes below (in-order DS: no wait needed)
        // the wave's 2 pooled rows x 16 px x 32 couts leave as 16-byte pieces: piece q = (16-cout block, row, pixel, half) -- 32 consecutive lanes
        // cover 16 px x 32 B = one contiguous run of the channel-blocked output
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = lane + 64 * j;
            const int half = q & 1, opx = (q >> 1) & 15, m = (q >> 5) & 1, cb16 = q >> 6;
            const uint4 val = *reinterpret_cast<const uint4 *>(ost + (m * 16 + opx) * 64 + cb16 * 32 + half * 16);
            const int oy = (y0 + rg * RW + 2 * m) >> 1, ox = (x0 >> 1) + opx;
            const bool ok = oy < OH && ox < OW;
            frcnn_buf_store_b128(ybuf, ok ? (uint32_t)((((cw * 2 + cb16) * OH + oy) * OW + ox) * 32 + half * 16) : kBufOob, val);
        }
        __builtin_amdgcn_wave_barrier();                               // the staging tile is rewritten by the next tile only after these reads
        PAIR_STAMP(3);
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
    const int xtiles = frcnn_cdiv(W, 32);
    const int cus = frcnn_cu_count() > 0 ? frcnn_cu_count() : 256;
    // FRCNN_BF16_PAIR_FORM=1: the one-wave-per-SIMD form (weights in registers; FRCNN_BF16_PAIR_RW = 4 | 6 rows per wave) -- kept for A/B measurements
    const char *fe = frcnn_tune("FRCNN_BF16_PAIR_FORM");
    if (!fe || atoi(fe) != 1) {
        const int ntiles = xtiles * frcnn_cdiv(H, Pair2::TR);
        // which half of a SIMD's wave pair issues first when both are ready (s_setprio): measured on the MI355X (profiles/r04_conv_pair_micro.txt) the
        // PRODUCERS first is the faster arrangement -- the MFMA stream loses ~8 % of its own pace and the tile ~15 % of its wait for the next image
        const char *pe = frcnn_tune("FRCNN_BF16_PAIR_PRIO");
        const int prio = pe ? atoi(pe) : 2;
        hipLaunchKernelGGL(conv1_pair_pc_bf16_kernel, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(512), 0, (hipStream_t)stream, x, w1, b1, w2_packed, b2, y,
                           Cin, H, W, xtiles, ntiles, prio);
        return frcnn_launch_status();
    }
#ifndef FRCNN_TUNING_FORMS
    return FRCNN_ERR_INVALID;                                       // form 1 (70 us = no gain): research builds only
#else
    const char *e = frcnn_tune("FRCNN_BF16_PAIR_RW");
    const int rw = e ? atoi(e) : 6;
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
#endif
}

}  // extern "C"
