// conv_f32s.hip -- fp32 3x3 convolution on the bf16 matrix cores: every fp32 operand is carried as THREE bf16 terms
// (x = h + m + l exactly: 8 + 8 + 8 mantissa bits) and a product block is six v_mfma_f32_32x32x16_bf16
//     h.h + h.m + m.h + h.l + l.h + m.m            (the dropped terms m.l, l.m, l.l are below 2^-24 of the product)
// accumulated in fp32 -- the same 2^-24-per-product accuracy class as v_mfma_f32_32x32x2_f32 (measured against a float64
// convolution the two differ from it by the same few 1e-7, tests/test_gpu_parity.py), at 16 / 6 = 2.7x the matrix-core rate
// of the native fp32 instruction.  Same reference interface as conv.hip (L.Convolution2D(ci, co, 3, 1, 1) + F.relu
// [+ F.max_pooling_2d(2, 2)], /root/reference/models/vgg16.py:39-82; rpn_conv_3x3, region_proposal_network.py:53).
//
// Layout ("split tensors").  Activations [3 parts][C/16][H][W][16] bf16 -- each part is exactly the channel-blocked tensor
// of conv_bf16.hip -- and weights [3 parts][C/16][tap][CoutP][16] bf16: 6 bytes per fp32 value.  A layer's epilogue splits its
// fp32 result (after bias / ReLU / the fused 2x2 max-pool, all in fp32) into the three parts, so the chain never round-trips
// through fp32 tensors; h + m + l reproduces the fp32 value bit for bit (frcnn_f32s_to_nchw_f32).
//
// Kernel = the LDS-DMA structure of conv_dma_bf16_kernel (tile 64 couts x 4 rows x 32 px, 4 waves = 2 cout blocks x 2 row
// pairs, buffer_load_dwordx4 ... lds, XOR-swizzled 32-byte rows) with a 74 KB stage per 16-channel K-chunk: the three parts of the
// 6 x 34 halo (3 x 208 rows) and of the 9 x 64 weight panel (3 x 576 rows).  Per chunk a wave reads 81 fragments (ds_read_b128)
// for 108 MFMAs -- 0.75 LDS reads and 0.7 KB of DMA per MFMA against 1.17 and 1.4 KB in the plain bf16 kernel, which is what
// lets the matrix pipe run instead of waiting on staging.  Two workgroups per CU (single-stage rings: one computes while the
// other's chunk lands).
#include "frcnn_common.h"
#include <stdlib.h>
#include <string.h>
#include <frcnn_buffer.h>   // angle brackets: shadowed by the test emulator
#include <frcnn_intrin.h>
#include <frcnn_sync.h>
#include "frcnn_reduce.h"

namespace {

constexpr int kCK = 16;                 // channels per K-chunk = the MFMA's k extent
constexpr int kParts = 3;

// (v0, v1) -> the three packed bf16 pairs (h, m, l) with h + m + l == v exactly: frcnn_split3_pair (frcnn_intrin.h)
__device__ __forceinline__ void split3_pair(float v0, float v1, uint32_t &h, uint32_t &m, uint32_t &l) { frcnn_split3_pair(v0, v1, h, m, l); }

// ABL = timing ablations (WRONG results; scripts/conv_f32s_bench.py only): 1 no DMA, 4 no fragment reads / MFMAs
template <int WPS, int ABL = 0, int NS = 1>
__global__ void __launch_bounds__(256, WPS)
conv_f32s_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wp, const float *__restrict__ bias, void *__restrict__ y,
                 int CinP, int Cout, int CoutP, int H, int W, int relu, int out_mode, int xtiles, int ytiles, int nsplit,
                 float *__restrict__ partial_ws, int *__restrict__ tile_counters, int xcd_cotiles, float *__restrict__ y_nchw,
                 const float *__restrict__ mask) {
    constexpr int KS = 3, TAPS = 9, PAD = 1;
    constexpr int RW = 2, BROWS = 4, BCO = 64;
    constexpr int HR = BROWS + KS - 1, HPX = 32 + KS - 1;
    constexpr int IN_ROWS = HR * HPX;                         // 204 halo pixels per part, 32 B each
    constexpr int IN_ROWS_P = 208;                            // padded to a multiple of 16 rows: the swizzle phase is the same in every part
    constexpr int W_ROWS = TAPS * BCO;                        // 576 weight rows per part
    constexpr int IN_PIECES = (kParts * IN_ROWS_P * 2 + 63) / 64;   // 1 KB pieces (64 lanes x 16 B): 20
    constexpr int W_PIECES = kParts * W_ROWS * 2 / 64;        // 54
    constexpr int PIECES = IN_PIECES + W_PIECES;              // 74
    constexpr int IN_BYTES = IN_PIECES * 1024, STAGE_BYTES = PIECES * 1024;
    constexpr int PPW = (PIECES + 3) / 4;                     // pieces per wave (wave w moves pieces w, w+4, ...)
    static_assert(IN_PIECES % 4 == 0, "a group of four pieces comes from one tensor");
    constexpr int OP = BCO * 2 + 16;                          // epilogue tile: LDS bytes per pixel and part (128 B + pad)
    static_assert(kParts * BROWS * 32 * OP <= STAGE_BYTES, "epilogue tiles must fit in the stage");
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NS * STAGE_BYTES];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave & 1, wrow = wave >> 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    // split-K for launches with fewer tiles than the chip has room for (the 38x63 maps: 160 tiles, 512 workgroup slots): `nsplit`
    // consecutive workgroups share a tile, each takes a contiguous range of the K-chunks, the last to finish sums the pieces in
    // split order (deterministic) and runs the epilogue
    int tile = blockIdx.x / nsplit;
    int split = blockIdx.x - tile * nsplit;
    if (xcd_cotiles > 0) {
        // XCD-aware order (launches with 1, 2, 4 or 8 cout tiles): workgroups are dealt to the 8 XCDs round-robin, so
        // XCD x works on cout tile x % cotiles only -- its L2 holds ONE 64-cout weight slab (<= 1.8 MB) instead of all of them
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
    const uint32_t x_part_bytes = (uint32_t)((size_t)H * W * CinP * 2), w_part_bytes = (uint32_t)((size_t)TAPS * CoutP * CinP * 2);
    const frcnn_buf_t xbuf = frcnn_make_buf(x, kParts * x_part_bytes);
    const frcnn_buf_t wbuf = frcnn_make_buf(wp, kParts * w_part_bytes);
    const uint32_t x_chunk_bytes = (uint32_t)(H * W) * 32u, w_chunk_bytes = (uint32_t)(TAPS * CoutP) * 32u;

    // source offset (chunk 0) of the 16 bytes this lane contributes to each of its wave's pieces: slot s of a region holds
    // (row R = s >> 1, half (s & 1) ^ ((R >> 3) & 1)); input rows R = part * 208 + halo pixel, weight rows R = part * 576 + tap * 64 + cout
    uint32_t poff[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pid = wave + 4 * q;
        if (pid < IN_PIECES) {
            const int sl = pid * 64 + lane, R = sl >> 1, half = (sl & 1) ^ ((R >> 3) & 1);
            const int part = R / IN_ROWS_P, P = R - part * IN_ROWS_P;
            const int hr = P / HPX, hx = P - hr * HPX;
            const int gy = y0 - PAD + hr, gx = x0 - PAD + hx;
            const bool inside = part < kParts && P < IN_ROWS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            poff[q] = inside ? (uint32_t)part * x_part_bytes + (uint32_t)((gy * W + gx) * 32 + half * 16) : kBufOob;
        } else {
            const int sl = (pid - IN_PIECES) * 64 + lane, R = sl >> 1, half = (sl & 1) ^ ((R >> 3) & 1);
            const int part = R / W_ROWS, r = R - part * W_ROWS;
            const int tap = r / BCO, col = r - tap * BCO;
            poff[q] = (pid < PIECES && co0 + col < CoutP) ? (uint32_t)part * w_part_bytes + (uint32_t)((tap * CoutP + co0 + col) * 32 + half * 16) : kBufOob;
        }
    }
    auto issue = [&](int chunk, int stage) {
        if constexpr ((ABL & 1) != 0) return;
        unsigned char *dst = ring + stage * STAGE_BYTES + wave * 1024;
        const uint32_t xs = (uint32_t)(c_first + chunk) * x_chunk_bytes, ws = (uint32_t)(c_first + chunk) * w_chunk_bytes;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            if (4 * q + 3 < IN_PIECES) frcnn_buf_load_lds_b128(xbuf, dst + q * 4096, poff[q], xs);
            else if (4 * q + 3 < PIECES || wave + 4 * q < PIECES) frcnn_buf_load_lds_b128(wbuf, dst + q * 4096, poff[q], ws);
        }
    };

    // fragment byte offsets inside the stage (swizzled).  A = weight row part*576 + tap*64 + wco*32 + l31 (the part / tap offsets are
    // multiples of 16 rows: compile-time immediates), B = halo row part*208 + (2 wrow + r)*34 + l31 + kx
    const uint32_t a_off = (uint32_t)(IN_BYTES + (wco * 32 + l31) * 32 + ((khalf ^ ((l31 >> 3) & 1)) << 4));
    uint32_t b_off[RW + KS - 1][KS];
#pragma unroll
    for (int r = 0; r < RW + KS - 1; ++r)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int P = (wrow * RW + r) * HPX + l31 + kx;
            b_off[r][kx] = (uint32_t)(P * 32 + ((khalf ^ ((P >> 3) & 1)) << 4));
        }

    // two accumulator sets per output row: the large terms (h.h, h.m, m.h) and the small ones (h.l, l.h, m.m) -- four independent
    // MFMA chains per wave, and the 2^-16-sized terms are summed among themselves before they meet the large sum
    frcnn_f32x16 acc[RW], acs[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = acs[j][r] = 0.0f;

    auto compute = [&](int stage) {
        if constexpr ((ABL & 4) != 0) return;
        const unsigned char *st = ring + stage * STAGE_BYTES;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            uint4 a[kParts][KS], b[kParts][RW][KS];
#pragma unroll
            for (int p = 0; p < kParts; ++p)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    if constexpr ((ABL & 8) != 0) {                   // no fragment reads: the MFMA ceiling of this loop
                        a[p][kx] = make_uint4(lane + p, kx, ky, stage);
#pragma unroll
                        for (int j = 0; j < RW; ++j) b[p][j][kx] = make_uint4(lane + j, kx, p, ky);
                    } else {
                    a[p][kx] = *reinterpret_cast<const uint4 *>(st + a_off + (p * W_ROWS + (ky * KS + kx) * BCO) * 32);
#pragma unroll
                    for (int j = 0; j < RW; ++j) b[p][j][kx] = *reinterpret_cast<const uint4 *>(st + b_off[ky + j][kx] + p * IN_ROWS_P * 32);
                    }
                }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int j = 0; j < RW; ++j) acs[j] = frcnn_mfma_32x32x16_bf16(a[2][kx], b[0][j][kx], acs[j]);     // l.h
#pragma unroll
                for (int j = 0; j < RW; ++j) acc[j] = frcnn_mfma_32x32x16_bf16(a[1][kx], b[0][j][kx], acc[j]);     // m.h
#pragma unroll
                for (int j = 0; j < RW; ++j) acs[j] = frcnn_mfma_32x32x16_bf16(a[0][kx], b[2][j][kx], acs[j]);     // h.l
#pragma unroll
                for (int j = 0; j < RW; ++j) acc[j] = frcnn_mfma_32x32x16_bf16(a[0][kx], b[1][j][kx], acc[j]);     // h.m
#pragma unroll
                for (int j = 0; j < RW; ++j) acs[j] = frcnn_mfma_32x32x16_bf16(a[1][kx], b[1][j][kx], acs[j]);     // m.m
#pragma unroll
                for (int j = 0; j < RW; ++j) acc[j] = frcnn_mfma_32x32x16_bf16(a[0][kx], b[0][j][kx], acc[j]);     // h.h
            }
        }
    };

    // single-stage ring: the co-resident workgroup's MFMAs cover this one's wait for its chunk.  (Starting half of the resident
    // workgroups late, so that pairs cannot run in lockstep, measured no different: r02j.)
    if constexpr (NS == 1) {
    for (int c = 0; c < nchunks; ++c) {
        issue(c, 0);
        frcnn_wait_vmcnt<0>();
        frcnn_barrier_nofence();
        compute(0);
        if (c + 1 < nchunks) frcnn_barrier_nofence();           // everybody is done reading before the stage is refilled
    }
    } else {
        // two stages, ONE workgroup per CU: chunk c+1 lands while chunk c feeds the MFMAs
        issue(0, 0);
        frcnn_wait_vmcnt<0>();
        frcnn_barrier_nofence();
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) issue(c + 1, (c + 1) & 1);
            compute(c & 1);
            if (c + 1 < nchunks) {
                frcnn_wait_vmcnt<0>();                            // chunk c+1 has landed for this wave ...
                frcnn_barrier_nofence();                          // ... and for everybody, and everybody is done reading stage c & 1
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += acs[j][r];
    if (nsplit > 1) {
        // publish this split's accumulators (fragment-linear float4s, write-through: no release fence needed), take a ticket
        const size_t slot_floats = (size_t)256 * RW * 16;
        const frcnn_buf_t pbuf = frcnn_make_buf(partial_ws + ((size_t)tile * nsplit + split) * slot_floats, (uint32_t)(slot_floats * sizeof(float)));
#pragma unroll
        for (int j = 0; j < RW; ++j)
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
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        for (int q = 0; q < nsplit; ++q) {
            const float4 *piece = reinterpret_cast<const float4 *>(partial_ws + ((size_t)tile * nsplit + q) * slot_floats);
            float4 v[RW * 4];
#pragma unroll
            for (int e = 0; e < RW * 4; ++e) v[e] = piece[(size_t)e * 256 + tid];
#pragma unroll
            for (int e = 0; e < RW * 4; ++e) frcnn_pin(v[e]);           // one batch of loads per piece (else: load - wait - add per vector)
#pragma unroll
            for (int j = 0; j < RW; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 t = v[j * 4 + r4];
                    acc[j][4 * r4] += t.x; acc[j][4 * r4 + 1] += t.y; acc[j][4 * r4 + 2] += t.z; acc[j][4 * r4 + 3] += t.w;
                }
        }
    }
    __syncthreads();                                            // the stage becomes the epilogue's output tiles

    // ---- epilogue: register r of lane l = cout (r&3) + 8*(r>>2) + 4*khalf of pixel l31, rows y0 + 2 wrow + j
    float v[RW][16];
    {
        const frcnn_buf_t bbuf = frcnn_make_buf(bias, (uint32_t)Cout * 4u);    // couts past Cout read 0 (range check): one branch-free batch
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = frcnn_buf_load_f32(bbuf, (uint32_t)(co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * 4u);
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = acc[j][r] + bv[r];
                if (relu) t = fmaxf(t, 0.0f);
                v[j][r] = t;
            }
    }
    const int px = x0 + l31;
    if (mask != nullptr || y_nchw != nullptr) {
        // training forms.  mask (Cout,H,W) fp32: y = (mask > 0) ? y : 0 -- the input-gradient convolution of the backward pass with the
        // producing ReLU's mask fused in (conv.hip act = 2).  y_nchw: the result ALSO as fp32 NCHW (the weight-gradient kernel, the
        // bias gradient, the pool and the next layer's mask read fp32; the next convolution reads the split tensor)
        // (buffer loads / stores with 32-bit offsets: all of a row's mask values are fetched in one batch before the first store -- as
        // load - test - store per element every load waited for the previous store's acknowledgement; lanes and couts outside the
        // map get an out-of-range offset: they load 0 and store nothing)
        const uint32_t map_bytes = (uint32_t)((size_t)Cout * H * W * 4);
        const frcnn_buf_t mbuf = frcnn_make_buf(mask, mask != nullptr ? map_bytes : 0u);
        const frcnn_buf_t nbuf = frcnn_make_buf(y_nchw, y_nchw != nullptr ? map_bytes : 0u);
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int py = y0 + wrow * RW + j;
            const bool inside = px < W && py < H;
            uint32_t o[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                o[r] = (inside && co < Cout) ? (uint32_t)((co * H + py) * W + px) * 4u : kBufOob;
            }
            if (mask != nullptr) {
                float mv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) mv[r] = frcnn_buf_load_f32(mbuf, o[r]);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!(mv[r] > 0.0f)) v[j][r] = 0.0f;
            }
            if (y_nchw != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) frcnn_buf_store_f32(nbuf, o[r], v[j][r]);
            }
        }
    }
    if (out_mode == 1) {                                        // fp32 NCHW (the last layer of a chain)
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int py = y0 + wrow * RW + j;
            if (px >= W || py >= H) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co < Cout) reinterpret_cast<float *>(y)[(size_t)co * H * W + (size_t)py * W + px] = v[j][r];
            }
        }
        return;
    }
    uint16_t *y16 = reinterpret_cast<uint16_t *>(y);
    if (out_mode == 2) {
        // F.max_pooling_2d(2, 2) (cover_all) fused, in fp32 BEFORE the split: the wave's two rows are one window row pair (tiles
        // start at even rows), the horizontal neighbour is the next lane; windows cut by the map's edge use what is inside
        const int py = y0 + wrow * RW;
        const bool has_row1 = py + 1 < H, has_col1 = px + 1 < W;
        const int OH = (H + 1) / 2, OW = (W + 1) / 2;
        const size_t y_part = (size_t)CoutP * OH * OW;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float pv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float top = v[0][4 * g + t], col = has_row1 ? fmaxf(top, v[1][4 * g + t]) : top;
                const float nb = __shfl_xor(col, 1);
                pv[t] = has_col1 ? fmaxf(col, nb) : col;
            }
            uint32_t hp[2], mp[2], lp[2];
            split3_pair(pv[0], pv[1], hp[0], mp[0], lp[0]);
            split3_pair(pv[2], pv[3], hp[1], mp[1], lp[1]);
            if ((l31 & 1) == 0) {
                const int colc = wco * 32 + 8 * g + 4 * khalf;                 // first of four consecutive couts (within the tile)
                unsigned char *o = ring + (wrow * 16 + (l31 >> 1)) * OP + colc * 2;
                *reinterpret_cast<uint2 *>(o) = make_uint2(hp[0], hp[1]);
                *reinterpret_cast<uint2 *>(o + (BROWS / 2) * 16 * OP) = make_uint2(mp[0], mp[1]);
                *reinterpret_cast<uint2 *>(o + 2 * (BROWS / 2) * 16 * OP) = make_uint2(lp[0], lp[1]);
            }
        }
        __syncthreads();
        for (int e = tid; e < kParts * (BROWS / 2) * 16 * 8; e += 256) {       // 16-byte vectors: (part, cout block of 16, pooled pixel, half)
            const int part = e / ((BROWS / 2) * 16 * 8), e1 = e - part * ((BROWS / 2) * 16 * 8);
            const int cbl = e1 / ((BROWS / 2) * 16 * 2), rem = e1 - cbl * ((BROWS / 2) * 16 * 2);
            const int opix = rem >> 1, half = rem & 1;
            const int oy = (y0 >> 1) + (opix >> 4), ox = (x0 >> 1) + (opix & 15), co = co0 + cbl * 16;
            if (oy < OH && ox < OW && co < CoutP)
                *reinterpret_cast<uint4 *>(y16 + part * y_part + (((size_t)(co >> 4) * OH + oy) * OW + ox) * 16 + half * 8) =
                    *reinterpret_cast<const uint4 *>(ring + (part * (BROWS / 2) * 16 + opix) * OP + (cbl * 2 + half) * 16);
        }
        return;
    }
    // out_mode 0: split tensor, same size.  Transposed through LDS so that each 16-cout block of a tile row leaves as one
    // contiguous run of 32 px x 32 B per part
    const size_t y_part = (size_t)CoutP * H * W;
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t hp[2], mp[2], lp[2];
            split3_pair(v[j][4 * g], v[j][4 * g + 1], hp[0], mp[0], lp[0]);
            split3_pair(v[j][4 * g + 2], v[j][4 * g + 3], hp[1], mp[1], lp[1]);
            const int colc = wco * 32 + 8 * g + 4 * khalf;
            unsigned char *o = ring + ((wrow * RW + j) * 32 + l31) * OP + colc * 2;
            *reinterpret_cast<uint2 *>(o) = make_uint2(hp[0], hp[1]);
            *reinterpret_cast<uint2 *>(o + BROWS * 32 * OP) = make_uint2(mp[0], mp[1]);
            *reinterpret_cast<uint2 *>(o + 2 * BROWS * 32 * OP) = make_uint2(lp[0], lp[1]);
        }
    __syncthreads();
    for (int e = tid; e < kParts * BROWS * 32 * 8; e += 256) {                 // 16-byte vectors: (part, cout block of 16, pixel, half)
        const int part = e / (BROWS * 32 * 8), e1 = e - part * (BROWS * 32 * 8);
        const int cbl = e1 / (BROWS * 32 * 2), rem = e1 - cbl * (BROWS * 32 * 2);
        const int pix = rem >> 1, half = rem & 1;
        const int py = y0 + (pix >> 5), qx = x0 + (pix & 31), co = co0 + cbl * 16;
        if (py < H && qx < W && co < CoutP)
            *reinterpret_cast<uint4 *>(y16 + part * y_part + (((size_t)(co >> 4) * H + py) * W + qx) * 16 + half * 8) =
                *reinterpret_cast<const uint4 *>(ring + (part * BROWS * 32 + pix) * OP + (cbl * 2 + half) * 16);
    }
}

// First layer (Cin * 9 <= 32: conv1_1, 3 channels): the fp32 NCHW image goes in as it is.  K = (ci, tap) padded to 32 = two MFMA
// k-steps; the weight fragments (64 couts x 32 k, three parts) live in registers for the whole kernel; a workgroup holds the fp32
// halo of a 4-row x 64-px tile in LDS (5 KB), each wave builds the im2col B fragments of a 32-px row segment from it (16 ds_read_b32,
// the 3-way split in registers) -- 24 MFMAs per 32 px x 64 couts -- and the split output leaves through the wave's LDS tile as
// contiguous 1 KB runs.  The generic kernel would stage 74 KB per tile for 3 real channels and run 108 MFMAs per 2 rows: 131 us on
// the 600x1000 image; this one is bound by its 230 MB of output.
// PERSISTENT: the launch holds as many workgroups as the chip seats at once and each strides over the tiles, so the prologue (32
// weight loads per lane, gathered across the (Cout, K) matrix -- issued as ONE batch: clamped index, unconditional load, value
// select; the first version's predicated loads each waited for the previous one, ~20 us per workgroup) is paid once, and the next
// tile's halo (5 loads per thread, same batch form) is in flight while the current tile's units run.
// SPLIT = false: the plain bf16 form of the same kernel (conv_bf16.hip's chain: operands rounded to bf16, one MFMA per k-step, the
// result rounded to bf16 once) -- only the h terms exist.
// F32: native fp32 arithmetic (v_mfma_f32_32x32x2_f32, K = 28 = 14 k-steps, fp32 weights in registers) and ONLY the fp32 NCHW output:
// conv1_1 of the fp32 chain (conv.hip's generic kernel writes that layer's 154 MB with 4-byte stores: 68 us).
// The fp32 NCHW map (DUAL, F32) leaves through a wave-private LDS tile [cout][32 px] as 16-byte stores along the row (eight couts x 128
// bytes per instruction) -- as 4-byte stores per accumulator register the same 154 MB took ~70 us.
template <int NCB, bool SPLIT = true, bool DUAL = false, bool F32 = false>        // cout blocks of 32; DUAL: the fp32 NCHW map as well (training)
__global__ void __launch_bounds__(256, F32 ? 3 : (SPLIT ? 2 : 4))
conv1_f32s_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, uint16_t *__restrict__ y, int Cin,
                  int Cout, int H, int W, int relu, int w_is_packed, float *__restrict__ y_nchw, int xtiles, int ntiles) {
    constexpr int TR = 4, TW = 64, HR = TR + 2, PITCH = TW + 4;        // LDS row: 66 used of 68 floats; tile row = wave
    constexpr int CMAX = 3;
    __shared__ float xt[CMAX * HR * PITCH];
    __shared__ __attribute__((aligned(16))) float sbias[32 * NCB];
    constexpr int NPARTS = SPLIT ? kParts : 1;
    constexpr int OPX = 32 + 16;                                       // output staging: bytes per (pixel, 16-cout block) + pad
    __shared__ __attribute__((aligned(16))) unsigned char ot[F32 ? 1 : 4][F32 ? 1 : 2 * NCB][F32 ? 16 : 32 * OPX];     // per wave: one part of a unit's tile, [cout block of 16][px]
    constexpr bool NCHW = DUAL || F32;
    constexpr int NTP = 36;                                            // fp32 NCHW staging: floats per cout row (32 px + pad)
    __shared__ __attribute__((aligned(16))) float nt[NCHW ? 4 : 1][NCHW ? 32 * NCB : 1][NCHW ? NTP : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int K = Cin * 9;
    if (tid < 32 * NCB) sbias[tid] = tid < Cout ? bias[tid] : 0.0f;
    const frcnn_buf_t wbuf = frcnn_make_buf(w, (uint32_t)(Cout * K) * 4u);
    const int wk = w_is_packed ? Cout : 1, wc = w_is_packed ? 1 : K;   // element strides of (k, cout) in w
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)Cin * H * W * 4));
    // ---- weight fragments: lane (cout l31 of block cb, k = 16 s + 8 khalf + e), k = ci * 9 + tap; the three parts of (Cout, K) fp32
    constexpr int NP = SPLIT ? kParts : 1;
    constexpr int KS32 = 14;                                           // F32: k-steps of two (k = 2 s + khalf < 28)
    float aw[F32 ? NCB : 1][F32 ? KS32 : 1];
    int boff32[F32 ? KS32 : 1];
    if constexpr (F32) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int st = 0; st < KS32; ++st) {
                const int k = 2 * st + khalf, co = cb * 32 + l31;
                aw[cb][st] = frcnn_buf_load_f32(wbuf, (co < Cout && k < K) ? (uint32_t)(k * wk + co * wc) * 4u : kBufOob);
            }
#pragma unroll
        for (int st = 0; st < KS32; ++st) {
            const int k = 2 * st + khalf, kk = k < K ? k : 0;
            const int ci = kk / 9, tap = kk - ci * 9, ky = tap / 3, kx = tap - ky * 3;
            boff32[st] = (ci * HR + wave + ky) * PITCH + kx + l31;
        }
    }
    uint4 a[NCB][2][NP];
    if constexpr (!F32) {
        float wv[NCB][2][8];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * s2 + 8 * khalf + e, co = cb * 32 + l31;
                    // w: Chainer's (Cout, Cin, 3, 3), or (w_is_packed) the trainers' packed [(ci * 9 + tap)][co]; out of range -> 0
                    wv[cb][s2][e] = frcnn_buf_load_f32(wbuf, (co < Cout && k < K) ? (uint32_t)(k * wk + co * wc) * 4u : kBufOob);
                }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint32_t hp[4], mp[4], lp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split3_pair(wv[cb][s2][2 * e], wv[cb][s2][2 * e + 1], hp[e], mp[e], lp[e]);
                a[cb][s2][0] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                if constexpr (SPLIT) {
                    a[cb][s2][1] = make_uint4(mp[0], mp[1], mp[2], mp[3]);
                    a[cb][s2][2] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
                }
            }
    }
    // ---- per-lane LDS offsets of the 16 im2col elements (floats), relative to column 0 of the wave's tile row
    // (k >= K: the weight is zero and the element read instead -- k = 0, the pixel's own first tap -- is a finite image value)
    int boff[2][8];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * s2 + 8 * khalf + e;
            const int kk = k < K ? k : 0;
            const int ci = kk / 9, tap = kk - ci * 9, ky = tap / 3, kx = tap - ky * 3;
            boff[s2][e] = (ci * HR + wave + ky) * PITCH + kx + l31;
        }
    // ---- halo loads: wave w fetches halo rows w, w + 4, ... (row = ci * HR + r, wave-uniform) at column `lane`; the two columns past
    // 64 of all 18 rows are one more load of threads 0..35.  Zero outside the image and for channels >= Cin (out-of-range offset).
    constexpr int HROWS = CMAX * HR, HIT = (HROWS + 3) / 4;
    float hv[HIT + 1];
    auto load_halo = [&](int tile) {
        const int ty = tile / xtiles, tx = tile - ty * xtiles;
        const int gx0 = tx * TW - 1, gy0 = ty * TR - 1;
#pragma unroll
        for (int j = 0; j < HIT; ++j) {
            const int row = wave + 4 * j, ci = row / HR, gy = gy0 + row - ci * HR, gx = gx0 + lane;
            const bool ok = row < HROWS && ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W;
            hv[j] = frcnn_buf_load_f32(xbuf, ok ? (uint32_t)((ci * H + gy) * W + gx) * 4u : kBufOob);
        }
        {
            const int row = tid >> 1, ci = row / HR, gy = gy0 + row - ci * HR, gx = gx0 + TW + (tid & 1);
            const bool ok = row < HROWS && ci < Cin && gy >= 0 && gy < H && gx < W;
            hv[HIT] = frcnn_buf_load_f32(xbuf, ok ? (uint32_t)((ci * H + gy) * W + gx) * 4u : kBufOob);
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int j = 0; j < HIT; ++j)
            if (wave + 4 * j < HROWS) xt[(wave + 4 * j) * PITCH + lane] = hv[j];
        if (tid < 2 * HROWS) xt[(tid >> 1) * PITCH + TW + (tid & 1)] = hv[HIT];
    };
    const int CoutP = (Cout + 15) / 16 * 16;
    const size_t y_part = (size_t)CoutP * H * W;
    const uint32_t plane_bytes = (uint32_t)(H * W) * 32u;              // one 16-cout block of one part
    frcnn_buf_t ybuf[NPARTS];
#pragma unroll
    for (int part = 0; part < NPARTS; ++part) ybuf[part] = frcnn_make_buf(y + part * y_part, (uint32_t)(y_part * 2));
    const frcnn_buf_t nbuf = frcnn_make_buf(y_nchw, NCHW ? (uint32_t)((size_t)Cout * H * W * 4) : 0u);
    int tile = blockIdx.x;
    if (tile < ntiles) load_halo(tile);
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
        const int ty = tile / xtiles, tx = tile - ty * xtiles;
        const int x0 = tx * TW, y0 = ty * TR;
        __syncthreads();                                               // the previous tile's units have read xt
        store_halo();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_halo(tile + gridDim.x);        // in flight under the units below
        const int py = y0 + wave;
        if (py >= H) continue;                                         // wave-uniform (no barrier below in this iteration)
        // ---- units: wave w owns tile row w, two 32-px segments
#pragma unroll 1
        for (int seg = 0; seg < 2; ++seg) {
            if (x0 + seg * 32 >= W) continue;                          // wave-uniform
            uint4 b[2][NP];
            frcnn_f32x16 acc[NCB];                                     // (one accumulator per cout block: 24 MFMAs per unit, registers matter more)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
            if constexpr (F32) {
                float bx[KS32];
#pragma unroll
                for (int st = 0; st < KS32; ++st) bx[st] = xt[boff32[st] + seg * 32];
#pragma unroll
                for (int st = 0; st < KS32; ++st)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[cb][st], bx[st], acc[cb], 0, 0, 0);
            } else {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = xt[boff[s2][e] + seg * 32];
                uint32_t hp[4], mp[4], lp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split3_pair(v[2 * e], v[2 * e + 1], hp[e], mp[e], lp[e]);
                b[s2][0] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                if constexpr (SPLIT) {
                    b[s2][1] = make_uint4(mp[0], mp[1], mp[2], mp[3]);
                    b[s2][2] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if constexpr (SPLIT) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[cb] = frcnn_mfma_32x32x16_bf16(a[cb][s2][NP - 1], b[s2][0], acc[cb]);     // l.h
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[cb] = frcnn_mfma_32x32x16_bf16(a[cb][s2][NP / 2], b[s2][0], acc[cb]);     // m.h
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[cb] = frcnn_mfma_32x32x16_bf16(a[cb][s2][0], b[s2][NP - 1], acc[cb]);     // h.l
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[cb] = frcnn_mfma_32x32x16_bf16(a[cb][s2][0], b[s2][NP / 2], acc[cb]);     // h.m
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[cb] = frcnn_mfma_32x32x16_bf16(a[cb][s2][NP / 2], b[s2][NP / 2], acc[cb]);     // m.m
                }
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb] = frcnn_mfma_32x32x16_bf16(a[cb][s2][0], b[s2][0], acc[cb]);     // h.h
            }
            }
            // the unit's 32 px x (32 NCB) couts go through the wave's LDS tile, one part at a time, so that every 16-cout block leaves as
            // ONE contiguous run of 32 px x 32 B (16-byte stores, consecutive lanes consecutive addresses) instead of 8-byte pieces 32
            // bytes apart
            uint2 pk[NPARTS][NCB][4];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb * 32 + 8 * g + 4 * khalf;        // first of four consecutive couts
                    const float4 bq = *reinterpret_cast<const float4 *>(&sbias[co]);
                    float v[4] = {acc[cb][4 * g] + bq.x, acc[cb][4 * g + 1] + bq.y, acc[cb][4 * g + 2] + bq.z, acc[cb][4 * g + 3] + bq.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (relu) v[t] = fmaxf(v[t], 0.0f);
                        if constexpr (NCHW) nt[wave][co + t][l31] = v[t];      // the fp32 NCHW map (training forms / the fp32 chain)
                    }
                    if constexpr (F32) continue;
                    uint32_t hp[2], mp[2], lp[2];
                    split3_pair(v[0], v[1], hp[0], mp[0], lp[0]);
                    split3_pair(v[2], v[3], hp[1], mp[1], lp[1]);
                    pk[0][cb][g] = make_uint2(hp[0], hp[1]);
                    if constexpr (SPLIT) {
                        pk[NPARTS / 2][cb][g] = make_uint2(mp[0], mp[1]);
                        pk[NPARTS - 1][cb][g] = make_uint2(lp[0], lp[1]);
                    }
                }
            if constexpr (NCHW) {
                __builtin_amdgcn_wave_barrier();
                const int cr = lane >> 3, p4 = (lane & 7) * 4, qx4 = x0 + seg * 32 + p4;
                const bool ragged = x0 + seg * 32 + 32 > W;            // wave-uniform: the segment crosses the map's right edge
#pragma unroll
                for (int i = 0; i < 4 * NCB; ++i) {
                    const int co = 8 * i + cr;
                    const float4 q4 = *reinterpret_cast<const float4 *>(&nt[wave][co][p4]);
                    const uint32_t o = (uint32_t)((co * H + py) * W + qx4) * 4u;
                    frcnn_buf_store_b128(nbuf, (co < Cout && qx4 + 3 < W) ? o : kBufOob, make_uint4(__float_as_uint(q4.x), __float_as_uint(q4.y), __float_as_uint(q4.z), __float_as_uint(q4.w)));
                    if (ragged) {                                      // the partial group of four at the edge, element by element
                        const bool part_grp = co < Cout && !(qx4 + 3 < W);
                        frcnn_buf_store_f32(nbuf, (part_grp && qx4 < W) ? o : kBufOob, q4.x);
                        frcnn_buf_store_f32(nbuf, (part_grp && qx4 + 1 < W) ? o + 4u : kBufOob, q4.y);
                        frcnn_buf_store_f32(nbuf, (part_grp && qx4 + 2 < W) ? o + 8u : kBufOob, q4.z);
                    }
                }
                __builtin_amdgcn_wave_barrier();                       // nt is rewritten by the next unit only after these reads
            }
            if constexpr (F32) continue;
            const int q = lane >> 1, half = lane & 1;                  // pixel, 16-byte half of its 32-byte cout block
            const int qx = x0 + seg * 32 + q;
            const uint32_t yoff = qx < W ? (uint32_t)((py * W + qx) * 32 + half * 16) : kBufOob;       // bytes inside a 16-cout block plane
#pragma unroll
            for (int part = 0; part < NPARTS; ++part) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = cb * 32 + 8 * g + 4 * khalf;
                        *reinterpret_cast<uint2 *>(&ot[wave][co >> 4][l31 * OPX + (co & 15) * 2]) = pk[part][cb][g];
                    }
                __builtin_amdgcn_wave_barrier();                       // the wave's own LDS writes, read by other lanes below
#pragma unroll
                for (int b16 = 0; b16 < 2 * NCB; ++b16) {
                    const uint4 val = *reinterpret_cast<const uint4 *>(&ot[wave][b16][q * OPX + half * 16]);
                    if (b16 * 16 < CoutP) frcnn_buf_store_b128(ybuf[part], yoff + (uint32_t)b16 * plane_bytes, val);
                }
                __builtin_amdgcn_wave_barrier();                       // the tile is rewritten only after these reads were issued
            }
        }
    }
}

// (Cout, Cin, 3, 3) fp32 -> [3 parts][CinP/16][tap][CoutP][16] bf16, zero padded
__global__ void __launch_bounds__(256)
pack_w_f32s_kernel(const float *__restrict__ w, int Cout, int Cin, int taps, int CoutP, int CinP, uint16_t *__restrict__ wp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)taps * CoutP * CinP;
    if (i >= total) return;
    const int c16 = (int)(i % 16), co = (int)((i / 16) % CoutP), tap = (int)((i / (16 * (size_t)CoutP)) % taps);
    const int ci = (int)(i / (16 * (size_t)CoutP * taps)) * 16 + c16;
    const float v = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + tap] : 0.0f;
    uint32_t h, m, l;
    split3_pair(v, 0.0f, h, m, l);
    wp[i] = (uint16_t)h; wp[total + i] = (uint16_t)m; wp[2 * total + i] = (uint16_t)l;
}

// The trainer's PACKED fp32 weights wp[(ci * 9 + tap)][co] (conv.hip's layout, updated in place by the optimizer) -> split weights of
// the forward convolution, or (dgrad) of the input-gradient convolution: channels swapped, taps rotated by 180 degrees --
// w'[ci][co][tap'] = w[co][ci][8 - tap'] (train.hip pack_dgrad_w_kernel's rule)
__global__ void __launch_bounds__(256)
pack_w_f32s_from_packed_kernel(const float *__restrict__ wp, int Cin, int Cout, int dgrad, uint16_t *__restrict__ dst) {
    const int KI = dgrad ? Cout : Cin, KO = dgrad ? Cin : Cout;        // reduction / output channels of the convolution being packed
    const int KIP = (KI + 15) / 16 * 16, KOP = (KO + 15) / 16 * 16;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)9 * KOP * KIP;
    if (i >= total) return;
    const int c16 = (int)(i % 16), o = (int)((i / 16) % KOP), tap = (int)((i / (16 * (size_t)KOP)) % 9);
    const int k = (int)(i / (16 * (size_t)KOP * 9)) * 16 + c16;
    float v = 0.0f;
    if (o < KO && k < KI) v = dgrad ? wp[((size_t)o * 9 + (8 - tap)) * Cout + k] : wp[((size_t)k * 9 + tap) * Cout + o];
    uint32_t h, m, l;
    split3_pair(v, 0.0f, h, m, l);
    dst[i] = (uint16_t)h; dst[total + i] = (uint16_t)m; dst[2 * total + i] = (uint16_t)l;
}

// All layers of a trainer in ONE launch (26 launches of 8 us each otherwise, every step): up to 16 layers, each block looks its layer up
struct PackManyArgs {
    const float *wp[16];
    uint16_t *fwd[16], *dgr[16];
    int cin[16], cout[16];
    unsigned block_end[16];                          // exclusive prefix of 256-element blocks
    int n;
};
__global__ void __launch_bounds__(256)
pack_w_f32s_many_kernel(const PackManyArgs a) {
    int li = 0;
    while (li + 1 < a.n && blockIdx.x >= a.block_end[li]) ++li;
    const unsigned b0 = li == 0 ? 0u : a.block_end[li - 1];
    const int Cin = a.cin[li], Cout = a.cout[li];
    const int CinP = (Cin + 15) / 16 * 16, CoutP = (Cout + 15) / 16 * 16;
    const size_t i = (size_t)(blockIdx.x - b0) * 256 + threadIdx.x;
    const size_t total = (size_t)9 * CoutP * CinP;
    if (i >= total) return;
    const float *wp = a.wp[li];
    {   // forward: [CinP/16][tap][CoutP][16]
        const int c16 = (int)(i % 16), o = (int)((i / 16) % CoutP), tap = (int)((i / (16 * (size_t)CoutP)) % 9);
        const int k = (int)(i / (16 * (size_t)CoutP * 9)) * 16 + c16;
        uint32_t h, m, l;
        split3_pair((o < Cout && k < Cin) ? wp[((size_t)k * 9 + tap) * Cout + o] : 0.0f, 0.0f, h, m, l);
        uint16_t *d = a.fwd[li];
        d[i] = (uint16_t)h; d[total + i] = (uint16_t)m; d[2 * total + i] = (uint16_t)l;
    }
    if (a.dgr[li] != nullptr) {   // input gradient: [CoutP/16][tap][CinP][16], taps rotated
        const int c16 = (int)(i % 16), o = (int)((i / 16) % CinP), tap = (int)((i / (16 * (size_t)CinP)) % 9);
        const int k = (int)(i / (16 * (size_t)CinP * 9)) * 16 + c16;
        uint32_t h, m, l;
        split3_pair((o < Cin && k < Cout) ? wp[((size_t)o * 9 + (8 - tap)) * Cout + k] : 0.0f, 0.0f, h, m, l);
        uint16_t *d = a.dgr[li];
        d[i] = (uint16_t)h; d[total + i] = (uint16_t)m; d[2 * total + i] = (uint16_t)l;
    }
}

// (C,H,W) fp32 -> [3][CP/16][H*W][16] bf16 parts, channels C..CP-1 zero
__global__ void __launch_bounds__(256)
nchw_to_f32s_kernel(const float *__restrict__ x, int C, int HW, int CP, uint16_t *__restrict__ y) {
    const size_t total = (size_t)HW * CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c16 = (int)(i % 16);
        const size_t p = (i / 16) % HW;
        const int c = (int)(i / (16 * (size_t)HW)) * 16 + c16;
        uint32_t h, m, l;
        split3_pair(c < C ? x[(size_t)c * HW + p] : 0.0f, 0.0f, h, m, l);
        y[i] = (uint16_t)h; y[total + i] = (uint16_t)m; y[2 * total + i] = (uint16_t)l;
    }
}

// [3][CP/16][H*W][16] bf16 parts -> (C,H,W) fp32 (h + m + l: exact) through a 64x65 LDS tile
__global__ void __launch_bounds__(256)
f32s_to_nchw_kernel(const uint16_t *__restrict__ x, int C, int HW, int CP, float *__restrict__ y) {
    __shared__ float tile[64][65];
    const size_t part = (size_t)HW * CP;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int p = p0 + i, c = c0 + tx;
        float v = 0.0f;
        if (p < HW && c < C) {
            const size_t o = ((size_t)(c >> 4) * HW + p) * 16 + (c & 15);
            v = (__uint_as_float((uint32_t)x[o] << 16) + __uint_as_float((uint32_t)x[part + o] << 16)) + __uint_as_float((uint32_t)x[2 * part + o] << 16);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, p = p0 + tx;
        if (c < C && p < HW) y[(size_t)c * HW + p] = tile[tx][i];
    }
}

}  // namespace

extern "C" {

int frcnn_f32s_pack_conv_w(const float *w, int Cout, int Cin, uint16_t *w_packed, void *stream) {
    if (!w || !w_packed || Cout < 1 || Cin < 1) return FRCNN_ERR_INVALID;
    const int CoutP = (Cout + 15) / 16 * 16, CinP = (Cin + 15) / 16 * 16;
    const size_t total = (size_t)9 * CoutP * CinP;
    hipLaunchKernelGGL(pack_w_f32s_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, 9, CoutP, CinP, w_packed);
    return frcnn_launch_status();
}

int frcnn_f32s_pack_from_packed(const float *w_packed_f32, int Cin, int Cout, int dgrad, uint16_t *w_split, void *stream) {
    if (!w_packed_f32 || !w_split || Cin < 1 || Cout < 1) return FRCNN_ERR_INVALID;
    const size_t total = (size_t)9 * ((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16);
    hipLaunchKernelGGL(pack_w_f32s_from_packed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed_f32, Cin, Cout,
                       dgrad ? 1 : 0, w_split);
    return frcnn_launch_status();
}

int frcnn_f32s_pack_many(const frcnn_f32s_pack_desc *layers, int n, void *stream) {
    if (!layers || n < 1 || n > 16) return FRCNN_ERR_INVALID;
    PackManyArgs a;
    memset(&a, 0, sizeof(a));
    unsigned blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!layers[i].w_packed_f32 || !layers[i].w_split_fwd || layers[i].Cin < 1 || layers[i].Cout < 1) return FRCNN_ERR_INVALID;
        a.wp[i] = layers[i].w_packed_f32; a.fwd[i] = layers[i].w_split_fwd; a.dgr[i] = layers[i].w_split_dgrad;
        a.cin[i] = layers[i].Cin; a.cout[i] = layers[i].Cout;
        const size_t total = (size_t)9 * ((layers[i].Cout + 15) / 16 * 16) * ((layers[i].Cin + 15) / 16 * 16);
        blocks += (unsigned)((total + 255) / 256);
        a.block_end[i] = blocks;
    }
    a.n = n;
    hipLaunchKernelGGL(pack_w_f32s_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return frcnn_launch_status();
}

int frcnn_f32s_from_nchw_f32(const float *x, int C, int H, int W, uint16_t *y, void *stream) {
    if (!x || !y || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int CP = (C + 15) / 16 * 16;
    const size_t total = (size_t)H * W * CP;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(nchw_to_f32s_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, C, H * W, CP, y);
    return frcnn_launch_status();
}

int frcnn_f32s_to_nchw_f32(const uint16_t *x, int C, int H, int W, float *y, void *stream) {
    if (!x || !y || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int CP = (C + 15) / 16 * 16;
    hipLaunchKernelGGL(f32s_to_nchw_kernel, dim3(frcnn_cdiv(H * W, 64), frcnn_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, C, H * W, CP, y);
    return frcnn_launch_status();
}

// persistent first-layer launch: as many workgroups as the chip seats at once (2 per CU for the split form, 4 for the bf16 form, 3
// for the fp32 form: the kernels' __launch_bounds__), each strides over the tiles
static int conv1_grid(int ntiles, bool split, int seats = 0) {
    const char *e = frcnn_tune("FRCNN_CONV1_WGS_PER_CU");
    const int per_cu = e && atoi(e) > 0 ? atoi(e) : (seats > 0 ? seats : (split ? 2 : 4));
    const char *g = frcnn_tune("FRCNN_CONV1_GRID");                        // tests: an exact workgroup count (the strided tile loop on small images)
    const long slots = g && atoi(g) > 0 ? atoi(g) : (long)frcnn_cu_count() * per_cu;
    return (int)(ntiles < slots ? ntiles : slots);
}

int frcnn_conv1_f32s(const float *x, const float *w, const float *bias, uint16_t *y, int Cin, int Cout, int H, int W, int relu, void *stream) {
    if (!x || !w || !bias || !y || Cin < 1 || Cin > 3 || Cout < 1 || Cout > 64 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if ((size_t)H * W * 64 * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;          // one part of the output behind a 32-bit buffer range
    const int xtiles = frcnn_cdiv(W, 64), ntiles = xtiles * frcnn_cdiv(H, 4);
    const dim3 grid(conv1_grid(ntiles, true));
    if (Cout > 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_f32s_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, Cin, Cout, H, W, relu, 0, (float *)nullptr, xtiles, ntiles);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_f32s_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, Cin, Cout, H, W, relu, 0, (float *)nullptr, xtiles, ntiles);
    return frcnn_launch_status();
}

int frcnn_conv1_f32s_train(const float *x, const float *w_packed_f32, const float *bias, uint16_t *y_split, float *y_nchw, int Cin, int Cout, int H, int W,
                           int relu, void *stream) {
    if (!x || !w_packed_f32 || !bias || !y_split || Cin < 1 || Cin > 3 || Cout < 1 || Cout > 64 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if ((size_t)H * W * 64 * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;          // one part of the output behind a 32-bit buffer range
    const int xtiles = frcnn_cdiv(W, 64), ntiles = xtiles * frcnn_cdiv(H, 4);
    const dim3 grid(conv1_grid(ntiles, true));
#define FRCNN_CONV1_TRAIN(NCB, DUAL_) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_f32s_kernel<NCB, true, DUAL_>), grid, dim3(256), 0, (hipStream_t)stream, x, \
                                                         w_packed_f32, bias, y_split, Cin, Cout, H, W, relu, 1, y_nchw, xtiles, ntiles)
    if (y_nchw != nullptr) { if (Cout > 32) FRCNN_CONV1_TRAIN(2, true); else FRCNN_CONV1_TRAIN(1, true); }
    else { if (Cout > 32) FRCNN_CONV1_TRAIN(2, false); else FRCNN_CONV1_TRAIN(1, false); }
#undef FRCNN_CONV1_TRAIN
    return frcnn_launch_status();
}

// conv1_1 of the fp32 chain (frcnn_conv3x3_f32 hands layers with Cin <= 3 and Cout <= 64 over): fp32 NCHW image in, the trainers' packed
// weights [(ci * 9 + tap)][Cout], fp32 NCHW out, native fp32 MFMA arithmetic
int frcnn_conv1_f32(const float *x, const float *w_packed, const float *bias, float *y, int Cin, int Cout, int H, int W, int relu, void *stream) {
    if (!x || !w_packed || !bias || !y || Cin < 1 || Cin > 3 || Cout < 1 || Cout > 64 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if ((size_t)H * W * 64 * 4 >= (1ull << 31)) return FRCNN_ERR_INVALID;          // the output behind a 32-bit buffer range
    const int xtiles = frcnn_cdiv(W, 64), ntiles = xtiles * frcnn_cdiv(H, 4);
    const dim3 grid(conv1_grid(ntiles, false, 3));
    if (Cout > 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_f32s_kernel<2, false, false, true>), grid, dim3(256), 0, (hipStream_t)stream, x, w_packed, bias, (uint16_t *)nullptr, Cin, Cout, H, W, relu, 1, y, xtiles, ntiles);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_f32s_kernel<1, false, false, true>), grid, dim3(256), 0, (hipStream_t)stream, x, w_packed, bias, (uint16_t *)nullptr, Cin, Cout, H, W, relu, 1, y, xtiles, ntiles);
    return frcnn_launch_status();
}

int frcnn_conv1_bf16(const float *x, const float *w, const float *bias, uint16_t *y, int Cin, int Cout, int H, int W, int relu, void *stream) {
    if (!x || !w || !bias || !y || Cin < 1 || Cin > 3 || Cout < 1 || Cout > 64 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if ((size_t)H * W * 64 * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;          // one part of the output behind a 32-bit buffer range
    const int xtiles = frcnn_cdiv(W, 64), ntiles = xtiles * frcnn_cdiv(H, 4);
    const dim3 grid(conv1_grid(ntiles, false));
    if (Cout > 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_f32s_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, Cin, Cout, H, W, relu, 0, (float *)nullptr, xtiles, ntiles);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1_f32s_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, Cin, Cout, H, W, relu, 0, (float *)nullptr, xtiles, ntiles);
    return frcnn_launch_status();
}

constexpr size_t kF32sCounterPageBytes = 64 * 1024;

// split-K factor: launches that leave most of the chip's 2 x CUs workgroup slots empty split their K range (FRCNN_F32S_SPLIT overrides)
constexpr long kF32sMaxSplitTiles = 2048;     // launches with more tiles than this never split

// split-K factor: only launches that cannot fill the chip's 2 x CUs workgroup slots ONCE split (160 tiles on 512 slots -> 3).
// Filling the last partial round of bigger launches the same way (conv3_x: 1216 tiles -> 2 splits, conv4_x: 608 -> 4) measured
// 5 % faster on conv4_2/3 and 12 % SLOWER on conv3_x / conv4_1 (r02o): a workgroup left alone on its CU runs almost twice as fast,
// so a thin last round costs far less than the slot count suggests, and every split pays its own prologue and partial tile.
// (FRCNN_F32S_SPLIT overrides.)
static int conv_f32s_pick_split(long tiles, int chunks) {
    const char *e = frcnn_tune("FRCNN_F32S_SPLIT");
    int s = e ? atoi(e) : (int)((2L * frcnn_cu_count()) / (tiles > 0 ? tiles : 1));
    if (tiles > kF32sMaxSplitTiles && !e) s = 1;
    // a launch that fills the slots once and leaves a thin second round (conv4_2/3: 608 tiles) gains 5 % from two splits when its
    // chunk loop is long enough to carry them (512 input channels); with 256 channels (conv4_1) it loses (r02o)
    if (!e && tiles > 2L * frcnn_cu_count() && 2 * tiles <= 3 * 2L * frcnn_cu_count() && chunks >= 32) s = 2;
    if (s > 4) s = 4;
    if (s < 1) s = 1;
    while (s > 1 && chunks / s < 4) --s;                          // a split should still carry a few chunks
    return s;
}

size_t frcnn_conv_f32s_workspace_bytes(int Cin, int Cout, int H, int W) {
    if (Cin < 1 || Cout < 1 || H < 1 || W < 1) return 0;
    long tiles = (long)frcnn_cdiv(W, 32) * frcnn_cdiv(H, 4) * frcnn_cdiv((Cout + 15) / 16 * 16, 64);
    if (tiles > kF32sMaxSplitTiles) tiles = 0;                                         // never split: counters only
    return kF32sCounterPageBytes + (size_t)tiles * 4 * 256 * 32 * sizeof(float);      // up to 4 splits x 32 KB of accumulators per tile
}

int frcnn_conv_f32s_workspace_init(void *workspace, size_t workspace_bytes, void *stream) {
    if (!workspace || workspace_bytes < kF32sCounterPageBytes) return FRCNN_ERR_INVALID;
    FRCNN_HIP_TRY(hipMemsetAsync(workspace, 0, kF32sCounterPageBytes, (hipStream_t)stream));
    return FRCNN_OK;
}

static int conv3x3_f32s_launch(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, float *y_nchw, const float *mask, int Cin, int Cout,
                               int H, int W, int relu, int out_mode, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !w_packed || !bias || !y || Cin < 1 || Cout < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if (out_mode < 0 || out_mode > 2 || (out_mode == 2 && !relu)) return FRCNN_ERR_INVALID;
    const int CinP = (Cin + 15) / 16 * 16, CoutP = (Cout + 15) / 16 * 16;
    if ((size_t)kParts * H * W * CinP * 2 >= (1ull << 31) || (size_t)kParts * 9 * CoutP * CinP * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;   // 32-bit buffer offsets, top bit = out of range
    if ((mask || y_nchw) && (size_t)Cout * H * W * 4 >= (1ull << 31)) return FRCNN_ERR_INVALID;     // the fp32 maps sit behind 32-bit buffer ranges
    const int xtiles = frcnn_cdiv(W, 32), ytiles = frcnn_cdiv(H, 4), cotiles = frcnn_cdiv(CoutP, 64);
    const long tiles = (long)xtiles * ytiles * cotiles;
    // split-K needs the workspace (partial tiles + the zeroed counter page); without one every tile is whole
    int nsplit = 1;
    if (workspace && tiles * 4 <= 16384) {
        nsplit = conv_f32s_pick_split(tiles, CinP / kCK);
        if (workspace_bytes < kF32sCounterPageBytes + (size_t)tiles * nsplit * 256 * 32 * sizeof(float)) nsplit = 1;
    }
    float *partials = nsplit > 1 ? (float *)((char *)workspace + kF32sCounterPageBytes) : nullptr;
    int *counters = nsplit > 1 ? (int *)workspace : nullptr;
    dim3 grid((unsigned)(tiles * nsplit));
    int xcd_cotiles = 0;
    // 1 enables.  Four back-to-back launches of one layer gain 5-10 % from it (r02j), the real 14-layer chain nothing (r02o: 2.131 vs
    // 2.142 ms) -- between different layers the slabs are cold either way -- so the plain order stays the default
    const char *xcd_env = frcnn_tune("FRCNN_F32S_XCD");
    if ((cotiles == 1 || cotiles == 2 || cotiles == 4 || cotiles == 8) && xcd_env && xcd_env[0] == '1') {
        xcd_cotiles = cotiles;
        grid = dim3((unsigned)(8 * frcnn_cdiv(xtiles * ytiles * nsplit, 8 / cotiles)));
    }
    const char *abl_env = frcnn_tune("FRCNN_F32S_ABL");
    const int abl = abl_env ? atoi(abl_env) : 0;
#define FRCNN_F32S_LAUNCH(...) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f32s_kernel<__VA_ARGS__>), grid, dim3(256), 0, stream, x, w_packed, bias, y, CinP, Cout, CoutP, H, W, relu, out_mode, xtiles, ytiles, nsplit, partials, counters, xcd_cotiles, y_nchw, mask)
    switch (abl) {
#ifdef FRCNN_TIMING_ABLATIONS                                   // WRONG results: sweeps only, never shipped
        case 1: FRCNN_F32S_LAUNCH(2, 1); break;
        case 4: FRCNN_F32S_LAUNCH(2, 4); break;
        case 5: FRCNN_F32S_LAUNCH(2, 5); break;
        case 9: FRCNN_F32S_LAUNCH(2, 9); break;
        case 8: FRCNN_F32S_LAUNCH(2, 8); break;
        case 21: FRCNN_F32S_LAUNCH(1, 1, 2); break;
#endif
        case 20: FRCNN_F32S_LAUNCH(1, 0, 2); break;              // two-stage ring, one workgroup per CU (measured slower: kept for A/B runs)
        default: FRCNN_F32S_LAUNCH(2);
    }
#undef FRCNN_F32S_LAUNCH
    return frcnn_launch_status();
}

int frcnn_conv3x3_f32s_ws(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H, int W, int relu, int out_mode,
                          void *workspace, size_t workspace_bytes, void *stream) {
    return conv3x3_f32s_launch(x, w_packed, bias, y, nullptr, nullptr, Cin, Cout, H, W, relu, out_mode, workspace, workspace_bytes, (hipStream_t)stream);
}

int frcnn_conv3x3_f32s_train(const uint16_t *x, const uint16_t *w_packed, const float *bias, uint16_t *y_split, float *y_nchw, const float *mask, int Cin,
                             int Cout, int H, int W, int relu, void *workspace, size_t workspace_bytes, void *stream) {
    if (!y_split && !y_nchw) return FRCNN_ERR_INVALID;
    if (!y_split) return conv3x3_f32s_launch(x, w_packed, bias, y_nchw, nullptr, mask, Cin, Cout, H, W, relu, 1, workspace, workspace_bytes, (hipStream_t)stream);
    return conv3x3_f32s_launch(x, w_packed, bias, y_split, y_nchw, mask, Cin, Cout, H, W, relu, 0, workspace, workspace_bytes, (hipStream_t)stream);
}

int frcnn_conv3x3_f32s(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H, int W, int relu, int out_mode,
                       void *stream) {
    return frcnn_conv3x3_f32s_ws(x, w_packed, bias, y, Cin, Cout, H, W, relu, out_mode, nullptr, 0, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Fully connected layers on split tensors: y(M,N) = act(x(M,K) @ W(N,K)^T + b) in fp32, the products as six bf16 MFMAs of the 3-way
// split operands (L.Linear + F.relu, /root/reference/models/faster_rcnn.py:33-36,127-134).  x and W are [3 parts][rows][K] bf16 (K
// contiguous: a lane's eight consecutive k-values are one 16-byte read).  Workgroup = 4 waves along N: tile (32*AM) x 128, K panels of
// 32 (64 B per row and part) through LDS-DMA -- 1 KB pieces of 16 rows, 16-byte group g of row r in slot 4r + (g ^ ((r >> 2) & 3)), so
// the 16 lanes of a ds_read_b128 group (16 consecutive rows, one group index) fall on 16 distinct bank slots -- single-stage ring, two
// workgroups per CU, split-K partial slabs + one reduce / bias / ReLU pass that can write the result split again.
namespace {

constexpr int kFK = 32;             // k per panel

template <int AM>
__global__ void __launch_bounds__(256, 2)
linear_f32s_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w, float *__restrict__ part, int M, int N, int K, int k_per_split) {
    constexpr int BM = 32 * AM, BN = 128;
    constexpr int XP = kParts * BM / 16, WP = kParts * BN / 16;   // 1 KB pieces (16 rows each)
    constexpr int PIECES = XP + WP, PPW = (PIECES + 3) / 4;
    constexpr int X_BYTES = XP * 1024, STAGE = PIECES * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_per_split, k_end = min(K, k_begin + k_per_split);
    const int nchunks = (k_end - k_begin) / kFK;             // whole panels: the host guarantees K % 32 == 0
    const uint32_t x_part_bytes = (uint32_t)((size_t)M * K * 2), w_part_bytes = (uint32_t)((size_t)N * K * 2);
    const frcnn_buf_t xbuf = frcnn_make_buf(x, kParts * x_part_bytes);
    const frcnn_buf_t wbuf = frcnn_make_buf(w, kParts * w_part_bytes);
    // piece p of the x region: rows R = p * 16 .. + 15 of [part][BM]; of the w region likewise over [part][BN]
    uint32_t poff[PPW];
    bool isx[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pid = wave + 4 * q;
        isx[q] = pid < XP;
        const int sl = (isx[q] ? pid : pid - XP) * 64 + lane, R = sl >> 2, g = (sl & 3) ^ ((R >> 2) & 3);
        const int rows = isx[q] ? BM : BN;
        const int p = R / rows, r = R - p * rows;
        const int gr = (isx[q] ? m0 : n0) + r;
        const bool ok = pid < PIECES && p < kParts && gr < (isx[q] ? M : N);
        poff[q] = ok ? (uint32_t)p * (isx[q] ? x_part_bytes : w_part_bytes) + (uint32_t)(((size_t)gr * K + k_begin + 8 * g) * 2) : kBufOob;
    }
    auto issue = [&](int chunk) {
        unsigned char *dst = lds + wave * 1024;
        const uint32_t so = (uint32_t)chunk * (kFK * 2);
#pragma unroll
        for (int q = 0; q < PPW; ++q)
            if (4 * q + 3 < PIECES || wave + 4 * q < PIECES) frcnn_buf_load_lds_b128(isx[q] ? xbuf : wbuf, dst + q * 4096, poff[q], so);
    };
    frcnn_f32x16 acc[AM];
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int l31 = lane & 31, khalf = lane >> 5;
    // fragment offsets: row r of a region, k-step ks -> group 2 ks + khalf
    auto frag_off = [&](int r, int ks) { return (uint32_t)(r * 64 + (((2 * ks + khalf) ^ ((r >> 2) & 3)) << 4)); };
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        issue(chunk);
        frcnn_wait_vmcnt<0>();
        frcnn_barrier_nofence();
#pragma unroll
        for (int ks = 0; ks < kFK / 16; ++ks) {
            uint4 b[kParts];
#pragma unroll
            for (int p = 0; p < kParts; ++p) b[p] = *reinterpret_cast<const uint4 *>(lds + X_BYTES + frag_off(p * BN + wave * 32 + l31, ks));
#pragma unroll
            for (int i = 0; i < AM; ++i) {
                uint4 a[kParts];
#pragma unroll
                for (int p = 0; p < kParts; ++p) a[p] = *reinterpret_cast<const uint4 *>(lds + frag_off(p * BM + 32 * i + l31, ks));
                acc[i] = frcnn_mfma_32x32x16_bf16(a[2], b[0], acc[i]);     // l.h
                acc[i] = frcnn_mfma_32x32x16_bf16(a[0], b[2], acc[i]);     // h.l
                acc[i] = frcnn_mfma_32x32x16_bf16(a[1], b[1], acc[i]);     // m.m
                acc[i] = frcnn_mfma_32x32x16_bf16(a[1], b[0], acc[i]);     // m.h
                acc[i] = frcnn_mfma_32x32x16_bf16(a[0], b[1], acc[i]);     // h.m
                acc[i] = frcnn_mfma_32x32x16_bf16(a[0], b[0], acc[i]);     // h.h
            }
        }
        if (chunk + 1 < nchunks) frcnn_barrier_nofence();         // everybody is done reading before the stage is refilled
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

// sum of the split-K slabs + bias (+ ReLU) -> fp32 (out_split 0) or the three bf16 parts [3][M][N] (out_split 1)
__global__ void __launch_bounds__(256)
linear_reduce_f32s_kernel(const float *__restrict__ part, const float *__restrict__ bias, void *__restrict__ y, int M, int N, int splits, int relu, int out_split) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float b = bias[i % N];                         // issued ahead of the slab loads: it is used last
        float v = frcnn_sum_splits(part, total, i, splits);
        v += b;
        if (relu) v = fmaxf(v, 0.0f);
        if (out_split) {
            uint32_t h, m, l;
            split3_pair(v, 0.0f, h, m, l);
            uint16_t *o = reinterpret_cast<uint16_t *>(y);
            o[i] = (uint16_t)h; o[total + i] = (uint16_t)m; o[2 * total + i] = (uint16_t)l;
        } else reinterpret_cast<float *>(y)[i] = v;
    }
}

// flat fp32 array -> [3][n] bf16 parts, and back (h + m + l: exact)
__global__ void __launch_bounds__(256)
f32s_split_kernel(const float *__restrict__ x, size_t n, uint16_t *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h, m, l;
        split3_pair(x[i], 0.0f, h, m, l);
        y[i] = (uint16_t)h; y[n + i] = (uint16_t)m; y[2 * n + i] = (uint16_t)l;
    }
}
__global__ void __launch_bounds__(256)
f32s_join_kernel(const uint16_t *__restrict__ x, size_t n, float *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = (__uint_as_float((uint32_t)x[i] << 16) + __uint_as_float((uint32_t)x[n + i] << 16)) + __uint_as_float((uint32_t)x[2 * n + i] << 16);
}

struct LinPlanS { int am, mblocks, nblocks, splits, k_per_split; };
static LinPlanS plan_linear_f32s(int M, int N, int K) {
    LinPlanS p;
    p.am = (M > 96) ? 5 : (M > 32 ? 3 : 1);
    p.mblocks = frcnn_cdiv(M, 32 * p.am);
    p.nblocks = frcnn_cdiv(N, 128);
    const int tiles = p.mblocks * p.nblocks, kchunks = K / kFK;
    int splits = frcnn_cdiv(2 * frcnn_cu_count(), tiles);          // two workgroups per CU
    if (splits > kchunks / 8) splits = kchunks / 8;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    p.k_per_split = frcnn_cdiv(kchunks, splits) * kFK;
    p.splits = frcnn_cdiv(K, p.k_per_split);
    return p;
}

}  // namespace

extern "C" {

int frcnn_f32s_split(const float *x, size_t n, uint16_t *y, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!x || !y) return FRCNN_ERR_INVALID;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(f32s_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, y);
    return frcnn_launch_status();
}

int frcnn_f32s_join(const uint16_t *x, size_t n, float *y, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!x || !y) return FRCNN_ERR_INVALID;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(f32s_join_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, y);
    return frcnn_launch_status();
}

size_t frcnn_linear_f32s_workspace_bytes(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    const LinPlanS p = plan_linear_f32s(M, N, K);
    return frcnn_align256((size_t)p.splits * M * N * sizeof(float));
}

int frcnn_linear_f32s(const uint16_t *x, const uint16_t *w, const float *bias, void *y, int M, int N, int K, int relu, int out_split, void *workspace,
                      size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w || !bias || !y || M < 1 || N < 1 || K < 1 || (K % kFK) != 0) return FRCNN_ERR_INVALID;
    if ((size_t)kParts * M * K * 2 >= (1ull << 31) || (size_t)kParts * N * K * 2 >= (1ull << 31)) return FRCNN_ERR_INVALID;
    const LinPlanS p = plan_linear_f32s(M, N, K);
    if (!workspace || workspace_bytes < (size_t)p.splits * M * N * sizeof(float)) return FRCNN_ERR_INVALID;
    float *part = (float *)workspace;
    const dim3 grid(p.nblocks, p.mblocks, p.splits);
    if (p.am == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_f32s_kernel<5>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else if (p.am == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_f32s_kernel<3>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_f32s_kernel<1>), grid, dim3(256), 0, stream, x, w, part, M, N, K, p.k_per_split);
    const size_t total = (size_t)M * N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(linear_reduce_f32s_kernel, dim3(blocks), dim3(256), 0, stream, part, bias, y, M, N, p.splits, relu, out_split);
    return frcnn_launch_status();
}

}  // extern "C"
