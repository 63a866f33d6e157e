// roi_pool.hip -- RoIPooling2D forward / backward for gfx950 (HBM-bound: 35 MB algorithmic per image).
//
// Replaces chainer.functions.roi_pooling_2d as called at /root/reference/models/faster_rcnn.py:125-126
// (Chainer v1 ROIPooling2D; semantics restated in oracle/c/frcnn_oracle.c:oracle_roi_pool_fwd).
//
// Layout decision.  All C channels of one RoI share the same 7x7 bin geometry, so the natural wave
// mapping is lane = channel: every bin cell is then ONE fully coalesced read (64 lanes x VEC floats,
// uniform address + lane), there is no per-lane index arithmetic and no divergence.  That needs the
// feature map channel-last, so the forward runs in two launches:
//   chw_to_hwc_kernel   (C,H*W) -> (H*W,C) through a 64x65 LDS tile          4.9 MB in, 4.9 MB out
//   roi_pool_hwc_kernel block = (roi, 64*VEC channels), wave = output row ph; the 49 maxima per channel go
//                       through LDS ([channel][49], odd stride -> conflict-free) so the (R,C,7,7) result
//                       leaves as one contiguous, float4-coalesced 64*VEC*49-float run per block.
// The pipeline can skip the first launch by keeping conv5_3's output channel-last (frcnn_roi_pool_fwd_hwc).
//
// Bin edges replicate Chainer's CPU arithmetic exactly: round-half-even of the float32 product
// x*spatial_scale, then floor(p*stride) / ceil((p+1)*stride) with stride = (double)extent/out in IEEE
// double -- NOT integer division (7*(29/7.) = 29.000000000000004 -> ceil 30; see
// tests/test_oracle_pinned.py::test_roi_bin_edges_need_double_arithmetic).  Empty bins give 0 / argmax -1.
#include "frcnn_common.h"
#include <frcnn_intrin.h>   // angle brackets: shadowed by the test emulator
#include <frcnn_buffer.h>
#include <math.h>
#include <algorithm>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

namespace {

__global__ void __launch_bounds__(256)
chw_to_hwc_kernel(const float *__restrict__ x, float *__restrict__ xt, int C, int HW) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < HW) ? x[(size_t)c * HW + p] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) xt[(size_t)p * C + c] = tile[tx][i];
    }
}

constexpr int kBatch = 8;   // bin cells fetched per round

struct RoiGeom { int xs, ys, rw, rh; };

__device__ __forceinline__ RoiGeom roi_geometry(const float *__restrict__ roi, float scale) {
    // Python round() on the float32 product = round-half-to-even (rintf under the default rounding mode)
    const int xs = (int)rintf(roi[1] * scale), ys = (int)rintf(roi[2] * scale);
    const int xe = (int)rintf(roi[3] * scale), ye = (int)rintf(roi[4] * scale);
    RoiGeom g;
    g.xs = xs; g.ys = ys;
    g.rw = max(xe - xs + 1, 1);
    g.rh = max(ye - ys + 1, 1);
    return g;
}
__device__ __forceinline__ void bin_range(int p, int extent, int out, int offset, int limit, int &lo, int &hi) {
    const double stride = (double)extent / (double)out;
    lo = (int)floor((double)p * stride) + offset;
    hi = (int)ceil((double)(p + 1) * stride) + offset;
    lo = min(max(lo, 0), limit);
    hi = min(max(hi, 0), limit);
}

// One block = one RoI x (64*VEC) channels; OUTH waves, wave ph computes output row ph.
template <int VEC, bool ARGMAX>
__global__ void __launch_bounds__(448)
roi_pool_hwc_kernel(const float *__restrict__ xt, int C, int H, int W, const float *__restrict__ rois, int roi_cols, int outh,
                    int outw, float scale, float *__restrict__ y, int32_t *__restrict__ argmax) {
    constexpr int CB = 64 * VEC;
    __shared__ __attribute__((aligned(16))) float s_val[CB * 49];
    __shared__ int32_t s_idx[ARGMAX ? CB * 49 : 1];
    const int r = blockIdx.x, c0 = blockIdx.y * CB;
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int bins = outh * outw;
    // rows are [batch,x1,y1,x2,y2] (roi_cols 5) or bare [x1,y1,x2,y2] (roi_cols 4, ProposalLayer's output)
    const RoiGeom g = roi_geometry(rois + (size_t)roi_cols * r + (roi_cols - 5), scale);
    if (ph < outh) {
        int hs, he;
        bin_range(ph, g.rh, outh, g.ys, H, hs, he);
        const int cl = c0 + lane * VEC;
        for (int pw = 0; pw < outw; ++pw) {
            int ws, we;
            bin_range(pw, g.rw, outw, g.xs, W, ws, we);
            float m[VEC];
            int32_t mi[VEC];
            const bool empty = (he <= hs) || (we <= ws);
#pragma unroll
            for (int v = 0; v < VEC; ++v) { m[v] = 0.0f; mi[v] = -1; }
            if (!empty && cl < C) {
                // Cells of the bin in row-major order, kBatch independent loads in flight at a time (the wave
                // is latency-bound otherwise).  The tail re-reads the bin's last cell: with the strict `>`
                // below a duplicate can never displace the first maximum, so no predication is needed.
                const int bw = we - ws, ncell = (he - hs) * bw;
                const float *base = xt + cl;
                int hh = hs, ww = ws;                       // wave-uniform cursor
#pragma unroll
                for (int v = 0; v < VEC; ++v) { m[v] = 0.0f; mi[v] = hs * W + ws; }
                for (int i0 = 0; i0 < ncell; i0 += kBatch) {
                    float val[kBatch][VEC];
                    int pos[kBatch];
#pragma unroll
                    for (int q = 0; q < kBatch; ++q) {
                        pos[q] = hh * W + ww;
                        const float *src = base + (size_t)pos[q] * C;
                        if constexpr (VEC == 2) { const float2 t2 = *reinterpret_cast<const float2 *>(src); val[q][0] = t2.x; val[q][1] = t2.y; }
                        else val[q][0] = src[0];
                        if (i0 + q + 1 < ncell) { if (++ww == we) { ww = ws; ++hh; } }   // stay on the last cell past the end
                    }
#pragma unroll
                    for (int q = 0; q < kBatch; ++q)
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
                            if ((i0 == 0 && q == 0) || val[q][v] > m[v]) { m[v] = val[q][v]; mi[v] = pos[q]; }   // first maximum wins
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                s_val[(lane * VEC + v) * bins + ph * outw + pw] = m[v];
                if (ARGMAX) s_idx[(lane * VEC + v) * bins + ph * outw + pw] = mi[v];
            }
        }
    }
    __syncthreads();
    // (r, c0 .. c0+CB, :, :) is one contiguous run of CB*bins floats in y
    const int nch = min(CB, C - c0);
    const int total = nch * bins;
    const size_t base = ((size_t)r * C + c0) * bins;
    if (((base | (size_t)total) & 3) == 0) {
        float4 *dst = reinterpret_cast<float4 *>(y + base);
        const float4 *srcv = reinterpret_cast<const float4 *>(s_val);
        for (int i = threadIdx.x; i < total / 4; i += blockDim.x) dst[i] = srcv[i];
    } else {
        for (int i = threadIdx.x; i < total; i += blockDim.x) y[base + i] = s_val[i];
    }
    if (ARGMAX)
        for (int i = threadIdx.x; i < total; i += blockDim.x) argmax[base + i] = s_idx[i];
}

// ------------------------------------------------------------------------------------------------
// Plane-resident forward (the fast path; NCHW in, no transpose).  The 300 RoIs of an image overlap ~50x: summed
// over RoIs the bins cover 56 M cells (x 4 B = 225 MB) of a 4.9 MB map, so the kernel is bound by how fast cells
// can be re-read, not by HBM.  Each workgroup therefore copies CG whole channel planes into LDS ONCE (row pitch
// 64, plane stride H*64+8 dwords: ~2-way bank conflicts on real RoI sets) and serves a slice of the RoIs from there:
//   * grid = (C/CG channel groups) x (RoI groups); 1024 threads = 16 waves, one workgroup per CU;
//   * while the plane loads are in flight every 16-lane group works out one RoI's bin edges (double arithmetic,
//     once per RoI and workgroup); after ONE barrier the waves never synchronise again;
//   * a wave pulls the next RoI from an LDS counter; lane = (channel, pw) walks the RoI's rows ONCE: per row the
//     maximum over the lane's column range (clamped columns min(k, bw-1): re-reading the last column cannot
//     displace the first maximum under strict `>`, so all lanes run the same wave-uniform loops, two rows =
//     up to eight independent ds_read_b32 in flight), and a row that two adjacent output rows share (floor/ceil
//     edges overlap by one) is read once and carried over;
//   * the RoI's [CG][7][7] block is staged in the wave's own LDS slot and leaves as ONE contiguous
//     float4-coalesced run of CG*49 floats -- HBM sees only full-line writes.
// Exactly the oracle's scan order: first cell, then strict `>` row-major (first maximum wins; later NaNs never win).
constexpr int kRowPitch = 64;            // LDS row pitch (dwords)
constexpr int kPlanePad = 8;
constexpr int kPlaneFloats = 19584;      // 8 planes of 38 x 64 + pad (+ slack for the clamped over-reads)
constexpr int kMaxBins = 49;
constexpr int kPlaneWaves = 16;
constexpr int kMaxRoisPerBlock = 128;
constexpr int kMaxBinW = 12;             // columns of one bin (<= ceil(64/outw)+1 for outw >= 6; checked on the host)
constexpr int kMaxBinH = 8;              // rows of one bin (ceil(H/outh)+1; checked on the host)

__device__ __forceinline__ void frcnn_wave_sync() { __builtin_amdgcn_wave_barrier(); }

// running (max, argmax) under the oracle's rule: replace only on strictly greater
// (inference: the index is not tracked and the update is one v_max_f32; a NaN in m is restored by the caller)
template <bool ARGMAX>
__device__ __forceinline__ void take_gt(float &m, int &mi, float v, int vi) {
    if constexpr (ARGMAX) {
        const bool gt = v > m;
        m = gt ? v : m;
        mi = gt ? vi : mi;
    } else {
        m = frcnn_max_f32(m, v);
    }
}

// One RoI, one lane = (channel, pw): walk the RoI's map rows once and emit the outh bin maxima of this lane's column
// range into the wave's staging slot.  NC = columns read per row (the wave's widest bin rounded up to 4; columns past
// the lane's own width are clamped duplicates).  Every branch is wave-uniform (scalar) and the loops are real loops:
// the kernel's code must stay far below the 64 KB instruction cache -- a fully unrolled version of this scan ran
// 8 k cycles per RoI on instruction fetch alone.
template <bool ARGMAX, int NC>
__device__ __forceinline__ void roi_scan_lane(const float *pc, int ws, int bw, int W, const int *hr, int outh, int outw,
                                              bool lane_on, float *sv_lane, int32_t *si_lane) {
    const int bw1 = max(bw, 1);
    int off[NC];
    const float *col[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { off[k] = min(k, bw1 - 1); col[k] = pc + off[k]; }
    // maximum of one map row over the lane's columns, left to right; advances the row pointers
    auto row_max = [&](const float *(&b)[NC], float &rm, int &ri, float &f) {
        float v[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) { v[k] = *b[k]; b[k] += kRowPitch; }
        rm = v[0];
        ri = 0;
        f = v[0];
#pragma unroll
        for (int k = 1; k < NC; ++k) take_gt<ARGMAX>(rm, ri, v[k], off[k]);
    };
    float carry_m = 0.0f, carry_f = 0.0f;      // last map row the previous bin read: its maximum and its first cell
    int carry_i = -1, carry_h = -1;
    int hr_next = hr[0];
#pragma unroll 1
    for (int ph = 0; ph < outh; ++ph) {
        const int hrv = hr_next;
        if (ph + 1 < outh) hr_next = hr[ph + 1];                            // one bin ahead: hides the LDS latency
        const int hs = hrv & 0xffff, bh = (hrv >> 16) - hs;
        float m = 0.0f, first = 0.0f;
        int mi = -1;
        if (bh > 0) {                                                       // wave-uniform
            const float *b[NC];
            int dh = 0;
            if (carry_h == hs) {                                            // first row already read by the bin above
                m = carry_m; mi = carry_i; first = carry_f; dh = 1;
#pragma unroll
                for (int k = 0; k < NC; ++k) b[k] = col[k] + (hs + 1) * kRowPitch;
            } else {
#pragma unroll
                for (int k = 0; k < NC; ++k) b[k] = col[k] + hs * kRowPitch;
                row_max(b, m, mi, first);
                if (ARGMAX) mi += hs * W + ws;
                carry_m = m; carry_i = mi; carry_f = first; carry_h = hs; dh = 1;
            }
#pragma unroll 1
            for (; dh < bh; ++dh) {
                float rm, f;
                int ri;
                row_max(b, rm, ri, f);
                if (ARGMAX) ri += (hs + dh) * W + ws;
                take_gt<ARGMAX>(m, mi, rm, ri);
                carry_m = rm; carry_i = ri; carry_f = f; carry_h = hs + dh;
            }
            if (!ARGMAX && first != first) m = first;                       // a NaN first cell stays (`>` never replaces it)
        }
        if (bw <= 0) { m = 0.0f; mi = -1; }
        if (lane_on) {
            sv_lane[ph * outw] = m;
            if (ARGMAX) si_lane[ph * outw] = mi;
        }
    }
}

// The same for a RoI whose widest bin exceeds kMaxBinW columns -- only a RoI LARGER THAN THE MAP has such bins (an in-map RoI's are at most
// ceil(W / outw) + 1 wide, which the host checks against kMaxBinW) -- the oracle's scan itself, cell by cell with per-lane trip counts: the bin's
// first cell seeds the maximum, a later cell replaces it only under a strict `>`.  Rare and slow on purpose (round 5 sent these bins through the
// 12-column form, which read 12 of their columns: wrong maxima for e.g. [-1e4, -1e4, 2e4, 2e4]).
template <bool ARGMAX>
__device__ __attribute__((noinline)) void roi_scan_lane_wide(const float *plane, int ws, int bw, int W, const int *hr, int outh, int outw, bool lane_on,
                                                float *sv_lane, int32_t *si_lane) {
    for (int ph = 0; ph < outh; ++ph) {
        const int hrv = hr[ph];
        const int hs = hrv & 0xffff, he = hrv >> 16;
        float m = 0.0f;
        int mi = -1;
        if (lane_on && he > hs && bw > 0) {
            m = plane[hs * kRowPitch + ws];
            mi = hs * W + ws;
            for (int h = hs; h < he; ++h)
                for (int w = ws; w < ws + bw; ++w) {
                    const float v = plane[h * kRowPitch + w];
                    if (v > m) { m = v; mi = h * W + w; }
                }
        }
        if (lane_on) {
            sv_lane[ph * outw] = m;
            if (ARGMAX) si_lane[ph * outw] = mi;
        }
    }
}

__device__ __forceinline__ uint32_t roi_f32_to_bf16(float f) { return frcnn_pack_bf16x2(f, 0.0f) & 0xffffu; }   // nearest even (v_cvt_pk_bf16_f32)

// OUT16: y is raw bf16 (uint16) -- what the bf16 FC head consumes; pooling itself stays fp32 (a max of fp32 values, then ONE rounding:
// the same bits as pooling to fp32 and converting afterwards, without the 30 MB fp32 round trip)
template <bool ARGMAX, bool OUT16 = false>
__global__ void __launch_bounds__(64 * kPlaneWaves)
roi_pool_planes_kernel(const float *__restrict__ x, int C, int H, int W, const float *__restrict__ rois, int roi_cols, int R,
                       int outh, int outw, float scale, float *__restrict__ y, int32_t *__restrict__ argmax, int CG,
                       int rois_per_block) {
    __shared__ __attribute__((aligned(16))) float planes[kPlaneFloats];
    __shared__ __attribute__((aligned(16))) float stage_val[kPlaneWaves][8 * kMaxBins];
    __shared__ __attribute__((aligned(16))) int32_t stage_idx[ARGMAX ? kPlaneWaves : 1][ARGMAX ? 8 * kMaxBins : 4];
    __shared__ int hrange[kMaxRoisPerBlock][8], wrange[kMaxRoisPerBlock][8], wmax[kMaxRoisPerBlock];
    __shared__ int cost[kMaxRoisPerBlock], order[kMaxRoisPerBlock];
    __shared__ int next_roi;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W, bins = outh * outw;
    const int pstride = H * kRowPitch + kPlanePad;
    const int c0 = blockIdx.x * CG;
    const int cg = min(CG, C - c0);                 // channels this workgroup really has
    // RoI group g of ng takes RoIs g, g + ng, g + 2 ng, ...: sizes are unrelated to rank, so the groups carry equal work
    const int ng = gridDim.y, g0 = blockIdx.y;
    const int nr = (R - g0 + ng - 1) / ng;
    (void)rois_per_block;

    // ---- channel planes -> LDS, one map row per wave and step (rows are contiguous in NCHW: coalesced); the
    //      loads of the first kLoadRows rows are issued before the bin-edge arithmetic below and land during it
    constexpr int kLoadRows = 20;     // 8 planes x 38 rows / 16 waves = 19 rows per wave: one batch, one round of HBM latency
    const float *src = x + (size_t)c0 * HW;
    const int nrows = cg * H;
    if (tid == 0) next_roi = 0;
    for (int row0 = wave; row0 < nrows; row0 += kPlaneWaves * kLoadRows) {
        float v[kLoadRows];
#pragma unroll
        for (int q = 0; q < kLoadRows; ++q) {
            const int row = row0 + q * kPlaneWaves;
            v[q] = (row < nrows && lane < W) ? src[(size_t)row * W + lane] : 0.0f;
        }
        if (row0 == wave) {
            // bin edges of every RoI of this workgroup: 16 lanes per RoI (k < outh: row bins, 8 <= k < 8 + outw: column bins)
            for (int rl0 = 0; rl0 < nr; rl0 += 64) {
                const int rl = rl0 + (tid >> 4), k = tid & 15;
                int bw = 0;
                if (rl < nr) {
                    const RoiGeom g = roi_geometry(rois + (size_t)roi_cols * (g0 + rl * ng) + (roi_cols - 5), scale);
                    int lo, hi;
                    if (k < outh) { bin_range(k, g.rh, outh, g.ys, H, lo, hi); hrange[rl][k] = lo | (hi << 16); }
                    else if (k >= 8 && k < 8 + outw) { bin_range(k - 8, g.rw, outw, g.xs, W, lo, hi); wrange[rl][k - 8] = lo | (hi << 16); bw = max(hi - lo, 0); }
                    if (k == 15) cost[rl] = min(g.rh, H) * 64 + min(g.rw, W);      // rows dominate the scan time
                }
                for (int d = 1; d < 16; d <<= 1) bw = max(bw, __shfl_xor(bw, d, 16));
                if (k == 0 && rl < nr) wmax[rl] = bw;
            }
        }
#pragma unroll
        for (int q = 0; q < kLoadRows; ++q) {
            const int row = row0 + q * kPlaneWaves;
            if (row < nrows && lane < W) { const int c = row / H; planes[c * pstride + (row - c * H) * kRowPitch + lane] = v[q]; }
        }
    }
    __syncthreads();
    // longest RoIs first (rank sort): the dynamic hand-out below then ends with the cheap ones and the waves finish together
    if (tid < nr) {
        const int mine = cost[tid];
        int rank = 0;
        for (int j = 0; j < nr; ++j) { const int cj = cost[j]; rank += (cj > mine) || (cj == mine && j < tid); }
        order[rank] = tid;
    }
    __syncthreads();

    const int cpp = min(cg, 64 / outw);            // channels per wave pass
    const int lc = lane / outw, pw = lane - lc * outw;
    const int run = cg * bins;
    float *sv = stage_val[wave];
    int32_t *si = stage_idx[ARGMAX ? wave : 0];

    for (;;) {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&next_roi, 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= nr) break;
        const int rl = order[slot];
        const int bwm = wmax[rl];
        for (int cb = 0; cb < cg; cb += cpp) {
            const int c = cb + lc;
            const bool lane_on = lc < cpp && c < cg;
            int ws = 0, bw = 0;
            if (lane_on) { const int wr = wrange[rl][pw]; ws = wr & 0xffff; bw = (wr >> 16) - ws; }
            const float *pc = planes + (lane_on ? c : 0) * pstride + min(ws, W - 1);
            float *svl = sv + (lane_on ? c * bins + pw : 0);
            int32_t *sil = si + ((ARGMAX && lane_on) ? c * bins + pw : 0);
            if (bwm <= 4) roi_scan_lane<ARGMAX, 4>(pc, ws, bw, W, hrange[rl], outh, outw, lane_on, svl, sil);
            else if (bwm <= 8) roi_scan_lane<ARGMAX, 8>(pc, ws, bw, W, hrange[rl], outh, outw, lane_on, svl, sil);
            else if (bwm <= kMaxBinW) roi_scan_lane<ARGMAX, 12>(pc, ws, bw, W, hrange[rl], outh, outw, lane_on, svl, sil);
            else roi_scan_lane_wide<ARGMAX>(planes + (lane_on ? c : 0) * pstride, ws, bw, W, hrange[rl], outh, outw, lane_on, svl, sil);
        }
        frcnn_wave_sync();     // the wave's own LDS writes above are read by other lanes below (DS ops of a wave are in order)
        // ---- (r, c0 .. c0 + cg, :, :) is one contiguous run of cg*bins floats of y
        const size_t dst = ((size_t)(g0 + rl * ng) * C + c0) * bins;
        if constexpr (OUT16) {
            uint16_t *y16 = reinterpret_cast<uint16_t *>(y);
            if ((run & 3) == 0 && (dst & 3) == 0) {
                for (int i = lane; i < run / 4; i += 64) {
                    const float4 v = reinterpret_cast<const float4 *>(sv)[i];
                    uint2 pk;
                    pk.x = frcnn_pack_bf16x2(v.x, v.y);
                    pk.y = frcnn_pack_bf16x2(v.z, v.w);
                    reinterpret_cast<uint2 *>(y16 + dst)[i] = pk;
                }
            } else {
                for (int i = lane; i < run; i += 64) y16[dst + i] = (uint16_t)roi_f32_to_bf16(sv[i]);
            }
        } else if ((run & 3) == 0 && (dst & 3) == 0) {
            for (int i = lane; i < run / 4; i += 64) {
                reinterpret_cast<float4 *>(y + dst)[i] = reinterpret_cast<const float4 *>(sv)[i];
                if (ARGMAX) reinterpret_cast<int4 *>(argmax + dst)[i] = reinterpret_cast<const int4 *>(si)[i];
            }
        } else {
            for (int i = lane; i < run; i += 64) {
                y[dst + i] = sv[i];
                if (ARGMAX) argmax[dst + i] = si[i];
            }
        }
        frcnn_wave_sync();     // staging slot is rewritten by the next RoI only after these reads were issued
    }
}

// ------------------------------------------------------------------------------------------------
// Cell-major forward (inference; the default when the map fits): the same LDS-resident idea with the data laid out for the
// LDS itself.  The plane kernel above reads one float per lane (ds_read_b32: 128 B/clk per CU) from addresses that collide at
// random across its (channel, pw) lanes -- PMC: ~45 % of its LDS cycles are bank conflicts, and the LDS array is what it waits
// for.  Here a workgroup holds EIGHT channels of the whole map as 32-byte cells, cells[h][swz(w)] = {c0 .. c0+7}:
//   * a lane reads 16 B = four channels of one cell (ds_read_b128: 256 B/clk per CU, 4x fewer LDS instructions per value, one
//     address computation for four maxima);
//   * lane = (pg, cq, pw): pg = which pair of output rows (ph = 2 pg, 2 pg + 1), cq = channel quad, pw = output column.  The
//     hardware serves a ds_read_b128 in four fixed 16-lane groups; pg is exactly that group, so the lanes that share an LDS cycle
//     read the two quads of SEVEN cells of one map row (the idle eighth lane pair shadows the seventh: a broadcast).  Seven cells
//     in the eight 32-byte slots of a bank row still collide two-way more often than not (measured: half of the LDS-active cycles;
//     a column swizzle made no difference and was dropped) -- but the kernel is bound by VALU issue, not by the LDS (r02 counters);
//   * every lane owns its bins outright (no cross-lane reduction, no staging pass, no wave barrier): it walks the wave's largest
//     bin shape with saturating rows / columns -- a duplicate read cannot displace the first maximum -- so all loops are
//     wave-uniform, and stores its 2 x 4 results straight to y;
//   * bin edges come from two 72-entry tables built on the host with the oracle's double arithmetic (floor(p * (e / out)),
//     ceil((p + 1) * (e / out))), passed by value and copied to LDS: a lookup per RoI instead of fourteen double divisions
//     (extents >= 72 cells take the arithmetic path);
//   * 78 KB of LDS and 512 threads per workgroup: TWO workgroups per CU.
// Scan order = the oracle's: the bin's first cell, then NaN-ignoring maxima; a NaN first cell is restored at the end.
constexpr int kCellPitch = 64;           // cells per LDS row
constexpr int kTabExt = 72;              // the bin tables cover extents 1 .. kTabExt-1

// Bin-edge tables (relative to the RoI origin) for every extent below kTabExt, built ON THE HOST with the oracle's double arithmetic
// -- floor(p * (e / out)), ceil((p + 1) * (e / out)): IEEE double division / multiplication / floor / ceil are correctly rounded, so
// the host's values are the device's -- and handed to the kernel by value; a workgroup copies them into LDS (2.4 KB) on entry.
struct RoiBinTables {
    uint16_t tab[2][kTabExt][8];        // [0] rows (outh), [1] columns (outw): lo | hi << 8
    uint8_t tabmax[2][kTabExt];         // tallest / widest bin of an extent
};

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(frcnn_max_f32(a.x, b.x), frcnn_max_f32(a.y, b.y), frcnn_max_f32(a.z, b.z), frcnn_max_f32(a.w, b.w));
}
__device__ __forceinline__ float4 max4_3(float4 a, float4 b, float4 c) {
    return make_float4(frcnn_max3_f32(a.x, b.x, c.x), frcnn_max3_f32(a.y, b.y, c.y), frcnn_max3_f32(a.z, b.z, c.z), frcnn_max3_f32(a.w, b.w, c.w));
}

// One chunk of NC bin columns starting at k0: both of the lane's bins walk the rows of the wave's tallest bin together (8 reads
// in flight per step), row addresses advance by one pitch and saturate at the bin's last row (a duplicate read cannot displace a
// maximum), and the next step's cells are fetched before the current step's maxima are taken.
template <int NC>
__device__ __forceinline__ void cells_scan_chunk(const float4 *__restrict__ cells, int k0, int c_lo2, int c_hi2, int cq,
                                                 const int (&r0)[2], const int (&r1)[2], int mbh, float4 (&acc)[2]) {
    int ca[NC];                                     // float4 index of column k0 + q of the lane's bin, saturating at its last column
#pragma unroll                                      // (c_lo2 / c_hi2 = 2 x the column: a cell is two float4)
    for (int q = 0; q < NC; ++q) ca[q] = min(c_lo2 + 2 * (k0 + q), c_hi2) + cq;
    int row[2] = {r0[0], r0[1]};                    // float4 index of the current map row of each bin
    float4 buf[2][NC];                              // one register buffer per bin: bin 1's cells are in flight while bin 0's maxima are
                                                    // taken and vice versa (two buffers per bin did not fit the 128-register budget of
                                                    // a 16-wave workgroup)
    auto fetch = [&](int s) {
#pragma unroll
        for (int q = 0; q < NC; ++q) buf[s][q] = cells[row[s] + ca[q]];
    };
    auto take = [&](int s) {
        if constexpr (NC == 1) acc[s] = max4(acc[s], buf[s][0]);
        else if constexpr (NC == 2) acc[s] = max4_3(acc[s], buf[s][0], buf[s][1]);
        else if constexpr (NC == 3) acc[s] = max4(max4_3(acc[s], buf[s][0], buf[s][1]), buf[s][2]);
        else acc[s] = max4_3(max4_3(acc[s], buf[s][0], buf[s][1]), buf[s][2], buf[s][3]);
    };
    fetch(0);
    fetch(1);
#pragma unroll 1
    for (int left = mbh - 1; left > 0; --left) {
        take(0);
        row[0] = min(row[0] + kCellPitch * 2, r1[0]);
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);          // bin 0's next row is requested BEFORE bin 1's maxima are taken
        take(1);
        row[1] = min(row[1] + kCellPitch * 2, r1[1]);
        fetch(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    take(0);
    take(1);
}

// kRows = 38: one 1024-thread workgroup per CU (16 waves share one 78 KB image of the eight planes; each wave has a 1.5 KB LDS slot
// in which a RoI's [8][7][7] block is assembled, so it leaves as ONE contiguous float4-coalesced run: full 128-byte lines, 2 store
// instructions per RoI instead of 8 scattered ones -- the direct stores cost ~4 us of the r02a kernel's 17).  kRows = 76 (taller
// maps): 155 KB image, 8 waves, direct stores.
// OUT16: 0 = fp32, 1 = raw bf16 (one rounding of the fp32 maximum), 2 = the three bf16 terms of the fp32 maximum ([3][R][C*bins]: the
// split tensor the fully connected layers of conv_f32s.hip read -- h + m + l is the fp32 value, exactly)
// IN16: the map is the bf16 chain's channel-blocked tensor [CP/16][H][W][16] (x reinterpreted): a cell's eight channels are ONE 16-byte
// load (three loads per lane for the 38-row image instead of 24) and no fp32 NCHW copy of the map has to exist.
template <int kRows, int OUT16, bool IN16 = false>
__global__ void __launch_bounds__(kRows <= 38 ? 1024 : 512)
roi_pool_cells_kernel(const float *__restrict__ x, int C, int H, int W, const float *__restrict__ rois, int roi_cols, int R,
                      int outh, int outw, float scale, float *__restrict__ y, int rsplit, const RoiBinTables tables, int dbg_arg) {
#ifdef FRCNN_TIMING_ABLATIONS                    // tuning builds only (scripts/roi_ablate.py): 2 prologue only, 4 no scan, 8 no output -- WRONG results
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    constexpr int kCellWaves = kRows <= 38 ? 16 : 8;
    constexpr bool STAGED = kRows <= 38;
    __shared__ __attribute__((aligned(16))) float4 cells[kRows * kCellPitch * 2];
    __shared__ __attribute__((aligned(16))) RoiBinTables tb;
    __shared__ __attribute__((aligned(16))) float stage[STAGED ? kCellWaves : 1][STAGED ? 8 * kMaxBins : 4];
    // Per-RoI geometry is computed ONCE per workgroup (by 8 threads per RoI, while the map loads are in flight) instead of by every
    // wave that picks the RoI up: the pass itself starts from three LDS reads.  geo_w[slot][pw] = 2 c_lo | 2 c_hi << 8 | empty << 16
    // (first / last column of the bin as float4 indices), geo_h[slot][ph] = r0 | empty << 15 | r1 << 16 (first / last row as float4
    // indices), geo_m = widest | tallest << 8 bin of the RoI; entries past outw / outh repeat the last bin (the idle lanes shadow it:
    // same addresses -> LDS broadcast).  The waves draw slots with an atomic counter, one draw AHEAD: the next
    // slot's number and geometry are fetched while the current RoI is scanned, so a pass does not start with a chain of dependent
    // LDS round trips (longest-first ordering of the slots measured 0.3 us slower than arrival order and was dropped).
    constexpr int kGeoSlots = kRows <= 38 ? 128 : 32;        // 8 threads per slot: one pass of the workgroup (the 76-row image leaves 5 KB)
    __shared__ uint32_t geo_w[kGeoSlots][8];
    __shared__ uint32_t geo_h[kGeoSlots][8];
    __shared__ uint16_t geo_m[kGeoSlots];
    __shared__ int next_roi;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W, bins = outh * outw;
    const int c0 = blockIdx.x * 8;
    const int g0 = blockIdx.y;
    const int n_slots = g0 < R ? (R - g0 + rsplit - 1) / rsplit : 0;        // this workgroup's RoIs: g0 + slot * rsplit
    constexpr int kThreads = 64 * kCellWaves;
    static_assert(kGeoSlots * 8 <= kThreads, "one thread per (slot, bin index)");
    const int my_sl = tid >> 3, my_p = tid & 7;

    auto load_roi = [&](int base) -> float4 {                             // x1, y1, x2, y2 of this thread's slot
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (my_sl < kGeoSlots && base + my_sl < n_slots) {
            const float *pr = rois + (size_t)roi_cols * (g0 + (base + my_sl) * rsplit) + (roi_cols - 4);
            q = make_float4(pr[0], pr[1], pr[2], pr[3]);
        }
        return q;
    };
    auto build_geometry = [&](int base, const float4 q) {
        const int nb = min(n_slots - base, kGeoSlots);
        if (my_sl < nb) {
            const int sl = my_sl, p = my_p;
            // Python round() on the float32 product = round-half-to-even (roi_geometry)
            const int xs = (int)rintf(q.x * scale), ys = (int)rintf(q.y * scale);
            const int rw = max((int)rintf(q.z * scale) - xs + 1, 1), rh = max((int)rintf(q.w * scale) - ys + 1, 1);
            const int pw = min(p, outw - 1), ph = min(p, outh - 1);
            int ws, we, hs, he, mbw, mbh;
            if (rw < kTabExt && rh < kTabExt) {
                const int tw = tb.tab[1][rw][pw], th = tb.tab[0][rh][ph];
                ws = min(max((tw & 255) + xs, 0), W); we = min(max((tw >> 8) + xs, 0), W);
                hs = min(max((th & 255) + ys, 0), H); he = min(max((th >> 8) + ys, 0), H);
                mbw = tb.tabmax[1][rw]; mbh = tb.tabmax[0][rh];
            } else {                                                        // huge RoI: the arithmetic itself, trip counts = the map
                bin_range(pw, rw, outw, xs, W, ws, we);
                bin_range(ph, rh, outh, ys, H, hs, he);
                mbw = W; mbh = H;
            }
            mbw = min(mbw, W); mbh = min(mbh, H);
            // clamped into the map so that an empty bin still reads valid cells
            const int c_lo = min(ws, W - 1), c_hi = min(max(we - 1, c_lo), W - 1);
            const int h0 = min(max(hs, 0), H - 1), h1 = min(max(he - 1, h0), H - 1);
            geo_w[sl][p] = (uint32_t)(c_lo * 2) | ((uint32_t)(c_hi * 2) << 8) | ((we <= ws) ? 1u << 16 : 0u);
            geo_h[sl][p] = (uint32_t)(h0 * (kCellPitch * 2)) | ((he <= hs) ? 1u << 15 : 0u) | ((uint32_t)(h1 * (kCellPitch * 2)) << 16);
            if (p == 0) geo_m[sl] = (uint16_t)(mbw | (mbh << 8));
        }
        if (tid == 0) next_roi = 0;
        __syncthreads();
    };

    // ---- the first batch's RoIs, then 8 channel planes -> cells: one map row per wave and step (coalesced NCHW rows) through a
    //      buffer descriptor -- rows past H, columns past W and channels past C are out-of-range offsets that load 0: no branches;
    //      all loads are issued before the table copy to LDS and the geometry pass below and land during them
    const float4 roi_first = load_roi(0);
    // the bin tables (a by-value kernel argument, 2.4 KB): their loads go out BEFORE the map's -- loads return in order, so issued after
    // them the copy to LDS (and with it the geometry pass) waited for the whole map to land
    constexpr int kTabWords = (int)(sizeof(RoiBinTables) / 4), kTabPerThread = (kTabWords + 64 * kCellWaves - 1) / (64 * kCellWaves);
    uint32_t tabv[kTabPerThread];
#pragma unroll
    for (int q = 0; q < kTabPerThread; ++q) {
        const int e = tid + q * 64 * kCellWaves;
        tabv[q] = e < kTabWords ? reinterpret_cast<const uint32_t *>(&tables)[e] : 0u;
    }
    constexpr int kRowsPerWave = (kRows + kCellWaves - 1) / kCellWaves;
    const frcnn_buf_t xbuf = frcnn_make_buf(x, IN16 ? (uint32_t)((size_t)((C + 15) / 16 * 16) * HW * 2) : (uint32_t)((size_t)C * HW * sizeof(float)));
    float v[kRowsPerWave][8];
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
        const int h = wave + i * kCellWaves;
        if constexpr (IN16) {
            // channels c0 .. c0 + 7 of pixel (h, lane): half (c0 / 8) & 1 of channel block c0 / 16 (channels past C are the tensor's zero padding)
            const uint32_t off = (h < H && lane < W) ? (uint32_t)((((c0 >> 4) * HW + h * W + lane) * 32) + ((c0 >> 3) & 1) * 16) : kBufOob;
            const float4 q = frcnn_buf_load_f32x4(xbuf, off);
            const uint32_t u[4] = {__float_as_uint(q.x), __float_as_uint(q.y), __float_as_uint(q.z), __float_as_uint(q.w)};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[i][2 * c] = frcnn_h16_to_f32((uint16_t)u[c]);               // (bf16: u << 16 and u & 0xffff0000 -- the compiler folds the casts; fp16 twin: v_cvt_f32_f16)
                v[i][2 * c + 1] = frcnn_h16_to_f32((uint16_t)(u[c] >> 16));
            }
        } else {
            const uint32_t base = (h < H && lane < W) ? (uint32_t)((c0 * HW + h * W + lane) * 4) : kBufOob;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[i][c] = frcnn_buf_load_f32(xbuf, base + (uint32_t)(c * HW * 4));   // c0 + c >= C: past the end -> 0
        }
    }
#pragma unroll
    for (int q = 0; q < kTabPerThread; ++q) {
        const int e = tid + q * kThreads;
        if (e < kTabWords) reinterpret_cast<uint32_t *>(&tb)[e] = tabv[q];
    }
    __syncthreads();
    build_geometry(0, roi_first);
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
        const int h = wave + i * kCellWaves;
        if (h < H && lane < W) {
            float4 *dst = cells + (h * kCellPitch + lane) * 2;
            dst[0] = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
            dst[1] = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
        }
    }
    __syncthreads();

    // ---- lane -> (pg, cq, pw): pg = the hardware's ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32)
    const int l5 = lane & 31;
    const bool in_a = (l5 < 4) || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    const int j = in_a ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12)) : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16));
    const int pg = (lane >> 5) * 2 + (in_a ? 0 : 1);
    const int cq = j & 1;
    const int pw = j >> 1;                          // 0 .. 7; bins past outw - 1 (and rows past outh - 1) shadow the last one
    const bool pw_on = pw < outw;

    if (dbg & 2) return;                                                 // ablation: prologue only
    for (int base = 0; base < n_slots; base += kGeoSlots) {
    if (base > 0) { __syncthreads(); build_geometry(base, load_roi(base)); }
    const int nb = min(n_slots - base, kGeoSlots);
    auto grab = [&]() -> int {
        int d = 0;
        if (lane == 0) d = atomicAdd(&next_roi, 1);
        return __builtin_amdgcn_readfirstlane(d);
    };
    int draw = grab();
    uint32_t gw_n = 0u, gm_n = 0u;
    uint2 gh_n = make_uint2(0u, 0u);
    auto fetch_geo = [&](int d) {
        if (d < nb) {
            gw_n = geo_w[d][pw];
            gh_n = *reinterpret_cast<const uint2 *>(&geo_h[d][2 * pg]);
            gm_n = geo_m[d];
        }
    };
    fetch_geo(draw);
    for (;;) {
        if (draw >= nb) break;
        const int sl = draw;
        const int r = g0 + (base + sl) * rsplit;
        const uint32_t gw = gw_n;
        const uint2 gh = gh_n;
        const int gm = __builtin_amdgcn_readfirstlane((int)gm_n);
        draw = grab();                                                   // one draw ahead
        fetch_geo(draw);
        const int mbw = (dbg & 4) ? 0 : (gm & 255), mbh = gm >> 8;           // (ablation 4: no scan)
        const int c_lo2 = (int)(gw & 255u), c_hi2 = (int)((gw >> 8) & 255u);
        const int c_first = c_lo2 + cq;
        int r0[2], r1[2];
        r0[0] = (int)(gh.x & 0x7fffu); r1[0] = (int)(gh.x >> 16);
        r0[1] = (int)(gh.y & 0x7fffu); r1[1] = (int)(gh.y >> 16);
        const bool empty_w = (gw >> 16) & 1u;
        const bool empty_h[2] = {(bool)((gh.x >> 15) & 1u), (bool)((gh.y >> 15) & 1u)};
        const bool any_empty = empty_w || empty_h[0] || empty_h[1];
        // the bin's first cell seeds the maximum (the oracle's scan order)
        float4 acc[2];
        float nan_probe = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[s] = cells[r0[s] + c_first];
            nan_probe += (acc[s].x + acc[s].y) + (acc[s].z + acc[s].w);     // NaN iff one of the eight first cells is
        }
#pragma unroll 1
        for (int k0 = 0; k0 < mbw; k0 += 4) {
            const int nc = min(mbw - k0, 4);                            // wave-uniform
            if (nc >= 4) cells_scan_chunk<4>(cells, k0, c_lo2, c_hi2, cq, r0, r1, mbh, acc);
            else if (nc == 3) cells_scan_chunk<3>(cells, k0, c_lo2, c_hi2, cq, r0, r1, mbh, acc);
            else if (nc == 2) cells_scan_chunk<2>(cells, k0, c_lo2, c_hi2, cq, r0, r1, mbh, acc);
            else cells_scan_chunk<1>(cells, k0, c_lo2, c_hi2, cq, r0, r1, mbh, acc);
        }
        // rare fix-ups behind ONE wave-level test: a NaN first cell stays (`>` never replaces it: the sum of the eight first cells
        // is NaN iff one of them is; they are re-read then), an empty bin is 0
        if (__any((nan_probe != nan_probe) || any_empty)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool empty = empty_h[s] || empty_w;
                const float4 first = cells[r0[s] + c_first];
                if (first.x != first.x) acc[s].x = first.x;
                if (first.y != first.y) acc[s].y = first.y;
                if (first.z != first.z) acc[s].z = first.z;
                if (first.w != first.w) acc[s].w = first.w;
                if (empty) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (dbg & 8) {                                                   // ablation: no output (keep the maxima alive)
            if (acc[0].x + acc[1].y + acc[0].z + acc[1].w == 12345.678f) y[0] = acc[0].x;
            continue;
        }
        if constexpr (STAGED) {
            float *sv = stage[wave];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ph = 2 * pg + s;
                const float o[4] = {acc[s].x, acc[s].y, acc[s].z, acc[s].w};
                if (pw_on && ph < outh) {
                    float *d = sv + (4 * cq) * bins + ph * outw + pw;
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[q * bins] = o[q];
                }
            }
            frcnn_wave_sync();     // the wave's own LDS writes above are read by other lanes below (DS ops of a wave are in order)
            // (r, c0 .. c0 + cg, :, :) is one contiguous run of cg*bins floats of y
            const int cg = min(8, C - c0), run = cg * bins;
            const size_t dst = ((size_t)r * C + c0) * bins;
            if constexpr (OUT16 == 2) {
                uint16_t *y16 = reinterpret_cast<uint16_t *>(y);
                const size_t part = (size_t)R * C * bins;
                if ((run & 3) == 0 && (dst & 3) == 0) {
                    for (int i = lane; i < run / 4; i += 64) {
                        const float4 v4 = reinterpret_cast<const float4 *>(sv)[i];
                        uint32_t h0, m0, l0, h1, m1, l1;
                        frcnn_split3_pair(v4.x, v4.y, h0, m0, l0);
                        frcnn_split3_pair(v4.z, v4.w, h1, m1, l1);
                        reinterpret_cast<uint2 *>(y16 + dst)[i] = make_uint2(h0, h1);
                        reinterpret_cast<uint2 *>(y16 + part + dst)[i] = make_uint2(m0, m1);
                        reinterpret_cast<uint2 *>(y16 + 2 * part + dst)[i] = make_uint2(l0, l1);
                    }
                } else {
                    for (int i = lane; i < run; i += 64) {
                        uint32_t h, m, l;
                        frcnn_split3_pair(sv[i], 0.0f, h, m, l);
                        y16[dst + i] = (uint16_t)h; y16[part + dst + i] = (uint16_t)m; y16[2 * part + dst + i] = (uint16_t)l;
                    }
                }
            } else if constexpr (OUT16 == 1) {
                uint16_t *y16 = reinterpret_cast<uint16_t *>(y);
                if ((run & 3) == 0 && (dst & 3) == 0) {
                    for (int i = lane; i < run / 4; i += 64) {
                        const float4 v4 = reinterpret_cast<const float4 *>(sv)[i];
                        uint2 pk;
                        pk.x = frcnn_pack_bf16x2(v4.x, v4.y);
                        pk.y = frcnn_pack_bf16x2(v4.z, v4.w);
                        reinterpret_cast<uint2 *>(y16 + dst)[i] = pk;
                    }
                } else {
                    for (int i = lane; i < run; i += 64) y16[dst + i] = (uint16_t)roi_f32_to_bf16(sv[i]);
                }
            } else if ((run & 3) == 0 && (dst & 3) == 0) {
                for (int i = lane; i < run / 4; i += 64) reinterpret_cast<float4 *>(y + dst)[i] = reinterpret_cast<const float4 *>(sv)[i];
            } else {
                for (int i = lane; i < run; i += 64) y[dst + i] = sv[i];
            }
            frcnn_wave_sync();     // the slot is rewritten by the next RoI only after these reads were issued
        } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ph = 2 * pg + s;
            const float o[4] = {acc[s].x, acc[s].y, acc[s].z, acc[s].w};
            if (pw_on && ph < outh) {
                const int cbase = c0 + 4 * cq;
                const size_t dst = ((size_t)r * C + cbase) * bins + ph * outw + pw;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (cbase + q < C) {
                        if constexpr (OUT16 == 2) {
                            uint32_t h, m, l;
                            frcnn_split3_pair(o[q], 0.0f, h, m, l);
                            uint16_t *y16 = reinterpret_cast<uint16_t *>(y) + dst + (size_t)q * bins;
                            const size_t part = (size_t)R * C * bins;
                            y16[0] = (uint16_t)h; y16[part] = (uint16_t)m; y16[2 * part] = (uint16_t)l;
                        } else if constexpr (OUT16 == 1) reinterpret_cast<uint16_t *>(y)[dst + (size_t)q * bins] = (uint16_t)roi_f32_to_bf16(o[q]);
                        else y[dst + (size_t)q * bins] = o[q];
                    }
                }
            }
        }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Quad-cell forward (round 3; the default when the map fits 38 x 64).  What round 2's counters and round 3's micro-measurements
// (scripts/micro/store_micro.hip) say about this operation on the MI355X:
//   * a launch in a chain of graph nodes costs 2.7 us before it does anything, and 30.1 MB of output cost 3.6 us more with
//     write-through (sc1) 16-byte stores -- but 4.5 us with plain stores: the dirty lines sit in the 32 MB of L2 and are written back
//     AFTER the last wave has finished (the "parts that add up" of round 2's ablation), and 7-10 us as direct 4-byte stores;
//   * the prologue is pure latency: 76 KB of map per CU cost 1.05 us, 38 KB 0.6 us;
//   * everything in between is VALU and LDS issue (4 clocks per VALU instruction and wave, 4+ per ds_read_b128; SQ counters): the
//     cell-major kernel spent 219 VALU instructions and ~24 LDS reads per (RoI, 8 channels), 37 % of its LDS cycles on bank conflicts.
// So: a workgroup holds FOUR channels of the map as 16-byte cells (38 KB; one ds_read_b128 = the four channels of one cell), a lane
// owns ONE bin (ph, pw) of the RoI its wave is working on and all four channels of it, and the address arithmetic is gone:
//   * the bins of one RoI differ in width (height) by at most one cell -- floor / ceil of multiples of the same stride -- so with
//     mbw (mbh) the RoI's widest (tallest) bin, EVERY lane reads the same number of columns: c0, c0 + tw, ... (immediate offsets of
//     the ds_read: the count is wave-uniform, one unrolled body per value) and its own last column; likewise rows.  A duplicate read
//     cannot change a maximum and nothing outside the bin is ever read.  Per pass a lane builds four base addresses
//     (first | last row) + (first | last column) from two packed LDS words and advances two of them per row step;
//   * a second image holds T11[h][w] = max of the 2 x 2 cells at (h, w): RoIs whose bins are all at least 2 x 2 (extent >= 8 cells
//     both ways: most) are covered with 2 x 2 tiles -- a quarter of the reads and of the maxima (3.5 instead of 8.5 reads per pass on
//     the benchmark's RoIs);
//   * lanes are dealt to bins by the hardware's ds_read_b128 lane groups: the 16 lanes that share an LDS cycle hold two adjacent bin
//     rows (8 columns each, the eighth shadows the seventh), which with the odd row pitch keeps them on different banks most of the
//     time (simulated 1.36 cycles per group instead of 2.04 for lane = 8 ph + pw; scripts/micro/lds_sim.py);
//   * RoIs that do not fit the pattern (bins emptied or narrowed by the map's edge) and maps that hold a NaN take the general path
//     (saturating addresses on the plain image, first-cell rule restored): correct, not fast, and rare;
//   * bin edges are the oracle's double arithmetic (bin_range), ONE thread per RoI on three waves that do nothing else, while the
//     other thirteen fetch the map (three rows + one halo row each) and build the tile image from registers (the horizontal pair
//     maximum is a DPP wave shift);
//   * the RoI's [4][7][7] block is assembled in the wave's 784-byte LDS slot and leaves as ONE run of 16-byte write-through stores
//     through a buffer descriptor (scalar base: no address VALU).
// Scan order = the oracle's: the bin's first cell seeds the maximum, later NaNs never win, a NaN first cell stays (general path).
constexpr int kQuadPitch = 65;           // cells per LDS row: odd, so vertically adjacent cells sit 16 B apart in the bank pattern
constexpr int kQuadRows = 38;
constexpr int kQuadImgBytes = kQuadRows * kQuadPitch * 16;
constexpr int kQuadWaves = 16;
constexpr int kQuadLoadWaves = 13;       // waves 0 .. 12 fetch three map rows each (+ one halo row)
constexpr int kQuadProducers = kQuadWaves - kQuadLoadWaves;              // waves 13 .. 15 build the RoI geometry, one RoI per lane
constexpr int kQuadGeoSlots = 64 * kQuadProducers;                       // RoIs per geometry batch
constexpr int kQuadTabExt = 64;
// LDS layout (one block, carved by hand: ds_append addresses its counter through M0[15:0], so the counter must sit below 64 KB --
// the compiler put a separate __shared__ int behind the images and the hardware wrapped the address into them)
constexpr int kQuadOffCtr = 0;                                           // int: draw counter; bytes 8 .. 23: a NaN flag per wave
constexpr int kQuadOffGeoM = 32;                                         // uint2 [slots]: column word, row word
constexpr int kQuadOffGeoW = kQuadOffGeoM + kQuadGeoSlots * 8;           // uint32 [slots][8]
constexpr int kQuadOffGeoH = kQuadOffGeoW + kQuadGeoSlots * 32;          // uint2 [slots][8]
constexpr int kQuadOffStage = kQuadOffGeoH + kQuadGeoSlots * 64;         // float [waves][2][4 * kMaxBins]
constexpr int kQuadStageFloats = 4 * kMaxBins;                          // one RoI's [4][7][7] block
// (the arg-max form stages values AND indices: twice the slots)
constexpr int quad_off_img(bool argmax) { return (kQuadOffStage + kQuadWaves * (argmax ? 4 : 2) * kQuadStageFloats * 4 + 15) / 16 * 16; }   // float4 [2][rows * pitch]
constexpr int quad_off_tab(bool argmax) { return quad_off_img(argmax) + 2 * kQuadImgBytes; }      // uint16 [producers][2][ext][8]: each producer's copy of the edge rows
constexpr int quad_lds_bytes(bool argmax) { return quad_off_tab(argmax) + kQuadProducers * 2 * kQuadTabExt * 16; }
static_assert(quad_off_img(false) % 16 == 0 && quad_off_img(true) % 16 == 0 && kQuadOffStage % 16 == 0, "16-byte LDS accesses");
static_assert(quad_lds_bytes(true) <= 160 * 1024, "LDS");

// Bin edges per extent, built on the HOST with the oracle's double arithmetic -- floor(p * (e / out)), ceil((p + 1) * (e / out)): IEEE
// double division / multiplication / floor / ceil are correctly rounded, so the host's values are the device's (7 * (29 / 7.) =
// 29.000000000000004 -> ceil 30 included) -- and handed to the kernel by value (2 KB of kernel arguments).  Row = seven bins as
// lo | hi << 8 and a meta word = largest bin | last hi << 8.  Extents of 64 and more (a RoI larger than the map) take the double
// arithmetic itself on the device.
struct RoiEdgeTable {
    uint16_t row[2][kQuadTabExt][8];     // [0] columns (outw), [1] rows (outh)
};

// The waves of a workgroup draw RoIs from an LDS counter with ds_append: ONE wave-level operation that adds the number of active
// lanes (64: the draw loop runs with every lane on) and returns the old value -- no single-lane branch and none of the ten-instruction
// mbcnt sequence `if (lane == 0) atomicAdd(...)` compiles to.  The counter therefore counts in units of 64.
__device__ __forceinline__ int quad_draw(unsigned char *lds) {
    return frcnn_lds_append(reinterpret_cast<int *>(lds + kQuadOffCtr)) >> 6;
}
__device__ __forceinline__ float4 quad_cell(const unsigned char *lds, uint32_t off) { return *reinterpret_cast<const float4 *>(lds + off); }

// One RoI's bin for this lane: gw / gh = the lane's column and row words, mw / mh = the RoI's (wave-uniform) meta words.
__device__ __forceinline__ float4 quad_scan_slot(const unsigned char *img, uint32_t gw, uint2 gh, uint32_t mw, uint32_t mh) {
    const int nc = (int)(mw & 255u), nr = (int)(mh & 255u);
    float4 acc;
    if (!(mw >> 17)) {
        // tiles: columns cl, cl + tw, ... and the bin's own last tile column ch; rows likewise (a / b advance, c / d = last row).
        // A duplicate read cannot change a maximum.
        const uint32_t t2 = (mw >> 16) & 1u;
        const uint32_t cl = gw & 0xffffu, ch = gw >> 16;
        uint32_t a = gh.x + cl, b = gh.x + ch;
        const uint32_t c = gh.y + cl, d = gh.y + ch;
        const uint32_t cstep = 16u << t2, rstep = (uint32_t)(kQuadPitch * 16) << t2;
        acc = max4(quad_cell(img, c), quad_cell(img, d));
#pragma unroll 1
        for (int k = 1; k < nc - 1; ++k) acc = max4(acc, quad_cell(img, c + k * cstep));
#pragma unroll 1
        for (int j = 1; j < nr; ++j) {
            acc = max4_3(acc, quad_cell(img, a), quad_cell(img, b));
#pragma unroll 1
            for (int k = 1; k < nc - 1; ++k) acc = max4(acc, quad_cell(img, a + k * cstep));
            a += rstep; b += rstep;
        }
    } else {
        // general path: saturating rows / columns over the RoI's largest bin shape on the plain cells; the first cell seeds the
        // maximum and a NaN first cell stays (`>` never replaces it), an empty bin is 0
        const uint32_t cl = gw & 0x7fffu, ch = (gw >> 16) & 0x7fffu, r0 = gh.x, r1 = gh.y & 0x7fffffffu;
        const bool empty = ((gw >> 15) & 1u) || (gh.y >> 31);
        const float4 first = quad_cell(img, r0 + cl);
        acc = first;
#pragma unroll 1
        for (int j = 0; j < nr; ++j) {
            const uint32_t rr = min(r0 + (uint32_t)j * (kQuadPitch * 16), r1);
#pragma unroll 1
            for (int k = 0; k < nc; ++k) acc = max4(acc, quad_cell(img, rr + min(cl + 16u * k, ch)));
        }
        if (first.x != first.x) acc.x = first.x;
        if (first.y != first.y) acc.y = first.y;
        if (first.z != first.z) acc.z = first.z;
        if (first.w != first.w) acc.w = first.w;
        if (empty) acc = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return acc;
}

// The training form: maximum AND the position of the FIRST maximum (Chainer's argmax_data: h * W + w, -1 for an empty bin), exactly the
// oracle's scan -- the bin's first cell seeds both, then strict `>` in row-major order.  Always on the plain cells with saturating
// rows / columns over the RoI's largest bin shape: a cell read twice cannot displace itself under `>`, a NaN never wins and a NaN first
// cell stays.  `pos` is tracked as the cell's byte offset in the LDS image and converted at the end.
__device__ __forceinline__ void quad_scan_slot_argmax(const unsigned char *img, uint32_t gw, uint2 gh, uint32_t mw, uint32_t mh, int W,
                                                      float4 &val, int4 &idx) {
    const int nc = (int)(mw & 255u), nr = (int)(mh & 255u);
    const uint32_t cl = gw & 0x7fffu, ch = (gw >> 16) & 0x7fffu, r0 = gh.x, r1 = gh.y & 0x7fffffffu;
    const bool empty = ((gw >> 15) & 1u) || (gh.y >> 31);
    float4 m = quad_cell(img, r0 + cl);
    uint32_t px = r0 + cl, py = px, pz = px, pw_ = px;
#pragma unroll 1
    for (int j = 0; j < nr; ++j) {
        const uint32_t rr = min(r0 + (uint32_t)j * (kQuadPitch * 16), r1);
#pragma unroll 1
        for (int k = 0; k < nc; ++k) {
            const uint32_t a = rr + min(cl + 16u * k, ch);
            const float4 v = quad_cell(img, a);
            if (v.x > m.x) { m.x = v.x; px = a; }
            if (v.y > m.y) { m.y = v.y; py = a; }
            if (v.z > m.z) { m.z = v.z; pz = a; }
            if (v.w > m.w) { m.w = v.w; pw_ = a; }
        }
    }
    auto to_index = [&](uint32_t off) -> int {
        const uint32_t q = off >> 4, h = (q * 64528u) >> 22;               // q / 65 for q < 2^18 (64528 * 65 = 2^22 + 16)
        return (int)(q - h * (uint32_t)(kQuadPitch - W));                  // h * W + (q - 65 h)
    };
    static_assert(kQuadPitch == 65, "the division constant above");
    val = empty ? make_float4(0.f, 0.f, 0.f, 0.f) : m;
    idx = empty ? make_int4(-1, -1, -1, -1) : make_int4(to_index(px), to_index(py), to_index(pz), to_index(pw_));
}

// BINS: outh * outw as a compile-time constant (49 for the 7 x 7 head: the staging stores get immediate offsets and pair up as
// ds_write2_b32), or 0 = any shape up to 7 x 7
// ARGMAX: the training form -- also writes argmax_data (int32, same shape as y); every RoI takes the plain-cell scan above.
// IN16 (round 4): the map is the bf16 chain's channel-blocked tensor [CP/16][H][W][16] -- a cell's four channels are ONE 8-byte load, widened to fp32 on
// the way into the LDS images (exact), everything after that is the fp32 kernel; OUT16: the output is raw bf16 bits (R, C * bins) -- the staged fp32 maxima
// are rounded once (v_cvt_pk_bf16_f32; of bf16 inputs: exact) and leave as 8-byte stores.  The bf16 line's RoI stage used the round-2 cell-major kernel (13.9 us).
template <int ST, int BINS, bool ARGMAX = false, bool IN16 = false, bool OUT16 = false>
__global__ void __launch_bounds__(64 * kQuadWaves)
roi_pool_quads_kernel(const float *__restrict__ x, int C, int H, int W, const float *__restrict__ rois, int roi_cols, int R,
                      int outh, int outw, float scale, float *__restrict__ y, int32_t *__restrict__ argmax, int rsplit, const RoiEdgeTable etab,
                      int dbg_arg) {
    constexpr int kQuadOffImg = quad_off_img(ARGMAX), kQuadOffTab = quad_off_tab(ARGMAX), kQuadLdsBytes = quad_lds_bytes(ARGMAX);
#ifdef FRCNN_TIMING_ABLATIONS                    // tuning builds only (scripts/micro): 2 prologue only, 4 no scan, 8 no staging / output, 16 no stores -- WRONG results
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    __shared__ __attribute__((aligned(16))) unsigned char lds[kQuadLdsBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W, bins = BINS ? BINS : outh * outw;
    const int c0 = blockIdx.x * 4;
    const int g0 = blockIdx.y;
    const int n_slots = g0 < R ? (R - g0 + rsplit - 1) / rsplit : 0;        // this workgroup's RoIs: g0 + slot * rsplit
    int *const ctr = reinterpret_cast<int *>(lds + kQuadOffCtr);
#ifdef FRCNN_TIMING_ABLATIONS
    // dbg & 128: per-wave s_memtime stamps into y (use with dbg & 16: no output stores): [wg][wave][32] uint32
    uint32_t *stamps = reinterpret_cast<uint32_t *>(y) + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kQuadWaves + wave) * 32;
    int n_stamp = 0;
    auto stamp = [&]() { if ((dbg & 128) && n_stamp < 32) { const uint64_t t = __builtin_amdgcn_s_memtime(); if (lane == 0) stamps[n_stamp] = (uint32_t)t; ++n_stamp; } };
#else
    auto stamp = [&]() {};
#endif
    stamp();

    // ---- RoI geometry: the three waves that fetch no map rows, ONE LANE PER RoI, all fourteen bin entries of it in sequence.  (Eight
    //      lanes per RoI, one bin each, repeats the per-RoI arithmetic eight times: 100 VALU instructions per eight RoIs against 110 per
    //      sixty-four here, and instruction issue is what the prologue waits for.)  The lane reads its extents' two 16-byte edge rows
    //      from the wave's own LDS copy of the host table (no workgroup barrier between the copy and the look-up: DS operations of one
    //      wave are in order) and writes its 8 + 8 entries as six 16-byte stores.
    //   fast form (no bin touches the map's edge, so bin_range()'s clamps do nothing and every bin has its unclamped size >= 1):
    //     geo_w[p] = byte offset of the bin's first tile column | of its last tile column << 16; geo_h[p] = byte offsets of its first and
    //     last tile row, the image's offset included; meta = tile columns | 2 x 2 tiles << 16, tile rows
    //   general form (some bin clamped, or the map holds a NaN): plain cells, clamped into the map, bit 15 / bit 31 = bin empty;
    //     meta = widest bin | 1 << 17, tallest bin
    const int pk = wave - kQuadLoadWaves;                                    // producer index, < 0 on the map waves
    uint4 *const my_tab = reinterpret_cast<uint4 *>(lds + kQuadOffTab) + max(pk, 0) * (2 * kQuadTabExt);
    auto roi_of_lane = [&](int base) -> float4 {
        const int sl = pk * 64 + lane;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base + sl < n_slots) {
            const float *pr = rois + (size_t)roi_cols * (g0 + (base + sl) * rsplit) + (roi_cols - 4);
            q = make_float4(pr[0], pr[1], pr[2], pr[3]);
        }
        return q;
    };
    auto build_geometry = [&](int base, const float4 q, bool force_general) __attribute__((always_inline)) {
        const int sl = pk * 64 + lane;
        if (base + sl >= n_slots) return;
        int xs = (int)rintf(q.x * scale), ys = (int)rintf(q.y * scale);            // roi_geometry(): round-half-even of the fp32 product
        const int rw = max((int)rintf(q.z * scale) - xs + 1, 1), rh = max((int)rintf(q.w * scale) - ys + 1, 1);
        uint32_t rwq[4], rhq[4];                                             // the two edge rows: seven bins lo | hi << 8, then the meta word
        if (rw < kQuadTabExt && rh < kQuadTabExt) {
            const uint4 a = my_tab[rw], b = my_tab[kQuadTabExt + rh];
            rwq[0] = a.x; rwq[1] = a.y; rwq[2] = a.z; rwq[3] = a.w; rhq[0] = b.x; rhq[1] = b.y; rhq[2] = b.z; rhq[3] = b.w;
        } else {                                                           // a RoI larger than the map: bin_range()'s arithmetic itself.
            // The row's eight bits per edge cannot hold an offset from a far-away origin (round 5 clamped the OFFSET at 255 and then added
            // xs: a RoI reaching from -10^4 px lost every bin), so the edges are clamped into the map HERE -- 64-bit sums, exact for every
            // RoI in the header's stated domain -- and stored relative to the origin 0.
            const double sw = (double)rw / (double)outw, sh = (double)rh / (double)outh;
#pragma unroll
            for (int i = 0; i < 4; ++i) { rwq[i] = 0u; rhq[i] = 0u; }
#pragma unroll
            for (int p = 0; p < 7; ++p) {
                const int pw = min(p, outw - 1), ph = min(p, outh - 1);
                const long long LW = W, LH = H;
                const int lw = (int)min(max((long long)floor((double)pw * sw) + xs, 0ll), LW), hw = (int)min(max((long long)ceil((double)(pw + 1) * sw) + xs, 0ll), LW);
                const int lh = (int)min(max((long long)floor((double)ph * sh) + ys, 0ll), LH), hh = (int)min(max((long long)ceil((double)(ph + 1) * sh) + ys, 0ll), LH);
                rwq[p >> 1] |= (uint32_t)(lw | (hw << 8)) << (16 * (p & 1));
                rhq[p >> 1] |= (uint32_t)(lh | (hh << 8)) << (16 * (p & 1));
            }
            xs = 0; ys = 0;
            rwq[3] |= 0xffffu << 16; rhq[3] |= 0xffffu << 16;                // meta: largest bin 255, last hi 255 -> never "fast"; bounds only
        }
        const int mbw = (int)((rwq[3] >> 16) & 255u), hlw = (int)(rwq[3] >> 24), mbh = (int)((rhq[3] >> 16) & 255u), hlh = (int)(rhq[3] >> 24);
        const bool fast = !ARGMAX && xs >= 0 && ys >= 0 && xs + hlw <= W && ys + hlh <= H && !force_general;
        const bool t2 = rw > outw && rh > outh;                            // stride > 1 both ways <=> every (unclamped) bin is at least 2 x 2
        uint32_t ew[8];
        uint2 eh[8];
        uint2 meta;
        if (fast) {
            const int tw = t2 ? 2 : 1;
            const uint32_t off = t2 ? (uint32_t)kQuadImgBytes : 0u;
            const uint32_t xlo = (uint32_t)(xs * 16), xhi = (uint32_t)((xs - tw) * 16);
            const uint32_t ylo = off + (uint32_t)(ys * (kQuadPitch * 16)), yhi = off + (uint32_t)((ys - tw) * (kQuadPitch * 16));
#pragma unroll
            for (int p = 0; p < 7; ++p) {
                const uint32_t cw = rwq[p >> 1] >> (16 * (p & 1)), rw_ = rhq[p >> 1] >> (16 * (p & 1));
                const uint32_t lw = cw & 255u, hw = (cw >> 8) & 255u, lh = rw_ & 255u, hh = (rw_ >> 8) & 255u;
                ew[p] = (lw * 16u + xlo) | ((hw * 16u + xhi) << 16);
                eh[p] = make_uint2(lh * (uint32_t)(kQuadPitch * 16) + ylo, hh * (uint32_t)(kQuadPitch * 16) + yhi);
            }
            meta = make_uint2((uint32_t)((mbw + tw - 1) >> (tw - 1)) | (t2 ? 1u << 16 : 0u), (uint32_t)((mbh + tw - 1) >> (tw - 1)));
        } else {
#pragma unroll
            for (int p = 0; p < 7; ++p) {
                const uint32_t cw = rwq[p >> 1] >> (16 * (p & 1)), rw_ = rhq[p >> 1] >> (16 * (p & 1));
                const int ws = min(max((int)(cw & 255u) + xs, 0), W), we = min(max((int)((cw >> 8) & 255u) + xs, 0), W);
                const int hs = min(max((int)(rw_ & 255u) + ys, 0), H), he = min(max((int)((rw_ >> 8) & 255u) + ys, 0), H);
                // clamped into the map so that an empty bin still reads valid cells
                const int c_lo = min(ws, W - 1), c_hi = min(max(we - 1, c_lo), W - 1);
                const int h0 = min(hs, H - 1), h1 = min(max(he - 1, h0), H - 1);
                ew[p] = (uint32_t)(c_lo * 16) | ((uint32_t)(c_hi * 16) << 16) | ((we <= ws) ? 1u << 15 : 0u);
                eh[p] = make_uint2((uint32_t)(h0 * (kQuadPitch * 16)), (uint32_t)(h1 * (kQuadPitch * 16)) | ((he <= hs) ? 1u << 31 : 0u));
            }
            meta = make_uint2((uint32_t)min(max(mbw, 1), W) | (1u << 17), (uint32_t)min(max(mbh, 1), H));
        }
        ew[7] = ew[6]; eh[7] = eh[6];                                        // the host table repeats the last bin past out - 1; the eighth entry does too
        uint4 *gw = reinterpret_cast<uint4 *>(lds + kQuadOffGeoW) + sl * 2;
        uint4 *gh = reinterpret_cast<uint4 *>(lds + kQuadOffGeoH) + sl * 4;
        gw[0] = make_uint4(ew[0], ew[1], ew[2], ew[3]); gw[1] = make_uint4(ew[4], ew[5], ew[6], ew[7]);
#pragma unroll
        for (int i = 0; i < 4; ++i) gh[i] = make_uint4(eh[2 * i].x, eh[2 * i].y, eh[2 * i + 1].x, eh[2 * i + 1].y);
        reinterpret_cast<uint2 *>(lds + kQuadOffGeoM)[sl] = meta;
    };

    // ---- prologue.  Waves 0 .. 12: map rows 3 w .. 3 w + 3 (the fourth is the halo of the tile image) of four channel planes through
    //      a buffer descriptor -- rows past H, columns past W, channels past C are out-of-range offsets that load 0.  Producers: their
    //      copy of the edge rows, then their first step.  One barrier.
    if (tid == 0) ctr[0] = 0;
    // NaN flag of the map: one BYTE per wave (bytes 8 .. 23 of the counter block), each written by its own wave only and unconditionally -- no
    // initialisation another wave could overwrite (tid 0's zero raced the map waves' ones before the barrier)
    unsigned char *const nan_of_wave = lds + kQuadOffCtr + 8;
    float4 roi_q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pk >= 0) {
        const uint4 t0 = *reinterpret_cast<const uint4 *>(&etab.row[0][lane][0]), t1 = *reinterpret_cast<const uint4 *>(&etab.row[1][lane][0]);
        roi_q = roi_of_lane(0);
        my_tab[lane] = t0;
        my_tab[kQuadTabExt + lane] = t1;
        frcnn_wave_sync();         // (the rows are read by other lanes of this wave: DS operations of one wave are in order)
        if (!(dbg & 256)) build_geometry(0, roi_q, false);
        if (lane == 0) nan_of_wave[wave] = 0;
    } else {
        float v[4][4];
        if constexpr (IN16) {
            const int CP16 = (C + 15) / 16 * 16;                                  // the blocked tensor is zero-padded to whole 16-channel blocks
            const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)CP16 * HW * 2));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int h = 3 * wave + i;
                const uint32_t off = (h < H && lane < W) ? (uint32_t)((((c0 >> 4) * HW + h * W + lane) * 16 + (c0 & 15)) * 2) : kBufOob;
                const uint2 q = frcnn_buf_load_b64(xbuf, off);                    // channels c0 .. c0 + 3 of the cell: 4 x bf16
                v[i][0] = frcnn_h16_to_f32((uint16_t)q.x); v[i][1] = frcnn_h16_to_f32((uint16_t)(q.x >> 16));
                v[i][2] = frcnn_h16_to_f32((uint16_t)q.y); v[i][3] = frcnn_h16_to_f32((uint16_t)(q.y >> 16));
            }
        } else {
            const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)C * HW * sizeof(float)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int h = 3 * wave + i;
                const uint32_t base = (h < H && lane < W) ? (uint32_t)((c0 * HW + h * W + lane) * 4) : kBufOob;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[i][c] = frcnn_buf_load_f32(xbuf, base + (uint32_t)(c * HW * 4));   // c0 + c >= C: past the end -> 0
            }
        }
        float4 *img0 = reinterpret_cast<float4 *>(lds + kQuadOffImg), *img1 = reinterpret_cast<float4 *>(lds + kQuadOffImg + kQuadImgBytes);
        float t10[4][4];
        float nan_sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // the right-hand neighbour's value: DPP wave_shl:1 (lane l receives lane l + 1)
                const float nb = frcnn_wave_shl1_f32(v[i][c]);
                t10[i][c] = frcnn_max_f32(v[i][c], nb);
                nan_sum += v[i][c];                                      // NaN iff one of the values is (or +inf and -inf meet: the
            }                                                            // general path is merely slower)
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int h = 3 * wave + i;
            if (h < kQuadRows && !(dbg & 1024)) {
                img0[h * kQuadPitch + lane] = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
                if (!(dbg & 512)) img1[h * kQuadPitch + lane] = make_float4(frcnn_max_f32(t10[i][0], t10[i + 1][0]), frcnn_max_f32(t10[i][1], t10[i + 1][1]),
                                                          frcnn_max_f32(t10[i][2], t10[i + 1][2]), frcnn_max_f32(t10[i][3], t10[i + 1][3]));
            }
        }
        const bool wave_nan = __any(nan_sum != nan_sum);
        if (lane == 0) nan_of_wave[wave] = wave_nan ? 1 : 0;
    }
    // lane -> bin by the hardware's ds_read_b128 lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): group g holds the bin
    // rows 2 g and 2 g + 1, eight columns each
    const int l5 = lane & 31;
    const bool in_a = (l5 < 4) || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    const int gi = in_a ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12)) : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16));
    const int ph = 2 * ((lane >> 5) * 2 + (in_a ? 0 : 1)) + (gi >> 3), pw = gi & 7;
    const bool lane_on = ph < outh && pw < outw;
    const int cg = min(4, C - c0), run = cg * bins;
    const bool vec_ok = (run & 3) == 0 && ((C * bins) & 3) == 0 && ((c0 * bins) & 3) == 0;
    static_assert(!(ARGMAX && OUT16), "the training form is fp32");
    constexpr uint32_t kOutBytes = OUT16 ? 2u : 4u;                               // bytes per output value
    const frcnn_buf_t ybuf = frcnn_make_buf(y, (uint32_t)((size_t)R * C * bins * kOutBytes));
    const frcnn_buf_t abuf = frcnn_make_buf(ARGMAX ? (const void *)argmax : (const void *)y, (uint32_t)((size_t)R * C * bins * sizeof(float)));
    float *sv = reinterpret_cast<float *>(lds + kQuadOffStage) + wave * ((ARGMAX ? 4 : 2) * kQuadStageFloats);
    float *sv_lane = sv + min(ph, outh - 1) * outw + min(pw, outw - 1);
    const uint32_t st_off = (vec_ok && lane < run / 4 && !(dbg & 16)) ? (uint32_t)(lane * 4) * kOutBytes : kBufOob;     // four values per lane
    const uint32_t run_bytes = (uint32_t)(C * bins) * kOutBytes, c0_bytes = (uint32_t)(c0 * bins) * kOutBytes;
    const unsigned char *img = lds + kQuadOffImg;
    const uint32_t *geo_w_lane = reinterpret_cast<const uint32_t *>(lds + kQuadOffGeoW) + pw;
    const uint2 *geo_h_lane = reinterpret_cast<const uint2 *>(lds + kQuadOffGeoH) + ph;
    const uint2 *geo_m = reinterpret_cast<const uint2 *>(lds + kQuadOffGeoM);
    stamp();
    __syncthreads();
    const bool nan_map = __any(ctr[2 + (lane & 3)] != 0);                  // the sixteen waves' bytes
    stamp();
    if (dbg & 2) return;

    if (nan_map) {                                                       // rare: every RoI in the general form
        if (pk >= 0) build_geometry(0, roi_q, true);
        __syncthreads();
    }
    for (int base = 0; base < n_slots; base += kQuadGeoSlots) {
    const int nb = min(n_slots - base, kQuadGeoSlots), npairs = (nb + 1) >> 1;
    if (base > 0) {
        __syncthreads();                                                     // everybody is done with the previous batch's entries
        if (tid == 0) ctr[0] = 0;
        if (pk >= 0) build_geometry(base, roi_of_lane(base), nan_map);
        __syncthreads();
    }
    // The waves draw PAIRS of RoIs (2 d, 2 d + 1) and work on both at once: a pass is a chain of dependent LDS round trips (geometry ->
    // cells -> staging slot -> store), and with four waves per SIMD nobody else hides them; two independent chains per wave do.
    int static_draw = wave;
    int draw = (dbg & 32) ? static_draw : __builtin_amdgcn_readfirstlane(quad_draw(lds));
    uint32_t gw_n[2] = {0u, 0u};
    uint2 gh_n[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)}, gm_n[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
    auto fetch_geo = [&](int d) {
        if (d < npairs) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int sl = min(2 * d + s2, nb - 1);                      // an odd tail repeats the last RoI (not stored)
                gw_n[s2] = geo_w_lane[sl * 8]; gh_n[s2] = geo_h_lane[sl * 8]; gm_n[s2] = geo_m[sl];
            }
        }
    };
    fetch_geo(draw);
    for (;;) {
        if (draw >= npairs) break;
        const int r0i = g0 + (base + 2 * draw) * rsplit;
        const bool second = 2 * draw + 1 < nb;
        int draw_v;
        if (dbg & 32) { static_draw += kQuadWaves; draw_v = static_draw; }        // ablation: static hand-out
        else draw_v = quad_draw(lds);                                    // one draw ahead: issued now, looked at after this pass's scan
        const uint32_t gw0 = gw_n[0], gw1 = gw_n[1];
        const uint2 gh0 = gh_n[0], gh1 = gh_n[1];
        uint32_t mw0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gm_n[0].x), mh0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gm_n[0].y);
        uint32_t mw1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gm_n[1].x), mh1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gm_n[1].y);
        if (dbg & 4) { mw0 = (mw0 & ~255u) | 1u; mh0 = 1u; mw1 = (mw1 & ~255u) | 1u; mh1 = 1u; }
        float4 acc0, acc1;
        int4 ai0 = make_int4(0, 0, 0, 0), ai1 = ai0;
        if constexpr (ARGMAX) {
            quad_scan_slot_argmax(img, gw0, gh0, mw0, mh0, W, acc0, ai0);
            quad_scan_slot_argmax(img, gw1, gh1, mw1, mh1, W, acc1, ai1);
        } else if (!((mw0 | mw1) >> 17) && (mw0 & 255u) <= 2u && mh0 <= 2u && (mw1 & 255u) <= 2u && mh1 <= 2u) {
            // the common case: both RoIs take at most 2 x 2 tiles -- four reads each, eight in flight together (two each where the
            // RoI has ONE tile row: its first row is its last row)
            const uint32_t cl0 = gw0 & 0xffffu, ch0 = gw0 >> 16, cl1 = gw1 & 0xffffu, ch1 = gw1 >> 16;
            const float4 c0v = quad_cell(img, gh0.y + cl0), d0 = quad_cell(img, gh0.y + ch0);
            const float4 c1v = quad_cell(img, gh1.y + cl1), d1 = quad_cell(img, gh1.y + ch1);
            acc0 = max4(c0v, d0);
            acc1 = max4(c1v, d1);
            if ((mh0 | mh1) > 1u) {
                const float4 a0 = quad_cell(img, gh0.x + cl0), b0 = quad_cell(img, gh0.x + ch0);
                const float4 a1 = quad_cell(img, gh1.x + cl1), b1 = quad_cell(img, gh1.x + ch1);
                acc0 = max4_3(acc0, a0, b0);
                acc1 = max4_3(acc1, a1, b1);
            }
        } else {
            acc0 = quad_scan_slot(img, gw0, gh0, mw0, mh0);
            acc1 = quad_scan_slot(img, gw1, gh1, mw1, mh1);
        }
        draw = __builtin_amdgcn_readfirstlane(draw_v);
        fetch_geo(draw);
        if (dbg & 8) {
            if (acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w == 12345.678f) y[0] = acc0.x;
            stamp();
            continue;
        }
        if (lane_on) {
            sv_lane[0] = acc0.x; sv_lane[bins] = acc0.y; sv_lane[2 * bins] = acc0.z; sv_lane[3 * bins] = acc0.w;
            float *s1 = sv_lane + kQuadStageFloats;
            s1[0] = acc1.x; s1[bins] = acc1.y; s1[2 * bins] = acc1.z; s1[3 * bins] = acc1.w;
            if constexpr (ARGMAX) {
                int *i0 = reinterpret_cast<int *>(sv_lane + 2 * kQuadStageFloats), *i1 = i0 + kQuadStageFloats;
                i0[0] = ai0.x; i0[bins] = ai0.y; i0[2 * bins] = ai0.z; i0[3 * bins] = ai0.w;
                i1[0] = ai1.x; i1[bins] = ai1.y; i1[2 * bins] = ai1.z; i1[3 * bins] = ai1.w;
            }
        }
        frcnn_wave_sync();     // the wave's own LDS writes above are read by other lanes below (DS ops of a wave are in order)
        // (r, c0 .. c0 + cg, :, :) is one contiguous run of cg * bins floats of y
        const uint32_t dst0 = (uint32_t)r0i * run_bytes + c0_bytes, dst1 = dst0 + (uint32_t)rsplit * run_bytes;
        if (vec_ok) {
            const int li = min(lane, kQuadStageFloats / 4 - 1);
            const float4 o0 = reinterpret_cast<const float4 *>(sv)[li], o1 = reinterpret_cast<const float4 *>(sv + kQuadStageFloats)[li];
            if constexpr (OUT16) {
                frcnn_buf_store_b64_soff<ST>(ybuf, st_off, dst0, make_uint2(frcnn_pack_bf16x2(o0.x, o0.y), frcnn_pack_bf16x2(o0.z, o0.w)));
                if (second) frcnn_buf_store_b64_soff<ST>(ybuf, st_off, dst1, make_uint2(frcnn_pack_bf16x2(o1.x, o1.y), frcnn_pack_bf16x2(o1.z, o1.w)));
            } else {
                frcnn_buf_store_f32x4_soff<ST>(ybuf, st_off, dst0, o0);
                if (second) frcnn_buf_store_f32x4_soff<ST>(ybuf, st_off, dst1, o1);
            }
            if constexpr (ARGMAX) {
                const float4 j0 = reinterpret_cast<const float4 *>(sv + 2 * kQuadStageFloats)[li], j1 = reinterpret_cast<const float4 *>(sv + 3 * kQuadStageFloats)[li];
                frcnn_buf_store_f32x4_soff<ST>(abuf, st_off, dst0, j0);
                if (second) frcnn_buf_store_f32x4_soff<ST>(abuf, st_off, dst1, j1);
            }
        } else {
            for (int i = lane; i < run; i += 64) {
                if constexpr (OUT16) {
                    uint16_t *y16 = reinterpret_cast<uint16_t *>(y);
                    y16[(size_t)(dst0 / 2) + i] = (uint16_t)frcnn_pack_bf16x2(sv[i], 0.0f);
                    if (second) y16[(size_t)(dst1 / 2) + i] = (uint16_t)frcnn_pack_bf16x2(sv[kQuadStageFloats + i], 0.0f);
                    continue;
                }
                y[(size_t)(dst0 / 4) + i] = sv[i];
                if (second) y[(size_t)(dst1 / 4) + i] = sv[kQuadStageFloats + i];
                if constexpr (ARGMAX) {
                    argmax[(size_t)(dst0 / 4) + i] = reinterpret_cast<const int *>(sv)[2 * kQuadStageFloats + i];
                    if (second) argmax[(size_t)(dst1 / 4) + i] = reinterpret_cast<const int *>(sv)[3 * kQuadStageFloats + i];
                }
            }
        }
        frcnn_wave_sync();     // the slots are rewritten by the next pair only after these reads were issued
        stamp();
    }
    }
    stamp();
}

// dx[c, argmax] += dy for every (roi, c, bin) with argmax >= 0 (Chainer backward_cpu).  fp32 atomics:
// the accumulation order differs from the reference's roi-major loop only in rounding.
__global__ void __launch_bounds__(256)
roi_pool_bwd_kernel(const float *__restrict__ dy, const int32_t *__restrict__ argmax, int C, int HW, int bins, size_t total,
                    float *__restrict__ dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int32_t a = argmax[i];
        if (a >= 0) {
            const int c = (int)((i / bins) % C);
            atomicAdd(&dx[(size_t)c * HW + a], dy[i]);
        }
    }
}

// Backward, plane-resident: dx[c, argmax] += dy for every (roi, c, bin) with argmax >= 0 (Chainer backward_cpu).  One workgroup owns NCH channels of
// dx as planar fp32 accumulators in LDS, streams the dy / argmax runs of all RoIs for those channels, adds into the LDS cells and writes its planes
// out once -- no memset of dx, no global atomics, each dy / argmax byte read once, each dx byte written once (65.1 MB).  The adds are compare-and-swap
// loops on the LDS cells: the LDS's INTEGER atomics run at full rate on this chip, ds_add_f32 does not (78 us with it in round 3).  The summation
// order over RoIs differs from the reference's loop only in rounding (test bar 1e-4).
// Round 5 (25.9 -> 17.0 us on the 300 x 512 x 7 x 7 case, scripts/micro/roi_micro, profiles/r05_roi_micro.txt):
//  * NCH = 2 channels per workgroup instead of four: 256 workgroups for the 512-channel map -- the four-channel launch left half the CUs without a
//    workgroup and was bound by what 128 CUs can pull (its "no update" ablation cost the same 26 us).  A lane owns NCH consecutive floats of the
//    RoI's (NCH x bins)-float run: 392 contiguous bytes, one 8-byte load per lane.  (NCH = 1, 512 workgroups of 196-byte runs: 25-27 us.)
//  * a wave issues the loads of DEPTH = 5 RoIs (ten loads) before it touches the first value.  MORE in flight is slower, not faster: DEPTH 10 / 20
//    21.4 / 23.8 us against 19.1; the next step's loads issued before the current step's adds (two register sets) 20.8; the adds of a step as one
//    batch of LDS reads + one batch of swaps 22.1; both 24.2; the first step's loads before the planes are zeroed 19.3 against 18.2.  Ten loads per
//    wave are already 16 MB in flight chip-wide -- queueing, not latency, is what a wave waits for, and a deeper queue only disorders the stream.
//  * workgroups are dealt to the eight XCDs round-robin; the XCD-major mapping gives the workgroups of ONE XCD consecutive channel pairs, so the
//    pieces that share a 128-byte line (a 392-byte piece starts and ends inside one) and a RoI's whole 12.5 KB stretch meet in one L2: 19.1 -> 18.2;
//  * the reads are non-temporal (read once: nothing of them should stay in the L2): 18.2 -> 17.6, and depth stops mattering (17.6-17.8 for 5..10);
//  * the planes leave as write-through stores (they drain while other workgroups still stream instead of after the last wave): 17.7 -> 17.05.
constexpr int kBwdMaxPlane = 38 * 64;      // floats per channel plane the LDS image holds
template <int NCH, int DEPTH, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
roi_pool_bwd_runs_kernel(const float *__restrict__ dy, const int32_t *__restrict__ argmax, int R, int C, int HW, int bins,
                         float *__restrict__ dx) {
    __shared__ __attribute__((aligned(16))) float planes[NCH * kBwdMaxPlane];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int grp = blockIdx.x;
    if ((gridDim.x & 7) == 0) grp = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);          // XCD-major (see above)
    const int c0 = grp * NCH;
    for (int i = tid; i < NCH * HW; i += 64 * WAVES) planes[i] = 0.0f;
    __syncthreads();
    int ch[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) ch[i] = min((NCH * lane + i) / bins, NCH - 1) * HW;
    const uint32_t total_bytes = (uint32_t)((size_t)R * C * bins * sizeof(float));
    const frcnn_buf_t dbuf = frcnn_make_buf(dy, total_bytes), abuf = frcnn_make_buf(argmax, total_bytes);
    const bool own = lane < bins;
    const uint32_t voff = own ? (uint32_t)(lane * 4 * NCH) : kBufOob;
    const uint32_t roi_bytes = (uint32_t)(C * bins) * 4u, c0_bytes = (uint32_t)(c0 * bins) * 4u;
    constexpr int kNt = 2;                                                  // buffer-load cache policy: non-temporal
#pragma unroll 1
    for (int r0 = wave; r0 < R; r0 += DEPTH * WAVES) {
        float gv[DEPTH][NCH];
        int av[DEPTH][NCH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const int r = r0 + k * WAVES;                                   // wave-uniform
            const uint32_t s = (uint32_t)min(r, R - 1) * roi_bytes + c0_bytes;
            const uint32_t vo = r < R ? voff : kBufOob;
            if constexpr (NCH == 4) {
                const float4 g = frcnn_buf_load_f32x4_soff_aux<kNt>(dbuf, vo, s), a = frcnn_buf_load_f32x4_soff_aux<kNt>(abuf, vo, s);
                gv[k][0] = g.x; gv[k][1] = g.y; gv[k][2] = g.z; gv[k][3] = g.w;
                av[k][0] = __float_as_int(a.x); av[k][1] = __float_as_int(a.y); av[k][2] = __float_as_int(a.z); av[k][3] = __float_as_int(a.w);
            } else if constexpr (NCH == 2) {
                const uint2 g = frcnn_buf_load_b64_soff_aux<kNt>(dbuf, vo, s), a = frcnn_buf_load_b64_soff_aux<kNt>(abuf, vo, s);
                gv[k][0] = __uint_as_float(g.x); gv[k][1] = __uint_as_float(g.y);
                av[k][0] = (int)a.x; av[k][1] = (int)a.y;
            } else {
                gv[k][0] = frcnn_buf_load_f32_soff_aux<kNt>(dbuf, vo, s);
                av[k][0] = __float_as_int(frcnn_buf_load_f32_soff_aux<kNt>(abuf, vo, s));
            }
        }
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const bool live = own && r0 + k * WAVES < R;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (live && av[k][i] >= 0) {
                    int *ic = reinterpret_cast<int *>(&planes[ch[i] + av[k][i]]);
                    int old = *ic;
                    for (;;) {
                        const int want = __float_as_int(__int_as_float(old) + gv[k][i]);
                        const int prev = atomicCAS(ic, old, want);
                        if (prev == old) break;
                        old = prev;
                    }
                }
        }
    }
    __syncthreads();
    float *dst = dx + (size_t)c0 * HW;                                      // (c0 .. c0 + NCH - 1, :, :) is one contiguous run of dx
    if (((NCH * HW) & 3) == 0 && (((size_t)c0 * HW) & 3) == 0) {
        const frcnn_buf_t obuf = frcnn_make_buf(dst, (uint32_t)(NCH * HW * 4));
        for (int i = tid; i < NCH * HW / 4; i += 64 * WAVES) frcnn_buf_store_f32x4_wt(obuf, (uint32_t)i * 16u, reinterpret_cast<const float4 *>(planes)[i]);
    } else {
        for (int i = tid; i < NCH * HW; i += 64 * WAVES) dst[i] = planes[i];
    }
}

static int frcnn_roi_cu_count();

static void roi_fill_tables(RoiBinTables &tables, int outh, int outw) {
    memset(&tables, 0, sizeof(tables));
    for (int t = 0; t < 2; ++t) {
        const int out = t ? outw : outh;
        for (int ext = 1; ext < kTabExt; ++ext) {
            int m = 0;
            for (int p = 0; p < out; ++p) {
                const double stride = (double)ext / (double)out;                // bin_range()'s arithmetic; offset and clamp are applied per RoI
                const int lo = (int)floor((double)p * stride), hi = (int)ceil((double)(p + 1) * stride);
                tables.tab[t][ext][p] = (uint16_t)(lo | (hi << 8));
                if (hi - lo > m) m = hi - lo;
            }
            tables.tabmax[t][ext] = (uint8_t)m;
        }
    }
}

// RoiEdgeTable for one output shape: bin_range()'s double expressions for every extent below kQuadTabExt
static void roi_fill_edge_table(RoiEdgeTable &t, int outh, int outw) {
    memset(&t, 0, sizeof(t));
    for (int dim = 0; dim < 2; ++dim) {
        const int out = dim ? outh : outw;
        for (int e = 0; e < kQuadTabExt; ++e) {
            const double stride = (double)e / (double)out;
            int mb = 0, hl = 0;
            for (int p = 0; p < out; ++p) {
                const int lo = (int)floor((double)p * stride), hi = (int)ceil((double)(p + 1) * stride);
                t.row[dim][e][p] = (uint16_t)(lo | (hi << 8));
                mb = std::max(mb, hi - lo); hl = hi;
            }
            for (int p = out; p < 7; ++p) t.row[dim][e][p] = t.row[dim][e][out - 1];   // entries past out - 1 repeat the last bin
            t.row[dim][e][7] = (uint16_t)(mb | (hl << 8));                   // out <= 7: word 7 is free
        }
    }
}

// Launch the quad-cell kernel (fp32 NCHW in, fp32 out; argmax != nullptr: the training form) wherever its 38-row image and its
// 32-bit output offsets fit.  Returns false when it does not apply.
template <bool IN16 = false, bool OUT16 = false>
static bool roi_quads_launch(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw,
                             float scale, float *y, int32_t *argmax, hipStream_t stream) {
    if (H > kQuadRows || W > 64) return false;
    if ((size_t)C * H * W * sizeof(float) >= (1ull << 31)) return false;      // the map is read through a 32-bit buffer descriptor
    if ((size_t)R * C * outh * outw * sizeof(float) >= (1ull << 32)) return false;
    if constexpr (IN16 || OUT16) {                                            // the bf16 forms: one instantiation each (write-through stores, any bin count)
        if (argmax) return false;
        RoiEdgeTable qk16;
        roi_fill_edge_table(qk16, outh, outw);
        const int cq = frcnn_cdiv(C, 4);
        int rs = frcnn_cdiv(frcnn_roi_cu_count(), cq);
        const int ms = frcnn_cdiv(R, 2 * kQuadWaves);
        if (rs > ms) rs = ms;
        if (rs < 1) rs = 1;
        const char *fx = frcnn_tune("FRCNN_ROI_RSPLIT");
        if (fx && atoi(fx) > 0) rs = atoi(fx);
        if (outh * outw == 49) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_quads_kernel<16, 49, false, IN16, OUT16>), dim3(cq, rs), dim3(64 * kQuadWaves), 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, nullptr, rs, qk16, 0);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_quads_kernel<16, 0, false, IN16, OUT16>), dim3(cq, rs), dim3(64 * kQuadWaves), 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, nullptr, rs, qk16, 0);
        return true;
    }
    RoiEdgeTable qk;
    roi_fill_edge_table(qk, outh, outw);
    const int cquads = frcnn_cdiv(C, 4);
    int rsplit = frcnn_cdiv(frcnn_roi_cu_count(), cquads);                   // one resident workgroup per CU
    const int max_split = frcnn_cdiv(R, 2 * kQuadWaves);                      // at least one pair of RoIs per wave
    if (rsplit > max_split) rsplit = max_split;
    if (rsplit < 1) rsplit = 1;
    const char *fix = frcnn_tune("FRCNN_ROI_RSPLIT");                             // test hook: RoI groups per channel group
    if (fix && atoi(fix) > 0) rsplit = atoi(fix);
    const char *st = frcnn_tune("FRCNN_ROI_ST");                                  // A/B hook: 0 = plain stores, default write-through
    int qdbg = 0;
#ifdef FRCNN_TIMING_ABLATIONS
    const char *qdbg_s = frcnn_tune("FRCNN_ROI_DBG");
    qdbg = qdbg_s ? atoi(qdbg_s) : 0;
#endif
    const dim3 grid(cquads, rsplit), blk(64 * kQuadWaves);
    const bool b49 = outh * outw == 49;
    if (argmax) {
        if (b49) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_quads_kernel<16, 49, true>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, argmax, rsplit, qk, qdbg);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_quads_kernel<16, 0, true>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, argmax, rsplit, qk, qdbg);
    } else if (st && atoi(st) == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_quads_kernel<0, 0, false>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, argmax, rsplit, qk, qdbg);
    else if (b49) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_quads_kernel<16, 49, false>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, argmax, rsplit, qk, qdbg);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_quads_kernel<16, 0, false>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, argmax, rsplit, qk, qdbg);
    return true;
}

// Launch the cell-major kernel when the map fits its LDS image (W <= 64 cells per row; H <= 38: two workgroups per CU,
// H <= 76: one).  FRCNN_ROI_KERNEL=planes keeps the plane kernel (A/B measurements).  Returns false when it does not apply.
template <int OUT16, bool IN16 = false>
static bool roi_cells_launch(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw,
                             float scale, float *y, hipStream_t stream) {
    const char *sel = frcnn_tune("FRCNN_ROI_KERNEL");
    if (sel && sel[0] == 'p' && OUT16 != 2 && !IN16) return false;           // A/B hook (the split-tensor and blocked-map forms have no plane kernel: they ignore it)
    if (W > kCellPitch || H > 76) return false;
    if ((size_t)C * H * W * sizeof(float) >= (1ull << 31)) return false;      // the map is read through a 32-bit buffer descriptor
    if constexpr (OUT16 == 0 && !IN16) {
        if (!(sel && sel[0] == 'c') && roi_quads_launch(x, C, H, W, rois, R, roi_cols, outh, outw, scale, y, nullptr, stream)) return true;
    }
    if constexpr (IN16 && OUT16 != 2) {                                       // round 4: the quad kernel reads the blocked bf16 map itself (FRCNN_ROI_KERNEL=cells: the round-2 kernel, A/B)
        if (!(sel && sel[0] == 'c') && roi_quads_launch<true, OUT16 == 1>(x, C, H, W, rois, R, roi_cols, outh, outw, scale, y, nullptr, stream)) return true;
    }
    const int cgroups = frcnn_cdiv(C, 8);
    const int waves = H <= 38 ? 16 : 8;
    const char *mul = frcnn_tune("FRCNN_ROI_SPLIT_MUL");                         // tuning hook: workgroups per CU (default 1)
    const int rounds = mul && atoi(mul) > 0 ? atoi(mul) : 1;
    int rsplit = frcnn_cdiv(rounds * frcnn_roi_cu_count(), cgroups);         // one resident workgroup per CU
    const int max_split = frcnn_cdiv(R, waves);                              // at least one RoI per wave
    if (rsplit > max_split) rsplit = max_split;
    if (rsplit < 1) rsplit = 1;
    const char *fix = frcnn_tune("FRCNN_ROI_RSPLIT");                             // test hook: RoI groups per channel group
    if (fix && atoi(fix) > 0) rsplit = atoi(fix);
    RoiBinTables tables;
    roi_fill_tables(tables, outh, outw);
    int dbg = 0;
#ifdef FRCNN_TIMING_ABLATIONS
    const char *dbg_s = frcnn_tune("FRCNN_ROI_DBG");
    dbg = dbg_s ? atoi(dbg_s) : 0;
#endif
    const dim3 grid(cgroups, rsplit), blk(64 * waves);
    if (H <= 38) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_cells_kernel<38, OUT16, IN16>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, rsplit, tables, dbg);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_cells_kernel<76, OUT16, IN16>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, scale, y, rsplit, tables, dbg);
    return true;
}

static int frcnn_roi_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus;
}

}  // namespace

extern "C" {

#ifndef FRCNN_HALF_F16      // roi_f16.hip compiles this file a second time for the two 16-bit entry points only (their fp16 twins)
size_t frcnn_roi_pool_workspace_bytes(int C, int H, int W) {
    if (C < 1 || H < 1 || W < 1) return 0;
    return frcnn_align256((size_t)C * H * W * sizeof(float));
}

int frcnn_chw_to_hwc(const float *x, int C, int H, int W, float *xt, void *stream_) {
    if (!x || !xt || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(chw_to_hwc_kernel, dim3(frcnn_cdiv(H * W, 64), frcnn_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream_, x,
                       xt, C, H * W);
    return frcnn_launch_status();
}

int frcnn_roi_pool_fwd_hwc(const float *xt, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw,
                           float spatial_scale, float *y, int32_t *argmax, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!xt || !rois || !y || C < 1 || H < 1 || W < 1 || R < 0) return FRCNN_ERR_INVALID;
    if (outh < 1 || outw < 1 || outh > 7 || outw > 7 || (roi_cols != 4 && roi_cols != 5)) return FRCNN_ERR_INVALID;   // LDS tile is sized for <= 7x7 bins
    if (R == 0) return FRCNN_OK;
    const dim3 blk(64 * outh);
    if (C % 128 == 0) {
        const dim3 grid(R, C / 128);
        if (argmax) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_hwc_kernel<2, true>), grid, blk, 0, stream, xt, C, H, W, rois, roi_cols, outh, outw, spatial_scale, y, argmax);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_hwc_kernel<2, false>), grid, blk, 0, stream, xt, C, H, W, rois, roi_cols, outh, outw, spatial_scale, y, argmax);
    } else {
        const dim3 grid(R, frcnn_cdiv(C, 64));
        if (argmax) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_hwc_kernel<1, true>), grid, blk, 0, stream, xt, C, H, W, rois, roi_cols, outh, outw, spatial_scale, y, argmax);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_hwc_kernel<1, false>), grid, blk, 0, stream, xt, C, H, W, rois, roi_cols, outh, outw, spatial_scale, y, argmax);
    }
    return frcnn_launch_status();
}

#endif
// Channel planes per workgroup for the plane-resident kernel, or 0 when a plane does not fit in LDS.
static int roi_planes_per_group(int C, int H, int W, int outh, int outw) {
    if (W > kRowPitch || frcnn_cdiv(W, outw) + 1 > kMaxBinW || frcnn_cdiv(H, outh) + 1 > kMaxBinH) return 0;
    for (int cg = 8; cg >= 1; cg >>= 1)
        if ((size_t)cg * (H * kRowPitch + kPlanePad) + kRowPitch <= (size_t)kPlaneFloats) return cg < C ? cg : C;
    return 0;
}

#ifndef FRCNN_HALF_F16
int frcnn_roi_pool_fwd_chw(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw,
                           float spatial_scale, float *y, int32_t *argmax, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !rois || !y || C < 1 || H < 1 || W < 1 || R < 0) return FRCNN_ERR_INVALID;
    if (outh < 1 || outw < 1 || outh > 7 || outw > 7 || (roi_cols != 4 && roi_cols != 5)) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    if (!argmax && roi_cells_launch<0>(x, C, H, W, rois, R, roi_cols, outh, outw, spatial_scale, y, stream)) return frcnn_launch_status();
    {
        const char *sel = frcnn_tune("FRCNN_ROI_KERNEL");                         // A/B hook: p = the plane kernel for the arg-max form too
        if (argmax && !(sel && (sel[0] == 'p' || sel[0] == 'c')) && roi_quads_launch(x, C, H, W, rois, R, roi_cols, outh, outw, spatial_scale, y, argmax, stream))
            return frcnn_launch_status();
    }
    const int cg = roi_planes_per_group(C, H, W, outh, outw);
    if (cg == 0) {     // map too large for LDS-resident planes: channel-last gather kernel
        if (!workspace || workspace_bytes < frcnn_roi_pool_workspace_bytes(C, H, W)) return FRCNN_ERR_INVALID;
        const int st = frcnn_chw_to_hwc(x, C, H, W, (float *)workspace, stream);
        if (st != FRCNN_OK) return st;
        return frcnn_roi_pool_fwd_hwc((const float *)workspace, C, H, W, rois, R, roi_cols, outh, outw, spatial_scale, y, argmax, stream);
    }
    const int cgroups = frcnn_cdiv(C, cg);
    int rgroups = frcnn_cdiv(frcnn_roi_cu_count(), cgroups);           // about one workgroup per CU
    const int max_rgroups = frcnn_cdiv(R, kPlaneWaves);                  // at least one RoI per wave
    if (rgroups > max_rgroups) rgroups = max_rgroups;
    if (rgroups < frcnn_cdiv(R, kMaxRoisPerBlock)) rgroups = frcnn_cdiv(R, kMaxRoisPerBlock);
    if (rgroups < 1) rgroups = 1;
    if (rgroups > R) rgroups = R;
    const int per_block = frcnn_cdiv(R, rgroups);
    const dim3 grid(cgroups, rgroups), blk(64 * kPlaneWaves);
    if (argmax) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_planes_kernel<true>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, spatial_scale, y, argmax, cg, per_block);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_planes_kernel<false>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw, spatial_scale, y, argmax, cg, per_block);
    return frcnn_launch_status();
}

int frcnn_roi_pool_fwd_chw_f32s(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw, float spatial_scale,
                                uint16_t *y_parts, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !rois || !y_parts || C < 1 || H < 1 || W < 1 || R < 0) return FRCNN_ERR_INVALID;
    if (outh < 1 || outw < 1 || outh > 7 || outw > 7 || (roi_cols != 4 && roi_cols != 5)) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    if (roi_cells_launch<2>(x, C, H, W, rois, R, roi_cols, outh, outw, spatial_scale, reinterpret_cast<float *>(y_parts), stream)) return frcnn_launch_status();
    return FRCNN_ERR_UNSUPPORTED;                    // cell-major kernel only (maps up to 76 x 64): pool in fp32 and frcnn_f32s_split otherwise
}

#endif

int frcnn_roi_pool_fwd_chw_bf16(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw,
                                float spatial_scale, uint16_t *y, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !rois || !y || C < 1 || H < 1 || W < 1 || R < 0) return FRCNN_ERR_INVALID;
    if (outh < 1 || outw < 1 || outh > 7 || outw > 7 || (roi_cols != 4 && roi_cols != 5)) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    if (roi_cells_launch<1>(x, C, H, W, rois, R, roi_cols, outh, outw, spatial_scale, reinterpret_cast<float *>(y), stream)) return frcnn_launch_status();
    const int cg = roi_planes_per_group(C, H, W, outh, outw);
    if (cg == 0) return FRCNN_ERR_UNSUPPORTED;      // plane-resident kernel only: convert an fp32 result with frcnn_f32_to_bf16 instead
    const int cgroups = frcnn_cdiv(C, cg);
    int rgroups = frcnn_cdiv(frcnn_roi_cu_count(), cgroups);
    const int max_rgroups = frcnn_cdiv(R, kPlaneWaves);
    if (rgroups > max_rgroups) rgroups = max_rgroups;
    if (rgroups < frcnn_cdiv(R, kMaxRoisPerBlock)) rgroups = frcnn_cdiv(R, kMaxRoisPerBlock);
    if (rgroups < 1) rgroups = 1;
    if (rgroups > R) rgroups = R;
    const int per_block = frcnn_cdiv(R, rgroups);
    const dim3 grid(cgroups, rgroups), blk(64 * kPlaneWaves);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_planes_kernel<false, true>), grid, blk, 0, stream, x, C, H, W, rois, roi_cols, R, outh, outw,
                       spatial_scale, reinterpret_cast<float *>(y), (int32_t *)nullptr, cg, per_block);
    return frcnn_launch_status();
}

int frcnn_roi_pool_fwd_blk_bf16(const uint16_t *x_blk, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw,
                                float spatial_scale, void *y, int out_bf16, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x_blk || !rois || !y || C < 1 || H < 1 || W < 1 || R < 0) return FRCNN_ERR_INVALID;
    if (outh < 1 || outw < 1 || outh > 7 || outw > 7 || (roi_cols != 4 && roi_cols != 5)) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    const float *x = reinterpret_cast<const float *>(x_blk);
    const bool ok = out_bf16 ? roi_cells_launch<1, true>(x, C, H, W, rois, R, roi_cols, outh, outw, spatial_scale, reinterpret_cast<float *>(y), stream)
                             : roi_cells_launch<0, true>(x, C, H, W, rois, R, roi_cols, outh, outw, spatial_scale, reinterpret_cast<float *>(y), stream);
    return ok ? frcnn_launch_status() : FRCNN_ERR_UNSUPPORTED;   // cell-major kernel only (maps up to 76 x 64): frcnn_bf16_to_nchw_f32 + frcnn_roi_pool_fwd_chw otherwise
}

#ifndef FRCNN_HALF_F16
int frcnn_roi_pool_fwd(const float *x, int C, int H, int W, const float *rois, int R, int outh, int outw, float spatial_scale,
                       float *y, int32_t *argmax, void *workspace, size_t workspace_bytes, void *stream) {
    return frcnn_roi_pool_fwd_chw(x, C, H, W, rois, R, 5, outh, outw, spatial_scale, y, argmax, workspace, workspace_bytes, stream);
}

int frcnn_roi_pool_bwd(const float *dy, const int32_t *argmax, int R, int C, int H, int W, int outh, int outw, float *dx,
                       void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !argmax || !dx || R < 0 || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    {
        // plane-resident kernel: NCH channel planes of dx in LDS per workgroup, no memset, no global atomics.  FRCNN_ROI_BWD (tuning registry):
        // "n1" / "n2" / "n4" force the channels per workgroup (tests), "atomic" the round-1 global-atomic kernel (also the path of maps beyond the LDS planes).
        const char *sel = frcnn_tune("FRCNN_ROI_BWD");
        const int bins = outh * outw;
        if (!(sel && sel[0] == 'a') && R > 0 && H * W <= kBwdMaxPlane && bins >= 1 && bins <= 64 &&
            (size_t)R * C * bins * sizeof(float) < (1ull << 32)) {
            int nch = (C & 1) ? 1 : 2;
            if (sel && sel[0] == 'n') {
                nch = atoi(sel + 1);
                if ((nch != 1 && nch != 2 && nch != 4) || C % nch) return FRCNN_ERR_INVALID;          // an unknown form: never a silent substitute
            }
            if (nch == 2) hipLaunchKernelGGL((roi_pool_bwd_runs_kernel<2, 5, 16>), dim3(C / 2), dim3(1024), 0, stream, dy, argmax, R, C, H * W, bins, dx);
            else if (nch == 1) hipLaunchKernelGGL((roi_pool_bwd_runs_kernel<1, 10, 8>), dim3(C), dim3(512), 0, stream, dy, argmax, R, C, H * W, bins, dx);
            else hipLaunchKernelGGL((roi_pool_bwd_runs_kernel<4, 5, 16>), dim3(C / 4), dim3(1024), 0, stream, dy, argmax, R, C, H * W, bins, dx);
            return frcnn_launch_status();
        }
    }
    FRCNN_HIP_TRY(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)C * H * W, stream));
    const size_t total = (size_t)R * C * outh * outw;
    if (total == 0) return FRCNN_OK;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dy, argmax, C, H * W, outh * outw, total, dx);
    return frcnn_launch_status();
}

#endif

}  // extern "C"
