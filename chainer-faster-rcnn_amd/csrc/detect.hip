// detect.hip -- ProposalLayer + greedy IoU NMS for gfx950, results identical to the reference's CPU path.
//
// Replaces /root/reference/models/proposal_layer.py:102-198 (ProposalLayer.__call__), the helpers it
// calls in models/bbox_transform.py (bbox_transform_inv :41-76, clip_boxes :79-99, filter_boxes :102-109)
// and models/cpu_nms.pyx:18-69.  Not a translation of the dead models/nms_kernel.cu: that kernel compares
// `>` in fp32, leaves the reduction to the host and is never called (proposal_layer.py:180-187).
//
// Device pipeline (one stream, no host round trip, caller-owned workspace):
//   tile_sort_kernel<true>  anchor (h,w,a) -> decode -> clip -> min-size test; writes the box and the fg score, makes the 64-bit
//                           sort key  (ordered score bits << 32) | ~index   (0 = filtered out)  in registers and bitonic-sorts
//                           its 1024-key tile (descending; wave shuffles, LDS for 3 of the 55 steps)
//   rank_scatter_kernel     global rank of a key = its position in its own tile + the number of larger
//                           keys in every other tile (branch-free binary searches, L2 resident);
//                           ranks < min(n_valid, pre_nms_top_n) are gathered into score order
//   nms_mask_kernel         upper-triangular 64x64 IoU tiles -> one uint64 suppression word per
//                           (row box, column chunk); lane = row box, column boxes broadcast from LDS
//   nms_scan_kernel         one workgroup walks the chunks in score order; the diagonal 64x64 block is
//                           resolved with scalar bit operations on one wave, survivors' mask rows are OR-ed
//                           into the `removed` bitmap by all waves; stops at post_nms_top_n and writes
//                           the final RoIs / scores / indices
// Ties between equal scores resolve to ascending index (canonical rule; NumPy's argsort()[::-1] leaves
// tie order implementation-defined).  +NaN scores sort first, as NumPy's do after the reversal.
//
// Arithmetic parity: fp32 with the reference's operation order, no FMA contraction (this file is built
// with -ffp-contract=off), IEEE division, exp evaluated in double and rounded to fp32, IoU threshold test
// `(double)iou >= thresh` (cpu_nms.pyx:18,66).
#include "frcnn_common.h"
#include <stdlib.h>
#include <algorithm>
#include <frcnn_intrin.h>
#include <frcnn_sync.h>     // angle brackets: shadowed by the test emulator

namespace {

constexpr int kSortTile = 1024;   // keys per bitonic tile (one 256-thread workgroup, 4 keys per thread)
constexpr int kSortThreads = 256;
constexpr int kRankTiles = 4;     // tiles searched together (8 = 128 KB of LDS, one workgroup per CU: measured slower) by rank_scatter_kernel (independent binary searches in flight)
constexpr int kChunk = 64;        // NMS chunk = wave width

// unsigned order == float order; EVERY NaN -- either sign bit, any payload -- takes the one top key: NumPy's argsort()[::-1]
// (proposal_layer.py:156-157, cpu_nms.pyx:26) sorts all NaNs to the end and the reversal puts them first.  0xFFC00000 (sign set) is
// what x86 makes of inf - inf, i.e. what a host-computed softmax hands over.  Several NaNs are ties: ascending index, like all ties.
__device__ __forceinline__ uint32_t ordered_bits(float f) {
    uint32_t b = __float_as_uint(f);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ unsigned long long make_key(float score, uint32_t index) {
    return ((unsigned long long)ordered_bits(score) << 32) | (uint32_t)(~index);
}
__device__ __forceinline__ uint32_t key_index(unsigned long long k) { return ~(uint32_t)k; }

// np.maximum(np.minimum(v, hi), 0) -- NumPy propagates NaN, fminf/fmaxf would swallow it
__device__ __forceinline__ float clip_like_numpy(float v, float hi) { return (v != v) ? v : fmaxf(fminf(v, hi), 0.0f); }

// Batched problems: every group owns one `slab` bytes of workspace with identical internal offsets.
template <typename T>
__device__ __forceinline__ T *slab_ptr(T *p, size_t slab) { return (T *)((char *)p + (size_t)blockIdx.z * slab); }
template <typename T>
__device__ __forceinline__ const T *slab_ptr(const T *p, size_t slab) { return (const T *)((const char *)p + (size_t)blockIdx.z * slab); }

struct Anchors {           // generate_anchors output, passed by value in the kernel arguments
    double a[32][4];
};

// ------------------------------------------------------------------------------------------------
// proposal_layer.py:135-154 + bbox_transform.py:41-109, one thread per anchor.
// Thread t -> (a = t / HW, p = t % HW) so the 4 delta reads and the score read are coalesced over p; the
// anchor's position in the reference's enumeration is idx = p*A + a (proposal_layer.py:219-220).
// Decode anchor t = a * HW + p (coalesced over p), write its box / score at the reference's enumeration index idx = p * A + a and
// return its sort key (0 = filtered out or past the end).
struct DecodeArgs {
    const float *cls_prob, *bbox_pred;
    int A, H, W, feat_stride, im_h, im_w;
    float min_size;
    float *boxes, *scores;
};
__device__ __forceinline__ unsigned long long decode_anchor(const DecodeArgs &d, const Anchors &anchors, int t) {
    const int HW = d.H * d.W;
    if (t >= d.A * HW) return 0ull;
    const int A = d.A, W = d.W;
    const int a = t / HW, p = t - a * HW;
    const int h = p / W, w = p - h * W;
    const uint32_t idx = (uint32_t)(p * A + a);
    // all_bbox = anchors + shifts in float64, then .astype(float32) (proposal_layer.py:200-221)
    const double sx = (double)(w * d.feat_stride), sy = (double)(h * d.feat_stride);
    const float bx1 = (float)(anchors.a[a][0] + sx), by1 = (float)(anchors.a[a][1] + sy);
    const float bx2 = (float)(anchors.a[a][2] + sx), by2 = (float)(anchors.a[a][3] + sy);
    // rpn_bbox_pred.transpose(1,2,0).reshape(-1,4): channel = a*4 + coord (proposal_layer.py:138)
    const float dx = d.bbox_pred[(size_t)(a * 4 + 0) * HW + p], dy = d.bbox_pred[(size_t)(a * 4 + 1) * HW + p];
    const float dw = d.bbox_pred[(size_t)(a * 4 + 2) * HW + p], dh = d.bbox_pred[(size_t)(a * 4 + 3) * HW + p];
    // bbox_transform_inv (bbox_transform.py:51-74): separate multiplies and adds, no clamp on dw/dh
    const float widths = bx2 - bx1 + 1.0f, heights = by2 - by1 + 1.0f;
    const float ctr_x = bx1 + 0.5f * widths, ctr_y = by1 + 0.5f * heights;
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = (float)exp((double)dw) * widths, ph = (float)exp((double)dh) * heights;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
    // clip_boxes (bbox_transform.py:88-98): maximum(minimum(v, int(dim-1)), 0)
    const float mx = (float)(d.im_w - 1), my = (float)(d.im_h - 1);
    x1 = clip_like_numpy(x1, mx); y1 = clip_like_numpy(y1, my);
    x2 = clip_like_numpy(x2, mx); y2 = clip_like_numpy(y2, my);
    // filter_boxes (bbox_transform.py:105-108); NaN compares false -> dropped
    const float ws = x2 - x1 + 1.0f, hs = y2 - y1 + 1.0f;
    const bool valid = (ws >= d.min_size) && (hs >= d.min_size);
    // fg score: rpn_cls_prob[A:].transpose(1,2,0) (proposal_layer.py:152-153)
    const float sc = d.cls_prob[(size_t)(A + a) * HW + p];
    reinterpret_cast<float4 *>(d.boxes)[idx] = make_float4(x1, y1, x2, y2);
    d.scores[idx] = sc;
    return valid ? make_key(sc, idx) : 0ull;
}

// keys for frcnn_nms(): every row of the (n,5) dets array takes part.
__global__ void __launch_bounds__(256)
dets_keys_kernel(const float *__restrict__ dets, int n, int n_pad, unsigned long long *__restrict__ keys,
                 int *__restrict__ counters, size_t dets_gs, size_t slab) {
    dets += blockIdx.z * dets_gs;
    keys = slab_ptr(keys, slab);
    counters = slab_ptr(counters, slab);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = make_key(dets[5 * (size_t)i + 4], (uint32_t)i);
    else if (i < n_pad) keys[i] = 0ull;
    if (i == 0) counters[0] = n;
}

// ------------------------------------------------------------------------------------------------
// Bitonic sort (descending) of one 1024-key tile by one 256-thread workgroup (one wave per SIMD of a CU; the 22 tiles of a
// 600 x 1000 image run on 22 CUs -- the network is VALU-issue-bound, so 4096-key tiles on 6 CUs were 1.5x slower: r02 measurement).
// Thread t holds the four CONSECUTIVE elements 4t .. 4t+3 in registers, so a compare-exchange at distance j is
//   j = 1, 2       between two of the thread's own registers          (no communication: 19 of the 55 network steps)
//   j = 4 .. 128   with lane ^ (j / 4) of the same wave                (__shfl_xor: no LDS image, no barrier: 33 steps)
//   j = 256, 512   with another wave                                   (through LDS, two barriers: 3 steps)
// (the all-LDS version of round 1 ran 55 barrier-separated passes).  Element i keeps the larger key of the pair (i, i ^ j) iff
// (i & j) == 0 is equal to ((i & k) == 0): the usual bitonic rule, descending.
// DECODE: the tile's keys are not read but MADE here -- 16 runs of 64 anchors, dealt round-robin over the tiles, are decoded (ProposalLayer's
// bbox_transform_inv / clip / filter, see decode_anchor) straight into the sort registers: the stand-alone decode launch (5 us of
// mostly launch latency) and its 8-byte-per-anchor key round trip are gone.  Which tile a key starts in is irrelevant: ranks are global.
// The tile's count of valid anchors goes to counters[1 + tile] (summed by rank_scatter_kernel).
template <bool DECODE>
__global__ void __launch_bounds__(kSortThreads)
tile_sort_kernel(unsigned long long *__restrict__ keys, size_t slab, DecodeArgs dec, Anchors anchors, int *__restrict__ counters) {
    __shared__ unsigned long long s[kSortTile];
    keys = slab_ptr(keys, slab);
    unsigned long long *g = keys + (size_t)blockIdx.x * kSortTile;
    const int tid = threadIdx.x;
    unsigned long long key[4];
    if constexpr (DECODE) {
        int n_valid = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // 64-anchor runs (coalesced) dealt round-robin to the tiles: every tile is a uniform sample of all positions and anchor shapes
            const int run = (r * (kSortThreads / 64) + (tid >> 6)) * (int)gridDim.x + (int)blockIdx.x;
            key[r] = decode_anchor(dec, anchors, run * 64 + (tid & 63));
            n_valid += key[r] != 0ull;
        }
        __shared__ int wave_count[kSortThreads / 64];
        for (int d = 32; d > 0; d >>= 1) n_valid += __shfl_xor(n_valid, d);
        if ((tid & 63) == 0) wave_count[tid >> 6] = n_valid;
        __syncthreads();
        if (tid == 0) {
            int c = 0;
            for (int w = 0; w < kSortThreads / 64; ++w) c += wave_count[w];
            counters[1 + blockIdx.x] = c;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) key[r] = g[tid * 4 + r];
    }
    for (int k = 2; k <= kSortTile; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if ((r & j) == 0) {
                        const int i = tid * 4 + r;
                        const bool desc = (i & k) == 0;
                        const unsigned long long a = key[r], b = key[r | j];
                        const bool swap = desc ? (a < b) : (a > b);
                        if (swap) { key[r] = b; key[r | j] = a; }
                    }
                }
            } else {
                const int dt = j >> 2;                             // partner thread = tid ^ dt, same register
                unsigned long long other[4];
                if (dt >= 64) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r * kSortThreads + tid] = key[r];
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < 4; ++r) other[r] = s[r * kSortThreads + (tid ^ dt)];
                    __syncthreads();
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) other[r] = __shfl_xor(key[r], dt);
                }
                // (i & j) and (i & k) do not depend on r here (j, k >= 4): one decision per thread and step
                const bool keep_max = (((tid * 4) & j) == 0) == (((tid * 4) & k) == 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool gt = other[r] > key[r];
                    key[r] = (keep_max == gt) ? other[r] : key[r];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) g[tid * 4 + r] = key[r];
}

// Global rank by merging: rank(key) = sum over all sorted tiles of (number of elements > key) (keys are unique, so in the key's
// own tile that count is just its position there).  A workgroup owns 256 consecutive keys and streams the tiles through LDS
// kRankTiles at a time (register-staged double buffer: the next group's 32 KB is in flight while the current one is searched); the
// kRankTiles branch-free binary searches of a thread are independent, so their 11 dependent LDS reads each overlap (one search at
// a time was a chain of 242 dependent reads for 22 tiles).  Ranks below top_k are gathered into score order.
__global__ void __launch_bounds__(256)
rank_scatter_kernel(const unsigned long long *__restrict__ keys, int n_tiles, const float *__restrict__ boxes_in, int box_stride,
                    const float *__restrict__ scores_in, int score_stride, int top_k, int *__restrict__ counters_rw, int n_count_blocks,
                    int32_t *__restrict__ order, float *__restrict__ sorted_boxes, float *__restrict__ sorted_scores,
                    size_t in_gs, size_t slab) {
    constexpr int GK = kRankTiles * kSortTile;                 // keys per group
    constexpr int SQ = GK / 256;                               // staged keys per thread
    __shared__ unsigned long long tile[2][GK];
    boxes_in += blockIdx.z * in_gs; scores_in += blockIdx.z * in_gs;
    keys = slab_ptr(keys, slab);
    order = slab_ptr(order, slab); sorted_boxes = slab_ptr(sorted_boxes, slab); sorted_scores = slab_ptr(sorted_scores, slab);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;       // position in the tile-sorted key array (< n_tiles*1024)
    const unsigned long long key = keys[t];
    const int n_keys = n_tiles * kSortTile;
    // n_valid = the decode blocks' counts, summed once (ProposalLayer path; frcnn_nms sets counters[0] itself: n_count_blocks = 0)
    if (blockIdx.x == 0 && n_count_blocks > 0) {
        __shared__ int partial[4];
        int *cw = slab_ptr(counters_rw, slab);
        int acc = 0;
        for (int b = threadIdx.x; b < n_count_blocks; b += 256) acc += cw[1 + b];
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
        if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) cw[0] = partial[0] + partial[1] + partial[2] + partial[3];
    }
    const int n_groups = (n_tiles + kRankTiles - 1) / kRankTiles;
    unsigned long long stage[SQ];
    // (all of a group's loads go out before the first LDS store: written as load + store per element the compiler put a full
    // vmcnt(0) wait between them -- SQ dependent memory round trips before the first search)
#pragma unroll
    for (int q = 0; q < SQ; ++q) { const int i = threadIdx.x + 256 * q; stage[q] = i < n_keys ? keys[i] : 0ull; }    // a missing tile is all zeros: nothing in it is > key
#pragma unroll
    for (int q = 0; q < SQ; ++q) tile[0][threadIdx.x + 256 * q] = stage[q];
    __syncthreads();
    int rank = 0, cur = 0;
    for (int o = 0; o < n_groups; ++o) {
        const bool more = o + 1 < n_groups;
        if (more) {
#pragma unroll
            for (int q = 0; q < SQ; ++q) { const int i = (o + 1) * GK + threadIdx.x + 256 * q; stage[q] = i < n_keys ? keys[i] : 0ull; }
        }
        int pos[kRankTiles];
#pragma unroll
        for (int g = 0; g < kRankTiles; ++g) pos[g] = 0;
#pragma unroll
        for (int st = kSortTile / 2; st >= 1; st >>= 1)
#pragma unroll
            for (int g = 0; g < kRankTiles; ++g)
                if (tile[cur][g * kSortTile + pos[g] + st - 1] > key) pos[g] += st;
#pragma unroll
        for (int g = 0; g < kRankTiles; ++g) rank += pos[g] + (tile[cur][g * kSortTile + pos[g]] > key ? 1 : 0);
        if (more) {
#pragma unroll
            for (int q = 0; q < SQ; ++q) tile[cur ^ 1][threadIdx.x + 256 * q] = stage[q];
        }
        __syncthreads();
        cur ^= 1;
    }
    if (key == 0ull) return;                                   // filtered-out / padding element
    // a valid key's rank (the number of larger keys) is below n_valid by construction: only the top-K cut needs testing
    if (top_k <= 0 || rank < top_k) {
        const uint32_t idx = key_index(key);
        order[rank] = (int32_t)idx;
        const float *b = boxes_in + (size_t)idx * box_stride;
        reinterpret_cast<float4 *>(sorted_boxes)[rank] = make_float4(b[0], b[1], b[2], b[3]);
        sorted_scores[rank] = scores_in[(size_t)idx * score_stride];
    }
}

// ------------------------------------------------------------------------------------------------
// Suppression bitmask, wave-ballot form.  mask[row*pitch + colchunk] bit t  <=>  box (colchunk*64+t) comes later in score
// order than `row` and IoU(row, it) >= thresh in the reference's arithmetic.
// One wave = one 64 x 64 tile of the UPPER triangle only (tiles are enumerated linearly: no launched-and-idle lower half).
// Lane = COLUMN box (its coordinates and area live in registers for the whole tile); the 64 row boxes are walked one per step,
// each broadcast from the lane that loaded it (v_readlane -> scalar operands), so the suppression test of a step is one
// wave-wide compare whose 64-bit result IS the row's mask word -- no per-lane bit loop, no LDS, no barrier.  Lane t keeps the
// word of row t and the tile leaves as 64 eight-byte stores.
__device__ __forceinline__ float ref_max(float a, float b) { return a >= b ? a : b; }   // cpu_nms.pyx:12-13
__device__ __forceinline__ float ref_min(float a, float b) { return a <= b ? a : b; }   // cpu_nms.pyx:15-16
__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// The 64 steps of one tile.  FASTMM: coordinates are known NaN-free, so cpu_nms.pyx's max / min helpers (a >= b ? a : b) are one
// v_max_f32 / v_min_f32 each instead of compare + select -- identical values (only the sign of a zero can differ, and xx2 - xx1 does
// not see it).  The threshold test avoids the division: |inter - thr * uni| outside a 1e-4 relative band decides at once; inside it
// (or for degenerate areas) the reference's own expression is evaluated.
// DIAG: the tile lies on the diagonal (rows and columns are the same 64 boxes).  Only later boxes can be suppressed by row t, and the
// lane additionally collects ITS COLUMN of the tile -- bit (t - t_begin) of `colpiece` = "row t (< lane) suppresses this lane's box" --
// which is what lets the sequential pass resolve a chunk with wave-wide steps instead of one box at a time (nms_scan_col_body).
template <bool FASTMM, bool DIAG>
__device__ __forceinline__ unsigned long long nms_tile_words(const float4 rb, const float rarea_l, const float4 cb, const float carea,
                                                             int t_begin, int t_rows, bool col_ok, int lane, double thresh,
                                                             uint32_t &colpiece) {
    const float thr_f = (float)thresh;
    const bool lane_fast = (thresh > 1e-6) && (carea > 0.0f);
    unsigned long long word = 0ull;
    for (int t = t_begin; t < t_rows; ++t) {
        const float rx1 = lane_bcast(rb.x, t), ry1 = lane_bcast(rb.y, t), rx2 = lane_bcast(rb.z, t), ry2 = lane_bcast(rb.w, t);
        const float rarea = lane_bcast(rarea_l, t);
        float xx1, yy1, xx2, yy2, w, h;
        if constexpr (FASTMM) {
            xx1 = frcnn_max_f32(rx1, cb.x); yy1 = frcnn_max_f32(ry1, cb.y);      // cpu_nms.pyx:58-61 (i = the row box, j = the column box)
            xx2 = frcnn_min_f32(rx2, cb.z); yy2 = frcnn_min_f32(ry2, cb.w);
            w = frcnn_max_f32(0.0f, xx2 - xx1 + 1.0f); h = frcnn_max_f32(0.0f, yy2 - yy1 + 1.0f);   // :62-63
        } else {
            xx1 = ref_max(rx1, cb.x); yy1 = ref_max(ry1, cb.y);
            xx2 = ref_min(rx2, cb.z); yy2 = ref_min(ry2, cb.w);
            w = ref_max(0.0f, xx2 - xx1 + 1.0f); h = ref_max(0.0f, yy2 - yy1 + 1.0f);
        }
        const float inter = w * h;                                               // :64
        const float uni = rarea + carea - inter;                                 // :65
        const float tt = thr_f * uni;
        const float d = inter - tt;
        bool sup = d > 0.0f;
        const bool in_band = !(fabsf(d) > 1e-4f * fabsf(tt)) || !lane_fast || !(rarea > 0.0f);     // NaNs land here too
        if (__any(in_band)) {
            if (in_band) sup = (double)(inter / uni) >= thresh;                  // exact: IEEE divide, double compare (:65-66)
        }
        unsigned long long bal = __ballot(sup && col_ok);
        if constexpr (DIAG) {
            bal &= (t == 63) ? 0ull : (~0ull << (t + 1));                        // only later boxes can be suppressed by row t
            if (sup && col_ok && lane > t) colpiece |= 1u << (t - t_begin);
        }
        if (lane == t) word = bal;
    }
    return word;
}

// Everything the sequential pass needs; passed by value so the second-stage mask launch can run it itself (see below).
struct ScanArgs {
    const unsigned long long *mask;
    int pitch;
    const int *counters_in;
    int top_k, max_out;
    const int32_t *order;
    const float *sorted_boxes, *sorted_scores;
    int32_t *keep_pos, *out_index;
    float *out_boxes, *out_scores;
    int32_t *n_out;
    int out_capacity;
    size_t slab, out_gs;
    unsigned long long *colw;     // [pitch][64]: column words of the diagonal tiles (bit t of colw[c][l]: box c*64+t suppresses box c*64+l, t < l)
};
__device__ __forceinline__ void nms_scan_col_body(const ScanArgs &a, int c_begin, int c_end, int first_stage, int last_stage, int lane);

// `tail`: this is the SECOND stage of a staged NMS.  When the first stage's scan already kept `limit` boxes, every workgroup returns at
// once (and no further launch follows).  Otherwise the workgroups compute their tiles, publish them (agent-scope release) and draw a
// ticket; the one that draws the last ticket acquires and runs the rest of the sequential pass on its first wave -- the second stage
// costs ONE launch, and ~4 us of launch latency instead of ~9 when it has nothing to do.
__global__ void __launch_bounds__(256)
nms_mask_kernel(const float *__restrict__ sorted_boxes, const int *__restrict__ counters, int top_k, double thresh,
                unsigned long long *__restrict__ mask, int pitch, size_t slab, int cc_lo, int cc_hi, int max_out, int tail, ScanArgs scan,
                int *__restrict__ ticket) {
    sorted_boxes = slab_ptr(sorted_boxes, slab); counters = slab_ptr(counters, slab); mask = slab_ptr(mask, slab);
    int m = counters[0];
    if (top_k > 0 && top_k < m) m = top_k;
    const int n_chunks = (m + kChunk - 1) / kChunk;
    if (tail && (scan.n_out[blockIdx.z] >= ((max_out > 0 && max_out < m) ? max_out : m) || cc_lo >= n_chunks)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // tiles of the upper triangle, enumerated column by column (column cc holds rows 0 .. cc); this launch covers columns
    // [cc_lo, cc_hi) of the STATIC pitch (it is sized before m is known): tile T' = T + tri(cc_lo), tri(k) = k (k + 1) / 2.
    // One workgroup = one tile, its four waves take 16 rows each: a tile is a chain of 64 dependent steps (~0.2 us each for a lone
    // wave), and the first stage of a staged NMS has fewer tiles than the chip has SIMDs -- its latency is the length of that chain
    const long long tri_lo = (long long)cc_lo * (cc_lo + 1) / 2, tri_hi = (long long)cc_hi * (cc_hi + 1) / 2;
    // a second-stage launch has a FIXED number of workgroups that stride over its tiles: when it has nothing to do, its cost is the
    // dispatch of those workgroups (4.9 us for the 4275 one-tile workgroups of the 6000-box problem)
    for (long long Tp = (long long)blockIdx.x + tri_lo; Tp < tri_hi; Tp += gridDim.x) {
    int cc = (int)((sqrt(8.0 * (double)Tp + 1.0) - 1.0) * 0.5);
    cc = min(max(cc, 0), pitch - 1);
    while (cc > 0 && (long long)cc * (cc + 1) / 2 > Tp) --cc;
    while ((long long)(cc + 1) * (cc + 2) / 2 <= Tp) ++cc;
    cc = __builtin_amdgcn_readfirstlane(cc);
    const int rc = __builtin_amdgcn_readfirstlane((int)(Tp - (long long)cc * (cc + 1) / 2));
    if (cc < cc_hi && rc < n_chunks && cc < n_chunks) {                // wave-uniform
        const int c = cc * kChunk + lane, r = rc * kChunk + lane;
        float4 cb = make_float4(0.f, 0.f, 0.f, 0.f), rb = cb;
        if (c < m) cb = reinterpret_cast<const float4 *>(sorted_boxes)[c];
        if (r < m) rb = reinterpret_cast<const float4 *>(sorted_boxes)[r];
        const float carea = (cb.z - cb.x + 1.0f) * (cb.w - cb.y + 1.0f);  // areas, cpu_nms.pyx:25
        const float rarea_l = (rb.z - rb.x + 1.0f) * (rb.w - rb.y + 1.0f);
        const int t_begin = wave * 16, t_rows = min(min(kChunk, m - rc * kChunk), t_begin + 16);
        const float probe = (cb.x + cb.y) + (cb.z + cb.w) + (rb.x + rb.y) + (rb.z + rb.w);       // NaN iff any coordinate of the tile is
        unsigned long long word;
        uint32_t colpiece = 0u;
        const bool has_nan = __any(probe != probe);
        if (cc == rc) {
            if (has_nan) word = nms_tile_words<false, true>(rb, rarea_l, cb, carea, t_begin, t_rows, c < m, lane, thresh, colpiece);
            else word = nms_tile_words<true, true>(rb, rarea_l, cb, carea, t_begin, t_rows, c < m, lane, thresh, colpiece);
            // the four waves' 16-row pieces of a lane's column word are the four 16-bit quarters of one 64-bit word
            reinterpret_cast<uint16_t *>(slab_ptr(scan.colw, slab))[((size_t)cc * kChunk + lane) * 4 + wave] = (uint16_t)colpiece;
        } else {
            if (has_nan) word = nms_tile_words<false, false>(rb, rarea_l, cb, carea, t_begin, t_rows, c < m, lane, thresh, colpiece);
            else word = nms_tile_words<true, false>(rb, rarea_l, cb, carea, t_begin, t_rows, c < m, lane, thresh, colpiece);
        }
        if (r < m && (lane >> 4) == wave) mask[(size_t)r * pitch + cc] = word;
    }
    }
    if (!tail) return;
    // publish, count in, and let the last workgroup finish the sequential pass
    __shared__ int s_last;
    frcnn_drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) {
        frcnn_release_agent();
        s_last = frcnn_ticket(slab_ptr(ticket, slab)) == (int)gridDim.x - 1;
        if (s_last) frcnn_acquire_agent();
    }
    __syncthreads();
    if (s_last && wave == 0) nms_scan_col_body(scan, cc_lo, pitch, 0, 1, lane);
}

// ------------------------------------------------------------------------------------------------
// Sequential part of greedy NMS, one workgroup (256 threads).
//   removed[c] (LDS)  bit t set <=> box c*64+t is suppressed by an already-kept box
// Per chunk c: wave 0 resolves the diagonal block with scalar ops -- walk the still-alive bits in order,
// keep the lowest, clear what its diagonal word suppresses -- then all waves OR the kept rows' words
// for chunks > c into removed[].  Outputs are written at the end by the whole block.
constexpr int kMaxChunks = 1024;   // up to 65536 boxes

__global__ void __launch_bounds__(256)
nms_scan_kernel(const unsigned long long *__restrict__ mask, int pitch, const int *__restrict__ counters_in, int top_k,
                int max_out, const int32_t *__restrict__ order, const float *__restrict__ sorted_boxes,
                const float *__restrict__ sorted_scores, int32_t *__restrict__ keep_pos, int32_t *__restrict__ out_index,
                float *__restrict__ out_boxes, float *__restrict__ out_scores, int32_t *__restrict__ n_out,
                int out_capacity, size_t slab, size_t out_gs) {
    __shared__ unsigned long long removed[kMaxChunks];
    __shared__ unsigned long long kept_word;
    __shared__ int n_kept_s;
    const int gz = blockIdx.z;
    counters_in = slab_ptr(counters_in, slab); mask = slab_ptr(mask, slab); order = slab_ptr(order, slab);
    sorted_boxes = slab_ptr(sorted_boxes, slab); sorted_scores = slab_ptr(sorted_scores, slab); keep_pos = slab_ptr(keep_pos, slab);
    if (out_index) out_index += gz * out_gs;
    if (out_boxes) out_boxes += gz * out_gs * 4;
    if (out_scores) out_scores += gz * out_gs;
    n_out += gz;
    int m = counters_in[0];
    if (top_k > 0 && top_k < m) m = top_k;
    const int limit = (max_out > 0 && max_out < m) ? max_out : m;
    const int n_chunks = (m + kChunk - 1) / kChunk;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < n_chunks; c += blockDim.x) removed[c] = 0ull;
    if (threadIdx.x == 0) n_kept_s = 0;
    __syncthreads();
    int n_kept = 0;
    for (int c = 0; c < n_chunks && n_kept < limit; ++c) {
        if (wave == 0) {
            const int r = c * kChunk + lane;
            const unsigned long long diag = (r < m) ? mask[(size_t)r * pitch + c] : 0ull;
            const int dlo = (int)(uint32_t)diag, dhi = (int)(uint32_t)(diag >> 32);
            const int in_chunk = min(kChunk, m - c * kChunk);
            const unsigned long long valid = in_chunk == 64 ? ~0ull : ((1ull << in_chunk) - 1ull);
            unsigned long long alive = ~removed[c] & valid;
            unsigned long long kept = 0ull;
            int budget = limit - n_kept;
            // alive is wave-uniform (every lane reads the same LDS word); make that explicit for readlane
            uint32_t alo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)alive);
            uint32_t ahi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(alive >> 32));
            alive = ((unsigned long long)ahi << 32) | alo;
            while (alive != 0ull && budget > 0) {
                const int i = __ffsll((long long)alive) - 1;
                kept |= 1ull << i;
                --budget;
                const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane(dlo, i);
                const uint32_t shi = (uint32_t)__builtin_amdgcn_readlane(dhi, i);
                alive &= ~(((unsigned long long)shi << 32) | slo);
                alive &= ~(1ull << i);
            }
            if ((kept >> lane) & 1ull) keep_pos[n_kept + __popcll(kept & ((1ull << lane) - 1ull))] = r;
            if (lane == 0) { kept_word = kept; n_kept_s = n_kept + __popcll(kept); }
        }
        __syncthreads();
        const unsigned long long kept = kept_word;
        n_kept = n_kept_s;
        if (n_kept < limit && kept != 0ull) {
            // OR the kept rows' words for the later chunks into removed[]: wave w takes every 4th kept row
            unsigned long long kk = kept;
            int ord = 0;
            while (kk != 0ull) {
                const int i = __ffsll((long long)kk) - 1;
                kk &= kk - 1ull;
                if ((ord & 3) == wave) {
                    const unsigned long long *row = mask + (size_t)(c * kChunk + i) * pitch;
                    for (int w = c + 1 + lane; w < n_chunks; w += 64) {
                        const unsigned long long v = row[w];
                        if (v) atomicOr(&removed[w], v);
                    }
                }
                ++ord;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) n_out[0] = n_kept;
    __syncthreads();   // keep_pos written by wave 0 through global memory: same workgroup, made visible by the barrier
    for (int k = threadIdx.x; k < n_kept; k += blockDim.x) {
        const int pos = keep_pos[k];
        if (out_index) out_index[k] = order[pos];
        if (out_boxes) reinterpret_cast<float4 *>(out_boxes)[k] = reinterpret_cast<const float4 *>(sorted_boxes)[pos];
        if (out_scores) out_scores[k] = sorted_scores[pos];
    }
    // rows past the survivors are defined (zero box / zero score / index -1) so fixed-capacity consumers
    // (RoI pooling over post_nms_top_n rows) never read stale memory
    for (int k = n_kept + threadIdx.x; k < out_capacity; k += blockDim.x) {
        if (out_index) out_index[k] = -1;
        if (out_boxes) reinterpret_cast<float4 *>(out_boxes)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (out_scores) out_scores[k] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// The same sequential pass on ONE wave, for up to kWaveChunks chunks (16384 boxes: both ProposalLayer modes).  The
// `removed` bitmap lives in registers (lane l holds words l, l+64, l+128, l+192), the word of the current chunk is fetched with
// v_readlane, kept rows are OR-ed in four at a time (their loads are independent and in flight together) and the next chunk's
// diagonal word is prefetched -- no LDS, no atomics, no barriers (76 -> 54 us for 6000 boxes; what remains is the dependent
// instruction chain of a lone wave, ~1200 cycles per chunk: taking the next chunk's word off the memory path as well -- from
// prefetched super-diagonal words -- measured no faster).
constexpr int kWaveChunks = 256;

template <int NJ>                          // bitmap words per lane: chunks <= 64 * NJ
__global__ void __launch_bounds__(64)
nms_scan_wave_kernel(const unsigned long long *__restrict__ mask, int pitch, const int *__restrict__ counters_in, int top_k,
                     int max_out, const int32_t *__restrict__ order, const float *__restrict__ sorted_boxes,
                     const float *__restrict__ sorted_scores, int32_t *__restrict__ keep_pos, int32_t *__restrict__ out_index,
                     float *__restrict__ out_boxes, float *__restrict__ out_scores, int32_t *__restrict__ n_out,
                     int out_capacity, size_t slab, size_t out_gs) {
    const int gz = blockIdx.z;
    counters_in = slab_ptr(counters_in, slab); mask = slab_ptr(mask, slab); order = slab_ptr(order, slab);
    sorted_boxes = slab_ptr(sorted_boxes, slab); sorted_scores = slab_ptr(sorted_scores, slab); keep_pos = slab_ptr(keep_pos, slab);
    if (out_index) out_index += gz * out_gs;
    if (out_boxes) out_boxes += gz * out_gs * 4;
    if (out_scores) out_scores += gz * out_gs;
    n_out += gz;
    int m = counters_in[0];
    if (top_k > 0 && top_k < m) m = top_k;
    const int limit = (max_out > 0 && max_out < m) ? max_out : m;
    const int n_chunks = (m + kChunk - 1) / kChunk;
    const int lane = threadIdx.x;
    unsigned long long rem[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rem[j] = 0ull;
    int n_kept = 0;
    unsigned long long diag_next = (lane < m) ? mask[(size_t)lane * pitch] : 0ull;
    for (int c = 0; c < n_chunks && n_kept < limit; ++c) {
        const unsigned long long diag = diag_next;
        if (c + 1 < n_chunks) {
            const int r1 = (c + 1) * kChunk + lane;
            diag_next = (r1 < m) ? mask[(size_t)r1 * pitch + c + 1] : 0ull;          // lands while this chunk is resolved
        }
        unsigned long long sel = rem[0];
#pragma unroll
        for (int j = 1; j < NJ; ++j) sel = ((c >> 6) == j) ? rem[j] : sel;
        const uint32_t rlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sel, c & 63);
        const uint32_t rhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sel >> 32), c & 63);
        const int in_chunk = min(kChunk, m - c * kChunk);
        const unsigned long long valid = in_chunk == 64 ? ~0ull : ((1ull << in_chunk) - 1ull);
        unsigned long long alive = ~(((unsigned long long)rhi << 32) | rlo) & valid;
        const int dlo = (int)(uint32_t)diag, dhi = (int)(uint32_t)(diag >> 32);
        unsigned long long kept = 0ull;
        int budget = limit - n_kept;
        while (alive != 0ull && budget > 0) {
            const int i = __ffsll((long long)alive) - 1;
            kept |= 1ull << i;
            --budget;
            const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane(dlo, i);
            const uint32_t shi = (uint32_t)__builtin_amdgcn_readlane(dhi, i);
            alive &= ~(((unsigned long long)shi << 32) | slo);
            alive &= ~(1ull << i);
        }
        if ((kept >> lane) & 1ull) keep_pos[n_kept + __popcll(kept & ((1ull << lane) - 1ull))] = c * kChunk + lane;
        n_kept += __popcll(kept);
        if (n_kept < limit) {
            unsigned long long kk = kept;
            // RB kept rows per round, all their loads in flight together: a round costs one L2 round trip whatever RB is, and a
            // chunk of well-separated boxes keeps up to 64 of them.  A short round repeats its first row (OR is idempotent).
            constexpr int RB = NJ <= 2 ? 16 : 8;
            while (kk != 0ull) {
                int idx[RB];
                idx[0] = __ffsll((long long)kk) - 1;
                kk &= kk - 1ull;
#pragma unroll
                for (int q = 1; q < RB; ++q) {
                    idx[q] = kk != 0ull ? __ffsll((long long)kk) - 1 : idx[0];
                    kk &= kk - 1ull;               // 0 stays 0
                }
                unsigned long long v[RB][NJ];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const unsigned long long *row = mask + (size_t)(c * kChunk + idx[q]) * pitch;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int w = lane + 64 * j;
                        v[q][j] = (w > c && w < n_chunks) ? row[w] : 0ull;
                    }
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    unsigned long long acc = 0ull;
#pragma unroll
                    for (int q = 0; q < RB; ++q) acc |= v[q][j];
                    rem[j] |= acc;
                }
            }
        }
    }
    if (lane == 0) n_out[0] = n_kept;
    frcnn_drain_vmem();                               // keep_pos was written by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < n_kept; k += 64) {
        const int pos = keep_pos[k];
        if (out_index) out_index[k] = order[pos];
        if (out_boxes) reinterpret_cast<float4 *>(out_boxes)[k] = reinterpret_cast<const float4 *>(sorted_boxes)[pos];
        if (out_scores) out_scores[k] = sorted_scores[pos];
    }
    for (int k = n_kept + lane; k < out_capacity; k += 64) {
        if (out_index) out_index[k] = -1;
        if (out_boxes) reinterpret_cast<float4 *>(out_boxes)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (out_scores) out_scores[k] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// The sequential pass in COLUMN form, one wave, any number of chunks.  The row form above ORs the whole mask row of every box it
// keeps into a bitmap: one dependent trip to memory per 16 kept rows, ~0.45 us each (the mask was written by other XCDs) -- with
// 300 survivors among the first 600 boxes, as on the benchmark image, 19 such trips are most of its 41 us.  Here the removed-word of
// chunk c+1 is gathered directly, and one chunk AHEAD of its use:
//     removed(c+1) = OR { mask[i][c+1] : i kept in chunks < c }        <- one 8-byte load per kept box, issued BEFORE chunk c is resolved
//                  | OR { mask[c*64+l][c+1] : lane l kept in chunk c }  <- the super-diagonal word of every row of chunk c, prefetched
// so no load sits on the dependent chain; only words that are actually needed are ever read (kept boxes x visited chunks).  The list
// of kept rows lives in LDS (first 2048; beyond that it is read back from keep_pos).
// Stages: the host may split the chunk range in two launches ([0, S) and [S, end)) with the mask of the second computed in
// between; a later stage finds n_out[0] >= limit (or no chunks left) and exits at once -- see frcnn_proposals.
constexpr int kKeptLds = 2048;

__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {      // wave-uniform result
    const uint32_t lo = frcnn_wave_or_u32((uint32_t)v), hi = frcnn_wave_or_u32((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ void nms_scan_col_body(const ScanArgs &a, int c_begin, int c_end, int first_stage, int last_stage, int lane) {
    const unsigned long long *mask = a.mask;
    const int pitch = a.pitch, top_k = a.top_k, max_out = a.max_out, out_capacity = a.out_capacity;
    const int *counters_in = a.counters_in;
    const int32_t *order = a.order;
    const float *sorted_boxes = a.sorted_boxes, *sorted_scores = a.sorted_scores;
    int32_t *keep_pos = a.keep_pos, *out_index = a.out_index, *n_out = a.n_out;
    float *out_boxes = a.out_boxes, *out_scores = a.out_scores;
    const size_t slab = a.slab, out_gs = a.out_gs;
    __shared__ int kept_list[kKeptLds];
    const int gz = blockIdx.z;
    counters_in = slab_ptr(counters_in, slab); mask = slab_ptr(mask, slab); order = slab_ptr(order, slab);
    const unsigned long long *colw = slab_ptr((const unsigned long long *)a.colw, slab);
    sorted_boxes = slab_ptr(sorted_boxes, slab); sorted_scores = slab_ptr(sorted_scores, slab); keep_pos = slab_ptr(keep_pos, slab);
    if (out_index) out_index += gz * out_gs;
    if (out_boxes) out_boxes += gz * out_gs * 4;
    if (out_scores) out_scores += gz * out_gs;
    n_out += gz;
    int m = counters_in[0];
    if (top_k > 0 && top_k < m) m = top_k;
    const int limit = (max_out > 0 && max_out < m) ? max_out : m;
    const int n_chunks = (m + kChunk - 1) / kChunk;
    int n_kept = 0;
    if (!first_stage) {
        n_kept = n_out[0];
        if (n_kept >= limit || c_begin >= n_chunks) return;        // an earlier stage finished the job and wrote the outputs
        for (int k = lane; k < min(n_kept, kKeptLds); k += 64) kept_list[k] = keep_pos[k];
    }
    __builtin_amdgcn_wave_barrier();
    const int c_stop = min(c_end, n_chunks);
    // this lane's share of OR { mask[kept row][col] : the first `upto` kept rows }.  Row indices first (LDS), then ALL the 8-byte
    // gathers of a batch in flight together (8 per lane = 512 kept rows per batch): a loop of dependent index -> word pairs would
    // serialise into two memory round trips per row
    auto gather = [&](int col, int upto) -> unsigned long long {
        unsigned long long acc = 0ull;
        if (col >= n_chunks) return acc;
        const int lds_upto = min(upto, kKeptLds);
        for (int k0 = 0; k0 < lds_upto; k0 += 512) {
            int rows[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + q * 64 + lane;
                rows[q] = k < lds_upto ? kept_list[k] : -1;
            }
            unsigned long long v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = rows[q] >= 0 ? mask[(size_t)rows[q] * pitch + col] : 0ull;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc |= v[q];
        }
        for (int k = kKeptLds + lane; k < upto; k += 64) acc |= mask[(size_t)keep_pos[k] * pitch + col];   // beyond the LDS list (rare)
        return acc;
    };
    auto uniform64 = [&](unsigned long long v) -> unsigned long long {       // tell the compiler a wave-uniform value IS uniform (-> SGPRs)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    auto load_word = [&](int c, int col) -> unsigned long long {
        const int row = c * kChunk + lane;
        return (c < n_chunks && col < n_chunks && row < m) ? mask[(size_t)row * pitch + col] : 0ull;
    };
    unsigned long long removed = first_stage ? 0ull : wave_or_u64(gather(c_begin, n_kept));
    // The diagonal / super-diagonal words of the NEXT chunk are fetched a chunk ahead and handed over through LDS: the resolve loop
    // then reads registers that an LDS load produced, so the compiler's s_waitcnt vmcnt(0) in front of it is gone and the gathers
    // issued just before it really are in flight while it runs (with a register hand-over the first v_readlane waited for them).
    __shared__ unsigned long long handover[2][64];
    auto load_col = [&](int c) -> unsigned long long { return c < n_chunks ? colw[(size_t)c * kChunk + lane] : 0ull; };
    handover[0][lane] = load_col(c_begin);
    handover[1][lane] = load_word(c_begin, c_begin + 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int c = c_begin; c < c_stop && n_kept < limit; ++c) {
        __builtin_amdgcn_wave_barrier();
        const unsigned long long suppressors = handover[0][lane] & below, sup = handover[1][lane];
        if (n_kept > kKeptLds) frcnn_drain_vmem();                  // rows past the LDS list are read back from keep_pos: written by this wave
        const unsigned long long part_next = gather(c + 1, n_kept); // in flight while the chunk is resolved
        const unsigned long long col_n = load_col(c + 1), sup_n = load_word(c + 1, c + 2);
        const int in_chunk = min(kChunk, m - c * kChunk);
        const unsigned long long valid = in_chunk == 64 ? ~0ull : ((1ull << in_chunk) - 1ull);
        // Resolve the diagonal block with wave-wide steps.  U = undecided boxes (wave-uniform), kept = decided survivors.  A box none of
        // whose suppressors (earlier boxes of the chunk with IoU over the threshold: its column word) is still undecided is KEPT -- each
        // of them has been removed, or this box would have been removed with them below; a box with a suppressor among the newly kept
        // is REMOVED.  The lowest undecided box always qualifies, so the loop ends after (longest suppression chain) rounds -- a
        // handful -- instead of one scalar round per kept box; the result is the greedy pass's own (unique) fixed point.
        unsigned long long U = ~removed & valid, kept = 0ull;
        while (U != 0ull) {
            const unsigned long long nk = __ballot(((U >> lane) & 1ull) && (suppressors & U) == 0ull);
            kept |= nk;
            U &= ~nk;
            U &= ~__ballot(((U >> lane) & 1ull) && (suppressors & nk) != 0ull);
        }
        const int budget = limit - n_kept;
        if (__popcll(kept) > budget) kept = __ballot(((kept >> lane) & 1ull) && __popcll(kept & below) < budget);   // the first `budget`
        const bool mine = (kept >> lane) & 1ull;
        if (mine) {
            const int idx = n_kept + __popcll(kept & ((1ull << lane) - 1ull));
            keep_pos[idx] = c * kChunk + lane;
            if (idx < kKeptLds) kept_list[idx] = c * kChunk + lane;
        }
        n_kept += __popcll(kept);
        removed = wave_or_u64(part_next | (mine ? sup : 0ull));
        handover[0][lane] = col_n;
        handover[1][lane] = sup_n;
        __builtin_amdgcn_wave_barrier();                            // kept_list: written above, read by other lanes in the next gather
    }
    if (lane == 0) n_out[0] = n_kept;
    if (!(n_kept >= limit || c_stop >= n_chunks || last_stage)) return;   // a later stage continues (and writes the outputs)
    frcnn_drain_vmem();                               // keep_pos was written by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < n_kept; k += 64) {
        const int pos = keep_pos[k];
        if (out_index) out_index[k] = order[pos];
        if (out_boxes) reinterpret_cast<float4 *>(out_boxes)[k] = reinterpret_cast<const float4 *>(sorted_boxes)[pos];
        if (out_scores) out_scores[k] = sorted_scores[pos];
    }
    for (int k = n_kept + lane; k < out_capacity; k += 64) {
        if (out_index) out_index[k] = -1;
        if (out_boxes) reinterpret_cast<float4 *>(out_boxes)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (out_scores) out_scores[k] = 0.0f;
    }
}

__global__ void __launch_bounds__(64)
nms_scan_col_kernel(ScanArgs a, int c_begin, int c_end, int first_stage, int last_stage, int *__restrict__ ticket) {
    if (ticket && threadIdx.x == 0) *slab_ptr(ticket, a.slab) = 0;       // the second stage's arrival counter (see nms_mask_kernel)
    nms_scan_col_body(a, c_begin, c_end, first_stage, last_stage, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
struct Layout {   // carve-up of the caller's workspace (per group)
    size_t counters, keys, boxes, scores, order, sboxes, sscores, keep_pos, mask, colw, total;
    int n_pad, n_tiles, m_max, pitch, ticket_index;
};

// FRCNN_NMS_SCAN=1: the one-chunk-per-trip single-wave scan of round 1 (A/B measurements); default: four chunks per trip
static bool scan_one_chunk_per_trip() {
    const char *e = frcnn_tune("FRCNN_NMS_SCAN");
    return e && e[0] == '1';
}

// blocks (one tile each) of a mask launch over columns [lo, hi) of the upper triangle
static int mask_blocks(int lo, int hi) { return (int)(((long long)hi * (hi + 1) - (long long)lo * (lo + 1)) / 2); }
// first-stage width of a staged NMS: 4 x post_nms_top_n boxes (a greedy NMS that keeps `limit` boxes has usually done so long before
// it has looked at 4 x limit of them: 610 of 6000 on the benchmark image); 0 = one stage
static int nms_stage_chunks(int pitch, int max_out) {
    const char *e = frcnn_tune("FRCNN_NMS_STAGE");
    if (e && e[0] == '0') return 0;
    if (max_out <= 0) return 0;
    const int s = (4 * max_out + kChunk - 1) / kChunk;
    return (s + 8 <= pitch) ? s : 0;                                   // not worth two launches for the last few columns
}

// workgroups of a second-stage mask launch (FRCNN_NMS_TAIL_WGS overrides: A/B measurements)
static int nms_tail_workgroups() {
    const char *e = frcnn_tune("FRCNN_NMS_TAIL_WGS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 4 * frcnn_cu_count();
}

static Layout make_layout(int n_total, int top_k, bool own_boxes) {
    Layout L;
    L.n_tiles = frcnn_cdiv(n_total > 0 ? n_total : 1, kSortTile);
    L.n_pad = L.n_tiles * kSortTile;
    L.m_max = (top_k > 0 && top_k < n_total) ? top_k : n_total;
    if (L.m_max < 1) L.m_max = 1;
    L.pitch = frcnn_cdiv(L.m_max, kChunk);
    size_t o = 0;
    L.ticket_index = 1 + L.n_pad / 256;                                               // [0] n_valid, [1 + b] tile b's count, then the stage-2 ticket
    L.counters = o; o += frcnn_align256((size_t)(2 + L.n_pad / 256) * sizeof(int));
    L.keys = o; o += frcnn_align256((size_t)L.n_pad * 8);
    L.boxes = o; o += own_boxes ? frcnn_align256((size_t)(n_total > 0 ? n_total : 1) * 16) : 0;
    L.scores = o; o += own_boxes ? frcnn_align256((size_t)(n_total > 0 ? n_total : 1) * 4) : 0;
    L.order = o; o += frcnn_align256((size_t)L.m_max * 4);
    L.sboxes = o; o += frcnn_align256((size_t)L.m_max * 16);
    L.sscores = o; o += frcnn_align256((size_t)L.m_max * 4);
    L.keep_pos = o; o += frcnn_align256((size_t)L.m_max * 4);
    L.mask = o; o += frcnn_align256((size_t)L.m_max * L.pitch * 8);
    L.colw = o; o += frcnn_align256((size_t)L.pitch * kChunk * 8);
    L.total = o;
    return L;
}

// Mask + sequential pass of one (possibly batched) NMS problem, in one or two stages (see nms_scan_col_kernel).
static void launch_mask_and_scan(hipStream_t stream, int groups, const Layout &L, const float *sboxes, int *counters, int top_k,
                                 double thresh, int max_out, unsigned long long *mask, const int32_t *order, const float *sscores,
                                 int32_t *keep_pos, int32_t *out_index, float *out_boxes, float *out_scores, int32_t *n_out,
                                 int out_capacity, size_t slab, size_t out_gs) {
    const dim3 blk(256);
    const ScanArgs sa{mask, L.pitch, counters, top_k, max_out, order, sboxes, sscores, keep_pos, out_index, out_boxes, out_scores, n_out,
                      out_capacity, slab, out_gs, (unsigned long long *)((char *)mask + (L.colw - L.mask))};
    int *ticket = counters + L.ticket_index;
    if (scan_one_chunk_per_trip()) {                       // round-1 row-form scan (A/B measurements)
        hipLaunchKernelGGL(nms_mask_kernel, dim3(mask_blocks(0, L.pitch), 1, groups), blk, 0, stream, sboxes, counters, top_k, thresh, mask,
                           L.pitch, slab, 0, L.pitch, max_out, 0, sa, ticket);
        if (L.pitch <= 128)
            hipLaunchKernelGGL(nms_scan_wave_kernel<2>, dim3(1, 1, groups), dim3(64), 0, stream, mask, L.pitch, counters, top_k, max_out, order,
                               sboxes, sscores, keep_pos, out_index, out_boxes, out_scores, n_out, out_capacity, slab, out_gs);
        else if (L.pitch <= kWaveChunks)
            hipLaunchKernelGGL(nms_scan_wave_kernel<4>, dim3(1, 1, groups), dim3(64), 0, stream, mask, L.pitch, counters, top_k, max_out, order,
                               sboxes, sscores, keep_pos, out_index, out_boxes, out_scores, n_out, out_capacity, slab, out_gs);
        else
            hipLaunchKernelGGL(nms_scan_kernel, dim3(1, 1, groups), blk, 0, stream, mask, L.pitch, counters, top_k, max_out, order, sboxes,
                               sscores, keep_pos, out_index, out_boxes, out_scores, n_out, out_capacity, slab, out_gs);
        return;
    }
    const int S = nms_stage_chunks(L.pitch, max_out);
    const int hi0 = S > 0 ? S : L.pitch;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(mask_blocks(0, hi0), 1, groups), blk, 0, stream, sboxes, counters, top_k, thresh, mask, L.pitch,
                       slab, 0, hi0, max_out, 0, sa, ticket);
    hipLaunchKernelGGL(nms_scan_col_kernel, dim3(1, 1, groups), dim3(64), 0, stream, sa, 0, hi0, 1, S > 0 ? 0 : 1, S > 0 ? ticket : (int *)nullptr);
    if (S > 0)            // second stage: the rest of the mask AND (by its last workgroup) the rest of the sequential pass, if still needed
        hipLaunchKernelGGL(nms_mask_kernel, dim3(std::min(mask_blocks(S, L.pitch), nms_tail_workgroups()), 1, groups), blk, 0, stream, sboxes,
                           counters, top_k, thresh, mask, L.pitch, slab, S, L.pitch, max_out, 1, sa, ticket);
}

}  // namespace

extern "C" {

size_t frcnn_nms_batched_workspace_bytes(int groups, int n) {
    if (groups < 1 || n < 0) return 0;
    return make_layout(n, 0, false).total * (size_t)groups;
}
size_t frcnn_nms_workspace_bytes(int n) { return frcnn_nms_batched_workspace_bytes(1, n); }

int frcnn_nms_batched(const float *dets, int groups, int n, double thresh, int max_out, int32_t *keep, int32_t *n_keep,
                      void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (groups < 1 || n < 0 || !n_keep || (n > 0 && (!dets || !keep))) return FRCNN_ERR_INVALID;
    if (n > kMaxChunks * kChunk) return FRCNN_ERR_INVALID;
    if (n == 0) { FRCNN_HIP_TRY(hipMemsetAsync(n_keep, 0, sizeof(int32_t) * groups, stream)); return FRCNN_OK; }
    const Layout L = make_layout(n, 0, false);
    if (!workspace || workspace_bytes < L.total * (size_t)groups) return FRCNN_ERR_INVALID;
    char *ws = (char *)workspace;
    const size_t gs = L.total;   // bytes of workspace per group (a multiple of 256)
    int *counters = (int *)(ws + L.counters);
    unsigned long long *keys = (unsigned long long *)(ws + L.keys);
    int32_t *order = (int32_t *)(ws + L.order);
    float *sboxes = (float *)(ws + L.sboxes);
    float *sscores = (float *)(ws + L.sscores);
    int32_t *keep_pos = (int32_t *)(ws + L.keep_pos);
    unsigned long long *mask = (unsigned long long *)(ws + L.mask);
    const dim3 blk(256);
    hipLaunchKernelGGL(dets_keys_kernel, dim3(frcnn_cdiv(L.n_pad, 256), 1, groups), blk, 0, stream, dets, n, L.n_pad, keys,
                       counters, (size_t)n * 5, gs);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_sort_kernel<false>), dim3(L.n_tiles, 1, groups), dim3(kSortThreads), 0, stream, keys, gs, DecodeArgs{},
                       Anchors{}, (int *)nullptr);
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(frcnn_cdiv(L.n_pad, 256), 1, groups), blk, 0, stream, keys, L.n_tiles, dets, 5,
                       dets + 4, 5, 0, counters, 0, order, sboxes, sscores, (size_t)n * 5, gs);
    launch_mask_and_scan(stream, groups, L, sboxes, counters, 0, thresh, max_out, mask, order, sscores, keep_pos, keep, (float *)nullptr,
                         (float *)nullptr, n_keep, (max_out > 0 && max_out < n) ? max_out : n, gs, (size_t)n);
    return frcnn_launch_status();
}

int frcnn_nms(const float *dets, int n, double thresh, int max_out, int32_t *keep, int32_t *n_keep, void *workspace,
              size_t workspace_bytes, void *stream) {
    return frcnn_nms_batched(dets, 1, n, thresh, max_out, keep, n_keep, workspace, workspace_bytes, stream);
}

size_t frcnn_proposals_workspace_bytes(int A, int H, int W, int pre_nms_top_n) {
    if (A < 1 || H < 1 || W < 1) return 0;
    return make_layout(A * H * W, pre_nms_top_n, true).total;
}

int frcnn_proposals(const float *rpn_cls_prob, const float *rpn_bbox_pred, int A, int H, int W, const double *anchors_host,
                    int feat_stride, int im_h, int im_w, float min_size, int pre_nms_top_n, int post_nms_top_n, double nms_thresh,
                    float *rois, float *probs, int32_t *n_out, int32_t *src_index, void *workspace, size_t workspace_bytes,
                    void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!rpn_cls_prob || !rpn_bbox_pred || !anchors_host || !rois || !probs || !n_out) return FRCNN_ERR_INVALID;
    if (A < 1 || A > 32 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int n = A * H * W;
    const Layout L = make_layout(n, pre_nms_top_n, true);
    if (L.m_max > kMaxChunks * kChunk) return FRCNN_ERR_INVALID;
    if (!workspace || workspace_bytes < L.total) return FRCNN_ERR_INVALID;
    char *ws = (char *)workspace;
    int *counters = (int *)(ws + L.counters);
    unsigned long long *keys = (unsigned long long *)(ws + L.keys);
    float *boxes = (float *)(ws + L.boxes);
    float *scores = (float *)(ws + L.scores);
    int32_t *order = (int32_t *)(ws + L.order);
    float *sboxes = (float *)(ws + L.sboxes);
    float *sscores = (float *)(ws + L.sscores);
    int32_t *keep_pos = (int32_t *)(ws + L.keep_pos);
    unsigned long long *mask = (unsigned long long *)(ws + L.mask);
    Anchors anc;
    for (int a = 0; a < A; ++a)
        for (int c = 0; c < 4; ++c) anc.a[a][c] = anchors_host[a * 4 + c];
    const dim3 blk(256);
    const DecodeArgs dec{rpn_cls_prob, rpn_bbox_pred, A, H, W, feat_stride, im_h, im_w, min_size, boxes, scores};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_sort_kernel<true>), dim3(L.n_tiles), dim3(kSortThreads), 0, stream, keys, (size_t)0, dec, anc, counters);
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(frcnn_cdiv(L.n_pad, 256)), blk, 0, stream, keys, L.n_tiles, boxes, 4, scores, 1,
                       pre_nms_top_n, counters, L.n_tiles, order, sboxes, sscores, (size_t)0, (size_t)0);
    launch_mask_and_scan(stream, 1, L, sboxes, counters, pre_nms_top_n, nms_thresh, post_nms_top_n, mask, order, sscores, keep_pos, src_index,
                         rois, probs, n_out, (post_nms_top_n > 0 && post_nms_top_n < L.m_max) ? post_nms_top_n : L.m_max, (size_t)0, (size_t)0);
    return frcnn_launch_status();
}

}  // extern "C"
