// linear_bf16.hip -- bf16 L.Linear for the RCNN head (fc6 / fc7 / cls_score ‖ bbox_pred of BASELINE configs[2]) as a WEIGHT STREAM:
// the weight matrix is re-tiled once, at load time, into the exact LDS image of the kernel's weight panels, and the GEMM walks it once
// through a five-stage LDS-DMA ring with counted vmcnt waits.  Replaces L.Linear (models/faster_rcnn.py:33-36,127-134) on the bf16 line.
//
// Why (profiles/r05_linear_bf16_splits_micro.txt): linear_dma_bf16_kernel (conv_bf16.hip) streams fc6's 206 MB of
// weights at 2.6 TB/s -- 79 us against a 33-35 us floor -- because it is a two-stage ring that DRAINS (vmcnt(0) + barrier) every 64-k
// chunk: one chunk of prefetch per workgroup, two workgroups per CU, so a CU has ~32 KB of weights in flight against a memory latency of
// ~1 us; its 160-row M tiles walk the weight matrix twice; and its weight reads are 128 rows x 128 B at a 50 KB stride (one DRAM page per
// 128 B).  Here:
//   * ONE workgroup per CU holds ALL rows of x (BM = 320 >= the 300 RoIs: the matrix is walked once) and 128 output columns; its four
//     waves are 2 (row halves) x 2 (column halves): a wave owns 5 x 2 tiles of 32 x 32 = 160 accumulator registers, and reads
//     5 + 2 fragments (ds_read_b128) per 10 MFMAs;
//   * a chunk is 32 k (two MFMA k-steps): 20 KB of x (from L2: the K slice of a split is 1.9 MB, and the workgroups of one split are
//     dealt to ONE XCD -- split = blockIdx % splits, eight splits on eight XCDs) + 8 KB of weights (from HBM); FIVE stages of 28 KB:
//     three to four chunks (~1 us of MFMA work) are in flight behind the one being multiplied, their 1 KB pieces issued BETWEEN the
//     MFMAs (conv_bf16_strip.h's schedule: a piece rides along with an MFMA for free), waits are counted (vmcnt((NS-2) * 7)), one
//     fence-less barrier per chunk, placed between the chunk's two k-steps so that the next chunk's first fragments are read under
//     the second k-step's MFMAs;
//   * the weights are read as 8 KB CONTIGUOUS tiles, a workgroup's stream is 98 consecutive tiles (784 KB): frcnn_linear_bf16_tile_w
//     writes tile (n block, k chunk) as the swizzled LDS image itself, so a DMA piece is a linear 1 KB copy;
//   * LDS image: row pitch 64 B = four 16-byte groups, group g of row r in slot g ^ ((r >> 2) & 3): the sixteen lanes the hardware
//     serves together in a ds_read_b128 ({0-3, 12-15, 20-27}, ...) fall on sixteen different 16-byte columns of the 256-byte bank row.
// Split-K partial slabs + linear_ring_reduce_kernel (bias, ReLU, optional bf16 output) as before: fp32 partial sums added in split order
// (deterministic).  Same products and the same fp32 accumulation per 32x32x16 MFMA as linear_dma_bf16_kernel; the order of the k-steps
// within a split is the same ascending order, so for equal split boundaries the results are bit-identical.
#include "frcnn_common.h"
#include "frcnn_reduce.h"
#include <frcnn_buffer.h>
#include <frcnn_intrin.h>

namespace {

constexpr int kRK = 32;                 // k per chunk
constexpr int kRBN = 128;               // output columns per workgroup
constexpr int kRWBytes = kRBN * kRK * 2;   // one weight tile: 8 KB
constexpr int kRQB = 4;                 // pieces of a chunk issued in the second k-step's MFMAs (the rest under the next chunk's first k-step)

__device__ __forceinline__ uint16_t ring_f32_to_bf16(float f) { return (uint16_t)(frcnn_pack_bf16x2(f, 0.0f) & 0xffffu); }

// w (N, K) bf16 row-major -> tiles [nb][kc] of 8 KB: 16-byte slot s = row * 4 + (g ^ ((row >> 2) & 3)) holds w[nb * 128 + row][kc * 32 + 8 g .. + 7]
// (rows past N are zero).  One thread per slot.
__global__ void __launch_bounds__(256)
linear_tile_w_bf16_kernel(const uint16_t *__restrict__ w, int N, int K, uint16_t *__restrict__ wt, size_t slots) {
    const int KC = K / kRK;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
        const int s = (int)(i & 511);
        const size_t tile = i >> 9;
        const int kc = (int)(tile % KC), nb = (int)(tile / KC);
        const int row = s >> 2, g = (s & 3) ^ ((row >> 2) & 3);
        const int n = nb * kRBN + row;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (n < N) v = *reinterpret_cast<const uint4 *>(w + (size_t)n * K + kc * kRK + g * 8);
        reinterpret_cast<uint4 *>(wt)[i] = v;
    }
}

// MT = 32-row tiles per wave (BM = 64 MT rows per workgroup), NS = ring stages
// ABL (timing ablations, WRONG results, only in -DFRCNN_TIMING_ABLATIONS builds: scripts/micro/build_micro.sh -> liblinear_abl.so): 1 no DMA after the
// prologue, 2 no fragment reads after the prologue, 4 no MFMAs (operands still waited for), 8 no hand-over (counted wait + barrier)
#ifdef FRCNN_TIMING_ABLATIONS
__device__ __forceinline__ void ring_keep(const uint4 &v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
#endif
// TRN (round 6, the default where N % 4 == 0): the MFMA's operands swapped -- A = the weight fragment, B = the x fragment: the accumulator tile is the
// TRANSPOSE (register r of lane l = output COLUMN (r & 3) + 8 (r >> 2) + 4 khalf of output ROW l31), the same sixteen products per element summed by the same
// instruction, so four consecutive registers are four consecutive floats of one slab row and the partial tile leaves as 40 sixteen-byte stores per lane
// instead of 160 four-byte ones.
template <int MT, int NS, int ABL = 0, bool TRN = false>
__global__ void __launch_bounds__(256)
linear_ring_bf16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wt, float *__restrict__ part, int M, int N, int K, int splits, int cps, int flags) {
    constexpr int BM = 64 * MT, XQ = MT, WQ = 2, PPW = XQ + WQ;
    constexpr int XBYTES = BM * 64, STAGE = XBYTES + kRWBytes;
    constexpr int QB = kRQB < PPW ? kRQB : PPW - 1, QA = PPW - QB;
    constexpr int NM = 2 * MT;                                              // MFMAs per k-step per wave
    static_assert(NS * STAGE <= 160 * 1024 - 64, "LDS");
    static_assert((NS - 2) * PPW + QB <= 63, "vmcnt holds 63");
    static_assert(2 * (MT + 2) <= 15, "two k-steps of fragment reads in flight: lgkmcnt holds 15");
    static_assert(NS - 2 <= 4, "the tail's counted waits are written out for k <= 4");
    static_assert(QA <= NM && QB <= NM, "a piece per MFMA at most");
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const bool wnt = (flags & 1) != 0;                                       // non-temporal weight stream (A/B: FRCNN_LINEAR_RING_FLAGS)
    const int l31 = lane & 31, khalf = lane >> 5;
    // split fastest: consecutive workgroups go to consecutive XCDs, so with eight splits every XCD's L2 holds ONE K slice of x
    const int split = (int)blockIdx.x % splits, nb = (int)blockIdx.x / splits;
    const int m0 = (int)blockIdx.y * BM;
    const int KC = K / kRK;
    const int c_begin = split * cps;
    const int nch = min(KC, c_begin + cps) - c_begin;                        // >= 1 (host)
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)M * K * 2));
    const frcnn_buf_t wbuf = frcnn_make_buf(wt, (uint32_t)((size_t)((N + kRBN - 1) / kRBN) * KC * kRWBytes));

    // per-lane source offsets (chunk c_begin) of this wave's pieces: x piece p = wave + 4 q covers rows 16 p .. 16 p + 15 of the panel
    uint32_t xoff[XQ], woff[WQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        const int s = (wave + 4 * q) * 64 + lane, row = s >> 2, g = (s & 3) ^ ((row >> 2) & 3);
        const int gr = m0 + row;
        xoff[q] = gr < M ? (uint32_t)(((size_t)gr * K + (size_t)c_begin * kRK + 8 * g) * 2) : kBufOob;
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q)
        woff[q] = (uint32_t)(((size_t)nb * KC + c_begin) * kRWBytes + (wave + 4 * q) * 1024 + lane * 16);
#ifdef FRCNN_TIMING_ABLATIONS
    float4 xr[XQ];                                                          // (ablation 16: the x pieces as plain 16-byte loads into registers instead of LDS-DMA -- the rate of that path)
#pragma unroll
    for (int q = 0; q < XQ; ++q) xr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    auto issue_piece = [&](int q, int chunk, int stage) {
        unsigned char *dst = ring + stage * STAGE + wave * 1024;
#ifdef FRCNN_TIMING_ABLATIONS
        if constexpr ((ABL & 16) != 0) {
            if (q < XQ) { xr[q] = frcnn_buf_load_f32x4_soff(xbuf, xoff[q], (uint32_t)chunk * (kRK * 2)); return; }
        }
#endif
        if (q < XQ) frcnn_buf_load_lds_b128(xbuf, dst + q * 4096, xoff[q], (uint32_t)chunk * (kRK * 2));
        else if (wnt) frcnn_buf_load_lds_b128_nt(wbuf, dst + XBYTES + (q - XQ) * 4096, woff[q - XQ], (uint32_t)chunk * kRWBytes);
        else frcnn_buf_load_lds_b128(wbuf, dst + XBYTES + (q - XQ) * 4096, woff[q - XQ], (uint32_t)chunk * kRWBytes);
    };
    // at most n of this wave's pieces still in flight (n = PPW * k + extra, a handful of values: a switch over compile-time immediates)
    auto wait_chunks = [&](int k, bool plus_qb) {
        if (plus_qb) {
            switch (k) {
            case 0: frcnn_wait_vmcnt<QB>(); break;
            case 1: frcnn_wait_vmcnt<PPW + QB>(); break;
            case 2: frcnn_wait_vmcnt<(NS > 3 ? 2 : 0) * PPW + (NS > 3 ? QB : 0)>(); break;
            case 3: frcnn_wait_vmcnt<(NS > 4 ? 3 : 0) * PPW + (NS > 4 ? QB : 0)>(); break;
            case 4: frcnn_wait_vmcnt<(NS > 5 ? 4 : 0) * PPW + (NS > 5 ? QB : 0)>(); break;
            default: frcnn_wait_vmcnt<(NS > 6 ? 5 : 0) * PPW + (NS > 6 ? QB : 0)>(); break;
            }
        } else {
            switch (k) {
            case 0: frcnn_wait_vmcnt<0>(); break;
            case 1: frcnn_wait_vmcnt<PPW>(); break;
            case 2: frcnn_wait_vmcnt<(NS > 3 ? 2 : 0) * PPW>(); break;
            case 3: frcnn_wait_vmcnt<(NS > 4 ? 3 : 0) * PPW>(); break;
            case 4: frcnn_wait_vmcnt<(NS > 5 ? 4 : 0) * PPW>(); break;
            default: frcnn_wait_vmcnt<(NS > 6 ? 5 : 0) * PPW>(); break;
            }
        }
    };

    // prologue: chunks 0 .. NS-2 whole, and the QB first pieces of chunk NS-1 ("the second k-step of iteration -1")
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nch) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) issue_piece(q, c, c);
        }
    if (NS - 1 < nch) {
#pragma unroll
        for (int q = 0; q < QB; ++q) issue_piece(q, NS - 1, NS - 1);
    }
    __builtin_amdgcn_sched_barrier(0);

    // fragment byte offsets inside a stage (swizzled): A = x row (wm MT + i) 32 + l31, B = weight row (2 wn + j) 32 + l31; group 2 ks + khalf
    const int sw = (l31 >> 2) & 3;
    uint32_t a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const uint32_t slot = (uint32_t)(((2 * ks + khalf) ^ sw) << 4);
        a_off[ks] = (uint32_t)((wm * MT * 32 + l31) * 64) + slot;
        b_off[ks] = (uint32_t)(XBYTES + (wn * 64 + l31) * 64) + slot;
    }
    frcnn_f32x16 acc[NM];
#pragma unroll
    for (int a = 0; a < NM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    uint4 fa[2][MT], fb[2][2];
    auto read_frags = [&](const unsigned char *st, int ks) {
        if constexpr ((ABL & 2) != 0) { if (st != ring) return; }           // (ablation: only the prologue's reads)
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[ks][i] = *reinterpret_cast<const uint4 *>(st + a_off[ks] + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[ks][j] = *reinterpret_cast<const uint4 *>(st + b_off[ks] + j * 2048);
    };
    // one k-step: NM MFMAs, with `npieces` DMA pieces q0 .. of chunk `chunk` (-> LDS stage `stage`) spread between them
    auto kstep = [&](int ks, int q0, int npieces, int chunk, int stage) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#ifdef FRCNN_TIMING_ABLATIONS
                if constexpr ((ABL & 4) != 0) { ring_keep(fa[ks][i]); ring_keep(fb[ks][j]); } else
#endif
                acc[i * 2 + j] = TRN ? frcnn_mfma_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i * 2 + j]) : frcnn_mfma_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i * 2 + j]);
                const int m = i * 2 + j;
                if (npieces > 0 && (ABL & 1) == 0) {
                    const int p0 = m * npieces / NM, p1 = (m + 1) * npieces / NM;          // folds: m and npieces are compile-time at every call site
                    if (p1 != p0) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(q0 + p1 - 1, chunk, stage);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    };

    // chunk 0 landed (behind it in this wave's queue: chunks 1 .. NS-2 and the QB pieces of chunk NS-1, as far as they exist)
    wait_chunks(min(NS - 2, nch - 1), NS - 1 < nch);
    frcnn_barrier_nofence();
    read_frags(ring, 0);
    __builtin_amdgcn_sched_barrier(0);

    // Iteration c (LDS stage s_cur): read k-step 1's fragments; k-step 0's MFMAs, under them the last QA pieces of chunk c - 1 + NS (its stage was
    // freed by the barrier of iteration c - 1); hand-over: chunk c + 1 has landed for this wave (counted wait), barrier (for every wave -- and every
    // wave's reads of chunk c are complete: its stage may be refilled), chunk c + 1's k-step-0 fragments are read under k-step 1's MFMAs, which
    // carry the first QB pieces of chunk c + NS into the stage just freed.
    int s_cur = 0, s_prev = NS - 1, c = 0;
    for (; c + NS < nch; ++c) {                                            // steady state: chunks c - 1 + NS and c + NS both exist
        const unsigned char *st = ring + s_cur * STAGE;
        const int s_next = s_cur + 1 == NS ? 0 : s_cur + 1;
        read_frags(st, 1);
        __builtin_amdgcn_sched_barrier(0);
        kstep(0, QB, QA, c - 1 + NS, s_prev);
#ifdef FRCNN_TIMING_ABLATIONS
        if constexpr ((ABL & 16) != 0) {                                    // the registers loaded one iteration ago are "used" here (their wait is the compiler's)
#pragma unroll
            for (int q = 0; q < XQ; ++q) asm volatile("" ::"v"(xr[q].x), "v"(xr[q].y), "v"(xr[q].z), "v"(xr[q].w));
        }
#endif
        if constexpr ((ABL & 8) == 0) {
            if constexpr ((ABL & 1) == 0) frcnn_wait_vmcnt<(NS - 2) * PPW>();
            frcnn_barrier_nofence();
        }
        read_frags(ring + s_next * STAGE, 0);
        __builtin_amdgcn_sched_barrier(0);
        kstep(1, 0, QB, c + NS, s_cur);
        s_prev = s_cur;
        s_cur = s_next;
    }
    // The last min(NS, nch) chunks, straight-line (one block per value of k = whole chunks still queued behind chunk c + 1 at the hand-over; a loop with
    // the piece / hand-over conditions inside made the compiler shuffle all 160 accumulators AGPR -> VGPR -> AGPR around its branches: conv_bf16_strip.h).
    // k == NS - 2 only happens at c == nch - NS, the one tail iteration that still has pieces to issue (the last QA of chunk nch - 1).
#pragma unroll
    for (int k = NS - 2; k >= 0; --k) {
        if (nch - c - 2 == k) {
            const int s_next = s_cur + 1 == NS ? 0 : s_cur + 1;
            read_frags(ring + s_cur * STAGE, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (k == NS - 2) kstep(0, QB, QA, c - 1 + NS, s_prev);
            else kstep(0, 0, 0, 0, 0);
            if (k == 0) frcnn_wait_vmcnt<0>();
            else if (k == 1) frcnn_wait_vmcnt<PPW>();
            else if (k == 2) frcnn_wait_vmcnt<2 * PPW>();
            else if (k == 3) frcnn_wait_vmcnt<3 * PPW>();
            else frcnn_wait_vmcnt<4 * PPW>();
            frcnn_barrier_nofence();
            read_frags(ring + s_next * STAGE, 0);
            __builtin_amdgcn_sched_barrier(0);
            kstep(1, 0, 0, 0, 0);
            s_prev = s_cur;
            s_cur = s_next;
            ++c;
        }
    }
    read_frags(ring + s_cur * STAGE, 1);                                   // the last chunk: nothing to hand over to
    __builtin_amdgcn_sched_barrier(0);
    kstep(0, 0, 0, 0, 0);
    kstep(1, 0, 0, 0, 0);

    // partial tile -> this split's slab (register r of lane l = row (r & 3) + 8 (r >> 2) + 4 khalf, column l31: 128-byte runs of a row); straight-line
    // buffer stores, rows past M / columns past N get an out-of-range offset and store nothing
    const frcnn_buf_t pbuf = frcnn_make_buf(part + (size_t)split * M * N, (uint32_t)((size_t)M * N * 4));
    if constexpr (TRN) {
        // transposed accumulators: lane = slab row m, registers 4 g .. 4 g + 3 = columns n0 + 8 g + 4 khalf .. + 3 (N % 4 == 0: a quad is inside N or outside)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m0 + (wm * MT + i) * 32 + l31;
                const int n0 = nb * kRBN + (wn * 2 + j) * 32 + 4 * khalf;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + 8 * g;
                    const uint32_t off = (m < M && n < N) ? (uint32_t)(m * N + n) * 4u : kBufOob;
                    const frcnn_f32x16 &a = acc[i * 2 + j];
                    const float4 v = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                    if (flags & 2) frcnn_buf_store_f32x4_soff<16>(pbuf, off, 0u, v);
                    else frcnn_buf_store_f32x4_soff<0>(pbuf, off, 0u, v);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nb * kRBN + (wn * 2 + j) * 32 + l31;
            const int mrow = m0 + (wm * MT + i) * 32 + 4 * khalf;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow + (r & 3) + 8 * (r >> 2);
                const uint32_t off = (m < M && n < N) ? (uint32_t)(m * N + n) * 4u : kBufOob;
                if (flags & 2) frcnn_buf_store_f32_aux<16>(pbuf, off, acc[i * 2 + j][r]);
                else frcnn_buf_store_f32(pbuf, off, acc[i * 2 + j][r]);
            }
        }
}

__global__ void __launch_bounds__(256)
linear_ring_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias, void *__restrict__ y, int M, int N, int splits, int relu, int out_bf16) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float b = bias[i % N];
        float v = frcnn_sum_splits(part, total, i, splits);
        v += b;
        if (relu) v = fmaxf(v, 0.0f);
        if (out_bf16) reinterpret_cast<uint16_t *>(y)[i] = ring_f32_to_bf16(v);
        else reinterpret_cast<float *>(y)[i] = v;
    }
}

struct RingPlan { int mt, ns, mblocks, nblocks, splits, cps; };
static RingPlan plan_ring(int M, int N, int K) {
    RingPlan p;
    p.mt = M > 192 ? 5 : (M > 64 ? 3 : 1);
    p.ns = p.mt == 5 ? 5 : 6;
    p.mblocks = frcnn_cdiv(M, 64 * p.mt);
    p.nblocks = frcnn_cdiv(N, kRBN);
    const int KC = K / kRK, tiles = p.mblocks * p.nblocks;
    int splits = frcnn_cdiv(frcnn_cu_count(), tiles);
    if (splits > KC / (2 * p.ns)) splits = KC / (2 * p.ns);               // a split is at least two rings long
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    p.cps = frcnn_cdiv(KC, splits);
    p.splits = frcnn_cdiv(KC, p.cps);
    return p;
}

}  // namespace

extern "C" {

size_t frcnn_linear_bf16_tiled_bytes(int N, int K) {
    if (N < 1 || K < 1 || (K % kRK) != 0) return 0;
    return (size_t)frcnn_cdiv(N, kRBN) * (K / kRK) * kRWBytes;
}

int frcnn_linear_bf16_tile_w(const uint16_t *w, int N, int K, uint16_t *w_tiled, void *stream) {
    if (!w || !w_tiled || N < 1 || K < 1 || (K % kRK) != 0) return FRCNN_ERR_INVALID;
    const size_t slots = frcnn_linear_bf16_tiled_bytes(N, K) / 16;
    if (slots * 16 >= (1ull << 31)) return FRCNN_ERR_UNSUPPORTED;          // the kernel addresses the tiles through a 32-bit buffer descriptor
    const int blocks = (int)((slots + 255) / 256 < 8192 ? (slots + 255) / 256 : 8192);
    hipLaunchKernelGGL(linear_tile_w_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, N, K, w_tiled, slots);
    return frcnn_launch_status();
}

size_t frcnn_linear_bf16_tiled_workspace_bytes(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1 || (K % kRK) != 0) return 0;
    const RingPlan p = plan_ring(M, N, K);
    return frcnn_align256((size_t)p.splits * M * N * sizeof(float));
}

int frcnn_linear_bf16_tiled(const uint16_t *x, const uint16_t *w_tiled, const float *bias, void *y, int M, int N, int K, int relu, int out_bf16,
                            void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w_tiled || !bias || !y || M < 1 || N < 1 || K < 1 || (K % kRK) != 0) return FRCNN_ERR_INVALID;
    if ((size_t)M * K * 2 >= (1ull << 31) || frcnn_linear_bf16_tiled_bytes(N, K) >= (1ull << 31) || (size_t)M * N * 4 >= (1ull << 31)) return FRCNN_ERR_UNSUPPORTED;
    const RingPlan p = plan_ring(M, N, K);
    if (!workspace || workspace_bytes < (size_t)p.splits * M * N * sizeof(float)) return FRCNN_ERR_INVALID;
    float *part = (float *)workspace;
    const dim3 grid(p.nblocks * p.splits, p.mblocks);
    const int flags = frcnn_tune_int("FRCNN_LINEAR_RING_FLAGS", 0);        // A/B hook: bit 0 = nt weight reads (measured SLOWER: 93.7 vs 83.6 us, the DMA stream alone 84 vs 62), bit 1 = write-through slab stores (fc6 +4 us, fc7 -1 us); 0 = the measured pick
#ifdef FRCNN_TIMING_ABLATIONS
    const int abl = frcnn_tune_int("FRCNN_LINEAR_RING_ABL", 0);
    if (p.mt == 5 && abl) {
        switch (abl) {
#define RING_ABL(A) case A: hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_ring_bf16_kernel<5, 5, A>), grid, dim3(256), 0, stream, x, w_tiled, part, M, N, K, p.splits, p.cps, flags); break;
        RING_ABL(1) RING_ABL(2) RING_ABL(3) RING_ABL(4) RING_ABL(5) RING_ABL(6) RING_ABL(7) RING_ABL(8) RING_ABL(9) RING_ABL(12) RING_ABL(15) RING_ABL(16) RING_ABL(22) RING_ABL(18) RING_ABL(20)
#undef RING_ABL
        default: return FRCNN_ERR_INVALID;
        }
    } else
#endif
    {
        const bool trn = (N & 3) == 0 && !(flags & 4);                       // (bit 2 of FRCNN_LINEAR_RING_FLAGS: round 6's first form, 4-byte slab stores -- A/B)
#define RING_LAUNCH(MT_, NS_) do { \
            if (trn) hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_ring_bf16_kernel<MT_, NS_, 0, true>), grid, dim3(256), 0, stream, x, w_tiled, part, M, N, K, p.splits, p.cps, flags); \
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(linear_ring_bf16_kernel<MT_, NS_, 0, false>), grid, dim3(256), 0, stream, x, w_tiled, part, M, N, K, p.splits, p.cps, flags); } while (0)
        if (p.mt == 5) RING_LAUNCH(5, 5);
        else if (p.mt == 3) RING_LAUNCH(3, 6);
        else RING_LAUNCH(1, 6);
#undef RING_LAUNCH
    }
    const size_t total = (size_t)M * N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(linear_ring_reduce_kernel, dim3(blocks), dim3(256), 0, stream, part, bias, y, M, N, p.splits, relu, out_bf16);
    return frcnn_launch_status();
}

}  // extern "C"
