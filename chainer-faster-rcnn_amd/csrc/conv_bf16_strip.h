// conv_bf16_strip.h -- the 3x3 bf16 convolution with one wave per SIMD and the latency hiding INSIDE the wave ("strip" forms; included by
// conv_bf16.hip inside its anonymous namespace).  Written late in round 3 on the host emulator, then timed and compared bit for bit with
// conv_dma_bf16_kernel on the MI355X through the torch-free harness (scripts/micro/conv_bf16_micro --check; profiles/r03_conv_bf16_strip_*.txt,
// DESIGN 3.8b): forms D and C are default picks of frcnn_conv_bf16_ws (conv2_2 ... conv4_3: -4 ... -15 %, the 38x63 launches 23.1 -> 15.1 us),
// A and B stay selectable (FRCNN_BF16_DMA=901 / 902; 903 = C, 909 = D; 907 / 908 are two measured-and-not-adopted shapes kept under test).
// (Measured and removed again: form A with a two-stage ring, with its DMA pieces front-loaded, form B with three stages: all within 0.5 us.)
//
// Why another form (profiles/r03_conv_bf16_micro.txt): conv_dma_bf16_kernel hides latency with co-resident workgroups,
// which ties it to small tiles (64 couts x 4..8 rows: 200..350 B of LDS-DMA and 0.75..1.2 fragment reads per MFMA), and its large
// tiles lose to tile-count quantisation (320 tiles on 256 CUs).  Here the launch is shaped to the chip instead: the grid of
// (cout tile, 32-px column, row block) is sized to about one workgroup per CU, each wave owns 32*COB couts x RW rows x 32 px
// (A fragments reused RW times, B fragments 3*COB times from registers: 0.43 reads and ~120 B of DMA per MFMA at COB 2, RW 5), and
// the latency that co-resident workgroups used to cover is covered inside the wave by a software pipeline:
//   * an NS-stage LDS-DMA ring whose pieces are issued BETWEEN the MFMAs (profiles/r03_mfma_filler_micro.txt: one instruction
//     rides along with a bf16 MFMA for free; bunched, a stage's pieces hold the vector-memory path for hundreds of clocks),
//   * fragments read one (tap) group ahead of their MFMAs, at most 14 reads in flight (lgkmcnt counts to 15: with more the
//     compiler's only correct wait is lgkmcnt(0)), in an order pinned with __builtin_amdgcn_sched_barrier -- left alone the
//     scheduler read every fragment just in time (ds_read, s_waitcnt lgkmcnt(0), MFMA: a round trip per few MFMAs),
//   * the stage hand-over (counted vmcnt wait + barrier) BEFORE the last tap group of a stage, so that the first two groups'
//     fragments of the next stage are in flight under that group's MFMAs and no LDS latency is exposed at a stage boundary.
// The four waves of a workgroup are RG row groups x CW cout ways x KW K ways (layers with too few pixels for 1024 waves split
// the K loop over waves; the KW partial accumulators meet in LDS at the end):
//   form A  COB 2, RW 5, RG 4, CW 1, KW 1: 64 couts x 20 rows x 32 px, 3 stages of 42 KB   (150 x 250 maps: 256 workgroups)
//   form B  COB 1, RW 5, RG 2, CW 2, KW 1: 64 couts x 10 rows x 32 px, 4 stages of 34 KB   ( 75 x 125 maps: 256 workgroups)
//   form C  COB 1, RW 5, RG 1, CW 1, KW 4: 32 couts x  5 rows x 32 px, 2 stages of 68 KB   ( 38 x  63 maps: 256 workgroups)
//   form D  form B's waves, 2 stages of 34 KB, <= 256 registers: TWO workgroups per CU -- each still one wave per SIMD with the pipeline
//           below, and each covering the other's prologue and epilogue; the best form on every launch with >= 8 K-chunks and >= one tile per CU
// Same operands, same LDS image (pitch 32 B, XOR swizzle on source offsets and fragment reads) and, for KW 1, the same
// (chunk, tap) accumulation order per output as every conv_dma_bf16_kernel variant: bit-identical results; KW > 1 sums KW
// partial accumulators in ascending K-way order (deterministic; fp32 rounding differs from the single chain).
//
// The waves of one K way (GW = RG * CW of them) move the pieces of that K way's chunk round-robin (so all of a wave's loads of a
// stage carry one scalar offset), halo pieces first: the halo region is padded to a multiple of GW pieces, which makes "halo or
// weights" a compile-time fact per piece slot.  Waits are the counted vmcnt of conv_dma_bf16_kernel (a wave's loads per stage
// are a compile-time pair of values), one fence-less barrier per stage.
#pragma once
#include <type_traits>

template <int COB, int RW, int RG, int CW, int KW, int NS, int NW = 4>
struct StripShape {
    static constexpr int KS = 3, TAPS = 9, PAD = 1;
    static constexpr int GW = RG * CW;                                    // waves per K way
    static constexpr int BCO = 32 * COB * CW;                             // couts per workgroup
    static constexpr int TR = RG * RW;                                    // tile rows
    static constexpr int HR = TR + KS - 1, HPX = 32 + KS - 1;
    static constexpr int IN_ROWS = HR * HPX;                              // halo pixels, 32 B each
    static constexpr int IN_PIECES = ((IN_ROWS * 2 + 63) / 64 + GW - 1) / GW * GW;   // 1 KB pieces, padded to a multiple of GW
    static constexpr int W_PIECES = TAPS * BCO * 2 / 64;                  // 18 (64 couts) or 9 (32): whole pieces
    static constexpr int IN_Q = IN_PIECES / GW;                           // halo pieces per wave
    static constexpr int W_Q = (W_PIECES + GW - 1) / GW;                  // weight pieces per wave (the last one may not exist)
    static constexpr int PPW = IN_Q + W_Q;
    static constexpr int W_FULL = W_PIECES - GW * (W_Q - 1);              // waves gw < W_FULL own W_Q weight pieces, the others W_Q - 1
    static constexpr int KW_BYTES = (IN_PIECES + W_PIECES) * 1024;        // one K-way's chunk
    static constexpr int IN_BYTES = IN_PIECES * 1024;
    static constexpr int STAGE_BYTES = KW * KW_BYTES;
    static constexpr int NACC = COB * RW;
    static constexpr int GM = NACC;                                       // MFMAs per tap group
    static constexpr int OP = BCO * 2 + 16;                               // epilogue tile: bytes per pixel (bf16 couts + pad)
    static constexpr int OT_BYTES = TR * 32 * OP;
    static constexpr int PART_BYTES = KW == 1 ? 0 : GW * KW * (KW - 1) * NACC * (4 / KW) * 1024;
    static constexpr int RING_BYTES = NS * STAGE_BYTES;
    static constexpr int LDS_BYTES = RING_BYTES > OT_BYTES + PART_BYTES ? RING_BYTES : OT_BYTES + PART_BYTES;
    // DMA pieces of the next-but-(NS-2) stage go out between the MFMAs of tap groups 0 .. 7 (the hand-over sits before group 8): evenly
    // over all of them with >= 3 stages (the data has more than a whole stage to arrive); with 2 stages it has to arrive within
    // THIS stage, so as early as one piece per two MFMAs allows
    static constexpr int SPAN = NS >= 3 ? 8 * GM : (2 * (PPW + 1) < 8 * GM ? 2 * (PPW + 1) : 8 * GM);
    static_assert(GW * KW == NW && (NW == 4 || NW == 8), "waves");
    static_assert(KW == 1 || KW == 2 || KW == 4, "K ways");
    static_assert(LDS_BYTES <= 160 * 1024 - 64, "LDS");                    // (the kernel checks WPE * LDS_BYTES)
    static_assert(NS >= 2 && NS <= 4 && (NS - 1) * PPW <= 63, "vmcnt holds 63");
    static_assert(PPW + 1 <= SPAN, "a piece per MFMA at most");
    static_assert(2 * (COB + RW) <= 15, "two tap groups of fragment reads in flight: lgkmcnt holds 15");
};

#ifdef FRCNN_TIMING_ABLATIONS
__device__ __forceinline__ void strip_keep(const uint4 &v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
#endif

// ABL (timing ablations, WRONG results, only in -DFRCNN_TIMING_ABLATIONS builds): 1 no DMA after the prologue, 2 no fragment reads after the
// prologue, 4 no MFMAs (their operands are still waited for), 8 no stage hand-over (wait + barrier)
// WPE = workgroups per CU the form is compiled for (2: at most 256 registers and half the LDS -- form D and experiment 908)
// DIRECT (910; the default rule's form D: -0.3 ... -0.9 us per launch, bit-identical on the MI355X, r03 probe 9): the bf16 output of a launch without the fused pool leaves straight from the accumulator quads
// -- a quad is four consecutive couts of one pixel = one aligned 8-byte piece of the channel-blocked record, a wave's store covers 32 pixels x 16 B -- instead
// of through the LDS transpose: no barrier, no LDS round trip, and no use of the ring by the epilogue (the condition for a persistent tile loop)
// NW = waves per workgroup: 4 (one per SIMD) or 8 (form E, round 4: two per SIMD in ONE workgroup -- form A's 64-cout x 20-row tile on form D's waves; the
// two waves of a SIMD share one stage's weight panel instead of each workgroup fetching its own: 42 KB of DMA per 360 MFMAs where form D moves 2 x 31 KB)
template <int COB, int RW, int RG, int CW, int KW, int NS, int ABL = 0, int WPE = 1, bool DIRECT = false, int NW = 4>
__global__ void __launch_bounds__(64 * NW, WPE)
conv_strip_bf16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wp, const float *__restrict__ bias, void *__restrict__ y,
                       int CinP, int Cout, int CoutP, int H, int W, int relu, int out_mode, int xtiles, int ytiles, int cotiles) {
    using S = StripShape<COB, RW, RG, CW, KW, NS, NW>;
    static_assert(WPE * S::LDS_BYTES <= 160 * 1024 - 64 * WPE, "LDS of WPE workgroups");
    constexpr int KS = 3, TAPS = 9, PAD = 1, HPX = S::HPX, BCO = S::BCO, TR = S::TR, NACC = S::NACC, PPW = S::PPW, IN_Q = S::IN_Q, OP = S::OP, GW = S::GW;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[S::LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = wave % GW, kw = wave / GW;                             // wave within its K way; K way
    const int rg = gw % RG, cw = gw / RG;                                 // row group; cout way
    const int l31 = lane & 31, khalf = lane >> 5;
    // cout tile fastest: consecutive workgroups go to consecutive XCDs, so with 8 (4, 2) cout tiles an XCD's L2 sees one (two, four)
    // weight slab(s) of the layer, and the workgroups of one XCD walk neighbouring pixel columns of the same rows
    const int cot = (int)blockIdx.x % cotiles, pt = (int)blockIdx.x / cotiles;
    const int tx = pt % xtiles, ty = pt / xtiles;
    const int x0 = tx * 32, y0 = ty * TR, co0 = cot * BCO;
    const int nsc = CinP / (kCK * KW);                                    // stages of KW chunks each (the host checks divisibility)
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)H * W * CinP * 2));
    const frcnn_buf_t wbuf = frcnn_make_buf(wp, (uint32_t)((size_t)TAPS * CoutP * CinP * 2));
    const uint32_t x_chunk_bytes = (uint32_t)(H * W) * 32u, w_chunk_bytes = (uint32_t)(TAPS * CoutP) * 32u;

    // per-lane source offsets (chunk 0) of this wave's pieces: slot s of a region holds (row P = s >> 1, half (s & 1) ^ ((P >> 3) & 1))
    uint32_t poff[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        if (q < IN_Q) {
            const int sl = (gw + GW * q) * 64 + lane, P = sl >> 1, half = (sl & 1) ^ ((P >> 3) & 1);
            const int hr = P / HPX, hx = P - hr * HPX;
            const int gy = y0 - PAD + hr, gx = x0 - PAD + hx;
            const bool inside = P < S::IN_ROWS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            poff[q] = inside ? (uint32_t)((gy * W + gx) * 32 + half * 16) : kBufOob;
        } else {
            const int pw = gw + GW * (q - IN_Q);
            const int sl = pw * 64 + lane, P = sl >> 1, half = (sl & 1) ^ ((P >> 3) & 1);
            const int tap = P / BCO, col = P - tap * BCO;
            poff[q] = (pw < S::W_PIECES && co0 + col < CoutP) ? (uint32_t)((tap * CoutP + co0 + col) * 32 + half * 16) : kBufOob;
        }
    }
    const bool wfull = gw < S::W_FULL;                                    // this wave owns W_Q (not W_Q - 1) weight pieces
    // piece q of stage `sc` -> LDS stage `stage`
    auto issue_piece = [&](int q, int sc, int stage) {
        unsigned char *dst = ring + stage * S::STAGE_BYTES + kw * S::KW_BYTES + gw * 1024;
        const uint32_t chunk = (uint32_t)(sc * KW + kw);
        if (q < IN_Q) frcnn_buf_load_lds_b128(xbuf, dst + q * GW * 1024, poff[q], chunk * x_chunk_bytes);
        else if (q < PPW - 1 || S::W_FULL == GW || wfull)
            frcnn_buf_load_lds_b128(wbuf, dst + q * GW * 1024, poff[q], chunk * w_chunk_bytes);
    };
    auto wait_allow = [&](int k) {                                        // at most k stages of this wave's loads still in flight
        switch (k) {
        case 0: frcnn_wait_vmcnt<0>(); break;
        case 1: if (wfull) frcnn_wait_vmcnt<PPW>(); else frcnn_wait_vmcnt<PPW - 1>(); break;
        case 2: if (wfull) frcnn_wait_vmcnt<(NS > 2 ? 2 : 0) * PPW>(); else frcnn_wait_vmcnt<(NS > 2 ? 2 : 0) * (PPW - 1)>(); break;
        default: if (wfull) frcnn_wait_vmcnt<(NS > 3 ? 3 : 0) * PPW>(); else frcnn_wait_vmcnt<(NS > 3 ? 3 : 0) * (PPW - 1)>(); break;
        }
    };

    // prologue, first half: NS-1 stages go out NOW -- their latency (the longest single wait of a one-round launch: nobody else is
    // resident to cover it) runs under the set-up below (fragment offsets, 160 accumulator writes)
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nsc) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) issue_piece(q, c, c);
        }
    __builtin_amdgcn_sched_barrier(0);

    // fragment byte offsets inside a K-way's chunk image (swizzled): A = weight row tap*BCO + (cw*COB + cb)*32 + l31, B = halo pixel
    // (rg*RW + r)*34 + l31 + kx
    const uint32_t a_off = (uint32_t)(S::IN_BYTES + (cw * COB * 32 + l31) * 32 + ((khalf ^ ((l31 >> 3) & 1)) << 4));
    uint32_t b_off[RW + KS - 1][KS];
#pragma unroll
    for (int r = 0; r < RW + KS - 1; ++r)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int P = (rg * RW + r) * HPX + l31 + kx;
            b_off[r][kx] = (uint32_t)(P * 32 + ((khalf ^ ((P >> 3) & 1)) << 4));
        }

    frcnn_f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;

    // Fragment registers live across stages (the next stage's first two tap groups are read under this stage's last one).  Tap group
    // g = ky*3 + kx uses weight fragments a[g % 3][cb] and halo fragments b[ky + j][kx]; its reads are a[g % 3][.] plus the halo rows
    // no earlier group of the same kx has brought in: rows 0 .. RW-1 for ky 0, row ky + RW - 1 after that.
    uint4 fa[3][COB], fb[RW + KS - 1][KS];
    auto read_group = [&](const unsigned char *st, int g) {
        const int ky = g / KS, kx = g - ky * KS;
#pragma unroll
        for (int cb = 0; cb < COB; ++cb) fa[g % 3][cb] = *reinterpret_cast<const uint4 *>(st + a_off + (g * BCO + cb * 32) * 32);
        if (ky == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r) fb[r][kx] = *reinterpret_cast<const uint4 *>(st + b_off[r][kx]);
        } else fb[ky + RW - 1][kx] = *reinterpret_cast<const uint4 *>(st + b_off[ky + RW - 1][kx]);
    };
    // One stage: nine tap groups of GM MFMAs.  Entered with groups 0 and 1 of the stage read (or in flight).  Before the MFMAs of group
    // g >= 1: the reads of group g + 1.  ISSUE: the PPW pieces of stage `sc_new` go out between the MFMAs of groups 0 .. 7 (piece q after
    // MFMA (q+1) * SPAN / (PPW+1)).  NEXT: before group 8, the hand-over to the next stage -- wait until its data has landed (`allow`
    // later stages of this wave's loads may still be in flight), barrier (everybody's has; and everybody's reads of THIS stage are
    // complete: its LDS stage may be refilled from now on), then the next stage's groups 0 and 1 are read under group 8's MFMAs.
    auto stage_body = [&](auto issue_tag, auto next_tag, int stage, int sc_new, int st_new, int stage_next, int allow) {
        constexpr bool ISSUE = decltype(issue_tag)::value, NEXT = decltype(next_tag)::value;
        const unsigned char *st = ring + stage * S::STAGE_BYTES + kw * S::KW_BYTES;
#pragma unroll
        for (int g = 0; g < TAPS; ++g) {
            if (g >= 1 && g + 1 < TAPS && (ABL & 2) == 0) {
                read_group(st, g + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == TAPS - 1 && NEXT) {
                if constexpr ((ABL & 8) == 0) {
                    wait_allow(allow);
                    frcnn_barrier_nofence();
                }
                if constexpr ((ABL & 2) == 0) {
                    const unsigned char *stn = ring + stage_next * S::STAGE_BYTES + kw * S::KW_BYTES;
                    read_group(stn, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    read_group(stn, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int ky = g / KS, kx = g - ky * KS;
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
#pragma unroll
                for (int j = 0; j < RW; ++j) {
#ifdef FRCNN_TIMING_ABLATIONS
                    if constexpr ((ABL & 4) != 0) { strip_keep(fa[g % 3][cb]); strip_keep(fb[ky + j][kx]); } else
#endif
                    acc[cb * RW + j] = frcnn_mfma_32x32x16_bf16(fa[g % 3][cb], fb[ky + j][kx], acc[cb * RW + j]);
                    if constexpr (ISSUE && (ABL & 1) == 0) {
                        const int m = (g * COB + cb) * RW + j;                             // folds to a constant in the unrolled body
                        const int q0 = m * (PPW + 1) / S::SPAN, q1 = (m + 1) * (PPW + 1) / S::SPAN;
                        if (q1 != q0 && q1 - 1 < PPW) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(q1 - 1, sc_new, st_new);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using Yes = std::true_type;
    using No = std::false_type;

    // prologue, second half (the first NS-1 stages were issued above, before the fragment offsets and the accumulators were set up): stage 0
    // landed, its first two tap groups on their way into registers
    wait_allow(min(NS - 2, nsc - 1));
    frcnn_barrier_nofence();
    read_group(ring + kw * S::KW_BYTES, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_group(ring + kw * S::KW_BYTES, 1);
    __builtin_amdgcn_sched_barrier(0);
    // (three call sites in two loops and a tail, one per loop: with two forms of the stage behind a branch inside ONE loop the compiler
    // copied all 160 accumulators AGPR -> VGPR -> AGPR around the branch, every stage)
    int s_cur = 0, s_new = NS - 1, c = 0;
    for (; c + NS - 1 < nsc; ++c) {                                       // stages that still have a successor to fetch
        const int s_next = s_cur + 1 == NS ? 0 : s_cur + 1;
        stage_body(Yes{}, Yes{}, s_cur, c + NS - 1, s_new, s_next, NS - 2);
        s_cur = s_next;
        s_new = s_new + 1 == NS ? 0 : s_new + 1;
    }
    for (; c + 1 < nsc; ++c) {                                            // nothing left to issue
        const int s_next = s_cur + 1 == NS ? 0 : s_cur + 1;
        stage_body(No{}, Yes{}, s_cur, 0, 0, s_next, nsc - 1 - (c + 1));
        s_cur = s_next;
    }
    stage_body(No{}, No{}, s_cur, 0, 0, 0, 0);                            // the last stage
    __syncthreads();                                                      // the ring becomes the reduction buffer and the output tile
    // K ways -> one: a reduce-scatter through LDS.  The register quad (a, g) -- four consecutive couts of one pixel -- is finished by
    // the wave whose K way is g % KW: every wave deposits the quads it does not own, the owner adds the KW terms in ascending K-way
    // order (its own from registers).  part[gw][owner][source slot][a][g / KW] x 1 KB (64 lanes x float4).
    unsigned char *ot = ring;
    if constexpr (KW > 1) {
        constexpr int GH = 4 / KW;
        unsigned char *part = ring + S::OT_BYTES;
        auto slot = [&](int owner, int src, int a, int gh) {
            const int sidx = src < owner ? src : src - 1;
            return part + ((size_t)((((gw * KW + owner) * (KW - 1) + sidx) * NACC + a) * GH + gh)) * 1024 + lane * 16;
        };
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g % KW != kw)
                    *reinterpret_cast<float4 *>(slot(g % KW, kw, a, g / KW)) = make_float4(acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]);
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g % KW == kw) {
                    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ks = 0; ks < KW; ++ks) {
                        float4 t;
                        if (ks == kw) t = make_float4(acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]);
                        else t = *reinterpret_cast<const float4 *>(slot(kw, ks, a, g / KW));
                        if (ks == 0) s = t;
                        else { s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
                    }
                    acc[a][4 * g] = s.x; acc[a][4 * g + 1] = s.y; acc[a][4 * g + 2] = s.z; acc[a][4 * g + 3] = s.w;
                }
    }

    // epilogue (the forms of conv_bf16_epilogue; here a wave writes the quads it owns): register r of lane l = cout (r&3) + 8*(r>>2) + 4*khalf
    // of pixel l31
    const frcnn_buf_t bbuf = frcnn_make_buf(bias, (uint32_t)Cout * 4u);
    float bv[COB][16];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                bv[cb][4 * g + t] = (g % KW == kw) ? frcnn_buf_load_f32(bbuf, (uint32_t)(co0 + (cw * COB + cb) * 32 + 8 * g + 4 * khalf + t) * 4u) : 0.0f;
    if (out_mode == 1) {
        // fp32 NCHW (Cout, H, W): straight-line buffer stores, lanes / couts outside the map store nothing
        const frcnn_buf_t ybuf = frcnn_make_buf(y, (uint32_t)((size_t)Cout * H * W * 4));
        const int px = x0 + l31;
#pragma unroll
        for (int cb = 0; cb < COB; ++cb)
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const int py = y0 + rg * RW + j;
                const bool inside = px < W && py < H;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g % KW != kw) continue;
                    const int co = co0 + (cw * COB + cb) * 32 + 8 * g + 4 * khalf;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float v = acc[cb * RW + j][4 * g + t] + bv[cb][4 * g + t];
                        if (relu) v = fmaxf(v, 0.0f);
                        frcnn_buf_store_f32(ybuf, (inside && co + t < Cout) ? (uint32_t)(((co + t) * H + py) * W + px) * 4u : kBufOob, v);
                    }
                }
            }
        return;
    }
    if constexpr (DIRECT) {
        if (out_mode == 0) {
            const frcnn_buf_t ybuf = frcnn_make_buf(y, (uint32_t)((size_t)CoutP * H * W * 2));         // (the host checks < 2 GiB)
            const int px = x0 + l31;
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
#pragma unroll
                for (int j = 0; j < RW; ++j) {
                    const int py = y0 + rg * RW + j;
                    const bool inside = px < W && py < H;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g % KW != kw) continue;
                        const int co = co0 + (cw * COB + cb) * 32 + 8 * g + 4 * khalf;                    // first of four consecutive couts: co % 4 == 0
                        float v[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            v[t] = acc[cb * RW + j][4 * g + t] + bv[cb][4 * g + t];
                            if (relu) v[t] = fmaxf(v[t], 0.0f);
                        }
                        uint2 pk;
                        pk.x = frcnn_pack_bf16x2(v[0], v[1]);
                        pk.y = frcnn_pack_bf16x2(v[2], v[3]);
                        frcnn_buf_store_b64(ybuf, (inside && co < CoutP) ? (uint32_t)((((co >> 4) * H + py) * W + px) * 32 + (co & 15) * 2) : kBufOob, pk);
                    }
                }
            return;
        }
    }
    // bf16 channel-blocked output: transpose through LDS so that each 16-cout block of a tile row leaves as one contiguous run
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g % KW != kw) continue;
                const int col = (cw * COB + cb) * 32 + 8 * g + 4 * khalf;
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v[t] = acc[cb * RW + j][4 * g + t] + bv[cb][4 * g + t];
                    if (relu) v[t] = fmaxf(v[t], 0.0f);
                }
                uint2 pk;
                pk.x = frcnn_pack_bf16x2(v[0], v[1]);
                pk.y = frcnn_pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(ot + ((rg * RW + j) * 32 + l31) * OP + col * 2) = pk;
            }
    __syncthreads();
    constexpr int CB16 = BCO / 16;                                        // 16-cout blocks of the tile
    if (out_mode == 2) {
        // F.MaxPooling2D(2, 2) (cover_all) fused: TR is even and tiles start at even rows / columns, so every 2x2 window lies inside
        // the tile; y is [CoutP/16][ceil(H/2)][ceil(W/2)][16]
        const int OH = (H + 1) / 2, OW = (W + 1) / 2;
        for (int v = tid; v < (TR / 2) * 16 * CB16 * 2; v += 64 * NW) {
            const int cbl = v / ((TR / 2) * 16 * 2), rem = v - cbl * ((TR / 2) * 16 * 2);
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
    for (int v = tid; v < TR * 32 * CB16 * 2; v += 64 * NW) {             // 16-byte vectors: (cout block of 16, pixel, half)
        const int cbl = v / (TR * 32 * 2), rem = v - cbl * (TR * 32 * 2);
        const int pix = rem >> 1, half = rem & 1;
        const int py = y0 + (pix >> 5), qx = x0 + (pix & 31), co = co0 + cbl * 16;
        if (py < H && qx < W && co < CoutP)
            *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(y) + (((size_t)(co >> 4) * H + py) * W + qx) * 16 + half * 8) =
                *reinterpret_cast<const uint4 *>(ot + pix * OP + (cbl * 2 + half) * 16);
    }
}
