// conv_bf16_res.h -- the 3x3 bf16 convolution for layers whose weight slab fits in LDS ("resident" forms; included by conv_bf16.hip inside its
// anonymous namespace).  Round 4, after the stamps of csrc/conv_bf16_pair.hip: a wave that does nothing but read fragments from LDS and issue MFMAs
// runs the matrix pipe at ~90 % of its pace (35 clocks per 32x32x16 MFMA), while every strip form of conv_bf16_strip.h pays its LDS-DMA pieces in the
// MFMA wave's own issue stream (SQ_VALU_MFMA_BUSY 49 % on conv3_2, profiles/r04_mfma_pmc_summary.json).  Here the two jobs are separate waves:
//   * a workgroup is 8 waves, two per SIMD: waves 0..3 (consumers) multiply, waves 4..7 (producers) move the halo chunks of the input -- and nothing
//     else is ever moved: the workgroup's whole weight slab (32 CB couts x Cin x 9 taps <= 73.7 KB) is copied into LDS ONCE, the launch is persistent
//     (one workgroup per CU walking its tiles), and the input ring runs on across tile boundaries, so a tile has no prologue;
//   * form R   CB 2, 4 chunks (Cin 64):  64 couts x 8 rows x 32 px per tile, consumers = 2 row groups x 2 cout ways, ring of 6 x 12 KB   (conv1_2, conv2_1)
//     form R2  CB 1, 8 chunks (Cin 128): 32 couts x 16 rows x 32 px,          consumers = 4 row groups,             ring of 4 x 20 KB   (conv2_2, conv3_1)
//   * per K-chunk one fence-less barrier, placed like the strip forms' hand-over: BEFORE the last tap group, so the next chunk's first fragments
//     are in flight under that group's MFMAs; the producers meet it after waiting (counted vmcnt) for the chunk after that to have landed;
//   * a producer's piece is one instruction: per-lane offsets inside a tile are computed once, the tile's origin and the chunk ride in the scalar
//     offset (tiles at the image's edge rebuild their offsets with the range check);
//   * epilogue by the consumers under the producers' next chunks: bf16 output as direct 8-byte quad stores (conv_bf16_strip.h's DIRECT form), or
//     ReLU + 2x2 ceil-mode pool through a wave-private LDS tile as 16-byte stores (csrc/conv_bf16_pair.hip's).
// Same operands, same swizzled LDS images, the same (chunk, tap) accumulation order per output as every other bf16 kernel: bit-identical results.
// MEASURED AND NOT ADOPTED (profiles/r04_conv_res_micro.txt): 8-30 % slower than the default picks on all four layers; the reason is in conv_bf16.hip next
// to conv_bf16_default_res_form.  Kept selectable (FRCNN_BF16_DMA=921 / 922, FRCNN_BF16_RES=1) and under test.
#pragma once
#include <type_traits>

template <int CB, int NCH, int RW, int NS>
struct ResShape {
    static constexpr int RG = CB == 2 ? 2 : 4, CW = CB == 2 ? 2 : 1;    // consumer waves = RG row groups x CW cout ways (32 couts each)
    static constexpr int BCO = 32 * CB, TR = RG * RW, HR = TR + 2, HPX = 34, NPIX = HR * HPX;
    static constexpr int PIECES = ((NPIX * 32 + 1023) / 1024 + 3) / 4 * 4;           // 1 KB pieces of a halo chunk, padded to four per producer round
    static constexpr int PPW = PIECES / 4;                                          // pieces per producer wave and chunk
    static constexpr int STAGE = PIECES * 1024;
    static constexpr int WCH = 9 * BCO * 32;                                        // one K-chunk of the workgroup's weight slab
    static constexpr int WBYTES = NCH * WCH;
    static constexpr int OSTAGE = (RW / 2) * 16 * 64;                               // a consumer wave's pooled output: RW/2 rows x 16 px x 32 couts bf16
    static constexpr int OFF_W = 0, OFF_RING = WBYTES, OFF_OS = OFF_RING + NS * STAGE, LDS = OFF_OS + 4 * OSTAGE;
    static_assert(RW % 2 == 0 && NS >= 3 && NS <= 6 && LDS <= 160 * 1024, "shape");
    static_assert((NS - 1) * PPW <= 63, "vmcnt holds 63");
    static_assert(2 * (RW + 1) <= 15, "two tap groups of fragment reads in flight: lgkmcnt holds 15");
};

template <int CB, int NCH, int RW, int NS>
__global__ void __launch_bounds__(512, 1)
conv_res_bf16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wp, const float *__restrict__ bias, uint16_t *__restrict__ y,
                     int Cout, int CoutP, int H, int W, int relu, int out_mode, int xtiles, int cotiles, int ntiles, int prio) {
    using S = ResShape<CB, NCH, RW, NS>;
    constexpr int HPX = S::HPX, NPIX = S::NPIX, BCO = S::BCO, TR = S::TR, PPW = S::PPW;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[S::LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    // tile t -> (cout tile fastest: the workgroups of neighbouring CUs share an input tile in L2), then x, then y.  This workgroup's tiles:
    // blockIdx.x + k * gridDim.x; its cout tile is the same for all of them when gridDim.x is a multiple of cotiles (the host sees to that).
    const int n_my = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int cot = (int)blockIdx.x % cotiles, co0 = cot * BCO;
    const int pstep = (int)gridDim.x / cotiles;                        // pixel tiles between two tiles of this workgroup
    const int step_y = pstep / xtiles, step_x = pstep - step_y * xtiles;
    const int p0 = (int)blockIdx.x / cotiles;
    // ---- once: the workgroup's weight slab into LDS, swizzled like a DMA piece lands (16-byte slot s of a chunk = row P = s >> 1, half (s & 1) ^ ((P >> 3) & 1))
    {
        const frcnn_buf_t wbuf = frcnn_make_buf(wp, (uint32_t)((size_t)NCH * 9 * CoutP * 32));
        for (int sidx = tid; sidx < S::WBYTES / 16; sidx += 512) {
            const int c = sidx / (S::WCH / 16), sl = sidx - c * (S::WCH / 16);
            const int P = sl >> 1, half = (sl & 1) ^ ((P >> 3) & 1);
            const int tap = P / BCO, col = P - tap * BCO;
            const float4 v = frcnn_buf_load_f32x4(wbuf, co0 + col < CoutP ? (uint32_t)(((c * 9 + tap) * CoutP + co0 + col) * 32 + half * 16) : kBufOob);
            *reinterpret_cast<float4 *>(lds + S::OFF_W + c * S::WCH + sl * 16) = v;
        }
    }
    const int nsteps = n_my * NCH;                                     // K-chunk steps of this workgroup, over all its tiles

    if (wave >= 4) {
        // =============================================================== producers: the halo chunk of step i + NS - 1 while the consumers multiply step i
        const int pw = wave - 4;
        if (prio == 2) __builtin_amdgcn_s_setprio(1);
        const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)NCH * H * W * 32));
        const uint32_t chunk_bytes = (uint32_t)(H * W) * 32u;
        // per-lane offsets of this wave's pieces inside a tile whose halo lies in the image: slot sl of the chunk image = (pixel P = sl >> 1, half)
        uint32_t poff[PPW], prc[PPW];
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int sl = (pw + 4 * j) * 64 + lane, P = sl >> 1, half = (sl & 1) ^ ((P >> 3) & 1);
            const int hr = P / HPX, hx = P - hr * HPX;
            poff[j] = P < NPIX ? (uint32_t)((hr * W + hx) * 32 + half * 16) : kBufOob;
            prc[j] = (uint32_t)hr | ((uint32_t)hx << 8);
        }
        int py = p0 / xtiles, px = p0 - py * xtiles;                   // pixel tile of the step being ISSUED
        int ic = 0;                                                    // its chunk
        auto issue_step = [&](int slot) {
            const int y0 = py * TR, x0 = px * 32;
            unsigned char *dst = lds + S::OFF_RING + slot * S::STAGE + pw * 1024;
            if (y0 >= 1 && y0 + TR + 1 <= H && x0 >= 1 && x0 + 33 <= W) {             // (wave-uniform) the halo inside the image: origin and chunk as the scalar offset
                const uint32_t soff = (uint32_t)(((y0 - 1) * W + (x0 - 1)) * 32) + (uint32_t)ic * chunk_bytes;
#pragma unroll
                for (int j = 0; j < PPW; ++j) frcnn_buf_load_lds_b128(xbuf, dst + j * 4096, poff[j], soff);
            } else {
#pragma unroll
                for (int j = 0; j < PPW; ++j) {
                    const int gy = y0 - 1 + (int)(prc[j] & 255u), gx = x0 - 1 + (int)(prc[j] >> 8);
                    const bool ok = poff[j] != kBufOob && gy >= 0 && gy < H && gx >= 0 && gx < W;
                    frcnn_buf_load_lds_b128(xbuf, dst + j * 4096, ok ? poff[j] + (uint32_t)(((y0 - 1) * W + (x0 - 1)) * 32) : kBufOob, (uint32_t)ic * chunk_bytes);
                }
            }
            if (++ic == NCH) {                                         // the next step belongs to the workgroup's next tile
                ic = 0;
                px += step_x; py += step_y;
                if (px >= xtiles) { px -= xtiles; ++py; }
            }
        };
        auto wait_allow = [&](int k) {                                 // at most k steps of this wave's pieces still in flight
            switch (k < NS - 2 ? k : NS - 2) {
            case 0: frcnn_wait_vmcnt<0>(); break;
            case 1: frcnn_wait_vmcnt<PPW>(); break;
            case 2: frcnn_wait_vmcnt<2 * PPW>(); break;
            case 3: frcnn_wait_vmcnt<(NS > 4 ? 3 : 0) * PPW>(); break;
            case 4: frcnn_wait_vmcnt<(NS > 5 ? 4 : 0) * PPW>(); break;
            default: frcnn_wait_vmcnt<0>(); break;                     // (k < 0: past the last step)
            }
        };
        int issued = 0;
        for (; issued < NS - 1 && issued < nsteps; ++issued) issue_step(issued);
        wait_allow(min(NS - 2, nsteps - 1));                           // step 0 has landed
        frcnn_barrier_nofence();                                       // barrier 0 (also: the weight slab)
#pragma unroll 1
        for (int i = 0; i < nsteps; ++i) {
            // the consumers are in step i (slot i % NS); slot (i - 1) % NS was released by the barrier that ended step i - 1
            if (issued < nsteps) { issue_step(issued % NS); ++issued; }
            wait_allow(min(NS - 2, nsteps - 2 - i));                   // step i + 1 has landed (nothing to wait for past the last step)
            frcnn_barrier_nofence();                                   // barrier i + 1
        }
        return;
    }

    // =================================================================== consumers
    constexpr int RG = S::RG;
    const int rg = wave % RG, cw = wave / RG;
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const frcnn_buf_t ybuf = frcnn_make_buf(y, (uint32_t)((size_t)CoutP * (out_mode == 2 ? OH * OW : H * W) * 2));
    const frcnn_buf_t bbuf = frcnn_make_buf(bias, (uint32_t)Cout * 4u);
    const uint32_t a_off = (uint32_t)(S::OFF_W + (cw * 32 + l31) * 32 + ((khalf ^ ((l31 >> 3) & 1)) << 4));       // + chunk * WCH + tap * BCO * 32
    uint32_t b_off[RW + 2][3];
#pragma unroll
    for (int r = 0; r < RW + 2; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = (rg * RW + r) * HPX + l31 + kx;
            b_off[r][kx] = (uint32_t)(S::OFF_RING + P * 32 + ((khalf ^ ((P >> 3) & 1)) << 4));
        }
    float bv[16];                                                      // this lane's 16 couts: cw * 32 + (r & 3) + 8 (r >> 2) + 4 khalf
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = frcnn_buf_load_f32(bbuf, (uint32_t)(co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * 4u);
    unsigned char *const ost = lds + S::OFF_OS + wave * S::OSTAGE;
    frcnn_f32x16 acc[RW];
    uint4 fa[3], fb[RW + 2][3];
    auto read_group = [&](int slot, int c, int g) {
        const int ky = g / 3, kx = g - ky * 3;
        fa[g % 3] = *reinterpret_cast<const uint4 *>(lds + a_off + c * S::WCH + g * (BCO * 32));
        const unsigned char *st = lds + slot * S::STAGE;
        if (ky == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r) fb[r][kx] = *reinterpret_cast<const uint4 *>(st + b_off[r][kx]);
        } else fb[ky + RW - 1][kx] = *reinterpret_cast<const uint4 *>(st + b_off[ky + RW - 1][kx]);
    };
    // one K-chunk: nine tap groups of RW MFMAs; entered with groups 0 and 1 read (or in flight); the hand-over -- barrier, then the next step's groups 0 and
    // 1 -- sits before group 8 (all of this step's fragment reads are complete by then: its slot may be refilled)
    auto step_body = [&](int slot, int c, int slot_next, int c_next, bool more) {
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            if (g >= 1 && g + 1 < 9) {
                read_group(slot, c, g + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == 8) {
                frcnn_barrier_nofence();
                if (more) {
                    read_group(slot_next, c_next, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    read_group(slot_next, c_next, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int ky = g / 3, kx = g - ky * 3;
#pragma unroll
            for (int j = 0; j < RW; ++j) acc[j] = frcnn_mfma_32x32x16_bf16(fa[g % 3], fb[ky + j][kx], acc[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    frcnn_barrier_nofence();                                           // barrier 0: the weight slab, step 0
    if (nsteps > 0) {
        read_group(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_group(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    int py = p0 / xtiles, px = p0 - py * xtiles, slot = 0;
#pragma unroll 1
    for (int k = 0; k < n_my; ++k) {
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            const int slot_next = slot + 1 == NS ? 0 : slot + 1, c_next = c + 1 == NCH ? 0 : c + 1;
            step_body(slot, c, slot_next, c_next, k * NCH + c + 1 < nsteps);
            slot = slot_next;
        }
        // ---- epilogue (under the producers' next chunks); register r of lane l = cout co0 + cw * 32 + (r & 3) + 8 (r >> 2) + 4 khalf of pixel l31
        const int x0 = px * 32, y0 = py * TR;
        const int gx = x0 + l31;
        if (out_mode == 0) {
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const int gy = y0 + rg * RW + j;
                const bool inside = gx < W && gy < H;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co0 + cw * 32 + 8 * g + 4 * khalf;  // first of four consecutive couts: one aligned 8-byte piece of the channel-blocked record
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        v[t] = acc[j][4 * g + t] + bv[4 * g + t];
                        if (relu) v[t] = frcnn_max_f32(v[t], 0.0f);
                    }
                    frcnn_buf_store_b64(ybuf, (inside && co < CoutP) ? (uint32_t)((((co >> 4) * H + gy) * W + gx) * 32 + (co & 15) * 2) : kBufOob,
                                        make_uint2(frcnn_pack_bf16x2(v[0], v[1]), frcnn_pack_bf16x2(v[2], v[3])));
                }
            }
        } else {
            // ReLU + 2x2 ceil-mode max-pool fused: the maximum of the fp32 sums, then + bias, ReLU, one rounding (monotone: the bits of the maximum of
            // the rounded values).  Even lanes deposit the quads g = 0, 1 of their pooled pixel, odd lanes g = 2, 3; INTERIOR: every window is whole.
            const int odd = l31 & 1;
            const uint32_t oddm = 0u - (uint32_t)odd;
            auto pool_rows = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                const bool own_ok = gx < W, other_ok = (gx ^ 1) < W;
#pragma unroll
                for (int m = 0; m < RW / 2; ++m) {
                    const bool row2 = y0 + rg * RW + 2 * m + 1 < H;
                    uint2 pk[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
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
                            v[t] = vm + bv[4 * g + t];
                            if (relu) v[t] = frcnn_max_f32(v[t], 0.0f);
                        }
                        pk[g] = make_uint2(frcnn_pack_bf16x2(v[0], v[1]), frcnn_pack_bf16x2(v[2], v[3]));
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint2 val = make_uint2((pk[2 + h].x & oddm) | (pk[h].x & ~oddm), (pk[2 + h].y & oddm) | (pk[h].y & ~oddm));
                        *reinterpret_cast<uint2 *>(ost + (m * 16 + (l31 >> 1)) * 64 + (8 * (2 * odd + h) + 4 * khalf) * 2) = val;
                    }
                }
            };
            if (y0 + TR <= H && x0 + 32 <= W) pool_rows(std::true_type{});
            else pool_rows(std::false_type{});
            __builtin_amdgcn_wave_barrier();                           // the wave's own LDS writes, read by other lanes below (DS operations of one wave are in order)
            // RW/2 pooled rows x 16 px x 32 couts leave as 16-byte pieces: q = (16-cout block, row, pixel, half): 32 consecutive lanes = one 512-byte run
#pragma unroll
            for (int j = 0; j < (RW / 2 * 16 * 4 + 63) / 64; ++j) {
                const int q = lane + 64 * j;
                const int half = q & 1, opx = (q >> 1) & 15, m = (q >> 5) % (RW / 2), cb16 = q / (32 * (RW / 2));
                const uint4 val = *reinterpret_cast<const uint4 *>(ost + (m * 16 + opx) * 64 + cb16 * 32 + half * 16);
                const int oy = (y0 + rg * RW + 2 * m) >> 1, ox = (x0 >> 1) + opx, co = co0 + cw * 32 + cb16 * 16;
                const bool ok = q < RW / 2 * 16 * 4 && oy < OH && ox < OW && co < CoutP;
                frcnn_buf_store_b128(ybuf, ok ? (uint32_t)((((co >> 4) * OH + oy) * OW + ox) * 32 + half * 16) : kBufOob, val);
            }
            __builtin_amdgcn_wave_barrier();                           // the staging tile is rewritten by the next tile only after these reads
        }
        px += step_x; py += step_y;
        if (px >= xtiles) { px -= xtiles; ++py; }
    }
}

// launch: 910 + form: 921 = form R (Cin 64, 64 couts per workgroup), 922 = form R2 (Cin 128, 32 couts per workgroup).  Returns 1 when the form does not apply.
// ONE place for "does the resident form apply to this launch" -- the launcher and frcnn_conv_bf16_plan both ask it (the plan query used to
// answer from the channel count alone and could name a kernel the launcher then declined)
static bool conv_bf16_res_applies(int form, int CinP, int CoutP, int H, int W, int out_mode) {
    if (out_mode != 0 && out_mode != 2) return false;
    if ((size_t)CoutP * H * W * 2 >= (1ull << 31) || (size_t)CinP * H * W * 2 >= (1ull << 31)) return false;
    if (!((form == 21 && CinP == 64) || (form == 22 && CinP == 128))) return false;
    const int cus = frcnn_cu_count() > 0 ? frcnn_cu_count() : 256;
    const int bco = form == 21 ? ResShape<2, 4, 4, 6>::BCO : ResShape<1, 8, 4, 4>::BCO, tr = form == 21 ? ResShape<2, 4, 4, 6>::TR : ResShape<1, 8, 4, 4>::TR;
    const int cotiles = frcnn_cdiv(CoutP, bco), ntiles = frcnn_cdiv(W, 32) * frcnn_cdiv(H, tr) * cotiles;
    const int grid = ntiles < cus ? ntiles : (cus / cotiles > 0 ? cus / cotiles * cotiles : cotiles);
    return grid >= 1 && grid % cotiles == 0;
}
static int conv_bf16_res(int form, const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int CinP, int Cout, int CoutP, int H, int W, int relu,
                         int out_mode, hipStream_t stream) {
    if (!conv_bf16_res_applies(form, CinP, CoutP, H, W, out_mode)) return 1;
    const char *pe = frcnn_tune("FRCNN_BF16_RES_PRIO");
    const int prio = pe ? atoi(pe) : 0;
    const int cus = frcnn_cu_count() > 0 ? frcnn_cu_count() : 256;
    const int xtiles = frcnn_cdiv(W, 32);
    if (form == 21 && CinP == 64) {
        using S = ResShape<2, 4, 4, 6>;
        const int cotiles = frcnn_cdiv(CoutP, S::BCO), ntiles = xtiles * frcnn_cdiv(H, S::TR) * cotiles;
        const int grid = ntiles < cus ? ntiles : (cus / cotiles > 0 ? cus / cotiles * cotiles : cotiles);   // a multiple of cotiles: a workgroup keeps its cout tile
        if (grid < 1 || grid % cotiles != 0) return 1;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_res_bf16_kernel<2, 4, 4, 6>), dim3((unsigned)grid), dim3(512), 0, stream, x, w_packed, bias, (uint16_t *)y, Cout, CoutP, H, W,
                           relu, out_mode, xtiles, cotiles, ntiles, prio);
        return 0;
    }
    if (form == 22 && CinP == 128) {
        using S = ResShape<1, 8, 4, 4>;
        const int cotiles = frcnn_cdiv(CoutP, S::BCO), ntiles = xtiles * frcnn_cdiv(H, S::TR) * cotiles;
        const int grid = ntiles < cus ? ntiles : (cus / cotiles > 0 ? cus / cotiles * cotiles : cotiles);   // a multiple of cotiles: a workgroup keeps its cout tile
        if (grid < 1 || grid % cotiles != 0) return 1;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_res_bf16_kernel<1, 8, 4, 4>), dim3((unsigned)grid), dim3(512), 0, stream, x, w_packed, bias, (uint16_t *)y, Cout, CoutP, H, W,
                           relu, out_mode, xtiles, cotiles, ntiles, prio);
        return 0;
    }
    return 1;
}
