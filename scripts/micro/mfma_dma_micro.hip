// mfma_dma_micro.hip -- how much matrix-pipe time does an LDS-DMA instruction cost the SIMD it is issued on?
// One workgroup per CU, 8 waves: waves 0-3 ("M", one per SIMD) run N dependent-free v_mfma_f32_32x32x2_f32 on 9 accumulators and time
// themselves with s_memtime; waves 4-7 ("D") issue LDS-DMA pieces from an L2-resident buffer in a loop until the M waves are done
// (flag in LDS), in one of several forms: none, b32 (64 lanes x 4 B), b32 with 34 lanes, b128 (64 x 16 B), with `gap` s_sleep units
// between pieces (0 = back to back, throttled only by vmcnt <= 8).  Also: M = bf16 32x32x16.  Prints cycles per MFMA and DMA pieces
// issued per MFMA for each form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v4i make_buf(const void *p, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    v4i d; d.x = (int)(uint32_t)a; d.y = (int)((uint32_t)(a >> 32) & 0xffffu); d.z = (int)bytes; d.w = 0x00020000;
    return d;
}
template <int FORM, int MF, int FILL = 0>      // FORM 0 none, 1 b32 x 64 lanes, 2 b32 x 34 lanes, 3 b128; MF 0 fp32 MFMA, 1 bf16 MFMA
__global__ void __launch_bounds__(512, 1) k(const float *src, uint32_t nbytes, int nmfma, int gap, unsigned long long *out, float *sink, int dprio) {
    __shared__ __attribute__((aligned(1024))) float img[4][4096];
    __shared__ volatile int done;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) done = 0;
    __syncthreads();
    if (wave < 4) {
        f32x16 acc[9];
        for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float a = (float)lane, b = 1.0f;
        bf16x8 ah, bh; for (int i = 0; i < 8; ++i) { ah[i] = (short)(lane + i); bh[i] = (short)i; }
        const v4i mbd = make_buf(src, nbytes);
        const uint32_t mla = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)&img[wave][0];
        const uint32_t moff = (uint32_t)lane * 4u;
        uint32_t sa = 0, sb = 0, va = 0, vb = 0;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < nmfma; i += 9) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (MF == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                else if (MF == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                else {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    f32x4 c4; c4[0] = acc[t][0]; c4[1] = acc[t][1]; c4[2] = acc[t][2]; c4[3] = acc[t][3];
                    c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
                    acc[t][0] = c4[0]; acc[t][1] = c4[1]; acc[t][2] = c4[2]; acc[t][3] = c4[3];
                }
                // fillers in the wave's own stream: 1 = one s_nop per MFMA, 2 = four, 3 = one v_mov, 4 = one ds_read_b32 (waited for at the end of the group)
                if constexpr (FILL == 5) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(mla + (uint32_t)t * 256u), "v"(moff), "s"(mbd) : "memory", "m0");
                if constexpr (FILL == 6) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(mla + (uint32_t)t * 1024u), "v"(moff * 4u), "s"(mbd) : "memory", "m0");
                if constexpr (FILL == 7) { asm volatile("s_mov_b32 %0, 1\n\ts_mov_b32 %1, 2\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, %0" : "=s"(sa), "=s"(sb)); }
                if constexpr (FILL == 8) asm volatile("v_mov_b32 %0, 1\n\tv_mov_b32 %1, 2\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %0" : "=v"(va), "=v"(vb));
                if constexpr (FILL == 9) asm volatile("v_mov_b32 %0, 1" : "=v"(va));
                if constexpr (FILL == 1) asm volatile("s_nop 0");
                if constexpr (FILL == 2) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
                if constexpr (FILL == 3) asm volatile("v_mov_b32 %0, %0" : "+v"(b));
                if constexpr (FILL == 4) { float tmp; asm volatile("ds_read_b32 %0, %1" : "=v"(tmp) : "v"(lane * 4)); asm volatile("" :: "v"(tmp)); }
            }
            if constexpr (FILL == 4) asm volatile("s_waitcnt lgkmcnt(0)");
            if constexpr (FILL == 5 || FILL == 6) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0; for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        if (s == 12345.f) sink[tid] = s + (float)(sa + sb + va + vb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; }
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) atomicAdd((int *)&done, 1);
    } else {
        const v4i bd = make_buf(src, nbytes);
        unsigned long long pieces = 0;
        if (dprio == 3) __builtin_amdgcn_s_setprio(3);
        if (FORM == 4) { while (done < 4) { ++pieces; } }
        if (FORM != 0 && FORM != 4) {
            const uint32_t la0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)&img[wave - 4][0];
            uint32_t off = (uint32_t)(lane * (FORM == 3 ? 16 : 4));
            int slot = 0;
            while (done < 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t la = la0 + (uint32_t)slot * 1024u;
                    if (FORM == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(la), "v"(off), "s"(bd) : "memory", "m0");
                    if (FORM == 2) { if (lane < 34) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(la), "v"(off), "s"(bd) : "memory", "m0"); }
                    if (FORM == 3) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(la), "v"(off), "s"(bd) : "memory", "m0");
                    slot = (slot + 1) & 15;
                    off = (off + 1024u) & (nbytes - 1u);
                    ++pieces;
                    if (gap > 0) __builtin_amdgcn_s_sleep(1);
                    if (gap > 1) __builtin_amdgcn_s_sleep(4);
                    if (gap > 2) __builtin_amdgcn_s_sleep(16);
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) out[(blockIdx.x * 8 + wave) * 2 + 1] = pieces;
    }
}
template <int FORM, int MF, int FILL = 0> static void run(const char *name, const float *d, uint32_t nb, int nmfma, int gap, unsigned long long *o, float *sink, int blocks, int dprio = 0) {
    const int fill = FILL;
    hipMemset(o, 0, blocks * 16 * 8);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k<FORM, MF, FILL>), dim3(blocks), dim3(512), 0, 0, d, nb, nmfma, gap, o, sink, dprio);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k<FORM, MF, FILL>), dim3(blocks), dim3(512), 0, 0, d, nb, nmfma, gap, o, sink, dprio);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16); hipMemcpy(h.data(), o, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, pcs = 0; int nm = 0, nd = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) { if (w < 4) { cyc += (double)h[(b * 8 + w) * 2]; ++nm; } else { pcs += (double)h[(b * 8 + w) * 2 + 1]; ++nd; } }
    // (s_memtime: shader clocks)
    if (getenv("MICRO_RAW")) { for (int w = 0; w < 8; ++w) printf("  wave %d: %llu %llu\n", w, h[w * 2], h[w * 2 + 1]); }
    printf("%-64s fill %d dprio %d gap %d: clocks per MFMA %8.4f   DMA pieces per MFMA (same SIMD) %6.3f\n", name, fill, dprio, gap, cyc / nm / nmfma, pcs / nd / nmfma);
}
int main() {
    const uint32_t nb = 1u << 20; float *d, *sink; unsigned long long *o; const int blocks = 256;
    hipMalloc(&d, nb); hipMemset(d, 0, nb); hipMalloc(&sink, 4096); hipMalloc(&o, blocks * 16 * 8);
    const int nm = 9 * 4000;
    printf("== a second wave on the same SIMD (polling an LDS flag, or issuing LDS-DMA pieces) next to a wave of back-to-back MFMAs:\n");
    printf("   `DMA pieces per MFMA` = iterations / pieces the second wave got through per MFMA of the first\n");
    run<0, 0, 0>("f32 32x32x2: alone", d, nb, nm, 0, o, sink, blocks);
    run<4, 0, 0>("f32 32x32x2: + polling wave", d, nb, nm, 0, o, sink, blocks);
    run<4, 0, 0>("f32 32x32x2: + polling wave at s_setprio 3", d, nb, nm, 0, o, sink, blocks, 3);
    run<1, 0, 0>("f32 32x32x2: + DMA wave", d, nb, nm, 0, o, sink, blocks);
    run<0, 2, 0>("f32 16x16x4: alone", d, nb, nm, 0, o, sink, blocks);
    run<4, 2, 0>("f32 16x16x4: + polling wave", d, nb, nm, 0, o, sink, blocks);
    run<4, 2, 0>("f32 16x16x4: + polling wave at s_setprio 3", d, nb, nm, 0, o, sink, blocks, 3);
    run<1, 2, 0>("f32 16x16x4: + DMA wave", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 0>("bf16 32x32x16: alone", d, nb, nm, 0, o, sink, blocks);
    run<4, 1, 0>("bf16 32x32x16: + polling wave", d, nb, nm, 0, o, sink, blocks);
    run<4, 1, 0>("bf16 32x32x16: + polling wave at s_setprio 3", d, nb, nm, 0, o, sink, blocks, 3);
    run<3, 1, 0>("bf16 32x32x16: + DMA wave", d, nb, nm, 0, o, sink, blocks);
    printf("== price of instructions in the MFMA wave's own stream (one wave per SIMD; clocks per MFMA):\n");
    run<0, 0, 0>("f32 32x32x2 + per MFMA: nothing", d, nb, nm, 0, o, sink, blocks);
    run<0, 0, 1>("f32 32x32x2 + per MFMA: 1 s_nop", d, nb, nm, 0, o, sink, blocks);
    run<0, 0, 2>("f32 32x32x2 + per MFMA: 4 s_nop", d, nb, nm, 0, o, sink, blocks);
    run<0, 0, 9>("f32 32x32x2 + per MFMA: 1 independent VALU", d, nb, nm, 0, o, sink, blocks);
    run<0, 0, 8>("f32 32x32x2 + per MFMA: 4 VALU (2 mov + 2 dependent add)", d, nb, nm, 0, o, sink, blocks);
    run<0, 0, 4>("f32 32x32x2 + per MFMA: 1 ds_read_b32", d, nb, nm, 0, o, sink, blocks);
    run<0, 0, 5>("f32 32x32x2 + per MFMA: 1 b32 LDS-DMA (+s_mov m0, s_nop)", d, nb, nm, 0, o, sink, blocks);
    run<0, 0, 6>("f32 32x32x2 + per MFMA: 1 b128 LDS-DMA", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 0>("bf16 32x32x16 + per MFMA: nothing", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 1>("bf16 32x32x16 + per MFMA: 1 s_nop", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 2>("bf16 32x32x16 + per MFMA: 4 s_nop", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 9>("bf16 32x32x16 + per MFMA: 1 independent VALU", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 8>("bf16 32x32x16 + per MFMA: 4 VALU (2 mov + 2 dependent add)", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 4>("bf16 32x32x16 + per MFMA: 1 ds_read_b32", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 5>("bf16 32x32x16 + per MFMA: 1 b32 LDS-DMA (+s_mov m0, s_nop)", d, nb, nm, 0, o, sink, blocks);
    run<0, 1, 6>("bf16 32x32x16 + per MFMA: 1 b128 LDS-DMA", d, nb, nm, 0, o, sink, blocks);
    return 0;
}
