// dma_align_micro.hip -- does `buffer_load_dwordx4 ... lds` accept source addresses that are only 4-byte aligned, and how does the range check
// treat a 16-byte access that straddles the end of the buffer?  One wave: lane l loads 16 bytes from src + 4 * (shift + 4 * l) into LDS at
// 16 * l, then the LDS image is copied out.  Prints mismatches per shift.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i make_buf(const void *p, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    v4i d; d.x = (int)(uint32_t)a; d.y = (int)((uint32_t)(a >> 32) & 0xffffu); d.z = (int)bytes; d.w = 0x00020000;
    return d;
}
__global__ void k(const float *src, uint32_t nbytes, int shift, float *out) {
    __shared__ __attribute__((aligned(1024))) float img[256];
    const int lane = threadIdx.x;
    const v4i b = make_buf(src, nbytes);
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)img;
    const uint32_t off = (uint32_t)(shift + 4 * lane) * 4u;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" : : "s"(la), "v"(off), "s"(b) : "memory", "m0");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = img[i];
}
int main() {
    const int N = 1024;
    std::vector<float> h(N); for (int i = 0; i < N; ++i) h[i] = (float)(i + 1);
    float *d, *o; hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4); hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 6; ++shift) {
        for (int tail = 0; tail < 2; ++tail) {
            // tail = 1: the buffer ends 2 floats into lane 63's vector (range check of a partly out-of-range 16-byte access)
            const uint32_t nb = tail ? (uint32_t)(shift + 4 * 63 + 2) * 4u : N * 4u;
            hipMemset(o, 0xff, 256 * 4);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nb, shift, o);
            std::vector<float> r(256); hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
            int bad = 0; for (int i = 0; i < 252; ++i) bad += r[i] != (float)(shift + i + 1);
            printf("shift %d (src %2d-byte aligned) tail %d: mismatches in lanes 0..62: %d; lane 63 = %g %g %g %g (expect %d %d %s)\n", shift, shift % 4 == 0 ? 16 : (shift % 2 == 0 ? 8 : 4), tail, bad,
                   r[252], r[253], r[254], r[255], shift + 253, shift + 254, tail ? "0 0 if the check is per dword" : "..");
        }
    }
    return 0;
}
