// mfma_peak_micro.hip -- what the matrix pipes SUSTAIN in wall-clock terms: every SIMD of the chip issues nothing but independent MFMAs
// (ten accumulators per wave, operands in registers, no memory traffic at all), timed with events over bursts of launches.  The other
// micro-benchmarks count shader clocks per MFMA (32.9 for bf16 32x32x16: the pipe is full); this one says what those clocks are worth
// under a chip-wide MFMA load -- the ceiling every "fraction of the 2.5 PFLOP/s peak" in DESIGN.md has to be read against.
// Operand DATA matters: the matrix datapath of a chip-wide MFMA load is power-limited, and constant operands toggle no bits.
// Usage: mfma_peak_micro [waves_per_simd=1] [mfmas_per_wave=20000] [launches_per_timed_burst=10]   (a burst of 10 lasts ~6 ms at bf16 rates; 2000 lasts over a second:
//        does the clock sag further when the load lasts?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
// round 5: the same stream with v_mfma_f32_16x16x32_bf16 (KIND 2: half the accumulator rows per instruction, twice the k -- the same flops per instruction
// and per operand byte; does the datapath draw the same power for them?), and with HALF of the B operand's values zero (zero_b: what a post-ReLU activation
// map looks like to the multipliers)
template <int KIND>
__global__ void __launch_bounds__(256) mfma_stream16(float *out, int n, const uint4 *rnd, int zero_b) {
    f32x4 acc[10];
    for (int a = 0; a < 10; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    bf16x8 va[5], vb[5];
    for (int k = 0; k < 5; ++k) {
        const uint4 ra = rnd[(k * 2) * 256 + threadIdx.x];
        uint4 rb = rnd[(k * 2 + 1) * 256 + threadIdx.x];
        if (zero_b) { rb.x &= 0xffff0000u; rb.y &= 0x0000ffffu; rb.z &= 0xffff0000u; rb.w &= 0x0000ffffu; }
        va[k] = __builtin_bit_cast(bf16x8, ra); vb[k] = __builtin_bit_cast(bf16x8, rb);
    }
    for (int i = 0; i < n; i += 10) {
#pragma unroll
        for (int a = 0; a < 10; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[a % 5], vb[a / 2], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 10; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>   // 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_32x32x2_f32
__global__ void __launch_bounds__(256) mfma_stream(float *out, int n, const uint4 *rnd, int zero_b = 0) {
    f32x16 acc[10];
    for (int a = 0; a < 10; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // operands: small constants (rnd == nullptr: almost no bits toggle between MFMAs) or random bf16 / fp32 values in [-1, 1), a different
    // fragment per lane and five A and five B fragments in rotation (what a convolution's operand stream looks like to the datapath)
    bf16x8 va[5], vb[5];
    float fa[5], fb[5];
    for (int k = 0; k < 5; ++k) {
        for (int i = 0; i < 8; ++i) { va[k][i] = (__bf16)(float)(threadIdx.x & 3); vb[k][i] = (__bf16)1.0f; }
        fa[k] = (float)(threadIdx.x & 3); fb[k] = 1.0f;
        if (rnd) {
            const uint4 ra = rnd[(k * 2) * 256 + threadIdx.x], rb = rnd[(k * 2 + 1) * 256 + threadIdx.x];
            uint4 rbz = rb;
            if (zero_b) { rbz.x &= 0xffff0000u; rbz.y &= 0x0000ffffu; rbz.z &= 0xffff0000u; rbz.w &= 0x0000ffffu; }
            va[k] = __builtin_bit_cast(bf16x8, ra); vb[k] = __builtin_bit_cast(bf16x8, rbz);
            fa[k] = __uint_as_float((ra.x & 0x007fffffu) | 0x3f000000u) - 0.75f; fb[k] = __uint_as_float((rb.x & 0x007fffffu) | 0x3f000000u) - 0.75f;
        }
    }
    for (int i = 0; i < n; i += 10) {
#pragma unroll
        for (int a = 0; a < 10; ++a) {
            if (KIND == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[a % 5], vb[a / 2], acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a % 5], fb[a / 2], acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 10; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

static int g_burst = 10;
template <int KIND>
static void run(const char *name, double flop_per_mfma, int wps, int n, int cus, bool random_data, int zero_b = 0) {
    float *out; CK(hipMalloc(&out, 4096));
    uint4 *rnd = nullptr;
    if (random_data) {                                                    // bf16 pairs with exponents around 2^-2 .. 2^0, random signs and mantissas
        std::vector<uint32_t> h(10 * 256 * 4);
        uint32_t st = 12345u;
        for (auto &w : h) {
            uint32_t v = 0;
            for (int half = 0; half < 2; ++half) { st = st * 1664525u + 1013904223u; const uint32_t r = st >> 8; v |= (((r & 0x80ffu) | ((125u + (r >> 16) % 3u) << 7)) & 0xffffu) << (16 * half); }
            w = v;
        }
        CK(hipMalloc(&rnd, h.size() * 4)); CK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(cus * wps), block(256);
    auto launch = [&]() {
        if (KIND == 2) hipLaunchKernelGGL(mfma_stream16<2>, grid, block, 0, s, out, n, rnd, zero_b);
        else hipLaunchKernelGGL(mfma_stream<(KIND == 2 ? 0 : KIND)>, grid, block, 0, s, out, n, rnd, zero_b);
    };
    for (int i = 0; i < 20; ++i) launch();      // warm-up: let the clocks settle under load
    CK(hipStreamSynchronize(s));
    std::vector<double> tf;
    for (int r = 0; r < 9; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < g_burst; ++i) launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        tf.push_back((double)g_burst * grid.x * 4 * (double)n * flop_per_mfma / (ms * 1e-3) / 1e12);
    }
    std::sort(tf.begin(), tf.end());
    const double med = tf[tf.size() / 2];
    // clocks per MFMA are known (32.9 / 64.0, profiles/r03_mfma_filler_micro.txt): the implied shader clock under this load
    const double clk_per = KIND == 0 ? 32.9 : (KIND == 2 ? 16.45 : 64.04), mfma_per_s_per_simd = med * 1e12 / flop_per_mfma / (cus * 4.0);
    printf("%-40s %d wave(s)/SIMD, %d MFMAs/wave: median %8.1f TFLOP/s (min %.1f max %.1f)  -> implied shader clock %.2f GHz\n", name, wps, n, med, tf.front(), tf.back(),
           mfma_per_s_per_simd * clk_per / wps / 1e9 * wps);
    CK(hipFree(out)); if (rnd) CK(hipFree(rnd));
}

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1, n = argc > 2 ? atoi(argv[2]) : 20000;
    if (argc > 3 && atoi(argv[3]) > 0) g_burst = atoi(argv[3]);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("bf16 32x32x16, constant operands", 32768.0, wps, n, p.multiProcessorCount, false);
    run<0>("bf16 32x32x16, random operands", 32768.0, wps, n, p.multiProcessorCount, true);
    run<1>("f32 32x32x2, constant operands", 4096.0, wps, n, p.multiProcessorCount, false);
    run<1>("f32 32x32x2, random operands", 4096.0, wps, n, p.multiProcessorCount, true);
    run<0>("bf16 32x32x16, random operands again", 32768.0, wps, n, p.multiProcessorCount, true);
    run<0>("bf16 32x32x16, random, half of B zero", 32768.0, wps, n, p.multiProcessorCount, true, 1);
    run<2>("bf16 16x16x32, random operands", 16384.0, wps, 2 * n, p.multiProcessorCount, true);
    run<2>("bf16 16x16x32, random, half of B zero", 16384.0, wps, 2 * n, p.multiProcessorCount, true, 1);
    run<0>("bf16 32x32x16, random operands once more", 32768.0, wps, n, p.multiProcessorCount, true);
    return 0;
}
