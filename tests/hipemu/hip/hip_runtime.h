// tests/hipemu/hip/hip_runtime.h -- a tiny host emulator of the HIP device model (TEST INFRASTRUCTURE).
//
// The build container has no GPU and GPU-minutes are rationed, so the *unmodified* kernel sources in
// chainer-faster-rcnn_amd/csrc/ are also compiled for the host with this header shadowing the real
// <hip/hip_runtime.h> (clang++ -I tests/hipemu ...).  That produces tests/hipemu/_build/libfrcnn_emu.so,
// which exports the same C ABI and lets the CPU test-suite exercise kernel *logic* (indexing, wave
// collectives, MFMA fragment layouts, barriers) against the oracle.  It is never loaded by the product
// package, is not a fallback, and says nothing about performance.
//
// Model: each workgroup runs as cooperative fibers (a hand-rolled x86-64 stack switch; ucontext elsewhere) on one OS thread, blocks run one after the
// other.  __syncthreads() and the wave-level collectives (ballot / shuffle / readlane / MFMA) are
// rendezvous points; a collective that not all live lanes of a wave reach from the same call site aborts
// with a diagnostic (catches divergent-collective bugs that real hardware would turn into garbage).
// LDS-DMA pieces land either when they are issued (default) or at the s_waitcnt that covers them (HIPEMU_DMA_DEFER=1): the two ends of what
// the hardware may do, so that a DMA ring's counted waits are tested and not only reasoned about (see dma_deposit / dma_wait below).
#pragma once
#include <ucontext.h>
#include <sys/mman.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }

typedef struct ihipStream_t *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 0; return hipSuccess; }  // the emulator is not a device
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
// a deliberately odd, tiny "chip" so persistent / stream-K decompositions split tiles unevenly under test
// (HIPEMU_CUS=n gives the emulated chip another size -- read when a library first asks, i.e. once per process: tests that want the launch
// heuristics of a bigger chip run in a subprocess)
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) {
    const char *e = getenv("HIPEMU_CUS");
    *v = (e && atoi(e) > 0) ? atoi(e) : 3;
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }

namespace hipemu {

enum State { RUNNABLE = 0, WAIT_BAR = 1, WAIT_WAVE = 2, DONE = 3 };

// One LDS-DMA deposit of one lane that has been issued but has not "landed" yet (HIPEMU_DMA_DEFER=1, see dma_deposit below)
struct PendingDma { void *dst; unsigned char data[16]; int bytes; };

// Context switch.  swapcontext() saves and restores the signal mask -- two system calls per switch, a quarter of the emulator's run time --
// which cooperative fibers on one thread have no use for: on x86-64 the switch is the callee-saved registers and the stack pointer.
#if defined(__x86_64__)
struct Ctx { void *sp; };
__attribute__((naked, noinline)) static void ctx_switch(Ctx * /*from: rdi*/, Ctx * /*to: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq (%rsi), %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
        "ret");
}
// a fresh context that enters `entry` (which never returns) on the given stack at its first switch-in
inline void ctx_make(Ctx &c, char *stack, size_t bytes, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;                    // the slot a caller's return address would occupy: keeps entry's frame 16-byte aligned as the ABI expects
    *--sp = (void *)entry;              // popped by ctx_switch's ret
    for (int i = 0; i < 6; ++i) *--sp = nullptr;      // rbp, rbx, r12 .. r15
    c.sp = (void *)sp;
}
#else
struct Ctx { ucontext_t uc; };
inline void ctx_switch(Ctx *from, Ctx *to) { swapcontext(&from->uc, &to->uc); }
inline void ctx_make(Ctx &c, char *stack, size_t bytes, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack;
    c.uc.uc_stack.ss_size = bytes;
    c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
#endif

struct Fiber {
    Ctx ctx;
    uint3_emu tid;
    int lin, wave, lane, state;
    char *stack;
    std::vector<PendingDma> dma;        // in issue order; [dma_head, size) are still in flight
    size_t dma_head = 0;
};

struct Wave {
    int gen = 0, arrived = 0, alive = 0;
    unsigned long long present[2] = {0, 0};
    const void *site[2] = {nullptr, nullptr};
    alignas(16) char buf[2][64][32];
};

struct Global {
    Ctx sched;
    Fiber *cur = nullptr;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    uint3_emu bid{0, 0, 0};
    dim3 bdim, gdim;
    int alive = 0, in_bar = 0;
    void (*body)(void *) = nullptr;
    void *body_arg = nullptr;
    size_t stack_bytes = 512 * 1024;
    bool dma_defer = false;             // HIPEMU_DMA_DEFER=1, read at every launch
};
inline Global &G() { static Global g; return g; }

// LDS-DMA timing model.  The hardware lands a buffer_load ... lds some time between its issue and the s_waitcnt vmcnt(N) that covers it.  By
// default the emulator lands it AT ISSUE (the earliest legal moment: what catches a stage refilled while somebody still reads it); with
// HIPEMU_DMA_DEFER=1 it lands AT THE WAIT THAT COVERS IT (the latest legal moment: a fragment read before its stage's counted wait sees the
// stage's previous contents, a wait that allows one stage too many in flight leaves a stage un-landed).  A ring protocol has to pass both.
// vmcnt is per wave and counts instructions; every lane of a wave executes every DMA instruction once, so a per-lane queue has the same length.
// (Ordinary global loads / stores, which the hardware also counts, land at once here: that only ever makes a real wait cover MORE pieces.)
inline void dma_deposit(void *dst, const void *src, int bytes) {
    Global &g = G();
    if (!g.dma_defer) { memcpy(dst, src, (size_t)bytes); return; }
    PendingDma p;
    p.dst = dst; p.bytes = bytes;
    memcpy(p.data, src, (size_t)bytes);
    g.cur->dma.push_back(p);
}
inline void dma_wait(int allow) {       // s_waitcnt vmcnt(allow) of the calling lane's wave
    Fiber *f = G().cur;
    if (!f) return;
    while ((long)(f->dma.size() - f->dma_head) > (long)allow) {
        const PendingDma &p = f->dma[f->dma_head++];
        memcpy(p.dst, p.data, (size_t)p.bytes);
    }
    if (f->dma_head == f->dma.size()) { f->dma.clear(); f->dma_head = 0; }
}

inline void yield_to_sched() { Global &g = G(); ctx_switch(&g.cur->ctx, &g.sched); }

inline void wave_release(Wave &w, int wave_id) {
    Global &g = G();
    w.arrived = 0;
    w.gen++;
    w.present[w.gen & 1] = 0;
    w.site[w.gen & 1] = nullptr;
    for (auto &f : g.fibers)
        if (f.wave == wave_id && f.state == WAIT_WAVE) f.state = RUNNABLE;
}

inline void trampoline() {
    Global &g = G();
    g.body(g.body_arg);
    dma_wait(0);                        // (a wave does not end with loads in flight)
    Fiber *f = g.cur;
    f->state = DONE;
    g.alive--;
    Wave &w = g.waves[f->wave];
    w.alive--;
    if (w.alive > 0 && w.arrived == w.alive) wave_release(w, f->wave);
    ctx_switch(&f->ctx, &g.sched);
}

// Rendezvous of all live lanes of the calling wave; returns the 64 deposits (valid until the
// wave's next-but-one collective) and the mask of lanes that took part.
struct Exchange { const char (*vals)[32]; unsigned long long present; };
__attribute__((noinline)) inline Exchange wave_exchange(const void *v, size_t sz, const void *site) {
    Global &g = G();
    Fiber *f = g.cur;
    Wave &w = g.waves[f->wave];
    int b = w.gen & 1;
    if (sz > 32) { fprintf(stderr, "hipemu: exchange too wide\n"); abort(); }
    if (w.site[b] && w.site[b] != site) {
        fprintf(stderr, "hipemu: DIVERGENT wave collective in block (%u,%u,%u) wave %d lane %d\n", g.bid.x, g.bid.y, g.bid.z, f->wave, f->lane);
        abort();
    }
    w.site[b] = site;
    memcpy(w.buf[b][f->lane], v, sz);
    w.present[b] |= 1ull << f->lane;
    w.arrived++;
    unsigned long long *pres = &w.present[b];
    if (w.arrived == w.alive) {
        unsigned long long p = *pres;
        wave_release(w, f->wave);
        w.present[b] = p;  // keep for the stragglers' read (cleared slot is the *next* generation's)
        return Exchange{w.buf[b], p};
    }
    f->state = WAIT_WAVE;
    yield_to_sched();
    return Exchange{w.buf[b], *pres};
}

inline void block_barrier() {
    Global &g = G();
    g.cur->state = WAIT_BAR;
    g.in_bar++;
    yield_to_sched();
}

inline void run_block() {
    Global &g = G();
    unsigned n = g.bdim.x * g.bdim.y * g.bdim.z;
    if (g.fibers.size() < n) {
        size_t old = g.fibers.size();
        g.fibers.resize(n);
        for (size_t i = old; i < n; ++i) {
            g.fibers[i].stack = (char *)mmap(nullptr, g.stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (g.fibers[i].stack == (char *)MAP_FAILED) { perror("hipemu mmap"); abort(); }
        }
    }
    unsigned nw = (n + 63) / 64;
    g.waves.assign(nw, Wave());
    for (unsigned i = 0; i < n; ++i) {
        Fiber &f = g.fibers[i];
        f.lin = (int)i;
        f.tid.x = i % g.bdim.x;
        f.tid.y = (i / g.bdim.x) % g.bdim.y;
        f.tid.z = i / (g.bdim.x * g.bdim.y);
        f.wave = (int)(i / 64);
        f.lane = (int)(i % 64);
        f.state = RUNNABLE;
        f.dma.clear();
        f.dma_head = 0;
        g.waves[f.wave].alive++;
        ctx_make(f.ctx, f.stack, g.stack_bytes, trampoline);
    }
    g.alive = (int)n;
    g.in_bar = 0;
    while (g.alive > 0) {
        bool progressed = false;
        for (unsigned i = 0; i < n; ++i) {
            Fiber &f = g.fibers[i];
            if (f.state != RUNNABLE) continue;
            g.cur = &f;
            ctx_switch(&g.sched, &f.ctx);
            progressed = true;
        }
        if (g.alive > 0 && g.in_bar == g.alive) {
            for (unsigned i = 0; i < n; ++i)
                if (g.fibers[i].state == WAIT_BAR) g.fibers[i].state = RUNNABLE;
            g.in_bar = 0;
            progressed = true;
        }
        if (!progressed) {
            fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): %d alive, %d at barrier\n", g.bid.x, g.bid.y, g.bid.z, g.alive, g.in_bar);
            for (unsigned i = 0; i < n && i < 8; ++i) fprintf(stderr, "  thread %u state %d\n", i, g.fibers[i].state);
            abort();
        }
    }
    g.cur = nullptr;
}

template <typename F>
inline void launch(dim3 grid, dim3 block, F &&f) {
    Global &g = G();
    if ((size_t)block.x * block.y * block.z > 1024 || block.x * block.y * block.z == 0) { fprintf(stderr, "hipemu: bad block size\n"); abort(); }
    g.gdim = grid;
    g.bdim = block;
    { const char *e = getenv("HIPEMU_DMA_DEFER"); g.dma_defer = e && e[0] == '1'; }
    using Fn = typename std::remove_reference<F>::type;
    Fn *fp = &f;
    g.body = [](void *p) { (*(Fn *)p)(); };
    g.body_arg = (void *)fp;
    // Workgroups run one after the other; HIPEMU_BLOCK_ORDER=reverse runs them last to first, so that "the last workgroup to arrive" of a
    // ticketed fix-up (stream-K / split-K pieces) is a different one: the results must not depend on it.
    const char *oe = getenv("HIPEMU_BLOCK_ORDER");
    const bool rev = oe && oe[0] == 'r';
    const unsigned long total = (unsigned long)grid.x * grid.y * grid.z;
    for (unsigned long k = 0; k < total; ++k) {
        const unsigned long i = rev ? total - 1 - k : k;
        g.bid = uint3_emu{(unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((unsigned long)grid.x * grid.y))};
        run_block();
    }
}

}  // namespace hipemu

#define threadIdx (hipemu::G().cur->tid)
#define blockIdx (hipemu::G().bid)
#define blockDim (hipemu::G().bdim)
#define gridDim (hipemu::G().gdim)
static const int warpSize = 64;

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    if (shmem != 0) { fprintf(stderr, "hipemu: dynamic LDS is not emulated; use static __shared__\n"); abort(); }
    hipemu::launch(grid, block, [&]() { kernel(static_cast<KArgs>(args)...); });
}

static inline void __syncthreads() { hipemu::dma_wait(0); hipemu::block_barrier(); }       // (its fences compile to s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier)
static inline void __builtin_amdgcn_s_barrier() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---------------------------------------------------------------- wave collectives
#define HIPEMU_SITE() __builtin_extract_return_addr(__builtin_return_address(0))
__attribute__((noinline)) static unsigned long long __ballot(int pred) {
    auto e = hipemu::wave_exchange(&pred, sizeof(pred), HIPEMU_SITE());
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if ((e.present >> l) & 1) { int p; memcpy(&p, e.vals[l], sizeof(p)); if (p) m |= 1ull << l; }
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { return __ballot(!pred) == 0; }
template <typename T>
__attribute__((noinline)) static T __shfl(T v, int src, int width = 64) {
    auto e = hipemu::wave_exchange(&v, sizeof(T), HIPEMU_SITE());
    int lane = hipemu::G().cur->lane;
    int base = lane & ~(width - 1);
    int s = base + (src & (width - 1));
    T r;
    memcpy(&r, e.vals[s], sizeof(T));
    return r;
}
template <typename T>
__attribute__((noinline)) static T __shfl_xor(T v, int mask, int width = 64) {
    auto e = hipemu::wave_exchange(&v, sizeof(T), HIPEMU_SITE());
    int lane = hipemu::G().cur->lane;
    int s = lane ^ mask;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    T r;
    memcpy(&r, e.vals[s], sizeof(T));
    return r;
}
template <typename T>
__attribute__((noinline)) static T __shfl_down(T v, unsigned delta, int width = 64) {
    auto e = hipemu::wave_exchange(&v, sizeof(T), HIPEMU_SITE());
    int lane = hipemu::G().cur->lane;
    int s = lane + (int)delta;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    T r;
    memcpy(&r, e.vals[s], sizeof(T));
    return r;
}
template <typename T>
__attribute__((noinline)) static T __shfl_up(T v, unsigned delta, int width = 64) {
    auto e = hipemu::wave_exchange(&v, sizeof(T), HIPEMU_SITE());
    int lane = hipemu::G().cur->lane;
    int s = lane - (int)delta;
    if (s < 0 || (s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    T r;
    memcpy(&r, e.vals[s], sizeof(T));
    return r;
}
__attribute__((noinline)) static int __builtin_amdgcn_readfirstlane(int v) {
    auto e = hipemu::wave_exchange(&v, sizeof(v), HIPEMU_SITE());
    int first = __builtin_ctzll(e.present);
    int r;
    memcpy(&r, e.vals[first], sizeof(r));
    return r;
}
__attribute__((noinline)) static int __builtin_amdgcn_readlane(int v, int lane) {
    auto e = hipemu::wave_exchange(&v, sizeof(v), HIPEMU_SITE());
    int r;
    memcpy(&r, e.vals[lane], sizeof(r));
    return r;
}

// a wave-level rendezvous: on hardware this is only a scheduling barrier (DS operations of one wave are in order);
// here lanes run one after the other, so cross-lane LDS traffic inside a wave needs a real meeting point
__attribute__((noinline)) static void __builtin_amdgcn_wave_barrier() {
    int z = 0;
    hipemu::wave_exchange(&z, sizeof(z), HIPEMU_SITE());
}

// v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; register r of lane l
// holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]; a k-ordered fmaf chain (cdna_hip_programming.md section 3).
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
__attribute__((noinline)) static hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    float ab[2] = {a, b};
    auto e = hipemu::wave_exchange(ab, sizeof(ab), HIPEMU_SITE());
    if (e.present != ~0ull) { fprintf(stderr, "hipemu: MFMA with a partial wave\n"); abort(); }
    int l = hipemu::G().cur->lane;
    int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, e.vals[i + 32 * k], 4);
            memcpy(&bv, e.vals[j + 32 * k] + 4, 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; reg r of lane l holds D[(l>>4)*4 + r][l&15].
__attribute__((noinline)) static hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    float ab[2] = {a, b};
    auto e = hipemu::wave_exchange(ab, sizeof(ab), HIPEMU_SITE());
    if (e.present != ~0ull) { fprintf(stderr, "hipemu: MFMA with a partial wave\n"); abort(); }
    int l = hipemu::G().cur->lane;
    int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        int i = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, e.vals[i + 16 * k], 4);
            memcpy(&bv, e.vals[j + 16 * k] + 4, 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// ---------------------------------------------------------------- scalar builtins / atomics (one OS thread)
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0; }   // HW_ID etc.: one wave slot on the host
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
