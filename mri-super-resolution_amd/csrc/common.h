// Shared host/device helpers for libinrhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "inrhip.h"

namespace inr {

// ---- error reporting (thread-local message, see inr_last_error) --------------------------------
void set_error(const char* fmt, ...);

#define INR_REQUIRE(cond, code, ...)             \
    do {                                         \
        if (!(cond)) {                           \
            ::inr::set_error(__VA_ARGS__);       \
            return (code);                       \
        }                                        \
    } while (0)

#define INR_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) {                                                        \
            ::inr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),    \
                             __FILE__, __LINE__);                                       \
            return (int)e__;                                                            \
        }                                                                               \
    } while (0)

// after a kernel launch
#define INR_LAUNCH_CHECK()                                                              \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            ::inr::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),\
                             __FILE__, __LINE__);                                       \
            return (int)e__;                                                            \
        }                                                                               \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- per-kernel-class event profiler (bench.py roofline) ---------------------------------------
enum KernelClass { KC_GEMM_FWD = 0, KC_GEMM_DX = 1, KC_GEMM_DW = 2, KC_OTHER = 3, KC_COUNT = 4 };
bool prof_enabled();
void prof_begin(int kernel_class, hipStream_t s);
void prof_end(int kernel_class, hipStream_t s);

struct ProfScope {
    int kc;
    hipStream_t s;
    bool on;
    ProfScope(int kc_, hipStream_t s_) : kc(kc_), s(s_), on(prof_enabled()) {
        if (on) prof_begin(kc, s);
    }
    ~ProfScope() {
        if (on) prof_end(kc, s);
    }
};

// ---- device math ---------------------------------------------------------------------------------
// sin and cos of the same fp32 argument, <= 1.6 ulp each for |x| < 65536 (3-term fma Cody-Waite
// reduction by pi/2 + minimax polynomials on [-pi/4, pi/4]); larger arguments take the libm path.
__device__ __forceinline__ void sincos_f32(float x, float& s, float& c) {
    if (__builtin_expect(!(fabsf(x) < 65536.0f), 0)) {
        sincosf(x, &s, &c);
        return;
    }
    const float kf = rintf(x * 0.636619772367581343f);
    float r = fmaf(-kf, 1.57079637050628662109e+00f, x);
    r = fmaf(-kf, -4.37113882867379288655e-08f, r);
    r = fmaf(-kf, -1.71512451000588187280e-15f, r);
    const float u = r * r;
    float ps = fmaf(2.7181986297364347e-06f, u, -0.00019839320157188922f);
    ps = fmaf(ps, u, 0.008333329111337662f);
    ps = fmaf(ps, u, -0.1666666716337204f);
    const float sr = fmaf(r * u, ps, r);
    float pc = fmaf(-2.7208204755879706e-07f, u, 2.479949216649402e-05f);
    pc = fmaf(pc, u, -0.0013888883404433727f);
    pc = fmaf(pc, u, 0.0416666679084301f);
    const float cr = fmaf(u * u, pc, fmaf(-0.5f, u, 1.0f));
    const int q = static_cast<int>(kf) & 3;
    const float s0 = (q & 1) ? cr : sr;
    const float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// Two-at-a-time form of sincos_f32 for the GEMM epilogues: the reduction and both polynomials are written
// on float2 so hipcc emits v_pk_fma_f32 / v_pk_mul_f32 (half the VALU instructions); quadrant fix-up by
// sign-bit xor.  Same constants and operation order per element as sincos_f32 => identical results.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sincos_f32x2(f32x2_t x, f32x2_t& s, f32x2_t& c) {
    if (__builtin_expect(!(fabsf(x[0]) < 65536.0f && fabsf(x[1]) < 65536.0f), 0)) {
        float s0, c0, s1, c1;
        sincos_f32(x[0], s0, c0);
        sincos_f32(x[1], s1, c1);
        s = f32x2_t{s0, s1};
        c = f32x2_t{c0, c1};
        return;
    }
    const f32x2_t t = x * 0.636619772367581343f;
    f32x2_t kf;
    kf[0] = rintf(t[0]);
    kf[1] = rintf(t[1]);
    const f32x2_t nk = -kf;
    f32x2_t r = __builtin_elementwise_fma(nk, (f32x2_t)(1.57079637050628662109e+00f), x);
    r = __builtin_elementwise_fma(nk, (f32x2_t)(-4.37113882867379288655e-08f), r);
    r = __builtin_elementwise_fma(nk, (f32x2_t)(-1.71512451000588187280e-15f), r);
    const f32x2_t u = r * r;
    f32x2_t ps = __builtin_elementwise_fma((f32x2_t)(2.7181986297364347e-06f), u, (f32x2_t)(-0.00019839320157188922f));
    ps = __builtin_elementwise_fma(ps, u, (f32x2_t)(0.008333329111337662f));
    ps = __builtin_elementwise_fma(ps, u, (f32x2_t)(-0.1666666716337204f));
    const f32x2_t sr = __builtin_elementwise_fma(r * u, ps, r);
    f32x2_t pc = __builtin_elementwise_fma((f32x2_t)(-2.7208204755879706e-07f), u, (f32x2_t)(2.479949216649402e-05f));
    pc = __builtin_elementwise_fma(pc, u, (f32x2_t)(-0.0013888883404433727f));
    pc = __builtin_elementwise_fma(pc, u, (f32x2_t)(0.0416666679084301f));
    const f32x2_t cr = __builtin_elementwise_fma(u * u, pc, __builtin_elementwise_fma((f32x2_t)(-0.5f), u, (f32x2_t)(1.0f)));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned q = static_cast<unsigned>(static_cast<int>(kf[e]));
        const float s0 = (q & 1u) ? cr[e] : sr[e];
        const float c0 = (q & 1u) ? sr[e] : cr[e];
        s[e] = __uint_as_float(__float_as_uint(s0) ^ ((q & 2u) << 30));
        c[e] = __uint_as_float(__float_as_uint(c0) ^ (((q + 1u) & 2u) << 30));
    }
}

// bit-exact torch.linspace(-1, 1, n)[i] in fp32 (oracle/inr_oracle.py: linspace_pm1)
__device__ __forceinline__ float linspace_pm1(int64_t i, int64_t n) {
    if (n <= 1) return -1.0f;
    const float step = 2.0f / static_cast<float>(n - 1);
    return (i < n / 2) ? fmaf(step, static_cast<float>(i), -1.0f)
                       : fmaf(-step, static_cast<float>(n - 1 - i), 1.0f);
}

}  // namespace inr
