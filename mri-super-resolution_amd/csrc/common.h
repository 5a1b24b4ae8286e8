// Shared host/device helpers for libinrhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
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
// gemm_f32.hip is compiled without packed fp32 VALU instructions (_build.py: SOURCE_FLAGS -- they slow the MFMA kernels' epilogues);
// kernels with no MFMA beside their VALU work switch them back on.  A no-op in translation units built with the default feature set.
#define INR_PACKED_F32 __attribute__((target("packed-fp32-ops")))

// inputs of the split-fp16 GEMM path (gemm_h3.inc); a null pointer / default-constructed value selects the fp32 MFMA
struct H3Args {
    const unsigned* a_amax = nullptr;   // float bits of max|A| (dz operands); null: A unscaled (activations, |A| <= 1)
    const unsigned* b_amax = nullptr;   // amax the pre-split B planes were scaled with
    const _Float16* Bh = nullptr;       // pre-split weight planes, k-contiguous for the GEMM at hand
    const _Float16* Bl = nullptr;
    unsigned* amax_out = nullptr;       // input-grad: receives max|dz_prev|
    int reverse_m = 0;                  // serpentine row-tile order between consecutive kernels (Infinity Cache reuse)
};

// power-of-two scale of an HL32 tensor (gemm_hp.inc) from an a-priori bound: bound = [*meas as float bits] * [*wn] * mul
// (missing factors = 1).  The producer scales by it, every consumer undoes it -- all by evaluating the same expression.
struct HpScale {
    const unsigned* meas = nullptr;   // float bits (an atomic-max slot of a FINISHED kernel), nullable
    const float* wn = nullptr;        // device float, nullable
    float mul = 0.f;                  // 0 = no scale at all
    int kmax = 126;                   // largest exponent of the scale (bounds with `wn`: gradients use the whole range; sine outputs
                                      // stop at 40 so that the folded bias b 2^(ka + kb) stays finite)
};

// one parameter-gradient GEMM of a step (hp_param_grad_multi: all of them in one launch)
struct HpParamGradJob {
    float* slabs;
    int splits;
    const char* dz_hl;
    const char* x_hl;
    int in_f, out_f;
    HpScale sa, sb;
};

// Process-global diagnostic switches behind inr_debug_set (atomics: a read races with nothing, but a switch flipped while
// another thread is enqueueing changes that thread's kernel selection -- diagnostic use only, see include/inrhip.h).
typedef std::atomic<int> tune_int;

// Which kernel family a host launcher picked: counted per process so that a test can assert that the family it means to
// cover is the one that ran (inr_launch_count).  Keep in step with INR_LF_* in include/inrhip.h.
enum LaunchFamily {
    LF_HP_PKD = 0,      // gemm_hp_pkd_kernel   persistent, deferred epilogue (HL32 operands)
    LF_HP_PKC = 1,      // gemm_hp_pkc_kernel   persistent, epilogue in line
    LF_HP_TILE = 2,     // gemm_hp_kernel<HP_KC> one block per tile
    LF_HP_RC = 3,       // gemm_hp_kernel<HP_RC> parameter gradient (row contraction)
    LF_H3 = 4,          // gemm_h3_kernel       split-fp16, operands split in the consumer
    LF_F32_PIPE16 = 5,  // gemm_f32_pipe16_kernel
    LF_F32_PIPE = 6,    // gemm_f32_pipe_kernel
    LF_F32_GENERIC = 7, // gemm_f32_kernel
    LF_SMALL_MULTI = 8, // siren_small_multi_kernel (persistent cooperative)
    LF_SMALL_STEP = 9,  // siren_small step kernel pair
    LF_HP_NARROW = 10,  // gemm_hp_nt_kernel    64 x 128 tiles (launches that cannot fill the chip with wide tiles)
    LF_HP_FUSED_FWD = 11,   // siren_fwd_fused_kernel: all sine layers + head of an inference forward in one launch
    LF_HP_ROW = 12,     // gemm_hp_row_kernel   persistent, one block owns 128 rows x all 512 columns, epilogue in line (round 5)
    LF_COUNT = 13
};
void count_launch(int family);
#define INR_E_FALLBACK (-100)   // internal: the chosen kernel cannot run on this device, the caller takes its next-best path

// ---- deferred gradient reduction of the fused fit (kernels.hip: finalize_kernel) -------------------------------------------
// Every gradient tensor of a step is a fixed-order sum of slab rows its producer left behind (row-split parameter-gradient
// GEMMs, per-tile column sums of the input-gradient epilogues, per-block sums of the head step).  Round 2 reduced each tensor
// right behind its producer -- 9 to 14 launches of ~5 us -- and ran Adam as one more; here the producers only write, and ONE
// launch at the end of the step sums every tensor, finishes the loss and takes the Adam step (two launches when a slab stack is
// tall enough to want a first stage).  No float atomics anywhere: runs stay bitwise reproducible.
constexpr int FIN_MAX_SEG = 20;
constexpr int FIN_GROUP = 32;         // rows per first-stage group
constexpr int FIN_TALL = 4 * FIN_GROUP;   // stacks above this many rows get a first stage
struct FinalizeSeg {
    const float* slab;     // [nslabs][len]
    float* stage1;         // [ceil(nslabs / FIN_GROUP)][len] when nslabs > FIN_TALL, else unused
    long long dst;         // offset of the tensor in the flat parameter / gradient buffers
    long long len;
    int nslabs;
};
struct FinalizeJob {
    FinalizeSeg seg[FIN_MAX_SEG];
    long long first[FIN_MAX_SEG + 1];   // prefix sums of len
    long long s1_first[FIN_MAX_SEG + 1];   // prefix sums of the first-stage blocks per segment (0 for stacks that need none)
    int nseg;
    const float* part_loss;             // [nparts] per-block loss terms (nullable)
    int nparts;
    float loss_scale;
    float* loss_out;
    float* grads;
    float *params, *m, *v;              // params == nullptr: reduce only (inr_siren_loss_grad)
    float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps;
};
int launch_finalize(FinalizeJob& job, long long adam_step, double lr, double b1, double b2, double eps, hipStream_t st);

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
// sin and cos of one fp32 argument.  On gfx950 the f32 MFMA and the f32 VALU share the same lanes (measured:
// a wave's VALU epilogue does not overlap the co-resident wave's v_mfma_f32_32x32x2_f32 stream), so every VALU
// instruction in a GEMM epilogue costs matrix time.  The argument is therefore reduced to a FRACTION OF A
// REVOLUTION with two FMAs -- f = x/(2*pi) - rint(x/(2*pi)), |f| <= 0.5, single rounding of the exact
// difference plus the 1/(2*pi) tail -- and handed to the transcendental unit (v_sin_f32 / v_cos_f32 take
// revolutions).  6 instructions per element instead of ~24; measured abs error <= 3e-7 for |x| < 2^20
// (hardware sin/cos: 1.4e-7 on [-0.5, 0.5] rev; argument: <= 3e-8 rev).  Larger arguments take the libm path.
#define INR_INV_2PI_HI 1.59154936671257019e-01f
#define INR_INV_2PI_LO 6.42063824329852650e-09f
#define INR_SINCOS_FAST_LIMIT 1048576.0f

__device__ __forceinline__ void sincos_f32(float x, float& s, float& c) {
    if (__builtin_expect(!(fabsf(x) < INR_SINCOS_FAST_LIMIT), 0)) {
        sincosf(x, &s, &c);
        return;
    }
    const float k = rintf(x * INR_INV_2PI_HI);
    float f = fmaf(x, INR_INV_2PI_HI, -k);
    f = fmaf(x, INR_INV_2PI_LO, f);
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
}

// Branch-free core of sincos_f32 on float2 (valid for |x| < INR_SINCOS_FAST_LIMIT; callers check that once per
// tile and redo the tile through sincos_f32 otherwise, so the unrolled epilogue carries no libm code).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sincos_f32x2_fast(f32x2_t x, f32x2_t& s, f32x2_t& c) {
    const f32x2_t t = x * INR_INV_2PI_HI;
    const f32x2_t k = f32x2_t{rintf(t[0]), rintf(t[1])};
    f32x2_t f = __builtin_elementwise_fma(x, (f32x2_t)(INR_INV_2PI_HI), -k);
    f = __builtin_elementwise_fma(x, (f32x2_t)(INR_INV_2PI_LO), f);
    s = f32x2_t{__builtin_amdgcn_sinf(f[0]), __builtin_amdgcn_sinf(f[1])};
    c = f32x2_t{__builtin_amdgcn_cosf(f[0]), __builtin_amdgcn_cosf(f[1])};
}

// bit-exact torch.linspace(-1, 1, n)[i] in fp32 (oracle/inr_oracle.py: linspace_pm1)
__device__ __forceinline__ float linspace_pm1(int64_t i, int64_t n) {
    if (n <= 1) return -1.0f;
    const float step = 2.0f / static_cast<float>(n - 1);
    return (i < n / 2) ? fmaf(step, static_cast<float>(i), -1.0f)
                       : fmaf(-step, static_cast<float>(n - 1 - i), 1.0f);
}

}  // namespace inr
