// fp32 MFMA GEMM with fused epilogues -- the three dense contractions of the SIREN fit step.
//
//   forward   (K3, SRDWI.py:58-59)  act = sin(w*(x W^T + b)), dact = w*cos(...)   A k-contig, B k-contig
//   input-grad (K6)                 dz_prev = (dz W) * dact_prev                  A k-contig, B n-contig
//   param-grad (K6)                 gW = dz^T x  (split over rows, slabs)         A m-contig, B n-contig
//
// One kernel template: 128x128x32 block tile, 256 threads = 4 waves in a 2x2 grid, each wave owns a
// 64x64 sub-tile as 2x2 v_mfma_f32_32x32x2_f32 accumulators (exact fp32 products, fp32 accumulate --
// the only MFMA form on gfx950 that meets the 1e-5 parity tier; peak 157.3 TFLOP/s).
// Operand tiles are staged global -> registers -> LDS (double-buffered, one barrier per K-step).
// LDS images: k-contiguous operands as [rows][BK+4] (row stride 9 x 16 B: ds_read_b128 of 16 rows
// hits 16 distinct 16-B slots), m/n-contiguous operands as [BK][128] read with ds_read_b32.
// Within an 8-wide k block lane-half h consumes k = 4h..4h+3 for BOTH operands (a fixed permutation
// of the k-sum, so one b128 read feeds four MFMAs).
// blockIdx is remapped so that consecutive logical tiles (which share an A row-panel) land on the
// same XCD and reuse it from that XCD's L2.
#include "common.h"

namespace inr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, NTHREADS = 256;
constexpr int LDK = BK + 4;  // k-contiguous LDS row stride (floats)
constexpr int LDM = BM;      // m-contiguous LDS row stride (floats)

enum Epilogue { EPI_SINE = 0, EPI_SINE_STASH = 1, EPI_MUL = 2, EPI_PLAIN = 3 };

struct GemmParams {
    const float* A;
    const float* B;
    float* C;    // main output (act / dz_prev / slab base)
    float* C2;   // EPI_SINE_STASH: dact
    const float* bias;  // EPI_SINE*: per-column bias (nullable)
    const float* mul;   // EPI_MUL: element-wise factor, same layout as C (nullable -> plain)
    int M, N, K;
    int lda, ldb, ldc;
    float omega;
    int k_per_split;        // multiple of BK; == K rounded up when there is a single split
    int splits;
    long long slab_stride;  // floats between consecutive split slabs of C
    int tiles_m, tiles_n;
};

// ---- tile movers ---------------------------------------------------------------------------------
// k-contiguous operand: element (r, k) at P[r*ld + k]; LDS [r][k] stride LDK.
// r-contiguous operand: element (r, k) at P[k*ld + r]; LDS [k][r] stride LDM.
template <bool KCONTIG, bool VEC>
__device__ __forceinline__ void load_tile(f32x4 (&reg)[4], const float* __restrict__ P, int ld, int r0, int k0,
                                          int r_end, int k_end, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r, k;
        if (KCONTIG) {
            r = r0 + (tid >> 3) + 32 * i;
            k = k0 + (tid & 7) * 4;
        } else {
            k = k0 + (tid >> 5) + 8 * i;
            r = r0 + (tid & 31) * 4;
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
            if (r < r_end && k < k_end) {
                const float* src = KCONTIG ? (P + (long long)r * ld + k) : (P + (long long)k * ld + r);
                v = *reinterpret_cast<const f32x4*>(src);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = KCONTIG ? r : r + j;
                const int kk = KCONTIG ? k + j : k;
                if (rr < r_end && kk < k_end)
                    v[j] = KCONTIG ? P[(long long)rr * ld + kk] : P[(long long)kk * ld + rr];
            }
        }
        reg[i] = v;
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const f32x4 (&reg)[4], int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (KCONTIG) {
            const int r = (tid >> 3) + 32 * i, k = (tid & 7) * 4;
            *reinterpret_cast<f32x4*>(S + r * LDK + k) = reg[i];
        } else {
            const int k = (tid >> 5) + 8 * i, r = (tid & 31) * 4;
            *reinterpret_cast<f32x4*>(S + k * LDM + r) = reg[i];
        }
    }
}

// fragment of 4 k-values (k = kb*8 + 4h + 0..3) for row `r` of the tile
template <bool KCONTIG>
__device__ __forceinline__ f32x4 read_frag(const float* __restrict__ S, int r, int kb, int h) {
    if (KCONTIG) {
        return *reinterpret_cast<const f32x4*>(S + r * LDK + kb * 8 + 4 * h);
    } else {
        const float* p = S + (kb * 8 + 4 * h) * LDM + r;
        f32x4 v;
        v[0] = p[0];
        v[1] = p[LDM];
        v[2] = p[2 * LDM];
        v[3] = p[3 * LDM];
        return v;
    }
}

template <bool KC>
struct TileSize {
    static constexpr int floats = KC ? BM * LDK : BK * LDM;
};

// bijective XCD-aware remap: physical block id -> logical id such that logical ids that are close
// together run on the same XCD (blocks are dealt round-robin over the 8 XCDs).
__device__ __forceinline__ int xcd_remap(int pid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = pid & 7, idx = pid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <bool A_KC, bool B_KC, int EPI, bool VEC>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (TileSize<A_KC>::floats + TileSize<B_KC>::floats)];
    constexpr int STAGE = TileSize<A_KC>::floats + TileSize<B_KC>::floats;
    constexpr int BOFF = TileSize<A_KC>::floats;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l32 = lane & 31;

    const int total = p.tiles_m * p.tiles_n * p.splits;
    int logical = xcd_remap(blockIdx.x, total);
    const int tile_n = logical % p.tiles_n;
    logical /= p.tiles_n;
    const int tile_m = logical % p.tiles_m;
    const int split = logical / p.tiles_m;

    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = split * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int ktiles = (k_end - k_begin + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    if (ktiles > 0) {
        load_tile<A_KC, VEC>(ra, p.A, p.lda, m0, k_begin, p.M, k_end, tid);
        load_tile<B_KC, VEC>(rb, p.B, p.ldb, n0, k_begin, p.N, k_end, tid);
        store_tile<A_KC>(smem, ra, tid);
        store_tile<B_KC>(smem + BOFF, rb, tid);
    }
    __syncthreads();

    for (int t = 0; t < ktiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < ktiles);
        if (more) {
            const int k0 = k_begin + (t + 1) * BK;
            load_tile<A_KC, VEC>(ra, p.A, p.lda, m0, k0, p.M, k_end, tid);
            load_tile<B_KC, VEC>(rb, p.B, p.ldb, n0, k0, p.N, k_end, tid);
        }
        const float* cA = smem + cur * STAGE;
        const float* cB = cA + BOFF;
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = read_frag<A_KC>(cA, wm * 64 + i * 32 + l32, kb, h);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<B_KC>(cB, wn * 64 + j * 32 + l32, kb, h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (more) {
            store_tile<A_KC>(smem + (cur ^ 1) * STAGE, ra, tid);
            store_tile<B_KC>(smem + (cur ^ 1) * STAGE + BOFF, rb, tid);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)
    float* __restrict__ C = p.C + (long long)split * p.slab_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l32;
        if (col >= p.N) continue;
        float bias = 0.f;
        if (EPI == EPI_SINE || EPI == EPI_SINE_STASH) bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= p.M) continue;
                const long long off = (long long)row * p.ldc + col;
                const float v = acc[i][j][r];
                if (EPI == EPI_SINE || EPI == EPI_SINE_STASH) {
                    float s, c;
                    sincos_f32(p.omega * (v + bias), s, c);
                    C[off] = s;
                    if (EPI == EPI_SINE_STASH) p.C2[off] = p.omega * c;
                } else if (EPI == EPI_MUL) {
                    C[off] = v * p.mul[off];
                } else {
                    C[off] = v;
                }
            }
        }
    }
}

// ---- host-side launch --------------------------------------------------------------------------
template <bool A_KC, bool B_KC, int EPI>
static int launch_gemm(GemmParams p, bool vec, hipStream_t stream) {
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const long long total = (long long)p.tiles_m * p.tiles_n * p.splits;
    INR_REQUIRE(total > 0 && total < (1ll << 31), INR_E_INVALID, "gemm grid out of range (%lld blocks)", total);
    dim3 grid((unsigned)total), block(NTHREADS);
    if (vec)
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, EPI, true>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, EPI, false>), grid, block, 0, stream, p);
    INR_LAUNCH_CHECK();
    return 0;
}

static inline bool vec_ok(const void* a, const void* b, int lda, int ldb, int ka, int kb) {
    return aligned16(a) && aligned16(b) && (lda % 4 == 0) && (ldb % 4 == 0) && (ka % 4 == 0) && (kb % 4 == 0);
}

// act[n][out] = sin(omega*(x[n][in] W[out][in]^T + b)), optional dact = omega*cos(...)
int gemm_sine_forward(float* act, float* dact, const float* x, const float* W, const float* b, int64_t n,
                      int in_f, int out_f, float omega, hipStream_t stream) {
    GemmParams p{};
    p.A = x; p.B = W; p.C = act; p.C2 = dact; p.bias = b; p.mul = nullptr;
    p.M = (int)n; p.N = out_f; p.K = in_f;
    p.lda = in_f; p.ldb = in_f; p.ldc = out_f;
    p.omega = omega;
    p.splits = 1;
    p.k_per_split = (in_f + BK - 1) / BK * BK;
    p.slab_stride = 0;
    const bool vec = vec_ok(x, W, in_f, in_f, in_f, in_f);
    ProfScope ps(KC_GEMM_FWD, stream);
    if (dact) return launch_gemm<true, true, EPI_SINE_STASH>(p, vec, stream);
    return launch_gemm<true, true, EPI_SINE>(p, vec, stream);
}

// dz_prev[n][in] = (dz[n][out] @ W[out][in]) * mul[n][in]   (mul nullable)
int gemm_input_grad(float* dz_prev, const float* dz, const float* W, const float* mul, int64_t n, int in_f,
                    int out_f, hipStream_t stream) {
    GemmParams p{};
    p.A = dz; p.B = W; p.C = dz_prev; p.C2 = nullptr; p.bias = nullptr; p.mul = mul;
    p.M = (int)n; p.N = in_f; p.K = out_f;
    p.lda = out_f; p.ldb = in_f; p.ldc = in_f;
    p.omega = 0.f;
    p.splits = 1;
    p.k_per_split = (out_f + BK - 1) / BK * BK;
    p.slab_stride = 0;
    // A k-contig: needs K%4; B n-contig: needs N%4 (a float4 runs along n)
    const bool vec = vec_ok(dz, W, out_f, in_f, out_f, in_f);
    ProfScope ps(KC_GEMM_DX, stream);
    if (mul) return launch_gemm<true, false, EPI_MUL>(p, vec, stream);
    return launch_gemm<true, false, EPI_PLAIN>(p, vec, stream);
}

// number of row-splits used for the parameter-gradient contraction over n rows
int param_grad_splits(int64_t n, int in_f, int out_f) {
    const long long tiles = (long long)((out_f + BM - 1) / BM) * ((in_f + BN - 1) / BN);
    const long long ksteps = (n + BK - 1) / BK;
    long long want = (1024 + tiles - 1) / tiles;        // ~4 blocks per CU in flight
    long long max_by_work = (ksteps + 7) / 8;           // at least 8 K-steps per split
    long long s = want < max_by_work ? want : max_by_work;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return (int)s;
}

// slabs[splits][out][in] = partial dz^T x over row ranges
int gemm_param_grad_slabs(float* slabs, int splits, const float* dz, const float* x, int64_t n, int in_f,
                          int out_f, hipStream_t stream) {
    GemmParams p{};
    p.A = dz; p.B = x; p.C = slabs; p.C2 = nullptr; p.bias = nullptr; p.mul = nullptr;
    p.M = out_f; p.N = in_f; p.K = (int)n;
    p.lda = out_f; p.ldb = in_f; p.ldc = in_f;
    p.omega = 0.f;
    p.splits = splits;
    const long long ksteps = (n + BK - 1) / BK;
    p.k_per_split = (int)((ksteps + splits - 1) / splits) * BK;
    p.slab_stride = (long long)out_f * in_f;
    const bool vec = vec_ok(dz, x, out_f, in_f, out_f, in_f);
    ProfScope ps(KC_GEMM_DW, stream);
    return launch_gemm<false, false, EPI_PLAIN>(p, vec, stream);
}

}  // namespace inr
