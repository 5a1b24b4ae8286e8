// fp32 MFMA GEMM with fused epilogues -- the three dense contractions of the SIREN fit step.
//
//   forward   (K3, SRDWI.py:58-59)  act = sin(w*(x W^T + b)), dact = w*cos(...)   A k-contig, B k-contig
//   input-grad (K6)                 dz_prev = (dz W) * dact_prev  (+ bias-grad sums) A k-contig, B n-contig
//   param-grad (K6)                 gW = dz^T x  (split over rows, slabs)         A m-contig, B n-contig
//
// 128x128x32 block tile, 256 threads = 4 waves in a 2x2 grid, each wave owns a 64x64 sub-tile.  Exact fp32 products
// with fp32 accumulation (f32-input MFMA): the only operand type on gfx950 that meets the 1e-5 parity tier; dense
// peak 157.3 TFLOP/s.  Three kernels share the tile geometry, the staging scheme and the epilogue:
//   gemm_f32_pipe16_kernel -- DEFAULT.  v_mfma_f32_16x16x4_f32, 4x4 accumulators per wave; within a 16-wide k block
//       lane group g = lane/16 consumes k = 4g..4g+3 for BOTH operands (a fixed permutation of the k-sum, so one
//       b128 LDS read feeds four MFMAs).  Operands arrive through buffer loads (per-block SRD: rows past the matrix
//       end read as 0, no exec-mask branches) into registers, fragments are prefetched one k block ahead into a
//       second register set, the next K-tile's global loads / LDS stores are interleaved between the MFMAs with
//       sched_group_barrier, and the single barrier per K-step sits in the middle of the second k block so the MFMAs
//       behind it already hold their operands.  Epilogue: the accumulators are parked in the (idle) operand LDS and
//       read back row-contiguous -> float4 buffer stores, hardware sin/cos on an FMA-reduced argument, dact fetched
//       under the last K-tile, bias-gradient column sums.
//   gemm_f32_pipe_kernel   -- the same structure on v_mfma_f32_32x32x2_f32 (2x2 accumulators, k = 4h..4h+3 per
//       8-block).  Kept for A/B (inr_debug_set(1, 0)): 5-7 % slower (tools/gemm_ab.py).
//   gemm_f32_kernel        -- generic fallback (any shape/alignment: scalar guarded loads), 32x32x2, plain loop.
// blockIdx is remapped so that consecutive logical tiles (which share an A row-panel) land on the same XCD.
//
// Measured facts that shaped it (s_memtime stamps: tools/stamp_timeline.py; tools/mfma_rate.hip, tools/gemm_ab.py):
//   * the f32 MFMA shares the f32 VALU lanes: a co-resident wave's VALU instructions issue about once per MFMA
//     boundary, so an epilogue that overlaps the other block's main loop is stretched ~4x and every VALU instruction
//     in it counts (v_sin/v_cos on a reduced argument, 16-byte staged stores, fused bias sums);
//   * register-only loops reach 138-142 TFLOP/s with 32x32x2 and 147-150 with 16x16x4 on this part; in the full
//     kernel 16x16x4 gives fwd 136 / input-grad 141 / param-grad 147 (32x32x2: 130 / 131 / 141);
//   * tried without effect: start-time staggering of co-resident blocks or of the whole first wave, s_setprio in
//     either direction; tried and slower: a persistent grid with cross-tile operand prefetch (-4 %).
#include "common.h"
#include <type_traits>

namespace inr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, NTHREADS = 256;
constexpr int LDK = BK + 4;  // k-contiguous LDS row stride (floats)
constexpr int LDM = BM;      // m-contiguous LDS row stride (floats)

enum Epilogue { EPI_SINE = 0, EPI_SINE_STASH = 1, EPI_MUL = 2, EPI_PLAIN = 3, EPI_TANH = 4, EPI_TANH_STASH = 5 };
// EPI_TANH*: act = omega*tanh(acc + bias), stash = omega*(1 - tanh^2)   (PerturbNet layers, SRDWI.py:103-107; `omega` = scale)

struct GemmParams {
    const float* A;
    const float* B;
    float* C;    // main output (act / dz_prev / slab base)
    float* C2;   // EPI_SINE_STASH: dact
    const float* bias;  // EPI_SINE*: per-column bias (nullable)
    const float* mul;   // EPI_MUL: element-wise factor, same layout as C (nullable -> plain)
    float* colsum;      // EPI_MUL fast path: slab [2*tiles_m][N] of per-64-row column sums of C (nullable)
    int M, N, K;
    int lda, ldb, ldc;
    float omega;
    int k_per_split;        // multiple of BK; == K rounded up when there is a single split
    int splits;
    long long slab_stride;  // floats between consecutive split slabs of C
    int tiles_m, tiles_n;
    long long a_elems, b_elems, c_elems;  // total element counts of A, B, C (SRD bounds of the fast path)
    unsigned long long* stamps;  // diagnostic builds (-DINR_STAMPS): 6 x u64 per wave
    // split-fp16 path (gemm_h3.inc); all null/zero on the fp32 path
    const unsigned* a_amax;      // float bits of max|A| (device); null = A is used unscaled (|A| <= 1: activations)
    const unsigned* b_amax;      // same for B (for pre-split planes: the amax the planes were scaled with)
    const _Float16* Bh;          // pre-split B planes (k-contiguous, leading dimension ldb halves)
    const _Float16* Bl;
    unsigned* amax_out;          // EPI_MUL: receives max|C| (atomic max on the float bits), nullable
    int reverse_m;               // walk the row tiles from the end: the rows the previous kernel wrote last (still in the
                                 // 256 MB Infinity Cache) are read first
};

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

struct TileCoord {
    int tile_m, tile_n, split;
};
__device__ __forceinline__ TileCoord decode_block(const GemmParams& p) {
    const int total = p.tiles_m * p.tiles_n * p.splits;
    int logical = xcd_remap(blockIdx.x, total);
    TileCoord c;
    c.tile_n = logical % p.tiles_n;
    logical /= p.tiles_n;
    c.tile_m = logical % p.tiles_m;
    c.split = logical / p.tiles_m;
    if (p.reverse_m) {   // row tiles (or, for the row-split parameter gradient, row ranges) from the end
        if (p.splits > 1) c.split = p.splits - 1 - c.split;
        else c.tile_m = p.tiles_m - 1 - c.tile_m;
    }
    return c;
}

// ---- LDS tile movers (shared by both kernels) --------------------------------------------------------
// k-contiguous operand: element (r, k) at P[r*ld + k]; LDS [r][k] stride LDK.
// r-contiguous operand: element (r, k) at P[k*ld + r]; LDS [k][r] stride LDM.
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

struct Frags {
    f32x4 a[2], b[2];
};

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void read_frags(Frags& f, const float* __restrict__ sA, const float* __restrict__ sB,
                                           int arow, int brow, int kb, int h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) f.a[i] = read_frag<A_KC>(sA, arow + i * 32, kb, h);
#pragma unroll
    for (int j = 0; j < 2; ++j) f.b[j] = read_frag<B_KC>(sB, brow + j * 32, kb, h);
}

__device__ __forceinline__ void mfma_block(f32x16 (&acc)[2][2], const Frags& f) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[j][s], acc[i][j], 0, 0, 0);
}

// ---- epilogue (C/D map of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)) ----
template <int EPI, bool CHECK>
__device__ __forceinline__ void epilogue(const GemmParams& p, const f32x16 (&acc)[2][2], int m0, int n0, int wm,
                                         int wn, int h, int l32, int split) {
    float* __restrict__ C = p.C + (long long)split * p.slab_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l32;
        if (CHECK && col >= p.N) continue;
        float bias = 0.f;
        if (EPI == EPI_SINE || EPI == EPI_SINE_STASH || EPI == EPI_TANH || EPI == EPI_TANH_STASH)
            bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row_base = m0 + wm * 64 + i * 32 + 4 * h;
            const long long base = (long long)row_base * p.ldc + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (CHECK && row_base + dr >= p.M) continue;
                const long long off = base + (long long)dr * p.ldc;
                const float v = acc[i][j][r];
                if (EPI == EPI_SINE || EPI == EPI_SINE_STASH) {
                    float s, c;
                    sincos_f32(p.omega * (v + bias), s, c);
                    C[off] = s;
                    if (EPI == EPI_SINE_STASH) p.C2[off] = p.omega * c;
                } else if (EPI == EPI_TANH || EPI == EPI_TANH_STASH) {
                    const float t = tanhf(v + bias);
                    C[off] = p.omega * t;
                    if (EPI == EPI_TANH_STASH) p.C2[off] = p.omega * (1.0f - t * t);
                } else if (EPI == EPI_MUL) {
                    C[off] = v * p.mul[off];
                } else {
                    C[off] = v;
                }
            }
        }
    }
}

// =====================================================================================================
// generic kernel: guarded (optionally scalar) global loads, plain double-buffered loop
// =====================================================================================================
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

template <bool A_KC, bool B_KC, int EPI, bool VEC>
__global__ void __launch_bounds__(NTHREADS, 2) INR_PACKED_F32 gemm_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (TileSize<A_KC>::floats + TileSize<B_KC>::floats)];
    constexpr int STAGE = TileSize<A_KC>::floats + TileSize<B_KC>::floats;
    constexpr int BOFF = TileSize<A_KC>::floats;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l32 = lane & 31;
    const TileCoord tc = decode_block(p);
    const int m0 = tc.tile_m * BM, n0 = tc.tile_n * BN;
    const int k_begin = tc.split * p.k_per_split;
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
            Frags f;
            read_frags<A_KC, B_KC>(f, cA, cB, wm * 64 + l32, wn * 64 + l32, kb, h);
            mfma_block(acc, f);
        }
        if (more) {
            store_tile<A_KC>(smem + (cur ^ 1) * STAGE, ra, tid);
            store_tile<B_KC>(smem + (cur ^ 1) * STAGE + BOFF, rb, tid);
        }
        __syncthreads();
    }
    epilogue<EPI, true>(p, acc, m0, n0, wm, wn, h, l32, tc.split);
}

// =====================================================================================================
// pipelined kernel (fast path)
// =====================================================================================================
// per-block buffer resource over [base, base + bytes): loads past the end return 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const float* base, long long bytes) {
    if (bytes < 0) bytes = 0;
    if (bytes > 0xFFFFFFFFll) bytes = 0xFFFFFFFFll;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(unsigned)bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t srd, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, 0);
    return __builtin_bit_cast(f32x4, v);
}

// byte offsets (relative to the block's SRD base) of this thread's 4 float4 of a tile at k-tile 0
template <bool KCONTIG>
__device__ __forceinline__ void tile_voffsets(int (&voff)[4], int ld, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (KCONTIG)
            voff[i] = (((tid >> 3) + 32 * i) * ld + (tid & 7) * 4) * 4;
        else
            voff[i] = (((tid >> 5) + 8 * i) * ld + (tid & 31) * 4) * 4;
    }
}

// scheduling recipe for one 8-wide k block: 16 MFMAs with NR LDS fragment reads and NX extra memory
// instructions (mask XMASK: 0x20 = VMEM read, 0x200 = DS write) spread evenly between them
template <int NR, int NX, int XMASK, int I>
__device__ __forceinline__ void sched_interleave() {
    if constexpr (I < 16) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int r = (I + 1) * NR / 16 - I * NR / 16;
        if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, r, 0);
        constexpr int x = (I + 1) * NX / 16 - I * NX / 16;
        if constexpr (x > 0) __builtin_amdgcn_sched_group_barrier(XMASK, x, 0);
        sched_interleave<NR, NX, XMASK, I + 1>();
    }
}

#ifdef INR_STAMPS
#define INR_STAMP(slot)                                                                              \
    do {                                                                                             \
        if (p.stamps && (threadIdx.x & 63) == 0) {                                                   \
            unsigned long long t_;                                                                   \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            p.stamps[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = t_;            \
        }                                                                                            \
    } while (0)
#else
#define INR_STAMP(slot)
#endif

// NOTE: the row offset is folded into the VGPR offset and soffset stays the constant 0.  With an SGPR soffset
// a 16-byte buffer store reads its data registers late, and on gfx950/ROCm 7.2 hipcc let the next VALU
// instruction overwrite them (observed: lanes 12-15 of every 16 stored the FOLLOWING store's second dword).
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t srd, int voff, int row_off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srd, voff + row_off, 0, 0);
}

// Epilogue of the pipelined kernel.  Each wave parks its 64x64 accumulator sub-tile in its own 16 KB of
// the (now idle) operand LDS and reads it back row-contiguous, so the activation is applied to float4s
// (packed-math sincos) and every global access is a 16-byte buffer op: 16 stores per output array per wave
// instead of 64 dword stores, out-of-range rows/columns dropped by the SRD bounds check (no branches).
//   q-th float4 of a lane: row = 4q + lane/16, cols = 4*(lane%16) .. +3 of the wave's sub-tile.
struct EpiAddr {
    __amdgpu_buffer_rsrc_t srdC, srdC2, srdMul;
    int voff;       // byte offset of the lane's first float4 inside the block's output window
    int row_step;   // bytes between q and q+1 (4 rows)
};

template <int EPI>
__device__ __forceinline__ EpiAddr epi_addr(const GemmParams& p, int m0, int n0, int wm, int wn, int lane, int split) {
    EpiAddr a;
    const long long first = (long long)split * p.slab_stride + (long long)m0 * p.ldc + n0;
    const long long c_end = (p.splits > 1 || p.slab_stride) ? (long long)(split + 1) * p.slab_stride : p.c_elems;
    const long long bytes = min(c_end - first, (long long)BM * p.ldc) * 4;   // this block's 128-row window only
    a.srdC = make_srd(p.C + first, bytes);
    constexpr bool STASH = (EPI == EPI_SINE_STASH || EPI == EPI_TANH_STASH);
    a.srdC2 = make_srd(STASH ? p.C2 + first : p.C, STASH ? bytes : 0);
    a.srdMul = make_srd(EPI == EPI_MUL ? p.mul + first : p.A, EPI == EPI_MUL ? bytes : 0);
    const int row = wm * 64 + (lane >> 4), col = wn * 64 + (lane & 15) * 4;
    // a lane whose columns fall outside N is pushed out of the SRD range: its loads read 0, its stores drop
    a.voff = (n0 + col < p.N) ? (row * p.ldc + col) * 4 : 0x7FFFFF00;
    a.row_step = 4 * p.ldc * 4;
    return a;
}

template <int EPI, int SUBS>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, float* __restrict__ sub, const EpiAddr& a,
                                              const f32x4 (&mulreg)[16], int n0, int wn, int lane, int slab_row);

template <int EPI>
__device__ __forceinline__ void epilogue_staged(const GemmParams& p, const f32x16 (&acc)[2][2], float* __restrict__ sub,
                                                const EpiAddr& a, const f32x4 (&mulreg)[16], int n0, int wn, int lane,
                                                int slab_row) {
    const int h = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sub[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + l32] = acc[i][j][r];
    epilogue_rows<EPI, 64>(p, sub, a, mulreg, n0, wn, lane, slab_row);
}

// second half of the staged epilogue: the wave's 64x64 sub-tile sits in `sub` as [row][SUBS]; lane's q-th float4 is
// row 4q + lane/16, columns 4*(lane%16) .. +3
template <int EPI, int SUBS>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, float* __restrict__ sub, const EpiAddr& a,
                                              const f32x4 (&mulreg)[16], int n0, int wn, int lane, int slab_row) {
    INR_STAMP(5);
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_SINE || EPI == EPI_SINE_STASH || EPI == EPI_TANH || EPI == EPI_TANH_STASH) {
        const int col = n0 + wn * 64 + (lane & 15) * 4;
        if (p.bias && col < p.N) bias = *reinterpret_cast<const f32x4*>(p.bias + col);
    }
    const float* rd = sub + (lane >> 4) * SUBS + (lane & 15) * 4;
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    float zmax = 0.f;   // largest |omega*z| seen by this lane (sine epilogues)
    float omax = 0.f;   // largest |output| (EPI_MUL with amax_out: scale of the next split-fp16 GEMM)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rd + q * 4 * SUBS);
        const int so = q * a.row_step;
        if (q == 8) INR_STAMP(6);
        if (EPI == EPI_SINE || EPI == EPI_SINE_STASH) {
            const f32x4 z = p.omega * (v + bias);
            zmax = fmaxf(fmaxf(zmax, fmaxf(fabsf(z[0]), fabsf(z[1]))), fmaxf(fabsf(z[2]), fabsf(z[3])));
            f32x2_t s01, c01, s23, c23;
            sincos_f32x2_fast(f32x2_t{z[0], z[1]}, s01, c01);
            sincos_f32x2_fast(f32x2_t{z[2], z[3]}, s23, c23);
            buf_store4(f32x4{s01[0], s01[1], s23[0], s23[1]}, a.srdC, a.voff, so);
            if (EPI == EPI_SINE_STASH)
                buf_store4(p.omega * f32x4{c01[0], c01[1], c23[0], c23[1]}, a.srdC2, a.voff, so);
        } else if (EPI == EPI_TANH || EPI == EPI_TANH_STASH) {
            const f32x4 z = v + bias;
            const f32x4 t = f32x4{tanhf(z[0]), tanhf(z[1]), tanhf(z[2]), tanhf(z[3])};
            buf_store4(p.omega * t, a.srdC, a.voff, so);
            if (EPI == EPI_TANH_STASH) buf_store4(p.omega * (1.0f - t * t), a.srdC2, a.voff, so);
        } else if (EPI == EPI_MUL) {
            const f32x4 o = v * mulreg[q];   // rows past M: v == 0 and mulreg == 0, so they add nothing below
            csum += o;
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            buf_store4(o, a.srdC, a.voff, so);
        } else {
            buf_store4(v, a.srdC, a.voff, so);
        }
    }
    if (EPI == EPI_SINE || EPI == EPI_SINE_STASH) {
        // rare: an argument beyond the fast range (or NaN) somewhere in the wave's tile -> redo it through libm
        if (__builtin_expect(__any(!(zmax < INR_SINCOS_FAST_LIMIT)), 0)) {
#pragma unroll 1
            for (int q = 0; q < 16; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rd + q * 4 * SUBS);
                const f32x4 z = p.omega * (v + bias);
                f32x4 sv, cv;
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {
                    float s1, c1;
                    sincos_f32(z[e], s1, c1);
                    sv[e] = s1;
                    cv[e] = c1;
                }
                buf_store4(sv, a.srdC, a.voff, q * a.row_step);
                if (EPI == EPI_SINE_STASH) buf_store4(p.omega * cv, a.srdC2, a.voff, q * a.row_step);
            }
        }
    }
    if (EPI == EPI_MUL && p.amax_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off, 64));
        if (lane == 0) atomicMax(p.amax_out, __float_as_uint(omax));   // max is order independent: runs stay reproducible
    }
    if (EPI == EPI_MUL && p.colsum) {
        // bias gradient of the layer below = column sums of this tile: 16 rows per lane, then the 4 lanes that
        // share a column group (lane, lane^16, lane^32, lane^48) in a fixed order -> one slab row per wave
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            csum[e] += __shfl_xor(csum[e], 16, 64);
            csum[e] += __shfl_xor(csum[e], 32, 64);
        }
        const int col = n0 + wn * 64 + (lane & 15) * 4;
        if (lane < 16 && col < p.N) *reinterpret_cast<f32x4*>(p.colsum + (long long)slab_row * p.N + col) = csum;
    }
}

template <bool A_KC, bool B_KC, int EPI>
__global__ void __launch_bounds__(NTHREADS, 2) INR_PACKED_F32 gemm_f32_pipe_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (TileSize<A_KC>::floats + TileSize<B_KC>::floats)];
    constexpr int STAGE = TileSize<A_KC>::floats + TileSize<B_KC>::floats;
    constexpr int BOFF = TileSize<A_KC>::floats;
    constexpr int NR = (A_KC ? 2 : 8) + (B_KC ? 2 : 8);  // LDS read instructions per k block

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l32 = lane & 31;
    const TileCoord tc = decode_block(p);
    const int m0 = tc.tile_m * BM, n0 = tc.tile_n * BN;
    const int k_begin = tc.split * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int ktiles = (k_end - k_begin + BK - 1) / BK;
#ifdef INR_STAMPS
    if (p.stamps && lane == 0)
        p.stamps[((long long)blockIdx.x * 4 + wave) * 8 + 7] =
            ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20) << 32) |
            __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
#endif
    INR_STAMP(0);
    // block-local SRDs: base at the tile's first element, extent to the end of the operand
    const long long a_first = A_KC ? ((long long)m0 * p.lda + k_begin) : ((long long)k_begin * p.lda + m0);
    const long long b_first = B_KC ? ((long long)n0 * p.ldb + k_begin) : ((long long)k_begin * p.ldb + n0);
    // extent: what this block may touch, never past the end of the operand (reads beyond return 0)
    const long long a_span = A_KC ? (long long)BM * p.lda : (long long)(k_end - k_begin) * p.lda;
    const long long b_span = B_KC ? (long long)BN * p.ldb : (long long)(k_end - k_begin) * p.ldb;
    const __amdgpu_buffer_rsrc_t srdA = make_srd(p.A + a_first, min(p.a_elems - a_first, a_span) * 4);
    const __amdgpu_buffer_rsrc_t srdB = make_srd(p.B + b_first, min(p.b_elems - b_first, b_span) * 4);
    const int a_step = (A_KC ? BK : BK * p.lda) * 4;  // bytes per K-tile
    const int b_step = (B_KC ? BK : BK * p.ldb) * 4;
    int va[4], vb[4];
    tile_voffsets<A_KC>(va, p.lda, tid);
    tile_voffsets<B_KC>(vb, p.ldb, tid);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = wm * 64 + l32, brow = wn * 64 + l32;
    f32x4 ra[4], rb[4];
    Frags f0, f1;

    if (ktiles > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = buf_load4(srdA, va[i], 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = buf_load4(srdB, vb[i], 0);
        store_tile<A_KC>(smem, ra, tid);
        store_tile<B_KC>(smem + BOFF, rb, tid);
    }
    __syncthreads();
    if (ktiles > 0) read_frags<A_KC, B_KC>(f0, smem, smem + BOFF, arow, brow, 0, h);

    // Issue priority: a wave in its MFMA stream outranks a co-resident wave that is in its (VALU-dense)
    // epilogue.  Arbitration is priority-then-age; without this an OLDER wave's epilogue starves the younger
    // wave's MFMA issue and the matrix pipe idles for the length of every epilogue.
    INR_STAMP(1);
    int a_off = 0, b_off = 0;
    for (int t = 0; t + 1 < ktiles; ++t) {
        const float* cA = smem + (t & 1) * STAGE;
        const float* cB = cA + BOFF;
        float* nA = smem + ((t & 1) ^ 1) * STAGE;
        float* nB = nA + BOFF;
        a_off += a_step;
        b_off += b_step;
        // k block 0: MFMAs on f0; prefetch fragments of k block 1; issue the next tile's global loads
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = buf_load4(srdA, va[i], a_off);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = buf_load4(srdB, vb[i], b_off);
        read_frags<A_KC, B_KC>(f1, cA, cB, arow, brow, 1, h);
        mfma_block(acc, f0);
        sched_interleave<NR, 8, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
        // k block 1
        read_frags<A_KC, B_KC>(f0, cA, cB, arow, brow, 2, h);
        mfma_block(acc, f1);
        sched_interleave<NR, 0, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
        // k block 2: also park the next tile in the other LDS buffer
        read_frags<A_KC, B_KC>(f1, cA, cB, arow, brow, 3, h);
        store_tile<A_KC>(nA, ra, tid);
        store_tile<B_KC>(nB, rb, tid);
        mfma_block(acc, f0);
        sched_interleave<NR, 8, 0x200, 0>();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // k block 3: its operands are already in registers; prefetch k block 0 of the next tile
        read_frags<A_KC, B_KC>(f0, nA, nB, arow, brow, 0, h);
        mfma_block(acc, f1);
        sched_interleave<NR, 0, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
    }
    const EpiAddr ea = epi_addr<EPI>(p, m0, n0, wm, wn, lane, tc.split);
    f32x4 mulreg[16];
    if (EPI == EPI_MUL) {  // the element-wise factor of the epilogue is fetched under the last tile's MFMAs
#pragma unroll
        for (int q = 0; q < 16; ++q) mulreg[q] = buf_load4(ea.srdMul, ea.voff, q * ea.row_step);
    }
    if (ktiles > 0) {  // last tile: nothing left to prefetch
        const float* cA = smem + ((ktiles - 1) & 1) * STAGE;
        const float* cB = cA + BOFF;
        read_frags<A_KC, B_KC>(f1, cA, cB, arow, brow, 1, h);
        mfma_block(acc, f0);
        sched_interleave<NR, 0, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
        read_frags<A_KC, B_KC>(f0, cA, cB, arow, brow, 2, h);
        mfma_block(acc, f1);
        sched_interleave<NR, 0, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
        read_frags<A_KC, B_KC>(f1, cA, cB, arow, brow, 3, h);
        mfma_block(acc, f0);
        sched_interleave<NR, 0, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(acc, f1);
    }

    INR_STAMP(2);
    __syncthreads();  // every wave is done with the operand tiles: LDS becomes the epilogue staging area
    INR_STAMP(3);
    epilogue_staged<EPI>(p, acc, smem + wave * 4096, ea, mulreg, n0, wn, lane, tc.tile_m * 2 + wm);
    INR_STAMP(4);
}

// =====================================================================================================
// pipelined kernel on v_mfma_f32_16x16x4_f32  (same tiles, same staging; 32-cycle MFMAs)
// =====================================================================================================
// A register-only loop of the 16x16x4 form runs 6 % faster than 32x32x2 on this part (147-150 vs 138-142 TFLOP/s,
// tools/mfma_rate.hip) and offers VALU/LDS instructions an issue slot every 32 instead of every 64 cycles.
// Wave tile 64x64 = 4x4 accumulators of 16x16; inside a 16-wide k block lane group g = lane/16 consumes
// k = 4g..4g+3 for both operands (again a fixed permutation of the k-sum; one b128 read feeds four MFMAs).
// r/n-contiguous LDS images get a row stride of 132 floats so the four lane groups of a ds_read_b32 hit four bank
// quarters; the epilogue stage uses a row stride of 68 for the same reason.
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int LDM16 = BM + 4;

template <bool KC>
struct TileSize16 {
    static constexpr int floats = KC ? BM * LDK : BK * LDM16;
};

template <bool KCONTIG>
__device__ __forceinline__ void store_tile16(float* __restrict__ S, const f32x4 (&reg)[4], int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (KCONTIG) {
            const int r = (tid >> 3) + 32 * i, k = (tid & 7) * 4;
            *reinterpret_cast<f32x4*>(S + r * LDK + k) = reg[i];
        } else {
            const int k = (tid >> 5) + 8 * i, r = (tid & 31) * 4;
            *reinterpret_cast<f32x4*>(S + k * LDM16 + r) = reg[i];
        }
    }
}

struct Frags16 {
    f32x4 a[4], b[4];
};

// fragment of 4 k-values (k = kb*16 + 4g + 0..3) for row `r`
template <bool KCONTIG>
__device__ __forceinline__ f32x4 read_frag16(const float* __restrict__ S, int r, int kb, int g) {
    if (KCONTIG) {
        return *reinterpret_cast<const f32x4*>(S + r * LDK + kb * 16 + 4 * g);
    } else {
        const float* p = S + (kb * 16 + 4 * g) * LDM16 + r;
        f32x4 v;
        v[0] = p[0];
        v[1] = p[LDM16];
        v[2] = p[2 * LDM16];
        v[3] = p[3 * LDM16];
        return v;
    }
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void read_frags16(Frags16& f, const float* __restrict__ sA, const float* __restrict__ sB,
                                             int arow, int brow, int kb, int g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = read_frag16<A_KC>(sA, arow + i * 16, kb, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) f.b[j] = read_frag16<B_KC>(sB, brow + j * 16, kb, g);
}

// MFMAs of k sub-steps [S0, S1) of one 16-wide k block
template <int S0, int S1>
__device__ __forceinline__ void mfma_block16(f32x4v (&acc)[4][4], const Frags16& f) {
#pragma unroll
    for (int s = S0; s < S1; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i][s], f.b[j][s], acc[i][j], 0, 0, 0);
}

// NM MFMAs with NR LDS reads and NX other memory instructions (mask XMASK) spread evenly between them
template <int NM, int NR, int NX, int XMASK, int I>
__device__ __forceinline__ void sched_interleave_n() {
    if constexpr (I < NM) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int r = (I + 1) * NR / NM - I * NR / NM;
        if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, r, 0);
        constexpr int x = (I + 1) * NX / NM - I * NX / NM;
        if constexpr (x > 0) __builtin_amdgcn_sched_group_barrier(XMASK, x, 0);
        sched_interleave_n<NM, NR, NX, XMASK, I + 1>();
    }
}

constexpr int SUB16 = 68;                       // epilogue stage row stride (floats)
constexpr int STAGE16_FLOATS = 4 * 64 * SUB16;  // four waves x 64 rows

template <bool A_KC, bool B_KC, int EPI>
__global__ void __launch_bounds__(NTHREADS, 2) INR_PACKED_F32 gemm_f32_pipe16_kernel(const GemmParams p) {
    constexpr int STAGE = TileSize16<A_KC>::floats + TileSize16<B_KC>::floats;
    constexpr int BOFF = TileSize16<A_KC>::floats;
    constexpr int SMEM = (2 * STAGE > STAGE16_FLOATS) ? 2 * STAGE : STAGE16_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    constexpr int NR = (A_KC ? 4 : 16) + (B_KC ? 4 : 16);  // LDS read instructions per 16-wide k block

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, l16 = lane & 15;
    const TileCoord tc = decode_block(p);
    const int m0 = tc.tile_m * BM, n0 = tc.tile_n * BN;
    const int k_begin = tc.split * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int ktiles = (k_end - k_begin + BK - 1) / BK;
    const long long a_first = A_KC ? ((long long)m0 * p.lda + k_begin) : ((long long)k_begin * p.lda + m0);
    const long long b_first = B_KC ? ((long long)n0 * p.ldb + k_begin) : ((long long)k_begin * p.ldb + n0);
    const long long a_span = A_KC ? (long long)BM * p.lda : (long long)(k_end - k_begin) * p.lda;
    const long long b_span = B_KC ? (long long)BN * p.ldb : (long long)(k_end - k_begin) * p.ldb;
    const __amdgpu_buffer_rsrc_t srdA = make_srd(p.A + a_first, min(p.a_elems - a_first, a_span) * 4);
    const __amdgpu_buffer_rsrc_t srdB = make_srd(p.B + b_first, min(p.b_elems - b_first, b_span) * 4);
    const int a_step = (A_KC ? BK : BK * p.lda) * 4;
    const int b_step = (B_KC ? BK : BK * p.ldb) * 4;
    int va[4], vb[4];
    tile_voffsets<A_KC>(va, p.lda, tid);
    tile_voffsets<B_KC>(vb, p.ldb, tid);

    f32x4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int arow = wm * 64 + l16, brow = wn * 64 + l16;
    f32x4 ra[4], rb[4];
    Frags16 f0, f1;

    if (ktiles > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = buf_load4(srdA, va[i], 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = buf_load4(srdB, vb[i], 0);
        store_tile16<A_KC>(smem, ra, tid);
        store_tile16<B_KC>(smem + BOFF, rb, tid);
    }
    __syncthreads();
    if (ktiles > 0) read_frags16<A_KC, B_KC>(f0, smem, smem + BOFF, arow, brow, 0, g);

    int a_off = 0, b_off = 0;
    for (int t = 0; t + 1 < ktiles; ++t) {
        const float* cA = smem + (t & 1) * STAGE;
        const float* cB = cA + BOFF;
        float* nA = smem + ((t & 1) ^ 1) * STAGE;
        float* nB = nA + BOFF;
        a_off += a_step;
        b_off += b_step;
        // k block 0 (64 MFMAs): prefetch the fragments of k block 1, issue the next K-tile's global loads
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = buf_load4(srdA, va[i], a_off);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = buf_load4(srdB, vb[i], b_off);
        read_frags16<A_KC, B_KC>(f1, cA, cB, arow, brow, 1, g);
        mfma_block16<0, 4>(acc, f0);
        sched_interleave_n<64, NR, 8, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
        // first half of k block 1: park the next K-tile in the other LDS buffer
        store_tile16<A_KC>(nA, ra, tid);
        store_tile16<B_KC>(nB, rb, tid);
        mfma_block16<0, 2>(acc, f1);
        sched_interleave_n<32, 0, 8, 0x200, 0>();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // second half: operands already in registers; fetch k block 0 of the next K-tile
        read_frags16<A_KC, B_KC>(f0, nA, nB, arow, brow, 0, g);
        mfma_block16<2, 4>(acc, f1);
        sched_interleave_n<32, NR, 0, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
    }
    const EpiAddr ea = epi_addr<EPI>(p, m0, n0, wm, wn, lane, tc.split);
    f32x4 mulreg[16];
    if (EPI == EPI_MUL) {
#pragma unroll
        for (int q = 0; q < 16; ++q) mulreg[q] = buf_load4(ea.srdMul, ea.voff, q * ea.row_step);
    }
    if (ktiles > 0) {
        const float* cA = smem + ((ktiles - 1) & 1) * STAGE;
        const float* cB = cA + BOFF;
        read_frags16<A_KC, B_KC>(f1, cA, cB, arow, brow, 1, g);
        mfma_block16<0, 4>(acc, f0);
        sched_interleave_n<64, NR, 0, 0x020, 0>();
        __builtin_amdgcn_sched_barrier(0);
        mfma_block16<0, 4>(acc, f1);
    }

    __syncthreads();  // every wave is done with the operand tiles: LDS becomes the epilogue staging area
    // C/D map of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4*(lane>>4) + reg
    float* sub = smem + wave * (64 * SUB16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sub[(i * 16 + 4 * g + r) * SUB16 + j * 16 + l16] = acc[i][j][r];
    epilogue_rows<EPI, SUB16>(p, sub, ea, mulreg, n0, wn, lane, tc.tile_m * 2 + wm);
}

#include "gemm_h3.inc"
#include "gemm_hp.inc"
#include "gemm_hp_nt.inc"
#include "gemm_hp_row.inc"
#include "gemm_hp_fwd.inc"

unsigned long long* g_stamps = nullptr;  // diagnostic builds only
tune_int g_stamp_class{-1}, g_stamp_nth{0}; // hp kernels: which launch receives g_stamps (class, countdown)
int gemm_build_flags() {
    int f = 0;
#ifdef INR_STAMPS
    f |= 1;
#endif
    if (H3_ABLATE != 0 || HP_ABLATE != 0) f |= 2;
    if (H3_EXTRA_LDS != 0) f |= 4;
    if (HP_A_AUX != 2 || HP_MUL_AUX != 0 || HP_RC_A_AUX != 0 || HP_RC_B_AUX != 0 || HP_HEAD_NT != 0 || HP_UNSCALE_LDEXP != 1 || HP_HEAD_PREFETCH != 1 || HP_COLSUM_TRANSPOSED != 1 || HP_DIAG_NO_OMEGA_STASH != 0) f |= 8;     // cache-policy experiments (gemm_hp.inc)
    return f;
}
static unsigned long long* hp_stamp_target(int kernel_class) {
    if (!g_stamps || kernel_class != g_stamp_class) return nullptr;
    return g_stamp_nth-- == 0 ? g_stamps : nullptr;
}
tune_int g_force_generic{0}; // tuning/debug: inr_debug_set(0, 1) routes every GEMM through the generic kernel
tune_int g_mfma16{1};     // 1 = 16x16x4 pipelined kernel (default, faster); inr_debug_set(1, 0) selects the 32x32x2 one
tune_int g_h3{1};         // split-fp16 GEMMs: 0 off, 1 on where the caller supplies scales/planes (fused fit), 2 also in
                          // the standalone layer calls (debug: planes and amax built per call in g_h3_scratch)
tune_int g_h3_wide{1};    // forward GEMMs on 128 x 256 tiles (inr_debug_set(6, 0): 128 x 128 everywhere)
char* g_h3_scratch = nullptr;   // inr_debug_set_ptr(1, ...): >= 16 MB of device memory for mode 2

// ---- host-side launch --------------------------------------------------------------------------
template <bool A_KC, bool B_KC, int EPI>
static int launch_gemm(GemmParams p, bool vec, hipStream_t stream, bool* used_fast = nullptr) {
    p.stamps = g_stamps;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const long long total = (long long)p.tiles_m * p.tiles_n * p.splits;
    INR_REQUIRE(total > 0 && total < (1ll << 31), INR_E_INVALID, "gemm grid out of range (%lld blocks)", total);
    dim3 grid((unsigned)total), block(NTHREADS);
    // fast path: 16-B aligned float4 traffic, K-tiles never straddle the contiguous axis of a k-contiguous
    // operand, and every byte offset inside one block's SRD stays below 2^31
    const long long a_span = A_KC ? (long long)BM * p.lda : (long long)p.k_per_split * p.lda;
    const long long b_span = B_KC ? (long long)BN * p.ldb : (long long)p.k_per_split * p.ldb;
    const bool fast = vec && !g_force_generic && (!A_KC || p.K % BK == 0) && (!B_KC || p.K % BK == 0) &&
                      (a_span + (long long)BK * p.lda) * 4 < (1ll << 31) &&
                      (b_span + (long long)BK * p.ldb) * 4 < (1ll << 31);
    if (used_fast) *used_fast = fast;
    if (fast && p.Bh && EPI != EPI_TANH && EPI != EPI_TANH_STASH) {
        if constexpr (A_KC && B_KC) {
            // 128 x 256 tiles (512 threads, one block per CU) pay off for the forward pass only (-4 %; input-grad +3 %)
            // ... and only when the wide grid still covers the chip twice over (small row counts keep the narrow tiles)
            if (g_h3_wide && p.N >= 256 && (EPI == EPI_SINE || EPI == EPI_SINE_STASH) &&
                (long long)p.tiles_m * ((p.N + 255) / 256) >= 512) {
                p.tiles_n = (p.N + 255) / 256;
                const dim3 wgrid((unsigned)((long long)p.tiles_m * p.tiles_n * p.splits)), wblock(512);
                if (p.a_amax)
                    hipLaunchKernelGGL((gemm_h3_kernel<H3_F32_KC, H3_SPLIT_KC, true, false, EPI, 4>), wgrid, wblock, 0, stream, p);
                else
                    hipLaunchKernelGGL((gemm_h3_kernel<H3_F32_KC, H3_SPLIT_KC, false, false, EPI, 4>), wgrid, wblock, 0, stream, p);
            } else if (p.a_amax) {
                hipLaunchKernelGGL((gemm_h3_kernel<H3_F32_KC, H3_SPLIT_KC, true, false, EPI, 2>), grid, block, 0, stream, p);
            } else {
                hipLaunchKernelGGL((gemm_h3_kernel<H3_F32_KC, H3_SPLIT_KC, false, false, EPI, 2>), grid, block, 0, stream, p);
            }
            INR_LAUNCH_CHECK();
            count_launch(LF_H3);
            return 0;
        }
    }
    if constexpr (!A_KC && !B_KC && EPI == EPI_PLAIN) {
        if (fast && p.a_amax) {
            if (p.b_amax)
                hipLaunchKernelGGL((gemm_h3_kernel<H3_F32_RC, H3_F32_RC, true, true, EPI, 2>), grid, block, 0, stream, p);
            else
                hipLaunchKernelGGL((gemm_h3_kernel<H3_F32_RC, H3_F32_RC, true, false, EPI, 2>), grid, block, 0, stream, p);
            INR_LAUNCH_CHECK();
            count_launch(LF_H3);
            return 0;
        }
    }
    if (fast && g_mfma16)
        hipLaunchKernelGGL((gemm_f32_pipe16_kernel<A_KC, B_KC, EPI>), grid, block, 0, stream, p);
    else if (fast)
        hipLaunchKernelGGL((gemm_f32_pipe_kernel<A_KC, B_KC, EPI>), grid, block, 0, stream, p);
    else if (vec)
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, EPI, true>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, EPI, false>), grid, block, 0, stream, p);
    INR_LAUNCH_CHECK();
    count_launch(fast ? (g_mfma16 ? LF_F32_PIPE16 : LF_F32_PIPE) : LF_F32_GENERIC);
    return 0;
}

static inline bool vec_ok(const void* a, const void* b, int lda, int ldb, int ka, int kb) {
    return aligned16(a) && aligned16(b) && (lda % 4 == 0) && (ldb % 4 == 0) && (ka % 4 == 0) && (kb % 4 == 0);
}

// act[n][out] = sin(omega*(x[n][in] W[out][in]^T + b)), optional dact = omega*cos(...)
// ---- split-fp16 support ---------------------------------------------------------------------------
// layout of a weight-plane set for one [out][in] matrix inside a caller-provided region: hi, lo, hiT, loT (halves)
size_t h3_planes_bytes(long long weights) { return (size_t)weights * 4 * sizeof(_Float16); }

int h3_weight_split(const float* const* W, const int* out_f, const int* in_f, int layers, _Float16* planes,
                    unsigned* amax, unsigned* zero_slots, int n_zero, hipStream_t stream) {
    INR_REQUIRE(layers >= 1 && layers <= 8, INR_E_INVALID, "h3_weight_split: %d layers", layers);
    WeightSplitJobs jobs{};
    _Float16* cur = planes;
    for (int l = 0; l < layers; ++l) {
        const long long n = (long long)out_f[l] * in_f[l];
        jobs.job[l] = WeightSplitJob{W[l], cur, cur + n, cur + 2 * n, cur + 3 * n, amax + l, out_f[l], in_f[l]};
        cur += 4 * n;
    }
    if (n_zero > 0) INR_HIP(hipMemsetAsync(zero_slots, 0, sizeof(unsigned) * n_zero, stream));
    int max_tiles = 1;
    for (int l = 0; l < layers; ++l) {
        const int t = ((out_f[l] + 63) / 64) * ((in_f[l] + 63) / 64);
        if (t > max_tiles) max_tiles = t;
    }
    ProfScope ps(KC_OTHER, stream);
    hipLaunchKernelGGL(weight_amax_kernel, dim3(32, layers), dim3(256), 0, stream, jobs);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(weight_split_kernel, dim3(max_tiles, layers), dim3(256), 0, stream, jobs);
    INR_LAUNCH_CHECK();
    return 0;
}

// floor_bits: the slot starts from this value (float bits).  The network input uses 1.0f: any tensor within [-1, 1]
// (Fourier features) then gets the same scale whatever its actual maximum, so a voxel's value cannot depend on which
// rows happen to share its re-sampling chunk.
int h3_tensor_amax(unsigned* out, const float* x, long long n, hipStream_t stream, unsigned floor_bits) {
    INR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(out), (int)floor_bits, 1, stream));
    ProfScope ps(KC_OTHER, stream);
    hipLaunchKernelGGL(tensor_amax_kernel, dim3(1024), dim3(256), 0, stream, out, x, n);
    INR_LAUNCH_CHECK();
    return 0;
}

static inline bool h3_shape_ok(int in_f, int out_f) { return in_f % BK == 0 && out_f % 8 == 0 && in_f % 8 == 0; }

// debug mode 2: build planes / amax for a standalone layer call inside g_h3_scratch
static int h3_debug_prepare(H3Args& h, const float* W, int out_f, int in_f, bool transposed, const float* scaled_a,
                            long long a_elems, hipStream_t stream) {
    unsigned* slots = reinterpret_cast<unsigned*>(g_h3_scratch);
    if (W) {
        _Float16* planes = reinterpret_cast<_Float16*>(g_h3_scratch + 256);
        if (int rc = h3_weight_split(&W, &out_f, &in_f, 1, planes, slots, slots, 1, stream)) return rc;
        const long long n = (long long)out_f * in_f;
        h.Bh = planes + (transposed ? 2 * n : 0);
        h.Bl = planes + (transposed ? 3 * n : n);
        h.b_amax = slots;
    }
    if (scaled_a) {
        if (int rc = h3_tensor_amax(slots + 1, scaled_a, a_elems, stream, 0u)) return rc;
        h.a_amax = slots + 1;
    }
    return 0;
}

static inline void h3_apply(GemmParams& p, const H3Args* h) {
    if (!h) return;
    p.a_amax = h->a_amax; p.b_amax = h->b_amax; p.Bh = h->Bh; p.Bl = h->Bl; p.amax_out = h->amax_out;
    p.reverse_m = h->reverse_m;
}

int gemm_sine_forward(float* act, float* dact, const float* x, const float* W, const float* b, int64_t n,
                      int in_f, int out_f, float omega, hipStream_t stream, const H3Args* h3) {
    H3Args dbg;
    if (!h3 && g_h3 == 2 && g_h3_scratch && h3_shape_ok(in_f, out_f) && (long long)in_f * out_f <= (1 << 20)) {
        if (int rc = h3_debug_prepare(dbg, W, out_f, in_f, false, nullptr, 0, stream)) return rc;
        h3 = &dbg;
    }
    GemmParams p{};
    h3_apply(p, h3);
    p.A = x; p.B = W; p.C = act; p.C2 = dact; p.bias = b; p.mul = nullptr;
    p.M = (int)n; p.N = out_f; p.K = in_f;
    p.lda = in_f; p.ldb = in_f; p.ldc = out_f;
    p.omega = omega;
    p.splits = 1;
    p.k_per_split = (in_f + BK - 1) / BK * BK;
    p.slab_stride = 0;
    p.a_elems = (long long)n * in_f; p.b_elems = (long long)out_f * in_f; p.c_elems = (long long)n * out_f;
    const bool vec = vec_ok(x, W, in_f, in_f, in_f, in_f);
    ProfScope ps(KC_GEMM_FWD, stream);
    if (dact) return launch_gemm<true, true, EPI_SINE_STASH>(p, vec, stream);
    return launch_gemm<true, true, EPI_SINE>(p, vec, stream);
}

// act[n][out] = scale*tanh(x W^T + b), optional dact = scale*(1 - tanh^2)   (PerturbNet hidden layer)
int gemm_tanh_forward(float* act, float* dact, const float* x, const float* W, const float* b, int64_t n, int in_f,
                      int out_f, float scale, hipStream_t stream) {
    GemmParams p{};
    p.A = x; p.B = W; p.C = act; p.C2 = dact; p.bias = b; p.mul = nullptr;
    p.M = (int)n; p.N = out_f; p.K = in_f;
    p.lda = in_f; p.ldb = in_f; p.ldc = out_f;
    p.omega = scale;
    p.splits = 1;
    p.k_per_split = (in_f + BK - 1) / BK * BK;
    p.slab_stride = 0;
    p.a_elems = (long long)n * in_f; p.b_elems = (long long)out_f * in_f; p.c_elems = (long long)n * out_f;
    const bool vec = vec_ok(x, W, in_f, in_f, in_f, in_f);
    ProfScope ps(KC_OTHER, stream);
    if (dact) return launch_gemm<true, true, EPI_TANH_STASH>(p, vec, stream);
    return launch_gemm<true, true, EPI_TANH>(p, vec, stream);
}

// dz_prev[n][in] = (dz[n][out] @ W[out][in]) * mul[n][in]   (mul nullable)
// colsum_slab (nullable, needs mul): receives [*slab_rows][in] partial column sums of dz_prev when the fast
// kernel ran (*slab_rows = 2*ceil(n/128)); *slab_rows = 0 means the caller must run its own column sum.
int input_grad_colsum_rows(int64_t n) { return 2 * (int)((n + BM - 1) / BM); }

int gemm_input_grad(float* dz_prev, const float* dz, const float* W, const float* mul, int64_t n, int in_f,
                    int out_f, float* colsum_slab, int* slab_rows, hipStream_t stream, const H3Args* h3) {
    H3Args dbg;
    if (!h3 && g_h3 == 2 && g_h3_scratch && h3_shape_ok(out_f, in_f) && (long long)in_f * out_f <= (1 << 20)) {
        if (int rc = h3_debug_prepare(dbg, W, out_f, in_f, true, dz, (long long)n * out_f, stream)) return rc;
        h3 = &dbg;
    }
    GemmParams p{};
    h3_apply(p, h3);
    // with planes, B = W^T [in][out]: k-contiguous, leading dimension out_f
    p.A = dz; p.B = W; p.C = dz_prev; p.C2 = nullptr; p.bias = nullptr; p.mul = mul;
    p.M = (int)n; p.N = in_f; p.K = out_f;
    p.lda = out_f; p.ldb = p.Bh ? out_f : in_f; p.ldc = in_f;
    p.omega = 0.f;
    p.splits = 1;
    p.k_per_split = (out_f + BK - 1) / BK * BK;
    p.slab_stride = 0;
    p.a_elems = (long long)n * out_f; p.b_elems = (long long)out_f * in_f; p.c_elems = (long long)n * in_f;
    // A k-contig: needs K%4; B n-contig: needs N%4 (a float4 runs along n)
    const bool vec = vec_ok(dz, W, out_f, in_f, out_f, in_f);
    ProfScope ps(KC_GEMM_DX, stream);
    if (slab_rows) *slab_rows = 0;
    if (mul) {
        p.colsum = (in_f % 4 == 0) ? colsum_slab : nullptr;
        bool fast = false;
        const int rc = p.Bh ? launch_gemm<true, true, EPI_MUL>(p, vec, stream, &fast)
                            : launch_gemm<true, false, EPI_MUL>(p, vec, stream, &fast);
        if (rc == 0 && fast && p.colsum && slab_rows) *slab_rows = input_grad_colsum_rows(n);
        return rc;
    }
    if (p.Bh) return launch_gemm<true, true, EPI_PLAIN>(p, vec, stream);
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
    // every split's row range must stay inside a 2 GiB operand window (wide layers at millions of rows)
    const long long widest = out_f > in_f ? out_f : in_f;
    const long long rows_max = ((1ll << 31) - 1) / (widest * 4) / BK * BK;
    const long long need = rows_max > 0 ? (n + rows_max - 1) / rows_max : (1ll << 30);
    if (need > s) s = need;
    if (s > (1 << 20)) s = 1 << 20;     // (the caller's range check rejects what still does not fit)
    return (int)s;
}

// slabs[splits][out][in] = partial dz^T x over row ranges
int gemm_param_grad_slabs(float* slabs, int splits, const float* dz, const float* x, int64_t n, int in_f,
                          int out_f, hipStream_t stream, const H3Args* h3) {
    H3Args dbg;
    if (!h3 && g_h3 == 2 && g_h3_scratch) {
        if (int rc = h3_debug_prepare(dbg, nullptr, 0, 0, false, dz, (long long)n * out_f, stream)) return rc;
        h3 = &dbg;
    }
    GemmParams p{};
    h3_apply(p, h3);
    p.A = dz; p.B = x; p.C = slabs; p.C2 = nullptr; p.bias = nullptr; p.mul = nullptr;
    p.M = out_f; p.N = in_f; p.K = (int)n;
    p.lda = out_f; p.ldb = in_f; p.ldc = in_f;
    p.omega = 0.f;
    p.splits = splits;
    const long long ksteps = (n + BK - 1) / BK;
    p.k_per_split = (int)((ksteps + splits - 1) / splits) * BK;
    // a block reads its row range through a 32-bit buffer window: the range must stay below 2 GiB in either operand
    INR_REQUIRE((long long)p.k_per_split * (out_f > in_f ? out_f : in_f) * 4 < (1ll << 31), INR_E_INVALID,
                "param-grad: %lld rows x %d columns per split do not fit a 2 GiB operand window (more splits needed)",
                (long long)p.k_per_split, out_f > in_f ? out_f : in_f);
    p.slab_stride = (long long)out_f * in_f;
    p.a_elems = (long long)n * out_f; p.b_elems = (long long)n * in_f; p.c_elems = (long long)splits * out_f * in_f;
    const bool vec = vec_ok(dz, x, out_f, in_f, out_f, in_f);
    ProfScope ps(KC_GEMM_DW, stream);
    return launch_gemm<false, false, EPI_PLAIN>(p, vec, stream);
}


// =====================================================================================================
// host side of the pre-split path (gemm_hp.inc)
// =====================================================================================================
tune_int g_hp_stagger{0};  // inr_debug_set(11, n): start phases of the persistent blocks, n * 64 cycles apart (0 = together)
tune_int g_hp_persistent{2}; // inr_debug_set(10, v): 2 persistent walk with the epilogue of tile T under the K-loop of tile T+1
                           // (K = 256 / 512), 1 persistent walk with the epilogue in line, 0 one block per tile
static int hp_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
bool hp_head_ok(int hidden) { return hidden == 128 || hidden == 256 || hidden == 512 || hidden == 1024; }

// per-step weight preparation (gemm_hp.inc): slots[l] = max|W_l|, slots[8 + l] = 0 (dz maxima), slots[16 + l] = wnorm_l; `part`
// = 8 x HP_PREP_MAXB x 2 words of scratch; head_bound nullable (forward-only callers)
size_t hp_prep_part_bytes() { return (size_t)8 * HP_PREP_MAXB * 2 * sizeof(unsigned); }
int hp_weight_prep(const float* const* W, const int* out_f, const int* in_f, int layers, char* planes, unsigned* slots,
                   unsigned* part, float* head_bound, const float* head_W, const float* head_b, int hidden, const unsigned* tmax,
                   const unsigned* wtmax, float inv_count, float omega, hipStream_t stream, const float* const* bias,
                   const float* layer_omega, float* act_bound, const unsigned* x_amax) {
    INR_REQUIRE(layers >= 1 && layers <= 8, INR_E_INVALID, "hp_weight_prep: %d layers", layers);
    HpWeightJobs jobs{};
    char* cur = planes;
    int max_tiles = 1, max_sb = 1;
    for (int l = 0; l < layers; ++l) {
        const long long n = (long long)out_f[l] * in_f[l];
        jobs.job[l] = HpWeightJob{W[l], cur, cur + 4 * n, slots + l, slots + 16 + l, out_f[l], in_f[l], bias ? bias[l] : nullptr,
                                  layer_omega ? layer_omega[l] : 0.f};
        cur += 8 * n;
        const int t = ((out_f[l] + 63) / 64) * ((in_f[l] + 63) / 64);
        if (t > max_tiles) max_tiles = t;
        const int sb = (in_f[l] >> 4) < HP_PREP_MAXB ? (in_f[l] >> 4) : HP_PREP_MAXB;
        if (sb > max_sb) max_sb = sb;
    }
    jobs.layers = layers;
    jobs.part = part;
    jobs.dz_slots = slots + 8;
    jobs.head_bound = head_bound;
    jobs.head_W = head_W; jobs.head_b = head_b; jobs.hidden = hidden;
    jobs.tmax = tmax; jobs.wtmax = wtmax; jobs.inv_count = inv_count; jobs.omega = omega;
    jobs.act_bound = act_bound; jobs.x_amax = x_amax;
    ProfScope ps(KC_OTHER, stream);
    hipLaunchKernelGGL(hp_weight_stats_kernel, dim3(max_sb, layers), dim3(256), 0, stream, jobs);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(hp_weight_split_kernel, dim3(max_tiles, layers + 1), dim3(256), 0, stream, jobs);
    INR_LAUNCH_CHECK();
    return 0;
}

int hp_convert(char* out, const float* x, long long rows, int cols, HpScale sc, hipStream_t stream) {
    if (rows <= 0) return 0;
    const long long n8 = rows * (cols / 8);
    long long blocks = (n8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(KC_OTHER, stream);
    hipLaunchKernelGGL(hp_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, x, rows, cols, sc);
    INR_LAUNCH_CHECK();
    return 0;
}

int hp_unconvert(float* out, const char* x, long long rows, int cols, HpScale sc, hipStream_t stream) {
    if (rows <= 0) return 0;
    long long blocks = (rows * cols + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(hp_unconvert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, x, rows, cols, sc);
    INR_LAUNCH_CHECK();
    return 0;
}

static int hp_check_grid(const HpParams& p) {
    const long long total = (long long)p.tiles_m * p.tiles_n * p.splits;
    INR_REQUIRE(total > 0 && total < (1ll << 31), INR_E_INVALID, "hp gemm grid out of range (%lld blocks)", total);
    return 0;
}

// act (HL32 [n][out_f]) = sin(omega (x W^T + b)); dact (fp32, nullable) = omega cos(.)
// may the last sine layer of a fit step stash z only (HPE_Z)?  (deferred-epilogue kernel shapes; debug key 16)
tune_int g_hp_zhead{1};
tune_int g_hp_fused_fwd{0};   // inr_debug_set(19, 1): inference forwards of eligible networks run all layers in one launch (gemm_hp_fwd.inc:
                              // measured SLOWER than the layer-wise launches at hidden = 512 -- 117 against 144 M voxels/s -- so off by default)
tune_int g_hp_narrow_max_tiles{192};   // inr_debug_set(29, v): most wide tiles (per 256 CUs) of a launch that still goes to the narrow kernel
tune_int g_hp_narrow{1};    // inr_debug_set(18, v): 1 = launches with fewer 128-row tiles than two per CU take 64-row tiles (default), 0 = never
// The row-owning kernel (gemm_hp_row.inc): K-contiguous launches of 512 output columns whose 128-row panels number at least
// g_hp_row_min_tiles.  OFF by default (inr_debug_set(27, 1) selects it, key 28 moves the threshold; bit-identical results either way):
// its K-loop is 10 % cheaper than the 128 x 256 shape's (tools/kloop_probe.hip V4 / V0: 0.631 against 0.702 ms at the package cap) but
// it has no registers for a second accumulator set, and the epilogue it therefore runs in line costs 0.19 ms per launch with the
// matrix pipe idle: forward 0.815 against 0.753 ms, input gradient 0.931 against 0.842, step 8.85 against 8.46 ms at 128^3
// (profiles/r05_headline_ab.txt).
tune_int g_hp_row{0};
tune_int g_hp_row_min_tiles{1024};
// (only the launches the deferred-epilogue kernel serves -- forward K = 256 / 512, input gradient K = 512: that kernel's MFMA block order
//  is the one this kernel reproduces bit for bit; the in-line kernel that takes the other K runs its blocks in another order)
static bool hp_row_ok(int64_t rows, int n_cols, int k, bool forward) {
    return g_hp_row && g_hp_persistent == 2 && n_cols == HR_BN && (k == 512 || (forward && k == 256)) &&
           (rows + HP_BM - 1) / HP_BM >= g_hp_row_min_tiles;
}
bool hp_z_stash_ok(int in_f) { return g_hp_zhead && g_hp_persistent == 2 && (in_f == 512 || in_f == 256); }

// Which rows of a K-contiguous GEMM go to the wide persistent kernels and which to the 64 x 128 tiles of gemm_hp_nt_kernel:
// the whole launch goes narrow when its wide tiles would keep at most three quarters of the CUs busy (four narrow tiles per
// wide one, each a little more than a quarter of its time: gemm_hp_nt.inc), otherwise all of it stays wide.
struct HpRowPlan {
    int64_t wide_rows, narrow_rows;
};
static HpRowPlan hp_row_plan(int64_t n, int width) {
    if (!g_hp_narrow || !g_hp_persistent) return {n, 0};
    const long long G = hp_num_cus();
    const long long tiles_n = (width + HP_BN - 1) / HP_BN, tiles_m = (n + HP_BM - 1) / HP_BM, tiles = tiles_m * tiles_n;
    if (256 * tiles <= (long long)g_hp_narrow_max_tiles * G) return {0, n};   // (default 192 per 256 CUs: three quarters of the chip)
    // (Handing the remainder rows of a LARGE launch to the narrow tiles was measured and dropped: these kernels move their
    //  bytes at ~3.6 TB/s whatever the tile count, so the round the remainder adds to a few CUs costs ~9 us at 69,632 rows,
    //  less than a second launch: 1.39 against 1.34 ms per step.)
    return {n, 0};
}

// xzy: the MFMA block order of the wide kernel that serves this K (gemm_hp_nt.inc)
template <int EPI>
static int hp_launch_narrow(HpParams p, int64_t row0, int64_t rows, bool xzy, hipStream_t stream) {
    p.A += row0 * p.pitchA;
    p.a_rows = rows;
    p.M = (int)rows;
    if (p.C_hl) p.C_hl += row0 * (long long)p.N * 4;
    if (p.C2) p.C2 += row0 * (long long)p.N;
    if (p.mul) p.mul += row0 * (long long)p.N;
    if (p.colsum) p.colsum += 2 * (row0 / HP_BM) * (long long)p.N;   // (row0 is a multiple of the wide tile height)
    p.tiles_m = (int)((rows + NT_BM - 1) / NT_BM);
    p.tiles_n = (p.N + NT_BN - 1) / NT_BN;
    p.splits = 1;
    const long long tiles = (long long)p.tiles_m * p.tiles_n;
    INR_REQUIRE(tiles > 0 && tiles < (1ll << 31), INR_E_INVALID, "hp narrow gemm grid out of range (%lld blocks)", tiles);
    if (xzy) hipLaunchKernelGGL((gemm_hp_nt_kernel<EPI, true>), dim3((unsigned)tiles), dim3(NT_NTH), 0, stream, p);
    else hipLaunchKernelGGL((gemm_hp_nt_kernel<EPI, false>), dim3((unsigned)tiles), dim3(NT_NTH), 0, stream, p);
    INR_LAUNCH_CHECK();
    count_launch(LF_HP_NARROW);
    return 0;
}

int hp_sine_forward(char* act_hl, float* dact, const char* x_hl, const char* W_hl, const float* bias, int64_t n, int in_f,
                    int out_f, float omega, HpScale sa, HpScale sb, int reverse_m, hipStream_t stream, bool z_only, HpScale so) {
    HpParams p{};
    p.A = x_hl; p.B = W_hl;
    p.M = (int)n; p.N = out_f; p.K = in_f;
    p.pitchA = (long long)in_f * 4; p.pitchB = (long long)in_f * 4;
    p.a_rows = n; p.b_rows = out_f;
    p.sa = sa; p.sb = sb; p.so = so;
    p.C_hl = act_hl; p.C2 = dact; p.bias = bias; p.omega = omega;
    p.k_per_split = in_f; p.reverse_m = reverse_m; p.stagger = g_hp_stagger; p.splits = 1;
    if (z_only) INR_REQUIRE(dact && hp_z_stash_ok(in_f), INR_E_INVALID, "hp_sine_forward: z-only stash needs the deferred-epilogue kernel");
    const HpRowPlan plan = hp_row_plan(n, out_f);
    p.stamps = hp_stamp_target(KC_GEMM_FWD);
    ProfScope ps(KC_GEMM_FWD, stream);
    if (plan.wide_rows > 0) {
        p.M = (int)plan.wide_rows;
        p.a_rows = plan.wide_rows;
        p.tiles_m = (int)((plan.wide_rows + HP_BM - 1) / HP_BM); p.tiles_n = (out_f + HP_BN - 1) / HP_BN;
        if (int rc = hp_check_grid(p)) return rc;
        const long long tiles = (long long)p.tiles_m * p.tiles_n;
        const dim3 grid((unsigned)tiles), block(HP_NTH);
        const dim3 pgrid((unsigned)(tiles < hp_num_cus() ? tiles : hp_num_cus()));
        if (hp_row_ok(plan.wide_rows, out_f, in_f, true)) {   // one block per 128 rows x all 512 columns (gemm_hp_row.inc)
            p.tiles_n = 1;
            const dim3 rgrid((unsigned)(p.tiles_m < hp_num_cus() ? p.tiles_m : hp_num_cus()));
            if (z_only) hipLaunchKernelGGL((gemm_hp_row_kernel<HPE_Z>), rgrid, block, 0, stream, p);
            else if (dact) hipLaunchKernelGGL((gemm_hp_row_kernel<HPE_SINE_STASH>), rgrid, block, 0, stream, p);
            else hipLaunchKernelGGL((gemm_hp_row_kernel<HPE_SINE>), rgrid, block, 0, stream, p);
            count_launch(LF_HP_ROW);
        } else if (z_only) {   // z + b as fp32 into `dact`, nothing else
            if (in_f == 512) hipLaunchKernelGGL((gemm_hp_pkd_kernel<HPE_Z, 16>), pgrid, block, 0, stream, p);
            else hipLaunchKernelGGL((gemm_hp_pkd_kernel<HPE_Z, 8>), pgrid, block, 0, stream, p);
            count_launch(LF_HP_PKD);
        } else if (g_hp_persistent == 2 && (in_f == 512 || in_f == 256)) {   // epilogue of tile T under the K-loop of tile T+1
            if (in_f == 512) {
                if (dact) hipLaunchKernelGGL((gemm_hp_pkd_kernel<HPE_SINE_STASH, 16>), pgrid, block, 0, stream, p);
                else hipLaunchKernelGGL((gemm_hp_pkd_kernel<HPE_SINE, 16>), pgrid, block, 0, stream, p);
            } else {
                if (dact) hipLaunchKernelGGL((gemm_hp_pkd_kernel<HPE_SINE_STASH, 8>), pgrid, block, 0, stream, p);
                else hipLaunchKernelGGL((gemm_hp_pkd_kernel<HPE_SINE, 8>), pgrid, block, 0, stream, p);
            }
            count_launch(LF_HP_PKD);
        } else if (g_hp_persistent && in_f >= 3 * HP_BK) {
            if (dact) hipLaunchKernelGGL((gemm_hp_pkc_kernel<HPE_SINE_STASH>), pgrid, block, 0, stream, p);
            else hipLaunchKernelGGL((gemm_hp_pkc_kernel<HPE_SINE>), pgrid, block, 0, stream, p);
            count_launch(LF_HP_PKC);
        } else if (dact) {
            hipLaunchKernelGGL((gemm_hp_kernel<HP_KC, HPE_SINE_STASH>), grid, block, 0, stream, p);
            count_launch(LF_HP_TILE);
        } else {
            hipLaunchKernelGGL((gemm_hp_kernel<HP_KC, HPE_SINE>), grid, block, 0, stream, p);
            count_launch(LF_HP_TILE);
        }
        INR_LAUNCH_CHECK();
    }
    if (plan.narrow_rows > 0) {
        if (plan.wide_rows > 0) p.stamps = nullptr;      // (one stamp buffer: the wide launch has it when there is one)
        const bool xzy = g_hp_persistent == 2 && (in_f == 512 || in_f == 256);   // the rule of the wide dispatch above
        p.fold_bias = (g_hp_persistent && in_f >= 3 * HP_BK) ? 1 : 0;
        if (z_only) return hp_launch_narrow<HPE_Z>(p, plan.wide_rows, plan.narrow_rows, xzy, stream);
        if (dact) return hp_launch_narrow<HPE_SINE_STASH>(p, plan.wide_rows, plan.narrow_rows, xzy, stream);
        return hp_launch_narrow<HPE_SINE>(p, plan.wide_rows, plan.narrow_rows, xzy, stream);
    }
    return 0;
}

// The last sine layer of a fit step with the head step in its epilogue (gemm_hp_row_kernel<HPE_HEAD>, gemm_hp_row.inc): dz_L (HL32, scale
// `dz_so`) over the bytes of `dact`, per 64-row half panel one row of slab_b (column sums of dz_L = the layer's bias gradient), slab_w
// (sum_n g_n sin(.) = the head's weight gradient), part_loss and part_g; max|dz_L| into `amax_out`.  2 * ceil(n / 128) slab rows.
tune_int g_hp_row_head{1};              // inr_debug_set(30, 0): never (the z-only layer + hp_head_step_kernel instead)
tune_int g_hp_row_head_min_tiles{768};  // inr_debug_set(31, v): fewest 128-row panels of a launch that takes the fused form (98,304 rows:
                                        // measured break-even at ~65-70 k rows, -1.6 % at 98 k, -2.2 % at 139 k, -2.4 % at 524 k: profiles/r05_head_fuse_sweep.txt)
bool hp_row_head_ok(int64_t n, int hidden, int in_f) {
    return g_hp_row_head && g_hp_persistent == 2 && hidden == HR_BN && (in_f == 512 || in_f == 256) &&
           (n + HP_BM - 1) / HP_BM >= g_hp_row_head_min_tiles && (n + HP_BM - 1) / HP_BM < (1ll << 30);
}
int hp_row_head_rows(int64_t n) { return 2 * (int)((n + HP_BM - 1) / HP_BM); }
int hp_sine_forward_head(char* dz_hl, const char* x_hl, const char* W_hl, const float* bias, int64_t n, int in_f, int out_f, float omega,
                         HpScale sa, HpScale sb, HpScale dz_so, const float* head_w, const float* head_b, const float* target,
                         const float* weight, int64_t count_total, float* slab_b, float* slab_w, float* part_loss, float* part_g,
                         unsigned* amax_out, hipStream_t stream) {
    INR_REQUIRE(hp_row_head_ok(n, out_f, in_f), INR_E_INVALID, "hp_sine_forward_head: shape not served (n = %lld, %d -> %d)", (long long)n, in_f, out_f);
    HpParams p{};
    p.A = x_hl; p.B = W_hl;
    p.M = (int)n; p.N = out_f; p.K = in_f;
    p.pitchA = (long long)in_f * 4; p.pitchB = (long long)in_f * 4;
    p.a_rows = n; p.b_rows = out_f;
    p.sa = sa; p.sb = sb; p.so = dz_so;
    p.C_hl = dz_hl; p.bias = bias; p.omega = omega;
    p.k_per_split = in_f; p.stagger = g_hp_stagger; p.splits = 1;
    p.tiles_m = (int)((n + HP_BM - 1) / HP_BM); p.tiles_n = 1;
    p.colsum = slab_b; p.slab_w = slab_w; p.part_loss = part_loss; p.part_g = part_g; p.amax_out = amax_out;
    p.head_w = head_w; p.head_b = head_b; p.target = target; p.tweight = weight;
    p.inv_count = (float)(1.0 / (double)(count_total > 0 ? count_total : n));
    p.stamps = hp_stamp_target(KC_GEMM_FWD);
    ProfScope ps(KC_GEMM_FWD, stream);
    const dim3 rgrid((unsigned)(p.tiles_m < hp_num_cus() ? p.tiles_m : hp_num_cus())), block(HP_NTH);
    hipLaunchKernelGGL((gemm_hp_row_kernel<HPE_HEAD>), rgrid, block, 0, stream, p);
    INR_LAUNCH_CHECK();
    count_launch(LF_HP_ROW);
    return 0;
}

// rows of column sums an input-grad launch may write (two per 64-row tile is the finest any of the kernels goes)
int hp_input_grad_max_rows(int64_t n) { return 2 * (int)((n + 63) / 64); }

int hp_input_grad(char* dzprev_hl, const char* dz_hl, const char* WT_hl, const float* mul, int64_t n, int in_f, int out_f,
                  float* colsum_slab, int* colsum_rows, unsigned* amax_out, HpScale sa, HpScale sb, HpScale so,
                  hipStream_t stream) {
    HpParams p{};
    p.A = dz_hl; p.B = WT_hl;
    p.M = (int)n; p.N = in_f; p.K = out_f;
    p.pitchA = (long long)out_f * 4; p.pitchB = (long long)out_f * 4;
    p.a_rows = n; p.b_rows = in_f;
    p.sa = sa; p.sb = sb; p.so = so;
    p.C_hl = dzprev_hl; p.mul = mul; p.colsum = colsum_slab; p.amax_out = amax_out;
    p.splits = 1;
    p.k_per_split = out_f; p.stagger = g_hp_stagger;
    const HpRowPlan plan = hp_row_plan(n, in_f);
    *colsum_rows = 2 * (int)((plan.wide_rows + HP_BM - 1) / HP_BM) + 2 * (int)((plan.narrow_rows + NT_BM - 1) / NT_BM);
    p.stamps = hp_stamp_target(KC_GEMM_DX);
    ProfScope ps(KC_GEMM_DX, stream);
    if (plan.wide_rows > 0) {
        p.M = (int)plan.wide_rows;
        p.a_rows = plan.wide_rows;
        p.tiles_m = (int)((plan.wide_rows + HP_BM - 1) / HP_BM); p.tiles_n = (in_f + HP_BN - 1) / HP_BN;
        if (int rc = hp_check_grid(p)) return rc;
        const long long tiles = (long long)p.tiles_m * p.tiles_n;
        const dim3 grid((unsigned)tiles), block(HP_NTH);
        const dim3 pgrid((unsigned)(tiles < hp_num_cus() ? tiles : hp_num_cus()));
        if (hp_row_ok(plan.wide_rows, in_f, out_f, false)) {
            p.tiles_n = 1;
            const dim3 rgrid((unsigned)(p.tiles_m < hp_num_cus() ? p.tiles_m : hp_num_cus()));
            hipLaunchKernelGGL((gemm_hp_row_kernel<HPE_MUL>), rgrid, block, 0, stream, p);
            count_launch(LF_HP_ROW);
        } else if (g_hp_persistent == 2 && out_f == 512) {   // (K = 256 would spill: the in-line epilogue serves it)
            hipLaunchKernelGGL((gemm_hp_pkd_kernel<HPE_MUL, 16>), pgrid, block, 0, stream, p);
            count_launch(LF_HP_PKD);
        } else if (g_hp_persistent && out_f >= 3 * HP_BK) {
            hipLaunchKernelGGL((gemm_hp_pkc_kernel<HPE_MUL>), pgrid, block, 0, stream, p);
            count_launch(LF_HP_PKC);
        } else {
            hipLaunchKernelGGL((gemm_hp_kernel<HP_KC, HPE_MUL>), grid, block, 0, stream, p);
            count_launch(LF_HP_TILE);
        }
        INR_LAUNCH_CHECK();
    }
    if (plan.narrow_rows > 0) {
        if (plan.wide_rows > 0) p.stamps = nullptr;
        return hp_launch_narrow<HPE_MUL>(p, plan.wide_rows, plan.narrow_rows, g_hp_persistent == 2 && out_f == 512, stream);
    }
    return 0;
}

// ---- grid -> Fourier features -> HL32 in one kernel (dense re-sampling) --------------------------------------------------------
// inr_siren_reconstruct built its network input per chunk in four passes: fourier_kernel (1 KB per voxel written at 256
// features), tensor_amax (1 KB read), hp_convert (1 KB read, 1 KB written), then layer 0 reads the HL32 image.  Features are
// sines and cosines, so max|x| <= 1 and the input scale (floor 1.0) is 2^14 whatever the data: here one thread computes
// eight frequencies of a row exactly as fourier_kernel does (same grid rule, same fma order, same sincos) and writes their
// sin and cos octets as HL32 directly -- 1 KB written, 1 KB read per voxel, bit-identical operands.  Needs m % 32 == 0.
struct HlGrid {
    int dim;
    long long n[8];
};
__global__ void __launch_bounds__(256) INR_PACKED_F32 grid_fourier_hl_kernel(char* __restrict__ out, HlGrid g, long long row_begin, long long n_rows,
                                                              const float* __restrict__ B, int m, unsigned* __restrict__ x_amax) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) *x_amax = 0x3f800000u;                       // max(1, max|x|) = 1: what tensor_amax would have found
    const int per_row = m >> 3;
    if (t >= n_rows * per_row) return;
    const long long row = t / per_row;
    const int j0 = (int)(t - row * per_row) * 8;
    const float two_pi = 6.283185307179586f;
    long long rem = row_begin + row;
    float c[8];
#pragma unroll
    for (int a = 7; a >= 0; --a) {
        if (a < g.dim) {
            const long long idx = rem % g.n[a];
            rem /= g.n[a];
            c[a] = linspace_pm1(idx, g.n[a]);
        }
    }
    float sv[8], cv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float proj = 0.f;
#pragma unroll
        for (int a = 0; a < 8; ++a)
            if (a < g.dim) proj = fmaf(two_pi * c[a], B[(j0 + e) * g.dim + a], proj);
        sincos_f32(proj, sv[e], cv[e]);
    }
    const float s = h3_pow2(14);
    u32x4 hi, lo;
    const int cols = 2 * m;
    hp_split8(sv, s, hi, lo);
    char* dst = out + hp_off(row, j0, cols);
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 64) = lo;
    hp_split8(cv, s, hi, lo);
    dst = out + hp_off(row, m + j0, cols);
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 64) = lo;
}

bool hp_grid_fourier_ok(int m, int dim) { return m >= 32 && m % 32 == 0 && dim >= 1 && dim <= 8; }
int hp_grid_fourier_hl(char* x_hl, unsigned* x_amax, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows, const float* B,
                       int m, hipStream_t stream) {
    INR_REQUIRE(hp_grid_fourier_ok(m, dim), INR_E_INVALID, "hp_grid_fourier_hl: m = %d, dim = %d", m, dim);
    if (n_rows == 0) return 0;
    HlGrid g{};
    g.dim = dim;
    for (int a = 0; a < 8; ++a) g.n[a] = a < dim ? shape[a] : 1;
    const long long work = n_rows * (m >> 3);
    ProfScope ps(KC_OTHER, stream);
    hipLaunchKernelGGL(grid_fourier_hl_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, x_hl, g, (long long)row_begin,
                       (long long)n_rows, B, m, x_amax);
    INR_LAUNCH_CHECK();
    return 0;
}

// ---- cross-layer fused forward (gemm_hp_fwd.inc) ---------------------------------------------------------------------------
bool hp_fused_forward_ok(int in_f, int hidden, int n_sine) {
    return g_hp_fused_fwd && (hidden == 512 || hidden == 256) && in_f % 32 == 0 && in_f >= 32 && in_f <= hidden && n_sine >= 1 &&
           n_sine <= FW_MAX_LAYERS;
}
// y [n] = head(sine layers(x)); x_hl: HL32 image of the network input; W_hl[l] / bias[l] / w_amax[l] per sine layer
int hp_fused_forward(float* y, const char* x_hl, const unsigned* x_amax, int64_t n, int in_f, int hidden, int n_sine,
                     const char* const* W_hl, const float* const* bias, const unsigned* const* w_amax, float first_omega,
                     float hidden_omega, const float* head_W, const float* head_b, int use_clamp, float clamp_min,
                     hipStream_t stream) {
    INR_REQUIRE(hp_fused_forward_ok(in_f, hidden, n_sine), INR_E_INVALID, "hp_fused_forward: unsupported network");
    FwParams p{};
    for (int l = 0; l < n_sine; ++l) {
        p.layer[l].W = W_hl[l];
        p.layer[l].bias = bias[l];
        p.layer[l].w_amax = w_amax[l];
        p.layer[l].K = l == 0 ? in_f : hidden;
        p.layer[l].omega = l == 0 ? first_omega : hidden_omega;
    }
    p.n_sine = n_sine; p.H = hidden;
    p.x = x_hl; p.x_amax = x_amax; p.n = n;
    p.head_W = head_W; p.head_b = head_b; p.y = y;
    p.use_clamp = use_clamp; p.clamp_min = clamp_min;
    const long long panels = (n + FW_ROWS - 1) / FW_ROWS;
    const dim3 grid((unsigned)(panels < hp_num_cus() ? panels : hp_num_cus())), block(FW_NTH);
    ProfScope ps(KC_GEMM_FWD, stream);
    if (hidden == 512) hipLaunchKernelGGL((siren_fwd_fused_kernel<4>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((siren_fwd_fused_kernel<2>), grid, block, 0, stream, p);
    INR_LAUNCH_CHECK();
    count_launch(LF_HP_FUSED_FWD);
    return 0;
}

int hp_param_grad_splits(int64_t n, int in_f, int out_f) {
    const long long tiles = (long long)((out_f + HP_BM - 1) / HP_BM) * ((in_f + HP_BN - 1) / HP_BN);
    const long long ksteps = (n + HP_BK - 1) / HP_BK;
    long long want = (256 + tiles - 1) / tiles;          // one 512-thread block per CU
    const long long max_by_work = (ksteps + 7) / 8;      // at least 8 K-tiles per split
    long long s = want < max_by_work ? want : max_by_work;
    const long long min_by_offset = (n * (long long)(in_f > out_f ? in_f : out_f) * 4 + (1ll << 30) - 1) >> 30;   // 32-bit offsets
    if (s < min_by_offset) s = min_by_offset;
    // fp32 accumulation over one split's rows is a plain running sum: beyond ~16k rows its rounding error (relative to a
    // gradient that is a small mean of large terms) shows at the 1e-5 tier (256^3 volume: 8.8e-5 with 131,072 rows per
    // split) -- keep the register accumulation short and let the fixed-order slab reduction do the rest
    const long long min_by_len = (n + 16383) / 16384;
    if (s < min_by_len) s = min_by_len;
    if (s < 1) s = 1;
    if (s > 4096) s = 4096;
    return (int)s;
}

// slabs[splits][out_f][in_f] = partial dz^T x over row ranges (dz, x: HL32)
static int hp_param_grad_params(HpParams& p, float* slabs, int splits, const char* dz_hl, const char* x_hl, int64_t n, int in_f,
                                int out_f, HpScale sa, HpScale sb) {
    p = HpParams{};
    p.A = dz_hl; p.B = x_hl;
    p.M = out_f; p.N = in_f; p.K = (int)n;
    p.pitchA = (long long)out_f * 4; p.pitchB = (long long)in_f * 4;
    p.a_rows = n; p.b_rows = n;
    p.sa = sa; p.sb = sb;
    p.C2 = slabs;
    p.tiles_m = (out_f + HP_BM - 1) / HP_BM; p.tiles_n = (in_f + HP_BN - 1) / HP_BN; p.splits = splits;
    const long long ksteps = (n + HP_BK - 1) / HP_BK;
    p.k_per_split = (int)((ksteps + splits - 1) / splits) * HP_BK;
    p.slab_stride = (long long)out_f * in_f;
    INR_REQUIRE((long long)p.k_per_split * (p.pitchA > p.pitchB ? p.pitchA : p.pitchB) < (1ll << 31), INR_E_INVALID,
                "hp_param_grad_slabs: row range per split too large for 32-bit offsets");
    return hp_check_grid(p);
}
int hp_param_grad_slabs(float* slabs, int splits, const char* dz_hl, const char* x_hl, int64_t n, int in_f, int out_f,
                        HpScale sa, HpScale sb, hipStream_t stream) {
    HpParams p;
    if (int rc = hp_param_grad_params(p, slabs, splits, dz_hl, x_hl, n, in_f, out_f, sa, sb)) return rc;
    const dim3 grid((unsigned)((long long)p.tiles_m * p.tiles_n * p.splits)), block(HP_NTH);
    p.stamps = hp_stamp_target(KC_GEMM_DW);
    ProfScope ps(KC_GEMM_DW, stream);
    hipLaunchKernelGGL((gemm_hp_kernel<HP_RC, HPE_SLAB>), grid, block, 0, stream, p);
    INR_LAUNCH_CHECK();
    count_launch(LF_HP_RC);
    return 0;
}

// the same GEMMs, `jobs` of them (<= hp_param_grad_multi_max()) in one launch; every job counts as one launch of its family
int hp_param_grad_multi_max() { return HP_MULTI_MAX; }
int hp_param_grad_multi(const HpParamGradJob* jobs, int njobs, int64_t n, hipStream_t stream) {
    INR_REQUIRE(njobs >= 1 && njobs <= HP_MULTI_MAX, INR_E_INVALID, "hp_param_grad_multi: %d jobs (1 .. %d)", njobs, HP_MULTI_MAX);
    HpMultiParams m{};
    long long blocks = 0;
    for (int j = 0; j < njobs; ++j) {
        if (int rc = hp_param_grad_params(m.p[j], jobs[j].slabs, jobs[j].splits, jobs[j].dz_hl, jobs[j].x_hl, n, jobs[j].in_f,
                                          jobs[j].out_f, jobs[j].sa, jobs[j].sb))
            return rc;
        m.first[j] = (int)blocks;
        blocks += (long long)m.p[j].tiles_m * m.p[j].tiles_n * m.p[j].splits;
    }
    m.first[njobs] = (int)blocks;
    m.jobs = njobs;
    INR_REQUIRE(blocks < (1ll << 30), INR_E_INVALID, "hp_param_grad_multi: %lld blocks", blocks);
    ProfScope ps(KC_GEMM_DW, stream);
    hipLaunchKernelGGL(gemm_hp_rc_multi_kernel, dim3((unsigned)blocks), dim3(HP_NTH), 0, stream, m);
    INR_LAUNCH_CHECK();
    for (int j = 0; j < njobs; ++j) count_launch(LF_HP_RC);
    return 0;
}

int hp_head_forward(float* y, const char* a_hl, const float* W, const float* bias, int64_t n, int hidden, int use_clamp,
                    float clamp_min, hipStream_t stream, bool from_z, float omega, HpScale sa) {
    long long blocks = (n + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    const dim3 grid((unsigned)blocks), block(256);
    ProfScope ps(KC_OTHER, stream);
    if (from_z) {
        switch (hidden) {
            case 128: hipLaunchKernelGGL((hp_head_forward_kernel<2, true>), grid, block, 0, stream, y, a_hl, W, bias, n, use_clamp, clamp_min, omega, sa); break;
            case 256: hipLaunchKernelGGL((hp_head_forward_kernel<4, true>), grid, block, 0, stream, y, a_hl, W, bias, n, use_clamp, clamp_min, omega, sa); break;
            case 512: hipLaunchKernelGGL((hp_head_forward_kernel<8, true>), grid, block, 0, stream, y, a_hl, W, bias, n, use_clamp, clamp_min, omega, sa); break;
            case 1024: hipLaunchKernelGGL((hp_head_forward_kernel<16, true>), grid, block, 0, stream, y, a_hl, W, bias, n, use_clamp, clamp_min, omega, sa); break;
            default: INR_REQUIRE(false, INR_E_INVALID, "hp_head_forward: hidden = %d", hidden);
        }
        INR_LAUNCH_CHECK();
        return 0;
    }
    switch (hidden) {
        case 128: hipLaunchKernelGGL((hp_head_forward_kernel<2, false>), grid, block, 0, stream, y, a_hl, W, bias, n, use_clamp, clamp_min, 0.f, sa); break;
        case 256: hipLaunchKernelGGL((hp_head_forward_kernel<4, false>), grid, block, 0, stream, y, a_hl, W, bias, n, use_clamp, clamp_min, 0.f, sa); break;
        case 512: hipLaunchKernelGGL((hp_head_forward_kernel<8, false>), grid, block, 0, stream, y, a_hl, W, bias, n, use_clamp, clamp_min, 0.f, sa); break;
        case 1024: hipLaunchKernelGGL((hp_head_forward_kernel<16, false>), grid, block, 0, stream, y, a_hl, W, bias, n, use_clamp, clamp_min, 0.f, sa); break;
        default: INR_REQUIRE(false, INR_E_INVALID, "hp_head_forward: hidden = %d", hidden);
    }
    INR_LAUNCH_CHECK();
    return 0;
}

// bound of the head's dz for an external dL/dy (the autograd path): gmax = slot holding max|g|
int hp_head_bound_ext(float* head_bound, const unsigned* gmax, const float* head_W, int hidden, float omega, hipStream_t stream) {
    ProfScope ps(KC_OTHER, stream);
    hipLaunchKernelGGL(hp_head_bound_ext_kernel, dim3(1), dim3(256), 0, stream, head_bound, gmax, head_W, hidden, omega);
    INR_LAUNCH_CHECK();
    return 0;
}

// rows per block of the head step: at most 128, fewer when that would leave CUs idle (round 2: 16 blocks at 4,096 rows, 49 us for
// 8 MB).  128, not the 256 of rounds 1-3: at 78 VGPRs six blocks share a CU, and 2,048 blocks on 1,536 slots are 1.33 rounds --
// 4,096 blocks leave a shorter tail (profiles/r04_nt_ab.txt, box 8: 0.542 ms per step outside the GEMMs against 0.551; 64 rows
// 0.546, 344 rows = one block per slot 0.576)
tune_int g_hp_head_min_rows{16};   // inr_debug_set(21, .): fewest rows a block of the head step takes (a multiple of 4)
tune_int g_hp_head_rows{0};        // inr_debug_set(23, .): rows per block of the head step, 0 = the rule below
int hp_head_rows_per_block(int64_t n) {
    if (g_hp_head_rows > 0) return (g_hp_head_rows + 3) / 4 * 4;
    long long r = (n + 1023) / 1024;
    r = (r + 3) / 4 * 4;
    const int lo = g_hp_head_min_rows;
    return (int)(r < lo ? lo : (r > 128 ? 128 : r));
}
int64_t hp_head_blocks(int64_t n) {
    const int rpb = hp_head_rows_per_block(n);
    return (n + rpb - 1) / rpb;
}

// slab_b / slab_w: [blocks][hidden], part_loss / part_g: [blocks], blocks = hp_head_blocks(n)
int hp_head_step(char* dz_hl, float* slab_b, float* slab_w, float* part_loss, float* part_g, const char* a_hl,
                 const float* dact, const float* W, const float* bias, const float* t, const float* wgt, int64_t n, int hidden,
                 int64_t count_total, unsigned* amax_out, HpScale so, hipStream_t stream, bool from_z, float omega,
                 const float* g_ext, HpScale sa) {
    const float inv = (float)(1.0 / (double)(count_total > 0 ? count_total : n));
    const int rpb = hp_head_rows_per_block(n);
    const dim3 grid((unsigned)((n + rpb - 1) / rpb)), block(256);
    ProfScope ps(KC_OTHER, stream);
#define HP_HEAD_STEP(CPL)                                                                                                   \
    do {                                                                                                                    \
        if (from_z && g_ext)                                                                                                \
            hipLaunchKernelGGL((hp_head_step_kernel<CPL, true, true>), grid, block, 0, stream, dz_hl, slab_b, slab_w,        \
                               part_loss, part_g, a_hl, dact, W, bias, t, wgt, n, inv, amax_out, so, rpb, omega, g_ext, sa);  \
        else if (from_z)                                                                                                    \
            hipLaunchKernelGGL((hp_head_step_kernel<CPL, true, false>), grid, block, 0, stream, dz_hl, slab_b, slab_w,       \
                               part_loss, part_g, a_hl, dact, W, bias, t, wgt, n, inv, amax_out, so, rpb, omega, g_ext, sa);  \
        else if (g_ext)                                                                                                     \
            hipLaunchKernelGGL((hp_head_step_kernel<CPL, false, true>), grid, block, 0, stream, dz_hl, slab_b, slab_w,       \
                               part_loss, part_g, a_hl, dact, W, bias, t, wgt, n, inv, amax_out, so, rpb, omega, g_ext, sa);  \
        else                                                                                                                \
            hipLaunchKernelGGL((hp_head_step_kernel<CPL, false, false>), grid, block, 0, stream, dz_hl, slab_b, slab_w,      \
                               part_loss, part_g, a_hl, dact, W, bias, t, wgt, n, inv, amax_out, so, rpb, omega, g_ext, sa);  \
    } while (0)
    switch (hidden) {
        case 128: HP_HEAD_STEP(2); break;
        case 256: HP_HEAD_STEP(4); break;
        case 512: HP_HEAD_STEP(8); break;
        case 1024: HP_HEAD_STEP(16); break;
        default: INR_REQUIRE(false, INR_E_INVALID, "hp_head_step: hidden = %d", hidden);
    }
#undef HP_HEAD_STEP
    INR_LAUNCH_CHECK();
    return 0;
}

}  // namespace inr
