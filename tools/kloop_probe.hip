// K-loop probe for the dominant GEMM (gemm_hp_pkd_kernel, csrc/gemm_hp.inc): the SAME K-loop -- HL32 operand images in a
// three-stage LDS ring (A 16 KB + B 32 KB per stage, the kernel's swizzle), fragments by ds_read_b128, three fp16 MFMA products
// per fp32 product (hi x lo, hi x hi, lo x hi), one s_barrier per K-tile, optionally the ring refilled by buffer_load ... lds
// (A panel streamed from a 1 GiB buffer = HBM, B panel from a 1 MiB buffer = L2) -- in the shapes verdict r03 item 1 asks about:
//   V0  8 waves (2 x 4) of 64 x 64,  v_mfma_f32_16x16x32_f16   (the product kernel's shape)
//   V1  8 waves (2 x 4) of 64 x 64,  v_mfma_f32_32x32x16_f16   (half the MFMA issues and operand-register reads per FLOP)
//   V2  4 waves (2 x 2) of 64 x 128, v_mfma_f32_16x16x32_f16   (LDS fragment bytes per K-tile 128 KB -> 96 KB; 512-VGPR budget)
//   V3  4 waves (2 x 2) of 64 x 128, v_mfma_f32_32x32x16_f16
// No epilogue: accumulators are summed into one word per lane at the end.  Every variant runs for --secs seconds of back-to-back
// launches on RANDOM operands (zeros raise the clock: MI355X_MICROARCH.md 'DVFS give-back'); reported per variant: wall time per
// K-tile and block, executed TFLOP/s, and the in-kernel shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (median over
// blocks of the last launch).  Rounds are interleaved (cdna_hip_programming.md rule 24).  Sample rocm-smi beside it
// (tools/power_trace.sh) for package power.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kloop_probe tools/kloop_probe.hip
//   ./tools/kloop_probe [--secs 12] [--rounds 2] [--dma 0|1]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <type_traits>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int A_IMG = 16384, B_IMG = 32768, STAGE = A_IMG + B_IMG, NSTAGE = 3;

__device__ __forceinline__ int hp_f(int r) {
    const int p = (r >> 1) & 7;
    return (p & 1) ^ (((p >> 1) & 1) * 5) ^ (((p >> 2) & 1) * 7);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t srd(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

struct Args {
    const char* A;       // [tiles][128 rows][K bytes-per-row = ktiles * 128]  (HL32 lines), streamed
    const char* B;       // [256 rows][ktiles * 128]
    float* out;
    unsigned long long* clk;   // [blocks][4]: memtime0, realtime0, memtime1, realtime1
    int ktiles;          // K-tiles per "tile"
    int tiles;           // tiles per block
    int dma;
    long long a_tile_bytes;
    int share;           // V0..V3: 1 = two consecutive 128 x 256 tiles read the SAME A panel (the product: the two column tiles of a row
                         // panel; the second read comes from L2), 0 = every tile streams its own panel from HBM
};

// WAVES: 8 or 4.  M32: 32x32x16 MFMA.
template <int WAVES, bool M32>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) kloop(const Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];
    constexpr int WN = WAVES == 8 ? 4 : 2;          // waves along the 256 columns
    constexpr int NB = 256 / WN / 16;               // 16-column B fragments per wave: 4 or 8
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // ---- fill the ring once from global (random HL32 lines), whatever the refill mode
    for (int i = tid; i < NSTAGE * STAGE / 16; i += WAVES * 64)
        reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(p.B)[i % (256 * 128 / 16 * 4)];
    __syncthreads();

    // ---- LDS-DMA bookkeeping: a K-tile is 48 transfers of 1 KB (8 rows x 128 B); wave w issues transfers w, w + WAVES, ...
    constexpr int NDMA = 48 / WAVES;
    const int drow = (lane >> 3), dslot = ((lane & 7) ^ hp_f(drow)) * 16;   // (row & 7 does not touch (r >> 1) & 7's upper bits here:
                                                                           //  f depends on bits 1..3 of the row = bits of drow)
    const __amdgpu_buffer_rsrc_t srdB = srd(p.B, 256u * (unsigned)p.ktiles * 128u);
    auto dma_tile = [&](int stage, const __amdgpu_buffer_rsrc_t& sa, int kt) {
        char* S = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int q = wave + WAVES * i;          // 0..15: A rows 8q.., 16..47: B rows 8(q-16)..
            const bool isA = q < 16;
            const int row0 = 8 * (isA ? q : q - 16);
            const int voff = (row0 + drow) * p.ktiles * 128 + (((lane & 7) ^ hp_f(row0 + drow)) * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? sa : srdB, (__attribute__((address_space(3))) void*)(S + q * 1024), 16, voff,
                                                     kt * 128, 0, 0);
        }
    };
    (void)dslot;

    // ---- fragment addresses
    int fa, fb;
    if (!M32) {
        const int g = lane >> 4, l16 = lane & 15;
        const int fx = (g ^ hp_f(l16)) << 4;
        fa = (64 * wm + l16) * 128 + fx;
        fb = A_IMG + ((256 / WN) * wn + l16) * 128 + fx;
    } else {
        const int h = lane >> 5, l32 = lane & 31;
        const int fx = (h ^ hp_f(l32)) << 4;          // k-step s adds slot 2 s: (2 s + h) ^ f = (h ^ f) ^ 2 s
        fa = (64 * wm + l32) * 128 + fx;
        fb = A_IMG + ((256 / WN) * wn + l32) * 128 + fx;
    }

    f32x4 acc16[M32 ? 1 : 4][M32 ? 1 : NB];
    f32x16 acc32[M32 ? 2 : 1][M32 ? NB / 2 : 1];
    if (!M32) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB / 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
    }

    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int total = p.tiles * p.ktiles;     // (even)
    auto a_srd = [&](int t) {
        return srd(p.A + ((long long)blockIdx.x * p.tiles + ((t / p.ktiles) >> p.share)) * p.a_tile_bytes, (unsigned)p.a_tile_bytes);
    };
    // fragment register sets (two: K-tile t computes on one while t+1's fragments land in the other, as in the product kernel)
    half8 ah[2][4], al[2][4], bh[2][NB], bl[2][NB];      // 16x16x32: [set][fragment]; 32x32x16: [set][2 * block + k-step]
    auto read_frags = [&](const char* S, auto setc) {
        constexpr int Q = decltype(setc)::value;
        if (!M32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[Q][i] = *reinterpret_cast<const half8*>(S + fa + i * 2048);
                al[Q][i] = *reinterpret_cast<const half8*>(S + (fa ^ 64) + i * 2048);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                bh[Q][j] = *reinterpret_cast<const half8*>(S + fb + j * 2048);
                bl[Q][j] = *reinterpret_cast<const half8*>(S + (fb ^ 64) + j * 2048);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    ah[Q][2 * i + s] = *reinterpret_cast<const half8*>(S + (fa ^ (32 * s)) + i * 4096);
                    al[Q][2 * i + s] = *reinterpret_cast<const half8*>(S + (fa ^ (32 * s) ^ 64) + i * 4096);
                }
#pragma unroll
            for (int j = 0; j < NB / 2; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bh[Q][2 * j + s] = *reinterpret_cast<const half8*>(S + (fb ^ (32 * s)) + j * 4096);
                    bl[Q][2 * j + s] = *reinterpret_cast<const half8*>(S + (fb ^ (32 * s) ^ 64) + j * 4096);
                }
        }
    };
    auto compute = [&](auto setc) {
        constexpr int Q = decltype(setc)::value;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            if (!M32) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 2 ? al[Q][i] : ah[Q][i], pr == 0 ? bl[Q][j] : bh[Q][j],
                                                                             acc16[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NB / 2; ++j)
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? al[Q][2 * i + s] : ah[Q][2 * i + s],
                                                                                 pr == 0 ? bl[Q][2 * j + s] : bh[Q][2 * j + s], acc32[i][j], 0, 0, 0);
            }
        }
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    constexpr int W2 = ((2 * NDMA) & 0xF) | (((2 * NDMA) >> 4) << 14), W1 = (NDMA & 0xF) | ((NDMA >> 4) << 14);
    if (p.dma) {
        dma_tile(0, a_srd(0), 0);
        dma_tile(1, a_srd(1), 1 % p.ktiles);
        dma_tile(2, a_srd(2), 2 % p.ktiles);
        __builtin_amdgcn_s_waitcnt(0x0F70 | W2);
    }
    __builtin_amdgcn_s_barrier();
    read_frags(smem, Q0{});
    int stage = 0;
    // iteration t: all fragments of K-tile t are in registers; wait for K-tile t+1 (t+2 stays in flight), barrier (stage t % 3 is dead,
    // t+1 visible), refill stage t % 3 with K-tile t+3, read t+1's fragments under t's MFMAs
    auto iter = [&](int t, auto cur, auto nxt) {
        if (p.dma) {
            if (t + 2 < total) __builtin_amdgcn_s_waitcnt(0x0070 | W1);
            else __builtin_amdgcn_s_waitcnt(0x0070);
        } else {
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        __builtin_amdgcn_s_barrier();
        if (p.dma && t + 3 < total) dma_tile(stage, a_srd(t + 3), (t + 3) % p.ktiles);
        const int s1 = stage == 2 ? 0 : stage + 1;
        if (t + 1 < total) read_frags(smem + s1 * STAGE, nxt);
        compute(cur);
        stage = s1;
    };
    for (int t = 0; t < total; t += 2) {
        iter(t, Q0{}, Q1{});
        iter(t + 1, Q1{}, Q0{});
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        p.clk[blockIdx.x * 4 + 0] = t0;
        p.clk[blockIdx.x * 4 + 1] = r0;
        p.clk[blockIdx.x * 4 + 2] = t1;
        p.clk[blockIdx.x * 4 + 3] = r1;
    }
    float s = 0.f;
    if (!M32) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) s += acc16[i][j][0] + acc16[i][j][1] + acc16[i][j][2] + acc16[i][j][3];
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB / 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc32[i][j][r];
    }
    p.out[(long long)blockIdx.x * WAVES * 64 + tid] = s;
}


// V4: the ROW-OWNING shape of verdict r04 item 5 (b) -- one block owns 128 rows x all 512 columns (the head step could then run
// in the last sine layer's epilogue: y_n needs the whole row): 8 waves (2 x 4) of 64 x 128, 128 accumulator registers per lane and NO
// second register set (the epilogue would run in line), a stage = A 16 KB + B 64 KB = 80 KB, so the 160 KB LDS hold a TWO-stage ring:
// K-tile t+1 lands while t computes, its fragments can only be read behind the barrier that ends iteration t (single fragment set).
constexpr int R_A_IMG = 16384, R_B_IMG = 65536, R_STAGE = R_A_IMG + R_B_IMG;
__global__ void __launch_bounds__(512, 2) kloop_rowown(const Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * R_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    for (int i = tid; i < 2 * R_STAGE / 16; i += 512)
        reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(p.B)[i % (256 * 128 / 16 * 4)];
    __syncthreads();
    const int drow = lane >> 3;
    const __amdgpu_buffer_rsrc_t srdB = srd(p.B, 512u * (unsigned)p.ktiles * 128u);
    // a K-tile = 80 transfers of 1 KB (8 rows x 128 B): wave w issues q = w, w + 8, ... (ten each): q < 16 A rows, else B rows
    auto dma_tile = [&](int stage, const __amdgpu_buffer_rsrc_t& sa, int kt) {
        char* S = smem + stage * R_STAGE;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int q = wave + 8 * i;
            const bool isA = q < 16;
            const int row0 = 8 * (isA ? q : q - 16);
            const int voff = (row0 + drow) * p.ktiles * 128 + (((lane & 7) ^ hp_f(row0 + drow)) * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? sa : srdB, (__attribute__((address_space(3))) void*)(S + q * 1024), 16, voff,
                                                     kt * 128, 0, 0);
        }
    };
    const int g = lane >> 4, l16 = lane & 15;
    const int fx = (g ^ hp_f(l16)) << 4;
    const int fa = (64 * wm + l16) * 128 + fx;
    const int fb = R_A_IMG + (128 * wn + l16) * 128 + fx;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int total = p.tiles * p.ktiles;
    auto a_srd = [&](int t) {
        return srd(p.A + ((long long)blockIdx.x * p.tiles + t / p.ktiles) * p.a_tile_bytes, (unsigned)p.a_tile_bytes);
    };
    if (p.dma) {
        dma_tile(0, a_srd(0), 0);
        dma_tile(1, a_srd(1), 1 % p.ktiles);
        __builtin_amdgcn_s_waitcnt(0x0F70 | 10);      // vmcnt(10): K-tile 0 has landed, K-tile 1 in flight
    }
    __builtin_amdgcn_s_barrier();
    half8 ah[4], al[4], bh[8], bl[8];
    for (int t = 0; t < total; ++t) {
        const char* S = smem + (t & 1) * R_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) ah[i] = *reinterpret_cast<const half8*>(S + fa + i * 2048);
#pragma unroll
        for (int j = 0; j < 8; ++j) bl[j] = *reinterpret_cast<const half8*>(S + (fb ^ 64) + j * 2048);
#pragma unroll
        for (int j = 0; j < 8; ++j) bh[j] = *reinterpret_cast<const half8*>(S + fb + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) al[i] = *reinterpret_cast<const half8*>(S + (fa ^ 64) + i * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        // K-tile t+1 has landed (this wave's transfers), everybody is done with stage t & 1: refill it with K-tile t+2
        if (p.dma) __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        if (p.dma && t + 2 < total) dma_tile(t & 1, a_srd(t + 2), (t + 2) % p.ktiles);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        p.clk[blockIdx.x * 4 + 0] = t0;
        p.clk[blockIdx.x * 4 + 1] = r0;
        p.clk[blockIdx.x * 4 + 2] = t1;
        p.clk[blockIdx.x * 4 + 3] = r1;
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    p.out[(long long)blockIdx.x * 512 + tid] = sum;
}
static void launch_rowown(const Args& a, int blocks) {
    Args b = a;
    b.share = 0;
    b.tiles = a.tiles / 2;        // 128 x 512 tiles: half as many per block for the same FLOPs (and half the A bytes)
    hipLaunchKernelGGL(kloop_rowown, dim3(blocks), dim3(512), 0, 0, b);
}

struct Var {
    const char* name;
    void (*launch)(const Args&, int blocks);
    int waves;
    std::vector<double> ms_per_launch, ghz;
};
template <int W, bool M>
static void launch(const Args& a, int blocks) {
    hipLaunchKernelGGL((kloop<W, M>), dim3(blocks), dim3(W * 64), 0, 0, a);
}

int main(int argc, char** argv) {
    double secs = 12;
    int rounds = 2, dma = 1, only = -1, share = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--secs")) secs = atof(argv[i + 1]);
        if (!strcmp(argv[i], "--rounds")) rounds = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--dma")) dma = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--share")) share = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--only")) only = atoi(argv[i + 1]);      // V0 and this variant only
    }
    const int blocks = 256, ktiles = 16, tiles = 32;     // a launch = 32 tiles x 16 K-tiles per block (the 128^3 forward: 32 tiles per CU)
    const long long a_tile = 128ll * ktiles * 128;       // bytes of one 128-row A panel
    const size_t a_bytes = (size_t)blocks * tiles * a_tile;   // 2 GiB: streamed once per launch
    char *A, *B;
    float* out;
    unsigned long long* clk;
    hipMalloc(&A, a_bytes);
    hipMalloc(&B, 512 * ktiles * 128 + 2 * R_STAGE);
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipMalloc(&clk, blocks * 4 * 8);
    {   // random HL32 lines: hi ~ uniform in [-2^14, 2^14), lo up to half an ulp of hi
        std::mt19937 rng(1);
        std::vector<_Float16> h((512 * ktiles * 128 + 2 * R_STAGE) / 2);
        auto fill = [&](std::vector<_Float16>& v) {
            for (size_t line = 0; line + 64 <= v.size(); line += 64)
                for (int e = 0; e < 32; ++e) {
                    const float x = ((int)(rng() % 32768) - 16384) * 1.0f;
                    v[line + e] = (_Float16)x;
                    v[line + 32 + e] = (_Float16)(((int)(rng() % 2048) - 1024) * (1.0f / 4096.f) * (x == 0 ? 1.f : 8.f));
                }
        };
        fill(h);
        hipMemcpy(B, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        std::vector<_Float16> ha(64 << 20 >> 1);        // 64 MiB of random lines, replicated over A
        fill(ha);
        for (size_t off = 0; off < a_bytes; off += (64 << 20))
            hipMemcpy(A + off, ha.data(), std::min<size_t>(64 << 20, a_bytes - off), hipMemcpyHostToDevice);
    }
    Args a{A, B, out, clk, ktiles, tiles, dma, a_tile, share};
    std::vector<Var> vars = {{"V0 8w 64x64  16x16x32", launch<8, false>, 8}, {"V1 8w 64x64  32x32x16", launch<8, true>, 8},
                             {"V2 4w 64x128 16x16x32", launch<4, false>, 4}, {"V3 4w 64x128 32x32x16", launch<4, true>, 4},
                             {"V4 8w 64x128 row-owning 128x512, 2 stages", launch_rowown, 8}};
    if (only >= 0) {
        std::vector<Var> pick = {vars[0], vars[only]};
        if (only == 0) pick.pop_back();
        vars = pick;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<unsigned long long> hc(blocks * 4);
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vars) {
            const auto start = std::chrono::steady_clock::now();
            double last_ms = 0;
            int n = 0;
            printf("## %s round %d start %.0f\n", v.name, r, (double)std::chrono::duration_cast<std::chrono::seconds>(
                                                                    std::chrono::system_clock::now().time_since_epoch()).count());
            fflush(stdout);
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() < secs) {
                hipEventRecord(e0);
                for (int k = 0; k < 20; ++k) v.launch(a, blocks);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                last_ms = ms / 20;
                ++n;
            }
            hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> g;
            for (int b = 0; b < blocks; ++b) {
                const double dt = (double)(hc[b * 4 + 2] - hc[b * 4 + 0]), dr = (double)(hc[b * 4 + 3] - hc[b * 4 + 1]);
                if (dr > 0) g.push_back(dt / dr * 0.1);
            }
            std::sort(g.begin(), g.end());
            v.ms_per_launch.push_back(last_ms);
            v.ghz.push_back(g.empty() ? 0 : g[g.size() / 2]);
            const double flop = 3.0 * 2.0 * 128 * 256 * 32 * (double)ktiles * tiles * blocks;
            printf("%s round %d: %.3f ms per launch (last of %d x 20), %.0f TFLOP/s executed, %.1f ns per K-tile, in-kernel clock %.3f GHz, "
                   "%.0f cycles per K-tile\n",
                   v.name, r, last_ms, n, flop / last_ms / 1e9, last_ms * 1e6 / (ktiles * tiles), v.ghz.back(),
                   last_ms * 1e6 / (ktiles * tiles) * v.ghz.back());
            fflush(stdout);
        }
    return 0;
}
