// Fused fit step for SMALL SIRENs (the master.py regime: Siren(2, 64, 6, 1) on a 60x60 slice, 24,000 tiny
// optimizer steps per fit -- master.py:130-148).  The layer-by-layer path needs ~45 launches per step and is
// launch-bound there (364 us/step measured); here one step is TWO launches:
//   siren_small_step_kernel  -- each wave owns 32 coordinate rows and carries them through the whole network:
//       forward (all sine layers + head), residual / loss, backward (dz, per-wave partial dW/db for every layer).
//       Layer weights are staged in LDS once per layer and shared by the block's 4 waves; a wave's activations move
//       between the MFMA accumulator layout and the next GEMM's operand layout through its own LDS stage;
//       activations needed again in backward go to an L2-resident scratch.
//   small_reduce_adam_kernel -- fixed-order sum of the per-wave gradient slabs + the Adam update (+ loss).
// Same arithmetic as the big kernels (fp32 MFMA 32x32x2, k = 4h..4h+3 per 8-block, hardware sin/cos on FMA-reduced
// revolutions), so results agree with the layer-by-layer path to rounding.
// Eligibility (host): hidden in {32, 64}, in_features <= 32, out_features == 1.
#include "common.h"
#include <atomic>

namespace inr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SMALL_MAX_LAYERS = 16;   // sine layers (1 + hidden_layers)

struct SmallParams {
    const float* params;        // flat parameter buffer (inr_siren_param_offsets layout)
    float* slabs;               // [nwaves_alloc][P] per-wave partial gradients (flat layout)
    float* loss_partial;        // [nwaves_alloc]
    float* acts;                // [(S+1)][N][H]: acts[0] unused (input is x), acts[l] = input of sine layer l
    float* dacts;               // [S][N][H]: omega*cos(omega z_l)
    const float* x;             // [N][F]
    const float* target;        // [N]
    const float* weight;        // [N] or null
    long long w_off[SMALL_MAX_LAYERS + 1], b_off[SMALL_MAX_LAYERS + 1];
    long long P;                // flat parameter count (padded)
    int N, F, S;                // rows, in_features, sine layers
    float first_omega, hidden_omega, inv_count;
};

__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int H>
__global__ void __launch_bounds__(256) siren_small_step_kernel(const SmallParams p) {
    constexpr int CT = H / 32;          // 32-wide column tiles of a hidden activation
    constexpr int LDS_STRIDE = H + 4;   // [row][feature] images; 16-B aligned rows, odd number of 16-B slots
    __shared__ __attribute__((aligned(16))) float ldsW[H * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float stage[4][2][32 * LDS_STRIDE];
    __shared__ float gbuf[4][32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, l32 = lane & 31;
    const int gwave = blockIdx.x * 4 + wave;
    const int r0 = gwave * 32;
    float* stageA = stage[wave][0];     // operand image of the current layer's input a_l   [row][feature]
    float* stageD = stage[wave][1];     // operand image of dz_l                            [row][feature]
    float* slab = p.slabs + (long long)gwave * p.P;
    const long long NH = (long long)p.N * H;

    // Hidden-layer weights W_l [H][H] travel global -> registers -> ldsW [H][H+4] in two halves so the global latency
    // hides behind the previous layer's arithmetic: fetch_weights(l) is issued a layer early, commit_weights() (whole
    // block, fenced by barriers) makes them the current LDS image.
    constexpr int WQ = (H * H / 4) / 256;   // float4 per thread
    f32x4 wreg[WQ];
    auto fetch_weights = [&](int l) {
        const f32x4* W = reinterpret_cast<const f32x4*>(p.params + p.w_off[l]);
#pragma unroll
        for (int i = 0; i < WQ; ++i) wreg[i] = W[tid + 256 * i];
    };
    auto commit_weights = [&]() {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            const int f = tid + 256 * i;
            *reinterpret_cast<f32x4*>(ldsW + (f / (H / 4)) * LDS_STRIDE + (f % (H / 4)) * 4) = wreg[i];
        }
        __syncthreads();
    };

    // ------------------------------------------------ forward ------------------------------------------------
    f32x16 acc[CT], dlast[CT];
    if (p.S > 1) fetch_weights(1);
    {   // layer 0: z0 = x W0^T + b0, K = F (k = 2*kp + hh, zero beyond F); operands straight from global memory
        const float* W0 = p.params + p.w_off[0];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        for (int kp = 0; kp < (p.F + 1) / 2; ++kp) {
            const int k = 2 * kp + hh;
            const float a = (k < p.F && r0 + l32 < p.N) ? p.x[(long long)(r0 + l32) * p.F + k] : 0.f;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float b = (k < p.F) ? W0[(ct * 32 + l32) * p.F + k] : 0.f;
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ct], 0, 0, 0);
            }
        }
    }
    for (int l = 0; l < p.S; ++l) {
        if (l > 0) {   // z_l = a_l W_l^T: A from the wave's stage (k-contiguous b128), B = W_l rows (k-contiguous b128)
            commit_weights();
            if (l + 1 < p.S) fetch_weights(l + 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < H / 8; ++kb) {
                const f32x4 fa = *reinterpret_cast<const f32x4*>(stageA + l32 * LDS_STRIDE + 8 * kb + 4 * hh);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f32x4 fb = *reinterpret_cast<const f32x4*>(ldsW + (ct * 32 + l32) * LDS_STRIDE + 8 * kb + 4 * hh);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc[ct], 0, 0, 0);
                }
            }
        }
        // epilogue: a_{l+1} = sin(omega z), d_l = omega cos(omega z)   (SRDWI.py:58-59)
        const float omega = (l == 0) ? p.first_omega : p.hidden_omega;
        const float* bias = p.params + p.b_off[l];
        float* a_out = p.acts + (long long)(l + 1) * NH;
        float* d_out = p.dacts + (long long)l * NH;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int j = ct * 32 + l32;
            const float bj = bias[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = acc_row(r, hh);
                float sv, cv;
                sincos_f32(omega * (acc[ct][r] + bj), sv, cv);
                const float dv = omega * cv;
                stageA[i * LDS_STRIDE + j] = sv;           // becomes the next layer's A operand / the head's input
                dlast[ct][r] = dv;
                if (r0 + i < p.N) {
                    a_out[(long long)(r0 + i) * H + j] = sv;
                    d_out[(long long)(r0 + i) * H + j] = dv;
                }
            }
        }
    }
    // head (SRDWI.py:75-83): y = a_S . w + b -- lane = (row, half of the features), halves combined by one shuffle
    const float* wh = p.params + p.w_off[p.S];
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < H / 8; ++q) {
        const int j0 = hh * (H / 2) + 4 * q;
        const f32x4 av = *reinterpret_cast<const f32x4*>(stageA + l32 * LDS_STRIDE + j0);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wh + j0);
        part = fmaf(av[0], wv[0], part);
        part = fmaf(av[1], wv[1], part);
        part = fmaf(av[2], wv[2], part);
        part = fmaf(av[3], wv[3], part);
    }
    const float y = part + __shfl_xor(part, 32, 64) + p.params[p.b_off[p.S]];
    const int row = r0 + l32;
    const bool rvalid = row < p.N;
    const float resid = rvalid ? y - p.target[row] : 0.f;
    const float wr = (p.weight && rvalid) ? p.weight[row] * resid : resid;
    const float g = 2.0f * wr * p.inv_count;                       // dL/dy  (superresDWI.py:135, master.py:143-145)
    if (hh == 0) gbuf[wave][l32] = g;
    float lsum = (hh == 0) ? wr * resid : 0.f;
    float gsum = (hh == 0) ? g : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lsum += __shfl_xor(lsum, off, 64);
        gsum += __shfl_xor(gsum, off, 64);
    }
    if (lane == 0) {
        p.loss_partial[gwave] = lsum;
        slab[p.b_off[p.S]] = gsum;                                 // head bias gradient
    } else if (lane < 4) {
        slab[p.b_off[p.S] + lane] = 0.f;                           // 16-byte padding of the 1-float head bias
    }

    // ------------------------------------------------ backward -----------------------------------------------
    // dz_{S-1} = g (x) w_head * d_{S-1};   gW_head[j] = sum_i g_i a_S[i][j]
    f32x16 dz[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int j = ct * 32 + l32;
        const float wj = wh[j];
        float gw = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = acc_row(r, hh);
            const float gi = gbuf[wave][i];
            gw = fmaf(gi, stageA[i * LDS_STRIDE + j], gw);
            dz[ct][r] = gi * wj * dlast[ct][r];
        }
        gw += __shfl_xor(gw, 32, 64);
        if (hh == 0) slab[p.w_off[p.S] + j] = gw;
    }

    // (ldsW still holds W_{S-1} from the forward pass: the first backward layer needs no reload)
    // Order inside a layer is chosen for a single wave per SIMD (nothing else hides latency): the global reads this
    // layer needs later (its input a_l, the layer below's d_{l-1}, the next weights) are issued first, then the
    // input-grad MFMAs (LDS only) run while they are in flight, and the weight-grad contraction comes last.
    for (int l = p.S - 1; l >= 0; --l) {
        const int K = (l == 0) ? p.F : H;
        f32x4 a_pref[(32 * H / 4) / 64];
        f32x16 d_pref[CT];
        if (l > 0) {
            const float* a_in = p.acts + (long long)l * NH;
#pragma unroll
            for (int it = 0; it < (32 * H / 4) / 64; ++it) {
                const int f = it * 64 + lane, rr = f / (H / 4), c4 = (f % (H / 4)) * 4;
                a_pref[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r0 + rr < p.N) a_pref[it] = *reinterpret_cast<const f32x4*>(a_in + (long long)(r0 + rr) * H + c4);
            }
            const float* d_in = p.dacts + (long long)(l - 1) * NH;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = acc_row(r, hh);
                    d_pref[ct][r] = (r0 + i < p.N) ? d_in[(long long)(r0 + i) * H + ct * 32 + l32] : 0.f;
                }
        }
        if (l > 1) fetch_weights(l - 1);     // committed after this layer's input grad
        // bias gradient + dz into its operand image
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int j = ct * 32 + l32;
            float gb = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gb += dz[ct][r];
                stageD[acc_row(r, hh) * LDS_STRIDE + j] = dz[ct][r];
            }
            gb += __shfl_xor(gb, 32, 64);
            if (hh == 0) slab[p.b_off[l] + j] = gb;
        }
        if (l > 0) {   // da_l = dz_l W_l, then dz_{l-1} = da_l * d_{l-1}  (dz_l itself lives on in stageD for the weight grad)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < H / 8; ++kb) {
                const f32x4 fa = *reinterpret_cast<const f32x4*>(stageD + l32 * LDS_STRIDE + 8 * kb + 4 * hh);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float* wrow = ldsW + (8 * kb + 4 * hh + s) * LDS_STRIDE + l32;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], wrow[ct * 32], acc[ct], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) dz[ct][r] = acc[ct][r] * d_pref[ct][r];
            // a_l into the stage: 32 rows x H floats, one float4 per lane and pass
#pragma unroll
            for (int it = 0; it < (32 * H / 4) / 64; ++it) {
                const int f = it * 64 + lane, rr = f / (H / 4), c4 = (f % (H / 4)) * 4;
                *reinterpret_cast<f32x4*>(stageA + rr * LDS_STRIDE + c4) = a_pref[it];
            }
        }
        // weight gradient: gW_l[j][k] = sum_rows dz_l[row][j] * a_l[row][k]   (contraction over the wave's 32 rows)
        const int KT = (K + 31) / 32;
        for (int ht = 0; ht < CT; ++ht) {
            for (int kt = 0; kt < KT; ++kt) {
                f32x16 wacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
                const int kc = kt * 32 + l32;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int rr = 8 * kb + 4 * hh + s;
                        const float a = stageD[rr * LDS_STRIDE + ht * 32 + l32];
                        float b;
                        if (l > 0)
                            b = stageA[rr * LDS_STRIDE + kc];
                        else
                            b = (kc < p.F && r0 + rr < p.N) ? p.x[(long long)(r0 + rr) * p.F + kc] : 0.f;
                        wacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, wacc, 0, 0, 0);
                    }
                }
                if (kc < K) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        slab[p.w_off[l] + (long long)(ht * 32 + acc_row(r, hh)) * K + kc] = wacc[r];
                }
            }
        }
        if (l > 1) commit_weights();
    }
}

// grads[i] = sum over wave slabs (fixed order); Adam update (torch single-tensor formulation); block 0 also finishes
// the loss.  A block owns 64 consecutive parameters; its 4 waves each sum a quarter of the slabs (coalesced 256-B
// reads), the quarters are combined through LDS in a fixed order.
__global__ void __launch_bounds__(256) small_reduce_adam_kernel(float* __restrict__ params, float* __restrict__ grads,
                                                                float* __restrict__ m, float* __restrict__ v,
                                                                const float* __restrict__ slabs, int nslabs, long long P,
                                                                float one_minus_b1, float b2, float one_minus_b2,
                                                                float step_size, float bc2_sqrt, float eps,
                                                                float* __restrict__ loss_out,
                                                                const float* __restrict__ loss_partial, float inv_count) {
    __shared__ float part[4][64];
    __shared__ float red[256];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f;
    if (i < P) {
        int s = q;
        for (; s + 4 < nslabs; s += 8) {
            a0 += slabs[(long long)s * P + i];
            a1 += slabs[(long long)(s + 4) * P + i];
        }
        if (s < nslabs) a0 += slabs[(long long)s * P + i];
    }
    part[q][lane] = a0 + a1;
    __syncthreads();
    if (q == 0 && i < P) {
        const float gi = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        grads[i] = gi;
        const float mi = fmaf(gi - m[i], one_minus_b1, m[i]);
        const float vi = fmaf(one_minus_b2 * gi, gi, v[i] * b2);
        const float denom = __fsqrt_rn(vi) / bc2_sqrt + eps;
        m[i] = mi;
        v[i] = vi;
        params[i] = params[i] - step_size * (mi / denom);
    }
    if (blockIdx.x == 0 && loss_out) {
        float acc = 0.f;
        for (int k = threadIdx.x; k < nslabs; k += 256) acc += loss_partial[k];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss_out[0] = red[0] * inv_count;
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
bool small_path_ok(const inr_siren_desc_t* d, int64_t n) {
    return (d->hidden_features == 32 || d->hidden_features == 64) && d->in_features <= 32 && d->out_features == 1 &&
           d->hidden_layers + 1 <= SMALL_MAX_LAYERS && n >= 1 && n <= 32768;
}

static inline int small_blocks(int64_t n) { return (int)((n + 127) / 128); }

size_t small_workspace_floats(const inr_siren_desc_t* d, int64_t n, long long P) {
    const int S = d->hidden_layers + 1;
    const size_t nh = (size_t)n * d->hidden_features;
    return (size_t)(S + 1) * nh + (size_t)S * nh + (size_t)small_blocks(n) * 4 * (size_t)P + (size_t)small_blocks(n) * 4 + 64;
}

int small_fit_step(const inr_siren_desc_t* d, const long long* w_off, const long long* b_off, long long P, float* params,
                   float* grads, float* m, float* v, const float* x, const float* target, const float* weight, int64_t n,
                   int64_t step, double lr, double b1, double b2, double eps, float* loss_out, float* ws, hipStream_t st) {
    const int S = d->hidden_layers + 1, H = d->hidden_features;
    const int blocks = small_blocks(n), nwaves = blocks * 4;
    const size_t nh = (size_t)n * H;
    SmallParams p{};
    p.params = params;
    p.acts = ws;
    p.dacts = ws + (size_t)(S + 1) * nh;
    p.slabs = p.dacts + (size_t)S * nh;
    p.loss_partial = p.slabs + (size_t)nwaves * P;
    p.x = x; p.target = target; p.weight = weight;
    for (int l = 0; l <= S; ++l) { p.w_off[l] = w_off[l]; p.b_off[l] = b_off[l]; }
    p.P = P; p.N = (int)n; p.F = d->in_features; p.S = S;
    p.first_omega = d->first_omega; p.hidden_omega = d->hidden_omega;
    p.inv_count = (float)(1.0 / (double)n);
    ProfScope ps(KC_OTHER, st);
    if (H == 64)
        hipLaunchKernelGGL(siren_small_step_kernel<64>, dim3(blocks), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(siren_small_step_kernel<32>, dim3(blocks), dim3(256), 0, st, p);
    INR_LAUNCH_CHECK();
    count_launch(LF_SMALL_STEP);
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(small_reduce_adam_kernel, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, st, params, grads, m, v,
                       p.slabs, nwaves, P, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(lr / bc1),
                       (float)sqrt(bc2), (float)eps, loss_out, p.loss_partial, p.inv_count);
    INR_LAUNCH_CHECK();
    return 0;
}


// =====================================================================================================================
// Persistent multi-step kernel (one launch = up to SM_MAX_STEPS optimizer steps, the acquisition changing every step).
//
// master.py:137-148 takes 24,000 optimizer steps per fit, each on a 3,600-row batch whose target / weight image changes
// from step to step.  Two launches per step (above) leave the chip idle between kernels and use 113 of 1,024 SIMDs; here
// the whole grid stays resident and walks through the steps itself:
//   * a block = 8 waves carries 64 coordinate rows through the network; the 16x16 output tiles of a layer are dealt to
//     the waves (wave -> row tile wave/2, column tiles (wave%1)*CTW ..), so one layer's forward is 32 MFMAs
//     (v_mfma_f32_16x16x4_f32) per wave instead of 64 32x32x2 ones on a quarter of the waves;
//   * activations, dz and the weights live in double-buffered LDS images ([row][H+4]: b128 fragment reads without bank
//     conflicts), so a layer costs ONE block barrier; the weights of the next layer are fetched a layer ahead;
//   * the weight gradient contracts over the block's 64 rows (one slab per block, not per wave: a quarter of the slab
//     traffic); bias gradients ride on wave shuffles + a 4-way fixed-order LDS sum;
//   * after the backward pass a grid barrier (agent-scope release/acquire around one atomic counter), then every thread
//     of the grid reduces a few parameters over the block slabs in fixed order and applies Adam, a second grid barrier,
//     next step.  The launch is cooperative (the runtime refuses it unless all blocks are co-resident) and the spin has
//     an exit: a barrier that does not complete within ~2^22 polls raises the error word and every block leaves.
// Arithmetic is the layer-wise path's (fp32 MFMA, hardware sin/cos on FMA-reduced revolutions, fixed-order sums):
// results agree with it to rounding; runs are bitwise reproducible.
// =====================================================================================================================
constexpr int SM_ROWS = 64;
constexpr int SM_THREADS = 512;
constexpr int SM_MAX_STEPS = 64;
constexpr unsigned SM_SPIN_LIMIT = 1u << 22;

struct SmallMulti {
    float* params; float* grads; float* m; float* v;
    float* slabs;                 // [nblocks][P]
    float* loss_partial;          // [nblocks]
    float* acts;                  // [S][nblocks*64][H]: acts[l] = input of sine layer l (l = 1..S-1)
    float* dacts;                 // [S][nblocks][8 waves][CTW][64 lanes] float4, accumulator-native
    const float* x;               // [N][F]
    const float* targets;         // [n_acq][N]
    const float* weights;         // [n_acq][N] or null
    float* losses;                // [n_steps] or null
    unsigned* sync;               // [0] arrivals (zeroed by the host before the launch), [1] error word
    unsigned spin_limit;          // polls a block waits at a grid barrier before it raises the error word
    long long w_off[SMALL_MAX_LAYERS + 1], b_off[SMALL_MAX_LAYERS + 1];
    long long P;
    int N, F, S, n_acq, first_acq, n_steps, nblocks, tpp;
    float first_omega, hidden_omega, inv_count;
    float one_minus_b1, b2, one_minus_b2, eps;
    float step_size[SM_MAX_STEPS], bc2_sqrt[SM_MAX_STEPS];
    unsigned long long* stamps;   // diagnostic builds (-DINR_STAMPS): per-wave s_memtime stamps of the launch's last step
};

#ifdef INR_STAMPS
#define SM_STAMP(slot)                                                                                  \
    do {                                                                                                \
        if (p.stamps && (threadIdx.x & 63) == 0 && step == p.n_steps - 1) {                             \
            unsigned long long t_;                                                                      \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
            p.stamps[((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = t_;               \
        }                                                                                               \
    } while (0)
#else
#define SM_STAMP(slot)
#endif
extern unsigned long long* g_stamps;

// Data that crosses CUs (gradient slabs, updated parameters, loss partials) is stored write-through at agent scope, so a
// block only has to wait for its own stores (vmcnt) before it announces itself -- no L2 write-back of the (much larger)
// block-private stash traffic ...
__device__ __forceinline__ void store_shared(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and everything that crosses CUs is READ with agent-scope (sc1) loads, which do not trust a line cached before the
// barrier: the barriers need no cache maintenance at all, and the coordinates, targets and the block-private stash stay
// cache-resident from step to step (an acquire after the Adam barrier, i.e. an L2 invalidate, made the first loads of
// every step take ~7 us).
__device__ __forceinline__ float load_shared(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Arrive + wait on a monotonically increasing counter (`target` = barrier ordinal x blocks).  Returns false when the
// launch has to be abandoned (poll limit or another block's error word); the value is uniform over the block.
template <bool ACQUIRE>
__device__ __forceinline__ bool small_grid_barrier(unsigned* sync, unsigned target, int* flag, unsigned spin_limit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's write-through stores have reached memory
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned polls = 0;
        int ok = 1;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > spin_limit || __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *flag = ok;
    }
    __syncthreads();
    if (ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave drops the lines it cached before
    return *flag != 0;
}

template <int H, int ROWS>
__global__ void __launch_bounds__(SM_THREADS) siren_small_multi_kernel(const SmallMulti p) {
    constexpr int RT = ROWS / 16;       // 16-row tiles of the block
    constexpr int WPR = 8 / RT;         // waves that share a row tile
    static_assert((H / 16) % WPR == 0, "column tiles must divide over the waves of a row tile");
    constexpr int LS = H + 4;           // row stride of the [row][feature] images (floats): 16-B rows, odd number of 16-B slots
    constexpr int LT = ROWS + 4;     // row stride of the transposed [feature][row] images
    constexpr int IMG = (H * LT > ROWS * LS) ? H * LT : ROWS * LS;
    constexpr int CTW = (H / 16) / WPR; // 16-wide column tiles per wave
    constexpr int KBH = H / 16;         // 16-deep k blocks of a hidden layer
    constexpr int WQ = (H * H / 4 + SM_THREADS - 1) / SM_THREADS;     // float4 of a weight matrix per thread
    constexpr int AQ = (ROWS * H / 4) / SM_THREADS;                // float4 of an activation tile per thread
    // forward: ldsW[l&1] = W_l [j][k], ldsA[l&1] = a_l [row][k].   backward: ldsW[(l+1)&1] = W_l^T [k][j],
    // ldsA[l&1] = a_l^T [k][row], ldsD[l&1] = dz_l [row][j], ldsT[l&1] = dz_l^T [j][row].
    __shared__ __attribute__((aligned(16))) float ldsW[2][H * LS];
    __shared__ __attribute__((aligned(16))) float ldsA[2][IMG];
    __shared__ __attribute__((aligned(16))) float ldsD[2][ROWS * LS];
    __shared__ __attribute__((aligned(16))) float ldsT[2][H * LT];
    __shared__ float redb[2][RT][H];
    __shared__ float redh[SM_THREADS / H][H];
    __shared__ float ldsB[(SMALL_MAX_LAYERS + 1) * H + 4];   // this step's biases (layer-major), head weights, head bias
    __shared__ float gbuf[ROWS];
    __shared__ float wred[2][8];
    __shared__ int bar_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int rt = wave / WPR, cb = (wave % WPR) * CTW;
    const int blk = blockIdx.x, nblk = p.nblocks;
    const int r0 = blk * ROWS;
    const int S = p.S, F = p.F;
    const int KB0 = (F + 15) / 16;                       // k blocks of layer 0 (features zero-padded to 16)
    float* slab = p.slabs + (long long)blk * p.P;
    const long long tile_floats = (long long)nblk * ROWS * H;
    float* tstash = p.acts + (long long)blk * ROWS * H;            // + l * tile_floats: a_l^T [k][64 rows] of this block
    f32x4* dnat = reinterpret_cast<f32x4*>(p.dacts) + ((long long)blk * 8 + wave) * CTW * 64 + lane;
    const long long dnat_layer = (long long)nblk * 8 * CTW * 64;      // float4 per layer
    unsigned barrier_no = 0;
    // the parameter buffer as seen after other CUs updated it: agent-scope (sc1) buffer loads, 4 or 16 bytes wide
    const __amdgpu_buffer_rsrc_t prsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.params, 0, (int)(p.P * (long long)sizeof(float)), 0x00020000);
    auto param1 = [&](long long idx) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, (int)(idx * 4), 0, 16));
    };
    auto param4 = [&](long long idx) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, (int)(idx * 4), 0, 16));
    };

    f32x4 wreg[WQ];
    auto fetch_weights = [&](int l) {       // hidden layer l >= 1: [H][H] row-major
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            const int f = tid + SM_THREADS * i;
            if (f < H * H / 4) wreg[i] = param4(p.w_off[l] + 4 * f);
        }
    };
    auto commit_weights = [&](float* dst) {             // [j][k]
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            const int f = tid + SM_THREADS * i;
            if (f < H * H / 4) *reinterpret_cast<f32x4*>(dst + (f / (H / 4)) * LS + (f % (H / 4)) * 4) = wreg[i];
        }
    };
    auto commit_weights_t = [&](float* dst) {           // [k][j]
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            const int f = tid + SM_THREADS * i;
            if (f < H * H / 4) {
                const int j = f / (H / 4), k = (f % (H / 4)) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[(k + q) * LS + j] = wreg[i][q];
            }
        }
    };
    // one 16-deep k block of MFMAs for this wave's CTW output tiles: rows from `Aimg` (stride sa), columns from `Bimg`
    auto mfma_block = [&](f32x4* acc, const float* Aimg, int sa, const float* Bimg, int sb, int kb) {
        const f32x4 fa = *reinterpret_cast<const f32x4*>(Aimg + (16 * rt + l16) * sa + 16 * kb + 4 * g);
        f32x4 fb[CTW];
#pragma unroll
        for (int c = 0; c < CTW; ++c) fb[c] = *reinterpret_cast<const f32x4*>(Bimg + (16 * (cb + c) + l16) * sb + 16 * kb + 4 * g);
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[c][s], acc[c], 0, 0, 0);
    };

    // Per-step inputs that never change (coordinates, targets, weights) are fetched into registers BEFORE the barrier that
    // ends the previous step; what depends on the new parameters (W_0, biases, head, W_1) is fetched right after it, all
    // loads in flight together, and only then written to LDS: one memory round trip at the start of a step.
    constexpr int TPR = SM_THREADS / ROWS;                               // threads per row in the head
    constexpr int XQ = ROWS * 32 / SM_THREADS;                           // x tile as [row][32] (zero-padded) per thread
    constexpr int W0Q = H * 32 / SM_THREADS;                             // W_0 as [j][32]
    constexpr int BQ = ((SMALL_MAX_LAYERS + 1) * H + 1 + SM_THREADS - 1) / SM_THREADS;
    const bool row_ok = r0 + tid / TPR < p.N;
    float xv[XQ], tgt_pre = 0.f, wgt_pre = 1.f;
    auto prefetch_inputs = [&](int step) {
        const int acq = (p.first_acq + step) % p.n_acq;
#pragma unroll
        for (int i = 0; i < XQ; ++i) {
            const int e = tid + SM_THREADS * i, row = e >> 5, k = e & 31;
            xv[i] = (k < F && r0 + row < p.N) ? p.x[(long long)(r0 + row) * F + k] : 0.f;
        }
        tgt_pre = row_ok ? p.targets[(long long)acq * p.N + r0 + tid / TPR] : 0.f;
        wgt_pre = (p.weights && row_ok) ? p.weights[(long long)acq * p.N + r0 + tid / TPR] : 1.f;
    };
    prefetch_inputs(0);

    for (int step = 0; step < p.n_steps; ++step) {
        const bool weighted = p.weights != nullptr;
        SM_STAMP(0);

        // ---------------------------------------------- forward ----------------------------------------------------
        {   // W_0 [H][F] -> ldsW[0] ([j][k], zero-padded to 32), x tile -> ldsA[0], biases and the head -> ldsB
            float w0v[W0Q], bv[BQ];
#pragma unroll
            for (int i = 0; i < W0Q; ++i) {
                const int e = tid + SM_THREADS * i, j = e >> 5, k = e & 31;
                w0v[i] = (k < F) ? param1(p.w_off[0] + j * F + k) : 0.f;
            }
#pragma unroll
            for (int i = 0; i < BQ; ++i) {
                const int e = tid + SM_THREADS * i, l = e / H, j = e % H;
                bv[i] = (e < (S + 1) * H + 1) ? param1((l < S) ? p.b_off[l] + j : (l == S) ? p.w_off[S] + j : p.b_off[S]) : 0.f;
            }
            if (S > 1) fetch_weights(1);
#pragma unroll
            for (int i = 0; i < W0Q; ++i) {
                const int e = tid + SM_THREADS * i;
                ldsW[0][(e >> 5) * LS + (e & 31)] = w0v[i];
            }
#pragma unroll
            for (int i = 0; i < XQ; ++i) {
                const int e = tid + SM_THREADS * i;
                ldsA[0][(e >> 5) * LS + (e & 31)] = xv[i];
            }
#pragma unroll
            for (int i = 0; i < BQ; ++i) {
                const int e = tid + SM_THREADS * i;
                if (e < (S + 1) * H + 1) ldsB[e] = bv[i];
            }
        }
        __syncthreads();
        SM_STAMP(1);

        f32x4 acc[CTW], dl[CTW];
        for (int l = 0; l < S; ++l) {
            const float* A = ldsA[l & 1];
            const float* W = ldsW[l & 1];
#pragma unroll
            for (int c = 0; c < CTW; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (l == 0) {
                for (int kb = 0; kb < KB0; ++kb) mfma_block(acc, A, LS, W, LS, kb);
            } else {
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) mfma_block(acc, A, LS, W, LS, kb);
            }
            // a_{l+1} = sin(omega z), d_l = omega cos(omega z)   (SRDWI.py:58-59)
            const float omega = (l == 0) ? p.first_omega : p.hidden_omega;
            const float* bias = ldsB + l * H;
            float* An = ldsA[(l + 1) & 1];
            float* ts = tstash + (long long)(l + 1) * tile_floats;
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int j = 16 * (cb + c) + l16;
                const float bj = bias[j];
                f32x4 sv4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sv, cv;
                    sincos_f32(omega * (acc[c][r] + bj), sv, cv);
                    An[(16 * rt + 4 * g + r) * LS + j] = sv;
                    sv4[r] = sv;
                    dl[c][r] = omega * cv;
                }
                if (l + 1 < S) {
                    *reinterpret_cast<f32x4*>(ts + j * ROWS + 16 * rt + 4 * g) = sv4;     // a_{l+1}^T for the weight gradient
                    dnat[(long long)l * dnat_layer + c * 64] = dl[c];
                }
            }
            if (l + 1 < S) {
                commit_weights(ldsW[(l + 1) & 1]);
                if (l + 2 < S) fetch_weights(l + 2);
            } else if (S > 1) {
                commit_weights_t(ldsW[S & 1]);   // (registers still hold W_{S-1}): its transpose opens the backward pass
            }
            __syncthreads();
        }

        SM_STAMP(2);
        // ---------------------------------------------- head + loss ------------------------------------------------
        const float* AS = ldsA[S & 1];                      // a_S [row][k]
        const float* wh = ldsB + S * H;
        // a_{S-1}^T (input of the last sine layer) for the first weight gradient: from the stash / the network input
        f32x4 apre[AQ];
        auto fetch_at = [&](int l) {                        // l >= 1
            const float* src = tstash + (long long)l * tile_floats;
#pragma unroll
            for (int i = 0; i < AQ; ++i) apre[i] = *reinterpret_cast<const f32x4*>(src + (tid + SM_THREADS * i) * 4);
        };
        auto commit_at = [&](float* dst) {
#pragma unroll
            for (int i = 0; i < AQ; ++i) {
                const int f = tid + SM_THREADS * i, k = f / (ROWS / 4), r4 = (f % (ROWS / 4)) * 4;
                *reinterpret_cast<f32x4*>(dst + k * LT + r4) = apre[i];
            }
        };
        auto stage_input_t = [&](float* dst) {              // x^T [k][row] (zero beyond F / N), from the step's registers
#pragma unroll
            for (int i = 0; i < XQ; ++i) {
                const int e = tid + SM_THREADS * i, row = e >> 5, k = e & 31;
                if (k < 16 * KB0) dst[k * LT + row] = xv[i];
            }
        };
        if (S > 1) fetch_at(S - 1);
        {
            const int row = tid / TPR, sub = tid % TPR;
            float part = 0.f;
#pragma unroll
            for (int q = 0; q < H / TPR; ++q) part = fmaf(AS[row * LS + sub * (H / TPR) + q], wh[sub * (H / TPR) + q], part);
#pragma unroll
            for (int off = 1; off < TPR; off <<= 1) part += __shfl_xor(part, off, 64);
            const float y = part + ldsB[(S + 1) * H];
            const float resid = row_ok ? y - tgt_pre : 0.f;
            const float wr = (weighted && row_ok) ? wgt_pre * resid : resid;
            const float gr = 2.0f * wr * p.inv_count;                   // dL/dy (superresDWI.py:135, master.py:143-145)
            if (sub == 0) gbuf[row] = gr;
            float lsum = (sub == 0) ? wr * resid : 0.f;
            float gsum = (sub == 0) ? gr : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lsum += __shfl_xor(lsum, off, 64);
                gsum += __shfl_xor(gsum, off, 64);
            }
            if (lane == 0) {
                wred[0][wave] = lsum;
                wred[1][wave] = gsum;
            }
        }
        __syncthreads();
        if (tid == 0) {
            float ls = 0.f, gs = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                ls += wred[0][w];
                gs += wred[1][w];
            }
            store_shared(p.loss_partial + blk, ls);
            store_shared(slab + p.b_off[S], gs);                         // head bias gradient
        } else if (tid < 4) {
            store_shared(slab + p.b_off[S] + tid, 0.f);                  // 16-byte padding of the 1-float head bias
        }
        // dz_{S-1} = g (x) w_head * d_{S-1} -> its two LDS images and the bias-gradient partials
        auto publish_dz = [&](const f32x4* dz, int l) {
            float* Dn = ldsD[l & 1];
            float* Tn = ldsT[l & 1];
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int j = 16 * (cb + c) + l16;
                float gb = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Dn[(16 * rt + 4 * g + r) * LS + j] = dz[c][r];
                    gb += dz[c][r];
                }
                *reinterpret_cast<f32x4*>(Tn + j * LT + 16 * rt + 4 * g) = dz[c];
                gb += __shfl_xor(gb, 16, 64);
                gb += __shfl_xor(gb, 32, 64);
                if (g == 0) redb[l & 1][rt][j] = gb;
            }
        };
        f32x4 dz[CTW];
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const float wj = wh[16 * (cb + c) + l16];
#pragma unroll
            for (int r = 0; r < 4; ++r) dz[c][r] = gbuf[16 * rt + 4 * g + r] * wj * dl[c][r];
        }
        publish_dz(dz, S - 1);
        {   // gW_head[j] = sum_rows g_row a_S[row][j]: one partial per row group, combined in fixed order after the barrier
            constexpr int GROUPS = SM_THREADS / H, RPG = ROWS / GROUPS;
            const int j = tid % H, q = tid / H;
            float gw = 0.f;
#pragma unroll
            for (int i = 0; i < RPG; ++i) gw = fmaf(gbuf[q * RPG + i], AS[(q * RPG + i) * LS + j], gw);
            redh[q][j] = gw;
        }
        if (S > 1) commit_at(ldsA[(S - 1) & 1]);
        else stage_input_t(ldsA[0]);
        __syncthreads();
        SM_STAMP(3);

        // ---------------------------------------------- backward ---------------------------------------------------
        if (tid < H) {
            float gw = 0.f;
#pragma unroll
            for (int q = 0; q < SM_THREADS / H; ++q) gw += redh[q][tid];
            store_shared(slab + p.w_off[S] + tid, gw);
        }
        // Order inside an iteration: the loads the NEXT iterations need are issued first (memory operations retire in order:
        // a load queued behind this iteration's write-through gradient stores would wait for their trip to memory), the
        // gradient stores come last.
        constexpr int GT = (KBH * KBH + 7) / 8;                          // weight-gradient tiles per wave
        f32x4 pend[GT];
        long long pend_base[GT], pend_bias_at = -1;
        int pend_stride = 0;
        float pend_bias = 0.f;
#pragma unroll
        for (int u = 0; u < GT; ++u) pend_base[u] = -1;
        auto flush_grads = [&]() {
#pragma unroll
            for (int u = 0; u < GT; ++u) {
                if (pend_base[u] >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_shared(slab + pend_base[u] + (long long)r * pend_stride, pend[u][r]);
                }
            }
            if (pend_bias_at >= 0) store_shared(slab + pend_bias_at, pend_bias);
        };
        f32x4 dnow[CTW], dnext[CTW];
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            dnow[c] = (S > 1) ? dnat[(long long)(S - 2) * dnat_layer + c * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int l = S - 1; l >= 0; --l) {
            const float* D = ldsD[l & 1];
            const float* T = ldsT[l & 1];
            const float* At = ldsA[l & 1];
            const float* Wt = ldsW[(l + 1) & 1];
            const int K = (l == 0) ? F : H;
            float gb_l = 0.f;
            if (tid < H) {                                               // bias gradient of layer l: the row-tile partials
#pragma unroll
                for (int q = 0; q < RT; ++q) gb_l += redb[l & 1][q][tid];
            }
            if (l > 1) {                                                 // a_{l-1}^T, W_{l-1} for the layer below, d_{l-2} beyond
                fetch_at(l - 1);
                fetch_weights(l - 1);
#pragma unroll
                for (int c = 0; c < CTW; ++c) dnext[c] = dnat[(long long)(l - 2) * dnat_layer + c * 64];
            }
            if (l > 0) {
                // da_l = dz_l W_l ; dz_{l-1} = da_l * d_{l-1}
#pragma unroll
                for (int c = 0; c < CTW; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) mfma_block(acc, D, LS, Wt, LS, kb);
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dz[c][r] = acc[c][r] * dnow[c][r];
                publish_dz(dz, l - 1);
#pragma unroll
                for (int c = 0; c < CTW; ++c) dnow[c] = dnext[c];
            }
            // this layer's gradient stores wait until the NEXT iteration has queued its loads (see above); the previous
            // layer's go out now
            flush_grads();
            // weight gradient gW_l[j][k] = sum over the block's rows of dz_l[row][j] a_l[row][k]
            const int ktn = (l == 0) ? KB0 : KBH;
#pragma unroll
            for (int u = 0; u < GT; ++u) {
                const int t = wave + 8 * u;
                pend_base[u] = -1;
                if (t < KBH * ktn) {
                    const int jt = t / ktn, kt = t % ktn;
                    f32x4 wacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < ROWS / 16; ++kb) {
                        const f32x4 fa = *reinterpret_cast<const f32x4*>(T + (16 * jt + l16) * LT + 16 * kb + 4 * g);
                        const f32x4 fb = *reinterpret_cast<const f32x4*>(At + (16 * kt + l16) * LT + 16 * kb + 4 * g);
#pragma unroll
                        for (int s = 0; s < 4; ++s) wacc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[s], wacc, 0, 0, 0);
                    }
                    const int k = 16 * kt + l16;
                    pend[u] = wacc;
                    pend_stride = K;
                    if (k < K) pend_base[u] = p.w_off[l] + (long long)(16 * jt + 4 * g) * K + k;
                }
            }
            pend_bias = gb_l;
            pend_bias_at = (tid < H) ? p.b_off[l] + tid : -1;
            if (l > 1) {
                commit_at(ldsA[(l - 1) & 1]);
                commit_weights_t(ldsW[l & 1]);
            } else if (l == 1) {
                stage_input_t(ldsA[0]);                                  // a_0 = the network input
            }
            __syncthreads();
        }

        flush_grads();                                                   // layer 0's
        // ------------------------------- gradient reduction + Adam over the whole grid -----------------------------
        SM_STAMP(4);
        if (!small_grid_barrier<false>(p.sync, ++barrier_no * (unsigned)nblk, &bar_flag, p.spin_limit)) return;
        SM_STAMP(5);
        {
            const float step_size = p.step_size[step], bc2_sqrt = p.bc2_sqrt[step];
            // `tpp` adjacent lanes share a parameter: lane q sums the slabs b = q (mod tpp) -- every load of a batch in
            // flight together (they come from other CUs: a memory round trip each) -- then the lanes are folded in a fixed
            // order; the order of the whole sum depends on the grid only
            const int tpp = p.tpp;
            const long long gthreads = (long long)nblk * SM_THREADS;
            for (long long base = 0; base < p.P; base += gthreads / tpp) {
                const long long gt = (long long)blk * SM_THREADS + tid;
                const long long i = base + gt / tpp;
                const int q = (int)(gt % tpp);
                const bool live = i < p.P;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                float m0 = 0.f, v0 = 0.f, w0 = 0.f;
                if (live) {
                    if (q == 0) {                      // the optimizer state rides in the same round trip as the slabs
                        m0 = p.m[i];
                        v0 = p.v[i];
                        w0 = param1(i);
                    }
                    constexpr int BATCH = 64;
                    for (int b0 = q; b0 < nblk; b0 += BATCH * tpp) {
                        float t[BATCH];
#pragma unroll
                        for (int u = 0; u < BATCH; ++u)
                            t[u] = (b0 + u * tpp < nblk) ? load_shared(p.slabs + (long long)(b0 + u * tpp) * p.P + i) : 0.f;
#pragma unroll
                        for (int u = 0; u < BATCH; u += 4) {
                            a0 += t[u];
                            a1 += t[u + 1];
                            a2 += t[u + 2];
                            a3 += t[u + 3];
                        }
                    }
                }
                float gi = (a0 + a1) + (a2 + a3);
                for (int off = 1; off < tpp; off <<= 1) gi += __shfl_xor(gi, off, 64);
                if (live && q == 0) {
                    p.grads[i] = gi;
                    const float mi = fmaf(gi - m0, p.one_minus_b1, m0);
                    const float vi = fmaf(p.one_minus_b2 * gi, gi, v0 * p.b2);
                    const float denom = __fsqrt_rn(vi) / bc2_sqrt + p.eps;
                    p.m[i] = mi;
                    p.v[i] = vi;
                    store_shared(p.params + i, w0 - step_size * (mi / denom));
                }
            }
            if (blk == 0 && wave == 0 && p.losses) {
                float ls = 0.f;
                for (int b = lane; b < nblk; b += 64) ls += load_shared(p.loss_partial + b);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off, 64);
                if (lane == 0) p.losses[step] = ls * p.inv_count;
            }
        }
        if (step + 1 < p.n_steps) prefetch_inputs(step + 1);
        SM_STAMP(6);
        if (!small_grid_barrier<false>(p.sync, ++barrier_no * (unsigned)nblk, &bar_flag, p.spin_limit)) return;
        SM_STAMP(7);
    }
}

tune_int g_small_spin_limit{0};   // inr_debug_set(17, n): poll limit of the grid barrier (0 = SM_SPIN_LIMIT); tests force the abandon path with 1
tune_int g_small_rows{0};   // rows per block of the persistent kernel: 0 = choose (32 when that still fits one block per CU), 32, 64

// Cooperative launches need every block co-resident.  The limit comes from the device (CU count x blocks of the chosen
// instantiation that fit a CU -- one, at ~140 KB of static LDS), not from a constant: a partitioned (CPX / DPX) or CU-masked
// device has fewer than 256 CUs.  Cached per process; without a device (build container) the MI355X figure stands in.
template <int H, int ROWS>
static int multi_capacity_of() {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 256;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, siren_small_multi_kernel<H, ROWS>, SM_THREADS, 0) != hipSuccess ||
        per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    return prop.multiProcessorCount * per_cu;
}
static int multi_capacity(int H, int rows) {
    // per DEVICE (a process may drive several, or partitions of different size) and race-free: relaxed atomics -- two threads
    // that both find 0 compute the same value
    constexpr int MAX_DEV = 64;
    static std::atomic<int> cap[MAX_DEV][3];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        dev = 0;
    }
    const int k = (H == 32) ? 0 : (rows == 32 ? 1 : 2);
    auto compute = [&]() { return (k == 0) ? multi_capacity_of<32, 64>() : (k == 1) ? multi_capacity_of<64, 32>() : multi_capacity_of<64, 64>(); };
    if (dev < 0 || dev >= MAX_DEV) return compute();
    int c = cap[dev][k].load(std::memory_order_relaxed);
    if (!c) {
        c = compute();
        cap[dev][k].store(c, std::memory_order_relaxed);
    }
    return c;
}

static int multi_rows(const inr_siren_desc_t* d, int64_t n) {
    if (d->hidden_features != 64) return 64;
    if (g_small_rows == 64) return 64;
    return (n + 31) / 32 <= multi_capacity(64, 32) ? 32 : 64;
}

bool small_multi_ok(const inr_siren_desc_t* d, int64_t n) {
    return small_path_ok(d, n) && (n + 63) / 64 <= multi_capacity(d->hidden_features, 64);      // all blocks co-resident
}

static inline int multi_blocks(const inr_siren_desc_t* d, int64_t n) {
    const int rows = multi_rows(d, n);
    return (int)((n + rows - 1) / rows);
}

size_t small_multi_workspace_floats(const inr_siren_desc_t* d, int64_t n, long long P) {
    const int S = d->hidden_layers + 1;
    const size_t tile = (size_t)((n + 63) / 64) * 64 * d->hidden_features;      // rows padded to whole blocks (either block size)
    const size_t nb = (size_t)((n + 31) / 32);                                  // the larger block count
    return 2 * (size_t)S * tile + nb * (size_t)P + nb + 64 + 64;                // acts, dacts, slabs, loss partials, sync words
}

// n_steps optimizer steps; step `it` fits acquisition (first_acq + it) % n_acq (targets / weights: [n_acq][n] contiguous).
int small_fit_multi(const inr_siren_desc_t* d, const long long* w_off, const long long* b_off, long long P, float* params,
                    float* grads, float* m, float* v, const float* x, const float* targets, const float* weights, int n_acq,
                    int first_acq, int64_t n, int64_t first_step, int n_steps, double lr, double b1, double b2, double eps,
                    float* losses, float* ws, hipStream_t st) {
    const int S = d->hidden_layers + 1, H = d->hidden_features;
    const int rows = multi_rows(d, n);
    const int nb = multi_blocks(d, n);
    const size_t tile = (size_t)((n + 63) / 64) * 64 * H;
    SmallMulti p{};
    p.params = params; p.grads = grads; p.m = m; p.v = v;
    p.acts = ws;
    p.dacts = ws + (size_t)S * tile;
    p.slabs = p.dacts + (size_t)S * tile;
    p.loss_partial = p.slabs + (size_t)((n + 31) / 32) * P;
    p.sync = reinterpret_cast<unsigned*>(p.loss_partial + (((size_t)((n + 31) / 32) + 63) / 64) * 64);
    p.x = x; p.targets = targets; p.weights = weights;
    for (int l = 0; l <= S; ++l) { p.w_off[l] = w_off[l]; p.b_off[l] = b_off[l]; }
    p.P = P; p.N = (int)n; p.F = d->in_features; p.S = S; p.n_acq = n_acq; p.nblocks = nb;
    p.tpp = 1;
    while (p.tpp < 8 && (long long)nb * SM_THREADS / (2 * p.tpp) >= P) p.tpp *= 2;   // lanes per parameter in the reduction
    p.first_omega = d->first_omega; p.hidden_omega = d->hidden_omega;
    p.inv_count = (float)(1.0 / (double)n);
    p.one_minus_b1 = (float)(1.0 - b1); p.b2 = (float)b2; p.one_minus_b2 = (float)(1.0 - b2); p.eps = (float)eps;
    p.stamps = g_stamps;
    const int lim = g_small_spin_limit;
    p.spin_limit = lim > 0 ? (unsigned)lim : SM_SPIN_LIMIT;
    for (int done = 0; done < n_steps; done += SM_MAX_STEPS) {
        const int k = n_steps - done < SM_MAX_STEPS ? n_steps - done : SM_MAX_STEPS;
        p.n_steps = k;
        p.first_acq = (int)(((long long)first_acq + done) % n_acq);
        p.losses = losses ? losses + done : nullptr;
        for (int i = 0; i < k; ++i) {
            const double t = (double)(first_step + done + i);
            p.step_size[i] = (float)(lr / (1.0 - pow(b1, t)));
            p.bc2_sqrt[i] = (float)sqrt(1.0 - pow(b2, t));
        }
        ProfScope ps(KC_OTHER, st);
        INR_HIP(hipMemsetAsync(p.sync, 0, (done == 0 ? 2 : 1) * sizeof(unsigned), st));   // the error word is sticky within a call
        void* args[] = {(void*)&p};
        const void* fn = (H == 32) ? (const void*)siren_small_multi_kernel<32, 64>
                         : (rows == 32) ? (const void*)siren_small_multi_kernel<64, 32>
                                        : (const void*)siren_small_multi_kernel<64, 64>;
        const hipError_t e = hipLaunchCooperativeKernel(fn, dim3(nb), dim3(SM_THREADS), args, 0, st);   // refused unless co-resident
        if (e == hipErrorCooperativeLaunchTooLarge && done == 0) {
            (void)hipGetLastError();
            return INR_E_FALLBACK;   // (caller: the two-launch step serves this device)
        }
        INR_HIP(e);
        count_launch(LF_SMALL_MULTI);
    }
    // A grid barrier that ran into its poll limit makes every block leave mid-step: parameters, Adam moments and losses are
    // then partly updated.  That must not pass for success, so this path reads the error word back -- the one place where an
    // entry point waits for the stream (cooperative launches cannot be captured into a graph anyway; include/inrhip.h).
    unsigned err = 0;
    INR_HIP(hipMemcpyAsync(&err, p.sync + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    INR_HIP(hipStreamSynchronize(st));
    INR_REQUIRE(err == 0, INR_E_TIMEOUT,
                "small-network persistent kernel: a grid barrier exceeded its poll limit and the launch was abandoned; "
                "params / m / v / losses of this call are partly updated and must be discarded");
    return 0;
}

}  // namespace inr
