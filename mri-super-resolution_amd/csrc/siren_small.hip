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
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(small_reduce_adam_kernel, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, st, params, grads, m, v,
                       p.slabs, nwaves, P, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(lr / bc1),
                       (float)sqrt(bc2), (float)eps, loss_out, p.loss_partial, p.inv_count);
    INR_LAUNCH_CHECK();
    return 0;
}

}  // namespace inr
