// Bandwidth-bound kernels of the INR fit path: coordinate grid, Fourier features, linear head
// (wavefront-shuffle row reduction), MSE residual/gradient, head backward, fixed-order column sums
// and slab reductions (bias / weight gradients, loss), fused Adam.  All fp32, no float atomics.
#include "common.h"

namespace inr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GridShape {
    int64_t n[8];
    int dim;
};

// ---- a-1: get_mgrid (SRDWI.py:12-18) ---------------------------------------------------------------
__global__ void mgrid_kernel(float* __restrict__ out, GridShape g, int64_t row_begin, int64_t n_rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    int64_t rem = row_begin + i;
    float v[8];
#pragma unroll
    for (int a = 7; a >= 0; --a) {
        if (a < g.dim) {
            const int64_t idx = rem % g.n[a];
            rem /= g.n[a];
            v[a] = linspace_pm1(idx, g.n[a]);
        }
    }
    float* o = out + i * g.dim;
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (a < g.dim) o[a] = v[a];
}

// ---- a-3: input_mapping (SRDWI.py:111-116); FROM_GRID fuses get_mgrid in -----------------------------
// one thread per (row, frequency j): proj = sum_a fl(2*pi*x_a) * B[j][a]; out[row][j] = sin, [m+j] = cos
template <bool FROM_GRID>
__global__ void fourier_kernel(float* __restrict__ out, const float* __restrict__ x, GridShape g, int d,
                               int64_t row_begin, int64_t n_rows, const float* __restrict__ B, int m) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * m) return;
    const int64_t row = t / m;
    const int j = (int)(t - row * m);
    const float two_pi = 6.283185307179586f;
    float proj = 0.f;
    if (FROM_GRID) {
        int64_t rem = row_begin + row;
        float c[8];
#pragma unroll
        for (int a = 7; a >= 0; --a) {
            if (a < g.dim) {
                const int64_t idx = rem % g.n[a];
                rem /= g.n[a];
                c[a] = linspace_pm1(idx, g.n[a]);
            }
        }
#pragma unroll
        for (int a = 0; a < 8; ++a)
            if (a < g.dim) proj = fmaf(two_pi * c[a], B[j * g.dim + a], proj);
    } else {
        for (int a = 0; a < d; ++a) proj = fmaf(two_pi * x[row * d + a], B[j * d + a], proj);
    }
    float s, c;
    sincos_f32(proj, s, c);
    out[row * (2 * m) + j] = s;
    out[row * (2 * m) + m + j] = c;
}

// ---- a-5: head y = a W^T + b; one wave per row, shuffle reduction ---------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool VEC>
__global__ void __launch_bounds__(256) head_forward_kernel(float* __restrict__ y, float* __restrict__ dy,
                                                           const float* __restrict__ a,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ b, int64_t n, int hidden,
                                                           int out_f, int use_clamp, float clamp_min) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < n; row += nwaves) {
        const float* ar = a + row * hidden;
        for (int o = 0; o < out_f; ++o) {
            const float* w = W + (int64_t)o * hidden;
            float acc = 0.f;
            if (VEC) {
                for (int k = lane * 4; k < hidden; k += 256) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(ar + k);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
                    acc = fmaf(av[0], wv[0], acc);
                    acc = fmaf(av[1], wv[1], acc);
                    acc = fmaf(av[2], wv[2], acc);
                    acc = fmaf(av[3], wv[3], acc);
                }
            } else {
                for (int k = lane; k < hidden; k += 64) acc = fmaf(ar[k], w[k], acc);
            }
            acc = wave_sum(acc);
            if (lane == 0) {
                float v = acc + (b ? b[o] : 0.f);
                if (use_clamp == 1) v = fmaxf(v, clamp_min);
                if (use_clamp == 2) {   // PerturbNet output: clamp_min carries the scale eps (SRDWI.py:107)
                    const float t = tanhf(v);
                    v = clamp_min * t;
                    if (dy) dy[row * out_f + o] = clamp_min * (1.0f - t * t);
                }
                y[row * out_f + o] = v;
            }
        }
    }
}

// ---- a-6: residual, gradient, per-block loss partials ------------------------------------------------
__global__ void __launch_bounds__(256) mse_kernel(float* __restrict__ gy, float* __restrict__ partial,
                                                  const float* __restrict__ y, const float* __restrict__ t,
                                                  const float* __restrict__ w, int64_t count, float inv_count) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float r = y[i] - t[i];
        const float wr = w ? w[i] * r : r;
        gy[i] = 2.0f * wr * inv_count;
        acc = fmaf(wr, r, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// final fixed-order sum of `nparts` partials, scaled
__global__ void __launch_bounds__(256) finish_sum_kernel(float* __restrict__ out, const float* __restrict__ partial,
                                                         int nparts, float scale) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// ---- head backward: dz_last[row][k] = (sum_o gy[row][o] W[o][k]) * dact[row][k] --------------------
__global__ void __launch_bounds__(256) head_dz_kernel(float* __restrict__ dz, const float* __restrict__ gy,
                                                      const float* __restrict__ W,
                                                      const float* __restrict__ dact, int64_t n, int hidden,
                                                      int out_f) {
    const int64_t total = n * hidden;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / hidden;
        const int k = (int)(i - row * hidden);
        float da = 0.f;
        for (int o = 0; o < out_f; ++o) da = fmaf(gy[row * out_f + o], W[(int64_t)o * hidden + k], da);
        dz[i] = dact ? da * dact[i] : da;
    }
}

// ---- fused head backward (out_features == 1, hidden/4 divides 256) ---------------------------------------
// One pass over the last sine layer's rows does everything the head needs:
//   dz[row][k]  = gy[row] * W[k] * dact[row][k]                      (in place over dact is fine)
//   slab_b[blk][k] = sum_{rows of blk} dz[row][k]                    (bias grad of the last sine layer)
//   slab_w[blk][k] = sum_{rows of blk} gy[row] * a[row][k]           (weight grad of the head)
// float4 per thread, column group fixed per thread so the sums stay in registers; rows of a block are
// visited in a fixed order and the RP row-phases are combined through LDS in a fixed order (deterministic).
constexpr int HEAD_ROWS_PER_BLOCK = 256;
__global__ void __launch_bounds__(256) head_bwd_fused_kernel(float* dz, float* __restrict__ slab_b,
                                                             float* __restrict__ slab_w, const float* __restrict__ gy,
                                                             const float* __restrict__ W, const float* __restrict__ a,
                                                             const float* dact, int64_t n, int hidden,
                                                             unsigned* __restrict__ amax_out) {
    __shared__ f32x4 red[2][256];
    const int Q = hidden >> 2;           // float4 groups per row (divides 256)
    const int RP = 256 / Q;              // rows per pass
    const int c4 = threadIdx.x % Q, rsub = threadIdx.x / Q;
    const int64_t r0 = (int64_t)blockIdx.x * HEAD_ROWS_PER_BLOCK;
    const int64_t r1 = min(n, r0 + HEAD_ROWS_PER_BLOCK);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + 4 * c4);
    f32x4 sb = {0.f, 0.f, 0.f, 0.f}, sw = {0.f, 0.f, 0.f, 0.f};
    float omax = 0.f;   // max |dz| (scale of the split-fp16 GEMMs that consume dz); max is order independent
    for (int64_t r = r0 + rsub; r < r1; r += RP) {
        const float g = gy[r];
        const int64_t off = r * hidden + 4 * c4;
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(dact + off);
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(a + off);
        const f32x4 o = (g * w4) * d4;
        *reinterpret_cast<f32x4*>(dz + off) = o;
        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        sb += o;
        sw += g * a4;
    }
    if (amax_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off, 64));
        if ((threadIdx.x & 63) == 0 && omax > 0.f) atomicMax(amax_out, __float_as_uint(omax));
    }
    red[0][threadIdx.x] = sb;
    red[1][threadIdx.x] = sw;
    __syncthreads();
    if (rsub == 0) {
        for (int k = 1; k < RP; ++k) {
            sb += red[0][k * Q + c4];
            sw += red[1][k * Q + c4];
        }
        *reinterpret_cast<f32x4*>(slab_b + (int64_t)blockIdx.x * hidden + 4 * c4) = sb;
        *reinterpret_cast<f32x4*>(slab_w + (int64_t)blockIdx.x * hidden + 4 * c4) = sw;
    }
}

// ---- head forward + loss + head backward in ONE pass (out_features == 1, hidden = 256 * M) ------------------------------
// The loss gradient of a row depends on that row alone (superresDWI.py:135: g_n = 2 w_n (y_n - t_n) / count), so the wave
// that has just read act_L[n, :] for y_n can go straight on to everything the backward pass needs from the row:
// dz_L[n, :] = g_n w dact_L[n, :] (written over dact_L), its column sums (bias gradient of the last sine layer), the head's
// weight gradient sum_n g_n act_L[n, :], sum_n g_n and the loss term.  Against head_forward + mse + head_bwd_fused this
// reads act_L once instead of twice (-1.07 GB per step at N = 524,288) and saves three launches.
// One wave per row, lane l owns columns 256 j + 4 l .. + 3; rows r0 + wave, + 4, ... of the block's 256-row range;
// fixed-order cross-wave and cross-block reductions (slabs), max|dz| by atomic max on the float bits.
template <int M>
__global__ void __launch_bounds__(256) head_step_fused_kernel(float* dz, float* __restrict__ slab_b, float* __restrict__ slab_w,
                                                              float* __restrict__ part_loss, float* __restrict__ part_g,
                                                              const float* __restrict__ a, const float* dact,
                                                              const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ t, const float* __restrict__ wgt,
                                                              int64_t n, float inv_count, unsigned* __restrict__ amax_out) {
    constexpr int H = 256 * M;
    __shared__ f32x4 red[2][4][64 * M];
    __shared__ float red_s[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * HEAD_ROWS_PER_BLOCK;
    const int64_t r1 = min(n, r0 + HEAD_ROWS_PER_BLOCK);
    f32x4 w4[M], sb[M], sw[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        w4[j] = *reinterpret_cast<const f32x4*>(W + 256 * j + 4 * lane);
        sb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        sw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float b0 = bias ? bias[0] : 0.f;
    float loss_acc = 0.f, g_acc = 0.f, omax = 0.f;
    for (int64_t r = r0 + wave; r < r1; r += 4) {
        f32x4 a4[M];
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) {   // same order as head_forward_kernel: y is bitwise what that kernel returns
            a4[j] = *reinterpret_cast<const f32x4*>(a + r * H + 256 * j + 4 * lane);
            acc = fmaf(a4[j][0], w4[j][0], acc);
            acc = fmaf(a4[j][1], w4[j][1], acc);
            acc = fmaf(a4[j][2], w4[j][2], acc);
            acc = fmaf(a4[j][3], w4[j][3], acc);
        }
        const float y = wave_sum(acc) + b0;
        const float res = y - t[r];
        const float wr = wgt ? wgt[r] * res : res;
        const float g = 2.0f * wr * inv_count;
        loss_acc = fmaf(wr, res, loss_acc);
        g_acc += g;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int64_t off = r * H + 256 * j + 4 * lane;
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dact + off);
            const f32x4 o = (g * w4[j]) * d4;
            *reinterpret_cast<f32x4*>(dz + off) = o;
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            sb[j] += o;
            sw[j] += g * a4[j];
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        red[0][wave][64 * j + lane] = sb[j];
        red[1][wave][64 * j + lane] = sw[j];
    }
    if (lane == 0) {           // loss_acc / g_acc are wave-uniform (y comes out of a full-wave reduction)
        red_s[0][wave] = loss_acc;
        red_s[1][wave] = g_acc;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 64 * M; q += 256) {
        const int j = q >> 6, l = q & 63;
        const f32x4 vb = (red[0][0][q] + red[0][1][q]) + (red[0][2][q] + red[0][3][q]);
        const f32x4 vw = (red[1][0][q] + red[1][1][q]) + (red[1][2][q] + red[1][3][q]);
        *reinterpret_cast<f32x4*>(slab_b + (int64_t)blockIdx.x * H + 256 * j + 4 * l) = vb;
        *reinterpret_cast<f32x4*>(slab_w + (int64_t)blockIdx.x * H + 256 * j + 4 * l) = vw;
    }
    if (threadIdx.x == 0) {
        part_loss[blockIdx.x] = (red_s[0][0] + red_s[0][1]) + (red_s[0][2] + red_s[0][3]);
        part_g[blockIdx.x] = (red_s[1][0] + red_s[1][1]) + (red_s[1][2] + red_s[1][3]);
    }
    if (amax_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off, 64));
        if (lane == 0 && omax > 0.f) atomicMax(amax_out, __float_as_uint(omax));
    }
}

// ---- weighted column sums over a row chunk ----------------------------------------------------------
// slab[chunk][gi][c] = sum_{row in chunk} g[row][gi] * X[row][c]   (g == nullptr: G = 1, weight 1)
// grid = (chunks, ceil(C/256)); thread = one column; rows streamed, coalesced across the block.
__global__ void __launch_bounds__(256) colsum_kernel(float* __restrict__ slab, const float* __restrict__ X,
                                                     const float* __restrict__ g, int64_t n, int C, int G,
                                                     int64_t rows_per_chunk) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t r1 = min(n, r0 + rows_per_chunk);
    for (int gi = 0; gi < G; ++gi) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int64_t r = r0;
        if (g) {
            for (; r + 3 < r1; r += 4) {
                a0 = fmaf(g[(r + 0) * G + gi], X[(r + 0) * C + c], a0);
                a1 = fmaf(g[(r + 1) * G + gi], X[(r + 1) * C + c], a1);
                a2 = fmaf(g[(r + 2) * G + gi], X[(r + 2) * C + c], a2);
                a3 = fmaf(g[(r + 3) * G + gi], X[(r + 3) * C + c], a3);
            }
            for (; r < r1; ++r) a0 = fmaf(g[r * G + gi], X[r * C + c], a0);
        } else {
            for (; r + 3 < r1; r += 4) {
                a0 += X[(r + 0) * C + c];
                a1 += X[(r + 1) * C + c];
                a2 += X[(r + 2) * C + c];
                a3 += X[(r + 3) * C + c];
            }
            for (; r < r1; ++r) a0 += X[r * C + c];
        }
        slab[((int64_t)blockIdx.x * G + gi) * C + c] = (a0 + a1) + (a2 + a3);
    }
}

// sum of x[s * pitch], s in [s0, s1): four running sums over s mod 4 (the remainder into the first), then (a0 + a1) + (a2 + a3) --
// THE order of every slab reduction here
__device__ __forceinline__ float slab_run_sum(const float* x, int s0, int s1, int64_t pitch) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = s0;
    for (; s + 3 < s1; s += 4) {
        a0 += x[(int64_t)s * pitch];
        a1 += x[(int64_t)(s + 1) * pitch];
        a2 += x[(int64_t)(s + 2) * pitch];
        a3 += x[(int64_t)(s + 3) * pitch];
    }
    for (; s < s1; ++s) a0 += x[(int64_t)s * pitch];
    return (a0 + a1) + (a2 + a3);
}

// out[g][i] = sum_{s in group g} slab[s][i]  (fixed order; group g = slabs [g*per_group, (g+1)*per_group))
// grid = (ceil(len/256), groups).  With groups == 1 this is the plain slab reduction.
// (slab s starts at slab + s * pitch: pitch == len for a dense stack, larger for partials interleaved with other fields)
__global__ void __launch_bounds__(256) reduce_slabs_kernel(float* __restrict__ out, const float* __restrict__ slab,
                                                           int nslabs, int64_t len, int per_group, int64_t pitch) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const int s0 = blockIdx.y * per_group;
    const int s1 = min(nslabs, s0 + per_group);
    out[(int64_t)blockIdx.y * len + i] = slab_run_sum(slab + i, s0, s1, pitch);
}

// Both stages of a tall reduction in ONE launch: 64 outputs per block, four threads per output take the groups of
// REDUCE_GROUP slabs in turn (group sums through LDS), one of them adds the group sums -- the additions of the two-launch
// form in its order (slab_run_sum over a group's slabs, then over the group sums), so the same bits.  groups <= 32.
__global__ void __launch_bounds__(256) reduce_slabs_onepass_kernel(float* __restrict__ out, const float* __restrict__ slab,
                                                                   int nslabs, int64_t len, int per_group, int64_t pitch) {
    __shared__ float part[32][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
    const int groups = (nslabs + per_group - 1) / per_group;
    if (i < len)
        for (int g = q; g < groups; g += 4) part[g][o] = slab_run_sum(slab + i, g * per_group, min(nslabs, (g + 1) * per_group), pitch);
    __syncthreads();
    if (q == 0 && i < len) out[i] = slab_run_sum(&part[0][o], 0, groups, 64);
}

// ---- deferred reduction + loss + Adam of a fused step (common.h: FinalizeJob) ---------------------------------------------
// first stage for tall slab stacks: one block per (segment, group of FIN_GROUP rows, 256 columns), numbered through the
// prefix table job.s1_first -- no empty blocks (a dense 3-D grid over the maxima launched 21 M of them at 4 M rows: 5 ms)
__global__ void __launch_bounds__(256) finalize_stage1_kernel(const FinalizeJob job) {
    int k = 0;
    while ((long long)blockIdx.x >= job.s1_first[k + 1]) ++k;
    const FinalizeSeg sg = job.seg[k];
    const long long local = (long long)blockIdx.x - job.s1_first[k];
    const long long gx = (sg.len + 255) / 256;
    const int64_t i = (local % gx) * 256 + threadIdx.x;
    const int group = (int)(local / gx);
    const int s0 = group * FIN_GROUP;
    if (i >= sg.len || s0 >= sg.nslabs) return;
    const int s1 = min(sg.nslabs, s0 + FIN_GROUP);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = s0;
    for (; s + 3 < s1; s += 4) {
        a0 += sg.slab[(int64_t)s * sg.len + i];
        a1 += sg.slab[(int64_t)(s + 1) * sg.len + i];
        a2 += sg.slab[(int64_t)(s + 2) * sg.len + i];
        a3 += sg.slab[(int64_t)(s + 3) * sg.len + i];
    }
    for (; s < s1; ++s) a0 += sg.slab[(int64_t)s * sg.len + i];
    sg.stage1[(int64_t)group * sg.len + i] = (a0 + a1) + (a2 + a3);
}

// one thread per gradient element: fixed-order sum of its slab column, gradient out, Adam step (torch's single-tensor
// formulation, as adam_kernel); the block behind the last element block finishes the loss
__global__ void __launch_bounds__(256) finalize_kernel(const FinalizeJob job) {
    const int64_t total = job.first[job.nseg];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if ((int64_t)blockIdx.x * 256 >= total) {   // the loss block
        if (!job.part_loss) return;
        __shared__ float red[256];
        float acc = 0.f;
        for (int i = threadIdx.x; i < job.nparts; i += 256) acc += job.part_loss[i];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) job.loss_out[0] = red[0] * job.loss_scale;
        return;
    }
    if (e >= total) return;
    int k = 0;
    while (e >= job.first[k + 1]) ++k;
    const FinalizeSeg sg = job.seg[k];
    const int64_t i = e - job.first[k];
    const bool tall = sg.nslabs > FIN_TALL;
    const float* src = tall ? sg.stage1 : sg.slab;
    const int ns = tall ? (sg.nslabs + FIN_GROUP - 1) / FIN_GROUP : sg.nslabs;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 3 < ns; s += 4) {
        a0 += src[(int64_t)s * sg.len + i];
        a1 += src[(int64_t)(s + 1) * sg.len + i];
        a2 += src[(int64_t)(s + 2) * sg.len + i];
        a3 += src[(int64_t)(s + 3) * sg.len + i];
    }
    for (; s < ns; ++s) a0 += src[(int64_t)s * sg.len + i];
    const float gi = (a0 + a1) + (a2 + a3);
    const int64_t at = sg.dst + i;
    job.grads[at] = gi;
    if (job.params) {
        const float m0 = job.m[at], v0 = job.v[at];
        const float mi = fmaf(gi - m0, job.one_minus_b1, m0);
        const float vi = fmaf(job.one_minus_b2 * gi, gi, v0 * job.b2);
        const float denom = __fsqrt_rn(vi) / job.bc2_sqrt + job.eps;
        job.m[at] = mi;
        job.v[at] = vi;
        job.params[at] = job.params[at] - job.step_size * (mi / denom);
    }
}

// the same, four consecutive elements per thread (16-byte loads of every slab row, of m / v / params): every element keeps its own
// sum in the scalar kernel's order, so the two forms give identical bits.  Host: only when every segment starts at a multiple of
// four elements (all but the last have lengths that are multiples of four -- the one-element head bias comes last).
// At 4,096 rows the scalar form took 18 us of a 187 us step for 33 MB of slabs (3,594 blocks of 9 dependent 4-byte load rounds).
__global__ void __launch_bounds__(256) finalize_v4_kernel(const FinalizeJob job) {
    const int64_t total = job.first[job.nseg];
    const int64_t quads = (total + 3) / 4;
    if ((int64_t)blockIdx.x * 256 >= quads) {   // the loss block
        if (!job.part_loss) return;
        __shared__ float red[256];
        float acc = 0.f;
        for (int i = threadIdx.x; i < job.nparts; i += 256) acc += job.part_loss[i];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) job.loss_out[0] = red[0] * job.loss_scale;
        return;
    }
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= total) return;
    int k = 0;
    while (e >= job.first[k + 1]) ++k;
    const FinalizeSeg sg = job.seg[k];
    const int64_t i = e - job.first[k];
    const bool tall = sg.nslabs > FIN_TALL;
    const float* src = tall ? sg.stage1 : sg.slab;
    const int ns = tall ? (sg.nslabs + FIN_GROUP - 1) / FIN_GROUP : sg.nslabs;
    const int64_t at = sg.dst + i;
    if (i + 3 < sg.len) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        int s = 0;
        for (; s + 3 < ns; s += 4) {
            a0 += *reinterpret_cast<const f32x4*>(src + (int64_t)s * sg.len + i);
            a1 += *reinterpret_cast<const f32x4*>(src + (int64_t)(s + 1) * sg.len + i);
            a2 += *reinterpret_cast<const f32x4*>(src + (int64_t)(s + 2) * sg.len + i);
            a3 += *reinterpret_cast<const f32x4*>(src + (int64_t)(s + 3) * sg.len + i);
        }
        for (; s < ns; ++s) a0 += *reinterpret_cast<const f32x4*>(src + (int64_t)s * sg.len + i);
        const f32x4 g4 = (a0 + a1) + (a2 + a3);
        *reinterpret_cast<f32x4*>(job.grads + at) = g4;
        if (job.params) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(job.m + at), v0 = *reinterpret_cast<const f32x4*>(job.v + at);
            f32x4 p4 = *reinterpret_cast<const f32x4*>(job.params + at), m4, v4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float gi = g4[q];
                const float mi = fmaf(gi - m0[q], job.one_minus_b1, m0[q]);
                const float vi = fmaf(job.one_minus_b2 * gi, gi, v0[q] * job.b2);
                const float denom = __fsqrt_rn(vi) / job.bc2_sqrt + job.eps;
                m4[q] = mi;
                v4[q] = vi;
                p4[q] = p4[q] - job.step_size * (mi / denom);
            }
            *reinterpret_cast<f32x4*>(job.m + at) = m4;
            *reinterpret_cast<f32x4*>(job.v + at) = v4;
            *reinterpret_cast<f32x4*>(job.params + at) = p4;
        }
        return;
    }
    for (int64_t j = i; j < sg.len; ++j) {      // the (short) tail of a segment whose length is not a multiple of four
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = 0;
        for (; s + 3 < ns; s += 4) {
            a0 += src[(int64_t)s * sg.len + j];
            a1 += src[(int64_t)(s + 1) * sg.len + j];
            a2 += src[(int64_t)(s + 2) * sg.len + j];
            a3 += src[(int64_t)(s + 3) * sg.len + j];
        }
        for (; s < ns; ++s) a0 += src[(int64_t)s * sg.len + j];
        const float gi = (a0 + a1) + (a2 + a3);
        const int64_t aj = sg.dst + j;
        job.grads[aj] = gi;
        if (job.params) {
            const float m0 = job.m[aj], v0 = job.v[aj];
            const float mi = fmaf(gi - m0, job.one_minus_b1, m0);
            const float vi = fmaf(job.one_minus_b2 * gi, gi, v0 * job.b2);
            const float denom = __fsqrt_rn(vi) / job.bc2_sqrt + job.eps;
            job.m[aj] = mi;
            job.v[aj] = vi;
            job.params[aj] = job.params[aj] - job.step_size * (mi / denom);
        }
    }
}

// ---- a-7: Adam (torch single-tensor formulation) ---------------------------------------------------
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                   float one_minus_b1, float b2, float one_minus_b2,
                                                   float step_size, float bc2_sqrt, float eps) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float gi = g[i];
        const float mi = fmaf(gi - m[i], one_minus_b1, m[i]);           // m.lerp_(g, 1-b1)
        const float vi = fmaf(one_minus_b2 * gi, gi, v[i] * b2);        // v*b2 + (1-b2)*g*g
        const float denom = __fsqrt_rn(vi) / bc2_sqrt + eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

__global__ void __launch_bounds__(256) mul_kernel(float* out, const float* a, const float* b, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[i] = a[i] * b[i];
}

__global__ void sincos_probe_kernel(float* s, float* c, const float* x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_f32(x[i], s[i], c[i]);
}

// ================================= host launchers ====================================================
static inline unsigned blocks_for(int64_t work, int per_block, int64_t cap) {
    int64_t b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

static int fill_shape(GridShape& g, const int64_t* shape, int dim) {
    INR_REQUIRE(shape && dim >= 1 && dim <= 8, INR_E_INVALID, "grid dim must be 1..8 (got %d)", dim);
    g.dim = dim;
    for (int a = 0; a < 8; ++a) g.n[a] = 1;
    for (int a = 0; a < dim; ++a) {
        INR_REQUIRE(shape[a] >= 1, INR_E_INVALID, "grid shape[%d] = %lld must be >= 1", a, (long long)shape[a]);
        g.n[a] = shape[a];
    }
    return 0;
}

int launch_mgrid(float* out, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows, hipStream_t st) {
    GridShape g;
    if (int rc = fill_shape(g, shape, dim)) return rc;
    if (n_rows == 0) return 0;
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(mgrid_kernel, dim3(blocks_for(n_rows, 256, 1 << 30)), dim3(256), 0, st, out, g, row_begin,
                       n_rows);
    INR_LAUNCH_CHECK();
    return 0;
}

int launch_fourier(float* out, const float* x, const int64_t* shape, int dim, int64_t row_begin, int64_t n_rows,
                   const float* B, int m, hipStream_t st) {
    GridShape g;
    g.dim = dim;
    for (int a = 0; a < 8; ++a) g.n[a] = 1;
    if (!x) {
        if (int rc = fill_shape(g, shape, dim)) return rc;
    }
    if (n_rows == 0) return 0;
    const int64_t work = n_rows * m;
    ProfScope ps(KC_OTHER, st);
    if (x)
        hipLaunchKernelGGL(fourier_kernel<false>, dim3(blocks_for(work, 256, 1ll << 31)), dim3(256), 0, st, out, x,
                           g, dim, row_begin, n_rows, B, m);
    else
        hipLaunchKernelGGL(fourier_kernel<true>, dim3(blocks_for(work, 256, 1ll << 31)), dim3(256), 0, st, out, x, g,
                           dim, row_begin, n_rows, B, m);
    INR_LAUNCH_CHECK();
    return 0;
}

// mode (use_clamp): 0 plain, 1 clamp(min=clamp_min), 2 y = clamp_min*tanh(.) with optional derivative dy
int launch_head_forward(float* y, const float* a, const float* W, const float* b, int64_t n, int hidden,
                        int out_f, int use_clamp, float clamp_min, hipStream_t st, float* dy = nullptr) {
    if (n == 0) return 0;
    const bool vec = aligned16(a) && aligned16(W) && (hidden % 4 == 0);
    ProfScope ps(KC_OTHER, st);
    const dim3 grid(blocks_for(n, 4, 256 * 32));
    if (vec)
        hipLaunchKernelGGL(head_forward_kernel<true>, grid, dim3(256), 0, st, y, dy, a, W, b, n, hidden, out_f,
                           use_clamp, clamp_min);
    else
        hipLaunchKernelGGL(head_forward_kernel<false>, grid, dim3(256), 0, st, y, dy, a, W, b, n, hidden, out_f,
                           use_clamp, clamp_min);
    INR_LAUNCH_CHECK();
    return 0;
}

int mse_blocks(int64_t count) { return (int)blocks_for(count, 256 * 4, 2048); }

// count_total (0 = count): the divisor of the mean -- larger than `count` when this call sees one row shard of a
// fit that is split over several GPUs (gradients and losses of the shards then simply add up)
int launch_mse(float* gy, float* loss, const float* y, const float* t, const float* w, int64_t count,
               float* partial, hipStream_t st, int64_t count_total = 0) {
    const int nb = mse_blocks(count);
    const float inv = (float)(1.0 / (double)(count_total > 0 ? count_total : count));
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, st, gy, partial, y, t, w, count, inv);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, st, loss, partial, nb, inv);
    INR_LAUNCH_CHECK();
    return 0;
}

int launch_head_dz(float* dz, const float* gy, const float* W, const float* dact, int64_t n, int hidden,
                   int out_f, hipStream_t st) {
    if (n == 0) return 0;
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(head_dz_kernel, dim3(blocks_for(n * hidden, 256, 256 * 64)), dim3(256), 0, st, dz, gy, W,
                       dact, n, hidden, out_f);
    INR_LAUNCH_CHECK();
    return 0;
}

int64_t reduce_tmp_floats(int64_t nslabs, int64_t len);
// chunking used by the column-sum path: ~2048 blocks, at least 32 rows per chunk
int64_t colsum_rows_per_chunk(int64_t n, int C) {
    const int64_t col_groups = (C + 255) / 256;
    int64_t chunks = 2048 / col_groups;
    if (chunks < 1) chunks = 1;
    int64_t rpc = (n + chunks - 1) / chunks;
    if (rpc < 32) rpc = 32;
    return rpc;
}
int64_t colsum_chunks(int64_t n, int C) {
    const int64_t rpc = colsum_rows_per_chunk(n, C);
    return (n + rpc - 1) / rpc;
}
int64_t colsum_ws_floats(int64_t n, int C, int G) {
    const int64_t chunks = colsum_chunks(n, C);
    return chunks * G * C + reduce_tmp_floats(chunks, (int64_t)G * C);
}

int launch_reduce_slabs(float* out, const float* slab, int nslabs, int64_t len, float* tmp, hipStream_t st);
int64_t reduce_tmp_floats(int64_t nslabs, int64_t len);

// out[G][C] = sum_rows g[row][gi]*X[row][c]; slab must hold colsum_ws_floats(n, C, G) floats
int launch_colsum(float* out, const float* X, const float* g, int64_t n, int C, int G, float* slab,
                  hipStream_t st) {
    const int64_t rpc = colsum_rows_per_chunk(n, C);
    const int64_t chunks = (n + rpc - 1) / rpc;
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)chunks, (unsigned)((C + 255) / 256)), dim3(256), 0, st, slab, X,
                       g, n, C, G, rpc);
    INR_LAUNCH_CHECK();
    const int64_t len = (int64_t)G * C;
    return launch_reduce_slabs(out, slab, (int)chunks, len, slab + chunks * len, st);
}

// tall-and-narrow slab stacks (thousands of slabs of a few hundred floats) are reduced in two stages so the
// sum is spread over many blocks; `tmp` must hold reduce_tmp_floats(nslabs, len) floats and not overlap `slab`.
constexpr int REDUCE_GROUP = 32;
tune_int g_reduce_onepass{1};      // inr_debug_set(25, 0): tall reductions in two launches (the form of rounds 1-3; same bits)
int64_t reduce_tmp_floats(int64_t nslabs, int64_t len) {
    return nslabs > 4 * REDUCE_GROUP ? (nslabs + REDUCE_GROUP - 1) / REDUCE_GROUP * len : 0;
}
int launch_reduce_slabs_pitched(float* out, const float* slab, int nslabs, int64_t len, int64_t pitch, float* tmp, hipStream_t st) {
    ProfScope ps(KC_OTHER, st);
    const unsigned gx = blocks_for(len, 256, 1 << 30);
    if (reduce_tmp_floats(nslabs, len) == 0) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx, 1), dim3(256), 0, st, out, slab, nslabs, len, nslabs, pitch);
        INR_LAUNCH_CHECK();
        return 0;
    }
    const int groups = (nslabs + REDUCE_GROUP - 1) / REDUCE_GROUP;
    if (groups <= 32 && g_reduce_onepass) {      // (tmp stays unused)
        hipLaunchKernelGGL(reduce_slabs_onepass_kernel, dim3(blocks_for(len, 64, 1 << 30), 1), dim3(256), 0, st, out, slab, nslabs, len,
                           REDUCE_GROUP, pitch);
        INR_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx, groups), dim3(256), 0, st, tmp, slab, nslabs, len, REDUCE_GROUP, pitch);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx, 1), dim3(256), 0, st, out, tmp, groups, len, groups, len);
    INR_LAUNCH_CHECK();
    return 0;
}
int launch_reduce_slabs(float* out, const float* slab, int nslabs, int64_t len, float* tmp, hipStream_t st) {
    return launch_reduce_slabs_pitched(out, slab, nslabs, len, len, tmp, st);
}

// fused head backward for out_features == 1 (see head_bwd_fused_kernel)
bool head_fused_ok(int hidden, int out_f, const void* a, const void* b, const void* c, const void* d) {
    const int Q = hidden >> 2;
    return out_f == 1 && hidden % 4 == 0 && Q >= 1 && Q <= 256 && 256 % Q == 0 && aligned16(a) && aligned16(b) &&
           aligned16(c) && aligned16(d);
}
int64_t head_fused_blocks(int64_t n) { return (n + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK; }
int launch_head_bwd_fused(float* dz, float* slab_b, float* slab_w, const float* gy, const float* W, const float* a,
                          const float* dact, int64_t n, int hidden, hipStream_t st, unsigned* amax_out) {
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(head_bwd_fused_kernel, dim3((unsigned)head_fused_blocks(n)), dim3(256), 0, st, dz, slab_b,
                       slab_w, gy, W, a, dact, n, hidden, amax_out);
    INR_LAUNCH_CHECK();
    return 0;
}

// the one-pass head step; slab_b / slab_w: [blocks][hidden], part_loss / part_g: [blocks]
bool head_step_fused_ok(int hidden, int out_f, const void* a, const void* b, const void* c, const void* d) {
    return out_f == 1 && (hidden == 256 || hidden == 512 || hidden == 1024) && aligned16(a) && aligned16(b) &&
           aligned16(c) && aligned16(d);
}
int launch_head_step_fused(float* dz, float* slab_b, float* slab_w, float* part_loss, float* part_g, const float* a,
                           const float* dact, const float* W, const float* bias, const float* t, const float* wgt,
                           int64_t n, int hidden, int64_t count_total, hipStream_t st, unsigned* amax_out) {
    const float inv = (float)(1.0 / (double)(count_total > 0 ? count_total : n));
    const dim3 grid((unsigned)head_fused_blocks(n)), block(256);
    ProfScope ps(KC_OTHER, st);
    if (hidden == 256)
        hipLaunchKernelGGL(head_step_fused_kernel<1>, grid, block, 0, st, dz, slab_b, slab_w, part_loss, part_g, a, dact, W,
                           bias, t, wgt, n, inv, amax_out);
    else if (hidden == 512)
        hipLaunchKernelGGL(head_step_fused_kernel<2>, grid, block, 0, st, dz, slab_b, slab_w, part_loss, part_g, a, dact, W,
                           bias, t, wgt, n, inv, amax_out);
    else
        hipLaunchKernelGGL(head_step_fused_kernel<4>, grid, block, 0, st, dz, slab_b, slab_w, part_loss, part_g, a, dact, W,
                           bias, t, wgt, n, inv, amax_out);
    INR_LAUNCH_CHECK();
    return 0;
}
// out[0] = scale * sum(partial[0..nparts)) in a fixed order
int launch_finish_sum(float* out, const float* partial, int nparts, float scale, hipStream_t st) {
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, st, out, partial, nparts, scale);
    INR_LAUNCH_CHECK();
    return 0;
}

// sums every segment of `job`, finishes the loss and -- when job.params is set -- takes Adam step number `adam_step`
int launch_finalize(FinalizeJob& job, long long adam_step, double lr, double b1, double b2, double eps, hipStream_t st) {
    INR_REQUIRE(job.nseg >= 1 && job.nseg <= FIN_MAX_SEG, INR_E_INVALID, "finalize: %d segments", job.nseg);
    job.first[0] = 0;
    job.s1_first[0] = 0;
    for (int k = 0; k < job.nseg; ++k) {
        job.first[k + 1] = job.first[k] + job.seg[k].len;
        long long blocks = 0;
        if (job.seg[k].nslabs > FIN_TALL) {
            INR_REQUIRE(job.seg[k].stage1 != nullptr, INR_E_INVALID, "finalize: tall segment %d without a first-stage buffer", k);
            blocks = ((job.seg[k].len + 255) / 256) * ((job.seg[k].nslabs + FIN_GROUP - 1) / FIN_GROUP);
        }
        job.s1_first[k + 1] = job.s1_first[k] + blocks;
    }
    const long long s1_blocks = job.s1_first[job.nseg];
    if (job.params) {
        const double bc1 = 1.0 - pow(b1, (double)adam_step), bc2 = 1.0 - pow(b2, (double)adam_step);
        job.one_minus_b1 = (float)(1.0 - b1);
        job.b2 = (float)b2;
        job.one_minus_b2 = (float)(1.0 - b2);
        job.step_size = (float)(lr / bc1);
        job.bc2_sqrt = (float)sqrt(bc2);
        job.eps = (float)eps;
    }
    ProfScope ps(KC_OTHER, st);
    if (s1_blocks > 0) {
        INR_REQUIRE(s1_blocks < (1ll << 31), INR_E_INVALID, "finalize: too many first-stage blocks (%lld)", s1_blocks);
        hipLaunchKernelGGL(finalize_stage1_kernel, dim3((unsigned)s1_blocks), dim3(256), 0, st, job);
        INR_LAUNCH_CHECK();
    }
    // four elements per thread where every segment starts (and, but for the last, ends) on a 16-byte boundary of its buffers
    bool v4 = aligned16(job.grads) && (!job.params || (aligned16(job.params) && aligned16(job.m) && aligned16(job.v)));
    for (int k = 0; k < job.nseg && v4; ++k) {
        const FinalizeSeg& sg = job.seg[k];
        const bool tall = sg.nslabs > FIN_TALL;
        v4 = job.first[k] % 4 == 0 && sg.dst % 4 == 0 && aligned16(tall ? sg.stage1 : sg.slab) &&
             (sg.len % 4 == 0 || k == job.nseg - 1);
    }
    if (v4) {
        const long long blocks = ((job.first[job.nseg] + 3) / 4 + 255) / 256 + 1;   // + the loss block
        hipLaunchKernelGGL(finalize_v4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, job);
        INR_LAUNCH_CHECK();
        return 0;
    }
    const long long blocks = (job.first[job.nseg] + 255) / 256 + 1;   // + the loss block
    INR_REQUIRE(blocks < (1ll << 31), INR_E_INVALID, "finalize: too many elements");
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, job);
    INR_LAUNCH_CHECK();
    return 0;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t count, int64_t step, double lr, double b1,
                double b2, double eps, hipStream_t st) {
    if (count == 0) return 0;
    // host-side double bias corrections, as torch's _single_tensor_adam does for python-float lr
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const double step_size = lr / bc1;
    const double bc2_sqrt = sqrt(bc2);
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(count, 256, 4096)), dim3(256), 0, st, p, g, m, v, count,
                       (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)step_size, (float)bc2_sqrt,
                       (float)eps);
    INR_LAUNCH_CHECK();
    return 0;
}

// ---- (f)-2: every combination of one acquisition per b-value at every voxel (SRDWI.py:143-152 calculate_combinations,
// which superresDWI.py:57-76 maps over the voxels with a 32-process pool).  itertools.product order: the LAST b-value's
// acquisition index runs fastest.  raw_b: [nvox][n_b] fp32 (b = 0: [nvox]); out: [nvox][4][K], K = n1 n2 n3:
//   out[v][0][c] = raw0[v]; out[v][1][c] = raw1[v][c / (n2 n3)]; out[v][2][c] = raw2[v][(c / n3) % n2]; out[v][3][c] = raw3[v][c % n3]
__global__ void __launch_bounds__(256) acquisition_products_kernel(float* __restrict__ out, const float* __restrict__ r0,
                                                                   const float* __restrict__ r1, const float* __restrict__ r2,
                                                                   const float* __restrict__ r3, int64_t nvox, int n1, int n2,
                                                                   int n3) {
    const int K = n1 * n2 * n3;
    const int64_t total = nvox * 4 * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % K);
        const int b = (int)((i / K) & 3);
        const int64_t v = i / (4 * (int64_t)K);
        float val;
        if (b == 0) val = r0[v];
        else if (b == 1) val = r1[v * n1 + c / (n2 * n3)];
        else if (b == 2) val = r2[v * n2 + (c / n3) % n2];
        else val = r3[v * n3 + c % n3];
        out[i] = val;
    }
}

int launch_acquisition_products(float* out, const float* r0, const float* r1, const float* r2, const float* r3, int64_t nvox,
                                int n1, int n2, int n3, hipStream_t st) {
    const int64_t total = nvox * 4 * n1 * n2 * n3;
    if (total == 0) return 0;
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(acquisition_products_kernel, dim3(blocks_for(total, 256, 1 << 16)), dim3(256), 0, st, out, r0, r1, r2, r3,
                       nvox, n1, n2, n3);
    INR_LAUNCH_CHECK();
    return 0;
}

int launch_mul(float* out, const float* a, const float* b, int64_t count, hipStream_t st) {
    if (count == 0) return 0;
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(mul_kernel, dim3(blocks_for(count, 256, 256 * 32)), dim3(256), 0, st, out, a, b, count);
    INR_LAUNCH_CHECK();
    return 0;
}

int launch_sincos_probe(float* s, float* c, const float* x, int64_t n, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sincos_probe_kernel, dim3(blocks_for(n, 256, 1 << 30)), dim3(256), 0, st, s, c, x, n);
    INR_LAUNCH_CHECK();
    return 0;
}

}  // namespace inr
