// RAMS multi-image super-resolution network, forward pass:
//   /root/reference/multi-image-super-resolution/utils/network.py:91-155 (graph), :42-63 RFAB, :65-87 RTAB,
//   :37-39 reflective padding, :21-27 normalize/denormalize; utils/prediction.py:76-83 predict_tensor.
// Activations are NDHWC fp32 exactly as the reference's Keras graph keeps them: [B][H'][W'][T][32].
//
// The work is 25 + 4 3x3x3 convolutions 32 -> 32 (>= 99 % of the 265 GFLOP): `conv3d_c32_mfma_kernel` runs them
// as an implicit GEMM on v_mfma_f32_32x32x2_f32 -- M = 32 output voxels per wave tile, N = 32 output channels,
// K = 27 taps x 32 input channels.  The whole folded kernel (27x32x32 fp32 = 108 KB) sits in LDS for the life of
// the block (8 waves / CU share it); the A operand is read straight from global memory (each voxel's 32 channels
// are one 128-B line, re-used by 27 taps out of L1/L2) through a buffer resource, so 'same' zero padding and the
// ragged last tile are plain out-of-range reads that return 0 -- no halo staging, no branches.  Bias, ReLU and the
// per-channel sums needed by the attention block's global average pool are fused into the epilogue.
// Everything else (stem 1 -> 32 conv, 1x1x1 squeeze/excite gates, reflect padding, the 2-D branch, pixel shuffle)
// is < 1 % of the work and uses simple direct kernels.
// Weight normalisation (tfa WeightNormalization, g*v/||v||) is folded into the kernels when the parameters are
// packed (host side, once per model) -- the packed layouts are documented in include/inrhip.h.
#include "common.h"

namespace inr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int RC = 32;                      // feature channels of the 3-D trunk (network.py: filters = 32)
constexpr int CONV_THREADS = 512;           // 8 waves share one LDS copy of the kernel
constexpr int CONV_W_FLOATS = 27 * RC * RC;

struct Conv3dParams {
    const float* x;       // [B][D1][D2][D3][32]
    float* y;             // [B][O1][O2][O3][y_cstride]
    const float* w;       // [27][32 cin][32 cout] (cout zero-padded to 32)
    const float* bias;    // [32]
    float* chan_slab;     // nullable: [B][waves_per_b][32] per-channel sums of the (post-activation) output
    int B, D1, D2, D3, O1, O2, O3;
    int pad;              // 1 = 'same' (zero padding), 0 = 'valid'
    int cout, y_cstride;  // channels actually stored / channel stride of y
    int relu;
    int tiles_per_b, waves_per_b;
    long long x_elems_per_b;
};

// MT = output tiles (32 voxels each) a wave works on at once; they share every B fragment read from LDS.
// Per tap: a tap's input offset is the lane's base offset (VGPR, once per tile) plus a wave-uniform
// delta (one add), and a tap's in-bounds predicate is one bit of a per-lane 27-bit
// mask built once per tile -- the f32 MFMA shares the VALU lanes, so per-tap address arithmetic is kept to a
// bit test and a select.
template <int MT>
__global__ void __launch_bounds__(CONV_THREADS, 2) conv3d_c32_mfma_kernel(const Conv3dParams p) {
    __shared__ __attribute__((aligned(16))) float sw[CONV_W_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int b = blockIdx.y;
    for (int i = tid; i < CONV_W_FLOATS / 4; i += CONV_THREADS)
        reinterpret_cast<f32x4*>(sw)[i] = reinterpret_cast<const f32x4*>(p.w)[i];
    __syncthreads();

    const long long xbytes = p.x_elems_per_b * 4;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (long long)b * p.x_elems_per_b), 0, (int)(unsigned)(xbytes > 0x7FFFFFFFll ? 0x7FFFFFFFll : xbytes),
        0x00020000);
    const int ovox = p.O1 * p.O2 * p.O3;
    const float bias = p.bias[l32];
    float csum = 0.f;
    // The two waves that share a SIMD are `wave` and `wave + 4` of a block.  Number the waves so that ids below
    // half the total are the FIRST wave of every SIMD: left-over tiles (tiles_per_b mod waves) then land on
    // different SIMDs instead of doubling up on some.
    const int half = p.waves_per_b >> 1;
    const int wid = (wave < 4) ? (blockIdx.x * 4 + wave) : (half + blockIdx.x * 4 + (wave - 4));
    const int groups = (p.tiles_per_b + MT - 1) / MT;

    for (int grp = wid; grp < groups; grp += p.waves_per_b) {
        int base[MT];
        unsigned mask[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int v = (grp * MT + m) * 32 + l32;
            const bool vvalid = v < ovox;
            const int o3 = v % p.O3, o2 = (v / p.O3) % p.O2, o1 = v / (p.O3 * p.O2);
            base[m] = ((((o1 - p.pad) * p.D2 + (o2 - p.pad)) * p.D3 + (o3 - p.pad)) * RC + 4 * h) * 4;
            // 27-bit tap mask from three 3-bit per-axis masks (bit d = input index o + d - pad is inside the axis)
            unsigned m1 = 0, m2 = 0, m3 = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                m1 |= ((unsigned)(o1 + d - p.pad) < (unsigned)p.D1) ? (1u << d) : 0u;
                m2 |= ((unsigned)(o2 + d - p.pad) < (unsigned)p.D2) ? (1u << d) : 0u;
                m3 |= ((unsigned)(o3 + d - p.pad) < (unsigned)p.D3) ? (1u << d) : 0u;
            }
            unsigned mk = 0;
#pragma unroll
            for (int d1 = 0; d1 < 3; ++d1)
#pragma unroll
                for (int d2 = 0; d2 < 3; ++d2)
                    mk |= (((m1 >> d1) & (m2 >> d2) & 1u) ? m3 : 0u) << (9 * d1 + 3 * d2);
            if (!vvalid) mk = 0;
            mask[m] = mk;
        }
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

        f32x4 a0[MT][4], a1[MT][4];   // rotating operand registers (no register moves between taps)
        auto load_tap = [&](int tap, f32x4(&a)[MT][4]) {
            const int delta = (((tap / 9) * p.D2 + (tap / 3) % 3) * p.D3 + tap % 3) * RC * 4;   // wave-uniform
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                // out-of-range taps read from an offset past the SRD -> 0 ('same' zero padding, ragged last tile)
                // (the whole offset goes into the VGPR: a raw buffer's range check does not see soffset, and base[m]
                //  alone is negative on the low borders)
                const int off = ((mask[m] >> tap) & 1u) ? base[m] + delta : 0x7F000000;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    a[m][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, off + q * 32, 0, 0));
            }
        };
        auto mma_tap = [&](int tap, const f32x4(&a)[MT][4]) {
            const float* wt = sw + tap * RC * RC + l32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float bv = wt[(8 * q + 4 * h + s) * RC];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q][s], bv, acc[m], 0, 0, 0);
                }
            }
        };
        // operands are fetched two taps ahead of their MFMAs (three rotating register sets, 27 = 9 x 3 taps)
        f32x4 a2[MT][4];
        load_tap(0, a0);
        load_tap(1, a1);
#pragma unroll 1   // rolled on purpose: unrolled, hipcc hoists all 27x16 LDS weight reads and spills
        for (int tap = 0; tap < 24; tap += 3) {
            load_tap(tap + 2, a2);
            mma_tap(tap, a0);
            load_tap(tap + 3, a0);
            mma_tap(tap + 1, a1);
            load_tap(tap + 4, a1);
            mma_tap(tap + 2, a2);
        }
        load_tap(26, a2);
        mma_tap(24, a0);
        mma_tap(25, a1);
        mma_tap(26, a2);

        // epilogue: C/D map col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4h (voxel inside the tile)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int tile = grp * MT + m;
            float tsum = 0.f;
            float* yb = p.y + ((long long)b * ovox + (long long)tile * 32) * p.y_cstride + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                float o = acc[m][r] + bias;
                if (p.relu) o = fmaxf(o, 0.f);
                if (tile * 32 + row < ovox) {
                    tsum += o;
                    if (l32 < p.cout) yb[(long long)row * p.y_cstride] = o;
                }
            }
            if (p.chan_slab) csum += tsum + __shfl_xor(tsum, 32, 64);
        }
    }
    if (p.chan_slab && lane < 32) p.chan_slab[((long long)b * p.waves_per_b + wid) * RC + lane] = csum;
}

#include "rams_h3.inc"

// ---- stem: Conv3D 1 -> 32, 3x3x3 'same' (network.py:119) ----------------------------------------------------
// x [B][D1][D2][D3], w [27][32], y [B][D1][D2][D3][32]; one thread per (voxel, cout)
__global__ void __launch_bounds__(256) conv3d_c1_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int B, int D1, int D2, int D3, unsigned* amax = nullptr) {
    const long long total = (long long)B * D1 * D2 * D3 * RC;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) {
        if (amax) r3_block_amax(0.f, amax);     // (the whole block takes part)
        return;
    }
    const int c = (int)(i % RC);
    long long v = i / RC;
    const int i3 = (int)(v % D3); v /= D3;
    const int i2 = (int)(v % D2); v /= D2;
    const int i1 = (int)(v % D1);
    const int b = (int)(v / D1);
    float acc = bias[c];
    for (int d1 = 0; d1 < 3; ++d1)
        for (int d2 = 0; d2 < 3; ++d2)
            for (int d3 = 0; d3 < 3; ++d3) {
                const int j1 = i1 + d1 - 1, j2 = i2 + d2 - 1, j3 = i3 + d3 - 1;
                if ((unsigned)j1 < (unsigned)D1 && (unsigned)j2 < (unsigned)D2 && (unsigned)j3 < (unsigned)D3)
                    acc = fmaf(x[(((long long)b * D1 + j1) * D2 + j2) * D3 + j3], w[((d1 * 3 + d2) * 3 + d3) * RC + c], acc);
            }
    y[i] = acc;
    if (amax) r3_block_amax(acc, amax);
}

// ---- attention gate: global average pool finish + 1x1 squeeze (ReLU) + 1x1 excite (sigmoid) ---------------------
// slab [B][nslab][C] partial channel sums -> gate [B][C].  One block per batch element.
__global__ void __launch_bounds__(1024) gate_kernel(float* __restrict__ gate, const float* __restrict__ slab, int nslab,
                                                    float inv_count, const float* __restrict__ wsq /*[C][Cr]*/,
                                                    const float* __restrict__ bsq, const float* __restrict__ wex /*[Cr][C]*/,
                                                    const float* __restrict__ bex, int C, int Cr) {
    // 32 phases x 32 channels; a phase sums every 32nd slab with eight independent accumulators (a batch-1 call has 2,048
    // slabs and ONE block: a serial chain of dependent loads was 59 us), phases combined in fixed order
    __shared__ float part[1024];
    __shared__ float mean[32];
    __shared__ float sq[8];
    const int b = blockIdx.x, t = threadIdx.x;
    const int c = t % 32, ph = t / 32;
    // (eight independent sums per thread since round 4: at batch 1 -- 2,048 slabs -- four chains of sixteen dependent additions behind
    //  their loads were 8.8 us per call, sixteen calls per forward)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float* base = slab + (long long)b * nslab * C + c;
        int s = ph;
        for (; s + 224 < nslab; s += 256) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += base[(long long)(s + 32 * u) * C];
        }
        for (; s < nslab; s += 32) a[0] += base[(long long)s * C];
    }
    part[t] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (t < C) {
        float m = 0.f;
        for (int k = 0; k < 32; ++k) m += part[k * 32 + t];
        mean[t] = m * inv_count;
    }
    __syncthreads();
    if (t < Cr) {
        float s = bsq[t];
        for (int k = 0; k < C; ++k) s = fmaf(mean[k], wsq[k * Cr + t], s);
        sq[t] = fmaxf(s, 0.f);
    }
    __syncthreads();
    if (t < C) {
        float e = bex[t];
        for (int k = 0; k < Cr; ++k) e = fmaf(sq[k], wex[k * C + t], e);
        gate[b * C + t] = 1.0f / (1.0f + expf(-e));
    }
}

// ---- the gate of an attention block BEFORE its second convolution runs ------------------------------------------------------
// gate = f(mean over voxels of c2), c2 = conv2(r1) + b2 ('same', zero padding) -- and that mean is linear in r1:
//   sum_v c2[v][co] = V b2[co] + sum_tap sum_ci W2[tap][ci][co] S[tap][ci],   S[tap][ci] = sum of r1[.][ci] over the box of voxels
// tap (d1, d2, d3) reaches from inside the image: all of an axis for d = 1, all but the last index for d = 0, all but the first for
// d = 2.  The 27 boxes are unions of the 27 classes (first / middle / last index per axis) of a voxel; the class sums of r1 need
// the whole-tensor sum (the first convolution's epilogue leaves it, as it does for the gate of the other form) and the boundary
// classes: the two depth faces of every interior row (2 of D3 voxels) and the thin H / W frame.  With the gate known, the second
// convolution applies it and adds the residual in its epilogue (rams_h3.inc, AUX 4): c2 is never written, the scale_residual pass
// (three tensor passes at 5.3 TB/s = 16 % of a batch-25 forward) is gone.  Equal to the other form up to the rounding of the mean.
constexpr int RCLS_NB = 64;      // blocks per batch element of the class-sum kernel
// part [B][RCLS_NB][27][32] (class = 9 a1 + 3 a2 + a3, a = 0 first / 1 middle / 2 last index; class 13 = all-middle is not formed here)
__global__ void __launch_bounds__(256) rams_class_sums_kernel(float* __restrict__ part_out, const float* __restrict__ r1, int D1,
                                                              int D2, int D3) {
    __shared__ float part[8][17][32];
    const int b = blockIdx.y, k = blockIdx.x, c = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const float* img = r1 + (long long)b * D1 * D2 * D3 * RC + c;
    auto vox = [&](int i1, int i2, int t) { return img[(((long long)i1 * D2 + i2) * D3 + t) * RC]; };
    const int n1 = D1 - 2, n2 = D2 - 2;
    float acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
    // interior rows: the two depth faces (32-bit indices: the host checks the image size; four rows' loads in flight, added in order)
    {
        const unsigned nrow = (unsigned)n1 * (unsigned)n2, step = RCLS_NB * 8;
        unsigned r = (unsigned)k * 8 + ph;
        for (; r + 3 * step < nrow; r += 4 * step) {
            float a[4], z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned rr = r + u * step, q1 = rr / (unsigned)n2;
                a[u] = vox(1 + (int)q1, 1 + (int)(rr - q1 * n2), 0);
                z[u] = vox(1 + (int)q1, 1 + (int)(rr - q1 * n2), D3 - 1);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] += a[u];
                acc[1] += z[u];
            }
        }
        for (; r < nrow; r += step) {
            const unsigned q1 = r / (unsigned)n2;
            acc[0] += vox(1 + (int)q1, 1 + (int)(r - q1 * n2), 0);
            acc[1] += vox(1 + (int)q1, 1 + (int)(r - q1 * n2), D3 - 1);
        }
    }
    // the four edges of the H / W frame (corners apart): whole depth rows, split by depth class
    auto row3 = [&](int i1, int i2, float& s0, float& s1, float& s2) {
        s0 += vox(i1, i2, 0);
        for (int t = 1; t < D3 - 1; ++t) s1 += vox(i1, i2, t);
        s2 += vox(i1, i2, D3 - 1);
    };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int len = e < 2 ? n2 : n1;
        for (int q = k * 8 + ph; q < len; q += RCLS_NB * 8) {
            const int i1 = e == 0 ? 0 : e == 1 ? D1 - 1 : 1 + q, i2 = e == 2 ? 0 : e == 3 ? D2 - 1 : 1 + q;
            row3(i1, i2, acc[2 + 3 * e], acc[3 + 3 * e], acc[4 + 3 * e]);
        }
    }
    if (k == 0 && ph < 4) row3((ph & 2) ? D1 - 1 : 0, (ph & 1) ? D2 - 1 : 0, acc[14], acc[15], acc[16]);
#pragma unroll
    for (int i = 0; i < 17; ++i) part[ph][i][c] = acc[i];
    __syncthreads();
    float* out = part_out + ((long long)b * RCLS_NB + k) * 27 * RC;
    for (int i = threadIdx.x; i < 27 * RC; i += 256) {
        const int cls = i / RC, cc = i % RC, a1 = cls / 9, a2 = (cls / 3) % 3, a3 = cls % 3;
        float v = 0.f;
        int slot = -1;        // which of the 14 per-block sums this class is; corners sit with ONE phase group each
        if (a1 == 1 && a2 == 1) slot = a3 == 0 ? 0 : a3 == 2 ? 1 : -1;
        else if (a1 == 0 && a2 == 1) slot = 2 + a3;
        else if (a1 == 2 && a2 == 1) slot = 5 + a3;
        else if (a1 == 1 && a2 == 0) slot = 8 + a3;
        else if (a1 == 1 && a2 == 2) slot = 11 + a3;
        if (slot >= 0) {
            for (int g = 0; g < 8; ++g) v += part[g][slot][cc];
        } else if (!(a1 == 1 && a2 == 1)) {
            v = part[(a1 ? 2 : 0) + (a2 ? 1 : 0)][14 + a3][cc];          // (zeros in every block but the first)
        }
        out[i] = v;
    }
}

// gate [B][C] from the class sums; one block of 1,024 threads per batch element.  tot_slab [B][nslab][32]: partial whole-tensor sums
// of r1 (the convolution's channel slabs); w2 [27][32][32], b2 [32]: the SECOND convolution's kernel and bias
__global__ void __launch_bounds__(1024) gate_pre_kernel(float* __restrict__ gate, const float* __restrict__ tot_slab, int nslab,
                                                        const float* __restrict__ cls_part, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float inv_count,
                                                        const float* __restrict__ wsq, const float* __restrict__ bsq,
                                                        const float* __restrict__ wex, const float* __restrict__ bex, int Cr) {
    __shared__ float Cs[27][32], Ss[27][32], part[32][32], tot[32], mean[32], sq[8];
    const int b = blockIdx.x, t = threadIdx.x, c = t % 32, ph = t / 32;
    {   // whole-tensor sums (gate_kernel's first phase)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float* base = tot_slab + (long long)b * nslab * RC + c;
        int s = ph;
        for (; s + 96 < nslab; s += 128) {
            a0 += base[(long long)s * RC];
            a1 += base[(long long)(s + 32) * RC];
            a2 += base[(long long)(s + 64) * RC];
            a3 += base[(long long)(s + 96) * RC];
        }
        for (; s < nslab; s += 32) a0 += base[(long long)s * RC];
        part[ph][c] = (a0 + a1) + (a2 + a3);
    }
    if (t < 27 * RC) {   // boundary class sums over the blocks of the class-sum kernel, eight loads in flight, fixed order
        const float* src = cls_part + (long long)b * RCLS_NB * 27 * RC + t;
        float v = 0.f;
        for (int k = 0; k < RCLS_NB; k += 8) {
            float u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = src[(long long)(k + q) * 27 * RC];
#pragma unroll
            for (int q = 0; q < 8; ++q) v += u[q];
        }
        Cs[t / RC][t % RC] = v;
    }
    __syncthreads();
    if (t < RC) {
        float m = 0.f;
        for (int k = 0; k < 32; ++k) m += part[k][t];
        float others = 0.f;
        for (int cls = 0; cls < 27; ++cls)
            if (cls != 13) others += Cs[cls][t];
        tot[t] = m;
        Cs[13][t] = m - others;
    }
    __syncthreads();
    if (t < 27 * RC) {   // box sums: axis classes {0, 1} for d = 0, all for d = 1, {1, 2} for d = 2
        const int tap = t / RC, ci = t % RC, d1 = tap / 9, d2 = (tap / 3) % 3, d3 = tap % 3;
        float v = 0.f;
        for (int a1 = (d1 == 2); a1 <= 2 - (d1 == 0); ++a1)
            for (int a2 = (d2 == 2); a2 <= 2 - (d2 == 0); ++a2)
                for (int a3 = (d3 == 2); a3 <= 2 - (d3 == 0); ++a3) v += Cs[a1 * 9 + a2 * 3 + a3][ci];
        Ss[tap][ci] = v;
    }
    __syncthreads();
    {   // sum_tap sum_ci W2[tap][ci][co] S[tap][ci]: 864 terms over 32 phase groups of 27
        const float* Sf = &Ss[0][0];
        float v = 0.f;
        for (int j = 0; j < 27; ++j) v = fmaf(w2[(long long)(ph * 27 + j) * RC + c], Sf[ph * 27 + j], v);
        part[ph][c] = v;
    }
    __syncthreads();
    if (t < RC) {
        float m = 0.f;
        for (int k = 0; k < 32; ++k) m += part[k][t];
        mean[t] = fmaf(m, inv_count, b2[t]);
    }
    __syncthreads();
    if (t < Cr) {
        float s = bsq[t];
        for (int k = 0; k < RC; ++k) s = fmaf(mean[k], wsq[k * Cr + t], s);
        sq[t] = fmaxf(s, 0.f);
    }
    __syncthreads();
    if (t < RC) {
        float e = bex[t];
        for (int k = 0; k < Cr; ++k) e = fmaf(sq[k], wex[k * RC + t], e);
        gate[b * RC + t] = 1.0f / (1.0f + expf(-e));
    }
}

// out = y * gate[b][c] + res   over [B][per_b voxels][C]
__global__ void __launch_bounds__(256) scale_residual_kernel(float* __restrict__ out, const float* __restrict__ y,
                                                             const float* __restrict__ gate, const float* __restrict__ res,
                                                             long long per_b, int C, long long total,
                                                             unsigned* amax = nullptr) {
    // grid-stride (launch with nblk_capped): float4 where the channel count allows; one block maximum per block
    float m = 0.f;
    if ((C & 3) == 0) {
        const long long n4 = total >> 2, pb4 = per_b * C >> 2;
        const int c4n = C >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const int c4 = (int)(i % c4n);
            const int b = (int)(i / pb4);
            const float4 g = reinterpret_cast<const float4*>(gate + (long long)b * C)[c4];
            const float4 a = reinterpret_cast<const float4*>(y)[i];
            const float4 r = reinterpret_cast<const float4*>(res)[i];
            const float4 v = make_float4(fmaf(a.x, g.x, r.x), fmaf(a.y, g.y, r.y), fmaf(a.z, g.z, r.z), fmaf(a.w, g.w, r.w));
            reinterpret_cast<float4*>(out)[i] = v;
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int c = (int)(i % C);
            const int b = (int)(i / (per_b * C));
            const float v = fmaf(y[i], gate[b * C + c], res[i]);
            out[i] = v;
            m = fmaxf(m, fabsf(v));
        }
    }
    if (amax) r3_block_amax(m, amax);
}

// the same for C = 32 with 16-byte accesses; one batch element per blockIdx.y (no 64-bit divisions)
__global__ void __launch_bounds__(256) scale_residual_c32_kernel(float* __restrict__ out, const float* __restrict__ y,
                                                                 const float* __restrict__ gate, const float* __restrict__ res,
                                                                 long long per_b4 /* float4 per batch element */,
                                                                 unsigned* amax) {
    const int b = blockIdx.y;
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y) + (long long)b * per_b4;
    const f32x4* r4 = reinterpret_cast<const f32x4*>(res) + (long long)b * per_b4;
    f32x4* o4 = reinterpret_cast<f32x4*>(out) + (long long)b * per_b4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + b * RC + 4 * (threadIdx.x & 7));     // 256 % 8 == 0: fixed per thread
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_b4; i += (long long)gridDim.x * 256) {
        const f32x4 yv = y4[i], rv = r4[i];
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = fmaf(yv[q], g[q], rv[q]);
            m = fmaxf(m, fabsf(v[q]));
        }
        o4[i] = v;
    }
    if (amax) r3_block_amax(m, amax);
}

__global__ void __launch_bounds__(256) add_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                  const float* __restrict__ b, long long total, unsigned* amax = nullptr) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (i < total) {
        v = a[i] + b[i];
        out[i] = v;
    }
    if (amax) r3_block_amax(v, amax);
}

// (x - MEAN)/STD  (network.py:21-23)
__global__ void __launch_bounds__(256) normalize_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                        long long total, float mean, float stdv) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = (x[i] - mean) / stdv;
}

// tf.pad REFLECT by 1 on axes 1,2 of [B][D1][D2][inner]  (network.py:37-39,145)
__global__ void __launch_bounds__(256) reflect_pad_kernel(float* __restrict__ out, const float* __restrict__ x, int B,
                                                          int D1, int D2, int inner) {
    const long long total = (long long)B * (D1 + 2) * (D2 + 2) * inner;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % inner);
    long long v = i / inner;
    const int o2 = (int)(v % (D2 + 2)); v /= (D2 + 2);
    const int o1 = (int)(v % (D1 + 2));
    const int b = (int)(v / (D1 + 2));
    int i1 = o1 - 1, i2 = o2 - 1;
    i1 = i1 < 0 ? -i1 : (i1 >= D1 ? 2 * D1 - 2 - i1 : i1);
    i2 = i2 < 0 ? -i2 : (i2 >= D2 ? 2 * D2 - 2 - i2 : i2);
    out[i] = x[(((long long)b * D1 + i1) * D2 + i2) * inner + k];
}

// 16-byte versions for the 5-D trunk tensors (inner = T * 32 floats): one (b, o1, o2) row per block column
__global__ void __launch_bounds__(256) reflect_pad_v4_kernel(float* __restrict__ out, const float* __restrict__ x, int D1,
                                                             int D2, int inner4) {
    const int o2 = blockIdx.x % (D2 + 2), o1 = blockIdx.x / (D2 + 2), b = blockIdx.y;
    int i1 = o1 - 1, i2 = o2 - 1;
    i1 = i1 < 0 ? -i1 : (i1 >= D1 ? 2 * D1 - 2 - i1 : i1);
    i2 = i2 < 0 ? -i2 : (i2 >= D2 ? 2 * D2 - 2 - i2 : i2);
    const f32x4* src = reinterpret_cast<const f32x4*>(x) + (((long long)b * D1 + i1) * D2 + i2) * inner4;
    f32x4* dst = reinterpret_cast<f32x4*>(out) + (((long long)b * (D1 + 2) + o1) * (D2 + 2) + o2) * inner4;
    for (int k = threadIdx.x; k < inner4; k += 256) dst[k] = src[k];
}

// The stem by rows: one thread = four output channels of ALL D3 voxels of a (b, i1, i2) row.  Per (d1, d2) it loads the D3 + 2
// values of the neighbouring row once (zeros past the ends) and three weight quads, and every output takes its taps in the order
// (d1, d2, d3) conv3d_c1_v4_kernel takes them -- the same bits (a tap outside the image adds 0 * w there, nothing here: the same sum)
// with 14 loads per voxel instead of 54 (the per-voxel kernel was bound by its L1 round trips: 0.62 ms per 25 stacks for 0.9 GB).
template <int TD>      // D3 (compile time: the window lives in registers)
__global__ void __launch_bounds__(256) conv3d_c1_rows_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                             const float* __restrict__ w, const float* __restrict__ bias, int B,
                                                             int D1, int D2, unsigned* amax, float* __restrict__ y2) {
    const long long rows = (long long)B * D1 * D2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float m = 0.f;
    if (i < rows * (RC / 4)) {
        const int c4 = (int)(i % (RC / 4));
        long long v = i / (RC / 4);
        const int i2 = (int)(v % D2); v /= D2;
        const int i1 = (int)(v % D1);
        const int b = (int)(v / D1);
        f32x4 acc[TD];
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * c4);
#pragma unroll
        for (int t = 0; t < TD; ++t) acc[t] = bv;
#pragma unroll
        for (int d1 = 0; d1 < 3; ++d1)
#pragma unroll
            for (int d2 = 0; d2 < 3; ++d2) {
                const int j1 = i1 + d1 - 1, j2 = i2 + d2 - 1;
                if ((unsigned)j1 >= (unsigned)D1 || (unsigned)j2 >= (unsigned)D2) continue;      // (the per-voxel kernel skips these taps too)
                const float* xr = x + (((long long)b * D1 + j1) * D2 + j2) * TD;
                float xv[TD];
#pragma unroll
                for (int t = 0; t < TD; ++t) xv[t] = xr[t];
                f32x4 wv[3];
#pragma unroll
                for (int d3 = 0; d3 < 3; ++d3) wv[d3] = *reinterpret_cast<const f32x4*>(w + ((d1 * 3 + d2) * 3 + d3) * RC + 4 * c4);
#pragma unroll
                for (int t = 0; t < TD; ++t)
#pragma unroll
                    for (int d3 = 0; d3 < 3; ++d3) {
                        const int j3 = t + d3 - 1;
                        if (j3 < 0 || j3 >= TD) continue;
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[t][q] = fmaf(xv[j3], wv[d3][q], acc[t][q]);
                    }
            }
        const long long o4 = ((((long long)b * D1 + i1) * D2 + i2) * TD) * (RC / 4) + c4;
#pragma unroll
        for (int t = 0; t < TD; ++t) {
            reinterpret_cast<f32x4*>(y)[o4 + (long long)t * (RC / 4)] = acc[t];
            if (y2) reinterpret_cast<f32x4*>(y2)[o4 + (long long)t * (RC / 4)] = acc[t];
#pragma unroll
            for (int q = 0; q < 4; ++q) m = fmaxf(m, fabsf(acc[t][q]));
        }
    }
    if (amax) r3_block_amax(m, amax);
}
// the frame of a reflect-padded image whose interior is already in place (written there by the convolution, Conv3dLdsParams::y_pad):
// buf [B][D1 + 2][D2 + 2][inner]; one frame voxel per block column -- top and bottom rows first, then the two side columns
__global__ void __launch_bounds__(256) reflect_border_v4_kernel(float* __restrict__ buf, int D1, int D2, int inner4) {
    const int f = blockIdx.x, b = blockIdx.y, W2 = D2 + 2;
    int o1, o2;
    if (f < 2 * W2) {
        o1 = f < W2 ? 0 : D1 + 1;
        o2 = f < W2 ? f : f - W2;
    } else {
        const int gidx = f - 2 * W2;
        o1 = 1 + gidx % D1;
        o2 = gidx < D1 ? 0 : D2 + 1;
    }
    int i1 = o1 - 1, i2 = o2 - 1;
    i1 = i1 < 0 ? -i1 : (i1 >= D1 ? 2 * D1 - 2 - i1 : i1);
    i2 = i2 < 0 ? -i2 : (i2 >= D2 ? 2 * D2 - 2 - i2 : i2);
    f32x4* img = reinterpret_cast<f32x4*>(buf) + (long long)b * (D1 + 2) * W2 * inner4;
    const f32x4* src = img + ((long long)(i1 + 1) * W2 + (i2 + 1)) * inner4;
    f32x4* dst = img + ((long long)o1 * W2 + o2) * inner4;
    for (int k = threadIdx.x; k < inner4; k += 256) dst[k] = src[k];
}

__global__ void __launch_bounds__(256) add_v4_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                     const float* __restrict__ b, long long total4, unsigned* amax) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[i], bv = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = av[q] + bv[q];
            m = fmaxf(m, fabsf(v[q]));
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
    if (amax) r3_block_amax(m, amax);
}

// stem with four output channels per thread (one 16-byte store; the 27 inputs are read once per four outputs)
// (y2, nullable: a second copy of the output -- the long skip keeps the stem output; writing it here saves a device-to-device copy)
__global__ void __launch_bounds__(256) conv3d_c1_v4_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int B, int D1, int D2, int D3, unsigned* amax, float* __restrict__ y2 = nullptr) {
    const long long total4 = (long long)B * D1 * D2 * D3 * (RC / 4);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float m = 0.f;
    if (i < total4) {
        const int c4 = (int)(i % (RC / 4));
        long long v = i / (RC / 4);
        const int i3 = (int)(v % D3); v /= D3;
        const int i2 = (int)(v % D2); v /= D2;
        const int i1 = (int)(v % D1);
        const int b = (int)(v / D1);
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + 4 * c4);
        for (int d1 = 0; d1 < 3; ++d1)
            for (int d2 = 0; d2 < 3; ++d2)
                for (int d3 = 0; d3 < 3; ++d3) {
                    const int j1 = i1 + d1 - 1, j2 = i2 + d2 - 1, j3 = i3 + d3 - 1;
                    if ((unsigned)j1 < (unsigned)D1 && (unsigned)j2 < (unsigned)D2 && (unsigned)j3 < (unsigned)D3) {
                        const float xv = x[(((long long)b * D1 + j1) * D2 + j2) * D3 + j3];
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ((d1 * 3 + d2) * 3 + d3) * RC + 4 * c4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] = fmaf(xv, wv[q], acc[q]);
                    }
                }
        reinterpret_cast<f32x4*>(y)[i] = acc;
        if (y2) reinterpret_cast<f32x4*>(y2)[i] = acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) m = fmaxf(m, fabsf(acc[q]));
    }
    if (amax) r3_block_amax(m, amax);
}

// ---- 2-D branch (RTAB on the 9 normalised inputs, network.py:145-148): direct kernels ----------------------------
// x [B][D1][D2][Cin], w [9 taps][Cin][Cout], y [B][O1][O2][Cout]; thread per (pixel, cout)
// the same convolution, one thread per PIXEL and all CO output channels (the 9 -> 9 convolutions of the 2-D branch): the kernel
// [9][CI][CO] sits in LDS (every lane reads the same word: a broadcast), a pixel's 9 x CI inputs are loaded once instead of once
// per output channel -- conv2d_direct_kernel measured 0.25 ms per launch at batch 25 for 0.3 GFLOP
template <int CI, int CO>
__global__ void __launch_bounds__(256) conv2d_pixel_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                           const float* __restrict__ w, const float* __restrict__ bias, int B, int D1,
                                                           int D2, int pad, int relu) {
    __shared__ float ws[9 * CI * CO + CO];
    for (int i = threadIdx.x; i < 9 * CI * CO; i += 256) ws[i] = w[i];
    if (threadIdx.x < CO) ws[9 * CI * CO + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int O1 = D1 + 2 * pad - 2, O2 = D2 + 2 * pad - 2;
    const long long total = (long long)B * O1 * O2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int o2 = (int)(i % O2);
        const long long v = i / O2;
        const int o1 = (int)(v % O1), b = (int)(v / O1);
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = ws[9 * CI * CO + c];
#pragma unroll
        for (int d1 = 0; d1 < 3; ++d1)
#pragma unroll
            for (int d2 = 0; d2 < 3; ++d2) {
                const int j1 = o1 + d1 - pad, j2 = o2 + d2 - pad;
                if ((unsigned)j1 < (unsigned)D1 && (unsigned)j2 < (unsigned)D2) {
                    const float* xp = x + (((long long)b * D1 + j1) * D2 + j2) * CI;
                    const float* wp = ws + (d1 * 3 + d2) * CI * CO;
#pragma unroll
                    for (int k = 0; k < CI; ++k) {
                        const float xv = xp[k];            // (same accumulation order per output as conv2d_direct_kernel: taps, then k)
#pragma unroll
                        for (int c = 0; c < CO; ++c) acc[c] = fmaf(xv, wp[k * CO + c], acc[c]);
                    }
                }
            }
        float* yp = y + i * CO;
#pragma unroll
        for (int c = 0; c < CO; ++c) yp[c] = relu ? fmaxf(acc[c], 0.f) : acc[c];
    }
}
// launcher: the pixel kernel for the 9 -> 9 shape, the per-output kernel otherwise
__global__ void __launch_bounds__(256) conv2d_direct_kernel(float* y, const float* x, const float* w, const float* bias, int B, int D1,
                                                            int D2, int Cin, int Cout, int pad, int relu);
static void launch_conv2d(float* y, const float* x, const float* w, const float* bias, int B, int D1, int D2, int Cin, int Cout,
                          int pad, int relu, hipStream_t st) {
    const int O1 = D1 + 2 * pad - 2, O2 = D2 + 2 * pad - 2;
    if (Cin == 9 && Cout == 9) {
        long long g = ((long long)B * O1 * O2 + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL((conv2d_pixel_kernel<9, 9>), dim3((unsigned)g), dim3(256), 0, st, y, x, w, bias, B, D1, D2, pad, relu);
    } else {
        hipLaunchKernelGGL(conv2d_direct_kernel, dim3((unsigned)(((long long)B * O1 * O2 * Cout + 255) / 256)), dim3(256), 0, st, y, x, w,
                           bias, B, D1, D2, Cin, Cout, pad, relu);
    }
}
__global__ void __launch_bounds__(256) conv2d_direct_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            int B, int D1, int D2, int Cin, int Cout, int pad, int relu) {
    const int O1 = D1 + 2 * pad - 2, O2 = D2 + 2 * pad - 2;
    const long long total = (long long)B * O1 * O2 * Cout;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cout);
    long long v = i / Cout;
    const int o2 = (int)(v % O2); v /= O2;
    const int o1 = (int)(v % O1);
    const int b = (int)(v / O1);
    float acc = bias[c];
    for (int d1 = 0; d1 < 3; ++d1)
        for (int d2 = 0; d2 < 3; ++d2) {
            const int j1 = o1 + d1 - pad, j2 = o2 + d2 - pad;
            if ((unsigned)j1 < (unsigned)D1 && (unsigned)j2 < (unsigned)D2) {
                const float* xp = x + (((long long)b * D1 + j1) * D2 + j2) * Cin;
                const float* wp = w + (d1 * 3 + d2) * Cin * Cout + c;
                for (int k = 0; k < Cin; ++k) acc = fmaf(xp[k], wp[k * Cout], acc);
            }
        }
    y[i] = relu ? fmaxf(acc, 0.f) : acc;
}

// slab[b][blk][C] = partial channel sums of x [B][per_b][C]  (C <= 32; feeds gate_kernel)
__global__ void __launch_bounds__(256) chan_partial_kernel(float* __restrict__ slab, const float* __restrict__ x,
                                                           long long per_b, int C, int nblk) {
    __shared__ float part[256];
    const int b = blockIdx.y, t = threadIdx.x;
    const int c = t % 32, ph = t / 32;
    float acc = 0.f;
    if (c < C)
        for (long long v = (long long)blockIdx.x * 8 + ph; v < per_b; v += (long long)nblk * 8)
            acc += x[((long long)b * per_b + v) * C + c];
    part[t] = acc;
    __syncthreads();
    if (t < C) {
        float s = 0.f;
        for (int k = 0; k < 8; ++k) s += part[k * 32 + t];
        slab[((long long)b * nblk + blockIdx.x) * C + t] = s;
    }
}

// final assembly (network.py:140-152 + prediction.py:76-83):
// out[b][s*h+i][s*w+j] = (up[b][h][w][i*s+j] + glob[b][h][w][i*s+j]) * STD + MEAN ; optional clip [0, 2^16] + round-half-even
__global__ void __launch_bounds__(256) shuffle_sum_kernel(float* __restrict__ out, const float* __restrict__ up,
                                                          const float* __restrict__ glob, int B, int H, int W, int s,
                                                          float mean, float stdv, int clip_round) {
    const long long total = (long long)B * H * s * W * s;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ow = (int)(i % (W * s));
    long long v = i / (W * s);
    const int oh = (int)(v % (H * s));
    const int b = (int)(v / (H * s));
    const int hh = oh / s, ii = oh % s, ww = ow / s, jj = ow % s;
    const long long src = (((long long)b * H + hh) * W + ww) * (s * s) + ii * s + jj;
    float o = (up[src] + glob[src]) * stdv + mean;
    if (clip_round) o = rintf(fminf(fmaxf(o, 0.f), 65536.f));
    out[i] = o;
}

// =============================== host orchestration ====================================================
static inline unsigned nblk(long long total) { return (unsigned)((total + 255) / 256); }
static inline unsigned nblk_capped(long long total) { const unsigned b = nblk(total); return b < 2048u ? b : 2048u; }   // grid-stride kernels
// debug key 24: 1 = the round-4 fusions and kernels that keep every bit -- epilogue operands of the staged convolution (AUX: default
// kernel only), padded outputs, the stem by rows; 0 = the separate passes / per-voxel stem they replaced (the tests compare);
// 2 (default) = also the inference gate computed ahead of the second convolution (gate_pre_kernel: equal up to the rounding of a mean)
tune_int g_rams_epi_fuse{2};
tune_int g_rams_pregate_min_vox{600000};   // debug key 26: fewest voxels (batch x image) an attention block has for the gate-ahead form
// the stem launch: rows kernel for the depths the network uses, the per-voxel kernel otherwise (a, w, bias 16-byte aligned)
static void launch_stem(float* y, const float* x, const float* w, const float* bias, int B, int D1, int D2, int D3, unsigned* amax,
                        float* y2, hipStream_t st) {
    const long long row_threads = (long long)B * D1 * D2 * (RC / 4);
    const dim3 rgrid((unsigned)((row_threads + 255) / 256));
    if (g_rams_epi_fuse == 0) D3 = -D3;      // (no rows kernel for a negative depth)
    if (D3 == 9) hipLaunchKernelGGL(conv3d_c1_rows_kernel<9>, rgrid, dim3(256), 0, st, y, x, w, bias, B, D1, D2, amax, y2);
    else if (D3 == 7) hipLaunchKernelGGL(conv3d_c1_rows_kernel<7>, rgrid, dim3(256), 0, st, y, x, w, bias, B, D1, D2, amax, y2);
    else if (D3 == 5) hipLaunchKernelGGL(conv3d_c1_rows_kernel<5>, rgrid, dim3(256), 0, st, y, x, w, bias, B, D1, D2, amax, y2);
    else if (D3 == 3) hipLaunchKernelGGL(conv3d_c1_rows_kernel<3>, rgrid, dim3(256), 0, st, y, x, w, bias, B, D1, D2, amax, y2);
    else {
        D3 = D3 < 0 ? -D3 : D3;
        hipLaunchKernelGGL(conv3d_c1_v4_kernel, dim3(nblk((long long)B * D1 * D2 * D3 * (RC / 4))), dim3(256), 0, st, y, x, w, bias, B, D1,
                           D2, D3, amax, y2);
    }
}


struct RamsLayout {
    // offsets (floats) into the packed parameter buffer; see inr_rams_param_offsets
    long long stem_w, stem_b;
    long long total;
};

// packed-parameter walk: conv3d 32->32 block = [27*32*32 w][32 b]; gate = [C*Cr][Cr][Cr*C][C]
struct Cursor {
    const float* base;
    long long off = 0;
    const float* take(long long n) {
        const float* p = base ? base + off : nullptr;     // (a counting walk has no base: no arithmetic on a null pointer)
        off += (n + 3) / 4 * 4;
        return p;
    }
};

long long rams_param_floats(const inr_rams_desc_t* d) {
    Cursor c{nullptr};
    auto conv = [&]() { c.take(CONV_W_FLOATS); c.take(RC); };
    auto gate = [&](int C, int Cr) { c.take((long long)C * Cr); c.take(Cr); c.take((long long)Cr * C); c.take(C); };
    const int Cr = d->filters / d->r;
    c.take(27 * RC); c.take(RC);                       // stem
    for (int i = 0; i < d->n_rfab; ++i) { conv(); conv(); gate(RC, Cr); }
    conv();                                            // trunk
    for (int i = 0; i < d->channels / 3; ++i) { conv(); conv(); gate(RC, Cr); conv(); }
    conv();                                            // up (cout padded to 32)
    const int T = d->channels, Tr = T / d->r > 0 ? T / d->r : 1;
    c.take(9LL * T * T); c.take(T); c.take(9LL * T * T); c.take(T);  // rtab conv1, conv2
    gate(T, Tr);
    c.take(9LL * T * d->scale * d->scale); c.take(d->scale * d->scale);  // global conv
    return c.off;
}

static int conv3d_mfma(const float* x, float* y, const float* w, const float* bias, float* chan_slab, int B, int D1,
                       int D2, int D3, int pad, int cout, int y_cstride, int relu, int waves_per_b, hipStream_t st) {
    Conv3dParams p{};
    p.x = x; p.y = y; p.w = w; p.bias = bias; p.chan_slab = chan_slab;
    p.B = B; p.D1 = D1; p.D2 = D2; p.D3 = D3;
    p.O1 = D1 + 2 * pad - 2; p.O2 = D2 + 2 * pad - 2; p.O3 = D3 + 2 * pad - 2;
    p.pad = pad; p.cout = cout; p.y_cstride = y_cstride; p.relu = relu;
    const int ovox = p.O1 * p.O2 * p.O3;
    p.tiles_per_b = (ovox + 31) / 32;
    p.waves_per_b = waves_per_b;
    p.x_elems_per_b = (long long)D1 * D2 * D3 * RC;
    ProfScope ps(KC_OTHER, st);
    const dim3 grid(waves_per_b / (CONV_THREADS / 64), B);
    // two tiles per wave (shared B fragments) once every wave has at least ~3 pairs to chew on
    if ((long long)p.tiles_per_b >= 6ll * waves_per_b)
        hipLaunchKernelGGL(conv3d_c32_mfma_kernel<2>, grid, dim3(CONV_THREADS), 0, st, p);
    else
        hipLaunchKernelGGL(conv3d_c32_mfma_kernel<1>, grid, dim3(CONV_THREADS), 0, st, p);
    INR_LAUNCH_CHECK();
    return 0;
}

int rams_waves_per_b(int B, int ovox);
long long rams_slab_floats(int B, int ovox_max);
tune_int g_rams_lds_waves{42};  // LDS-staged kernel: 42 = two blocks of 4 waves x 2 tiles per CU, one staged image each (default: 26.6 ms
                            // per 25 stacks); 8 = 8 waves x 1 tile, two images (29.2 ms); 4 = 4 waves x 2 tiles, two images, one block
                            // per CU (one wave per SIMD hides less latency than the shared weight fragments save); 16 = two-pass
tune_int g_rams_force_lds{0};   // (kept for the debug key's bit 2; the LDS-staged kernel is the default at every batch size now)
tune_int g_rams_h3{2};   // inference convolutions: 2 = split-fp16 MFMA, activations staged in LDS (default); 1 = split-fp16,
                     // activations from global; 0 = f32-input MFMA

static int conv3d_h3(const float* x, float* y, const _Float16* planes, const float* bias, float* chan_slab,
                     const unsigned* x_amax, const unsigned* w_amax, unsigned* y_amax, int B, int D1, int D2, int D3, int pad,
                     int cout, int y_cstride, int relu, int waves_per_b, hipStream_t st) {
    Conv3dH3Params p{};
    p.x = x; p.y = y; p.planes = planes; p.bias = bias; p.chan_slab = chan_slab;
    p.x_amax = x_amax; p.w_amax = w_amax; p.y_amax = y_amax;
    p.B = B; p.D1 = D1; p.D2 = D2; p.D3 = D3;
    p.O1 = D1 + 2 * pad - 2; p.O2 = D2 + 2 * pad - 2; p.O3 = D3 + 2 * pad - 2;
    p.pad = pad; p.cout = cout; p.y_cstride = y_cstride; p.relu = relu;
    const int ovox = p.O1 * p.O2 * p.O3;
    p.tiles_per_b = (ovox + 31) / 32;
    p.waves_per_b = waves_per_b;
    p.x_elems_per_b = (long long)D1 * D2 * D3 * RC;
    ProfScope ps(KC_OTHER, st);
    const dim3 grid(waves_per_b / (CONV_THREADS / 64), B);
    if ((long long)p.tiles_per_b >= 6ll * waves_per_b)
        hipLaunchKernelGGL(conv3d_c32_h3_kernel<2>, grid, dim3(CONV_THREADS), 0, st, p);
    else
        hipLaunchKernelGGL(conv3d_c32_h3_kernel<1>, grid, dim3(CONV_THREADS), 0, st, p);
    INR_LAUNCH_CHECK();
    return 0;
}

// may the LDS-staged kernels take this image?  (a 3 x 3 halo of one output column must fit the staged image; one batch element
// must stay below the 1 GiB their 32-bit buffer offsets and range checks assume -- larger images take the other kernels)
static inline bool r3l_fits(int D1, int D2, int D3) {
    return D3 * 9 <= R3L_MAX_HVOX && (long long)(D1 + 2) * (D2 + 2) * (D3 + 2) * RC * 4 < (1ll << 30);
}
// patch of the LDS-staged kernel: the most outputs that fit one round of 8 wave tiles (<= 256) whose halo fits the image
static void r3l_choose_patch(int D3, int O3, int* PO1, int* PO2, int max_hvox = R3L_MAX_HVOX, int max_out = 256) {
    int best = 0, b1 = 1, b2 = 1;
    for (int a = 1; a <= 16; ++a)
        for (int c = a; c <= 32; ++c) {
            if ((a + 2) * (c + 2) * D3 > max_hvox || a * c * O3 > max_out) continue;
            if (a * c > best) { best = a * c; b1 = a; b2 = c; }
        }
    *PO1 = b1;
    *PO2 = b2;
}
static int rams_lds_blocks_per_b(int B, int npatch) {
    int blocks = 256 / B;
    if (blocks < 1) blocks = 1;
    return blocks < npatch ? blocks : npatch;
}

#define R3L_SLAB_GUARD(rows_per_b)                                                                                              \
    INR_REQUIRE(!chan_slab || slab_cap < 0 || (long long)B * (rows_per_b) * RC <= slab_cap, INR_E_INVALID,                       \
                "RAMS convolution: %d channel-sum slabs per batch element x %d do not fit the %lld floats planned for them",    \
                (int)(rows_per_b), B, slab_cap)

static inline bool rams_lds_aux_ok() { return g_rams_lds_waves == 42 && g_rams_epi_fuse != 0; }
extern unsigned long long* g_stamps;   // diagnostic builds (-DR3_STAMPS): inr_debug_set_ptr(0, device buffer of 16 x u64)
static int conv3d_h3_lds(const float* x, float* y, const _Float16* planes, const float* bias, float* chan_slab,
                         const unsigned* x_amax, const unsigned* w_amax, unsigned* y_amax, int B, int D1, int D2, int D3,
                         int pad, int cout, int y_cstride, int relu, int* nslab, hipStream_t st, long long slab_cap = -1,
                         const float* aux = nullptr, int aux_mode = 0, int y_pad = 0, const float* gate = nullptr) {
    INR_REQUIRE(aux_mode == 0 || (aux && rams_lds_aux_ok()), INR_E_INVALID, "RAMS convolution: epilogue operand without the default kernel");
    INR_REQUIRE(y_pad == 0 || (aux_mode == 0 && g_rams_lds_waves == 42), INR_E_INVALID, "RAMS convolution: padded output without the default kernel");
    Conv3dLdsParams p{};
    p.aux = aux;
    p.gate = gate;
    p.y_pad = y_pad;
    INR_REQUIRE(aux_mode != 4 || gate, INR_E_INVALID, "RAMS convolution: the gated form needs its gate");
    p.x = x; p.y = y; p.planes = planes; p.bias = bias; p.chan_slab = chan_slab;
    p.x_amax = x_amax; p.w_amax = w_amax; p.y_amax = y_amax;
    p.B = B; p.D1 = D1; p.D2 = D2; p.D3 = D3;
    p.O1 = D1 + 2 * pad - 2; p.O2 = D2 + 2 * pad - 2; p.O3 = D3 + 2 * pad - 2;
    p.pad = pad; p.cout = cout; p.y_cstride = y_cstride; p.relu = relu;
    const bool two_pass = g_rams_lds_waves == 16;       // 8 waves x 2 tiles, channels in two halves (conv3d_c32_lds2_kernel)
    if (two_pass) r3l_choose_patch(D3, p.O3, &p.PO1, &p.PO2, R3L2_MAX_HVOX, 512);
    else r3l_choose_patch(D3, p.O3, &p.PO1, &p.PO2);
    p.np1 = (p.O1 + p.PO1 - 1) / p.PO1;
    p.np2 = (p.O2 + p.PO2 - 1) / p.PO2;
    p.stamps = g_stamps;
    p.mD3 = r3_magic(D3); p.mHP2 = r3_magic(p.PO2 + 2); p.mO3 = r3_magic(p.O3); p.mPO2 = r3_magic(p.PO2);
    p.mNP2 = r3_magic(p.np2);
    INR_REQUIRE((long long)D1 * D2 * D3 * RC * 4 < (1ll << 30) &&
                    (long long)(p.O1 + 2 * y_pad) * (p.O2 + 2 * y_pad) * p.O3 * y_cstride * 4 < (1ll << 30),
                INR_E_INVALID, "RAMS convolution: an image of %d x %d x %d x 32 floats exceeds the kernel's 1 GiB per image", D1, D2, D3);
    int blocks = rams_lds_blocks_per_b(B, p.np1 * p.np2);
    ProfScope ps(KC_OTHER, st);
    if (g_rams_lds_waves == 42) {                        // two blocks of 4 waves x 2 tiles per CU, one staged image each
        blocks = 512 / B < 1 ? 1 : 512 / B;
        if (blocks > p.np1 * p.np2) blocks = p.np1 * p.np2;
        if (nslab) *nslab = blocks * 4;
        R3L_SLAB_GUARD(blocks * 4);
        if (aux_mode == 1) hipLaunchKernelGGL((conv3d_c32_lds_kernel<4, 2, true, 1>), dim3(blocks, B), dim3(256), 0, st, p);
        else if (aux_mode == 2) hipLaunchKernelGGL((conv3d_c32_lds_kernel<4, 2, true, 2>), dim3(blocks, B), dim3(256), 0, st, p);
        else if (aux_mode == 4) hipLaunchKernelGGL((conv3d_c32_lds_kernel<4, 2, true, 4>), dim3(blocks, B), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3d_c32_lds_kernel<4, 2, true>), dim3(blocks, B), dim3(256), 0, st, p);
    } else if (two_pass) {
        if (nslab) *nslab = blocks * 8;
        R3L_SLAB_GUARD(blocks * 8);
        hipLaunchKernelGGL(conv3d_c32_lds2_kernel, dim3(blocks, B), dim3(512), 0, st, p);
    } else if (g_rams_lds_waves == 8) {
        if (nslab) *nslab = blocks * 8;
        R3L_SLAB_GUARD(blocks * 8);
        hipLaunchKernelGGL((conv3d_c32_lds_kernel<8, 1>), dim3(blocks, B), dim3(512), 0, st, p);
    } else {
        if (nslab) *nslab = blocks * 4;
        R3L_SLAB_GUARD(blocks * 4);
        hipLaunchKernelGGL((conv3d_c32_lds_kernel<4, 2>), dim3(blocks, B), dim3(256), 0, st, p);
    }
    INR_LAUNCH_CHECK();
    return 0;
}
static inline int rams_conv3d_count(const inr_rams_desc_t* d) { return 2 * d->n_rfab + 1 + 3 * (d->channels / 3) + 1; }
constexpr int R3_MAX_CONVS = 96;
struct R3SplitJobs {
    R3SplitJob j[R3_MAX_CONVS];
};
__global__ void __launch_bounds__(512) rams_weight_split_all_kernel(const R3SplitJobs jobs) {
    rams_weight_split_body(jobs.j[blockIdx.x]);
}

#include "rams_wgrad_h3.inc"

// max|x| of a tensor into a slot (the training step's convolution inputs: activations and gradients written by kernels
// that do not track it)
__global__ void __launch_bounds__(256) r3_tensor_amax_kernel(unsigned* __restrict__ slot, const float* __restrict__ x, long long n4) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    r3_block_amax(m, slot);
}

// ---- building blocks of the training step (utils/training.py:193-209) -------------------------------
// data gradient of a 'same' 3x3x3 convolution = the same convolution of dy with the kernel flipped along every axis
// and its channel axes swapped:  wd[tap'][co][ci] = w[26 - tap'][ci][co]
__global__ void __launch_bounds__(256) conv3d_dgrad_weights_kernel(float* __restrict__ wd, const float* __restrict__ w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= CONV_W_FLOATS) return;
    const int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
    wd[i] = w[((26 - tap) * RC + ci) * RC + co];
}

// weight gradient: gw[tap][ci][co] = sum_v x[v + tap - pad][ci] * dy[v][co] -- an MFMA contraction over the voxels
// (v_mfma_f32_32x32x2_f32: M = ci, N = co, K = 2 voxels per instruction).  A block walks a contiguous range of output
// voxels; its 8 waves share the range and split the 27 taps (wave w: taps w, w+8, w+16, w+24), so every wave keeps at
// most four 32x32 accumulators.  Partial sums go to slab[block][27][32][32] and are reduced in a fixed order.
struct Conv3dWgradParams {
    const float* x;    // [B][D1][D2][D3][32]
    const float* dy;   // [B][O1][O2][O3][32]
    float* slab;       // [B * blocks_per_b][27][32][32]
    int B, D1, D2, D3, O1, O2, O3, pad;
    int vox_per_block;   // output voxels of ONE batch element per block (multiple of 8)
};

// grid = (blocks_per_b, B): a block stays inside one batch element, so both operands are read through per-block buffer
// resources with 32-bit offsets, and an out-of-range tap is an offset past the resource (reads 0).  The f32 MFMA shares the
// VALU lanes: address arithmetic is what this loop must not do (at ~106 VALU per voxel pair -- three 3-bit range masks, the
// 27-bit tap mask assembled from them, the offsets -- the kernel ran at 36 % of its MFMA time).  Now:
//   * rows (axis 1) need no check at all: a tap above / below the image is a negative / too large offset, the resource's range
//     check returns 0;
//   * the column / temporal validity of the nine (d2, d3) combinations and the in-plane offset depend on (o2, o3) only: one LDS
//     table entry per in-plane position, built once per block: bits 0-8 the mask (1 = outside), bits 9.. the offset
//     (o2 - pad) * D3 + (o3 - pad) biased by D3 + 1;
//   * per voxel pair and tap: a signed one-bit extract (0 or all ones), an add, an or -- all ones is past every resource.
constexpr int WGRAD_TAB_MAX = 16384;     // in-plane positions O2 * O3 the table may hold (dynamic LDS, 4 bytes each)
__global__ void __launch_bounds__(512, 2) conv3d_c32_wgrad_kernel(const Conv3dWgradParams p) {
    extern __shared__ unsigned tab[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l32 = lane & 31;
    const int b = blockIdx.y;
    const int plane = p.O2 * p.O3;
    const int ovox = p.O1 * plane;
    const int v0 = blockIdx.x * p.vox_per_block, v1 = min(ovox, v0 + p.vox_per_block);
    const long long xb = (long long)p.D1 * p.D2 * p.D3 * RC, yb = (long long)ovox * RC;
    const __amdgpu_buffer_rsrc_t srdx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + b * xb), 0, (int)(xb * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t srdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + b * yb), 0, (int)(yb * 4), 0x00020000);
    for (int i = threadIdx.x; i < plane; i += 512) {
        const int o2 = i / p.O3, o3 = i - o2 * p.O3;
        unsigned mk = 0;
#pragma unroll
        for (int e2 = 0; e2 < 3; ++e2)
#pragma unroll
            for (int e3 = 0; e3 < 3; ++e3)
                if (!((unsigned)(o2 + e2 - p.pad) < (unsigned)p.D2 && (unsigned)(o3 + e3 - p.pad) < (unsigned)p.D3)) mk |= 1u << (3 * e2 + e3);
        tab[i] = 0x80000000u | mk | ((unsigned)((o2 - p.pad) * p.D3 + (o3 - p.pad) + p.D3 + 1) << 9);   // bit 31: "outside" for idle slots
    }
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int delta[4], sh[4];   // wave-uniform: byte offset of the tap inside x (less the table's bias), bit of its (d2, d3) in the mask
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tap = wave + 8 * t;   // tap 27..31: idle slot of waves 3..7
        delta[t] = ((((tap / 9) - p.pad) * p.D2 + (tap / 3) % 3) * p.D3 + tap % 3 - (p.D3 + 1)) * RC * 4;
        sh[t] = tap < 27 ? tap % 9 : 31;
    }
    const bool fourth = wave + 24 < 27;
    __syncthreads();
    // this lane's running output voxel: v0 + h, + 2 per MFMA k-step; idx = its in-plane position, row = byte offset of its x row
    int v = v0 + h;
    int idx = v % plane;
    int row = (v / plane) * p.D2 * p.D3 * RC * 4 + l32 * 4;
    const int row_step = p.D2 * p.D3 * RC * 4;
    for (int vb = v0; vb < v1; vb += 8) {
        float dyv[4], xv[4][4] = {};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ok = v < v1;
            const unsigned e = ok ? tab[idx] : 0x800001ffu;
            const int base = row + (int)((e >> 9) & 0x3fffffu) * (RC * 4);
            dyv[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdy, ok ? (v * RC + l32) * 4 : 0x7F000000, 0, 0));
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < 3 || fourth)      // (wave-uniform: waves 3..7 have no fourth tap -- no load, no MFMA for it)
                    xv[t][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             srdx, (base + delta[t]) | __builtin_amdgcn_sbfe((int)e, sh[t], 1), 0, 0));
            v += 2;
            idx += 2;
            while (idx >= plane) {
                idx -= plane;
                row += row_step;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[t][s], dyv[s], acc[t], 0, 0, 0);
        if (fourth) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[3][s], dyv[s], acc[3], 0, 0, 0);
        }
    }
    // C/D map: col = lane & 31 (co), row = (r & 3) + 8 (r >> 2) + 4 h (ci)
    const long long blk = (long long)b * gridDim.x + blockIdx.x;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tap = wave + 8 * t;
        if (tap < 27) {
            float* dst = p.slab + (blk * 27 + tap) * RC * RC + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * h) * RC] = acc[t][r];
        }
    }
}

constexpr int WGRAD_BLOCKS = 512;        // target number of blocks (two per CU)
constexpr int WGRAD_BLOCKS_MAX = 1024;   // slabs the workspace is sized for (blocks_per_b * B never exceeds max(512, B))
int64_t reduce_tmp_floats(int64_t nslabs, int64_t len);
int launch_reduce_slabs(float* out, const float* slab, int nslabs, int64_t len, float* tmp, hipStream_t st);
int launch_reduce_slabs_pitched(float* out, const float* slab, int nslabs, int64_t len, int64_t pitch, float* tmp, hipStream_t st);
int64_t colsum_ws_floats(int64_t n, int C, int G);
int launch_colsum(float* out, const float* X, const float* g, int64_t n, int C, int G, float* slab, hipStream_t st);

size_t rams_conv3d_wgrad_ws_floats(long long nvox) {
    const size_t a = (size_t)WGRAD_BLOCKS_MAX * (CONV_W_FLOATS + RC) + (size_t)reduce_tmp_floats(WGRAD_BLOCKS_MAX, CONV_W_FLOATS + RC);
    const size_t b = (size_t)colsum_ws_floats(nvox, RC, 1);
    return (a > b ? a : b) + 64 + 2 * R3_SLOT + 64;      // (+ two max|.| slots for the stand-alone split-fp16 call)
}

int rams_conv3d_forward(float* y, const float* x, const float* w, const float* bias, int B, int D1, int D2, int D3, int pad,
                        int relu, hipStream_t st) {
    const int ovox = (D1 + 2 * pad - 2) * (D2 + 2 * pad - 2) * (D3 + 2 * pad - 2);
    return conv3d_mfma(x, y, w, bias, nullptr, B, D1, D2, D3, pad, RC, RC, relu, rams_waves_per_b(B, ovox), st);
}

// dx = conv_same(dy, flip-transpose(w)); ws: 27*32*32 + 32 floats
int rams_conv3d_dgrad_same(float* dx, const float* dy, const float* w, int B, int D1, int D2, int D3, float* ws,
                           hipStream_t st) {
    float* wd = ws;
    float* zero_bias = ws + CONV_W_FLOATS;
    INR_HIP(hipMemsetAsync(zero_bias, 0, RC * sizeof(float), st));
    {
        ProfScope ps(KC_OTHER, st);
        hipLaunchKernelGGL(conv3d_dgrad_weights_kernel, dim3((CONV_W_FLOATS + 255) / 256), dim3(256), 0, st, wd, w);
        INR_LAUNCH_CHECK();
    }
    return conv3d_mfma(dy, dx, wd, zero_bias, nullptr, B, D1, D2, D3, 1, RC, RC, 0, rams_waves_per_b(B, D1 * D2 * D3), st);
}

int rams_conv3d_wgrad(float* gw, float* gb, const float* x, const float* dy, int B, int D1, int D2, int D3, int pad, float* ws,
                      hipStream_t st) {
    Conv3dWgradParams p{};
    p.x = x; p.dy = dy; p.slab = ws;
    p.B = B; p.D1 = D1; p.D2 = D2; p.D3 = D3; p.pad = pad;
    p.O1 = D1 + 2 * pad - 2; p.O2 = D2 + 2 * pad - 2; p.O3 = D3 + 2 * pad - 2;
    const int ovox = p.O1 * p.O2 * p.O3;
    int blocks_per_b = WGRAD_BLOCKS / B;
    if (blocks_per_b < 1) blocks_per_b = 1;
    p.vox_per_block = ((ovox + blocks_per_b - 1) / blocks_per_b + 7) / 8 * 8;
    blocks_per_b = (ovox + p.vox_per_block - 1) / p.vox_per_block;
    const int nslabs = blocks_per_b * B;
    INR_REQUIRE((long long)D1 * D2 * D3 * RC * 4 < (1ll << 31) && nslabs <= WGRAD_BLOCKS_MAX, INR_E_INVALID,
                "conv3d wgrad: one batch element must stay below 2 GiB and B below %d", WGRAD_BLOCKS_MAX);
    INR_REQUIRE(p.O2 * p.O3 >= 1 && p.O2 * p.O3 <= WGRAD_TAB_MAX, INR_E_INVALID,
                "conv3d wgrad: %d x %d in-plane output positions exceed the kernel's table of %d", p.O2, p.O3, WGRAD_TAB_MAX);
    {
        ProfScope ps(KC_OTHER, st);
        hipLaunchKernelGGL(conv3d_c32_wgrad_kernel, dim3(blocks_per_b, B), dim3(512), (size_t)p.O2 * p.O3 * sizeof(unsigned), st, p);
        INR_LAUNCH_CHECK();
    }
    if (int rc = launch_reduce_slabs(gw, ws, nslabs, CONV_W_FLOATS, ws + (size_t)nslabs * CONV_W_FLOATS, st)) return rc;
    if (gb) return launch_colsum(gb, dy, nullptr, (long long)B * ovox, RC, 1, ws, st);
    return 0;
}

// number of waves per batch element for the conv kernel: fill the chip once, never more waves than tiles
// the same gradient on the split-fp16 kernel; xs / ds: slots holding max|x| and max|dy| (R3_SLOT words each)
int rams_conv3d_wgrad_h3(float* gw, float* gb, const float* x, const float* dy, const unsigned* xs, const unsigned* ds, int B, int D1,
                         int D2, int D3, int pad, float* ws, hipStream_t st) {
    Conv3dWgradH3Params p{};
    p.x = x; p.dy = dy; p.slab = ws; p.x_amax = xs; p.dy_amax = ds;
    p.B = B; p.D1 = D1; p.D2 = D2; p.D3 = D3; p.pad = pad;
    p.O1 = D1 + 2 * pad - 2; p.O2 = D2 + 2 * pad - 2; p.O3 = D3 + 2 * pad - 2;
    r3l_choose_patch(D3, p.O3, &p.PO1, &p.PO2);
    p.np1 = (p.O1 + p.PO1 - 1) / p.PO1;
    p.np2 = (p.O2 + p.PO2 - 1) / p.PO2;
    p.mD3 = r3_magic(D3); p.mHP2 = r3_magic(p.PO2 + 2); p.mO3 = r3_magic(p.O3); p.mPO2 = r3_magic(p.PO2); p.mNP2 = r3_magic(p.np2);
    const int ovox = p.O1 * p.O2 * p.O3;
    int blocks_per_b = 256 / B < 1 ? 1 : 256 / B;               // one block (107 KB of LDS) per CU
    if (blocks_per_b > p.np1 * p.np2) blocks_per_b = p.np1 * p.np2;
    const int nslabs = blocks_per_b * B;
    INR_REQUIRE((long long)D1 * D2 * D3 * RC * 4 < (1ll << 30) && nslabs <= WGRAD_BLOCKS_MAX && D3 * 9 <= R3L_MAX_HVOX, INR_E_INVALID,
                "conv3d wgrad (split-fp16): one batch element must stay below 1 GiB, B below %d, depth below %d", WGRAD_BLOCKS_MAX,
                R3L_MAX_HVOX / 9 + 1);
    {
        ProfScope ps(KC_OTHER, st);
        hipLaunchKernelGGL(conv3d_c32_wgrad_h3_kernel, dim3(blocks_per_b, B), dim3(512), 0, st, p);
        INR_LAUNCH_CHECK();
    }
    // slab rows: [27 x 32 x 32 of gw][32 of gb]; one reduction when the caller keeps gb right behind gw (the training step's folded
    // gradient buffer does), two otherwise
    float* tmp = ws + (size_t)nslabs * W3_SLAB;
    (void)ovox;
    if (gb == gw + CONV_W_FLOATS) return launch_reduce_slabs_pitched(gw, ws, nslabs, W3_SLAB, W3_SLAB, tmp, st);
    if (int rc = launch_reduce_slabs_pitched(gw, ws, nslabs, CONV_W_FLOATS, W3_SLAB, tmp, st)) return rc;
    if (gb) return launch_reduce_slabs_pitched(gb, ws + CONV_W_FLOATS, nslabs, RC, W3_SLAB, tmp, st);
    return 0;
}

// stand-alone weight gradient (the C-ABI building block): the arithmetic debug key 14 selects, like the training step
int rams_conv3d_wgrad_auto(float* gw, float* gb, const float* x, const float* dy, int B, int D1, int D2, int D3, int pad, float* ws,
                           hipStream_t st) {
    if (g_rams_h3 != 2 || !r3l_fits(D1, D2, D3) || B > WGRAD_BLOCKS_MAX) return rams_conv3d_wgrad(gw, gb, x, dy, B, D1, D2, D3, pad, ws, st);
    const long long nvox = (long long)B * (D1 + 2 * pad - 2) * (D2 + 2 * pad - 2) * (D3 + 2 * pad - 2);
    float* tail = ws + (rams_conv3d_wgrad_ws_floats(nvox > 0 ? nvox : 1) - (2 * R3_SLOT + 64));
    unsigned* slots = reinterpret_cast<unsigned*>((reinterpret_cast<uintptr_t>(tail) + 63) & ~(uintptr_t)63);
    INR_HIP(hipMemsetAsync(slots, 0, 2 * R3_SLOT * sizeof(unsigned), st));
    auto amax = [&](unsigned* slot, const float* t, long long n) {
        long long g = (n / 4 + 255) / 256;
        if (g > 2048) g = 2048;
        if (g < 1) g = 1;
        hipLaunchKernelGGL(r3_tensor_amax_kernel, dim3((unsigned)g), dim3(256), 0, st, slot, t, n / 4);
    };
    amax(slots, x, (long long)B * D1 * D2 * D3 * RC);
    amax(slots + R3_SLOT, dy, nvox * RC);
    INR_LAUNCH_CHECK();
    return rams_conv3d_wgrad_h3(gw, gb, x, dy, slots, slots + R3_SLOT, B, D1, D2, D3, pad, ws, st);
}

int rams_waves_per_b(int B, int ovox) {
    const int tiles = (ovox + 31) / 32;
    int blocks = 256 / B;                           // one 8-wave block per CU (108 KB of LDS each): never more than 256
                                                    // blocks in total, or a second, nearly empty round doubles the time
    if (blocks < 1) blocks = 1;
    const int max_blocks = (tiles + 7) / 8;
    if (blocks > max_blocks) blocks = max_blocks;
    return blocks * 8;
}

// Channel-sum slab region of a forward / training pass: [B][rows][32] floats, where `rows` is the most any producer writes per
// batch element -- 8 per block of the f32-input / global-operand kernels (rams_waves_per_b), 4 or 8 per block of the LDS-staged
// kernels, whose default (key 15 = 42) launches up to 512 / B blocks, and the 64 phases of chan_partial_kernel / the gate
// backward.  (Round 3 sized the inference region by the first term only: at B = 20, 22, 24, 26, 29, 30 the 42-kernel wrote into
// the gate rows behind it, from B = 33 on past the pad.)  + 4096: the [B][32] gates live in the last 2,048 floats.
long long rams_slab_floats(int B, int ovox_max) {
    const int per_b_lds = (512 / B < 1 ? 1 : 512 / B) * 8;
    int rows = rams_waves_per_b(B, ovox_max);
    if (rows < per_b_lds) rows = per_b_lds;
    if (rows < 64) rows = 64;
    long long gates = (long long)B * RC;
    return (long long)B * rows * RC + 2048 + (gates > 2048 ? gates : 2048);
}

size_t rams_workspace_floats(const inr_rams_desc_t* d, int B, int H, int W) {
    const long long T = d->channels;
    const long long big = (long long)B * (H + 4) * (W + 4) * T * RC;      // largest 5-D activation (padded reduction stage)
    const long long slab = rams_slab_floats(B, (H + 4) * (W + 4) * (int)T);
    const long long small = (long long)B * (H + 2) * (W + 2) * T * 4 + (long long)B * H * W * d->scale * d->scale * 2;
    // split-fp16 inference: hi/lo planes of every 32 -> 32 kernel + scale slots
    const long long h3 = (long long)rams_conv3d_count(d) * (R3_LAYER_HALVES / 2) + 1024 +
                         (3ll * rams_conv3d_count(d) + 8) * R3_SLOT;
    return (size_t)(5 * big + slab + small + 8192 + h3);
}

int rams_forward_impl(const inr_rams_desc_t* d, const float* params, const float* x, float* out, int B, int H, int W,
                      int clip_round, float* ws, hipStream_t st) {
    const int T = d->channels, Cr = d->filters / d->r, S2 = d->scale * d->scale;
    const int Tr = T / d->r > 0 ? T / d->r : 1;
    const long long big = (long long)B * (H + 4) * (W + 4) * T * RC;
    float* bufA = ws;               // current trunk activation
    float* bufB = bufA + big;       // conv1 output
    float* bufC = bufB + big;       // conv2 output (to be gated)
    float* bufR = bufC + big;       // trunk residual (stem output)
    float* bufP = bufR + big;       // padded copies
    float* slab = bufP + big;
    const long long slab_floats = rams_slab_floats(B, (H + 4) * (W + 4) * T);
    const long long gate_floats = (long long)B * RC > 2048 ? (long long)B * RC : 2048;
    const long long slab_cap = slab_floats - gate_floats - 2048;   // what the channel-sum producers may use
    float* gate = slab + slab_floats - gate_floats;  // [B][32]
    float* xn = slab + slab_floats;                  // normalised input [B][H][W][T]
    float* xpad = xn + (long long)B * H * W * T;     // reflect-padded [B][H+2][W+2][T]
    float* g1 = xpad + (long long)B * (H + 2) * (W + 2) * T;
    float* g2 = g1 + (long long)B * (H + 2) * (W + 2) * T;
    float* upo = g2 + (long long)B * (H + 2) * (W + 2) * T;   // [B][H][W][S2]
    float* glo = upo + (long long)B * H * W * S2;
    // split-fp16 state behind everything else: [slots: 1024 x u32][planes of conv 0][planes of conv 1]...
    const bool h3 = g_rams_h3 != 0;
    const int n_conv = rams_conv3d_count(d);
    INR_REQUIRE(n_conv <= R3_MAX_CONVS, INR_E_INVALID, "rams: too many 3-D convolutions (%d)", n_conv);
    unsigned* slots = reinterpret_cast<unsigned*>(ws + 5 * big + slab_floats + (long long)B * H * W * T +
                                                  3 * (long long)B * (H + 2) * (W + 2) * T + 2 * (long long)B * H * W * S2 + 64);
    slots = reinterpret_cast<unsigned*>((reinterpret_cast<uintptr_t>(slots) + 255) & ~(uintptr_t)255);
    // words [0, 1024): max|w| per convolution; then R3_MAX_SLOTS tensor slots of R3_SLOT words each; then the weight planes
    unsigned* tslots = slots + 1024;
    const int n_slots = 3 * n_conv + 8;       // (at most three new tensors per convolution + stem, skip, ...)
    _Float16* planes = reinterpret_cast<_Float16*>(tslots + (long long)n_slots * R3_SLOT);
    int conv_no = 0, next_slot = 0;
    auto new_slot = [&]() {     // (a graph with more tensors than slots would share the last one: a larger maximum, still a valid scale)
        const int k = next_slot < n_slots - 1 ? next_slot++ : n_slots - 1;
        return tslots + (long long)k * R3_SLOT;
    };
    if (h3) {
        INR_HIP(hipMemsetAsync(slots, 0, (1024 + (size_t)n_slots * R3_SLOT) * sizeof(unsigned), st));
        // the 3-D kernels in consumption order (same walk as below)
        R3SplitJobs jobs{};
        Cursor w{params};
        int k = 0;
        auto conv = [&]() { jobs.j[k] = {w.take(CONV_W_FLOATS), planes + (long long)k * R3_LAYER_HALVES, slots + k}; ++k; w.take(RC); };
        auto skip_gate = [&](int C, int Cr2) { w.take((long long)C * Cr2); w.take(Cr2); w.take((long long)Cr2 * C); w.take(C); };
        w.take(27 * RC); w.take(RC);
        for (int i = 0; i < d->n_rfab; ++i) { conv(); conv(); skip_gate(RC, Cr); }
        conv();
        for (int i = 0; i < T / 3; ++i) { conv(); conv(); skip_gate(RC, Cr); conv(); }
        conv();
        hipLaunchKernelGGL(rams_weight_split_all_kernel, dim3(n_conv), dim3(512), 0, st, jobs);
        INR_LAUNCH_CHECK();
    }
    // one 3-D convolution (either arithmetic); xs = slot of max|x|, ys = slot that receives max|y| (nullable)
    int last_nslab = 0;      // channel-sum slabs the last convolution wrote per batch element
    // add (nullable): the staged kernel adds this tensor to its output in the epilogue (rams_h3.inc, AUX 1); *added says whether it did
    auto conv3d = [&](const float* xin, float* yout, const float* w, const float* bias, float* chan, const unsigned* xs,
                      unsigned* ys, int D1, int D2, int D3, int pad, int cout, int cstride, int relu, int wpb,
                      const float* add = nullptr, bool* added = nullptr, bool* padded = nullptr, const float* gate_of_add = nullptr) -> int {
        // gate_of_add (with add): y = conv * gate + add instead of conv + add (AUX 4)
        // padded (nullable): the caller would like the output written as the interior of a reflect-padded image (y_pad); *padded: done
        const int k = conv_no++;
        last_nslab = wpb;
        if (added) *added = false;
        if (padded) *padded = false;
        if (h3 && g_rams_h3 == 2 && r3l_fits(D1, D2, D3)) {
            const bool with_add = add && added && rams_lds_aux_ok();
            const bool with_pad = padded && !with_add && rams_lds_aux_ok();
            if (added) *added = with_add;
            if (padded) *padded = with_pad;
            return conv3d_h3_lds(xin, yout, planes + (long long)k * R3_LAYER_HALVES, bias, chan, xs, slots + k, ys, B, D1, D2, D3,
                                 pad, cout, cstride, relu, &last_nslab, st, slab_cap, with_add ? add : nullptr,
                                 with_add ? (gate_of_add ? 4 : 1) : 0, with_pad ? 1 : 0, with_add ? gate_of_add : nullptr);
        }
        if (h3)
            return conv3d_h3(xin, yout, planes + (long long)k * R3_LAYER_HALVES, bias, chan, xs, slots + k, ys, B, D1, D2, D3, pad,
                             cout, cstride, relu, wpb, st);
        return conv3d_mfma(xin, yout, w, bias, chan, B, D1, D2, D3, pad, cout, cstride, relu, wpb, st);
    };
    Cursor c{params};

    unsigned* io_slot = nullptr;   // slot of max|.| of the tensor the next convolution reads
    auto rfab = [&](float* io, int D1, int D2, int D3) -> int {   // io updated in place: io = gate*conv2(relu(conv1(io))) + io
        const float* w1 = c.take(CONV_W_FLOATS); const float* b1 = c.take(RC);
        const float* w2 = c.take(CONV_W_FLOATS); const float* b2 = c.take(RC);
        const float* wsq = c.take((long long)RC * Cr); const float* bsq = c.take(Cr);
        const float* wex = c.take((long long)Cr * RC); const float* bex = c.take(RC);
        const int ovox = D1 * D2 * D3;
        const int wpb = rams_waves_per_b(B, ovox);
        unsigned* mid = h3 ? new_slot() : nullptr;
        // The gate from the FIRST convolution's output (gate_pre_kernel above), scale + residual in the second one's epilogue: needs
        // the staged kernel with its epilogue operand for both convolutions and three indices per axis (debug key 24 = 2, default)
        // -- and enough voxels: the two small kernels in front of the second convolution cost ~33 us + 1.1 us per stack against
        // 9 us + 10.6 us per stack of the gate and scale passes they replace (128 x 128 x 9 stacks; batch 1: 1.70 against 1.46 ms)
        if (h3 && g_rams_h3 == 2 && r3l_fits(D1, D2, D3) && rams_lds_aux_ok() && g_rams_epi_fuse >= 2 && D1 >= 3 && D2 >= 3 && D3 >= 3 &&
            (long long)B * ovox >= g_rams_pregate_min_vox && ovox >= RCLS_NB * 27) {      // (bufC holds the class partials: 64 x 27 x 32 floats per b)
            if (int rc = conv3d(io, bufB, w1, b1, slab, io_slot, mid, D1, D2, D3, 1, RC, RC, 1, wpb)) return rc;   // slab: sums of r1
            const int nslab_r1 = last_nslab;
            hipLaunchKernelGGL(rams_class_sums_kernel, dim3(RCLS_NB, B), dim3(256), 0, st, bufC, bufB, D1, D2, D3);
            INR_LAUNCH_CHECK();
            hipLaunchKernelGGL(gate_pre_kernel, dim3(B), dim3(1024), 0, st, gate, slab, nslab_r1, bufC, w2, b2, 1.0f / (float)ovox, wsq,
                               bsq, wex, bex, Cr);
            INR_LAUNCH_CHECK();
            unsigned* out_slot = new_slot();
            bool added = false;
            if (int rc = conv3d(bufB, io, w2, b2, nullptr, mid, out_slot, D1, D2, D3, 1, RC, RC, 0, wpb, io, &added, nullptr, gate)) return rc;
            INR_REQUIRE(added, INR_E_INVALID, "rams: the gated convolution epilogue was refused");
            io_slot = out_slot;
            return 0;
        }
        if (int rc = conv3d(io, bufB, w1, b1, nullptr, io_slot, mid, D1, D2, D3, 1, RC, RC, 1, wpb)) return rc;
        if (int rc = conv3d(bufB, bufC, w2, b2, slab, mid, nullptr, D1, D2, D3, 1, RC, RC, 0, wpb)) return rc;
        hipLaunchKernelGGL(gate_kernel, dim3(B), dim3(1024), 0, st, gate, slab, last_nslab, 1.0f / (float)ovox, wsq, bsq, wex,
                           bex, RC, Cr);
        INR_LAUNCH_CHECK();
        const long long total = (long long)B * ovox * RC;
        io_slot = h3 ? new_slot() : nullptr;
        {
            const long long per_b4 = (long long)ovox * RC / 4;
            long long gx = (per_b4 + 255) / 256;
            const long long cap = (8 * 256 + B - 1) / B;          // ~8 blocks per CU over the whole batch
            if (gx > cap) gx = cap;
            hipLaunchKernelGGL(scale_residual_c32_kernel, dim3((unsigned)gx, B), dim3(256), 0, st, io, bufC, gate, io, per_b4,
                               io_slot);
        }
        INR_LAUNCH_CHECK();
        return 0;
    };

    ProfScope ps(KC_OTHER, st);
    // normalise, lift, reflect-pad (network.py:113-116)
    const long long n_in = (long long)B * H * W * T;
    hipLaunchKernelGGL(normalize_kernel, dim3(nblk(n_in)), dim3(256), 0, st, xn, x, n_in, d->mean, d->std);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(nblk((long long)B * (H + 2) * (W + 2) * T)), dim3(256), 0, st, xpad, xn, B, H,
                       W, T);
    INR_LAUNCH_CHECK();
    int D1 = H + 2, D2 = W + 2, D3 = T;
    {   // stem (network.py:119)
        const float* w = c.take(27 * RC); const float* b = c.take(RC);
        io_slot = h3 ? new_slot() : nullptr;
        launch_stem(bufA, xpad, w, b, B, D1, D2, D3, io_slot, bufR, st);
        INR_LAUNCH_CHECK();     // (bufR: the stem output kept for the long skip)
    }
    for (int i = 0; i < d->n_rfab; ++i)
        if (int rc = rfab(bufA, D1, D2, D3)) return rc;
    {   // trunk close + long skip (network.py:127-129)
        const float* w = c.take(CONV_W_FLOATS); const float* b = c.take(RC);
        const int wpb = rams_waves_per_b(B, D1 * D2 * D3);
        // the long skip rides in the convolution's epilogue (conv + stem output, the operands and order of add_v4_kernel: same bits;
        // 0.28 ms per 25 stacks); the sum then lives in bufB, so the two buffers change names
        unsigned* sum_slot = h3 ? new_slot() : nullptr;
        bool added = false;
        if (int rc = conv3d(bufA, bufB, w, b, nullptr, io_slot, sum_slot, D1, D2, D3, 1, RC, RC, 0, wpb, bufR, &added)) return rc;
        const long long total = (long long)B * D1 * D2 * D3 * RC;
        if (added) {
            float* t = bufA;
            bufA = bufB;
            bufB = t;
        } else {
            if (sum_slot) INR_HIP(hipMemsetAsync(sum_slot, 0, R3_SLOT * sizeof(unsigned), st));   // (it holds max|conv|, not max|sum|)
            hipLaunchKernelGGL(add_v4_kernel, dim3(2048), dim3(256), 0, st, bufA, bufB, bufR, total / 4, sum_slot);
            INR_LAUNCH_CHECK();
        }
        io_slot = sum_slot;
    }
    bool have_padded = false;           // bufP already holds the next stage's padded input (written there by the previous stage)
    for (int i = 0; i < T / 3; ++i) {   // temporal reduction (network.py:132-136)
        if (!have_padded) {
            hipLaunchKernelGGL(reflect_pad_v4_kernel, dim3((D1 + 2) * (D2 + 2), B), dim3(256), 0, st, bufP, bufA, D1, D2, D3 * RC / 4);
            INR_LAUNCH_CHECK();
        }
        if (int rc = rfab(bufP, D1 + 2, D2 + 2, D3)) return rc;      // (reflect padding copies values: max|.| carries over)
        const float* w = c.take(CONV_W_FLOATS); const float* b = c.take(RC);
        const int wpb = rams_waves_per_b(B, D1 * D2 * (D3 - 2));
        unsigned* out_slot = h3 ? new_slot() : nullptr;
        // every stage but the last feeds another padded block: its convolution writes the interior of that padded image itself and
        // a kernel over the frame completes it (the values reflect_pad_v4_kernel would have copied: same bits; 0.2 ms per 25 stacks each)
        bool padded = false;
        if (int rc = conv3d(bufP, bufA, w, b, nullptr, io_slot, out_slot, D1 + 2, D2 + 2, D3, 0, RC, RC, 1, wpb, nullptr, nullptr,
                            i + 1 < T / 3 ? &padded : nullptr))
            return rc;
        io_slot = out_slot;
        D3 -= 2;
        have_padded = padded;
        if (padded) {
            hipLaunchKernelGGL(reflect_border_v4_kernel, dim3(2 * (D2 + 2) + 2 * D1, B), dim3(256), 0, st, bufA, D1, D2, D3 * RC / 4);
            INR_LAUNCH_CHECK();
            float* t = bufA;
            bufA = bufP;
            bufP = t;
        }
    }
    {   // up-scaling head: Conv3D 32 -> scale^2, valid; keep T index 0 (network.py:139-140)
        const float* w = c.take(CONV_W_FLOATS); const float* b = c.take(RC);
        INR_REQUIRE(D3 == 3, INR_E_INVALID, "rams: temporal depth before the head must be 3 (got %d)", D3);
        const int wpb = rams_waves_per_b(B, (D1 - 2) * (D2 - 2));
        if (int rc = conv3d(bufA, upo, w, b, nullptr, io_slot, nullptr, D1, D2, D3, 0, S2, S2, 0, wpb)) return rc;
    }
    {   // global residual path: RTAB on the padded normalised input + valid conv (network.py:145-148)
        const float* w1 = c.take(9LL * T * T); const float* b1 = c.take(T);
        const float* w2 = c.take(9LL * T * T); const float* b2 = c.take(T);
        const float* wsq = c.take((long long)T * Tr); const float* bsq = c.take(Tr);
        const float* wex = c.take((long long)Tr * T); const float* bex = c.take(T);
        const float* wg = c.take(9LL * T * S2); const float* bg = c.take(S2);
        const int P1 = H + 2, P2 = W + 2;
        const long long tot = (long long)B * P1 * P2 * T;
        launch_conv2d(g1, xpad, w1, b1, B, P1, P2, T, T, 1, 1, st);
        INR_LAUNCH_CHECK();
        launch_conv2d(g2, g1, w2, b2, B, P1, P2, T, T, 1, 0, st);
        INR_LAUNCH_CHECK();
        const int nb2 = 64;
        hipLaunchKernelGGL(chan_partial_kernel, dim3(nb2, B), dim3(256), 0, st, slab, g2, (long long)P1 * P2, T, nb2);
        INR_LAUNCH_CHECK();
        hipLaunchKernelGGL(gate_kernel, dim3(B), dim3(1024), 0, st, gate, slab, nb2, 1.0f / (float)(P1 * P2), wsq, bsq, wex, bex,
                           T, Tr);
        INR_LAUNCH_CHECK();
        hipLaunchKernelGGL(scale_residual_kernel, dim3(nblk(tot)), dim3(256), 0, st, g1, g2, gate, xpad, (long long)P1 * P2, T,
                           tot);
        INR_LAUNCH_CHECK();
        launch_conv2d(glo, g1, wg, bg, B, P1, P2, T, S2, 0, 0, st);
        INR_LAUNCH_CHECK();
    }
    const long long n_out = (long long)B * H * d->scale * W * d->scale;
    hipLaunchKernelGGL(shuffle_sum_kernel, dim3(nblk(n_out)), dim3(256), 0, st, out, upo, glo, B, H, W, d->scale, d->mean,
                       d->std, clip_round);
    INR_LAUNCH_CHECK();
    return 0;
}

#include "rams_train.inc"

}  // namespace inr
