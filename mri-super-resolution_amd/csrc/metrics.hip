// End-of-fit metrics on the device: PSNR, skimage-compatible SSIM, ADC maps.
// The reference computes them on the host after copying the reconstruction back
// (superresDWI.py:179-186 skimage structural_similarity; SRDWI.py:118-130 per-pixel np.polyfit in a Python
// double loop).  Here the volume stays in HBM; everything accumulates in fp64 with fixed-order reductions.
#include "common.h"

namespace inr {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum_f64(double v, double* red /*[4]*/) {
    v = wave_sum_f64(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

// partial[b][blk] = sum over a slice of image b of (x - y)^2
__global__ void __launch_bounds__(256) sqdiff_kernel(double* __restrict__ partial, const float* __restrict__ x,
                                                     const float* __restrict__ y, int64_t per_image, int blocks_per_image) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * per_image;
    const float* yb = y + (int64_t)b * per_image;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_image; i += (int64_t)blocks_per_image * 256) {
        const double d = (double)xb[i] - (double)yb[i];
        acc += d * d;
    }
    const double s = block_sum_f64(acc, red);
    if (threadIdx.x == 0) partial[(int64_t)b * blocks_per_image + blockIdx.x] = s;
}

// out[b] = 10*log10(range^2 / (sum_k partial[b][k] / per_image))
__global__ void psnr_finish_kernel(double* __restrict__ out, const double* __restrict__ partial, int blocks_per_image,
                                   int64_t per_image, double data_range, int nimg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nimg) return;
    double s = 0.0;
    for (int k = 0; k < blocks_per_image; ++k) s += partial[(int64_t)b * blocks_per_image + k];
    const double mse = s / (double)per_image;
    out[b] = 10.0 * log10(data_range * data_range / mse);
}

// skimage 0.20 structural_similarity(x, y, data_range=R) for 2-D float images with default arguments:
// uniform win x win window (win = 7), sample covariance (N/(N-1)), K1 = 0.01, K2 = 0.03, mean of S over the image
// cropped by (win-1)/2 on every side.  Optional masking as at superresDWI.py:183-186: both images are multiplied by
// (x > mask_thr) when use_mask != 0.  One thread per cropped pixel; grid = (blocks, images).
__global__ void __launch_bounds__(256) ssim_kernel(double* __restrict__ partial, const float* __restrict__ x,
                                                   const float* __restrict__ y, int H, int W, int win, double c1,
                                                   double c2, int use_mask, float mask_thr, int blocks_per_image) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * H * W;
    const float* yb = y + (int64_t)b * H * W;
    const int pad = (win - 1) / 2;
    const int h = H - 2 * pad, w = W - 2 * pad;
    const int64_t count = (int64_t)h * w;
    const double np_ = (double)win * win, cov_norm = np_ / (np_ - 1.0);
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)blocks_per_image * 256) {
        const int r = (int)(i / w), c = (int)(i - (int64_t)r * w);
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dr = 0; dr < win; ++dr) {
            const float* xr = xb + (int64_t)(r + dr) * W + c;
            const float* yr = yb + (int64_t)(r + dr) * W + c;
            for (int dc = 0; dc < win; ++dc) {
                double xv = xr[dc], yv = yr[dc];
                if (use_mask) {
                    const double m = xr[dc] > mask_thr ? 1.0 : 0.0;
                    xv *= m;
                    yv *= m;
                }
                sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
            }
        }
        const double ux = sx / np_, uy = sy / np_;
        const double vx = cov_norm * (sxx / np_ - ux * ux), vy = cov_norm * (syy / np_ - uy * uy);
        const double vxy = cov_norm * (sxy / np_ - ux * uy);
        acc += ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
    }
    const double s = block_sum_f64(acc, red);
    if (threadIdx.x == 0) partial[(int64_t)b * blocks_per_image + blockIdx.x] = s;
}

__global__ void mean_finish_kernel(double* __restrict__ out, const double* __restrict__ partial, int blocks_per_image,
                                   double count, int nimg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nimg) return;
    double s = 0.0;
    for (int k = 0; k < blocks_per_image; ++k) s += partial[(int64_t)b * blocks_per_image + k];
    out[b] = s / count;
}

// calculate_ADC (SRDWI.py:118-130): per pixel, -slope of the degree-1 least-squares fit of log(S + 1e-7) against
// b/1000, clipped to [-10, 3]; closed form of np.polyfit.  data[npix][nb], bvals[nb] (nb <= 32).
__global__ void __launch_bounds__(256) adc_kernel(float* __restrict__ out, const float* __restrict__ data,
                                                  const float* __restrict__ bvals, int64_t npix, int nb) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    double bm = 0.0;
    for (int k = 0; k < nb; ++k) bm += (double)bvals[k] / 1000.0;
    bm /= nb;
    double ym = 0.0, yv[32];
    for (int k = 0; k < nb; ++k) {
        yv[k] = log((double)data[i * nb + k] + 1e-7);
        ym += yv[k];
    }
    ym /= nb;
    double num = 0.0, den = 0.0;
    for (int k = 0; k < nb; ++k) {
        const double db = (double)bvals[k] / 1000.0 - bm;
        num += db * (yv[k] - ym);
        den += db * db;
    }
    double adc = -(num / den);
    adc = adc > 3.0 ? 3.0 : (adc < -10.0 ? -10.0 : adc);
    out[i] = (float)adc;
}

// ---- RAMS shift-tolerant losses (multi-image-super-resolution/utils/loss.py:26-75 l1_loss, :77-127 psnr) -----------
// For each of the (2*border+1)^2 label shifts: brightness bias b = mean_masked(label - pred) over the cropped window,
// then the masked mean of |label - (pred + b)| (mode 0) or of its square (mode 1).  One block per (shift, image),
// fp64 accumulation, two passes over the (size - 2*border)^2 window.  y_true/y_pred/mask: [B][size][size] fp32.
__global__ void __launch_bounds__(256) shift_loss_kernel(double* __restrict__ per_shift, const float* __restrict__ y_true,
                                                         const float* __restrict__ y_pred, const float* __restrict__ mask,
                                                         int size, int border, int mode) {
    __shared__ double red[4];
    const int nshift = 2 * border + 1;
    const int si = blockIdx.x / nshift, sj = blockIdx.x % nshift, b = blockIdx.y;
    const int c = size - 2 * border;
    const float* yt = y_true + (long long)b * size * size;
    const float* yp = y_pred + (long long)b * size * size;
    const float* mk = mask + (long long)b * size * size;
    double sm = 0.0, sd = 0.0;
    for (int i = threadIdx.x; i < c * c; i += 256) {
        const int r = i / c, q = i - r * c;
        const double m = mk[(long long)(si + r) * size + sj + q];
        sm += m;
        sd += m * ((double)yt[(long long)(si + r) * size + sj + q] - (double)yp[(long long)(border + r) * size + border + q]);
    }
    sm = block_sum_f64(sm, red);
    sd = block_sum_f64(sd, red);
    const double bias = sd / sm;
    double acc = 0.0;
    for (int i = threadIdx.x; i < c * c; i += 256) {
        const int r = i / c, q = i - r * c;
        const double m = mk[(long long)(si + r) * size + sj + q];
        // labels*m - (pred*m + b)*m  (loss.py:43-60)
        const double d = m * (double)yt[(long long)(si + r) * size + sj + q] -
                         m * (m * (double)yp[(long long)(border + r) * size + border + q] + bias);
        acc += mode == 0 ? fabs(d) : d * d;
    }
    acc = block_sum_f64(acc, red);
    if (threadIdx.x == 0) per_shift[(long long)b * nshift * nshift + blockIdx.x] = acc / sm;
}

// out[b] = min over shifts of cL1 (mode 0) / max over shifts of 10*log10(65535^2 / cMSE) (mode 1)
__global__ void shift_loss_finish_kernel(double* __restrict__ out, int* __restrict__ arg, const double* __restrict__ per_shift,
                                         int nshift2, int mode, int nimg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nimg) return;
    double best = 0.0;
    int at = 0;
    for (int k = 0; k < nshift2; ++k) {   // first best shift wins ties (tf.reduce_min / reduce_max pass the gradient to ... see below)
        const double v = per_shift[(long long)b * nshift2 + k];
        const double val = mode == 0 ? v : 10.0 * log10(65535.0 * 65535.0 / v);
        if (k == 0 || (mode == 0 ? val < best : val > best)) {
            best = val;
            at = k;
        }
    }
    out[b] = best;
    if (arg) arg[b] = at;
}

// Gradient of the cL1 loss of utils/loss.py:26-75 with respect to the prediction, through the best shift (i*, j*) = arg:
// with d_p = m_p lab_p - m_p (m_p pred_p + b), b = (1/T) sum_p m_p (lab_p - pred_p), T = sum_p m_p, s_p = sign(d_p):
//   dL/dpred_q = (1/T) [ -s_q m_q^2 + (m_q / T) sum_p s_p m_p ]      inside the cropped window, 0 on the border.
// (Exact ties between shifts split the gradient in TensorFlow; here the first best shift takes it all -- ties have
// measure zero on real data.)  One block per image, three passes over the window, fp64 accumulation.
__global__ void __launch_bounds__(256) shift_loss_grad_kernel(float* __restrict__ grad, const float* __restrict__ y_true,
                                                              const float* __restrict__ y_pred, const float* __restrict__ mask,
                                                              const int* __restrict__ arg, int size, int border,
                                                              const float* __restrict__ upstream) {
    __shared__ double red[4];
    const int b = blockIdx.x, nshift = 2 * border + 1, c = size - 2 * border;
    const int si = arg[b] / nshift, sj = arg[b] % nshift;
    const float* yt = y_true + (long long)b * size * size;
    const float* yp = y_pred + (long long)b * size * size;
    const float* mk = mask + (long long)b * size * size;
    float* gr = grad + (long long)b * size * size;
    for (int i = threadIdx.x; i < size * size; i += 256) gr[i] = 0.f;
    double sm = 0.0, sd = 0.0;
    for (int i = threadIdx.x; i < c * c; i += 256) {
        const int r = i / c, q = i - r * c;
        const double m = mk[(long long)(si + r) * size + sj + q];
        sm += m;
        sd += m * ((double)yt[(long long)(si + r) * size + sj + q] - (double)yp[(long long)(border + r) * size + border + q]);
    }
    sm = block_sum_f64(sm, red);
    sd = block_sum_f64(sd, red);
    const double bias = sd / sm;
    double ssm = 0.0;
    for (int i = threadIdx.x; i < c * c; i += 256) {
        const int r = i / c, q = i - r * c;
        const double m = mk[(long long)(si + r) * size + sj + q];
        const double d = m * (double)yt[(long long)(si + r) * size + sj + q] -
                         m * (m * (double)yp[(long long)(border + r) * size + border + q] + bias);
        ssm += (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0)) * m;
    }
    ssm = block_sum_f64(ssm, red);
    const double up = upstream ? (double)upstream[b] : 1.0;
    __syncthreads();   // the zero fill above is complete
    for (int i = threadIdx.x; i < c * c; i += 256) {
        const int r = i / c, q = i - r * c;
        const double m = mk[(long long)(si + r) * size + sj + q];
        const double d = m * (double)yt[(long long)(si + r) * size + sj + q] -
                         m * (m * (double)yp[(long long)(border + r) * size + border + q] + bias);
        const double s = d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0);
        gr[(long long)(border + r) * size + border + q] = (float)(up * (-s * m * m + m / sm * ssm) / sm);
    }
}

constexpr int METRIC_BLOCKS = 64;

int metric_workspace_doubles(int nimg) { return nimg * METRIC_BLOCKS; }

int launch_psnr(double* out, const float* x, const float* y, int nimg, int64_t per_image, double data_range,
                double* ws, hipStream_t st) {
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(sqdiff_kernel, dim3(METRIC_BLOCKS, nimg), dim3(256), 0, st, ws, x, y, per_image, METRIC_BLOCKS);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(psnr_finish_kernel, dim3((nimg + 63) / 64), dim3(64), 0, st, out, ws, METRIC_BLOCKS, per_image,
                       data_range, nimg);
    INR_LAUNCH_CHECK();
    return 0;
}

int launch_ssim(double* out, const float* x, const float* y, int nimg, int H, int W, int win, double data_range,
                int use_mask, float mask_thr, double* ws, hipStream_t st) {
    const int pad = (win - 1) / 2;
    const double c1 = (0.01 * data_range) * (0.01 * data_range), c2 = (0.03 * data_range) * (0.03 * data_range);
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(ssim_kernel, dim3(METRIC_BLOCKS, nimg), dim3(256), 0, st, ws, x, y, H, W, win, c1, c2, use_mask,
                       mask_thr, METRIC_BLOCKS);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_finish_kernel, dim3((nimg + 63) / 64), dim3(64), 0, st, out, ws, METRIC_BLOCKS,
                       (double)(H - 2 * pad) * (double)(W - 2 * pad), nimg);
    INR_LAUNCH_CHECK();
    return 0;
}

int launch_shift_loss(double* out, const float* y_true, const float* y_pred, const float* mask, int nimg, int size,
                      int border, int mode, double* ws, hipStream_t st) {
    const int ns = 2 * border + 1;
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(shift_loss_kernel, dim3(ns * ns, nimg), dim3(256), 0, st, ws, y_true, y_pred, mask, size, border, mode);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(shift_loss_finish_kernel, dim3((nimg + 63) / 64), dim3(64), 0, st, out, (int*)nullptr, ws, ns * ns, mode,
                       nimg);
    INR_LAUNCH_CHECK();
    return 0;
}

// loss (cL1, per image) and its gradient with respect to y_pred; ws: nimg * (2*border+1)^2 doubles + nimg ints
int launch_shift_loss_grad(double* loss, float* grad, const float* y_true, const float* y_pred, const float* mask,
                           const float* upstream, int nimg, int size, int border, double* ws, hipStream_t st) {
    const int ns = 2 * border + 1;
    int* arg = reinterpret_cast<int*>(ws + (size_t)nimg * ns * ns);
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(shift_loss_kernel, dim3(ns * ns, nimg), dim3(256), 0, st, ws, y_true, y_pred, mask, size, border, 0);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(shift_loss_finish_kernel, dim3((nimg + 63) / 64), dim3(64), 0, st, loss, arg, ws, ns * ns, 0, nimg);
    INR_LAUNCH_CHECK();
    hipLaunchKernelGGL(shift_loss_grad_kernel, dim3(nimg), dim3(256), 0, st, grad, y_true, y_pred, mask, arg, size, border,
                       upstream);
    INR_LAUNCH_CHECK();
    return 0;
}

// ---- spline baseline: skimage.transform.rescale(img, s, anti_aliasing=True) for s >= 1 (superresDWI.py:172-191) ---------
// skimage 0.20 resize -> scipy.ndimage.zoom(img, out/in, order=1, mode='mirror', grid_mode=True) (the anti-aliasing sigma
// max(0, (1/s - 1)/2) is 0 when up-scaling): out[o] = (1 - f) in[m(i)] + f in[m(i + 1)], x = (o + 0.5) in/out - 0.5,
// i = floor(x), f = x - i, m = reflection about the edge samples (period 2n - 2).  Coordinates and weights in double.
__device__ __forceinline__ int mirror_index(int i, int n) {
    if (n <= 1) return 0;
    const int period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

__global__ void __launch_bounds__(256) rescale_linear_kernel(float* __restrict__ out, const float* __restrict__ in, int nimg, int H,
                                                             int W, int OH, int OW) {
    const long long total = (long long)nimg * OH * OW;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ox = (int)(idx % OW);
        const int oy = (int)((idx / OW) % OH);
        const long long b = idx / ((long long)OW * OH);
        const double y = (oy + 0.5) * ((double)H / OH) - 0.5, x = (ox + 0.5) * ((double)W / OW) - 0.5;
        const double fy0 = floor(y), fx0 = floor(x);
        const double wy = y - fy0, wx = x - fx0;
        const int y0 = mirror_index((int)fy0, H), y1 = mirror_index((int)fy0 + 1, H);
        const int x0 = mirror_index((int)fx0, W), x1 = mirror_index((int)fx0 + 1, W);
        const float* img = in + b * H * W;
        // separable, rows first (axis 0), as scipy applies the 1-D splines
        const double c0 = (1.0 - wy) * img[(long long)y0 * W + x0] + wy * img[(long long)y1 * W + x0];
        const double c1 = (1.0 - wy) * img[(long long)y0 * W + x1] + wy * img[(long long)y1 * W + x1];
        out[idx] = (float)((1.0 - wx) * c0 + wx * c1);
    }
}

int launch_rescale_linear(float* out, const float* in, int nimg, int H, int W, int OH, int OW, hipStream_t st) {
    const long long total = (long long)nimg * OH * OW;
    if (total == 0) return 0;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(rescale_linear_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, in, nimg, H, W, OH, OW);
    INR_LAUNCH_CHECK();
    return 0;
}

int launch_adc(float* out, const float* data, const float* bvals, int64_t npix, int nb, hipStream_t st) {
    if (npix == 0) return 0;
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(adc_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, out, data, bvals, npix, nb);
    INR_LAUNCH_CHECK();
    return 0;
}

// ---- AutoERD acceptance weights (master.py:77-93) --------------------------------------------------------------------------------
// Per pixel: the n acquisition values (a 1-D sample) are split in two by complete-linkage agglomeration exactly as
// sklearn.cluster.AgglomerativeClustering(n_clusters=2, linkage='complete') does it -- scipy's nearest-neighbour-chain walk
// (scipy/cluster/_hierarchy.pyx: nn_chain, ties resolved by scan order and by preferring the previous chain element), a STABLE
// sort of the merges by distance, the tree cut at its last merge -- then one of the reference's two rejection rules.  Intensities
// are often integer-valued, so equal distances are the rule, not the exception: every tie falls as it does in scipy
// (oracle/erd_oracle.py restates the same steps in NumPy and is pinned against sklearn itself: tests/golden/erd.npz).
// One thread per pixel, float64 throughout (scipy converts to double); n <= ERD_MAXN values live in private arrays.  The
// reference runs this as a Python double loop over the ROI with one sklearn fit per pixel (3,600 fits per case).
constexpr int ERD_MAXN = 32;

// numpy's pairwise summation as np.mean runs it on a short contiguous float64 array (n < 8: plain loop; else eight partial sums)
__device__ double erd_numpy_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

__global__ void __launch_bounds__(64) auto_erd_kernel(float* __restrict__ accept, const double* __restrict__ values,
                                                      const float* __restrict__ erd_map, int64_t npix, int n, int rule,
                                                      double majority) {
    const int64_t pix = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (pix >= npix) return;
    double x[ERD_MAXN], D[ERD_MAXN][ERD_MAXN], md[ERD_MAXN];
    int size[ERD_MAXN], chain[ERD_MAXN], mlo[ERD_MAXN], mhi[ERD_MAXN], order[ERD_MAXN], parent[ERD_MAXN];
    for (int i = 0; i < n; ++i) x[i] = values[pix * n + i];
    for (int i = 0; i < n; ++i) {
        size[i] = 1;
        for (int j = 0; j < n; ++j) D[i][j] = fabs(x[i] - x[j]);
    }
    int chain_len = 0;
    for (int k = 0; k < n - 1; ++k) {
        if (chain_len == 0) {
            chain_len = 1;
            for (int i = 0; i < n; ++i)
                if (size[i] > 0) {
                    chain[0] = i;
                    break;
                }
        }
        int a, b;
        double cur;
        while (true) {
            a = chain[chain_len - 1];
            if (chain_len > 1) {
                b = chain[chain_len - 2];
                cur = D[a][b];
            } else {
                b = -1;
                cur = INFINITY;
            }
            for (int i = 0; i < n; ++i) {
                if (size[i] == 0 || i == a) continue;
                if (D[a][i] < cur) {
                    cur = D[a][i];
                    b = i;
                }
            }
            if (chain_len > 1 && b == chain[chain_len - 2]) break;
            chain[chain_len++] = b;
        }
        chain_len -= 2;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        mlo[k] = lo;
        mhi[k] = hi;
        md[k] = cur;
        size[hi] += size[lo];
        size[lo] = 0;
        for (int i = 0; i < n; ++i) {
            if (size[i] == 0 || i == hi) continue;
            const double d = fmax(D[i][lo], D[i][hi]);
            D[i][hi] = d;
            D[hi][i] = d;
        }
    }
    // stable sort of the n - 1 merges by distance (insertion sort keeps equal keys in order)
    for (int k = 0; k < n - 1; ++k) {
        int j = k;
        while (j > 0 && md[order[j - 1]] > md[k]) {
            order[j] = order[j - 1];
            --j;
        }
        order[j] = k;
    }
    for (int i = 0; i < n; ++i) parent[i] = i;
    auto find = [&](int i) {
        while (parent[i] != i) i = parent[i];
        return i;
    };
    for (int q = 0; q < n - 2; ++q) {
        const int k = order[q];
        parent[find(mlo[k])] = find(mhi[k]);
    }
    const int root = find(0);
    // the two clusters: in0 = with point 0, the rest
    double g0[ERD_MAXN], g1[ERD_MAXN];
    int n0 = 0, n1 = 0;
    bool in0[ERD_MAXN];
    for (int i = 0; i < n; ++i) {
        in0[i] = find(i) == root;
        if (in0[i]) g0[n0++] = x[i];
        else g1[n1++] = x[i];
    }
    bool drop0 = false, drop1 = false;      // reject cluster 0 / cluster 1
    if (rule == 1) {
        if ((double)n0 >= majority) drop1 = true;
        if ((double)n1 >= majority) drop0 = true;
    } else if (!erd_map || erd_map[pix] > 0.f) {
        const double m0 = erd_numpy_sum(g0, n0) / (double)n0, m1 = erd_numpy_sum(g1, n1) / (double)n1;
        if (m0 > m1) drop1 = true;
        if (m1 > m0) drop0 = true;
    }
    for (int i = 0; i < n; ++i) accept[pix * n + i] = (in0[i] ? drop0 : drop1) ? 0.f : 1.f;
}

int launch_auto_erd(float* accept, const double* values, const float* erd_map, int64_t npix, int n, int rule, hipStream_t st) {
    INR_REQUIRE(n >= 2 && n <= ERD_MAXN, INR_E_INVALID, "inr_auto_erd: 2 <= acquisitions <= %d (got %d)", ERD_MAXN, n);
    INR_REQUIRE(rule == 1 || rule == 2, INR_E_INVALID, "inr_auto_erd: rule must be 1 (majority voting) or 2 (intensity-cognisant)");
    if (npix == 0) return 0;
    const double majority = (2.0 / 3.0) * (double)n;     // master.py:86, evaluated as Python does
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(auto_erd_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(64), 0, st, accept, values, erd_map, npix, n, rule,
                       majority);
    INR_LAUNCH_CHECK();
    return 0;
}

}  // namespace inr
