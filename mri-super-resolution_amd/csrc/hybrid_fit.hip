// the three-compartment "hybrid" fit (PIA.py:240-283, called at superresHybrid.py:140).
// The reference loops over voxels in Python and calls scipy.optimize.curve_fit(method='trf') on each: 16 signals
// S(b, TE), 8 bounded parameters, forward-difference Jacobian.  Here one lane owns one voxel and runs the same
// trust-region-reflective iteration (scipy least_squares: trf_bounds, tr_solver='exact', x_scale=1,
// ftol=xtol=gtol=1e-8, max_nfev=5000) entirely in fp64: bound-aware 2-point Jacobian, Coleman-Li scaling, one SVD of
// the augmented scaled Jacobian per outer iteration (one-sided Jacobi on the 24x8 matrix instead of LAPACK gesdd),
// More' iteration on the secular equation, reflected / Cauchy candidate steps, the same radius update and
// termination tests.  oracle/pia_oracle.py is the line-by-line CPU twin.
// Per-lane working set (J 16x8, augmented copy 24x8, V 8x8) lives in private (scratch) memory; the problem is
// latency- not bandwidth-bound (a whole 120x120 slice is 225 waves).
#include "common.h"

namespace inr {

namespace {

constexpr int HN = 8;    // parameters
constexpr int HM = 16;   // residuals
constexpr int HA = HM + HN;
constexpr double H_EPS = 2.220446049250313e-16;
constexpr double H_TOL = 1e-8;       // ftol = xtol = gtol
constexpr int H_MAX_NFEV = 5000;

__constant__ double c_p0[HN] = {0.55, 1.3, 2.8, 50.0, 70.0, 750.0, 0.3, 0.4};     // PIA.py:269
__constant__ double c_lb[HN] = {0.3, 0.7, 2.7, 20.0, 40.0, 500.0, 0.0, 0.0};      // PIA.py:271
__constant__ double c_ub[HN] = {0.7, 1.7, 3.0, 70.0, 100.0, 1000.0, 1.0, 1.0};    // PIA.py:272
__constant__ double c_b[4] = {0.0, 150.0, 1000.0, 1500.0};                        // PIA.py:254
__constant__ double c_te[4] = {0.0, 13.0, 93.0, 143.0};                           // PIA.py:255

// residual f = 1000*(S_ep + S_st + S_lu) - y, signals ordered b-major (PIA.py:240-251, :263-265)
__device__ void residual(const double* __restrict__ p, const double* __restrict__ y, double* __restrict__ f) {
    double eb[3][4], et[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            eb[c][k] = exp(-c_b[k] / 1000.0 * p[c]);
            et[c][k] = exp(-c_te[k] / p[3 + c]);
        }
    const double vol[3] = {p[6], p[7], 1.0 - p[6] - p[7]};
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const double s = (vol[0] * eb[0][ib] * et[0][it] + vol[1] * eb[1][ib] * et[1][it]) +
                             vol[2] * eb[2][ib] * et[2][it];
            f[ib * 4 + it] = 1000.0 * s - y[ib * 4 + it];
        }
}

__device__ __forceinline__ double dot_n(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

// forward differences with scipy's bound handling (_numdiff.py: _compute_absolute_step, _adjust_scheme_to_bounds)
__device__ void fd_jacobian(const double* x, const double* f0, const double* y, double* J /*[HM][HN]*/) {
    double x1[HN], f1[HM];
    for (int i = 0; i < HN; ++i) x1[i] = x[i];
    const double rstep = 1.4901161193847656e-08;   // sqrt(eps)
    for (int i = 0; i < HN; ++i) {
        double h = rstep * (x[i] >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(x[i]));
        const double lower = x[i] - c_lb[i], upper = c_ub[i] - x[i];
        const double xh = x[i] + h;
        const bool violated = xh < c_lb[i] || xh > c_ub[i];
        const bool fitting = fabs(h) <= fmax(lower, upper);
        if (violated && fitting) h = -h;
        if (!fitting) h = upper >= lower ? upper : -lower;
        x1[i] = x[i] + h;
        const double dx = x1[i] - x[i];
        residual(x1, y, f1);
        for (int r = 0; r < HM; ++r) J[r * HN + i] = (f1[r] - f0[r]) / dx;
        x1[i] = x[i];
    }
}

__device__ void cl_scaling(const double* x, const double* g, double* v, double* dv) {
    for (int i = 0; i < HN; ++i) {
        v[i] = 1.0;
        dv[i] = 0.0;
        if (g[i] < 0.0) { v[i] = c_ub[i] - x[i]; dv[i] = -1.0; }
        if (g[i] > 0.0) { v[i] = x[i] - c_lb[i]; dv[i] = 1.0; }
    }
}

// smallest t >= 0 with x + t*s on a bound; hit[i] = sign(s[i]) for the components that reach it first
__device__ double step_to_bound(const double* x, const double* s, int* hit) {
    double steps[HN], t = INFINITY;
    for (int i = 0; i < HN; ++i) {
        steps[i] = s[i] != 0.0 ? fmax((c_lb[i] - x[i]) / s[i], (c_ub[i] - x[i]) / s[i]) : INFINITY;
        t = fmin(t, steps[i]);
    }
    if (hit)
        for (int i = 0; i < HN; ++i) hit[i] = steps[i] == t ? (s[i] > 0.0 ? 1 : (s[i] < 0.0 ? -1 : 0)) : 0;
    return t;
}

// (J*d) . s  for the scaled Jacobian J_h = J diag(d)
__device__ void jh_times(const double* J, const double* d, const double* s, double* out /*[HM]*/) {
    double ds[HN];
    for (int i = 0; i < HN; ++i) ds[i] = d[i] * s[i];
    for (int r = 0; r < HM; ++r) out[r] = dot_n(J + r * HN, ds, HN);
}

__device__ double eval_quadratic(const double* J, const double* d, const double* gh, const double* diag, const double* s) {
    double js[HM];
    jh_times(J, d, s, js);
    double q = dot_n(js, js, HM);
    for (int i = 0; i < HN; ++i) q += s[i] * diag[i] * s[i];
    return 0.5 * q + dot_n(s, gh, HN);
}

__device__ void min_quadratic_1d(double a, double b, double lo, double hi, double c, double* t_out, double* y_out) {
    double t = lo, yv = lo * (a * lo + b) + c;
    const double yh = hi * (a * hi + b) + c;
    if (yh < yv) { t = hi; yv = yh; }
    if (a != 0.0) {
        const double ext = -0.5 * b / a;
        if (lo < ext && ext < hi) {
            const double ye = ext * (a * ext + b) + c;
            if (ye < yv) { t = ext; yv = ye; }
        }
    }
    *t_out = t;
    *y_out = yv;
}

// One-sided Jacobi SVD of A [HA][HN] in place: on return the columns of A are u_k * s_k and V accumulates the
// rotations (A_in = U diag(s) V^T).
__device__ void jacobi_svd(double* A, double* V) {
    for (int i = 0; i < HN * HN; ++i) V[i] = 0.0;
    for (int i = 0; i < HN; ++i) V[i * HN + i] = 1.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < HN - 1; ++p)
            for (int q = p + 1; q < HN; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int r = 0; r < HA; ++r) {
                    const double a = A[r * HN + p], b = A[r * HN + q];
                    al += a * a;
                    be += b * b;
                    ga += a * b;
                }
                if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < HA; ++r) {
                    const double a = A[r * HN + p], b = A[r * HN + q];
                    A[r * HN + p] = c * a - s * b;
                    A[r * HN + q] = s * a + c * b;
                }
                for (int r = 0; r < HN; ++r) {
                    const double a = V[r * HN + p], b = V[r * HN + q];
                    V[r * HN + p] = c * a - s * b;
                    V[r * HN + q] = s * a + c * b;
                }
            }
        if (!rotated) break;
    }
}

__device__ double norm_n(const double* a, int n) { return sqrt(dot_n(a, a, n)); }

// scipy solve_lsq_trust_region (rtol = 0.01, max_iter = 10); s unsorted, smax/smin passed in
__device__ void solve_tr(const double* uf, const double* s, double smax, double smin, const double* V, double delta,
                         double* alpha_io, double* p /*[HN]*/) {
    double suf[HN], w[HN];
    for (int k = 0; k < HN; ++k) suf[k] = s[k] * uf[k];
    const bool full_rank = smin > H_EPS * HM * smax;
    if (full_rank) {
        for (int k = 0; k < HN; ++k) w[k] = uf[k] / s[k];
        for (int i = 0; i < HN; ++i) p[i] = -dot_n(V + i * HN, w, HN);
        if (norm_n(p, HN) <= delta) {
            *alpha_io = 0.0;
            return;
        }
    }
    double a_hi = norm_n(suf, HN) / delta, a_lo = 0.0;
    auto phi = [&](double alpha, double* ph, double* php) {
        double n2 = 0.0, sp = 0.0;
        for (int k = 0; k < HN; ++k) {
            const double den = s[k] * s[k] + alpha;
            const double q = suf[k] / den;
            n2 += q * q;
            sp += suf[k] * suf[k] / (den * den * den);
        }
        const double pn = sqrt(n2);
        *ph = pn - delta;
        *php = -sp / pn;
    };
    double f, fp;
    if (full_rank) {
        phi(0.0, &f, &fp);
        a_lo = -f / fp;
    }
    double alpha = *alpha_io;
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * a_hi, sqrt(a_lo * a_hi));
    for (int it = 0; it < 10; ++it) {
        if (alpha < a_lo || alpha > a_hi) alpha = fmax(0.001 * a_hi, sqrt(a_lo * a_hi));
        phi(alpha, &f, &fp);
        if (f < 0.0) a_hi = alpha;
        const double ratio = f / fp;
        a_lo = fmax(a_lo, alpha - ratio);
        alpha -= (f + delta) * ratio / delta;
        if (fabs(f) < 0.01 * delta) break;
    }
    for (int k = 0; k < HN; ++k) w[k] = suf[k] / (s[k] * s[k] + alpha);
    for (int i = 0; i < HN; ++i) p[i] = -dot_n(V + i * HN, w, HN);
    const double scale = delta / norm_n(p, HN);
    for (int i = 0; i < HN; ++i) p[i] *= scale;
    *alpha_io = alpha;
}

// scipy trf.select_step: plain, reflected or Cauchy step, whichever predicts the largest reduction
__device__ double select_step(const double* x, const double* J, const double* d, const double* diag, const double* gh,
                              const double* ph_in, double delta, double theta, double* step, double* step_h) {
    double p[HN], ph[HN];
    bool inside = true;
    for (int i = 0; i < HN; ++i) {
        ph[i] = ph_in[i];
        p[i] = d[i] * ph[i];
        const double xn = x[i] + p[i];
        inside = inside && xn >= c_lb[i] && xn <= c_ub[i];
    }
    if (inside) {
        for (int i = 0; i < HN; ++i) { step[i] = p[i]; step_h[i] = ph[i]; }
        return -eval_quadratic(J, d, gh, diag, ph);
    }
    int hit[HN];
    const double p_stride = step_to_bound(x, p, hit);
    double rh[HN], r[HN], x_on[HN];
    for (int i = 0; i < HN; ++i) {
        rh[i] = hit[i] != 0 ? -ph[i] : ph[i];
        r[i] = d[i] * rh[i];
        p[i] *= p_stride;
        ph[i] *= p_stride;
        x_on[i] = x[i] + p[i];
    }
    // intersect_trust_region(ph, rh, delta): positive root of |ph + t rh| = delta
    double to_tr;
    {
        const double a = dot_n(rh, rh, HN), b = dot_n(ph, rh, HN), c = dot_n(ph, ph, HN) - delta * delta;
        const double dd = sqrt(b * b - a * c);
        const double q = -(b + copysign(dd, b));
        const double t1 = q / a, t2 = c / q;
        to_tr = t1 < t2 ? t2 : t1;
    }
    const double to_bound = step_to_bound(x_on, r, nullptr);
    double r_stride = fmin(to_bound, to_tr), r_lo, r_hi;
    if (r_stride > 0.0) {
        r_lo = (1.0 - theta) * p_stride / r_stride;
        r_hi = r_stride == to_bound ? theta * to_bound : to_tr;
    } else {
        r_lo = 0.0;
        r_hi = -1.0;
    }
    double r_val = INFINITY;
    if (r_lo <= r_hi) {
        double vv[HM], uu[HM];
        jh_times(J, d, rh, vv);
        jh_times(J, d, ph, uu);
        double a = dot_n(vv, vv, HM), b = dot_n(gh, rh, HN), c = 0.5 * dot_n(uu, uu, HM) + dot_n(gh, ph, HN);
        double sd = 0.0, s0d = 0.0, s00 = 0.0;
        for (int i = 0; i < HN; ++i) {
            sd += rh[i] * diag[i] * rh[i];
            s0d += ph[i] * diag[i] * rh[i];
            s00 += ph[i] * diag[i] * ph[i];
        }
        a = 0.5 * (a + sd);
        b += dot_n(uu, vv, HM) + s0d;
        c += 0.5 * s00;
        double t;
        min_quadratic_1d(a, b, r_lo, r_hi, c, &t, &r_val);
        for (int i = 0; i < HN; ++i) {
            rh[i] = rh[i] * t + ph[i];
            r[i] = rh[i] * d[i];
        }
    }
    for (int i = 0; i < HN; ++i) {
        p[i] *= theta;
        ph[i] *= theta;
    }
    const double p_val = eval_quadratic(J, d, gh, diag, ph);
    double agh[HN], ag[HN];
    for (int i = 0; i < HN; ++i) {
        agh[i] = -gh[i];
        ag[i] = d[i] * agh[i];
    }
    const double tr2 = delta / norm_n(agh, HN);
    const double tb2 = step_to_bound(x, ag, nullptr);
    const double stride = tb2 < tr2 ? theta * tb2 : tr2;
    double ag_val, t_ag;
    {
        double vv[HM];
        jh_times(J, d, agh, vv);
        double a = dot_n(vv, vv, HM);
        for (int i = 0; i < HN; ++i) a += agh[i] * diag[i] * agh[i];
        a *= 0.5;
        const double b = dot_n(gh, agh, HN);
        min_quadratic_1d(a, b, 0.0, stride, 0.0, &t_ag, &ag_val);
    }
    if (p_val < r_val && p_val < ag_val) {
        for (int i = 0; i < HN; ++i) { step[i] = p[i]; step_h[i] = ph[i]; }
        return -p_val;
    }
    if (r_val < p_val && r_val < ag_val) {
        for (int i = 0; i < HN; ++i) { step[i] = r[i]; step_h[i] = rh[i]; }
        return -r_val;
    }
    for (int i = 0; i < HN; ++i) { step[i] = ag[i] * t_ag; step_h[i] = agh[i] * t_ag; }
    return -ag_val;
}

}  // namespace

__global__ void __launch_bounds__(64) hybrid_fit_kernel(double* __restrict__ params, int* __restrict__ status_out,
                                                        int* __restrict__ nfev_out, double* __restrict__ cost_out,
                                                        const double* __restrict__ signals, int64_t n) {
    const int64_t vox = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (vox >= n) return;
    double y[HM], x[HN], f[HM], J[HM * HN], g[HN];
    for (int r = 0; r < HM; ++r) y[r] = signals[vox * HM + r];
    for (int i = 0; i < HN; ++i) x[i] = c_p0[i];
    residual(x, y, f);
    int nfev = 1;
    fd_jacobian(x, f, y, J);
    double cost = 0.5 * dot_n(f, f, HM);
    auto gradient = [&]() {
        for (int i = 0; i < HN; ++i) {
            double s = 0.0;
            for (int r = 0; r < HM; ++r) s += J[r * HN + i] * f[r];
            g[i] = s;
        }
    };
    gradient();
    double v[HN], dv[HN];
    cl_scaling(x, g, v, dv);
    double delta = 0.0;
    for (int i = 0; i < HN; ++i) delta += (x[i] / sqrt(v[i])) * (x[i] / sqrt(v[i]));
    delta = sqrt(delta);
    if (delta == 0.0) delta = 1.0;
    double alpha = 0.0;
    int status = -1;
    while (true) {
        cl_scaling(x, g, v, dv);
        double g_norm = 0.0;
        for (int i = 0; i < HN; ++i) g_norm = fmax(g_norm, fabs(g[i] * v[i]));
        if (g_norm < H_TOL) status = 1;
        if (status >= 0 || nfev == H_MAX_NFEV) break;
        double d[HN], diag[HN], gh[HN];
        for (int i = 0; i < HN; ++i) {
            d[i] = sqrt(v[i]);
            diag[i] = g[i] * dv[i];
            gh[i] = d[i] * g[i];
        }
        double A[HA * HN], V[HN * HN], s[HN], uf[HN];
        for (int r = 0; r < HM; ++r)
            for (int i = 0; i < HN; ++i) A[r * HN + i] = J[r * HN + i] * d[i];
        for (int r = 0; r < HN; ++r)
            for (int i = 0; i < HN; ++i) A[(HM + r) * HN + i] = r == i ? sqrt(diag[i]) : 0.0;
        jacobi_svd(A, V);
        double smax = 0.0, smin = INFINITY;
        for (int k = 0; k < HN; ++k) {
            double n2 = 0.0, uy = 0.0;
            for (int r = 0; r < HA; ++r) n2 += A[r * HN + k] * A[r * HN + k];
            for (int r = 0; r < HM; ++r) uy += A[r * HN + k] * f[r];
            s[k] = sqrt(n2);
            uf[k] = s[k] > 0.0 ? uy / s[k] : 0.0;
            smax = fmax(smax, s[k]);
            smin = fmin(smin, s[k]);
        }
        const double theta = fmax(0.995, 1.0 - g_norm);
        double actual = -1.0, cost_new = cost;
        double x_new[HN], f_new[HM];
        while (actual <= 0.0 && nfev < H_MAX_NFEV) {
            double ph[HN], step[HN], step_h[HN];
            solve_tr(uf, s, smax, smin, V, delta, &alpha, ph);
            const double predicted = select_step(x, J, d, diag, gh, ph, delta, theta, step, step_h);
            for (int i = 0; i < HN; ++i) {   // make_strictly_feasible(rstep = 0)
                double xn = x[i] + step[i];
                const double lo = xn - c_lb[i], up = c_ub[i] - xn;
                if (lo <= fmin(up, 0.0)) xn = nextafter(c_lb[i], c_ub[i]);
                else if (up <= fmin(lo, 0.0)) xn = nextafter(c_ub[i], c_lb[i]);
                x_new[i] = xn;
            }
            residual(x_new, y, f_new);
            ++nfev;
            const double sh_norm = norm_n(step_h, HN);
            bool finite = true;
            for (int r = 0; r < HM; ++r) finite = finite && isfinite(f_new[r]);
            if (!finite) {
                delta = 0.25 * sh_norm;
                continue;
            }
            cost_new = 0.5 * dot_n(f_new, f_new, HM);
            actual = cost - cost_new;
            double ratio;
            if (predicted > 0.0) ratio = actual / predicted;
            else if (predicted == 0.0 && actual == 0.0) ratio = 1.0;
            else ratio = 0.0;
            double delta_new = delta;
            if (ratio < 0.25) delta_new = 0.25 * sh_norm;
            else if (ratio > 0.75 && sh_norm > 0.95 * delta) delta_new = 2.0 * delta;
            const double step_norm = norm_n(step, HN);
            const bool ft = actual < H_TOL * cost && ratio > 0.25;
            const bool xt = step_norm < H_TOL * (H_TOL + norm_n(x, HN));
            if (ft && xt) status = 4;
            else if (ft) status = 2;
            else if (xt) status = 3;
            if (status >= 0) break;
            alpha *= delta / delta_new;
            delta = delta_new;
        }
        if (actual > 0.0) {
            for (int i = 0; i < HN; ++i) x[i] = x_new[i];
            for (int r = 0; r < HM; ++r) f[r] = f_new[r];
            cost = cost_new;
            fd_jacobian(x, f, y, J);
            gradient();
        }
    }
    if (status < 0) status = 0;
    // curve_fit raises on status 0 and the reference falls back to p0 (PIA.py:276-277)
    for (int i = 0; i < HN; ++i) params[vox * HN + i] = status == 0 ? c_p0[i] : x[i];
    status_out[vox] = status;
    nfev_out[vox] = nfev;
    cost_out[vox] = cost;
}

int launch_hybrid_fit(double* params, int* status, int* nfev, double* cost, const double* signals, int64_t n,
                      hipStream_t st) {
    ProfScope ps(KC_OTHER, st);
    hipLaunchKernelGGL(hybrid_fit_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, params, status, nfev, cost,
                       signals, n);
    INR_LAUNCH_CHECK();
    return 0;
}

}  // namespace inr
