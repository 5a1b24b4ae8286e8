// the three-compartment "hybrid" fit (PIA.py:240-283, called at superresHybrid.py:140).
// The reference loops over voxels in Python and calls scipy.optimize.curve_fit(method='trf') on each: 16 signals
// S(b, TE), 8 bounded parameters, forward-difference Jacobian.  Here a group of eight lanes (default kernel, second half
// of this file) or a single lane (first half: the simple twin, kept as a cross-check) owns one voxel and runs the same
// trust-region-reflective iteration (scipy least_squares: trf_bounds, tr_solver='exact', x_scale=1,
// ftol=xtol=gtol=1e-8, max_nfev=5000) entirely in fp64: bound-aware 2-point Jacobian, Coleman-Li scaling, one SVD of
// the augmented scaled Jacobian per outer iteration (one-sided Jacobi on the 24x8 matrix instead of LAPACK gesdd),
// More' iteration on the secular equation, reflected / Cauchy candidate steps, the same radius update and
// termination tests.  oracle/pia_oracle.py is the line-by-line CPU twin.
// Measured (MI355X, 120x120 slice, 2 % noise): one lane per voxel 4.5 s (per-lane working set in scratch, the slowest
// voxel of the slice -- 5000 evaluations -- sets the time), eight lanes per voxel 0.28 s; 230,400 voxels 0.85 s
// (271 k fits/s; scipy on one host core: 104 fits/s).
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

// ---------------------------------------------------------------------------------------------------------------------
// Eight lanes per voxel.  The fit is a chain of a few dozen (worst case 5000) dependent iterations, so the time of a
// slice is the time of its slowest voxels: latency per iteration is what matters, not lanes in flight.  Lane j of a
// group owns parameter j / Jacobian column j / singular triplet j: the eight forward-difference columns are evaluated
// at once, 8-vectors live one component per lane (reductions = 3 xor-shuffles), and the one-sided Jacobi SVD runs as
// a round-robin tournament -- 7 rounds per sweep, 4 disjoint column pairs rotated at once, partners exchanged
// through LDS.  Every group-uniform scalar (cost, radius, alpha, termination) is computed redundantly on the 8 lanes
// from bitwise identical reductions, so control flow never diverges inside a group; different groups of a wave do
// diverge (the wave serialises their paths), which only costs throughput of an otherwise idle machine.
namespace g8 {

constexpr int OFF_A = 0, OFF_V = 192, OFF_J = 256, OFF_EX = 384, OFF_EXN = 408, OFF_F = 432, OFF_FN = 448, OFF_Y = 464;
constexpr int GSTRIDE = 481;   // doubles per group (odd multiple of a bank pair: groups land in different banks)

__device__ __forceinline__ double gsum(double v) {
    v += __shfl_xor(v, 1, 8);
    v += __shfl_xor(v, 2, 8);
    v += __shfl_xor(v, 4, 8);
    return v;
}
__device__ __forceinline__ double gmax(double v) {
    v = fmax(v, __shfl_xor(v, 1, 8));
    v = fmax(v, __shfl_xor(v, 2, 8));
    v = fmax(v, __shfl_xor(v, 4, 8));
    return v;
}
__device__ __forceinline__ double gmin(double v) {
    v = fmin(v, __shfl_xor(v, 1, 8));
    v = fmin(v, __shfl_xor(v, 2, 8));
    v = fmin(v, __shfl_xor(v, 4, 8));
    return v;
}
__device__ __forceinline__ bool gany(bool b) {
    int v = b ? 1 : 0;
    v |= __shfl_xor(v, 1, 8);
    v |= __shfl_xor(v, 2, 8);
    v |= __shfl_xor(v, 4, 8);
    return v != 0;
}
__device__ __forceinline__ double gget(double v, int k) { return __shfl(v, k, 8); }
// LDS traffic between lanes of ONE wave: in-order in hardware, this only pins the compiler
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the 24 exponentials of the model at parameters xj (3 per lane) -> ex[0..11] = exp(-b_k/1000 D_c), ex[12..23] = exp(-TE_k/T2_c)
__device__ __forceinline__ void model_exps(double xj, int j, double* ex) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = 3 * j + q, k = e & 3;
        const int c = (e >= 12 ? e - 12 : e) >> 2;
        const double pc = gget(xj, e >= 12 ? 3 + c : c);
        ex[e] = e >= 12 ? exp(-c_te[k] / pc) : exp(-c_b[k] / 1000.0 * pc);
    }
}

// residual rows 2j and 2j+1 from published exponentials (operation order of PIA.py:246-251, no fused multiply-add)
__device__ __forceinline__ void residual_pair(const double* ex, double xj, const double* y, int j, double* f0, double* f1) {
#pragma clang fp contract(off)
    const double v0 = gget(xj, 6), v1 = gget(xj, 7), v2 = 1.0 - v0 - v1;
    const int ib = j >> 1, it = (j & 1) * 2;
    const double e0 = ex[ib], e1 = ex[4 + ib], e2 = ex[8 + ib];
    const double sa = (v0 * e0 * ex[12 + it] + v1 * e1 * ex[16 + it]) + v2 * e2 * ex[20 + it];
    const double sb = (v0 * e0 * ex[13 + it] + v1 * e1 * ex[17 + it]) + v2 * e2 * ex[21 + it];
    *f0 = 1000.0 * sa - y[2 * j];
    *f1 = 1000.0 * sb - y[2 * j + 1];
}

// forward-difference Jacobian column j (scipy approx_derivative '2-point' with bounds), all 16 rows, in registers
__device__ __forceinline__ void jac_column(double xj, int j, double lb, double ub, const double* ex, const double* f,
                                           const double* y, double* Jc) {
#pragma clang fp contract(off)
    const double rstep = 1.4901161193847656e-08;
    double h = rstep * (xj >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(xj));
    const double lower = xj - lb, upper = ub - xj, xh = xj + h;
    const bool violated = xh < lb || xh > ub, fitting = fabs(h) <= fmax(lower, upper);
    if (violated && fitting) h = -h;
    if (!fitting) h = upper >= lower ? upper : -lower;
    const double x1 = xj + h, dx = x1 - xj;
    double ne[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ne[k] = j < 3 ? exp(-c_b[k] / 1000.0 * x1) : exp(-c_te[k] / (j < 6 ? x1 : 1.0));
    double eb[3][4], et[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            eb[c][k] = j == c ? ne[k] : ex[c * 4 + k];
            et[c][k] = j == 3 + c ? ne[k] : ex[12 + c * 4 + k];
        }
    const double p6 = gget(xj, 6), p7 = gget(xj, 7);
    const double v0 = j == 6 ? x1 : p6, v1 = j == 7 ? x1 : p7, v2 = 1.0 - v0 - v1;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const double s = (v0 * eb[0][ib] * et[0][it] + v1 * eb[1][ib] * et[1][it]) + v2 * eb[2][ib] * et[2][it];
            const int r = ib * 4 + it;
            Jc[r] = ((1000.0 * s - y[r]) - f[r]) / dx;
        }
}

// rows 2j, 2j+1 of (J diag(d)) s, J published column-major in LDS
__device__ __forceinline__ void jh_pair(const double* Jl, double ds, int j, double* o0, double* o1) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double w = gget(ds, i);
        a0 += Jl[i * 16 + 2 * j] * w;
        a1 += Jl[i * 16 + 2 * j + 1] * w;
    }
    *o0 = a0;
    *o1 = a1;
}

__device__ __forceinline__ double eval_quad(const double* Jl, double d, double gh, double diag, double s, int j) {
    double q0, q1;
    jh_pair(Jl, d * s, j, &q0, &q1);
    return 0.5 * gsum(q0 * q0 + q1 * q1 + s * diag * s) + gsum(s * gh);
}

__device__ __forceinline__ double bound_step(double x, double s, double lb, double ub, int* hit) {
    const double st = s != 0.0 ? fmax((lb - x) / s, (ub - x) / s) : INFINITY;
    const double t = gmin(st);
    if (hit) *hit = st == t ? (s > 0.0 ? 1 : (s < 0.0 ? -1 : 0)) : 0;
    return t;
}

}  // namespace g8

__global__ void __launch_bounds__(64) hybrid_fit_g8_kernel(double* __restrict__ params, int* __restrict__ status_out,
                                                           int* __restrict__ nfev_out, double* __restrict__ cost_out,
                                                           const double* __restrict__ signals, int64_t n) {
    using namespace g8;
    __shared__ double lds[8 * GSTRIDE];
    const int j = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const int64_t vox = (int64_t)blockIdx.x * 8 + grp;
    if (vox >= n) return;   // whole groups leave; nothing below synchronises across groups
    double* L = lds + grp * GSTRIDE;
    const double lb = c_lb[j], ub = c_ub[j];
    L[OFF_Y + 2 * j] = signals[vox * HM + 2 * j];
    L[OFF_Y + 2 * j + 1] = signals[vox * HM + 2 * j + 1];
    double x = c_p0[j];
    model_exps(x, j, L + OFF_EX);
    lds_sync();
    double f0, f1;
    residual_pair(L + OFF_EX, x, L + OFF_Y, j, &f0, &f1);
    L[OFF_F + 2 * j] = f0;
    L[OFF_F + 2 * j + 1] = f1;
    lds_sync();
    double cost = 0.5 * gsum(f0 * f0 + f1 * f1);
    int nfev = 1;
    double Jc[HM], g;
    auto new_jacobian = [&]() {
        jac_column(x, j, lb, ub, L + OFF_EX, L + OFF_F, L + OFF_Y, Jc);
        g = 0.0;
#pragma unroll
        for (int r = 0; r < HM; ++r) {
            L[OFF_J + j * 16 + r] = Jc[r];
            g += Jc[r] * L[OFF_F + r];
        }
        lds_sync();
    };
    new_jacobian();
    auto scaling = [&](double* v, double* dv) {
        *v = 1.0;
        *dv = 0.0;
        if (g < 0.0) { *v = ub - x; *dv = -1.0; }
        if (g > 0.0) { *v = x - lb; *dv = 1.0; }
    };
    double v, dv;
    scaling(&v, &dv);
    double delta = sqrt(gsum((x / sqrt(v)) * (x / sqrt(v))));
    if (delta == 0.0) delta = 1.0;
    double alpha = 0.0;
    int status = -1;
    while (true) {
        scaling(&v, &dv);
        const double g_norm = gmax(fabs(g * v));
        if (g_norm < H_TOL) status = 1;
        if (status >= 0 || nfev == H_MAX_NFEV) break;
        const double d = sqrt(v), diag = g * dv, gh = d * g;
        // ---- SVD of [J diag(d); diag(sqrt(diag))]: one-sided Jacobi, column j here, partners through LDS ------------
        double a[HA], vv[HN];
#pragma unroll
        for (int r = 0; r < HM; ++r) a[r] = Jc[r] * d;
#pragma unroll
        for (int r = 0; r < HN; ++r) {
            a[HM + r] = r == j ? sqrt(diag) : 0.0;
            vv[r] = r == j ? 1.0 : 0.0;
        }
        for (int sweep = 0; sweep < 40; ++sweep) {
            bool rot = false;
            for (int t = 0; t < 7; ++t) {
                int partner = j == 7 ? t : (j == t ? 7 : 2 * t - j);
                if (j != 7 && j != t) partner = partner < 0 ? partner + 7 : (partner >= 7 ? partner - 7 : partner);
#pragma unroll
                for (int r = 0; r < HA; ++r) L[OFF_A + j * HA + r] = a[r];
#pragma unroll
                for (int r = 0; r < HN; ++r) L[OFF_V + j * HN + r] = vv[r];
                lds_sync();
                double b[HA], vb[HN], own2 = 0.0, oth2 = 0.0, ga = 0.0;
#pragma unroll
                for (int r = 0; r < HA; ++r) {
                    b[r] = L[OFF_A + partner * HA + r];
                    own2 += a[r] * a[r];
                    oth2 += b[r] * b[r];
                    ga += a[r] * b[r];
                }
#pragma unroll
                for (int r = 0; r < HN; ++r) vb[r] = L[OFF_V + partner * HN + r];
                lds_sync();
                const bool first = j < partner;
                const double al = first ? own2 : oth2, be = first ? oth2 : own2;
                if (ga != 0.0 && ga * ga > 1e-30 * (al * be)) {
                    rot = true;
                    const double zeta = (be - al) / (2.0 * ga);
                    const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
                    const double sg = first ? -s : s;   // column p: c a_p - s a_q ; column q: s a_p + c a_q
#pragma unroll
                    for (int r = 0; r < HA; ++r) a[r] = c * a[r] + sg * b[r];
#pragma unroll
                    for (int r = 0; r < HN; ++r) vv[r] = c * vv[r] + sg * vb[r];
                }
            }
            if (!gany(rot)) break;
        }
        double s2 = 0.0, uy = 0.0;
#pragma unroll
        for (int r = 0; r < HA; ++r) s2 += a[r] * a[r];
#pragma unroll
        for (int r = 0; r < HM; ++r) uy += a[r] * L[OFF_F + r];
#pragma unroll
        for (int r = 0; r < HN; ++r) L[OFF_V + j * HN + r] = vv[r];
        lds_sync();
        const double sv = sqrt(s2), uf = sv > 0.0 ? uy / sv : 0.0, suf = sv * uf;
        const double smax = gmax(sv), smin = gmin(sv);
        const bool full_rank = smin > H_EPS * HM * smax;
        auto v_times = [&](double w) {   // component j of -V w
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < HN; ++k) acc += L[OFF_V + k * HN + j] * gget(w, k);
            return -acc;
        };
        const double theta = fmax(0.995, 1.0 - g_norm);
        double actual = -1.0, cost_new = cost, x_new = x, fn0 = f0, fn1 = f1;
        while (actual <= 0.0 && nfev < H_MAX_NFEV) {
            // ---- trust-region sub-problem (More') -----------------------------------------------------------------
            double ph;
            bool gauss_newton = false;
            if (full_rank) {
                ph = v_times(uf / sv);
                if (sqrt(gsum(ph * ph)) <= delta) {
                    alpha = 0.0;
                    gauss_newton = true;
                }
            }
            if (!gauss_newton) {
                double a_hi = sqrt(gsum(suf * suf)) / delta, a_lo = 0.0, fph, fpp;
                auto phi = [&](double al) {
                    const double den = sv * sv + al, q = suf / den;
                    const double pn = sqrt(gsum(q * q));
                    fph = pn - delta;
                    fpp = -gsum(suf * suf / (den * den * den)) / pn;
                };
                if (full_rank) {
                    phi(0.0);
                    a_lo = -fph / fpp;
                }
                if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * a_hi, sqrt(a_lo * a_hi));
                for (int it = 0; it < 10; ++it) {
                    if (alpha < a_lo || alpha > a_hi) alpha = fmax(0.001 * a_hi, sqrt(a_lo * a_hi));
                    phi(alpha);
                    if (fph < 0.0) a_hi = alpha;
                    const double ratio = fph / fpp;
                    a_lo = fmax(a_lo, alpha - ratio);
                    alpha -= (fph + delta) * ratio / delta;
                    if (fabs(fph) < 0.01 * delta) break;
                }
                ph = v_times(suf / (sv * sv + alpha));
                ph *= delta / sqrt(gsum(ph * ph));
            }
            // ---- select_step: plain / reflected / Cauchy -----------------------------------------------------------
            double step, step_h, predicted;
            {
                double p = d * ph;
                const double xn = x + p;
                if (!gany(xn < lb || xn > ub)) {
                    step = p;
                    step_h = ph;
                    predicted = -eval_quad(L + OFF_J, d, gh, diag, ph, j);
                } else {
                    int hit;
                    const double p_stride = bound_step(x, p, lb, ub, &hit);
                    double rh = hit != 0 ? -ph : ph, r = d * rh;
                    p *= p_stride;
                    double phs = ph * p_stride;
                    const double x_on = x + p;
                    double to_tr;
                    {
                        const double qa = gsum(rh * rh), qb = gsum(phs * rh), qc = gsum(phs * phs) - delta * delta;
                        const double dd = sqrt(qb * qb - qa * qc), q = -(qb + copysign(dd, qb));
                        const double t1 = q / qa, t2 = qc / q;
                        to_tr = t1 < t2 ? t2 : t1;
                    }
                    const double to_bound = bound_step(x_on, r, lb, ub, nullptr);
                    const double r_stride = fmin(to_bound, to_tr);
                    double r_lo, r_hi;
                    if (r_stride > 0.0) {
                        r_lo = (1.0 - theta) * p_stride / r_stride;
                        r_hi = r_stride == to_bound ? theta * to_bound : to_tr;
                    } else {
                        r_lo = 0.0;
                        r_hi = -1.0;
                    }
                    double r_val = INFINITY;
                    if (r_lo <= r_hi) {
                        double v0, v1, u0, u1;
                        jh_pair(L + OFF_J, d * rh, j, &v0, &v1);
                        jh_pair(L + OFF_J, d * phs, j, &u0, &u1);
                        const double qa = 0.5 * (gsum(v0 * v0 + v1 * v1) + gsum(rh * diag * rh));
                        const double qb = gsum(gh * rh) + gsum(u0 * v0 + u1 * v1) + gsum(phs * diag * rh);
                        const double qc = 0.5 * gsum(u0 * u0 + u1 * u1) + gsum(gh * phs) + 0.5 * gsum(phs * diag * phs);
                        double tq;
                        min_quadratic_1d(qa, qb, r_lo, r_hi, qc, &tq, &r_val);
                        rh = rh * tq + phs;
                        r = rh * d;
                    }
                    p *= theta;
                    phs *= theta;
                    const double p_val = eval_quad(L + OFF_J, d, gh, diag, phs, j);
                    double agh = -gh, ag = d * agh;
                    const double tr2 = delta / sqrt(gsum(agh * agh));
                    const double tb2 = bound_step(x, ag, lb, ub, nullptr);
                    const double stride = tb2 < tr2 ? theta * tb2 : tr2;
                    double ag_val, t_ag;
                    {
                        double v0, v1;
                        jh_pair(L + OFF_J, d * agh, j, &v0, &v1);
                        const double qa = 0.5 * (gsum(v0 * v0 + v1 * v1) + gsum(agh * diag * agh));
                        const double qb = gsum(gh * agh);
                        min_quadratic_1d(qa, qb, 0.0, stride, 0.0, &t_ag, &ag_val);
                    }
                    if (p_val < r_val && p_val < ag_val) {
                        step = p; step_h = phs; predicted = -p_val;
                    } else if (r_val < p_val && r_val < ag_val) {
                        step = r; step_h = rh; predicted = -r_val;
                    } else {
                        step = ag * t_ag; step_h = agh * t_ag; predicted = -ag_val;
                    }
                }
            }
            // ---- trial point ----------------------------------------------------------------------------------------
            {
                double xn = x + step;
                const double lo = xn - lb, up = ub - xn;
                if (lo <= fmin(up, 0.0)) xn = nextafter(lb, ub);
                else if (up <= fmin(lo, 0.0)) xn = nextafter(ub, lb);
                x_new = xn;
            }
            model_exps(x_new, j, L + OFF_EXN);
            lds_sync();
            residual_pair(L + OFF_EXN, x_new, L + OFF_Y, j, &fn0, &fn1);
            ++nfev;
            const double sh_norm = sqrt(gsum(step_h * step_h));
            if (gany(!isfinite(fn0) || !isfinite(fn1))) {
                delta = 0.25 * sh_norm;
                continue;
            }
            cost_new = 0.5 * gsum(fn0 * fn0 + fn1 * fn1);
            actual = cost - cost_new;
            double ratio;
            if (predicted > 0.0) ratio = actual / predicted;
            else if (predicted == 0.0 && actual == 0.0) ratio = 1.0;
            else ratio = 0.0;
            double delta_new = delta;
            if (ratio < 0.25) delta_new = 0.25 * sh_norm;
            else if (ratio > 0.75 && sh_norm > 0.95 * delta) delta_new = 2.0 * delta;
            const double step_norm = sqrt(gsum(step * step));
            const bool ft = actual < H_TOL * cost && ratio > 0.25;
            const bool xt = step_norm < H_TOL * (H_TOL + sqrt(gsum(x * x)));
            if (ft && xt) status = 4;
            else if (ft) status = 2;
            else if (xt) status = 3;
            if (status >= 0) break;
            alpha *= delta / delta_new;
            delta = delta_new;
        }
        if (actual > 0.0) {
            x = x_new;
            f0 = fn0;
            f1 = fn1;
            cost = cost_new;
#pragma unroll
            for (int q = 0; q < 3; ++q) L[OFF_EX + 3 * j + q] = L[OFF_EXN + 3 * j + q];
            L[OFF_F + 2 * j] = f0;
            L[OFF_F + 2 * j + 1] = f1;
            lds_sync();
            new_jacobian();
        }
    }
    if (status < 0) status = 0;
    params[vox * HN + j] = status == 0 ? c_p0[j] : x;   // PIA.py:276-277
    if (j == 0) {
        status_out[vox] = status;
        nfev_out[vox] = nfev;
        cost_out[vox] = cost;
    }
}

static tune_int g_hybrid_variant{1};   // 1 = eight lanes per voxel (default), 0 = one lane per voxel (cross-check)
void set_hybrid_variant(int v) { g_hybrid_variant = v; }

int launch_hybrid_fit(double* params, int* status, int* nfev, double* cost, const double* signals, int64_t n,
                      hipStream_t st) {
    ProfScope ps(KC_OTHER, st);
    if (g_hybrid_variant == 1)
        hipLaunchKernelGGL(hybrid_fit_g8_kernel, dim3((unsigned)((n + 7) / 8)), dim3(64), 0, st, params, status, nfev,
                           cost, signals, n);
    else
        hipLaunchKernelGGL(hybrid_fit_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, params, status, nfev,
                           cost, signals, n);
    INR_LAUNCH_CHECK();
    return 0;
}

}  // namespace inr
