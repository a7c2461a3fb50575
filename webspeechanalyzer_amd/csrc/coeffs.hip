// coeffs.hip — K5 (output_level 12 only): per-syllable polynomial coefficients, ONE LANE PER FIT.
//
// Stands in for make_coeffs h(e) + f(e,t,n,i) of the reference's inner module 4 (ref dist/main.js:2 @B34150, @B33793)
// and the slice of numeric.js 1.2.6 they use (inner module 5: dot* @B47148-48151, inv @B55496, gradient @B89174,
// uncmin @B89779): for every syllable four least-squares polynomials — 10 log10 of the frame energy sum (order 4),
// bin of formants 1 and 2 (order 3), bin of formant 3 (order 1) — each started from the normal equations (solution
// rounded to fp32, as `new Float32Array(...)` does), refined by BFGS with numerical gradients, then
// [coefficients..., rms error, points] -> 7 + 6 + 6 + 4 = 23 numbers per syllable.
// Every sum runs in numeric's loop order (dotVV: last element first, then pairs downwards; norm2: descending), the
// design matrix is in the ROW INDEX r while the residuals use r - first (the reference's own inconsistency), and
// Math.pow(integer, 0..4) is an exact product (checked against V8 for all bases below 3000; products stay below
// 2^53).  Level 12 stores what level 10 stores plus the energy sums (ref @B27240), so this kernel reads the
// compacted syllable rows, the straightened frames and the per-frame energy sum.
#include "wsa_internal.hpp"
#include "jsmath_device.hpp"

namespace wsa {

constexpr int CMAX = 5;                     // coefficients of the largest fit (order 4)
constexpr double kNumericEps = 2220446049250313e-31;

struct Fit {                                // one fit's points, compacted: y value and row index of every kept frame
    const double* ys; const double* rr; int stride; int m; double first;      // point k at [k * stride]: a lane's column of the wave's LDS block (stride 64), or its rows of the global scratch (stride 1)
    __device__ double y(int k) const { return ys[k * stride]; }
    __device__ double r(int k) const { return rr[k * stride]; }
};
constexpr int COEF_LDS_PTS = 32;            // points per fit the wave keeps in LDS (2 x 16 KB: four waves per CU); longer syllables read their points from the global scratch

__device__ inline double ipow(double t, int k) { double r = 1.0; for (int i = 0; i < k; i++) r *= t; return r; }   // Math.pow(t, k), exact here

// numeric.dotVV order over an implicit pair of sequences a(i), b(i), i < n
template <typename FA, typename FB>
__device__ inline double dot_vv(int n, FA a, FB b) {
    double r = a(n - 1) * b(n - 1);
    int i = n - 2;
    for (; i >= 1; i -= 2) r += a(i) * b(i) + a(i - 1) * b(i - 1);
    if (i == 0) r += a(0) * b(0);
    return r;
}

// d(e) of f(): sum over the points of (solve_poly(c, r - first) - y)^2   (solve_poly ref @B1521: ascending powers)
__device__ inline double cost(const Fit& F, const double* c, int n) {
    double cc[CMAX];
#pragma unroll
    for (int j = 0; j < CMAX; j++) cc[j] = j < n ? c[j] : 0.0;
    double t = 0.0;
    for (int k = 0; k < F.m; k++) {
        const double x = F.r(k) - F.first, y = F.y(k);
        double p = 0.0, pw = 1.0;
#pragma unroll
        for (int j = 0; j < CMAX; j++) if (j < n) { p += cc[j] * pw; pw *= x; }
        const double a = p - y;
        t += a * a;
    }
    return t;
}

// The FIRST trial of numeric.gradient for every coordinate in one pass over the points: f(x + h e_i) and f(x - h e_i), i < n (the first
// step h = max(1e-6 f0, 1e-8) is the same for all coordinates).  Each of the 2 n sums is cost()'s own operation sequence — the Horner-like
// chain p += c[j] * pw of a variant equals the unmodified chain up to j = i — , so the values are bit for bit those of 2 n separate cost()
// calls; what changes is the latency: the 2 n dependent chains run side by side and a point is loaded once (a gradient was 10 of the ~12
// evaluations of a BFGS step, one after the other: the kernel is as long as its slowest lane's chain of evaluations).
__device__ inline void cost_pm(const Fit& F, const double* x, int n, double h, double* f1, double* f2) {
    double tp[CMAX], tm[CMAX], cp[CMAX], cm[CMAX], c[CMAX];
#pragma unroll
    for (int i = 0; i < CMAX; i++) { tp[i] = tm[i] = 0.0; c[i] = i < n ? x[i] : 0.0; cp[i] = c[i] + h; cm[i] = c[i] - h; }
    for (int k = 0; k < F.m; k++) {
        const double xk = F.r(k) - F.first, y = F.y(k);
        double pw[CMAX], sj[CMAX], P[CMAX];          // pw_j = x^j as cost() forms it, s_j = c_j * pw_j, P_j = the chain in front of term j
        double p = 0.0, w = 1.0;
#pragma unroll
        for (int j = 0; j < CMAX; j++) { pw[j] = w; sj[j] = c[j] * w; P[j] = p; p += sj[j]; w *= xk; }
#pragma unroll
        for (int i = 0; i < CMAX; i++) {
            if (i >= n) continue;
            double pp = P[i] + cp[i] * pw[i], pm = P[i] + cm[i] * pw[i];
#pragma unroll
            for (int j = i + 1; j < CMAX; j++) if (j < n) { pp += sj[j]; pm += sj[j]; }
            const double ap = pp - y, am = pm - y;
            tp[i] += ap * ap; tm[i] += am * am;
        }
    }
#pragma unroll
    for (int i = 0; i < CMAX; i++) { f1[i] = tp[i]; f2[i] = tm[i]; }
}

// numeric.gradient; false when it would throw ("Numerical gradient fails" / NaN) — the reference's try/catch then
// drops the whole segment's rows, reported through *failed
// (f0 = f(x): numeric.gradient evaluates it itself; uncmin has just computed the same sum at the same x — the start value, or the accepted
//  point of its line search — so the value is handed in instead of walking the points once more)
__device__ inline bool gradient(const Fit& F, const double (&x)[CMAX], int n, double (&J)[CMAX], const double f0) {
    if (!(f0 == f0)) return false;
    double x0[CMAX];
#pragma unroll
    for (int i = 0; i < CMAX; i++) x0[i] = x[i];
    const double h0 = fmax(1e-6 * f0, 1e-8);
    double F1[CMAX], F2[CMAX];
    cost_pm(F, x, n, h0, F1, F2);
    int it = 0;
#pragma unroll
    for (int i = 0; i < CMAX; i++) {
        if (i >= n) continue;
        double h = h0;
        bool first = true;
        for (;;) {
            if (++it > 20) return false;
            double f1, f2;
            if (first) { f1 = F1[i]; f2 = F2[i]; first = false; }          // (a retry with h / 16 — errest > 1e-3 or a NaN — is rare: one by one, as before)
            else {
                x0[i] = x[i] + h; f1 = cost(F, x0, n);
                x0[i] = x[i] - h; f2 = cost(F, x0, n);
                x0[i] = x[i];
            }
            if (!(f1 == f1) || !(f2 == f2)) { h /= 16; continue; }
            J[i] = (f1 - f2) / (2 * h);
            const double t0 = x[i] - h, t1 = x[i], t2 = x[i] + h;
            const double d1 = (f1 - f0) / h, d2 = (f0 - f2) / h;
            const double N = fmax(fmax(fmax(fmax(fmax(fmax(fmax(fabs(J[i]), fabs(f0)), fabs(f1)), fabs(f2)), fabs(t0)), fabs(t1)), fabs(t2)), 1e-8);
            const double errest = fmin(fmax(fmax(fabs(d1 - J[i]), fabs(d2 - J[i])), fabs(d1 - d2)) / N, h / N);
            if (errest > 1e-3) h /= 16; else break;
        }
    }
    return true;
}

// The optimiser's vectors and its 5 x 5 matrix live in registers: every loop over the n <= 5 coefficients is written out over CMAX with the
// tail predicated off, and numeric.dotVV's order (last element first, then pairs downwards) is spelled out per length — indexed by a run-time
// n the arrays sat in scratch memory and every BFGS update was a chain of memory round trips.
__device__ inline bool all_finite(const double (&v)[CMAX], int n) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < CMAX; i++) if (i < n && !(fabs(v[i]) < __builtin_inf())) ok = false;
    return ok;
}
__device__ inline double norm2_r(const double (&x)[CMAX], int n) {          // numeric.norm2: descending
    double acc = 0.0;
#pragma unroll
    for (int i = CMAX - 1; i >= 0; i--) if (i < n) acc += x[i] * x[i];
    return sqrt(acc);
}
__device__ inline double dot_r(const double (&a)[CMAX], const double (&b)[CMAX], int n) {      // numeric.dotVV, n in 1 .. 5
    switch (n) {
        case 5: return (a[4] * b[4] + (a[3] * b[3] + a[2] * b[2])) + (a[1] * b[1] + a[0] * b[0]);
        case 4: return (a[3] * b[3] + (a[2] * b[2] + a[1] * b[1])) + a[0] * b[0];
        case 3: return a[2] * b[2] + (a[1] * b[1] + a[0] * b[0]);
        case 2: return a[1] * b[1] + a[0] * b[0];
        default: return a[0] * b[0];
    }
}

// numeric.uncmin(f, x0) with its defaults (tol 1e-8, maxit 1000, numeric gradient): BFGS + backtracking; x0 in/out
__device__ inline bool uncmin(const Fit& F, double (&x0)[CMAX], int n) {
    const double tol = fmax(1e-8, kNumericEps);
    const int maxit = 1000;
    double f0 = cost(F, x0, n);
    if (!(f0 == f0)) return false;
    double H[CMAX][CMAX], g0[CMAX], g1[CMAX], step[CMAX], s[CMAX], x1[CMAX], y[CMAX], Hy[CMAX];
#pragma unroll
    for (int i = 0; i < CMAX; i++) {
        g0[i] = g1[i] = step[i] = s[i] = x1[i] = y[i] = Hy[i] = 0.0;
#pragma unroll
        for (int j = 0; j < CMAX; j++) H[i][j] = i == j ? 1.0 : 0.0;
    }
    if (!gradient(F, x0, n, g0, f0)) return false;
    int it = 0;
    while (it < maxit) {
        if (!all_finite(g0, n)) break;
#pragma unroll
        for (int i = 0; i < CMAX; i++) if (i < n) step[i] = -dot_r(H[i], g0, n);
        if (!all_finite(step, n)) break;
        const double nstep = norm2_r(step, n);
        if (nstep < tol) break;
        double t = 1.0, f1 = f0;
        const double df0 = dot_r(g0, step, n);
#pragma unroll
        for (int i = 0; i < CMAX; i++) x1[i] = x0[i];
        while (it < maxit) {
            if (t * nstep < tol) break;
#pragma unroll
            for (int i = 0; i < CMAX; i++) if (i < n) { s[i] = step[i] * t; x1[i] = x0[i] + s[i]; }
            f1 = cost(F, x1, n);
            if (f1 - f0 >= 0.1 * t * df0 || !(f1 == f1)) { t *= 0.5; ++it; continue; }
            break;
        }
        if (t * nstep < tol) break;
        if (it == maxit) break;
        if (!gradient(F, x1, n, g1, f1)) return false;
#pragma unroll
        for (int i = 0; i < CMAX; i++) if (i < n) y[i] = g1[i] - g0[i];
        const double ys = dot_r(y, s, n);
#pragma unroll
        for (int i = 0; i < CMAX; i++) if (i < n) Hy[i] = dot_r(H[i], y, n);
        const double c = (ys + dot_r(y, Hy, n)) / (ys * ys);
#pragma unroll
        for (int i = 0; i < CMAX; i++)
#pragma unroll
            for (int j = 0; j < CMAX; j++) if (i < n && j < n)
                H[i][j] = (H[i][j] + c * (s[i] * s[j])) - (Hy[i] * s[j] + s[i] * Hy[j]) / ys;
#pragma unroll
        for (int i = 0; i < CMAX; i++) if (i < n) { x0[i] = x1[i]; g0[i] = g1[i]; }
        f0 = f1;
        ++it;
    }
    return true;
}

// numeric.inv on an n x n matrix (Gauss-Jordan, partial pivoting, its loop order); A is destroyed, I receives the inverse
__device__ inline bool inv(double (*A)[CMAX], double (*I)[CMAX], int n) {     // false: the reference throws (no pivot: a NaN column)
    int rowA[CMAX];                          // the reference swaps row REFERENCES; track the permutation instead
    for (int i = 0; i < n; i++) { rowA[i] = i; for (int j = 0; j < n; j++) I[i][j] = i == j ? 1.0 : 0.0; }
    for (int j = 0; j < n; j++) {
        int i0 = -1; double v0 = -1.0;
        for (int i = j; i < n; i++) { const double k = fabs(A[rowA[i]][j]); if (k > v0) { i0 = i; v0 = k; } }
        if (i0 < 0) return false;                 // every candidate is NaN: `d[-1]` is undefined in the reference -> TypeError
        const int t = rowA[i0]; rowA[i0] = rowA[j]; rowA[j] = t;
        double* Aj = A[rowA[j]]; double* Ij = I[rowA[j]];
        double x = Aj[j];
        for (int k = j; k < n; k++) Aj[k] /= x;
        for (int k = n - 1; k >= 0; k--) Ij[k] /= x;
        for (int i = n - 1; i >= 0; i--) if (i != j) {
            double* Ai = A[rowA[i]]; double* Ii = I[rowA[i]];
            x = Ai[j];
            for (int k = j + 1; k < n; k++) Ai[k] -= Aj[k] * x;
            for (int k = n - 1; k >= 0; k--) Ii[k] -= Ij[k] * x;
        }
    }
    // hand the rows back in position order
    double T[CMAX][CMAX];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) T[i][j] = I[rowA[i]][j];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) I[i][j] = T[i][j];
    return true;
}

__global__ __launch_bounds__(64) void coeffs_kernel(CoefParams p) {
    const uint32_t gid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t n_rows = p.totals[0];
    const uint32_t row = gid >> 2; const int q = (int)(gid & 3u);
    if (row >= n_rows) return;
    const int32_t* m = p.row_meta + (uint64_t)row * 8;
    const uint32_t clip = (uint32_t)m[0];
    const int st = m[6], sl = m[7];
    const uint64_t fbase = (uint64_t)p.frame_off[clip];                      // the clip's (stream's ring's) first row in the frame tables
    const uint64_t f0 = p.scratch_stride ? (uint64_t)clip * p.scratch_stride + ((uint32_t)st & p.ring_mask) : fbase + (uint32_t)st;   // first scratch row of the syllable
    // q = 0: 10 log10(energy sum), order 4; 1 / 2: bins of formants 1 / 2, order 3; 3: bin of formant 3, order 1
    const int order = q == 0 ? 4 : (q == 3 ? 1 : 3), n = order + 1;
    const int out_off = q == 0 ? 0 : (q == 1 ? 7 : (q == 2 ? 13 : 19));
    // the fit's points: in the wave's LDS block when the syllable is short enough (every evaluation of the cost function walks them:
    // out of the global scratch a point was two dependent memory round trips, ~10 x the arithmetic on it), else in the scratch rows
    __shared__ double s_y[COEF_LDS_PTS * 64], s_r[COEF_LDS_PTS * 64];
    const bool in_lds = sl <= COEF_LDS_PTS;
    double* ys = in_lds ? s_y + threadIdx.x : p.ws + ((uint64_t)(2 * q) * p.total_frames + f0);
    double* rr = in_lds ? s_r + threadIdx.x : p.ws + ((uint64_t)(2 * q + 1) * p.total_frames + f0);
    const int stride = in_lds ? 64 : 1;
    int cnt = 0; double first = -1.0;
    for (int r = 0; r < sl; r++) {
        const uint64_t fr_ = fbase + (((uint32_t)st + (uint32_t)r) & p.ring_mask);
        const float v = q == 0 ? p.sums[fr_] : p.formants[fr_ * 9 + 3 * (q - 1)];
        if (v > 0.f) {
            if (first < 0) first = (double)r;
            ys[cnt * stride] = q == 0 ? 10 * jsm::log10((double)v) : (double)v;
            rr[cnt * stride] = (double)r;
            cnt++;
        }
    }
    double* out = p.row_feat + (uint64_t)row * WSA_NFEAT + out_off;
    // slot 23 is the row's `numeric threw` marker (0 from the level-10 row, 1.0 set below by whichever fit fails)
    if (q == 0) for (int j = 24; j < WSA_NFEAT; j++) p.row_feat[(uint64_t)row * WSA_NFEAT + j] = 0.0;
    if (cnt <= 2) { for (int j = 0; j < n; j++) out[j] = 0.0; out[n] = 0.0; out[n + 1] = (double)cnt; return; }
    Fit F; F.ys = ys; F.rr = rr; F.stride = stride; F.m = cnt; F.first = first;
    // normal equations in the row index r: (X^T X) c = X^T y, X[k][e] = r_k^e
    double A[CMAX][CMAX], I[CMAX][CMAX], b[CMAX], c[CMAX];
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) A[i][j] = dot_vv(cnt, [&](int k) { return ipow(F.r(k), i); }, [&](int k) { return ipow(F.r(k), j); });
        b[i] = dot_vv(cnt, [&](int k) { return ipow(F.r(k), i); }, [&](int k) { return F.y(k); });
    }
    const bool inv_ok = inv(A, I, n);
    for (int i = 0; i < CMAX; i++) c[i] = 0.0;
    for (int i = 0; i < n; i++) c[i] = inv_ok ? (double)(float)dot_vv(n, [&](int k) { return I[i][k]; }, [&](int k) { return b[k]; }) : 0.0;   // new Float32Array(...)
    if (!inv_ok || !uncmin(F, c, n)) {
        // numeric threw (NaN cost after a singular normal matrix — e.g. three points for five coefficients — or "Numerical
        // gradient fails"): make_coeffs' try / catch then returns the rows collected so far, i.e. this syllable and the
        // segment's later ones are not reported.  The row is marked; the hosts cut the segment's feature list there.
        p.row_feat[(uint64_t)row * WSA_NFEAT + 23] = 1.0;
        for (int j = 0; j < n + 2; j++) out[j] = __builtin_nan("");
        return;
    }
    for (int j = 0; j < n; j++) out[j] = c[j];
    out[n] = sqrt(cost(F, c, n)) / (double)cnt;
    out[n + 1] = (double)cnt;
}

void launch_coeffs(const CoefParams& p, uint32_t rows_cap, hipStream_t s) {
    if (rows_cap == 0) return;
    hipLaunchKernelGGL(coeffs_kernel, dim3((rows_cap * 4 + 63) / 64), dim3(64), 0, s, p);
}

}  // namespace wsa
