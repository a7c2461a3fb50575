/*
 * frontend.c — ORACLE (test infrastructure only; see wsa_oracle.h).
 *
 * PCM -> window -> FFT -> power -> mel bands -> emphasis/gain -> Uint32 frame: the stage the
 * reference runs inside its "spectrum-processor" AudioWorklet.  That worklet's source
 * (analyzernode.min.js, formantanalyzer@1.1.6) is fetched from unpkg at run time
 * (/root/reference/dist/main.js:2 @B6480, configured @B6726, consumed @B8568) and is NOT in the
 * reference tree, so this half is OUR specification: PARITY UNPINNED against
 * the reference.  What is pinned: the wire contract (one Uint32Array(spec_bands) per frame,
 * @B8568; the config fields, @B6726) and the observed dynamic range of the 53 features
 * (dist/nnmodel/1/cats_emotion/model_meta.json).  The HIP kernel must equal this file BIT FOR BIT
 * on the u32 frames, and this file is checked against an fp64 textbook DFT in the tests.
 *
 * The arithmetic is a fixed fp32 dataflow graph ("FE-1", DESIGN.md §front end):
 *   - tables (window, twiddles, mel weights) are computed in fp64 and rounded once to fp32;
 *   - data path uses only fp32 + - * and explicit fmaf; compile with -ffp-contract=off;
 *   - NFFT = the smallest of {2^k, 3 * 2^k} >= max(window, ceil(fs * N_fft_bins / f_max)), >= 256 (F2: bin width
 *     ~ f_max / N_fft_bins, ref index.html:269; 1024 at 16 kHz, 3072 at 44.1 / 48 kHz);
 *   - real FFT of NFFT points = complex DIF FFT of N2 = NFFT/2 packed points.  N2 = 64 R: radices [R, 8, 8];
 *     N2 = 3 * 64 R: one radix-3 DIF stage over the thirds (forms in radix3() below) with twiddle W_N2^{n k3},
 *     then the three 64 R-point FFTs with radices [R, 8, 8]; Z[3 k' + k3] is output k' of the k3-th of them.
 *     Each radix-r butterfly (r = 2^p) is p radix-2 DIF stages with the internal twiddle forms below,
 *     inter-pass twiddles by the generic complex multiply `cmul`;
 *   - un-normalised power, mel sum as an fmaf chain in ascending bin order.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "wsa_oracle.h"

#define PI 3.14159265358979323846

struct wsa_or_fe {
    wsa_or_fe_cfg cfg;
    int32_t win, hop, nfft, n2, R, kmax, bands;
    int32_t three, M;         /* N2 = 3 M (three = 1) or N2 = M; M = 64 R     */
    float *window;            /* win */
    float *tw_n2;             /* 2*N2: W_N2^j = (cos, -sin), j < N2          */
    float *tw_m;              /* 2*M:  W_M^j (= tw_n2 when N2 = M)            */
    float *tw_64;             /* 2*64                                         */
    float *tw_nfft;           /* 2*(kmax+1): W_NFFT^k for the real split      */
    int32_t *mel_k0, *mel_cnt, *mel_off;
    float *mel_w;             /* 0.25 * triangle weight, flat                 */
    float *emph;              /* bands: 1 + m*high_f_emph                     */
    float gain;
    double *bins_hz;
};

/* exact-symmetry twiddle: W_N^j = (cos(2 pi j/N), -sin(2 pi j/N)) reduced to the first octant in
 * integer arithmetic so that quadrant/octant points are exact and symmetric partners are equal */
static void twiddle(int32_t j, int32_t N, float *wr, float *wi) {
    j %= N;
    int32_t q = (int32_t)(((int64_t)j * 8) / N);         /* octant 0..7 */
    int64_t r8 = (int64_t)j * 8 - (int64_t)q * N;         /* remainder numerator (angle = (q + r8/N) * pi/4) */
    double c, s;
    if (r8 == 0) {
        static const double C[8] = {1, 0.70710678118654752440, 0, -0.70710678118654752440, -1,
                                    -0.70710678118654752440, 0, 0.70710678118654752440};
        static const double S[8] = {0, 0.70710678118654752440, 1, 0.70710678118654752440, 0,
                                    -0.70710678118654752440, -1, -0.70710678118654752440};
        c = C[q]; s = S[q];
    } else {
        /* angle inside octant: t in (0, pi/4); fold odd octants */
        int64_t num = (q & 1) ? ((int64_t)N - r8) : r8;   /* distance to the nearer axis, exact */
        double tt = (double)num / (double)N * (PI / 4);
        double ca = cos(tt), sa = sin(tt);                /* ca > sa > 0 */
        switch (q) {
            case 0: c = ca; s = sa; break;
            case 1: c = sa; s = ca; break;
            case 2: c = -sa; s = ca; break;
            case 3: c = -ca; s = sa; break;
            case 4: c = -ca; s = -sa; break;
            case 5: c = -sa; s = -ca; break;
            case 6: c = sa; s = -ca; break;
            default: c = ca; s = -sa; break;
        }
    }
    *wr = (float)c; *wi = (float)(-s);
}

static int32_t ilog2(int32_t x) { int32_t l = 0; while ((1 << l) < x) l++; return l; }
static double mel_of(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double hz_of(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

int32_t wsa_or_fe_nfft_for(double fs, double window_width, int32_t n_fft_bins, double f_max) {
    int32_t win = (int32_t)floor(fs * window_width / 1000.0 + 0.5);
    int32_t need = (int32_t)ceil(fs * n_fft_bins / f_max);
    if (win > need) need = win;
    int32_t n = 256;                                       /* 256, 384, 512, 768, 1024, 1536, ... */
    for (;;) {
        if (n >= need) return n;
        if (n / 2 * 3 >= need) return n / 2 * 3;
        n <<= 1;
    }
}

wsa_or_fe *wsa_or_fe_new(const wsa_or_fe_cfg *cfg) {
    wsa_or_fe *f = calloc(1, sizeof(*f));
    f->cfg = *cfg;
    f->win = (int32_t)floor(cfg->fs * cfg->window_width / 1000.0 + 0.5);
    f->hop = (int32_t)floor(cfg->fs * cfg->window_step / 1000.0 + 0.5);
    f->nfft = wsa_or_fe_nfft_for(cfg->fs, cfg->window_width, cfg->n_fft_bins, cfg->f_max);
    f->n2 = f->nfft / 2;
    f->three = (f->n2 % 3) == 0;
    f->M = f->three ? f->n2 / 3 : f->n2;
    f->R = f->M / 64;
    f->kmax = (int32_t)floor(cfg->f_max * f->nfft / cfg->fs);
    if (f->kmax > f->n2) f->kmax = f->n2;
    f->bands = cfg->spec_type == 1 ? cfg->n_mel_bins : cfg->n_fft_bins;
    if (f->win < 2 || f->hop < 1 || f->R < (f->three ? 1 : 2) || f->R > 64 || f->bands < 1 ||
        (cfg->spec_type != 1 && cfg->n_fft_bins > f->n2 + 1)) { free(f); return NULL; }
    /* F3: periodic Hann */
    f->window = malloc(sizeof(float) * (size_t)f->win);
    for (int32_t n = 0; n < f->win; n++) f->window[n] = (float)(0.5 - 0.5 * cos(2.0 * PI * n / f->win));
    f->tw_n2 = malloc(sizeof(float) * 2 * (size_t)f->n2);
    for (int32_t j = 0; j < f->n2; j++) twiddle(j, f->n2, &f->tw_n2[2 * j], &f->tw_n2[2 * j + 1]);
    f->tw_m = f->tw_n2;
    if (f->three) {
        f->tw_m = malloc(sizeof(float) * 2 * (size_t)f->M);
        for (int32_t j = 0; j < f->M; j++) twiddle(j, f->M, &f->tw_m[2 * j], &f->tw_m[2 * j + 1]);
    }
    f->tw_64 = malloc(sizeof(float) * 2 * 64);
    for (int32_t j = 0; j < 64; j++) twiddle(j, 64, &f->tw_64[2 * j], &f->tw_64[2 * j + 1]);
    f->tw_nfft = malloc(sizeof(float) * 2 * (size_t)(f->kmax + 1));
    for (int32_t k = 0; k <= f->kmax; k++) twiddle(k, f->nfft, &f->tw_nfft[2 * k], &f->tw_nfft[2 * k + 1]);
    f->gain = (float)cfg->pre_norm_gain;
    f->emph = malloc(sizeof(float) * (size_t)f->bands);
    for (int32_t m = 0; m < f->bands; m++) f->emph[m] = (float)(1.0 + m * cfg->high_f_emph);
    f->bins_hz = malloc(sizeof(double) * (size_t)f->bands);
    if (cfg->spec_type == 1) {
        /* F5: HTK-mel triangles, unit peak, sampled at bin centres; an empty (narrower than one
         * bin) triangle takes the linearly interpolated power at its centre */
        int32_t M = f->bands;
        double df = cfg->fs / f->nfft, mlo = mel_of(cfg->f_min), mhi = mel_of(cfg->f_max);
        double *pts = malloc(sizeof(double) * (size_t)(M + 2));
        for (int32_t j = 0; j < M + 2; j++) pts[j] = hz_of(mlo + (mhi - mlo) * j / (M + 1));
        f->mel_k0 = malloc(sizeof(int32_t) * (size_t)M);
        f->mel_cnt = malloc(sizeof(int32_t) * (size_t)M);
        f->mel_off = malloc(sizeof(int32_t) * (size_t)M);
        f->mel_w = malloc(sizeof(float) * (size_t)M * (size_t)(f->kmax + 2));
        int32_t off = 0;
        for (int32_t m = 0; m < M; m++) {
            double lo = pts[m], ce = pts[m + 1], hi = pts[m + 2];
            f->bins_hz[m] = ce;
            int32_t k0 = -1, cnt = 0;
            for (int32_t k = 0; k <= f->kmax; k++) {
                double fk = k * df, w = 0;
                if (fk > lo && fk < hi) {
                    double up = (fk - lo) / (ce - lo), dn = (hi - fk) / (hi - ce);
                    w = up < dn ? up : dn;
                }
                if (w > 0) { if (k0 < 0) k0 = k; f->mel_w[off + (k - k0)] = (float)w * 0.25f; cnt = k - k0 + 1; }
            }
            if (k0 < 0) {
                double pos = ce / df;
                k0 = (int32_t)floor(pos);
                double fr = pos - k0;
                if (k0 >= f->kmax) { k0 = f->kmax; fr = 0; }
                f->mel_w[off] = (float)(1.0 - fr) * 0.25f; cnt = 1;
                if (fr > 0) { f->mel_w[off + 1] = (float)fr * 0.25f; cnt = 2; }
            }
            f->mel_k0[m] = k0; f->mel_cnt[m] = cnt; f->mel_off[m] = off; off += cnt;
        }
        free(pts);
    } else {
        for (int32_t m = 0; m < f->bands; m++) f->bins_hz[m] = m * cfg->fs / f->nfft;
    }
    return f;
}

void wsa_or_fe_free(wsa_or_fe *f) {
    if (!f) return;
    if (f->tw_m != f->tw_n2) free(f->tw_m);
    free(f->window); free(f->tw_n2); free(f->tw_64); free(f->tw_nfft);
    free(f->mel_k0); free(f->mel_cnt); free(f->mel_off); free(f->mel_w); free(f->emph); free(f->bins_hz);
    free(f);
}

int32_t wsa_or_fe_bands(const wsa_or_fe *f) { return f->bands; }
int32_t wsa_or_fe_nfft(const wsa_or_fe *f) { return f->nfft; }
int32_t wsa_or_fe_win(const wsa_or_fe *f) { return f->win; }
int32_t wsa_or_fe_hop(const wsa_or_fe *f) { return f->hop; }
int32_t wsa_or_fe_kmax(const wsa_or_fe *f) { return f->kmax; }
const double *wsa_or_fe_bins_hz(const wsa_or_fe *f) { return f->bins_hz; }
int32_t wsa_or_fe_n_frames(const wsa_or_fe *f, int64_t n_samples) {
    if (n_samples < f->win) return 0;
    return (int32_t)((n_samples - f->win) / f->hop) + 1;                        /* F1: tail dropped */
}
/* table access for the table-equality test against the product's host-side plan */
const float *wsa_or_fe_table(const wsa_or_fe *f, int32_t which, int32_t *n) {
    switch (which) {
        case 0: *n = f->win; return f->window;
        case 1: *n = 2 * f->n2; return f->tw_n2;
        case 2: *n = 2 * 64; return f->tw_64;
        case 3: *n = 2 * (f->kmax + 1); return f->tw_nfft;
        case 4: *n = f->mel_off ? f->mel_off[f->bands - 1] + f->mel_cnt[f->bands - 1] : 0; return f->mel_w;
        case 5: *n = 2 * f->M; return f->tw_m;
        default: *n = 0; return NULL;
    }
}

/* generic complex multiply (x * w), the ONLY form used for table twiddles */
static inline void cmul(float xr, float xi, float wr, float wi, float *yr, float *yi) {
    *yr = fmaf(-xi, wi, xr * wr);
    *yi = fmaf(xi, wr, xr * wi);
}

/* in-place radix-r DIF butterfly on v[0..r-1] (stride 1), r = 2^p <= 64: p radix-2 stages; the
 * (u - w) output of the stage with half-size h is multiplied by W_{2h}^j using
 *   j = 0: nothing;  j = h/2: (re,im) -> (im,-re);  j = h/4: s(re+im), s(im-re);
 *   j = 3h/4: s(im-re), -(s(re+im));  otherwise cmul with the W_64 table (W_{2h}^j = W_64^{j*32/h}).
 * Output is bit-reversed: v[bitrev_p(k)] = DFT_r[k]. */
static void butterfly(float *vr, float *vi, int32_t r, const float *tw64) {
    const float s = 0.70710678118654752440f;
    for (int32_t h = r / 2; h >= 1; h /= 2)
        for (int32_t blk = 0; blk < r; blk += 2 * h)
            for (int32_t j = 0; j < h; j++) {
                int32_t a = blk + j, b = a + h;
                float ur = vr[a], ui = vi[a], wr = vr[b], wi = vi[b];
                vr[a] = ur + wr; vi[a] = ui + wi;
                float tr = ur - wr, ti = ui - wi;
                if (j == 0) { vr[b] = tr; vi[b] = ti; }
                else if (2 * j == h) { vr[b] = ti; vi[b] = -tr; }
                else if (4 * j == h) { vr[b] = s * (tr + ti); vi[b] = s * (ti - tr); }
                else if (4 * j == 3 * h) { vr[b] = s * (ti - tr); vi[b] = -(s * (tr + ti)); }
                else cmul(tr, ti, tw64[2 * (j * 32 / h)], tw64[2 * (j * 32 / h) + 1], &vr[b], &vi[b]);
            }
}

static int32_t bitrev(int32_t x, int32_t bits) {
    int32_t r = 0;
    for (int32_t i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

/* one DIF pass: N points at x (stride 1), radix r, M = N/r sub-length; twiddle table tw = W_N */
static void dif_pass(float *xr, float *xi, int32_t N, int32_t r, const float *tw, const float *tw64) {
    int32_t M = N / r, p = ilog2(r);
    float vr[64], vi[64];
    for (int32_t m = 0; m < M; m++) {
        for (int32_t a = 0; a < r; a++) { vr[a] = xr[a * M + m]; vi[a] = xi[a * M + m]; }
        butterfly(vr, vi, r, tw64);
        for (int32_t k = 0; k < r; k++) {
            float yr = vr[bitrev(k, p)], yi = vi[bitrev(k, p)];
            if (tw && k > 0 && m > 0) cmul(yr, yi, tw[2 * (m * k)], tw[2 * (m * k) + 1], &yr, &yi);
            xr[k * M + m] = yr; xi[k * M + m] = yi;
        }
    }
}

/* radix-3 DIF butterfly (x0, x1, x2) -> (y0, y1, y2), W_3 = -1/2 - i sqrt(3)/2:
 *   t = x1 + x2;  y0 = x0 + t;  m = fma(-1/2, t, x0);  s = fl32(sqrt(3)/2) * (x1 - x2)   (per component)
 *   y1 = m - i s = (m.re + s.im, m.im - s.re);  y2 = m + i s = (m.re - s.im, m.im + s.re) */
static void radix3(const float *xr, const float *xi, float *yr, float *yi) {
    const float c3 = 0.86602540378443864676f;
    float tr = xr[1] + xr[2], ti = xi[1] + xi[2];
    yr[0] = xr[0] + tr; yi[0] = xi[0] + ti;
    float mr = fmaf(-0.5f, tr, xr[0]), mi = fmaf(-0.5f, ti, xi[0]);
    float sr = c3 * (xr[1] - xr[2]), si = c3 * (xi[1] - xi[2]);
    yr[1] = mr + si; yi[1] = mi - sr;
    yr[2] = mr - si; yi[2] = mi + sr;
}

/* M-point complex FFT, M = 64 R, radices [R, 8, 8]; the result is digit-reversed: Z[a + R (b + 8 c)] at a*64 + b*8 + c */
static void fft_m(const wsa_or_fe *f, float *zr, float *zi) {
    const int32_t R = f->R;
    if (R > 1) dif_pass(zr, zi, f->M, R, f->tw_m, f->tw_64);               /* pass 1: radix R, W_M */
    for (int32_t a = 0; a < R; a++) {
        dif_pass(zr + a * 64, zi + a * 64, 64, 8, f->tw_64, f->tw_64);      /* pass 2: radix 8, W_64 */
        for (int32_t b = 0; b < 8; b++)
            dif_pass(zr + a * 64 + b * 8, zi + a * 64 + b * 8, 8, 8, NULL, f->tw_64);   /* pass 3 */
    }
}

/* 4x power spectrum P'[0..kmax] of one frame (F1-F4) */
void wsa_or_fe_power4(const wsa_or_fe *f, const float *pcm, float *P) {
    const int32_t N2 = f->n2, R = f->R, M = f->M;
    float *zr = calloc((size_t)N2 * 2, sizeof(float)), *zi = zr + N2;
    for (int32_t n = 0; n < f->win; n++) {
        float xw = pcm[n] * f->window[n];
        if (n & 1) zi[n >> 1] = xw; else zr[n >> 1] = xw;
    }
    if (f->three) {
        /* radix-3 stage over the thirds of the packed frame: y_k3[n] = (sum_j z[n + j M] W_3^{j k3}) W_N2^{n k3} */
        for (int32_t n = 0; n < M; n++) {
            float xr[3] = {zr[n], zr[n + M], zr[n + 2 * M]}, xi[3] = {zi[n], zi[n + M], zi[n + 2 * M]}, yr[3], yi[3];
            radix3(xr, xi, yr, yi);
            for (int32_t k3 = 1; k3 < 3; k3++)
                if (n > 0) cmul(yr[k3], yi[k3], f->tw_n2[2 * (n * k3)], f->tw_n2[2 * (n * k3) + 1], &yr[k3], &yi[k3]);
            for (int32_t k3 = 0; k3 < 3; k3++) { zr[k3 * M + n] = yr[k3]; zi[k3 * M + n] = yi[k3]; }
        }
        for (int32_t k3 = 0; k3 < 3; k3++) fft_m(f, zr + k3 * M, zi + k3 * M);
    } else fft_m(f, zr, zi);
    /* Z[k] of the N2-point transform: k = 3 k' + k3 (or k = k'), output k' of sub-transform k3 digit-reversed */
#define ZSUB(kk) ((((kk) % R) * 64) + ((((kk) / R) % 8) * 8) + ((kk) / (R * 8)))
#define ZPOS(k) (f->three ? ((k) % 3) * M + ZSUB((k) / 3) : ZSUB(k))
    for (int32_t k = 0; k <= f->kmax; k++) {
        int32_t pa = ZPOS(k % N2), pb = ZPOS((N2 - k) % N2);
        float ar = zr[pa], ai = zi[pa], br = zr[pb], bi = -zi[pb];
        float er = ar + br, ei = ai + bi, orr = ar - br, oi = ai - bi;     /* 2E, 2O */
        float tr, ti;
        cmul(orr, oi, f->tw_nfft[2 * k], f->tw_nfft[2 * k + 1], &tr, &ti);
        float xr = er + ti, xi = ei - tr;                                  /* 2 X[k] */
        P[k] = fmaf(xr, xr, xi * xi);
    }
#undef ZPOS
#undef ZSUB
    free(zr);
}

static uint32_t to_u32(float x) {                                           /* F8 */
    if (!(x > 0)) return 0;                                                 /* also NaN */
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

/* one frame -> bands u32 (F5-F8) */
void wsa_or_fe_frame(const wsa_or_fe *f, const float *pcm, uint32_t *out) {
    float *P = malloc(sizeof(float) * (size_t)(f->kmax + 2));
    wsa_or_fe_power4(f, pcm, P);
    for (int32_t m = 0; m < f->bands; m++) {
        float e;
        if (f->cfg.spec_type == 1) {
            e = 0;
            const float *w = f->mel_w + f->mel_off[m];
            for (int32_t j = 0; j < f->mel_cnt[m]; j++) e = fmaf(w[j], P[f->mel_k0[m] + j], e);
        } else {
            e = 0.25f * P[m];
            if (f->cfg.spec_type == 3) e = sqrtf(e);
        }
        e = e * f->emph[m];
        e = e * f->gain;
        out[m] = to_u32(e);
    }
    free(P);
}

int32_t wsa_or_fe_run(const wsa_or_fe *f, const float *pcm, int64_t n_samples, uint32_t *out) {
    int32_t nf = wsa_or_fe_n_frames(f, n_samples);
    for (int32_t k = 0; k < nf; k++) wsa_or_fe_frame(f, pcm + (size_t)k * (size_t)f->hop, out + (size_t)k * (size_t)f->bands);
    return nf;
}
