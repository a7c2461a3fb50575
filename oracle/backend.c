/*
 * backend.c — ORACLE (test infrastructure only; see wsa_oracle.h).
 *
 * Plain-C restatement of the reference's pinned half: u32 spectrum frame -> peak scan -> voiced
 * state machine + auto noise gate -> formant track association -> segment finalize ->
 * straighten -> (syllable split) -> 53-feature vector.  Every function cites the site in
 * /root/reference/dist/main.js (two-line minified bundle; "@B<n>" = byte offset in the file,
 * SURVEY.md §0.2) that it follows.  All scalars are IEEE doubles exactly as JavaScript Numbers;
 * Float32Array stores are explicit (float) casts.  Build with -ffp-contract=off.
 *
 * Math.log10 / Math.pow are the fdlibm forms in jsmath.c (see DESIGN.md "JS-engine math").
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "wsa_oracle.h"

#define VEC(T) struct { T *p; int32_t n, cap; }
#define VPUSH(v, x) do { if ((v).n == (v).cap) { (v).cap = (v).cap ? 2 * (v).cap : 16; \
    (v).p = realloc((v).p, sizeof(*(v).p) * (size_t)(v).cap); } (v).p[(v).n++] = (x); } while (0)
#define VFREE(v) do { free((v).p); (v).p = NULL; (v).n = (v).cap = 0; } while (0)

/* parseInt(x) for the positive finite doubles that occur here (>= 1e-6, < 1e21): truncation. */
static double js_trunc(double x) { return trunc(x); }

/* one formant track: the 18-field record of accumulate_fm (@B35952; field map SURVEY.md App. A) */
typedef struct {
    double start, end, last_frame, vel, last_bin, last_amp;   /* [0] [1] [2]=[3] [4] [5] [6] */
    VEC(double) frames, starts, ends, bins, amps, energies;   /* [7] .. [12] */
    double sumE, count, sumEbin, sumW;                         /* [13] [14] [15] [17] */
} track_t;

typedef struct { int32_t i, s, l; } peak_t;

typedef struct {
    int32_t start, len, syl0, nsyl, has_feat;
    double feat[53];
    float *fr;              /* len x 9, level >= 4 */
    float *sm;              /* len x 3 (sum f*E, sum E, sum w*E), level >= 4 */
    double *trk; int32_t trk_n;   /* level 3: the ranked tracks the reference stores (`s.push(i)` @B28273), flattened:
                                   * n_tracks, then per track 10 scalars ([0] [1] [2] [4] [5] [6] [13] [14] [15] [17]) and
                                   * its six per-point arrays ([7] .. [12], `count` numbers each) */
} segment_t;

typedef struct { int32_t seg, start, len; double feat[53]; } syllable_t;

struct wsa_or_seg {
    wsa_or_cfg cfg;
    /* segmenter state `o` (@B23439) */
    int32_t bands, max_voiced_bin;
    double breaker, min_frames, step_s;
    double cur_frame, no_fm, c_ci;
    int32_t c_started;
    /* noise gate state y, v, x, _, w, T, k (@B24629) */
    double ctx_max, floor_, last_max, last_floor, w, T, k;
    /* tracker module state l, s, c (@B31818) */
    VEC(track_t) tracks;
    double accS, accC;
    /* outputs */
    VEC(segment_t) segs;
    VEC(syllable_t) syls;
    int32_t trace_on;
    VEC(double) trace;
};

static void track_free(track_t *t) {
    VFREE(t->frames); VFREE(t->starts); VFREE(t->ends); VFREE(t->bins); VFREE(t->amps); VFREE(t->energies);
}

/* clear_fm @B35919 */
static void clear_fm(wsa_or_seg *s) {
    for (int32_t i = 0; i < s->tracks.n; i++) track_free(&s->tracks.p[i]);
    s->tracks.n = 0; s->accS = 0; s->accC = 0;
}

/* L(e) reset_segment @B25649 */
static void reset_segment(wsa_or_seg *s, int32_t started) {
    s->c_ci = 0; s->c_started = started; s->no_fm = 0; clear_fm(s);
}

wsa_or_seg *wsa_or_seg_new(const wsa_or_cfg *cfg) {
    wsa_or_seg *s = calloc(1, sizeof(*s));
    s->cfg = *cfg;
    s->bands = cfg->bands;
    s->max_voiced_bin = (int32_t)js_trunc(0.7 * cfg->bands);                 /* @B25136 */
    s->step_s = cfg->window_step / 1e3;
    s->breaker = cfg->pause_length > 2 * cfg->window_step ? cfg->pause_length / cfg->window_step
                                                         : 250 / cfg->window_step;  /* @B25188 */
    s->min_frames = js_trunc(cfg->min_seg_length / cfg->window_step);        /* @B25218 */
    s->c_started = -1;
    if (cfg->auto_noise_gate) { s->ctx_max = 50; s->floor_ = 2; }           /* @B25471 */
    else {
        s->ctx_max = wsa_or_pow(10, cfg->voiced_max_dB / 20);
        s->floor_ = wsa_or_pow(10, cfg->voiced_min_dB / 20);
    }
    s->last_max = s->ctx_max; s->last_floor = s->floor_;
    return s;
}

void wsa_or_seg_free(wsa_or_seg *s) {
    if (!s) return;
    clear_fm(s); VFREE(s->tracks);
    for (int32_t i = 0; i < s->segs.n; i++) { free(s->segs.p[i].fr); free(s->segs.p[i].sm); free(s->segs.p[i].trk); }
    VFREE(s->segs); VFREE(s->syls); VFREE(s->trace);
    free(s);
}

/* match score `_` @B37340: (gap, dist, track length, track bin, peak bin, track amp, peak amp, velocity) */
static double match_score(double gap, double dist, double n, double tbin, double pbin, double tamp,
                          double pamp, double vel) {
    double s;
    if (tamp >= pamp) s = pamp / tamp;
    else { if (!(pamp > 0)) return 0; s = tamp / pamp; }
    if (gap == 0) return s > .1 ? 300 * s / dist : 0;
    if (s < .001) return 0;
    if (s >= 1) s = 10; else if (s < .1) s = 1; else s *= 10;
    double t = 10 - fabs(pbin - tbin - vel);
    if (t < 0) return 0;
    if (t < 1) t = 1;
    double i = n;
    if (i > 10) i = 10;
    return 10 / gap * (t * t + i * s);
}

/* test access to the module-private score (fixture tests/golden/score_expected.json) */
double wsa_or_match_score(double gap, double dist, double n, double tbin, double pbin, double tamp, double pamp, double vel) {
    return match_score(gap, dist, n, tbin, pbin, tamp, pamp, vel);
}

/* accumulate_fm `x(e,t,n,r,a)` @B35952 */
static void accumulate_fm(wsa_or_seg *S, const uint32_t *e, const peak_t *pk, int32_t U, double n,
                          double energy, double floor_) {
    static const double win[4] = {3, 4, 6, 9};                                /* @B32325 */
    if (U < 1) return;
    int32_t *asg = malloc(sizeof(int32_t) * (size_t)U);
    double *best = malloc(sizeof(double) * (size_t)U);
    for (int32_t o = 0; o < U; o++) { asg[o] = -1; best[o] = 0; }
    S->accS += energy;
    int32_t ntr = S->tracks.n;
    for (int32_t r = 0; r < ntr; r++) {
        track_t *tr = &S->tracks.p[r];
        double gap = n - tr->last_frame;
        if (gap >= 0 && gap < 4) {
            double len = tr->frames.n;
            for (int32_t o = 0; o < U; o++) {
                double dist = fabs(tr->last_bin - pk[o].l);
                if (dist < win[(int32_t)gap]) {
                    double sc = match_score(gap, dist, len, tr->last_bin, pk[o].l, tr->last_amp,
                                            (double)e[pk[o].l], tr->vel);
                    if (sc > 1 && sc > best[o]) { best[o] = sc; asg[o] = r; }
                }
            }
        }
    }
    for (int32_t r = 0; r < ntr; r++) {
        int32_t first = -1;
        for (int32_t o = 0; o < U; o++) if (asg[o] == r) { first = o; break; }
        if (first < 0) continue;
        track_t *tr = &S->tracks.p[r];
        int32_t pb = pk[first].l;
        double amp = e[pb];                              /* read before the arg-max loop (quirk 3) */
        if (amp > floor_) {
            int32_t st = pk[first].i, en = pk[first].s;
            for (int32_t o = first; o < U; o++) if (asg[o] == r) {
                if (pk[o].s > en) en = pk[o].s;
                if (pk[o].i < st) st = pk[o].i;
                if (e[pk[o].l] > e[pb]) pb = pk[o].l;
            }
            double be = 0;
            for (int32_t t = st; t <= en; t++) be += e[t];
            int32_t h = tr->bins.n;
            const double *P = tr->bins.p;
            if (h >= 3) tr->vel = (pb - P[h - 1] + (P[h - 2] - P[h - 1]) + (P[h - 3] - P[h - 2])) / 3;
            else if (h == 2) tr->vel = (pb - P[h - 1] + (P[h - 2] - P[h - 1])) / 2;
            else if (h == 1) tr->vel = pb - P[h - 1];
            tr->start = st; tr->end = en; tr->last_frame = n; tr->last_bin = pb; tr->last_amp = amp;
            VPUSH(tr->frames, n); VPUSH(tr->starts, (double)st); VPUSH(tr->ends, (double)en);
            VPUSH(tr->bins, (double)pb); VPUSH(tr->amps, amp); VPUSH(tr->energies, be);
            tr->sumE += be; tr->count += 1; tr->sumEbin += be * pb; tr->sumW += en - st + 1;
            S->accS -= be; S->accC += be;
        }
    }
    for (int32_t o = 0; o < U; o++) if (asg[o] == -1) {
        int32_t pb = pk[o].l;
        double amp = e[pb];
        if (amp > floor_) {
            int32_t st = pk[o].i, en = pk[o].s;
            double be = 0;
            for (int32_t t = st; t <= en; t++) be += e[t];
            track_t tr; memset(&tr, 0, sizeof(tr));
            tr.start = st; tr.end = en; tr.last_frame = n; tr.vel = 0; tr.last_bin = pb; tr.last_amp = amp;
            VPUSH(tr.frames, n); VPUSH(tr.starts, (double)st); VPUSH(tr.ends, (double)en);
            VPUSH(tr.bins, (double)pb); VPUSH(tr.amps, amp); VPUSH(tr.energies, be);
            tr.sumE = be; tr.count = 1; tr.sumEbin = be * pb; tr.sumW = en - st + 1;
            VPUSH(S->tracks, tr);
        }
    }
    free(asg); free(best);
}

/* stats helpers: array_mean_NZ @B2203, only_std_NZ @B1978 / mean_std_NZ @B2089 (= src/stats.js:29-55) */
static double mean_nz(const double *v, int32_t n) {
    double t = 0, c = 0;
    for (int32_t i = 0; i < n; i++) if (v[i] > 0) { t += v[i]; c++; }
    return t / c;
}
static double std_nz(const double *v, int32_t n, double m) {
    double t = 0;
    for (int32_t i = 0; i < n; i++) { double d = v[i] - m; t += d * d; }
    return sqrt(t / n);
}
static double arr_sum(const double *v, int32_t n) {
    double t = 0;
    for (int32_t i = 0; i < n; i++) t += v[i];
    return t;
}

/* formant_features `u(e,t,n)` @B32369; output order @B33436 */
void wsa_or_formant_features(const float *fr, int32_t a, double ctx_max, double floor_, double cs,
                             double x[53]) {
    for (int32_t i = 0; i < 53; i++) x[i] = 0;
    double *c = malloc(sizeof(double) * 6 * (size_t)(a > 0 ? a : 1));
    double *w = c + a, *M = w + a, *T = M + a, *K = T + a, *A = K + a;
    for (int32_t n = 0; n < 3; n++) {
        int32_t b = 5 + 16 * n;
        int32_t prev = 0, m = 0, nA = 0;
        double S = 0, L = 0, cnt = 0, runs = 0, up = 0, dn = 0;
        for (int32_t t = 0; t < a; t++) {
            double r = fr[9 * t + 3 * n], E = fr[9 * t + 3 * n + 1];
            if (r > 0 && E > 0) {
                double wd = fr[9 * t + 3 * n + 2], dB = 20 * wsa_or_log10(E);
                c[m] = r * dB; w[m] = r; M[m] = wd * dB; T[m] = E; K[m] = dB; m++;
                if (prev) {
                    double dl = r - fr[9 * (t - 1) + 3 * n];
                    if (dl > 1) up += dl; else if (dl < -1) dn += -1 * dl;
                    if (E > L) { L = E; S = 1; }
                    else if (S == 1 && E < L / 2) { if (L > 10) A[nA++] = dB; L = 0; S = -1; }
                }
                if (!prev) runs += 1;
                prev = 1; cnt += 1;
            } else { prev = 0; S = 0; L = 0; }
        }
        if (runs > 0) {
            double sT = arr_sum(T, m);
            x[b + 4] = sT / a * 100 / ctx_max;
            x[b + 5] = sT / cnt * 100 / ctx_max;
            double sK = arr_sum(K, m);
            x[b + 0] = arr_sum(c, m) / sK;
            x[b + 1] = std_nz(w, m, mean_nz(w, m));
            x[b + 6] = arr_sum(M, m) / sK;
            double mk = mean_nz(K, m);
            x[b + 2] = mk; x[b + 3] = std_nz(K, m, mk);
            x[b + 11] = nA;
            if (nA > 0) {
                double ma = mean_nz(A, nA);
                x[b + 12] = ma; x[b + 13] = std_nz(A, nA, ma);
                x[b + 14] = 100 * (x[b + 12] / (sK / m) - 1);
            }
        }
        x[b + 7] = cnt; x[b + 8] = runs; x[b + 9] = up; x[b + 10] = dn;
        x[b + 15] = 100 * cnt / a;
    }
    x[0] = a; x[1] = sqrt((double)a); x[2] = cs; x[3] = wsa_or_log10(ctx_max); x[4] = floor_;
    free(c);
}

/* O(e) finalize @B27088 with get_ranked_formants @B35670, straighten_formants @B35074,
 * sep_syllables @B34757, make_syl_features @B34407 */
static void finalize(wsa_or_seg *S, double e) {
    double len_d = e - S->no_fm;
    if (!(len_d > S->min_frames && S->c_started >= 2)) return;
    int32_t len = (int32_t)len_d;
    int32_t start = (int32_t)(S->cur_frame - len_d);
    int32_t level = S->cfg.level;
    /* ranked(): tracks with count>=2 and mean bin >= 7, stable ascending by mean bin */
    int32_t ntr = S->tracks.n, nr = 0;
    int32_t *rk = malloc(sizeof(int32_t) * (size_t)(ntr > 0 ? ntr : 1));
    for (int32_t t = 0; t < ntr; t++) {
        track_t *tr = &S->tracks.p[t];
        if (tr->count >= 2) {
            double mb = tr->sumEbin / tr->sumE;
            if (mb >= 7) {
                int32_t r = 0;
                while (r < nr) {
                    track_t *q = &S->tracks.p[rk[r]];
                    if (q->sumEbin / q->sumE > mb) break;
                    r++;
                }
                memmove(rk + r + 1, rk + r, sizeof(int32_t) * (size_t)(nr - r));
                rk[r] = t; nr++;
            }
        }
    }
    segment_t seg; memset(&seg, 0, sizeof(seg));
    seg.start = start; seg.len = len; seg.syl0 = S->syls.n;
    if (level == 3) {            /* ref @B28273: `u.push([e,a]), ..., s.push(i)` with i = get_ranked_formants() */
        size_t words = 1;
        for (int32_t t = 0; t < nr; t++) words += 10 + 6 * (size_t)S->tracks.p[rk[t]].frames.n;
        double *o = malloc(sizeof(double) * words); size_t w = 0;
        o[w++] = nr;
        for (int32_t t = 0; t < nr; t++) {
            const track_t *tr = &S->tracks.p[rk[t]];
            o[w++] = tr->start; o[w++] = tr->end; o[w++] = tr->last_frame; o[w++] = tr->vel; o[w++] = tr->last_bin; o[w++] = tr->last_amp;
            o[w++] = tr->sumE; o[w++] = tr->count; o[w++] = tr->sumEbin; o[w++] = tr->sumW;
            const int32_t c = tr->frames.n;
            for (int32_t q = 0; q < c; q++) o[w++] = tr->frames.p[q];
            for (int32_t q = 0; q < c; q++) o[w++] = tr->starts.p[q];
            for (int32_t q = 0; q < c; q++) o[w++] = tr->ends.p[q];
            for (int32_t q = 0; q < c; q++) o[w++] = tr->bins.p[q];
            for (int32_t q = 0; q < c; q++) o[w++] = tr->amps.p[q];
            for (int32_t q = 0; q < c; q++) o[w++] = tr->energies.p[q];
        }
        seg.trk = o; seg.trk_n = (int32_t)words;
        VPUSH(S->segs, seg); free(rk); return;
    }
    /* straighten(): the reference pushes segments_ci BEFORE straighten can throw (@B27240); a frame
     * index >= len would be a TypeError there -> segment kept in segments_ci without results.
     * Provably unreachable; flagged through has_feat = -1 if it ever happens. */
    float *fr = calloc((size_t)len * 9, sizeof(float));
    float *sm = calloc((size_t)len * 3, sizeof(float));
    double last = 0; int32_t slot = 0, bad = 0;
    for (int32_t t = 0; t < nr && !bad; t++) {
        track_t *tr = &S->tracks.p[rk[t]];
        double mb = tr->sumEbin / tr->sumE;
        if (fabs(mb - last) > 20) { last = mb; slot++; if (slot >= 3) break; }
        for (int32_t i = 0; i < (int32_t)tr->count; i++) {
            int32_t l = slot;
            double f = tr->bins.p[i];
            if (f > 0) {
                double E = tr->energies.p[i];
                int32_t d = (int32_t)tr->frames.p[i];
                double wd = tr->ends.p[i] - tr->starts.p[i] + 1;
                if (d >= len || d < 0) { bad = 1; break; }
                if (fr[9 * d + 3 * l] > S->floor_ && fr[9 * d + 3 * l] < f && l < 2) l++;
                fr[9 * d + 3 * l] = (float)f; fr[9 * d + 3 * l + 1] = (float)E; fr[9 * d + 3 * l + 2] = (float)wd;
                sm[3 * d + 0] = (float)((double)sm[3 * d + 0] + f * E);
                sm[3 * d + 1] = (float)((double)sm[3 * d + 1] + E);
                sm[3 * d + 2] = (float)((double)sm[3 * d + 2] + wd * E);
            }
        }
    }
    free(rk);
    if (bad) { seg.has_feat = -1; free(fr); free(sm); VPUSH(S->segs, seg); return; }
    double cs = S->accC / S->accS;
    if (level == 5) {
        wsa_or_formant_features(fr, len, S->ctx_max, S->floor_, cs, seg.feat);
        seg.has_feat = 1;
    }
    if (level == 10 || level == 13) {
        int32_t i = -1; double c = 0, u = 0;
        for (int32_t e2 = 0; e2 < len; e2++) {
            if (sm[3 * e2 + 1] > S->floor_) { c = 0; u++; if (i < 0) i = e2; } else c++;
            if ((u > 20 && c > 0) || (u > 10 && c > 1) || (u > 0 && c > 4) || (e2 >= len - 1 && u > 4)) {
                int32_t t = e2 - (int32_t)c;
                if (t - i > 1) {
                    syllable_t sy; memset(&sy, 0, sizeof(sy));
                    sy.seg = S->segs.n; sy.start = i; sy.len = t - i;
                    if (level == 13)
                        wsa_or_formant_features(fr + 9 * i, t - i, S->ctx_max, S->floor_, cs, sy.feat);
                    VPUSH(S->syls, sy);
                    i = -1; u = 0;
                }
            }
        }
        seg.nsyl = S->syls.n - seg.syl0;
    }
    seg.fr = fr; seg.sm = sm;
    VPUSH(S->segs, seg);
}

/* auto noise gate C(h) @B28506 */
static void noise_gate(wsa_or_seg *S, double h) {
    S->w++;
    if (h > S->ctx_max || (S->w > 40 && h > 2 * S->floor_)) {
        if (h >= S->ctx_max) { S->w = 0; S->last_max = S->ctx_max = h; }
        else if (h > S->last_max / 100) { S->ctx_max -= js_trunc(S->ctx_max / 8); S->w = 35; }
        double y = S->ctx_max, t = wsa_or_log10(y), v;
        if (t > 7) v = js_trunc(wsa_or_pow(10, t - 3) / 20);
        else if (t > 6) v = js_trunc(wsa_or_pow(10, t - 3) / 2);
        else if (t > 4) v = js_trunc(wsa_or_pow(10, t - 2) / 2);
        else if (t > 2) v = js_trunc(wsa_or_pow(10, t / 3));
        else if (t > 1) v = js_trunc(y / 10);
        else v = 1;
        S->floor_ = v; S->last_floor = v;
        if (S->k > 0 && S->T / S->k < 30 * v) { reset_segment(S, 0); S->k = 0; S->T = 0; }
        S->T += S->ctx_max; S->k += 1;
    } else if (S->floor_ > 10 && S->floor_ > S->last_floor / 10 && S->w > 20) {
        S->floor_ -= js_trunc(S->last_floor / 20);
        if (S->floor_ < 10) S->floor_ = 10;
    }
}

/* spectrum_push I(e,t) @B30392 (levels > 2) followed by one pass of the frame loop D() @B25717 */
void wsa_or_seg_push(wsa_or_seg *S, const uint32_t *e) {
    const int32_t B = S->bands;
    S->cur_frame++;
    double t = S->c_ci;                           /* captured before the start test (quirk 1) */
    double v = S->floor_;
    int32_t n = 0, i = 0, l = 0, s = 0, c = 0, u = 0, p = 0;
    double d = 0, h = 2 * v, g = 0;
    peak_t *pk = malloc(sizeof(peak_t) * (size_t)(B > 0 ? B : 1));
#define EMIT(upd) do { if ((upd) && e[l] > h) { h = e[l]; p = l; } \
        double thr = e[l] / 10.0; \
        while (i < l && e[i] < thr) i++; \
        while (s > l && e[s] < thr) s--; \
        pk[n].i = i; pk[n].s = s; pk[n].l = l; n++; d += e[l]; } while (0)
    for (int32_t a = 1; a < B; a++) {                                          /* @B25827 */
        g += e[a];
        if (e[a] > e[a - 1] && (a < 2 || e[a] > e[a - 2]) && (a < 3 || e[a] > e[a - 3])) {
            if (u == -1 || u == 0) {
                if (u == -1 && e[l] > v && i <= l && l < s) EMIT(1);
                i = a - 1; l = a;
            } else if (u == 1) l = a;
            u = 1;
        } else if (e[a] < e[a - 1] && (a < 2 || e[a] < e[a - 2]) && (a < 3 || e[a] < e[a - 3])) {
            if (u == 1 || u == -1) { s = a; u = -1; }
        } else if (u == -1) {
            c++;
            if (c > 2) { c = 0; if (e[l] > v && i <= l && l < s) EMIT(1); u = 0; }
        } else if (u == 1 && e[a] > e[a - 1]) l = a;
        if (a == B - 1 && u == 1) { s = a; l = a; if (e[l] > v && i < l && l <= s) EMIT(0); }
    }
#undef EMIT
    if (S->c_started < 0) {                                                    /* @B26527 */
        double r = d > h ? h * (n - 1) / (d - h) : 0;
        if (n > 0 && p > 7 && p < S->max_voiced_bin && n > 4 && r > 4) reset_segment(S, 0);
        else S->no_fm++;
    }
    int32_t do_reset = 0;
    if (S->c_started >= 0) {                                                   /* @B26646 */
        if (n == 0 || p < 7 || p >= S->max_voiced_bin || (n > 3 && d / (g - d) < .1)) {
            S->no_fm++;
            if (S->c_started < 2) S->c_started--;
            else if (S->no_fm >= S->breaker) { finalize(S, S->c_ci + 1); do_reset = 1; }
            else if (S->cfg.auto_noise_gate) noise_gate(S, h);
        } else {
            if (S->cfg.auto_noise_gate) noise_gate(S, h);
            accumulate_fm(S, e, pk, n, t, g, S->floor_);
            if (S->c_started < 2) S->c_started++; else S->no_fm = 0;
        }
    }
    if (S->trace_on) {
        double row[10] = {S->c_ci, (double)S->c_started, S->no_fm, S->ctx_max, S->floor_, (double)n,
                          (double)p, h, d, g};
        for (int k = 0; k < 10; k++) VPUSH(S->trace, row[k]);
    }
    S->c_ci++;
    /* the reference resets in the Promise .then microtask, i.e. after the frame completes (quirk 8) */
    if (do_reset) reset_segment(S, -1);
    free(pk);
}

/* segment_truncate N() @B30757 -> D() with play_end -> O(c_ci) -> L(1) */
void wsa_or_seg_finish(wsa_or_seg *S) {
    finalize(S, S->c_ci);
    reset_segment(S, 1);
}

int32_t wsa_or_n_segments(const wsa_or_seg *s) { return s->segs.n; }
void wsa_or_segment(const wsa_or_seg *s, int32_t i, int32_t out[5]) {
    const segment_t *g = &s->segs.p[i];
    out[0] = g->start; out[1] = g->len; out[2] = g->syl0; out[3] = g->nsyl; out[4] = g->has_feat;
}
const double *wsa_or_segment_features(const wsa_or_seg *s, int32_t i) { return s->segs.p[i].feat; }
const float *wsa_or_segment_formants(const wsa_or_seg *s, int32_t i) { return s->segs.p[i].fr; }
const float *wsa_or_segment_sums(const wsa_or_seg *s, int32_t i) { return s->segs.p[i].sm; }
const double *wsa_or_segment_tracks(const wsa_or_seg *s, int32_t i, int32_t *n) { *n = s->segs.p[i].trk_n; return s->segs.p[i].trk; }
int32_t wsa_or_n_syllables(const wsa_or_seg *s) { return s->syls.n; }
void wsa_or_syllable(const wsa_or_seg *s, int32_t j, int32_t out[3]) {
    const syllable_t *y = &s->syls.p[j];
    out[0] = y->seg; out[1] = y->start; out[2] = y->len;
}
const double *wsa_or_syllable_features(const wsa_or_seg *s, int32_t j) { return s->syls.p[j].feat; }
int32_t wsa_or_trace_len(const wsa_or_seg *s) { return s->trace.n / 10; }
const double *wsa_or_trace(const wsa_or_seg *s) { return s->trace.p; }
void wsa_or_enable_trace(wsa_or_seg *s, int32_t on) { s->trace_on = on; }

wsa_or_seg *wsa_or_run_clip(const wsa_or_cfg *cfg, const uint32_t *spectra, int32_t frames) {
    wsa_or_seg *s = wsa_or_seg_new(cfg);
    for (int32_t f = 0; f < frames; f++) wsa_or_seg_push(s, spectra + (size_t)f * (size_t)cfg->bands);
    wsa_or_seg_finish(s);
    return s;
}
