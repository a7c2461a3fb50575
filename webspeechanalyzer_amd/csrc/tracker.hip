// tracker.hip — K2: the sequential half of the hot path, ONE WAVEFRONT PER CLIP.
//
// Stands in for (ref = /root/reference/dist/main.js line 2, byte offsets):
//   frame loop D() after the peak scan            @B25717 (start test @B26527, voiced test @B26646)
//   auto noise gate C(h)                          @B28506
//   accumulate_fm x(e,t,n,r,a) + match score _    @B35952, @B37340
//   finalize O(e)                                 @B27088
//   get_ranked_formants y(), straighten m()       @B35670, @B35074
//   sep_syllables p(), formant_features u()       @B34757, @B32369 (+ stats helpers @B1978-2277)
//   reset_segment L(e), clear_fm                  @B25649, @B35919
// including the reference's quirks: stale first-frame index, finalize-then-reset
// ordering, first-peak amplitude of a merged association, fp32 storage in straighten, and the
// segments_ci entry that survives a throwing straighten step.
//
// All decision arithmetic is IEEE double exactly as JavaScript Numbers (translation unit compiled
// with -ffp-contract=off; Math.log10 / Math.pow from jsmath_device.hpp).  State that the reference
// keeps in module variables is wave-uniform register state here; lanes parallelise the inner loops:
// candidate gating (lane = peak), live-track compaction (lane = track), peak<->track scoring
// (lane = peak, loop over live tracks in LDS), new-track creation (lane = peak), ranking
// (lane = track), straighten (lane = frame), features (lane = formant).
#include "wsa_internal.hpp"
#include "jsmath_device.hpp"

namespace wsa {

constexpr int MAXC = 64;            // peak candidates per frame record (bands <= 128)
constexpr int AC = 320;             // active-track table: tracks not yet 4 filing indices old (<= 5 x 63)

struct Ws {                          // per-wave work space carved out of global memory
    int32_t *tr_len, *tr_slot, *tr_rank;           // per track id: write-through summary + finalize scratch
    double *tr_sumE, *tr_sumEbin;
    int32_t *pt_track, *pt_bw; double* pt_energy;
    int32_t *d_p0, *d_p1, *d_gen;
    float *fr, *sm1;
    double *dB, *Aev;
    int32_t *q_idx, *sorted; double* q_mb;
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// lays the per-wave arrays out back to back (16-byte aligned); returns the pointers by value so
// that they live in registers, and the total size through *bytes
__host__ __device__ __forceinline__ Ws carve_ws(char* base, int T, int P, int F, size_t* bytes) {
    Ws w;
    size_t o = 0;
#define WSA_CARVE(field, type, count) do { w.field = reinterpret_cast<type*>(base + o); \
        o = align16(o + sizeof(type) * (size_t)(count)); } while (0)
    WSA_CARVE(tr_len, int32_t, T); WSA_CARVE(tr_slot, int32_t, T); WSA_CARVE(tr_rank, int32_t, T);
    WSA_CARVE(tr_sumE, double, T); WSA_CARVE(tr_sumEbin, double, T);
    WSA_CARVE(pt_track, int32_t, P); WSA_CARVE(pt_bw, int32_t, P); WSA_CARVE(pt_energy, double, P);
    WSA_CARVE(d_p0, int32_t, F + 2); WSA_CARVE(d_p1, int32_t, F + 2); WSA_CARVE(d_gen, int32_t, F + 2);
    WSA_CARVE(fr, float, (size_t)(F + 2) * 9); WSA_CARVE(sm1, float, F + 2);
    WSA_CARVE(dB, double, (size_t)3 * (F + 2)); WSA_CARVE(Aev, double, (size_t)3 * (F + 2));
    WSA_CARVE(q_idx, int32_t, T); WSA_CARVE(sorted, int32_t, T); WSA_CARVE(q_mb, double, T);
#undef WSA_CARVE
    if (bytes) *bytes = o;
    return w;
}

size_t tracker_ws_bytes(int tcap, int pcap, int fcap) { size_t b = 0; (void)carve_ws(nullptr, tcap, pcap, fcap, &b); return align16(b) + 256; }

// Lanes of ONE wave exchange data through LDS / global memory here.  The hardware executes a wave's
// LDS (and vector-memory) instructions in issue order, so no s_waitcnt is needed — only the compiler
// must not move memory accesses across the exchange point.  (A wavefront-scope fence would also do,
// but it drains vmcnt and so kills the frame-record prefetch.)
__device__ __forceinline__ void wsync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

__device__ __forceinline__ double wave_sum_f64(double v) {          // exact for the integer-valued sums here
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_incl_scan_i32(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64); v = t > v ? t : v; }
    return v;
}

// match score `_` (ref @B37340)
__device__ __forceinline__ double match_score(int gap, double dist, double n, double tbin, double pbin,
                                              double tamp, double pamp, double vel) {
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
    return 10 / (double)gap * (t * t + i * s);
}

// formant_features (ref @B32369) for ONE formant column n over rows [0, a) of fr (row stride 9).
// Writes x[5+16n .. 5+16n+15].  dBs / Aev are per-formant scratch rows of >= a doubles.
__device__ void formant_column(const float* fr, int a, int n, double ctx_max, double* x, double* dBs, double* Aev) {
    const int b = 5 + 16 * n;
    bool prev = false;
    int m = 0, nA = 0;
    double S = 0, L = 0, cnt = 0, runs = 0, up = 0, dn = 0;
    double sc = 0, sw = 0, sM = 0, sT = 0, sK = 0, sKpos = 0, nKpos = 0;
    float rprev = 0.f;
    for (int t = 0; t < a; t++) {
        const float rf = fr[9 * t + 3 * n], Ef = fr[9 * t + 3 * n + 1];
        const double r = rf, E = Ef;
        if (r > 0 && E > 0) {
            const double wd = fr[9 * t + 3 * n + 2], dB = 20 * jsm::log10(E);
            sc += r * dB; sw += r; sM += wd * dB; sT += E; sK += dB;
            if (dB > 0) { sKpos += dB; nKpos += 1; }
            dBs[m] = dB; m++;
            if (prev) {
                const double dl = r - (double)rprev;
                if (dl > 1) up += dl; else if (dl < -1) dn += -1 * dl;
                if (E > L) { L = E; S = 1; }
                else if (S == 1 && E < L / 2) { if (L > 10) Aev[nA++] = dB; L = 0; S = -1; }
            }
            if (!prev) runs += 1;
            prev = true; cnt += 1;
        } else { prev = false; S = 0; L = 0; }
        rprev = rf;
    }
    for (int q = 0; q < 16; q++) x[b + q] = 0;
    if (runs > 0) {
        x[b + 4] = sT / a * 100 / ctx_max;
        x[b + 5] = sT / cnt * 100 / ctx_max;
        x[b + 0] = sc / sK;
        const double mw = sw / cnt;                                  // mean_nz(w): every w entry is > 0
        double vw = 0;
        for (int t = 0; t < a; t++) {
            const double r = fr[9 * t + 3 * n], E = fr[9 * t + 3 * n + 1];
            if (r > 0 && E > 0) { const double d = r - mw; vw += d * d; }
        }
        x[b + 1] = sqrt(vw / m);
        x[b + 6] = sM / sK;
        const double mk = sKpos / nKpos;
        double vk = 0;
        for (int q = 0; q < m; q++) { const double d = dBs[q] - mk; vk += d * d; }
        x[b + 2] = mk; x[b + 3] = sqrt(vk / m);
        x[b + 11] = nA;
        if (nA > 0) {
            double sa = 0, na = 0;
            for (int q = 0; q < nA; q++) if (Aev[q] > 0) { sa += Aev[q]; na += 1; }
            const double ma = sa / na;
            double va = 0;
            for (int q = 0; q < nA; q++) { const double d = Aev[q] - ma; va += d * d; }
            x[b + 12] = ma; x[b + 13] = sqrt(va / nA);
            x[b + 14] = 100 * (ma / (sK / m) - 1);
        }
    }
    x[b + 7] = cnt; x[b + 8] = runs; x[b + 9] = up; x[b + 10] = dn;
    x[b + 15] = 100 * cnt / a;
}

__global__ __launch_bounds__(64) void tracker_kernel(TrParams p) {
    // accepted peaks of the current frame, compacted (lane o <-> peak o)
    __shared__ uint32_t s_pk[MAXC], s_amp[MAXC];
    __shared__ double s_plo[MAXC], s_phi[MAXC];
    // active tracks (ref `l`, the live part), in track order
    __shared__ int32_t a_last_frame[AC], a_len[AC], a_gid[AC];
    __shared__ uint32_t a_bins[AC], a_amp[AC];              // a_bins = last bin | P[h-2] << 8 | P[h-3] << 16
    __shared__ double a_vel[AC], a_sumE[AC], a_sumEbin[AC];
    __shared__ unsigned long long a_mmask[AC];              // peaks assigned to the track this frame
    // (track, peak) pairs of one scoring pass and the per-peak arg-max scratch
    __shared__ int32_t s_pr_j[64], s_pr_o[64];
    __shared__ unsigned long long s_best[MAXC];
    __shared__ int32_t s_asg[MAXC];

    const int lane = threadIdx.x;
    const int RS = p.rec_stride;
    const Ws W = carve_ws(p.ws + (uint64_t)blockIdx.x * p.ws_stride, p.tcap, p.pcap, p.fcap, nullptr);
    int gen = 0;
    for (int d = lane; d < p.fcap + 2; d += 64) W.d_gen[d] = 0;
    wsync();

    for (uint32_t clip = blockIdx.x; clip < p.n_clips; clip += gridDim.x) {
        const uint32_t nfr = p.n_frames[clip];
        const uint32_t* rec = p.rec + (uint64_t)p.frame_off[clip] * (uint32_t)RS;
        int32_t* seg_out = p.seg_out + (uint64_t)clip * p.seg_cap * 4;
        int32_t* row_meta = p.row_meta + (uint64_t)clip * p.row_cap * 8;
        double* row_feat = p.row_feat + (uint64_t)clip * p.row_cap * WSA_NFEAT;

        // ---- launch state (ref reset_segmentation @B24629)
        int cur_frame = 0, no_fm = 0, c_ci = 0, c_started = -1;
        double ctx_max = p.ctx_max0, floor_ = p.floor0, last_max = p.ctx_max0, last_floor = p.floor0;
        double gw = 0, gT = 0, gk = 0;               // gate counters w, T, k
        double accS = 0, accC = 0;
        int n_tr = 0, n_pt = 0, n_act = 0, stale_d = -1, stale_p1 = 0;
        int nseg = 0, nres = 0, nrows = 0;
        bool overflow = false;
        gen++;

#define WSA_RESET_SEGMENT(x) do { c_ci = 0; c_started = (x); no_fm = 0; n_tr = 0; n_pt = 0; n_act = 0; \
            accS = 0; accC = 0; stale_d = -1; stale_p1 = 0; gen++; } while (0)

        // finalize O(e) (ref @B27088) as a lambda over the wave-uniform state
        auto finalize = [&](int e_arg) __attribute__((always_inline)) {
            const int len = e_arg - no_fm;
            if (!((double)len > p.min_frames && c_started >= 2)) return;
            const int start = cur_frame - len;
            if (nseg >= p.seg_cap) { overflow = true; return; }
            int32_t* sg = seg_out + 4 * nseg;
            const int my_seg = nseg;
            nseg++;
            if (p.level == 3) { if (lane == 0) { sg[0] = start; sg[1] = len; sg[2] = 1; sg[3] = 0; } nres++; return; }
            // ---- get_ranked_formants (ref @B35670): count >= 2 and mean bin >= 7, stable ascending
            int nq = 0;
            for (int base = 0; base < n_tr; base += 64) {
                const int t = base + lane;
                bool q = false; double mb = 0;
                if (t < n_tr) {
                    W.tr_slot[t] = -1;
                    if (W.tr_len[t] >= 2) { mb = W.tr_sumEbin[t] / W.tr_sumE[t]; q = mb >= 7; }
                }
                const uint64_t mask = __ballot(q);
                if (q) { const int pos = nq + __popcll(mask & lanemask_lt(lane)); W.q_idx[pos] = t; W.q_mb[pos] = mb; }
                nq += __popcll(mask);
            }
            wsync();
            for (int base = 0; base < nq; base += 64) {
                const int qi = base + lane;
                if (qi < nq) {
                    const double mb = W.q_mb[qi];
                    int rank = 0;
                    for (int u = 0; u < nq; u++) { const double o = W.q_mb[u]; rank += (o < mb || (o == mb && u < qi)) ? 1 : 0; }
                    W.sorted[rank] = qi;
                }
            }
            wsync();
            // ---- slot assignment of straighten_formants (ref @B35074, first loop header)
            if (lane == 0) {
                double last = 0; int slot = 0;
                for (int r = 0; r < nq; r++) {
                    const int qi = W.sorted[r];
                    const double mb = W.q_mb[qi];
                    if (fabs(mb - last) > 20) { last = mb; slot++; if (slot >= 3) break; }
                    const int t = W.q_idx[qi];
                    W.tr_slot[t] = slot; W.tr_rank[t] = r;
                }
            }
            wsync();
            // ---- a point of a processed track filed at an index >= len makes the reference throw
            //      (r[d] undefined, ref @B35484): segments_ci keeps the entry, nothing else is stored
            bool bad = false;
            for (int base = len; base <= c_ci + 1; base += 64) {
                const int d = base + lane;
                if (d <= c_ci + 1 && W.d_gen[d] == gen)
                    for (int q = W.d_p0[d]; q < W.d_p1[d]; q++) if (W.tr_slot[W.pt_track[q]] >= 0) bad = true;
            }
            if (stale_d >= len && lane == 0)
                for (int q = 0; q < stale_p1; q++) if (W.tr_slot[W.pt_track[q]] >= 0) bad = true;
            if (__ballot(bad) != 0ull) { if (lane == 0) { sg[0] = start; sg[1] = len; sg[2] = -1; sg[3] = 0; } return; }
            // ---- straighten body, lane = frame index d: apply this frame's points in
            //      (track rank, arrival) order
            for (int base = 0; base < len; base += 64) {
                const int d = base + lane;
                if (d < len) {
                    float f9[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) f9[q] = 0.f;
                    float sm = 0.f;
                    const int a0 = 0, a1 = (stale_d == d) ? stale_p1 : 0;
                    const bool has_main = W.d_gen[d] == gen;
                    const int b0 = has_main ? W.d_p0[d] : 0, b1 = has_main ? W.d_p1[d] : 0;
                    long long last_key = -1;
                    for (;;) {
                        long long best_key = 0x7fffffffffffffffLL; int best_q = -1;
                        for (int part = 0; part < 2; part++) {
                            const int q0 = part ? b0 : a0, q1 = part ? b1 : a1;
                            for (int q = q0; q < q1; q++) {
                                const int t = W.pt_track[q];
                                if (W.tr_slot[t] < 0) continue;
                                const long long key = (long long)W.tr_rank[t] * (long long)(p.pcap + 1) + q;
                                if (key > last_key && key < best_key) { best_key = key; best_q = q; }
                            }
                        }
                        if (best_q < 0) break;
                        last_key = best_key;
                        const int t = W.pt_track[best_q];
                        int l = W.tr_slot[t];
                        const int bw = W.pt_bw[best_q];
                        const double f = bw & 0xff, wd = bw >> 8, E = W.pt_energy[best_q];
                        const float cur = l == 0 ? f9[0] : (l == 1 ? f9[3] : f9[6]);
                        if ((double)cur > floor_ && (double)cur < f && l < 2) l++;
                        const float ff = (float)f, Ef = (float)E, wf = (float)wd;
                        if (l == 0) { f9[0] = ff; f9[1] = Ef; f9[2] = wf; }
                        else if (l == 1) { f9[3] = ff; f9[4] = Ef; f9[5] = wf; }
                        else { f9[6] = ff; f9[7] = Ef; f9[8] = wf; }
                        sm = (float)((double)sm + E);
                    }
#pragma unroll
                    for (int q = 0; q < 9; q++) W.fr[9 * d + q] = f9[q];
                    W.sm1[d] = sm;
                }
            }
            wsync();
            const double cs = accC / accS;
            const double lg_ctx = jsm::log10(ctx_max);
            if (p.level == 4 || p.level == 5) {
                if (nrows >= p.row_cap) { overflow = true; return; }
                double* x = row_feat + (uint64_t)nrows * WSA_NFEAT;
                if (p.level == 5) {
                    if (lane < 3) formant_column(W.fr, len, lane, ctx_max, x, W.dB + (size_t)lane * (p.fcap + 2), W.Aev + (size_t)lane * (p.fcap + 2));
                    if (lane == 0) { x[0] = len; x[1] = sqrt((double)len); x[2] = cs; x[3] = lg_ctx; x[4] = floor_; }
                } else if (lane < WSA_NFEAT) x[lane] = 0;
                if (lane == 0) {
                    int32_t* m = row_meta + (uint64_t)nrows * 8;
                    m[0] = (int32_t)clip; m[1] = nres; m[2] = 0; m[3] = 0; m[4] = my_seg; m[5] = 0; m[6] = start; m[7] = len;
                    sg[0] = start; sg[1] = len; sg[2] = 1; sg[3] = 1;
                }
                nrows++; nres++;
                return;
            }
            // ---- levels 10 / 13: sep_syllables (ref @B34757) then one feature row per syllable
            int si = -1, cc = 0, uu = 0, nsyl = 0;
            for (int base = 0; base < len; base += 64) {
                const int dd = base + lane;
                const float smv = dd < len ? W.sm1[dd] : 0.f;
                const int lim = min(64, len - base);
                for (int j = 0; j < lim; j++) {
                    const int e2 = base + j;
                    const double v = __shfl(smv, j, 64);
                    if (v > floor_) { cc = 0; uu++; if (si < 0) si = e2; } else cc++;
                    if ((uu > 20 && cc > 0) || (uu > 10 && cc > 1) || (uu > 0 && cc > 4) || (e2 >= len - 1 && uu > 4)) {
                        const int t = e2 - cc;
                        if (t - si > 1) {
                            if (nrows >= p.row_cap) { overflow = true; return; }
                            double* x = row_feat + (uint64_t)nrows * WSA_NFEAT;
                            const int sl = t - si;
                            if (p.level == 13) {
                                if (lane < 3) formant_column(W.fr + 9 * si, sl, lane, ctx_max, x, W.dB + (size_t)lane * (p.fcap + 2), W.Aev + (size_t)lane * (p.fcap + 2));
                                if (lane == 0) { x[0] = sl; x[1] = sqrt((double)sl); x[2] = cs; x[3] = lg_ctx; x[4] = floor_; }
                            } else if (lane < WSA_NFEAT) x[lane] = 0;
                            if (lane == 0) {
                                int32_t* m = row_meta + (uint64_t)nrows * 8;
                                m[0] = (int32_t)clip; m[1] = nres; m[2] = si; m[3] = sl; m[4] = my_seg; m[5] = nsyl; m[6] = start + si; m[7] = sl;
                            }
                            nrows++; nsyl++;
                            si = -1; uu = 0;
                        }
                    }
                }
            }
            if (lane == 0) { sg[0] = start; sg[1] = len; sg[2] = nsyl > 0 ? 1 : 0; sg[3] = nsyl; }
            nres++;
        };

        // auto noise gate C(h) (ref @B28506)
        auto noise_gate = [&](double h) __attribute__((always_inline)) {
            gw++;
            if (h > ctx_max || (gw > 40 && h > 2 * floor_)) {
                if (h >= ctx_max) { gw = 0; last_max = ctx_max = h; }
                else if (h > last_max / 100) { ctx_max -= trunc(ctx_max / 8); gw = 35; }
                const double y = ctx_max, t = jsm::log10(y);
                double v;
                if (t > 7) v = trunc(jsm::pow_pos(10, t - 3) / 20);
                else if (t > 6) v = trunc(jsm::pow_pos(10, t - 3) / 2);
                else if (t > 4) v = trunc(jsm::pow_pos(10, t - 2) / 2);
                else if (t > 2) v = trunc(jsm::pow_pos(10, t / 3));
                else if (t > 1) v = trunc(y / 10);
                else v = 1;
                floor_ = v; last_floor = v;
                if (gk > 0 && gT / gk < 30 * v) { WSA_RESET_SEGMENT(0); gk = 0; gT = 0; }
                gT += ctx_max; gk += 1;
            } else if (floor_ > 10 && floor_ > last_floor / 10 && gw > 20) {
                floor_ -= trunc(last_floor / 20);
                if (floor_ < 10) floor_ = 10;
            }
        };

        // ---- frame records are prefetched one frame (entries) / two frames (header) ahead
        double g_a = 0, g_b = 0; int n_a = 0, n_b = 0;          // headers of frames f, f+1
        uint32_t e_pk = 0, e_amp = 0; double e_plo = 0, e_phi = 0;   // this lane's entry of frame f
        auto load_hdr = [&](uint32_t f, double& g, int& n) __attribute__((always_inline)) {
            const uint32_t* r = rec + (uint64_t)f * (uint32_t)RS;
            g = *reinterpret_cast<const double*>(r); n = (int)r[2];
        };
        auto load_ent = [&](uint32_t f, int n, uint32_t& pk, uint32_t& amp, double& plo, double& phi) __attribute__((always_inline)) {
            if (lane < n) {
                const uint32_t* r = rec + (uint64_t)f * (uint32_t)RS;
                pk = r[4 + lane]; amp = r[4 + MAXC + lane];
                plo = reinterpret_cast<const double*>(r + 4 + 2 * MAXC)[lane];
                phi = reinterpret_cast<const double*>(r + 4 + 4 * MAXC)[lane];
            }
        };
        if (nfr > 0) { load_hdr(0, g_a, n_a); load_ent(0, n_a, e_pk, e_amp, e_plo, e_phi); }
        if (nfr > 1) load_hdr(1, g_b, n_b);

        for (uint32_t f = 0; f < nfr; f++) {
            const int ncand = n_a;
            const double g = g_a;
            const uint32_t pkw = e_pk, amp = e_amp; const double plo = e_plo, phi = e_phi;
            // prefetch: entries of f+1 (its header is already here), header of f+2
            uint32_t nx_pk = 0, nx_amp = 0; double nx_plo = 0, nx_phi = 0, g_c = 0; int n_c = 0;
            if (f + 1 < nfr) load_ent(f + 1, n_b, nx_pk, nx_amp, nx_plo, nx_phi);
            if (f + 2 < nfr) load_hdr(f + 2, g_c, n_c);

            cur_frame++;
            const int t_idx = c_ci;                                  // captured before the start test (quirk 1)
            const double v = floor_;
            // ---- gate the candidates: lane = candidate (ref @B25827: `e[l] > v`)
            const bool acc = lane < ncand && (double)amp > v;
            const uint64_t amask = __ballot(acc);
            const int n = __popcll(amask);
            const double d = wave_sum_f64(acc ? (double)amp : 0.0);
            const bool hp = acc && ((pkw >> 24) & 1u) == 0;        // the end-of-spectrum peak never updates h / p
            const uint32_t mx = wave_max_u32(hp ? amp : 0u);
            double h = 2 * v; int pbin = 0;
            if (n > 0 && (double)mx > h) {
                h = mx;
                const uint64_t fm = __ballot(hp && amp == mx);
                const int src = __ffsll((long long)fm) - 1;
                pbin = (int)((__shfl((int)pkw, src, 64) >> 16) & 0xff);
            }

            // ---- start test (ref @B26527)
            bool reset_this_frame = false;
            if (c_started < 0) {
                const double r = d > h ? h * (n - 1) / (d - h) : 0;
                if (n > 0 && pbin > 7 && pbin < p.max_voiced_bin && n > 4 && r > 4) { WSA_RESET_SEGMENT(0); reset_this_frame = true; }
                else no_fm++;
            }
            bool do_reset = false;
            if (c_started >= 0) {                                    // ref @B26646
                if (n == 0 || pbin < 7 || pbin >= p.max_voiced_bin || (n > 3 && d / (g - d) < .1)) {
                    no_fm++;
                    if (c_started < 2) c_started--;
                    else if ((double)no_fm >= p.breaker) { finalize(c_ci + 1); do_reset = true; }
                    else if (p.auto_gate) noise_gate(h);
                } else {
                    if (p.auto_gate) { const int g0 = gen; noise_gate(h); if (gen != g0) reset_this_frame = true; }
                    // ---- accumulate_fm(e, peaks, t_idx, g, floor_) (ref @B35952)
                    if (n >= 1) {
                        const int nfile = t_idx;
                        const double fl = floor_;
                        accS += g;
                        // compact the accepted peaks: lane o < n owns peak o
                        const int my_o = __popcll(amask & lanemask_lt(lane));
                        if (acc) { s_pk[my_o] = pkw; s_amp[my_o] = amp; s_plo[my_o] = plo; s_phi[my_o] = phi; }
                        wsync();
                        int pk_i = 0, pk_s = 0, pk_l = -1000; uint32_t pk_amp = 0; double pk_plo = 0, pk_phi = 0;
                        if (lane < n) {
                            const uint32_t w = s_pk[lane];
                            pk_i = w & 0xff; pk_s = (w >> 8) & 0xff; pk_l = (w >> 16) & 0xff;
                            pk_amp = s_amp[lane]; pk_plo = s_plo[lane]; pk_phi = s_phi[lane];
                        }
                        // 1. retire tracks whose last filing index is 4 or more behind (gap only grows)
                        {
                            int kept = 0;
                            for (int base = 0; base < n_act; base += 64) {
                                const int j = base + lane;
                                const bool valid = j < n_act;
                                int lf = 0, ln = 0, gi = 0; uint32_t bn = 0, am = 0; double ve = 0, se = 0, sb = 0;
                                if (valid) { lf = a_last_frame[j]; ln = a_len[j]; gi = a_gid[j]; bn = a_bins[j]; am = a_amp[j]; ve = a_vel[j]; se = a_sumE[j]; sb = a_sumEbin[j]; }
                                const bool keep = valid && (nfile - lf) < 4;
                                const uint64_t km = __ballot(keep);
                                wsync();
                                if (keep) {
                                    const int q = kept + __popcll(km & lanemask_lt(lane));
                                    a_last_frame[q] = lf; a_len[q] = ln; a_gid[q] = gi; a_bins[q] = bn; a_amp[q] = am; a_vel[q] = ve; a_sumE[q] = se; a_sumEbin[q] = sb;
                                }
                                kept += __popcll(km);
                                wsync();
                            }
                            n_act = kept;
                        }
                        // 2. score every (track, peak) pair inside the track's search window; per peak keep
                        //    the best score > 1, the EARLIER track on ties (ref: `i>1&&i>d[o]` in track order)
                        int asg = -1; double best = 0;
                        for (int tbase = 0; tbase < n_act; tbase += 64) {
                            const int j = tbase + lane;
                            const bool valid = j < n_act;
                            int gap = -1, bin = 0;
                            if (valid) { gap = nfile - a_last_frame[j]; bin = (int)(a_bins[j] & 0xff); a_mmask[j] = 0ull; }
                            const bool live = valid && gap >= 0 && gap < 4;
                            const int win = gap == 0 ? 3 : (gap == 1 ? 4 : (gap == 2 ? 6 : 9));      // ref @B32325
                            int o_lo = 0, o_hi = 0;
                            for (int o = 0; o < n; o++) {
                                const int lo = __builtin_amdgcn_readlane(pk_l, o);
                                o_lo += (lo <= bin - win) ? 1 : 0;
                                o_hi += (lo < bin + win) ? 1 : 0;
                            }
                            const int cnt = live ? o_hi - o_lo : 0;
                            const int incl = wave_incl_scan_i32(cnt, lane);
                            const int off = incl - cnt;
                            const int M = __builtin_amdgcn_readlane(incl, 63);
                            const int maxc = (int)wave_max_u32((uint32_t)cnt);
                            for (int base = 0; base < M; base += 64) {
                                if (lane < MAXC) { s_best[lane] = 0ull; s_asg[lane] = 0x7fffffff; }
                                for (int c = 0; c < maxc; c++) {
                                    const int slot = off + c - base;
                                    if (c < cnt && slot >= 0 && slot < 64) { s_pr_j[slot] = j; s_pr_o[slot] = o_lo + c; }
                                }
                                wsync();
                                const bool pv = base + lane < M;
                                int jj = 0, oo = 0; double sc = 0;
                                if (pv) {
                                    jj = s_pr_j[lane]; oo = s_pr_o[lane];
                                    const int tb = (int)(a_bins[jj] & 0xff), tg = nfile - a_last_frame[jj];
                                    const int pl = (int)((s_pk[oo] >> 16) & 0xff);
                                    sc = match_score(tg, (double)abs(tb - pl), (double)a_len[jj], (double)tb, (double)pl,
                                                     (double)a_amp[jj], (double)s_amp[oo], a_vel[jj]);
                                    if (sc > 1) atomicMax(&s_best[oo], (unsigned long long)__double_as_longlong(sc));
                                }
                                wsync();
                                if (pv && sc > 1 && (unsigned long long)__double_as_longlong(sc) == s_best[oo]) atomicMin(&s_asg[oo], jj);
                                wsync();
                                if (lane < n) {
                                    const int cj = s_asg[lane];
                                    if (cj != 0x7fffffff) {
                                        const double cs = __longlong_as_double((long long)s_best[lane]);
                                        if (cs > best) { best = cs; asg = cj; }
                                    }
                                }
                                wsync();
                            }
                        }
                        // 3. hand each matched track the set of its peaks
                        if (lane < n && asg >= 0) atomicOr(&a_mmask[asg], 1ull << lane);
                        wsync();
                        const int p_begin = n_pt;
                        // 4. matched tracks update themselves (lane = track), points in track order
                        for (int tbase = 0; tbase < n_act; tbase += 64) {
                            const int j = tbase + lane;
                            const unsigned long long mm = j < n_act ? a_mmask[j] : 0ull;
                            bool upd = false; int pb = 0, st = 0, en = 0; uint32_t a0 = 0; double be = 0;
                            if (mm) {
                                const int first = __ffsll((long long)mm) - 1;
                                const uint32_t w0 = s_pk[first];
                                pb = (w0 >> 16) & 0xff;
                                a0 = s_amp[first];                       // amplitude of the FIRST assigned peak (quirk 3)
                                if ((double)a0 > fl) {
                                    upd = true;
                                    st = w0 & 0xff; en = (w0 >> 8) & 0xff;
                                    double lo_sum = s_plo[first], hi_sum = s_phi[first];
                                    uint32_t pb_amp = a0;
                                    unsigned long long rest = mm;
                                    while (rest) {
                                        const int o = __ffsll((long long)rest) - 1; rest &= rest - 1;
                                        const uint32_t w = s_pk[o];
                                        const int oi = w & 0xff, os = (w >> 8) & 0xff, ol = (w >> 16) & 0xff;
                                        if (os > en) { en = os; hi_sum = s_phi[o]; }
                                        if (oi < st) { st = oi; lo_sum = s_plo[o]; }
                                        if (s_amp[o] > pb_amp) { pb = ol; pb_amp = s_amp[o]; }
                                    }
                                    be = hi_sum - lo_sum;                // sum e[st..en], exact
                                }
                            }
                            const uint64_t um = __ballot(upd);
                            const int nu = __popcll(um);
                            if (n_pt + nu > p.pcap) { overflow = true; }
                            else if (upd) {
                                const int q = n_pt + __popcll(um & lanemask_lt(lane));
                                const int hlen = a_len[j];
                                const uint32_t bn = a_bins[j];
                                const int P1 = bn & 0xff, P2 = (bn >> 8) & 0xff, P3 = (bn >> 16) & 0xff;
                                double vel = a_vel[j];
                                if (hlen >= 3) vel = (double)((pb - P1) + (P2 - P1) + (P3 - P2)) / 3;
                                else if (hlen == 2) vel = (double)((pb - P1) + (P2 - P1)) / 2;
                                else if (hlen == 1) vel = (double)(pb - P1);
                                const double se = a_sumE[j] + be, sb = a_sumEbin[j] + be * pb;
                                a_vel[j] = vel; a_bins[j] = (uint32_t)pb | ((uint32_t)P1 << 8) | ((uint32_t)P2 << 16);
                                a_amp[j] = a0; a_last_frame[j] = nfile; a_len[j] = hlen + 1; a_sumE[j] = se; a_sumEbin[j] = sb;
                                const int t = a_gid[j];
                                W.tr_len[t] = hlen + 1; W.tr_sumE[t] = se; W.tr_sumEbin[t] = sb;
                                W.pt_track[q] = t; W.pt_bw[q] = pb | ((en - st + 1) << 8); W.pt_energy[q] = be;
                            }
                            const double sbe = wave_sum_f64(upd ? be : 0.0);       // integer-valued: exact in any order
                            accS -= sbe; accC += sbe;
                            if (!overflow) n_pt += nu;
                        }
                        // 5. unassigned peaks above the floor open new tracks, in peak order (lane = peak)
                        const bool mk = lane < n && asg == -1 && (double)pk_amp > fl;
                        const uint64_t nm = __ballot(mk);
                        const int nnew = __popcll(nm);
                        if (n_tr + nnew > p.tcap || n_pt + nnew > p.pcap || n_act + nnew > AC) overflow = true;
                        else if (mk) {
                            const int r = __popcll(nm & lanemask_lt(lane));
                            const int t = n_tr + r, q = n_pt + r, j = n_act + r;
                            const double be = pk_phi - pk_plo;
                            a_last_frame[j] = nfile; a_len[j] = 1; a_gid[j] = t; a_bins[j] = (uint32_t)pk_l; a_amp[j] = pk_amp;
                            a_vel[j] = 0; a_sumE[j] = be; a_sumEbin[j] = be * pk_l;
                            W.tr_len[t] = 1; W.tr_sumE[t] = be; W.tr_sumEbin[t] = be * pk_l;
                            W.pt_track[q] = t; W.pt_bw[q] = pk_l | ((pk_s - pk_i + 1) << 8); W.pt_energy[q] = be;
                        }
                        if (!overflow) { n_tr += nnew; n_pt += nnew; n_act += nnew; }
                        // file this frame's point range under its (possibly stale) index
                        if (reset_this_frame) { stale_d = nfile; stale_p1 = n_pt; }
                        else if (lane == 0 && nfile < p.fcap + 2) { W.d_p0[nfile] = p_begin; W.d_p1[nfile] = n_pt; W.d_gen[nfile] = gen; }
                        wsync();
                    }
                    if (c_started < 2) c_started++; else no_fm = 0;
                }
            }
            if (p.trace && lane == 0) {      // same row the oracle / ref_driver.js trace records, + tracker totals
                double* tr = p.trace + ((uint64_t)p.frame_off[clip] + f) * 12;
                tr[0] = c_ci; tr[1] = c_started; tr[2] = no_fm; tr[3] = ctx_max; tr[4] = floor_; tr[5] = n; tr[6] = pbin;
                tr[7] = h; tr[8] = d; tr[9] = g; tr[10] = accS; tr[11] = accC;
            }
            c_ci++;
            if (do_reset) WSA_RESET_SEGMENT(-1);            // the reference's Promise .then (quirk 8)
            g_a = g_b; n_a = n_b; g_b = g_c; n_b = n_c;
            e_pk = nx_pk; e_amp = nx_amp; e_plo = nx_plo; e_phi = nx_phi;
        }
        // ---- end of input: segment_truncate (ref @B30757) -> O(c_ci) -> L(1)
        finalize(c_ci);
        WSA_RESET_SEGMENT(1);
        if (lane == 0) { p.counts[2 * clip] = (uint32_t)nseg; p.counts[2 * clip + 1] = (uint32_t)nrows; if (overflow) atomicOr(p.flags, 1u); }
#undef WSA_RESET_SEGMENT
    }
}

void launch_tracker(const TrParams& p, int n_waves, hipStream_t s) {
    if (p.n_clips == 0) return;
    hipLaunchKernelGGL(tracker_kernel, dim3(n_waves), dim3(64), 0, s, p);
}

// ---- compaction: per-clip fixed-stride outputs -> dense (clip, si, syllable)-ordered tables
__global__ void compact_scan_kernel(CompactParams p) {
    // single block; n_clips is small relative to the rest of the work
    __shared__ uint32_t s_rows[256], s_segs[256];
    const int tid = threadIdx.x;
    const uint32_t per = (p.n_clips + 255) / 256;
    const uint32_t c0 = tid * per, c1 = min(p.n_clips, c0 + per);
    uint32_t rs = 0, ss = 0;
    for (uint32_t c = c0; c < c1; c++) { ss += p.counts[2 * c]; rs += p.counts[2 * c + 1]; }
    s_rows[tid] = rs; s_segs[tid] = ss;
    __syncthreads();
    if (tid == 0) {
        uint32_t ar = 0, as = 0;
        for (int i = 0; i < 256; i++) { const uint32_t r = s_rows[i], s = s_segs[i]; s_rows[i] = ar; s_segs[i] = as; ar += r; as += s; }
        p.totals[0] = ar; p.totals[1] = as;
        p.clip_row_off[p.n_clips] = ar; p.clip_seg_off[p.n_clips] = as;
    }
    __syncthreads();
    uint32_t ar = s_rows[tid], as = s_segs[tid];
    for (uint32_t c = c0; c < c1; c++) { p.clip_row_off[c] = ar; p.clip_seg_off[c] = as; as += p.counts[2 * c]; ar += p.counts[2 * c + 1]; }
}

__global__ __launch_bounds__(64) void compact_gather_kernel(CompactParams p) {
    const uint32_t clip = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t nseg = p.counts[2 * clip], nrow = p.counts[2 * clip + 1];
    const uint32_t so = p.clip_seg_off[clip], ro = p.clip_row_off[clip];
    const int32_t* sin = p.seg_in + (uint64_t)clip * p.seg_cap * 4;
    for (uint32_t i = lane; i < nseg; i += 64) {
        int32_t* o = p.seg_out + (uint64_t)(so + i) * 4;
        o[0] = (int32_t)clip; o[1] = sin[4 * i]; o[2] = sin[4 * i + 1]; o[3] = sin[4 * i + 2];
    }
    const int32_t* min_ = p.row_meta_in + (uint64_t)clip * p.row_cap * 8;
    const double* fin = p.row_feat_in + (uint64_t)clip * p.row_cap * WSA_NFEAT;
    for (uint32_t i = lane; i < nrow; i += 64) {
        int32_t* o = p.row_meta_out + (uint64_t)(ro + i) * 8;
        const int32_t* m = min_ + 8 * i;
        // the dispatcher indexes segments_ci with the RESULT index (ref @B29138 / @B29622): after a
        // dropped segment the timestamps come from the wrong entry — reproduced, not repaired
        const int si = m[1];
        const int32_t ts = sin[4 * si], tl = sin[4 * si + 1];
        o[0] = m[0]; o[1] = si; o[4] = m[4]; o[5] = m[5]; o[6] = m[6]; o[7] = m[7];
        if (p.level == 10 || p.level == 13) { o[2] = ts + m[2]; o[3] = m[3]; }     // syllable row (ref @B31114)
        else { o[2] = ts; o[3] = tl; }                                            // segment row (ref @B31504)
    }
    for (uint64_t i = lane; i < (uint64_t)nrow * WSA_NFEAT; i += 64) p.row_feat_out[(uint64_t)ro * WSA_NFEAT + i] = fin[i];
}

void launch_compact(const CompactParams& p, hipStream_t s) {
    if (p.n_clips == 0) return;
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(256), 0, s, p);
    hipLaunchKernelGGL(compact_gather_kernel, dim3(p.n_clips), dim3(64), 0, s, p);
}

}  // namespace wsa
