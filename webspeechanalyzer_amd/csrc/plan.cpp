// plan.cpp — host side of the front end: derives frame geometry and the fp32 tables (window,
// twiddles, mel filterbank) from (config, sample rate).  Specification "FE-1" (DESIGN.md): the
// reference's own front end (analyzernode.min.js, fetched at run time, ref dist/main.js:2 @B6480)
// is not in the reference tree; the wire contract it must honour is @B6726 (config) and @B8568
// (one Uint32Array(spec_bands) per frame).
#include <cmath>
#include "wsa_internal.hpp"

namespace wsa {

static const double kPi = 3.14159265358979323846;

// W_N^j = (cos, -sin)(2 pi j / N), folded to the first octant with integer arithmetic so that the
// axis / diagonal points are exact and mirror-image entries are equal.
static void make_twiddle(int j, int N, float& wr, float& wi) {
    j %= N;
    const int oct = (int)(((long long)j * 8) / N);
    const long long rem = (long long)j * 8 - (long long)oct * N;
    const double h = 0.70710678118654752440;
    double c, s;
    if (rem == 0) {
        const double cs[8] = {1, h, 0, -h, -1, -h, 0, h};
        const double sn[8] = {0, h, 1, h, 0, -h, -1, -h};
        c = cs[oct]; s = sn[oct];
    } else {
        const long long num = (oct & 1) ? ((long long)N - rem) : rem;
        const double t = (double)num / (double)N * (kPi / 4);
        const double big = std::cos(t), small = std::sin(t);
        switch (oct) {
            case 0: c = big; s = small; break;
            case 1: c = small; s = big; break;
            case 2: c = -small; s = big; break;
            case 3: c = -big; s = small; break;
            case 4: c = -big; s = -small; break;
            case 5: c = -small; s = -big; break;
            case 6: c = small; s = -big; break;
            default: c = big; s = -small; break;
        }
    }
    wr = (float)c; wi = (float)(-s);
}

static double hz_to_mel(double f) { return 2595.0 * std::log10(1.0 + f / 700.0); }
static double mel_to_hz(double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); }

bool build_fe_plan(const wsa_config& cfg, double fs, FePlanHost& p, std::string& err) {
    if (!(fs > 0) || !(cfg.f_max > 0) || cfg.N_fft_bins < 1 || cfg.N_mel_bins < 1) { err = "invalid front-end settings"; return false; }
    p.spec_type = cfg.spec_type;
    p.win = (int)std::floor(fs * cfg.window_width / 1000.0 + 0.5);
    p.hop = (int)std::floor(fs * cfg.window_step / 1000.0 + 0.5);
    int need = (int)std::ceil(fs * cfg.N_fft_bins / cfg.f_max);
    if (p.win > need) need = p.win;
    // FE-1 F2: the smallest of {2^k, 3 * 2^k} >= need, >= 256 (bin width ~ f_max / N_fft_bins, ref index.html:269:
    // 1024 at 16 kHz, 3072 at 44.1 / 48 kHz)
    p.nfft = 256;
    for (;;) {
        if (p.nfft >= need) break;
        if (p.nfft / 2 * 3 >= need) { p.nfft = p.nfft / 2 * 3; break; }
        p.nfft <<= 1;
    }
    p.n2 = p.nfft / 2;
    p.three = (p.n2 % 3) == 0 ? 1 : 0;           // N2 = 3 M: one radix-3 stage in front of three M-point transforms
    p.M = p.three ? p.n2 / 3 : p.n2;
    p.R = p.M / 64;
    p.kmax = (int)std::floor(cfg.f_max * p.nfft / fs);
    if (p.kmax > p.n2) p.kmax = p.n2;
    p.bands = cfg.spec_type == 1 ? cfg.N_mel_bins : cfg.N_fft_bins;
    if (p.win < 2 || p.hop < 1) { err = "window_width / window_step too small for this sample rate"; return false; }
    if (cfg.spec_type < 1 || cfg.spec_type > 3) { err = "spec_type must be 1, 2 or 3"; return false; }
    if (cfg.spec_type != 1 && cfg.N_fft_bins > p.n2 + 1) { err = "N_fft_bins exceeds the FFT size"; return false; }
    if (p.bands > 256) { err = "more than 256 spectrum bands are not supported"; return false; }

    p.window.resize(p.win);
    for (int n = 0; n < p.win; n++) p.window[n] = (float)(0.5 - 0.5 * std::cos(2.0 * kPi * n / p.win));
    p.tw_n2.resize(2 * (size_t)p.n2);
    for (int j = 0; j < p.n2; j++) make_twiddle(j, p.n2, p.tw_n2[2 * j], p.tw_n2[2 * j + 1]);
    p.tw_m.clear();
    if (p.three) {
        p.tw_m.resize(2 * (size_t)p.M);
        for (int j = 0; j < p.M; j++) make_twiddle(j, p.M, p.tw_m[2 * j], p.tw_m[2 * j + 1]);
    }
    p.tw_64.resize(128);
    for (int j = 0; j < 64; j++) make_twiddle(j, 64, p.tw_64[2 * j], p.tw_64[2 * j + 1]);
    p.tw_nfft.resize(2 * (size_t)(p.kmax + 1));
    for (int k = 0; k <= p.kmax; k++) make_twiddle(k, p.nfft, p.tw_nfft[2 * k], p.tw_nfft[2 * k + 1]);
    p.gain = (float)cfg.pre_norm_gain;
    p.emph.resize(p.bands);
    for (int m = 0; m < p.bands; m++) p.emph[m] = (float)(1.0 + m * cfg.high_f_emph);
    p.bins_hz.resize(p.bands);
    p.mel_k0.clear(); p.mel_cnt.clear(); p.mel_off.clear(); p.mel_w.clear();
    if (cfg.spec_type == 1) {
        const int M = p.bands;
        const double df = fs / p.nfft, mlo = hz_to_mel(cfg.f_min), mhi = hz_to_mel(cfg.f_max);
        std::vector<double> edge(M + 2);
        for (int j = 0; j < M + 2; j++) edge[j] = mel_to_hz(mlo + (mhi - mlo) * j / (M + 1));
        for (int m = 0; m < M; m++) {
            const double lo = edge[m], ce = edge[m + 1], hi = edge[m + 2];
            p.bins_hz[m] = ce;
            int first = -1, cnt = 0;
            const int off = (int)p.mel_w.size();
            for (int k = 0; k <= p.kmax; k++) {
                const double fk = k * df;
                double w = 0;
                if (fk > lo && fk < hi) {
                    const double up = (fk - lo) / (ce - lo), dn = (hi - fk) / (hi - ce);
                    w = up < dn ? up : dn;
                }
                if (w > 0) {
                    if (first < 0) first = k;
                    p.mel_w.resize(off + (k - first) + 1, 0.0f);
                    p.mel_w[off + (k - first)] = (float)w * 0.25f;
                    cnt = k - first + 1;
                }
            }
            if (first < 0) {   // triangle narrower than one bin: interpolated power at its centre
                const double pos = ce / df;
                first = (int)std::floor(pos);
                double fr = pos - first;
                if (first >= p.kmax) { first = p.kmax; fr = 0; }
                p.mel_w.push_back((float)(1.0 - fr) * 0.25f); cnt = 1;
                if (fr > 0) { p.mel_w.push_back((float)fr * 0.25f); cnt = 2; }
            }
            p.mel_k0.push_back(first); p.mel_cnt.push_back(cnt); p.mel_off.push_back(off);
        }
    } else {
        for (int m = 0; m < p.bands; m++) p.bins_hz[m] = m * fs / p.nfft;
    }
    return true;
}

}  // namespace wsa
