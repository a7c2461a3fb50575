/* resample.c — CPU restatement of the sample-rate conversion in front of the path (spec RS-1).
 * TEST INFRASTRUCTURE (oracle): only tests/, smoke() and bench.py's cpu_baseline may use it.
 *
 * The reference never resamples itself: its offline path hands the file to the browser's decodeAudioData, which
 * converts to the context rate — always 48 kHz there (`new OfflineAudioContext(1, 48e6, 48e3)`, ref dist/main.js:2
 * @B18769) — with the browser's own converter.  That converter is not in the reference tree and nothing in it pins
 * its output ("parity unpinned", like the worklet front end).  RS-1 restates the published algorithm of the
 * windowed-sinc converter of the Chromium family (kernel of 32 taps, 32 sub-sample offsets + 1, Blackman window,
 * cut-off 0.9 x the lower Nyquist, linear interpolation between the two neighbouring offset kernels, 16 zeros of
 * history in front of the first sample, output length trunc(n_in * fs_out / fs_in)):
 *   ratio = fs_in / fs_out;  scale = (ratio > 1 ? 1 / ratio : 1) * 0.9
 *   K[o][i] (o = 0..32, i = 0..31), in double, rounded once to float:
 *       s = o / 32, pre = pi (i - 16 - s), x = (i - s) / 32, w = 0.42 - 0.5 cos(2 pi x) + 0.08 cos(4 pi x)
 *       K = w * (pre == 0 ? scale : sin(scale * pre) / pre)
 *   out[n]: pos = n * ratio (double), src = floor(pos), vo = (pos - src) * 32, o = (int)vo, f = vo - o
 *       s1 = fmaf(x[src + i - 16], K[o][i], s1) for i = 0..31 from 0,  s2 the same with K[o + 1]     (float, one rounding per tap)
 *       out[n] = (float)((1 - f) * (double)s1 + f * (double)s2)               (x = 0 outside the clip) */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "wsa_oracle.h"

#define RS_TAPS 32
#define RS_OFFS 32

uint64_t wsa_or_resample_length(uint64_t n_in, double fs_in, double fs_out) {
    const double ratio = fs_in / fs_out;
    return (uint64_t)((double)n_in / ratio);
}

void wsa_or_resample_table(double fs_in, double fs_out, float *K /* [33][32] */) {
    const double ratio = fs_in / fs_out;
    const double scale = (ratio > 1.0 ? 1.0 / ratio : 1.0) * 0.9;
    const double pi = 3.14159265358979323846;
    for (int o = 0; o <= RS_OFFS; o++) {
        const double s = (double)o / RS_OFFS;
        for (int i = 0; i < RS_TAPS; i++) {
            const double pre = pi * ((double)(i - RS_TAPS / 2) - s);
            const double x = ((double)i - s) / RS_TAPS;
            const double w = 0.42 - 0.5 * cos(2.0 * pi * x) + 0.08 * cos(4.0 * pi * x);
            K[o * RS_TAPS + i] = (float)(w * (pre == 0.0 ? scale : sin(scale * pre) / pre));
        }
    }
}

void wsa_or_resample(const float *in, uint64_t n_in, double fs_in, double fs_out, float *out) {
    float *K = malloc(sizeof(float) * (RS_OFFS + 1) * RS_TAPS);
    wsa_or_resample_table(fs_in, fs_out, K);
    const double ratio = fs_in / fs_out;
    const uint64_t n_out = wsa_or_resample_length(n_in, fs_in, fs_out);
    for (uint64_t n = 0; n < n_out; n++) {
        const double pos = (double)n * ratio;
        const double fl = floor(pos);
        const int64_t src = (int64_t)fl;
        const double vo = (pos - fl) * RS_OFFS;
        const int o = (int)vo;
        const double f = vo - (double)o;
        const float *k1 = K + o * RS_TAPS, *k2 = k1 + RS_TAPS;
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < RS_TAPS; i++) {
            const int64_t q = src + i - RS_TAPS / 2;
            const float x = (q >= 0 && (uint64_t)q < n_in) ? in[q] : 0.f;
            s1 = fmaf(x, k1[i], s1); s2 = fmaf(x, k2[i], s2);
        }
        out[n] = (float)((1.0 - f) * (double)s1 + f * (double)s2);
    }
    free(K);
}
