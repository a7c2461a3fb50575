// resample.hip — K0: sample-rate conversion in front of the path (spec RS-1), one LANE per output sample.
//
// Stands in for the conversion to the context rate that the browser's decodeAudioData performs before the reference
// ever sees the samples (its offline path decodes into `new OfflineAudioContext(1, 48e6, 48e3)`, ref dist/main.js:2
// @B18769, i.e. always 48 kHz).  The converter is the browser's, not the reference's: nothing in the tree pins it
// ("parity unpinned").  RS-1 = the published windowed-sinc scheme of the Chromium family: 32 taps, 32 + 1 sub-sample
// offset kernels (Blackman window, cut-off 0.9 x the lower Nyquist), linear interpolation between the two neighbouring
// kernels, 16 zeros of history.  Bit-exact against oracle/resample.c (same table, same operation order).
#include "wsa_internal.hpp"
#include <cmath>

namespace wsa {

void build_resample_table(double fs_in, double fs_out, std::vector<float>& K) {
    const double ratio = fs_in / fs_out;
    const double scale = (ratio > 1.0 ? 1.0 / ratio : 1.0) * 0.9;
    const double pi = 3.14159265358979323846;
    K.resize((size_t)(RS_OFFS + 1) * RS_TAPS);
    for (int o = 0; o <= RS_OFFS; o++) {
        const double s = (double)o / RS_OFFS;
        for (int i = 0; i < RS_TAPS; i++) {
            const double pre = pi * ((double)(i - RS_TAPS / 2) - s);
            const double x = ((double)i - s) / RS_TAPS;
            const double w = 0.42 - 0.5 * std::cos(2.0 * pi * x) + 0.08 * std::cos(4.0 * pi * x);
            K[(size_t)o * RS_TAPS + i] = (float)(w * (pre == 0.0 ? scale : std::sin(scale * pre) / pre));
        }
    }
}

uint64_t resample_length(uint64_t n_in, double fs_in, double fs_out) { return (uint64_t)((double)n_in / (fs_in / fs_out)); }

// A block converts RS_BLOCK consecutive outputs of one clip: their inputs are one contiguous run, staged in LDS together
// with the 33 x 32 kernel table (row stride 33 words: lanes in different offset rows hit different banks).
constexpr int RS_BLOCK = 256;
constexpr int RS_KSTRIDE = RS_TAPS + 1;

__global__ __launch_bounds__(RS_BLOCK) void resample_kernel(RsParams p) {
    extern __shared__ float s_mem[];
    float* const s_k = s_mem;                                   // [33][33]
    float* const s_x = s_mem + (RS_OFFS + 1) * RS_KSTRIDE;      // [span]
    const uint32_t clip = blockIdx.y;
    const uint64_t n_in = p.n_in[clip], n_out = p.n_out[clip];
    const uint64_t n0 = (uint64_t)blockIdx.x * RS_BLOCK;
    if (n0 >= n_out) return;
    const float* x = p.in + (uint64_t)clip * p.stride_in;
    for (int q = threadIdx.x; q < (RS_OFFS + 1) * RS_TAPS; q += RS_BLOCK) s_k[(q / RS_TAPS) * RS_KSTRIDE + (q % RS_TAPS)] = p.table[q];
    // inputs [lo, lo + span): from the first tap of output n0 to the last tap of the block's last output
    const int64_t lo = (int64_t)floor((double)n0 * p.ratio) - RS_TAPS / 2;
    for (int q = threadIdx.x; q < p.span; q += RS_BLOCK) {
        const int64_t g = lo + q;
        s_x[q] = (g >= 0 && (uint64_t)g < n_in) ? x[g] : 0.f;
    }
    __syncthreads();
    const uint64_t n = n0 + threadIdx.x;
    if (n < n_out) {
        const double pos = (double)n * p.ratio;
        const double fl = floor(pos);
        const double vo = (pos - fl) * RS_OFFS;
        const int o = (int)vo;
        const double f = vo - (double)o;
        const float* k1 = s_k + o * RS_KSTRIDE;
        const float* k2 = k1 + RS_KSTRIDE;
        const float* xs = s_x + ((int64_t)fl - RS_TAPS / 2 - lo);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < RS_TAPS; i++) {
            const float xv = xs[i];
            const float p1 = xv * k1[i], p2 = xv * k2[i];           // -ffp-contract=off: products and sums rounded one by one
            s1 = s1 + p1; s2 = s2 + p2;
        }
        p.out[(uint64_t)clip * p.stride_out + n] = (float)((1.0 - f) * (double)s1 + f * (double)s2);
    }
}

void launch_resample(const RsParams& p, uint32_t n_clips, uint64_t max_out, hipStream_t s) {
    if (n_clips == 0 || max_out == 0) return;
    const size_t lds = sizeof(float) * ((size_t)(RS_OFFS + 1) * RS_KSTRIDE + (size_t)p.span);
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((max_out + RS_BLOCK - 1) / RS_BLOCK), n_clips), dim3(RS_BLOCK), lds, s, p);
}

int resample_span(double ratio) { return (int)std::ceil((RS_BLOCK - 1) * ratio) + RS_TAPS + 2; }

}  // namespace wsa
