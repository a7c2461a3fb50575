// resample.hip — K0: sample-rate conversion in front of the path (spec RS-1), one LANE per output sample.
//
// Stands in for the conversion to the context rate that the browser's decodeAudioData performs before the reference
// ever sees the samples (its offline path decodes into `new OfflineAudioContext(1, 48e6, 48e3)`, ref dist/main.js:2
// @B18769, i.e. always 48 kHz).  The converter is the browser's, not the reference's: nothing in the tree pins it
// ("parity unpinned").  RS-1 = the published windowed-sinc scheme of the Chromium family: 32 taps, 32 + 1 sub-sample
// offset kernels (Blackman window, cut-off 0.9 x the lower Nyquist), linear interpolation between the two neighbouring
// kernels, 16 zeros of history, one fused multiply-add per tap.  Bit-exact against oracle/resample.c (same table, same operation order).
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

// A block converts S * RS_J consecutive outputs of one clip, lane t the outputs n0 + t + j S (j < RS_J).  S is a multiple of
// the period L of the conversion (fs_in / fs_out = M / L reduced) whenever that period is short, so that a lane's outputs
// share their sub-sample offset and with it the two kernel rows: the rows are read from LDS once per lane (64 registers)
// instead of once per output, which halves the LDS traffic the kernel is bound by.  (Positions are the fp64 products of the
// specification; where rounding moves an output to a neighbouring row — or the period is long — the rows are re-read.)
// The block's input run and the 33 x 32 kernel table (row stride 33 words: conflict-free) are staged in LDS.
constexpr int RS_J = 16;
constexpr int RS_KSTRIDE = RS_TAPS + 1;

__global__ __launch_bounds__(256) void resample_kernel(RsParams p) {
    extern __shared__ float s_mem[];
    float* const s_k = s_mem;                                   // [33][33]
    float* const s_x = s_mem + (RS_OFFS + 1) * RS_KSTRIDE;      // [span]
    const uint32_t clip = blockIdx.y;
    const uint64_t n_in = p.n_in[clip], n_out = p.n_out[clip];
    const uint64_t n0 = (uint64_t)blockIdx.x * (uint64_t)(p.S * RS_J);
    if (n0 >= n_out) return;
    const float* x = p.in + (uint64_t)clip * p.stride_in;
    for (int q = threadIdx.x; q < (RS_OFFS + 1) * RS_TAPS; q += blockDim.x) s_k[(q / RS_TAPS) * RS_KSTRIDE + (q % RS_TAPS)] = p.table[q];
    // inputs [lo, lo + span): from the first tap of output n0 to the last tap of the block's last output
    const int64_t lo = (int64_t)floor((double)n0 * p.ratio) - RS_TAPS / 2;
    for (int q = threadIdx.x; q < p.span; q += blockDim.x) {
        const int64_t g = lo + q;
        s_x[q] = (g >= 0 && (uint64_t)g < n_in) ? x[g] : 0.f;
    }
    __syncthreads();
    if ((int)threadIdx.x >= p.S) return;
    float k1[RS_TAPS], k2[RS_TAPS];
    int o_have = -1;
#pragma unroll 2
    for (int j = 0; j < RS_J; j++) {
        const uint64_t n = n0 + threadIdx.x + (uint64_t)j * p.S;
        if (n >= n_out) break;
        const double pos = (double)n * p.ratio;
        const double fl = floor(pos);
        const double vo = (pos - fl) * RS_OFFS;
        const int o = (int)vo;
        const double f = vo - (double)o;
        if (o != o_have) {
            const float* r1 = s_k + o * RS_KSTRIDE;
#pragma unroll
            for (int i = 0; i < RS_TAPS; i++) { k1[i] = r1[i]; k2[i] = r1[RS_KSTRIDE + i]; }
            o_have = o;
        }
        const float* xs = s_x + ((int64_t)fl - RS_TAPS / 2 - lo);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < RS_TAPS; i++) { const float xv = xs[i]; s1 = __builtin_fmaf(xv, k1[i], s1); s2 = __builtin_fmaf(xv, k2[i], s2); }
        p.out[(uint64_t)clip * p.stride_out + n] = (float)((1.0 - f) * (double)s1 + f * (double)s2);
    }
}

// outputs per block row: a multiple of the conversion's period when the rates are integers with a short period
int resample_stride(double fs_in, double fs_out) {
    const double ri = std::floor(fs_in), ro = std::floor(fs_out);
    if (ri == fs_in && ro == fs_out && ri > 0 && ro > 0 && ri < 4e9 && ro < 4e9) {
        uint64_t a = (uint64_t)ri, b = (uint64_t)ro;
        while (b) { const uint64_t t = a % b; a = b; b = t; }
        const uint64_t L = (uint64_t)ro / a;                  // outputs per period
        if (L <= 256) return (int)(L * ((128 + L - 1) / L) <= 256 ? L * ((128 + L - 1) / L) : L);
    }
    return 256;
}

void launch_resample(const RsParams& p, uint32_t n_clips, uint64_t max_out, hipStream_t s) {
    if (n_clips == 0 || max_out == 0) return;
    const size_t lds = sizeof(float) * ((size_t)(RS_OFFS + 1) * RS_KSTRIDE + (size_t)p.span);
    const uint64_t per_block = (uint64_t)p.S * RS_J;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((max_out + per_block - 1) / per_block), n_clips), dim3((unsigned)((p.S + 63) / 64 * 64)), lds, s, p);
}

int resample_span(double ratio, int S) { return (int)std::ceil((double)(S * RS_J - 1) * ratio) + RS_TAPS + 2; }

}  // namespace wsa
