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

// A block converts S * J consecutive outputs of one clip (J = rs_j(S)), lane t the outputs n0 + t + j S (j < J).  S is a multiple of
// the period L of the conversion (fs_in / fs_out = M / L reduced) whenever that period is short, so that a lane's outputs
// share their sub-sample offset and with it the two kernel rows: the rows are read from LDS once per lane (64 registers)
// instead of once per output.  (Positions are the fp64 products of the specification; where rounding moves an output to a
// neighbouring row — or the period is long — the rows are re-read.)
// Per output the lane needs 32 consecutive inputs and 64 multiply-adds (both rows on the same inputs):
//   * the two rows live interleaved, (K[o][i], K[o+1][i]) per register pair, and one v_pk_fma_f32 with the input broadcast to both
//     halves advances both sums: 32 VALU instructions per output instead of 64 (each half is the IEEE fma of the specification,
//     ascending i);
//   * the block's input run is staged in LDS TWICE, the second copy shifted by one sample, so that every lane finds its 32 inputs
//     8-byte aligned in one of the copies and reads them as 16 ds_read_b64 (256 B per LDS clock) instead of 32 ds_read_b32 (128).
// The 33 x 32 kernel table is staged as 32 rows of pairs [o][i] = (K[o][i], K[o+1][i]) (row stride 33 pairs: conflict-free).
constexpr int RS_OUT_PER_BLOCK = 4608;          // ~ outputs a block converts: J = RS_OUT_PER_BLOCK / S per lane (24 at S = 192, 28 at S = 160: 17 KB of staged input at 48 kHz out of 44.1 kHz)
__host__ __device__ inline int rs_j(int S) { const int j = RS_OUT_PER_BLOCK / S; return j < 4 ? 4 : j; }
constexpr int RS_KSTRIDE = RS_TAPS + 1;
typedef float rs_v2f __attribute__((ext_vector_type(2)));

template <bool TWO>          // TWO: the shifted second copy of the inputs (aligned 8-byte reads); else one copy, read as pairs of words (half the LDS: more blocks per CU)
__global__ __launch_bounds__(512) void resample_kernel(RsParams p) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    rs_v2f* const s_k = reinterpret_cast<rs_v2f*>(s_mem);                               // [32][33] pairs
    const int xlen = (p.span + 3) & ~1;                                                 // floats per copy (even)
    float* const s_x0 = s_mem + 2 * RS_OFFS * RS_KSTRIDE;                               // copy 0: s_x0[q] = x[lo + q]
    float* const s_x1 = s_x0 + xlen;                                                    // copy 1: s_x1[q] = x[lo + q - 1]
    const uint32_t clip = blockIdx.y;
    const uint64_t n_in = p.n_in[clip], n_out = p.n_out[clip];
    const int J = p.J;
    const uint64_t n0 = (uint64_t)blockIdx.x * (uint64_t)(p.S * J);
    if (n0 >= n_out) return;
    const float* x = p.in + (uint64_t)clip * p.stride_in;
    for (int q = threadIdx.x; q < RS_OFFS * RS_TAPS; q += blockDim.x) {
        const int o = q / RS_TAPS, i = q % RS_TAPS;
        rs_v2f w; w.x = p.table[o * RS_TAPS + i]; w.y = p.table[(o + 1) * RS_TAPS + i];
        s_k[o * RS_KSTRIDE + i] = w;
    }
    // inputs [lo, lo + span): from the first tap of output n0 to the last tap of the block's last output
    const int64_t lo = (int64_t)floor((double)n0 * p.ratio) - RS_TAPS / 2;
    for (int q = threadIdx.x; q < xlen; q += blockDim.x) {
        const int64_t g = lo + q;
        const float v = (g >= 0 && (uint64_t)g < n_in) ? x[g] : 0.f;
        s_x0[q] = v;
        if (TWO && q + 1 < xlen) s_x1[q + 1] = v;
    }
    if (TWO && threadIdx.x == 0) s_x1[0] = (lo - 1 >= 0 && (uint64_t)(lo - 1) < n_in) ? x[lo - 1] : 0.f;
    __syncthreads();
    if ((int)threadIdx.x >= p.S) return;
    rs_v2f k12[RS_TAPS];
    int o_have = -1;
#pragma unroll 4
    for (int j = 0; j < J; j++) {
        const uint64_t n = n0 + threadIdx.x + (uint64_t)j * p.S;
        if (n >= n_out) break;
        const double pos = (double)n * p.ratio;
        const double fl = floor(pos);
        const double vo = (pos - fl) * RS_OFFS;
        const int o = (int)vo;
        const double f = vo - (double)o;
        if (o != o_have) {
            const rs_v2f* r1 = s_k + o * RS_KSTRIDE;
#pragma unroll
            for (int i = 0; i < RS_TAPS; i++) k12[i] = r1[i];
            o_have = o;
        }
        const int w = (int)((int64_t)fl - RS_TAPS / 2 - lo);                 // first input of the window, relative to lo (>= 0)
        // an even w is 8-byte aligned in copy 0, an odd one in copy 1 (where the sample sits one place further up)
        const rs_v2f* xs = reinterpret_cast<const rs_v2f*>((TWO && (w & 1)) ? s_x1 + w + 1 : s_x0 + w);
        const float* xw = s_x0 + w;
        rs_v2f acc; acc.x = 0.f; acc.y = 0.f;
#pragma unroll
        for (int i = 0; i < RS_TAPS / 2; i++) {
            rs_v2f xv;
            if (TWO) xv = xs[i]; else { xv.x = xw[2 * i]; xv.y = xw[2 * i + 1]; }
            // (s1, s2) = fma(x[2i], (k1, k2)[2i], (s1, s2)), then the same with x[2i+1]: v_pk_fma_f32, the input broadcast to both halves by op_sel
            rs_v2f xa, xb; xa.x = xv.x; xa.y = xv.x; xb.x = xv.y; xb.y = xv.y;
            acc = __builtin_elementwise_fma(xa, k12[2 * i], acc);
            acc = __builtin_elementwise_fma(xb, k12[2 * i + 1], acc);
        }
        p.out[(uint64_t)clip * p.stride_out + n] = (float)((1.0 - f) * (double)acc.x + f * (double)acc.y);
    }
}

// outputs per block row: a multiple of the conversion's period when the rates are integers with a short period
int resample_stride(double fs_in, double fs_out) {
    const double ri = std::floor(fs_in), ro = std::floor(fs_out);
    if (ri == fs_in && ro == fs_out && ri > 0 && ro > 0 && ri < 4e9 && ro < 4e9) {
        uint64_t a = (uint64_t)ri, b = (uint64_t)ro;
        while (b) { const uint64_t t = a % b; a = b; b = t; }
        const uint64_t L = (uint64_t)ro / a;                  // outputs per period
        // a multiple of the period that fills whole waves where one exists up to 512 lanes (3 -> 192, 160 -> 320, 1 / 2 / 4 ... -> 256), else the
        // multiple closest to 256 from below (147 -> 147: 48 kHz -> 44.1 kHz leaves a fifth of its third wave idle)
        if (L <= 512) {
            for (uint64_t S = 256; S >= 128; S -= 64) if (S % L == 0) return (int)S;
            for (uint64_t S = 320; S <= 512; S += 64) if (S % L == 0) return (int)S;
            return (int)(L * (256 / L > 0 ? 256 / L : 1));
        }
    }
    return 256;
}

void launch_resample(const RsParams& p, uint32_t n_clips, uint64_t max_out, hipStream_t s) {
    if (n_clips == 0 || max_out == 0) return;
    const bool two = p.two != 0;
    const size_t lds = sizeof(float) * ((size_t)2 * RS_OFFS * RS_KSTRIDE + (two ? 2 : 1) * (size_t)((p.span + 3) & ~1));
    const uint64_t per_block = (uint64_t)p.S * (uint64_t)p.J;
    const dim3 grid((unsigned)((max_out + per_block - 1) / per_block), n_clips), block((unsigned)((p.S + 63) / 64 * 64));
    if (two) hipLaunchKernelGGL(resample_kernel<true>, grid, block, lds, s, p);
    else hipLaunchKernelGGL(resample_kernel<false>, grid, block, lds, s, p);
}

int resample_outputs_per_lane(int S) { return rs_j(S); }
int resample_span(double ratio, int S, int J) { return (int)std::ceil((double)(S * J - 1) * ratio) + RS_TAPS + 2; }

}  // namespace wsa
