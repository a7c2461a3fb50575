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

// The LDS image of the table the kernel stages with straight 16-byte copies: 32 rows of pairs [o][i] = (K[o][i], K[o+1][i]), row stride 33 pairs
void resample_table_image(const std::vector<float>& K, std::vector<float>& img) {
    img.assign((size_t)2 * RS_OFFS * (RS_TAPS + 1), 0.f);
    for (int o = 0; o < RS_OFFS; o++)
        for (int i = 0; i < RS_TAPS; i++) {
            img[2 * ((size_t)o * (RS_TAPS + 1) + i)] = K[(size_t)o * RS_TAPS + i];
            img[2 * ((size_t)o * (RS_TAPS + 1) + i) + 1] = K[(size_t)(o + 1) * RS_TAPS + i];
        }
}

// A block converts S * J consecutive outputs of one clip (J = rs_j(S, ratio)), lane t the outputs n0 + t + j S (j < J).  S is a multiple of
// the period L of the conversion (fs_in / fs_out = M / L reduced) whenever that period is short, so that a lane's outputs
// share their sub-sample offset and with it the two kernel rows: the rows are read from LDS once per lane (64 registers)
// instead of once per output.  (Positions are the fp64 products of the specification; where rounding moves an output to a
// neighbouring row — or the period is long — the rows are re-read.)
// Per output the lane needs 32 consecutive inputs and 64 multiply-adds (both rows on the same inputs): the two rows live interleaved,
// (K[o][i], K[o+1][i]) per register pair, and one v_pk_fma_f32 with the input broadcast to both halves advances both sums: 32 VALU
// instructions per output instead of 64 (each half is the IEEE fma of the specification, ascending i).
// A block is as long as its staging: the 8 KB table image and the block's run of inputs (17 KB at 44.1 -> 48 kHz) come in as batches of
// 16-byte loads that are all in flight before the first LDS store (one load per loop trip, as the first version had it, is a global
// round trip per 768 bytes: 27 trips = the whole 2.2 ms of the kernel; profiles/r04_notes.md).
// inputs a block stages: J = RS_STAGED / (S * ratio) outputs per lane, at most 96 (17 KB of staged input + the 8 KB table: five blocks of three
// waves per CU; 44.1 -> 48 kHz J = 20 / 28 / 36: 1.71 / 1.61 / 1.61 ms, 16 -> 48 kHz J = 24 / 48 / 64 / 96: 1.32 / 1.23 / 1.21 / 1.32 ms)
constexpr int RS_STAGED = 4300;
inline int rs_j(int S, double ratio) { const double j = (double)RS_STAGED / ((double)S * ratio); return j < 1.0 ? 1 : (j > 96.0 ? 96 : (int)j); }     // (a factor-16 reduction: one output per lane, 16 KB staged)
constexpr int RS_KSTRIDE = RS_TAPS + 1;
constexpr int RS_IMG4 = 2 * RS_OFFS * RS_KSTRIDE / 4;      // the table image in 16-byte words
typedef float rs_v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float rs_lds_cf;
__host__ __device__ inline int rs_xlen(int span) { return (span + 6) & ~3; }    // staged inputs: the span, up to 3 samples in front of it (the run starts at a multiple of 4), rounded up to whole 16-byte words

__global__ __launch_bounds__(512) void resample_kernel(RsParams p) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    rs_v2f* const s_k = reinterpret_cast<rs_v2f*>(s_mem);                               // [32][33] pairs
    float* const s_x0 = s_mem + 2 * RS_OFFS * RS_KSTRIDE;                               // s_x0[q] = x[lo4 + q]
    const uint32_t clip = blockIdx.y;
    const uint64_t n_in = p.n_in[clip], n_out = p.n_out[clip];
    const int J = p.J;
    if ((uint64_t)blockIdx.x * (uint64_t)p.chunks * (uint64_t)(p.S * J) >= n_out) return;
    const float* x = p.in + (uint64_t)clip * p.stride_in;
    const int tid = threadIdx.x, nt = blockDim.x;
    constexpr int U = 6;
    // (loads are unconditional from clamped addresses, so that a batch is straight-line code with every load in flight; what a clamped
    //  load fetched for a place outside the clip is replaced by zeros — or, on the clip's edges, by guarded single loads — before the store)
    {
        const float4* timg = reinterpret_cast<const float4*>(p.table);
        float4* d = reinterpret_cast<float4*>(s_mem);
        for (int q0 = 0; q0 < RS_IMG4; q0 += 3 * nt) {
            float4 v[3];
#pragma unroll
            for (int u = 0; u < 3; u++) v[u] = timg[min(q0 + u * nt + tid, RS_IMG4 - 1)];
            asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x));       // (keeps the loads from sinking into the guarded stores, one round trip each)
#pragma unroll
            for (int u = 0; u < 3; u++) { const int q = q0 + u * nt + tid; if (q < RS_IMG4) d[q] = v[u]; }
        }
    }
    const uint32_t n_out32 = (uint32_t)n_out;
    float* const out = p.out + (uint64_t)clip * p.stride_out;
    // the block takes p.chunks consecutive runs of S * J outputs: the table is staged once for all of them
    for (int c = 0; c < p.chunks; c++) {
    const uint64_t n0 = ((uint64_t)blockIdx.x * (uint64_t)p.chunks + (uint64_t)c) * (uint64_t)(p.S * J);
    if (n0 >= n_out) break;
    if (c) __syncthreads();                                                 // the previous run's inputs are not needed any more
    int ts = tid;
    asm volatile("" : "+v"(ts));                                            // (what the staging derives from the lane number stays inside the loop: hoisted, it held 90 registers across the multiply-adds)
    // inputs from the first tap of output n0 (rounded down to a multiple of 4 samples) to the last tap of the run's last output
    const int64_t lo = (int64_t)floor((double)n0 * p.ratio) - RS_TAPS / 2;
    const int64_t lo4 = lo & ~(int64_t)3;
    const int xlen4 = rs_xlen(p.span) / 4;
    // s_x0[q] = x[lo4 + q] where that is inside the clip (q_lo <= q <= q_hi), else 0; indices relative to the run's first staged sample fit 32 bits
    const float* xb = x + lo4;
    const int q_lo = lo4 < 0 ? (int)(-lo4) : 0;
    const int64_t hi64 = (int64_t)n_in - 1 - lo4;
    const int q_hi = hi64 > 0x7fffffff ? 0x7fffffff : (int)hi64;            // (>= q_lo: a run with outputs has inputs)
    const int v_lo = (q_lo + 3) & ~3, v_hi = (q_hi - 3) & ~3;               // first / last whole 16-byte word inside the clip
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && v_hi >= v_lo && q_hi >= 3) {
        float4* d = reinterpret_cast<float4*>(s_x0);
        for (int q0 = 0; q0 < xlen4; q0 += U * nt) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const float4*>(xb + min(max(4 * (q0 + u * nt + ts), v_lo), v_hi));
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q4 = q0 + u * nt + ts, q = 4 * q4;
                if (q < v_lo || q > v_hi) {                                 // the clip's edges: sample by sample, zeros outside
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (q4 < xlen4 && q + 3 >= q_lo && q <= q_hi) {
                        if (q >= q_lo) v[u].x = xb[q];
                        if (q + 1 >= q_lo && q + 1 <= q_hi) v[u].y = xb[q + 1];
                        if (q + 2 >= q_lo && q + 2 <= q_hi) v[u].z = xb[q + 2];
                        if (q + 3 <= q_hi) v[u].w = xb[q + 3];
                    }
                }
                if (q4 < xlen4) d[q4] = v[u];
            }
        }
    } else {                                                                // a clip that does not start on a 16-byte boundary (or a run without a whole word inside the clip): words
        const int xlen = 4 * xlen4;
        for (int q0 = 0; q0 < xlen; q0 += U * nt) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = xb[min(max(q0 + u * nt + ts, q_lo), max(q_hi, q_lo))];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = q0 + u * nt + ts;
                if (q < xlen) s_x0[q] = (q >= q_lo && q <= q_hi) ? v[u] : 0.f;
            }
        }
    }
    __syncthreads();
    if (tid >= p.S) continue;
    rs_v2f k12[RS_TAPS];                                                    // (rows held across the staging would cost its registers: 196 instead of 106)
    int o_have = -1;
    // per output beside the 32 packed multiply-adds and 16 LDS reads: position (7 instructions), window start (3), interpolation and
    // store (9).  Outputs are numbered in 32 bits here (a clip holds fewer than 2^32 samples: n_out is 32 bits wide).
    const uint32_t nb = (uint32_t)n0 + (uint32_t)tid;
    const double lo_taps = (double)(lo4 + RS_TAPS / 2);                      // |lo4| < 2^53: exact
    for (int j = 0; j < J; j++) {
        const uint32_t n = nb + (uint32_t)(j * p.S);
        if (n >= n_out32) break;
        // pos = n * ratio; vo = (pos - floor(pos)) * 32; o = (int)vo; f = vo - o — with v_fract_f64 for the two differences (both exact, as the
        // subtractions are) and the window start straight from pos - (lo4 + 16) (exact: 0 <= difference <= pos; the conversion truncates)
        const double pos = (double)n * p.ratio;
        const double vo = __builtin_amdgcn_fract(pos) * RS_OFFS;
        const int o = (int)vo;
        const double f = __builtin_amdgcn_fract(vo);
        if (o != o_have) {
            const rs_v2f* r1 = s_k + o * RS_KSTRIDE;
#pragma unroll
            for (int i = 0; i < RS_TAPS; i++) k12[i] = r1[i];
            o_have = o;
        }
        const int w = (int)(pos - lo_taps);                                  // first input of the window, relative to lo4 (>= 0)
        // one LDS address, sixteen reads with immediate offsets (left alone the compiler folds the array's base into sixteen separate addresses)
        rs_lds_cf* xw = (rs_lds_cf*)(s_x0 + w);
        asm volatile("" : "+v"(xw));
        // all sixteen reads first (LDS answers in order: the multiply-adds start with the first answer), then the chain of 32
        rs_v2f xv[RS_TAPS / 2];
#pragma unroll
        for (int i = 0; i < RS_TAPS / 2; i++) { xv[i].x = xw[2 * i]; xv[i].y = xw[2 * i + 1]; }
        __builtin_amdgcn_sched_barrier(0);
        rs_v2f acc; acc.x = 0.f; acc.y = 0.f;
#pragma unroll
        for (int i = 0; i < RS_TAPS / 2; i++) {
            // (s1, s2) = fma(x[2i], (k1, k2)[2i], (s1, s2)), then the same with x[2i+1]: v_pk_fma_f32, the input broadcast to both halves by op_sel
            rs_v2f xa, xb; xa.x = xv[i].x; xa.y = xv[i].x; xb.x = xv[i].y; xb.y = xv[i].y;
            acc = __builtin_elementwise_fma(xa, k12[2 * i], acc);
            acc = __builtin_elementwise_fma(xb, k12[2 * i + 1], acc);
        }
        out[n] = (float)((1.0 - f) * (double)acc.x + f * (double)acc.y);
    }
    }
}

// outputs per block row: a multiple of the conversion's period when the rates are integers with a short period
int resample_stride(double fs_in, double fs_out) {
    const double ri = std::floor(fs_in), ro = std::floor(fs_out);
    if (ri == fs_in && ro == fs_out && ri > 0 && ro > 0 && ri < 4e9 && ro < 4e9) {
        uint64_t a = (uint64_t)ri, b = (uint64_t)ro;
        while (b) { const uint64_t t = a % b; a = b; b = t; }
        const uint64_t L = (uint64_t)ro / a;                  // outputs per period
        // the multiple of the period closest to 256 from below if it leaves at most a fifth of its last wave's lanes idle (3 -> 255 of 256, 160 -> 160 of
        // 192: small blocks overlap their staging with the other blocks' arithmetic better than large ones — 44.1 -> 48 kHz: S = 160 1.65 ms, 320 2.2 ms),
        // else the multiple up to 512 that wastes the fewest lanes (147 -> 441 of 448)
        if (L <= 512) {
            const uint64_t S0 = L * (256 / L > 0 ? 256 / L : 1);
            auto waste = [](uint64_t S) { return (double)((S + 63) / 64 * 64 - S) / (double)((S + 63) / 64 * 64); };
            if (waste(S0) <= 0.2) return (int)S0;
            uint64_t best = S0;
            for (uint64_t S = L; S <= 512; S += L) if (S >= 128 && waste(S) < waste(best)) best = S;
            return (int)best;
        }
    }
    return 256;
}

void launch_resample(const RsParams& p, uint32_t n_clips, uint64_t max_out, hipStream_t s) {
    if (n_clips == 0 || max_out == 0) return;
    const size_t lds = sizeof(float) * ((size_t)2 * RS_OFFS * RS_KSTRIDE + (size_t)rs_xlen(p.span));
    const uint64_t per_block = (uint64_t)p.S * (uint64_t)p.J * (uint64_t)p.chunks;
    const dim3 grid((unsigned)((max_out + per_block - 1) / per_block), n_clips), block((unsigned)((p.S + 63) / 64 * 64));
    hipLaunchKernelGGL(resample_kernel, grid, block, lds, s, p);
}

int resample_outputs_per_lane(int S, double ratio) { return rs_j(S, ratio); }
int resample_span(double ratio, int S, int J) { return (int)std::ceil((double)(S * J - 1) * ratio) + RS_TAPS + 2; }

}  // namespace wsa
