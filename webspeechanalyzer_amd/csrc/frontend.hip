// frontend.hip — K1: batched PCM -> Hann window -> 1024-point real FFT -> power -> mel bands ->
// emphasis / gain -> Uint32 frame, for gfx950 (MI355X).
//
// Stands in for the reference's "spectrum-processor" AudioWorklet (source not in the reference
// tree: fetched from unpkg at run time, ref dist/main.js:2 @B6480; configured @B6726; its output,
// one Uint32Array(spec_bands) per frame, is consumed @B8568).  Arithmetic = specification FE-1
// (DESIGN.md): bit-exact with oracle/frontend.c.  Compiled with -ffp-contract=off: only the
// explicit __builtin_fmaf calls fuse.
//
// Mapping: ONE WAVEFRONT PER FRAME, everything in registers except two LDS transposes.
//   N2 = 512 packed complex points = 64 lanes x 8 points.  DIF radices [8, 8, 8]:
//   pass 1  lane m holds z[64a + m], a = 0..7           (only a < AZ are non-zero: window <= 128 AZ samples)
//   X1      LDS transpose: lane (a', c) gets b = 0..7 of sub-FFT a'      (row stride 72: conflict-free)
//   pass 2  radix 8 over b, twiddle W_64^{c b'}
//   X2      LDS transpose inside 8-lane groups: lane (a', b') gets c = 0..7  (row stride 9)
//   pass 3  radix 8 over c  ->  lane holds Z[k0 + 64 c'], k0 = a' + 8 b'
//   split   real-FFT recombination with the partner lane (k0 <-> 64 - k0) through ds_bpermute
//   mel     4|X|^2 to LDS, each lane sums its bands (fmaf chain, ascending bin), u32 store (coalesced)
// Twiddles, the window and the split factors are loop-invariant per lane and live in VGPRs for all
// frames a wave processes.  PCM is read exactly once with 512-byte-per-instruction coalesced loads.
#include "wsa_internal.hpp"
#include "wave_ops.hpp"

namespace wsa {

struct __attribute__((packed, aligned(4))) pcm2 { float x, y; };

__device__ __forceinline__ float2 cmul(float2 x, float2 w) {        // FE-1 generic complex multiply
    float2 y;
    y.x = __builtin_fmaf(-x.y, w.y, x.x * w.x);
    y.y = __builtin_fmaf(x.y, w.x, x.x * w.y);
    return y;
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 t) { return make_float2(t.y, -t.x); }                 // * (-i)
__device__ __forceinline__ float2 mul_w8(float2 t) {                                                   // * W8
    const float s = 0.70710678118654752440f;
    return make_float2(s * (t.x + t.y), s * (t.y - t.x));
}
__device__ __forceinline__ float2 mul_w83(float2 t) {                                                  // * W8^3
    const float s = 0.70710678118654752440f;
    return make_float2(s * (t.y - t.x), -(s * (t.x + t.y)));
}

// radix-8 DIF butterfly = three radix-2 stages (FE-1); result returned in natural order.
// NZ = number of leading non-zero inputs (inputs >= NZ are exactly zero and pruned).
template <int NZ>
__device__ __forceinline__ void radix8(float2 (&v)[8]) {
    float2 a0, a1, a2, a3, b0, b1, b2, b3;
    if (NZ > 4) {
        a0 = cadd(v[0], v[4]); b0 = csub(v[0], v[4]);
        a1 = cadd(v[1], v[5]); b1 = mul_w8(csub(v[1], v[5]));
        a2 = cadd(v[2], v[6]); b2 = mul_mi(csub(v[2], v[6]));
        a3 = cadd(v[3], v[7]); b3 = mul_w83(csub(v[3], v[7]));
    } else {           // v[4..7] == 0: u + 0 = u, (u - 0) * W = u * W
        a0 = v[0]; b0 = v[0];
        a1 = v[1]; b1 = mul_w8(v[1]);
        a2 = v[2]; b2 = mul_mi(v[2]);
        a3 = v[3]; b3 = mul_w83(v[3]);
    }
    // stage h = 2 on (a0..a3) and (b0..b3)
    float2 c0 = cadd(a0, a2), c2 = csub(a0, a2);
    float2 c1 = cadd(a1, a3), c3 = mul_mi(csub(a1, a3));
    float2 d0 = cadd(b0, b2), d2 = csub(b0, b2);
    float2 d1 = cadd(b1, b3), d3 = mul_mi(csub(b1, b3));
    // stage h = 1; bit-reversed positions -> natural order: out[k] = pos[bitrev3(k)]
    v[0] = cadd(c0, c1); v[4] = csub(c0, c1);      // pos 0,1 -> k 0,4
    v[2] = cadd(c2, c3); v[6] = csub(c2, c3);      // pos 2,3 -> k 2,6
    v[1] = cadd(d0, d1); v[5] = csub(d0, d1);      // pos 4,5 -> k 1,5
    v[3] = cadd(d2, d3); v[7] = csub(d2, d3);      // pos 6,7 -> k 3,7
}

// lanes of one wave exchange through LDS: hardware runs a wave's LDS instructions in order, so only
// the compiler has to be kept from moving accesses across (a fence would also drain the PCM prefetch)
__device__ __forceinline__ void wave_lds_sync() { wsync(); }

__device__ __forceinline__ uint32_t to_u32(float x) {               // FE-1 F8: trunc, saturate, NaN -> 0
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

constexpr int XROW = 72;                  // float2 row stride of the transpose buffer
constexpr int MELW = 12;                  // mel taps per band kept in registers (wider bands take the LDS loop)
constexpr int XBUF = 8 * XROW;            // float2 per wave

template <int AZ>
__global__ __launch_bounds__(256) void fe_kernel_r8(FeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int clip = blockIdx.y;
    // ---- LDS carve-up: [mel_w | mel_k0 | mel_cnt | mel_off | emph] shared, then per wave X + P
    float* s_melw = reinterpret_cast<float*>(smem);
    int* s_k0 = reinterpret_cast<int*>(s_melw + ((p.mel_total + 3) & ~3));
    int* s_cnt = s_k0 + p.bands;
    int* s_off = s_cnt + p.bands;
    float* s_emph = reinterpret_cast<float*>(s_off + p.bands);
    const int shared_words = ((p.mel_total + 3) & ~3) + 4 * p.bands;
    const int pstride = (p.kmax + 1 + 3) & ~3;
    float2* X = reinterpret_cast<float2*>(smem + (size_t)((shared_words + 3) & ~3) * 4) + (size_t)wave * XBUF;
    float* P = reinterpret_cast<float*>(reinterpret_cast<float2*>(smem + (size_t)((shared_words + 3) & ~3) * 4) + 4 * XBUF) + (size_t)wave * pstride;

    for (int i = threadIdx.x; i < p.mel_total; i += 256) s_melw[i] = p.mel_w[i];
    for (int i = threadIdx.x; i < p.bands; i += 256) {
        s_emph[i] = p.emph[i];
        if (p.spec_type == 1) { s_k0[i] = p.mel_k0[i]; s_cnt[i] = p.mel_cnt[i]; s_off[i] = p.mel_off[i]; }
    }
    __syncthreads();

    const uint32_t nfr = p.n_frames[clip];
    const uint32_t f_begin = (uint32_t)(blockIdx.x * 4 + wave) * (uint32_t)p.frames_per_wave;
    if (f_begin >= nfr) return;
    uint32_t f_end = f_begin + (uint32_t)p.frames_per_wave;
    if (f_end > nfr) f_end = nfr;

    // ---- loop-invariant per-lane constants (registers)
    float2 tw1[8], tw2[8];
#pragma unroll
    for (int k = 1; k < 8; k++) { tw1[k] = p.tw_n2[lane * k]; tw2[k] = p.tw_64[(lane & 7) * k]; }
    float w0[AZ], w1[AZ];
#pragma unroll
    for (int a = 0; a < AZ; a++) {
        const int n = 2 * (64 * a + lane);
        w0[a] = n < p.win ? p.window[n] : 0.0f;
        w1[a] = n + 1 < p.win ? p.window[n + 1] : 0.0f;
    }
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const int k0 = hi3 + 8 * lo3;                         // this lane ends up holding Z[k0 + 64 c']
    const int k0p = (64 - k0) & 63;
    const int partner = ((k0p & 7) << 3) | (k0p >> 3);    // lane holding Z[k0p + 64 c']
    const int nrow = p.kmax / 64 + 1;                     // rows c' with some k <= kmax
    float2 tws[9];
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const int k = k0 + 64 * c;
        tws[c] = (k <= p.kmax) ? p.tw_nfft[k] : make_float2(0.f, 0.f);
    }

    // mel taps of this lane's two bands (m = lane, lane + 64): loop invariant, zero padded.  A padded
    // tap contributes fmaf(0, P, e) = e exactly, so the fixed-length chain equals the FE-1 chain.
    float mw[2][MELW]; int mk[2], mn[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int m = lane + 64 * q;
        mk[q] = 0; mn[q] = 0;
#pragma unroll
        for (int j = 0; j < MELW; j++) mw[q][j] = 0.f;
        if (p.spec_type == 1 && m < p.bands) {
            mk[q] = s_k0[m]; mn[q] = s_cnt[m];
#pragma unroll
            for (int j = 0; j < MELW; j++) if (j < mn[q]) mw[q][j] = s_melw[s_off[m] + j];
        }
    }
    const bool mel_fast = p.spec_type == 1 && p.bands <= 128 && __all(mn[0] <= MELW && mn[1] <= MELW);
    const int pmax = p.kmax;                                    // padded taps read a valid P slot

    const float* clip_pcm = p.pcm + (uint64_t)clip * p.clip_stride;
    uint32_t* out_base = p.spec + (uint64_t)p.frame_off[clip] * (uint32_t)p.bands;

    // PCM of frame f+1 is requested before frame f is transformed (lane m takes complex points 64a + m)
    auto load_pcm = [&](uint32_t f, float2 (&x)[AZ]) __attribute__((always_inline)) {
        const float* fr = clip_pcm + (uint64_t)f * (uint32_t)p.hop;
#pragma unroll
        for (int a = 0; a < AZ; a++) {
            const int n = 2 * (64 * a + lane);
            float x0 = 0.f, x1 = 0.f;
            if (n + 1 < p.win) { const pcm2 q = *reinterpret_cast<const pcm2*>(fr + n); x0 = q.x; x1 = q.y; }
            else if (n < p.win) x0 = fr[n];
            x[a] = make_float2(x0, x1);
        }
    };
    float2 xin[AZ];
    load_pcm(f_begin, xin);

    for (uint32_t f = f_begin; f < f_end; f++) {
        // ---- window (F1-F3)
        float2 v[8];
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = make_float2(0.f, 0.f);
#pragma unroll
        for (int a = 0; a < AZ; a++) v[a] = make_float2(xin[a].x * w0[a], xin[a].y * w1[a]);
        if (f + 1 < f_end) load_pcm(f + 1, xin);
        // ---- pass 1: radix 8 over a, twiddle W_512^{m a'}
        radix8<AZ>(v);
#pragma unroll
        for (int k = 1; k < 8; k++) v[k] = cmul(v[k], tw1[k]);
        // ---- X1: [a'][m] -> lane (a', c) reads b = 0..7
#pragma unroll
        for (int k = 0; k < 8; k++) X[k * XROW + lane] = v[k];
        wave_lds_sync();
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = X[hi3 * XROW + 8 * b + lo3];
        wave_lds_sync();
        // ---- pass 2: radix 8 over b, twiddle W_64^{c b'}
        radix8<8>(v);
#pragma unroll
        for (int k = 1; k < 8; k++) v[k] = cmul(v[k], tw2[k]);
        // ---- X2: [a'][b'][c] (row stride 9) -> lane (a', b') reads c = 0..7
#pragma unroll
        for (int k = 0; k < 8; k++) X[hi3 * XROW + k * 9 + lo3] = v[k];
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = X[hi3 * XROW + lo3 * 9 + c];
        wave_lds_sync();
        // ---- pass 3: radix 8 over c -> v[c'] = Z[k0 + 64 c']
        radix8<8>(v);
        // ---- real-FFT split + 4x power (F4): X[k] from Z[k] and conj(Z[512 - k])
#pragma unroll
        for (int c = 0; c < 9; c++) {
            if (c < nrow) {
                // partner value Z[512 - k]: general lanes: partner lane's register 7 - c;
                // the k0 == 0 lane pairs with itself: register (8 - c) & 7
                float2 zb;
                if (c < 8) {
                    const float2 src = v[7 - c];
                    zb.x = __shfl(src.x, partner, 64);
                    zb.y = __shfl(src.y, partner, 64);
                } else zb = make_float2(0.f, 0.f);
                if (k0 == 0) zb = v[(8 - c) & 7];
                const float2 za = v[c & 7];                // c == 8 only for k0 == 0: Z[512] = Z[0]
                const float2 bb = make_float2(zb.x, -zb.y);
                const float2 e = cadd(za, bb), o = csub(za, bb);
                const float2 t = cmul(o, tws[c]);
                const float xr = e.x + t.y, xi = e.y - t.x;
                const int k = k0 + 64 * c;
                if (k <= p.kmax) P[k] = __builtin_fmaf(xr, xr, xi * xi);
            }
        }
        wave_lds_sync();
        // ---- bands (F5-F8)
        uint32_t* out = out_base + (uint64_t)f * (uint32_t)p.bands;
        if (mel_fast) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int m = lane + 64 * q;
                float pv[MELW];
#pragma unroll
                for (int j = 0; j < MELW; j++) { const int k = mk[q] + j; pv[j] = P[k <= pmax ? k : pmax]; }
                float e = 0.f;
#pragma unroll
                for (int j = 0; j < MELW; j++) e = __builtin_fmaf(mw[q][j], pv[j], e);
                e = e * s_emph[m < p.bands ? m : 0];
                e = e * p.gain;
                if (m < p.bands) out[m] = to_u32(e);
            }
        } else
        for (int m = lane; m < p.bands; m += 64) {
            float e;
            if (p.spec_type == 1) {
                e = 0.f;
                const int kb = s_k0[m], n = s_cnt[m];
                const float* w = s_melw + s_off[m];
                for (int j = 0; j < n; j++) e = __builtin_fmaf(w[j], P[kb + j], e);
            } else {
                e = 0.25f * P[m];
                if (p.spec_type == 3) e = __fsqrt_rn(e);
            }
            e = e * s_emph[m];
            e = e * p.gain;
            out[m] = to_u32(e);
        }
        wave_lds_sync();
    }
}

size_t fe_lds_bytes(const FeParams& p) {
    const size_t shared_words = (size_t)((p.mel_total + 3) & ~3) + 4 * (size_t)p.bands;
    const size_t pstride = (size_t)((p.kmax + 1 + 3) & ~3);
    return ((shared_words + 3) & ~(size_t)3) * 4 + 4 * XBUF * sizeof(float2) + 4 * pstride * 4;
}

void launch_frontend(const FeParams& p, int n_clips, int max_frames, int R, hipStream_t s) {
    (void)R;
    if (n_clips <= 0 || max_frames <= 0) return;
    const int frames_per_block = 4 * p.frames_per_wave;
    dim3 grid((max_frames + frames_per_block - 1) / frames_per_block, n_clips, 1);
    const size_t lds = fe_lds_bytes(p);
    const int az = (p.win + 127) / 128;       // non-zero 64-point blocks of packed input
    if (az <= 2) hipLaunchKernelGGL(fe_kernel_r8<2>, grid, dim3(256), lds, s, p);
    else if (az <= 4) hipLaunchKernelGGL(fe_kernel_r8<4>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(fe_kernel_r8<8>, grid, dim3(256), lds, s, p);
}

}  // namespace wsa
