// Dependent-load latency as the tracker sees it: W waves, each walking its own run of 1552-byte frame records
// (one 16-byte load per lane of the first 8 lanes + one header dword), every load address depending on the data before.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/stride_latency tools/microbench/stride_latency.hip ; run: /tmp/stride_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void chase(const uint32_t* rec, uint32_t stride_words, uint32_t frames_per_wave, uint32_t iters, unsigned long long* cyc, uint32_t* sink) {
    const uint32_t w = blockIdx.x;
    const int lane = threadIdx.x;
    uint32_t f = w * frames_per_wave, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < iters; i++) {
        const uint32_t* r = rec + (uint64_t)(f + (i % frames_per_wave)) * stride_words + (acc & 1u);   // acc & 1 is 0: data are zeros
        uint32_t x = 0;
        if (lane < 10) x = r[4 + 6 * lane];
        acc += __builtin_amdgcn_readfirstlane(x) + r[2];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { cyc[w] = t1 - t0; sink[w] = acc; }
}
int main(int argc, char** argv) {
    const uint32_t frames = 409600, stride = 388;
    uint32_t* rec; hipMalloc(&rec, (size_t)frames * stride * 4); hipMemset(rec, 0, (size_t)frames * stride * 4);
    for (int waves : {256, 1024, 3072, 8192}) {
        unsigned long long* cyc; uint32_t* sink; hipMalloc(&cyc, waves * 8); hipMalloc(&sink, waves * 4);
        const uint32_t fpw = frames / waves, iters = fpw < 200 ? fpw : 200;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        chase<<<waves, 64>>>(rec, stride, fpw, iters, cyc, sink);
        hipEventRecord(e0); chase<<<waves, 64>>>(rec, stride, fpw, iters, cyc, sink); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(waves); hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto c : h) s += (double)c;
        printf("waves %5d: %.0f cycles per dependent load (s_memtime), kernel %.3f ms -> %.2f us per load\n", waves, s / waves / iters, ms, ms * 1e3 / iters);
        hipFree(cyc); hipFree(sink);
    }
    return 0;
}
