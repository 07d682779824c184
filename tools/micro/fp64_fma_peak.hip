// Micro-benchmark: FP64 peak of an MI355X as this path can reach it ("verify with an FMA
// micro-benchmark on the box and use the measured number as denominator").
//   * v_fma_f64: NCHAIN independent dependent-chains per lane (no chain limits issue), 1 / 2 / 4 / 8 waves per SIMD, every CU;
//   * v_pk_fma_f32 for orientation (the 157 TFLOP/s row of the guide);
//   * v_mfma_f64_16x16x4_f64: 4 independent accumulator tiles per wave, 1 / 2 / 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/fp64_fma_peak.hip -o tools/micro/fp64_fma_peak
// run:   tools/micro/fp64_fma_peak            (prints one JSON object; joltqc_amd/data/measured_peaks.json keeps the numbers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NCHAIN = 16;      // independent accumulators per lane (FMA latency x issue rate is covered many times over)
constexpr int UNROLL = 8;

__global__ void __launch_bounds__(256) fma_f64(double* out, const int iters, const double a, const double b)
{
    double acc[NCHAIN];
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) acc[c] = threadIdx.x * 1e-9 + c;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int c = 0; c < NCHAIN; c++) acc[c] = __builtin_fma(acc[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) s += acc[c];
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;     // (never true: keeps the chains alive)
}

typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) pk_fma_f32(float* out, const int iters, const float a, const float b)
{
    v2f acc[NCHAIN];
    const v2f va = {a, a}, vb = {b, b};
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) acc[c] = (v2f){threadIdx.x * 1e-6f + c, 1.f + c};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int c = 0; c < NCHAIN; c++) acc[c] = __builtin_elementwise_fma(acc[c], va, vb);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) s += acc[c].x + acc[c].y;
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int NTILE = 4;
__global__ void __launch_bounds__(256) mfma_f64(double* out, const int iters, const double a, const double b)
{
    v4d acc[NTILE];
#pragma unroll
    for (int c = 0; c < NTILE; c++) acc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double va = a + threadIdx.x * 1e-9, vb = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int c = 0; c < NTILE; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, acc[c], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NTILE; c++) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();                                   // warm-up (clocks, code load)
    CHECK(hipDeviceSynchronize());
    std::vector<float> t(reps);
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&t[r], e0, e1));
    }
    std::sort(t.begin(), t.end());
    return t[reps / 2];
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    double* out; CHECK(hipMalloc(&out, 8ull << 20));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", p.name, ncu, p.clockRate / 1000);
    const int iters = 4096;
    // one 256-thread workgroup = one wave per SIMD of a CU; wps workgroups per CU = wps waves per SIMD
    const int wps_list[] = {1, 2, 4, 8};
    double best64 = 0, best32 = 0, bestm = 0;
    printf(", \"v_fma_f64\": {");
    for (int n = 0; n < 4; n++) {
        const int wps = wps_list[n];
        const double ms = time_ms([&] { fma_f64<<<ncu * wps, 256>>>(out, iters, 0.999999, 1e-9); }, 7);
        const double tf = 2.0 * NCHAIN * UNROLL * (double)iters * 256.0 * ncu * wps / (ms * 1e-3) * 1e-12;
        best64 = std::max(best64, tf);
        printf("%s\"%d_waves_per_simd\": %.2f", n ? ", " : "", wps, tf);
    }
    printf("}, \"v_pk_fma_f32\": {");
    for (int n = 0; n < 4; n++) {
        const int wps = wps_list[n];
        const double ms = time_ms([&] { pk_fma_f32<<<ncu * wps, 256>>>((float*)out, iters, 0.999999f, 1e-9f); }, 7);
        const double tf = 4.0 * NCHAIN * UNROLL * (double)iters * 256.0 * ncu * wps / (ms * 1e-3) * 1e-12;
        best32 = std::max(best32, tf);
        printf("%s\"%d_waves_per_simd\": %.2f", n ? ", " : "", wps, tf);
    }
    printf("}, \"v_mfma_f64_16x16x4_f64\": {");
    for (int n = 0; n < 3; n++) {
        const int wps = wps_list[n];
        const double ms = time_ms([&] { mfma_f64<<<ncu * wps, 256>>>(out, iters, 0.5, 1e-9); }, 7);
        // one MFMA 16x16x4 = 2 * 16 * 16 * 4 flop per wave
        const double tf = 2048.0 * NTILE * UNROLL * (double)iters * 4.0 * ncu * wps / (ms * 1e-3) * 1e-12;
        bestm = std::max(bestm, tf);
        printf("%s\"%d_waves_per_simd\": %.2f", n ? ", " : "", wps, tf);
    }
    printf("}, \"fp64_valu_peak_tflops\": %.2f, \"fp32_packed_valu_peak_tflops\": %.2f, \"fp64_mfma_peak_tflops\": %.2f}\n", best64, best32, bestm);
    return 0;
}
