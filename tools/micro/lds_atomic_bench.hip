// Micro-benchmark: LDS f64 atomic add / read / write throughput on gfx950 as a function of same-address fan-in.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/micro/lds_atomic_bench.hip -o /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int fan, int iters, unsigned long long* cyc)
{
    __shared__ double s[4096];
    const int tid = threadIdx.x;
    for (int n = tid; n < 4096; n += 256) s[n] = 0;
    __syncthreads();
    // fan lanes share one address; distinct groups -> distinct consecutive addresses
    const int addr = (tid / fan) % 2048;
    double v = 1.0 + tid, acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) atomicAdd(&s[addr + u * 256], v);
            else if (MODE == 1) acc += s[addr + u * 256];
            else if (MODE == 2) s[(tid + u * 256) % 4096] = v;
            else if (MODE == 3) { float* f = (float*)s; atomicAdd(&f[addr + u * 256], (float)v); }
        }
    }
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + tid] = s[tid] + acc;
}
int main()
{
    double* out; unsigned long long* cyc;
    const int nb = 256 * 2;
    hipMalloc(&out, nb * 256 * 8); hipMalloc(&cyc, nb * 8);
    std::vector<unsigned long long> h(nb);
    const char* names[] = {"ds_add_f64", "ds_read_b64", "ds_write_b64", "ds_add_f32"};
    for (int mode = 0; mode < 4; mode++)
        for (int fan : {1, 2, 4, 8, 16, 32, 64}) {
            const int iters = 200;
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, out, fan, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, out, fan, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, out, fan, iters, cyc);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, out, fan, iters, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
            double m = 0; for (auto x : h) m += x; m /= nb;
            // cycles per wave-instruction seen by one workgroup (4 waves issue 8*iters each; 2 WGs per CU share the LDS)
            printf("%-13s fan-in %2d: %8.1f cycles per wave-instruction (WG wall / (8*iters))\n", names[mode], fan, m / (8.0 * iters));
        }
    return 0;
}
