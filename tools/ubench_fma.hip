// Micro-benchmark: issue rate of v_fmac_f32 (SGPR operand), v_fma_f32, v_pk_fma_f32 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fma.hip -o tools/ubench_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

template<int MODE>
__global__ void __launch_bounds__(256) K(float *out, const float *cin, int iters, long long *cyc)
{
    float a[16];
#pragma unroll
    for(int i = 0; i < 16; ++i) a[i] = float(threadIdx.x + i);
    float w[20];
#pragma unroll
    for(int i = 0; i < 20; ++i) w[i] = float(threadIdx.x * 3 + i) * 1e-3f;
    const float c0 = cin[0], c1 = cin[1], c2 = cin[2], c3 = cin[3];
    long long t0 = __builtin_readcyclecounter();
    for(int it = 0; it < iters; ++it)
    {
        if constexpr(MODE == 0)
        {   // 64 x v_fmac_f32 with scalar coefficient
#pragma unroll
            for(int j = 0; j < 4; ++j)
            {
                const float c = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : c3;
#pragma unroll
                for(int r = 0; r < 16; ++r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[r]) : "s"(c), "v"(w[r + 3 - j]));
            }
        }
        else if constexpr(MODE == 1)
        {   // 32 x v_pk_fma_f32 : aligned pairs, scalar coefficient broadcast
            typedef float float2_ __attribute__((ext_vector_type(2)));
#pragma unroll
            for(int j = 0; j < 4; ++j)
            {
                const float c = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : c3;
                float2_ cc = {c, c};
#pragma unroll
                for(int r = 0; r < 16; r += 2)
                {
                    float2_ acc = {a[r], a[r + 1]};
                    float2_ ww = {w[r + (j & 2)], w[r + 1 + (j & 2)]};
                    acc = __builtin_elementwise_fma(cc, ww, acc);
                    a[r] = acc.x; a[r + 1] = acc.y;
                }
            }
        }
        else
        {   // 64 x v_fma_f32 all-VGPR
#pragma unroll
            for(int j = 0; j < 4; ++j)
            {
#pragma unroll
                for(int r = 0; r < 16; ++r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[r]) : "v"(w[j]), "v"(w[r + 3 - j]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for(int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if(threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template<int MODE>
int run(const char *name, int blocks, int threads, int iters)
{
    float *out, *cin; long long *cyc;
    CHECK(hipMalloc(&out, size_t(blocks) * threads * 4));
    CHECK(hipMalloc(&cin, 16));
    CHECK(hipMalloc(&cyc, 8));
    float h[4] = {0.5f, 0.25f, -0.5f, 0.125f};
    CHECK(hipMemcpy(cin, h, 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(K<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cin, 10, cyc);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL(K<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cin, iters, cyc);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double fmas = double(blocks) * threads * iters * 64.0;
    printf("%-28s blocks %5d x %4d: %8.3f ms  %7.2f TFLOP/s  wave0 clk/iter %.1f (64 lane-FMAs per iter per lane)\n", name, blocks, threads, ms,
        2.0 * fmas / (ms * 1e-3) / 1e12, double(c) / iters);
    hipFree(out); hipFree(cin); hipFree(cyc);
    return 0;
}

int main()
{
    const int iters = 20000;
    for(int wpc : {4, 8, 16})   // waves per CU
    {
        const int threads = 256, blocks = 256 * wpc / 4;
        printf("--- %d waves/CU\n", wpc);
        run<0>("v_fmac_f32 sgpr coef", blocks, threads, iters);
        run<1>("v_pk_fma_f32", blocks, threads, iters);
        run<2>("v_fma_f32 vgpr", blocks, threads, iters);
    }
    return 0;
}
