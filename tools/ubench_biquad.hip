// Micro-benchmark: the serial dual-biquad recurrence (BiquadFilter::dualProcess order, no
// contraction) on 4 lanes of one wavefront over a 256-sample LDS row -- scalar vs packed/skewed.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ubench_biquad ubench_biquad.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kRow = 260, kN = 256;

template<int MODE>
__global__ void Bench(float *out, unsigned long long *ticks, const float *coef)
{
    __shared__ float rows[4 * kRow];
    const unsigned lane = threadIdx.x;
    for(unsigned k = lane; k < 4 * kRow; k += 64) rows[k] = 0.001f * float(k % 97);
    __syncthreads();
    const float b00 = coef[0], b01 = coef[1], b02 = coef[2], a01 = coef[3], a02 = coef[4];
    const float b10 = coef[5], b11 = coef[6], b12 = coef[7], a11 = coef[8], a12 = coef[9];
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long c0 = wall_clock64();
    for(int rep = 0; rep < 16; ++rep)
    {
        if(lane < 4)
        {
            float *row = rows + lane * kRow;
            if(MODE == 0)
            {
                float z01 = 0, z02 = 0, z11 = 0, z12 = 0;
#pragma unroll 8
                for(int i = 0; i < kN; ++i)
                {
                    const float x0 = row[i];
                    const float y0 = x0 * b00 + z01;
                    z01 = x0 * b01 - y0 * a01 + z02;
                    z02 = x0 * b02 - y0 * a02;
                    const float y1 = y0 * b10 + z11;
                    z11 = y0 * b11 - y1 * a11 + z12;
                    z12 = y0 * b12 - y1 * a12;
                    row[i] = y1;
                }
            }
            else if(MODE == 1)
            {
                f2 Z1 = {0, 0}, Z2 = {0, 0};
                const f2 B0 = {b00, b10}, B1 = {b01, b11}, B2 = {b02, b12}, A1 = {a01, a11}, A2 = {a02, a12};
                float yprev = 0;
#pragma unroll 8
                for(int i = 1; i < kN; ++i)
                {
                    const f2 X = {row[i], yprev};
                    const f2 Y = X * B0 + Z1;
                    Z1 = X * B1 - Y * A1 + Z2;
                    Z2 = X * B2 - Y * A2;
                    row[i - 1] = Y.y;
                    yprev = Y.x;
                }
            }
            else
            {   // skewed, scalar instructions (two independent chains interleaved)
                float z01 = 0, z02 = 0, z11 = 0, z12 = 0, yprev = 0;
#pragma unroll 8
                for(int i = 1; i < kN; ++i)
                {
                    const float x0 = row[i];
                    const float y0 = x0 * b00 + z01;
                    const float y1 = yprev * b10 + z11;
                    z01 = x0 * b01 - y0 * a01 + z02;
                    z11 = yprev * b11 - y1 * a11 + z12;
                    z02 = x0 * b02 - y0 * a02;
                    z12 = yprev * b12 - y1 * a12;
                    row[i - 1] = y1;
                    yprev = y0;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long c1 = wall_clock64();
    if(lane == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
    out[lane] = rows[lane];
}

int main()
{
    float *out, *coef; unsigned long long *ticks;
    hipMalloc(&out, 256); hipMalloc(&coef, 64); hipMalloc(&ticks, 16);
    const float hc[10] = {1.01f, -1.9f, 0.92f, -1.89f, 0.91f, 0.99f, -1.8f, 0.85f, -1.79f, 0.84f};
    hipMemcpy(coef, hc, sizeof(hc), hipMemcpyHostToDevice);
    for(int pass = 0; pass < 2; ++pass)
        for(int mode = 0; mode < 3; ++mode)
        {
            if(mode == 0) hipLaunchKernelGGL(Bench<0>, dim3(1), dim3(64), 0, 0, out, ticks, coef);
            if(mode == 1) hipLaunchKernelGGL(Bench<1>, dim3(1), dim3(64), 0, 0, out, ticks, coef);
            if(mode == 2) hipLaunchKernelGGL(Bench<2>, dim3(1), dim3(64), 0, 0, out, ticks, coef);
            unsigned long long h[2];
            hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
            printf("mode %d: %.1f counter ticks/sample, %.2f ns/sample (wall clock 100 MHz)\n", mode,
                double(h[0]) / (16.0 * kN), double(h[1]) * 10.0 / (16.0 * kN));
        }
    return 0;
}
