// Probe: does ds_read_b64 at a 4-byte-aligned (not 8-byte-aligned) LDS address return the right data on
// gfx950, and what does it cost next to ds_read2_b32 (the resampler's source-pair reads)?  Addresses follow
// the resampler's pattern: lane l reads sample pairs starting at floor(l * 0.92) + 2 q.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template<int MODE>
__global__ void __launch_bounds__(512) probe(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float s[4096];
    for(int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = float(i);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)(uintptr_t)(s) + 4u * ((lane * 60211u) >> 16) + 512u * wave;   // bytes
    f2 acc = {0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for(int it = 0; it < iters; ++it)
    {
#pragma unroll
        for(int q = 0; q < 12; ++q)
        {
            f2 v;
            if constexpr (MODE == 0) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(2 * q), "n"(2 * q + 1));
            else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(8 * q));
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            acc += v;
        }
        addr ^= (it & 1) ? 4u : 0u;      // alternate parity
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + 2.0f * acc.y;
    if(threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template<int MODE> void run(const char *name)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    probe<MODE><<<256, 512>>>(out, cyc, iters); hipDeviceSynchronize();
    std::vector<unsigned long long> c(256); std::vector<float> o(512);
    hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(o.data(), out, 512 * 4, hipMemcpyDeviceToHost);
    double m = 0; for(auto x : c) m += double(x); m /= 256;
    printf("%s: %.1f cycles per wave-instruction-slot (8 waves/CU, %d x 12 reads each): %.2f LDS cycles per read per CU; check %g %g %g\n", name,
        m / (iters * 12.0), iters, m / (iters * 12.0) / 8.0, o[0], o[1], o[3]);
}
int main() { run<0>("ds_read2_b32 (offset, offset+1)"); run<1>("ds_read_b64 at 4-byte alignment"); run<0>("ds_read2_b32 again"); run<1>("ds_read_b64 again"); return 0; }
