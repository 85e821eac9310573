// Measurement aid: can the host store straight into device memory on this box (large BAR), and what does a one-wavefront
// kernel's first dependent read cost from (a) pinned host memory, (b) fine-grained device memory the host wrote?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_largebar tools/ubench_largebar.hip && /tmp/ubench_largebar
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)
__global__ void ReadKernel(const uint32_t *src, uint32_t *dst, uint32_t n)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if(i < n) dst[i] = src[src[i] % n] + 1u;          // two dependent reads of the source
}
int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    std::printf("isLargeBar %d\n", p.isLargeBar);
    const uint32_t n = 1024 * 6;                       // 24 KB, the size of a move block
    uint32_t *pinned, *fine = nullptr, *dst; CK(hipHostMalloc((void**)&pinned, n * 4, hipHostMallocDefault));
    hipError_t e = hipExtMallocWithFlags((void**)&fine, n * 4, hipDeviceMallocFinegrained);
    std::printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    CK(hipMalloc((void**)&dst, n * 4));
    std::vector<uint32_t> host(n), back(n);
    for(uint32_t i = 0; i < n; ++i) host[i] = i * 7u + 3u;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for(int which = 0; which < 2; ++which)
    {
        uint32_t *src = which ? fine : pinned;
        if(!src) continue;
        bool ok = true; double hostUs = 0; float kernMs = 0;
        for(int rep = 0; rep < 50; ++rep)
        {
            for(uint32_t i = 0; i < n; ++i) host[i] = i * 7u + 3u + uint32_t(rep);
            const auto t0 = std::chrono::steady_clock::now();
            std::memcpy(src, host.data(), n * 4);      // (which = 1: CPU stores through the BAR)
            hostUs += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            hipExtLaunchKernelGGL(ReadKernel, dim3(n / 64), dim3(64), 0, s, a, b, 0u, src, dst, n);
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); kernMs += ms;
            CK(hipMemcpy(back.data(), dst, n * 4, hipMemcpyDeviceToHost));
            for(uint32_t i = 0; i < n; ++i) if(back[i] != host[host[i] % n] + 1u) ok = false;
        }
        std::printf("%s: results %s, host copy %.2f us, kernel %.2f us\n", which ? "fine-grained device memory" : "pinned host memory",
            ok ? "ok" : "WRONG", hostUs / 50, kernMs / 50 * 1e3);
    }
    return 0;
}
