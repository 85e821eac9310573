// Check of the wavefront biquad scan (wave_common.hpp WaveDoFilters) against the serial recurrence, on a full machine:
// every wavefront of a grid of 256-thread workgroups filters its own copy of the same signal, 40 times over, and -- mode 1 --
// every second workgroup of a CU keeps the matrix pipe busy meanwhile.  Built with hipcc's defaults the scan comes out wrong
// in mode 1 only (the SLP vectoriser's v_pk_*_f32 ... op_sel:[0,1,..], see tools/ubench_pk_opsel.hip); built with the
// Makefile's flags (-fno-slp-vectorize -mllvm -disable-vector-combine) it is right in every mode.
//   ubench_bqscan [mode] [code object] [kernel symbol]     mode 0: all scan, 1: MFMA partner, 2: VALU partner
#include "../openal-soft_amd/csrc/wave_common.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
using namespace oalgpu;
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(const float *in, float *out, const BiquadState *f, BiquadState *fout, uint32_t n, uint32_t reps, uint32_t mode)
{
    __shared__ float xs[4][1100];
    __shared__ float ballast[13000];            // 52 KB more: two workgroups per CU
    if(n == 0xffffffffu) ballast[threadIdx.x] = 1.0f;
    if(((blockIdx.x >> 8) & 1u) && mode != 0u)
    {   // the co-resident workgroup keeps the matrix pipe busy
        h8v a, b; f4v c = {0, 0, 0, 0};
        for(int i = 0; i < 8; ++i) { a[i] = _Float16(float(threadIdx.x + i) * 1e-3f); b[i] = _Float16(float(threadIdx.x * 3 + i) * 1e-3f); }
        if(mode == 1u)
        for(uint32_t r = 0; r < reps * 600u; ++r)
        {
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0);
        }
        else
        for(uint32_t r = 0; r < reps * 2400u; ++r)
        {   // plain VALU work instead
            c = __builtin_elementwise_fma(c, c, c);
            c = __builtin_elementwise_fma(c, c, c);
        }
        if(c[0] == 12345.0f) out[0] = c[1];
        return;
    }
    __shared__ float fsts[4][32];
    __shared__ BiquadSlot slotss[4][2];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    float *x = xs[wv]; float *fst = fsts[wv]; BiquadSlot *slots = slotss[wv];
    uint32_t h0 = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    uint64_t tf = 0;
    for(uint32_t r = 0; r < reps; ++r)
    {
        for(uint32_t i = lane; i < n; i += 64) x[i] = in[i];
        if(lane < 2) { slots[lane].f = f[lane]; }
        WaveSync();
        if(lane < 32) fst[lane] = reinterpret_cast<const float*>(&slots[lane >> 4])[lane & 15];
        WaveSync();
        const uint64_t ta = __builtin_readcyclecounter();
        WaveDoFilters(fst, slots, true, x, n, lane);
        WaveSync();
        tf += __builtin_readcyclecounter() - ta;
        uint32_t h = 0;
        for(uint32_t i = lane; i < n; i += 64) h ^= __builtin_bit_cast(uint32_t, x[i]) * (i + 1u);
        h ^= __builtin_bit_cast(uint32_t, slots[1].f.z1) ^ __builtin_bit_cast(uint32_t, slots[0].f.z2);
        if(r == 0) h0 = h;
        else if(h != h0) atomicAdd(reinterpret_cast<uint32_t*>(fout + size_t{gridDim.x} * 8), 1u);
        WaveSync();
    }
    const size_t w = size_t{blockIdx.x} * 4 + wv;
    if(lane == 0) { fout[size_t{gridDim.x} * 8 + 1 + w].z1 = float(tf) / float(reps); fout[size_t{gridDim.x} * 8 + 1 + w].z2 = float(__builtin_readcyclecounter() - t0) / float(reps); }
    for(uint32_t i = lane; i < n; i += 64) out[w * 1024 + i] = x[i];
    if(lane < 2) { fout[w * 2 + lane] = slots[lane].f; }
}
int main(int argc, char **argv)
{
    const uint32_t mode = argc > 1 ? uint32_t(atoi(argv[1])) : 1u;
    printf("mode %u (0: every workgroup scans, 1: half of them run MFMAs, 2: half of them run VALU work)\n", mode);
    const uint32_t G = 512;
    for(uint32_t n : {1024u, 700u, 37u, 1000u})
    {
        std::vector<float> x(n); srand(n);
        for(auto &v : x) v = float(rand()) / RAND_MAX * 2 - 1;
        BiquadState f[2]{};
        f[0].b0 = 0.8f; f[0].b1 = -1.2f; f[0].b2 = 0.5f; f[0].a1 = -1.3f; f[0].a2 = 0.55f; f[0].z1 = 0.3f; f[0].z2 = -0.2f;
        f[1].b0 = 1.1f; f[1].b1 = -0.4f; f[1].b2 = 0.1f; f[1].a1 = -0.5f; f[1].a2 = 0.2f; f[1].z1 = -0.1f; f[1].z2 = 0.05f;
        std::vector<double> y(n);
        double s[2][2] = {{f[0].z1, f[0].z2}, {f[1].z1, f[1].z2}};
        for(uint32_t i = 0; i < n; ++i)
        {
            double v = x[i];
            for(int q = 0; q < 2; ++q)
            {
                const double o = v * f[q].b0 + s[q][0];
                s[q][0] = v * f[q].b1 - o * f[q].a1 + s[q][1];
                s[q][1] = v * f[q].b2 - o * f[q].a2;
                v = o;
            }
            y[i] = v;
        }
        float *dx, *dy; BiquadState *df, *dfo;
        (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dy, size_t{G} * 4 * 1024 * 4); (void)hipMalloc(&df, sizeof(f)); const size_t dfoBytes = sizeof(BiquadState) * (size_t{G} * 8 + 2 + size_t{G} * 4); (void)hipMalloc(&dfo, dfoBytes); (void)hipMemset(dfo, 0, dfoBytes);
        (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(df, f, sizeof(f), hipMemcpyHostToDevice);
        if(argc > 2)
        {   // the kernel out of a separately assembled code object (hand-edited ISA)
            hipModule_t mod; hipFunction_t fn;
            if(hipModuleLoad(&mod, argv[2]) != hipSuccess || hipModuleGetFunction(&fn, mod, argv[3]) != hipSuccess) { printf("cannot load %s\n", argv[2]); return 1; }
            uint32_t reps = 40, nn = n, md = mode;
            void *args[] = {&dx, &dy, &df, &dfo, &nn, &reps, &md};
            if(hipModuleLaunchKernel(fn, G, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr) != hipSuccess) { printf("launch failed\n"); return 1; }
        }
        else k<<<G, 256>>>(dx, dy, df, dfo, n, 40, mode); (void)hipDeviceSynchronize();
        std::vector<float> g(size_t{G} * 4 * 1024); std::vector<BiquadState> fo(G * 4 * 2);
        (void)hipMemcpy(g.data(), dy, g.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(fo.data(), dfo, fo.size() * sizeof(BiquadState), hipMemcpyDeviceToHost);
        double err = 0, mx = 0, zerr = 0; size_t bad = 0;
        for(size_t w = 0; w < size_t{G} * 4; ++w)
        {
            if((((w / 4) >> 8) & 1u) && mode) continue;
            double we = 0;
            for(uint32_t i = 0; i < n; ++i) { const double e = fabs(g[w * 1024 + i] - y[i]); we = fmax(we, e); mx = fmax(mx, fabs(y[i])); }
            zerr = fmax(zerr, fmax(fmax(fabs(fo[w * 2].z1 - s[0][0]), fabs(fo[w * 2].z2 - s[0][1])), fmax(fabs(fo[w * 2 + 1].z1 - s[1][0]), fabs(fo[w * 2 + 1].z2 - s[1][1]))));
            if(we > 1e-5) { if(bad < 4) printf("  wave %zu (wg %zu) err %.3e\n", w, w / 4, we); ++bad; }
            err = fmax(err, we);
        }
        uint32_t mism = 0; (void)hipMemcpy(&mism, reinterpret_cast<char*>(dfo) + sizeof(BiquadState) * G * 8, 4, hipMemcpyDeviceToHost);
        { std::vector<BiquadState> tm(size_t{G} * 4); (void)hipMemcpy(tm.data(), dfo + size_t{G} * 8 + 1, tm.size() * sizeof(BiquadState), hipMemcpyDeviceToHost);
          double a = 0; size_t c = 0; for(size_t w = 0; w < tm.size(); ++w) if(tm[w].z1 > 0) { a += tm[w].z1; ++c; }
          printf("filter pair over %u samples: %.0f cycles (s_memtime) per call, mean of %zu wavefronts\n", n, c ? a / c : 0.0, c); }
        printf("repetitions that differ from the first: %u lanes\n", mism);
        printf("n=%u: %u waves, max err %.3e of max %.3e, state err %.3e, bad waves %zu\n", n, G * 4, err, mx, zerr, bad);
    }
    return 0;
}
