// Numerics probe: what v_mfma_f32_16x16x32_f16 does with products and their sum (MI355X).
// D = A x B (+0) against the exact double-precision result of the same f16 inputs, and the split-half
// product scheme of FirMfmaH (hi*hi + hi*lo + lo*hi) against the double product of the f32 inputs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void one(const _Float16 *A /*16x32 row major*/, const _Float16 *B /*32x16 row major*/, float *D /*16x16*/)
{
    const unsigned l = threadIdx.x, i = l & 15, g = l >> 4;
    h8 a, b;
    for(int k = 0; k < 8; ++k) { a[k] = A[i * 32 + 8 * g + k]; b[k] = B[(8 * g + k) * 16 + i]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for(int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}
// split scheme: Af/Bf f32; out = (hi*hi + hi*lo + lo*hi)
__global__ void split(const float *Af, const float *Bf, float *D, float *Dparts)
{
    const unsigned l = threadIdx.x, i = l & 15, g = l >> 4;
    h8 ah, al, bh, bl;
    for(int k = 0; k < 8; ++k)
    {
        const float a = Af[i * 32 + 8 * g + k], b = Bf[(8 * g + k) * 16 + i];
        auto pa = __builtin_amdgcn_cvt_pkrtz(a, 0.0f); auto pb = __builtin_amdgcn_cvt_pkrtz(b, 0.0f);
        ah[k] = pa[0]; bh[k] = pb[0];
        al[k] = __builtin_amdgcn_cvt_pkrtz(a - float(pa[0]), 0.0f)[0];
        bl[k] = __builtin_amdgcn_cvt_pkrtz(b - float(pb[0]), 0.0f)[0];
    }
    f4 z = {0, 0, 0, 0};
    f4 ta = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, z, 0, 0, 0);
    f4 tb = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, z, 0, 0, 0);
    f4 tc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, z, 0, 0, 0);
    for(int r = 0; r < 4; ++r)
    {
        D[(4 * g + r) * 16 + i] = (ta[r] + tb[r]) + tc[r];
        Dparts[((4 * g + r) * 16 + i) * 3 + 0] = ta[r]; Dparts[((4 * g + r) * 16 + i) * 3 + 1] = tb[r]; Dparts[((4 * g + r) * 16 + i) * 3 + 2] = tc[r];
    }
}
int main()
{
    std::vector<_Float16> A(512), B(512); std::vector<float> Af(512), Bf(512);
    srand(1);
    auto rnd = [] { return float(rand()) / RAND_MAX * 2.0f - 1.0f; };
    for(int t = 0; t < 512; ++t) { Af[t] = rnd() * 20000.0f; Bf[t] = rnd() * 20000.0f * expf(-float(t / 16) / 6.0f); A[t] = _Float16(Af[t]); B[t] = _Float16(Bf[t]); }
    _Float16 *dA, *dB; float *dD, *dAf, *dBf, *dP;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024); hipMalloc(&dAf, 2048); hipMalloc(&dBf, 2048); hipMalloc(&dP, 3072);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dAf, Af.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dBf, Bf.data(), 2048, hipMemcpyHostToDevice);
    std::vector<float> D(256), P(768);
    one<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double worst = 0, mx = 0;
    for(int i = 0; i < 16; ++i) for(int j = 0; j < 16; ++j)
    {
        double s = 0; for(int k = 0; k < 32; ++k) s += double(float(A[i * 32 + k])) * double(float(B[k * 16 + j]));
        worst = fmax(worst, fabs(s - D[i * 16 + j])); mx = fmax(mx, fabs(s));
    }
    printf("single MFMA, f16 inputs: max |err| %.3e of max |D| %.3e = %.3e relative\n", worst, mx, worst / mx);
    split<<<1, 64>>>(dAf, dBf, dD, dP); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(P.data(), dP, 3072, hipMemcpyDeviceToHost);
    worst = 0; mx = 0; double worstc = 0;
    for(int i = 0; i < 16; ++i) for(int j = 0; j < 16; ++j)
    {
        double s = 0, sc = 0;
        for(int k = 0; k < 32; ++k)
        {
            s += double(Af[i * 32 + k]) * double(Bf[k * 16 + j]);
        }
        worst = fmax(worst, fabs(s - D[i * 16 + j])); mx = fmax(mx, fabs(s));
        worstc = fmax(worstc, fabs(s - P[(i * 16 + j) * 3 + 2]));
    }
    printf("split scheme, f32 inputs: max |err| %.3e of max %.3e = %.3e relative; hi*hi alone %.3e relative\n", worst, mx, worst / mx, worstc / mx);
    return 0;
}
