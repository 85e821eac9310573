// The panning half of the parameter stage on the GPU: CalcDirectionCoeffs / CalcAmbiCoeffs
// (core/mixer.h:68-73, core/ambidefs.h:219-271, core/mixer.cpp:16-90) and ComputePanGains (core/mixer.cpp:92-102)
// as CalcPanningAndFilters calls them for a point source (alc/alu.cpp: the dry bus with DryGain.Base, every
// send's slot with WetGain[i].Base) -- so that a moving voice's parameter record is a direction, a spread
// and a few gains instead of up to 32 + 6 x 25 resolved line gains.
//
// The encoder coefficients are the real spherical harmonics up to order 4 in ACN order with N3D normalisation;
// thread = one bus line, which evaluates the ONE coefficient its AmbiMap entry names, in the reference's
// operation order (no contraction), so that a direction without spread yields bit-identical gains.  The
// spread's zonal scaling uses cos and sqrt: within an ulp or two of the host's libm.
#include <hip/hip_runtime.h>
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

// ACN `acn` of CalcAmbiCoeffs(y, z, x) (ambisonic coordinates), core/ambidefs.h:219-271
__device__ __forceinline__ float AmbiCoeff(uint32_t acn, float y, float z, float x)
{
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    switch(acn)
    {
    case 0: return 1.0f;
    case 1: return 1.7320508075688772f * y;
    case 2: return 1.7320508075688772f * z;
    case 3: return 1.7320508075688772f * x;
    case 4: return 3.872983346e+00f * xy;
    case 5: return 3.872983346e+00f * yz;
    case 6: return 1.118033989e+00f * (3.0f * zz - 1.0f);
    case 7: return 3.872983346e+00f * xz;
    case 8: return 1.936491673e+00f * (xx - yy);
    case 9: return 2.091650066e+00f * (y * (3.0f * xx - yy));
    case 10: return 1.024695076e+01f * (z * xy);
    case 11: return 1.620185175e+00f * (y * (5.0f * zz - 1.0f));
    case 12: return 1.322875656e+00f * (z * (5.0f * zz - 3.0f));
    case 13: return 1.620185175e+00f * (x * (5.0f * zz - 1.0f));
    case 14: return 5.123475383e+00f * (z * (xx - yy));
    case 15: return 2.091650066e+00f * (x * (xx - 3.0f * yy));
    case 16: return 8.874119675e+00f * (xy * (xx - yy));
    case 17: return 6.274950199e+00f * ((3.0f * xx - yy) * yz);
    case 18: return 3.354101966e+00f * (xy * (7.0f * zz - 1.0f));
    case 19: return 2.371708245e+00f * (yz * (7.0f * zz - 3.0f));
    case 20: return 3.750000000e-01f * (35.0f * (zz * zz) - 30.0f * zz + 3.0f);
    case 21: return 2.371708245e+00f * (xz * (7.0f * zz - 3.0f));
    case 22: return 1.677050983e+00f * ((xx - yy) * (7.0f * zz - 1.0f));
    case 23: return 6.274950199e+00f * ((xx - 3.0f * yy) * xz);
    default: return 2.218529919e+00f * ((xx * xx) - 6.0f * (xx * yy) + (yy * yy));
    }
}

// the spread's zonal-harmonic factor of ambisonic order `order`, core/mixer.cpp:20-86
__device__ __forceinline__ float SpreadFactor(uint32_t order, float spread)
{
    const float ca = cosf(spread * 0.5f);
    const float scale = sqrtf(1.0f + 0.31830988618379067154f * 0.5f * spread);
    const float caca = ca * ca;
    switch(order)
    {
    case 0: return scale;
    case 1: return scale * 0.5f * (ca + 1.0f);
    case 2: return scale * 0.5f * ((ca + 1.0f) * ca);
    case 3: return scale * 0.125f * ((ca + 1.0f) * (5.0f * caca - 1.0f));
    default: return scale * 0.125f * ((ca + 1.0f) * (7.0f * caca - 3.0f) * ca);
    }
}

// one workgroup per record; thread t < numDry: dry line t; then numSends x wetChannels send lines
__global__ void __launch_bounds__(256) PanGainsKernel(DeviceLayout L, const PanRecord *recs, const AmbiMapEntry *dryMap,
    const AmbiMapEntry *wetMaps /* [slot][wetChannels] */)
{
    const PanRecord r = recs[blockIdx.x];
    const uint32_t v = r.voice, t = threadIdx.x;
    // CalcDirectionCoeffs: OpenAL -> ambisonic coordinates (core/mixer.h:71-72)
    const float y = -r.dir[0], z = r.dir[1], x = -r.dir[2];
    auto gainFor = [&](const AmbiMapEntry &m, float ingain)
    {
        float c = AmbiCoeff(m.index, y, z, x);
        if(r.spread > 0.0f)
        {
            const uint32_t order = m.index == 0 ? 0u : (m.index < 4 ? 1u : (m.index < 9 ? 2u : (m.index < 16 ? 3u : 4u)));
            c *= SpreadFactor(order, r.spread);
        }
        return m.scale * c * ingain;              // ComputePanGains, core/mixer.cpp:99
    };
    if(!L.hrtf && t < L.numDry) L.gainTgt[size_t{v} * L.numDry + t] = gainFor(dryMap[t], r.dryGain);
    const uint32_t nsend = L.numSends * L.wetChannels;
    if(t >= 64u && t - 64u < nsend)
    {
        const uint32_t k = t - 64u, s = k / L.wetChannels, c = k % L.wetChannels;
        const int32_t slot = L.ctl[v].sendSlot[s];
        // (a send without a slot keeps mSend[i].Buffer empty: its gains are never read)
        L.sendTgt[size_t{v} * nsend + k] = slot >= 0 ? gainFor(wetMaps[size_t(slot) * L.wetChannels + c], r.sendGain[s]) : 0.0f;
    }
}

} // namespace

void LaunchPanGains(hipStream_t s, const DeviceLayout &L, const PanRecord *recs, uint32_t count, const AmbiMapEntry *dryMap,
    const AmbiMapEntry *wetMaps)
{ hipLaunchKernelGGL(PanGainsKernel, dim3(count), dim3(256), 0, s, L, recs, dryMap, wetMaps); }

} // namespace oalgpu
