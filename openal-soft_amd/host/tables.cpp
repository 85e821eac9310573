// Resampler coefficient tables, generated on the host exactly as the reference generates
// them at static-init time (f64 math narrowed to f32), then uploaded to HBM as kernel operands.
//
//   bsinc12/24/48 : core/bsinc_tables.cpp:147-371 (Kaiser-windowed sinc, 16 scales x 32 phases,
//                   per scale {filter, phase delta} x 32 followed by {scale delta, scale-phase
//                   delta} x 32; layout :255-339, header :114-145, public view :352-365)
//   spline/gaussian: core/cubic_tables.cpp:39-106 (32 phases x {coeffs[4], deltas[4]})
//
// tests/ compares these byte-for-byte with the tables dumped from the compiled reference.
#include "tables.hpp"

#include <algorithm>
#include <cmath>
#include <mutex>

namespace oalgpu {
namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr unsigned kScales = 16;
constexpr unsigned kPhases = 32;

// I0(x), summed until the term stops contributing (bsinc_tables.cpp:33-57).
double BesselI0(double x)
{
    const double half = x / 2.0;
    double term = 1.0, sum = 1.0, prev;
    int k = 1;
    do {
        const double y = half / k;
        ++k;
        prev = sum;
        term *= y * y;
        sum += term;
    } while(sum != prev);
    return sum;
}

double NormSinc(double x)
{
    constexpr double eps = 2.2204460492503131e-16;
    if(!(x > eps || x < -eps)) return 1.0;
    return std::sin(kPi * x) / (kPi * x);
}

double KaiserWindow(double beta, double k, double i0beta)
{
    if(!(k >= -1.0 && k <= 1.0)) return 0.0;
    return BesselI0(beta * std::sqrt(1.0 - k * k)) / i0beta;
}

// std::lerp as libstdc++ evaluates it for these operand ranges (a>0, b=1, 0<t<=1).
double Lerp(double a, double b, double t)
{
    if((a <= 0 && b >= 0) || (a >= 0 && b <= 0)) return t * b + (1 - t) * a;
    if(t == 1) return b;
    const double x = a + t * (b - a);
    return ((t > 1) == (b > a)) ? (b < x ? x : b) : (b > x ? x : b);
}

struct Design {
    double beta, scaleBase, scaleLimit;
    double a[kScales];
    unsigned m[kScales];
    size_t total;

    Design(double rejection, double order, double maxScale)
    {
        beta = (rejection > 50.0) ? 0.1102 * (rejection - 8.7)
            : (rejection >= 21.0) ? 0.5842 * std::pow(rejection - 21.0, 0.4) + 0.07886 * (rejection - 21.0)
            : 0.0;
        const double width = (rejection > 21.19) ? (rejection - 7.95) / (2.285 * kPi * 2.0 * order)
            : 5.79 / (kPi * 2.0) / order;
        scaleBase = width / 2.0;
        scaleLimit = 1.0 / maxScale;
        total = 0;
        const double baseA = (order + 1.0) / 2.0;
        for(unsigned si = 0; si < kScales; ++si)
        {
            const double scale = Lerp(scaleBase, 1.0, double(si + 1u) / double(kScales));
            a[si] = std::min(baseA / scale, baseA * maxScale);
            unsigned ai = static_cast<unsigned>(a[si]);
            if(double(ai) != a[si]) ++ai; // ceil
            m[si] = ai * 2u;
            total += size_t{4} * kPhases * ((m[si] + 3u) & ~3u);
        }
    }
};

void BuildBsinc(BsincTable &out, double rejection, double order, double maxScale)
{
    const Design d{rejection, order, maxScale};
    const unsigned width = (d.m[0] + 3u) & ~3u;
    std::vector<double> flt(size_t{kScales} * kPhases * width, 0.0);
    auto F = [&](unsigned si, unsigned pi, size_t i) -> double& {
        return flt[(size_t{si} * kPhases + pi) * width + i];
    };
    const double i0beta = BesselI0(d.beta);
    for(unsigned si = 0; si < kScales; ++si)
    {
        const unsigned m = d.m[si];
        const double l = std::floor(m * 0.5) - 1.0;
        const size_t o = (width - m) / 2u;
        const double scale = Lerp(d.scaleBase, 1.0, double(si + 1u) / double(kScales));
        const double maxCutoff = (0.5 - d.scaleBase) * scale;
        const double tw = d.scaleBase * std::max(d.scaleLimit, scale);
        const double cutoff2 = std::min(maxCutoff, (scale - tw) * 0.5) * 2.0;
        for(unsigned pi = 0; pi < kPhases; ++pi)
        {
            const double phase = l + double(pi) / kPhases;
            for(unsigned i = 0; i < m; ++i)
            {
                const double x = double(i) - phase;
                F(si, pi, o + i) = KaiserWindow(d.beta, x / d.a[si], i0beta) * cutoff2 * NormSinc(cutoff2 * x);
            }
        }
    }

    out.tab.assign(d.total, 0.0f);
    size_t idx = 0;
    for(unsigned si = 0; si < kScales; ++si)
    {
        const size_t m = (size_t{d.m[si]} + 3u) & ~size_t{3};
        const size_t o = (width - m) / 2u;
        for(unsigned pi = 0; pi < kPhases; ++pi)
        {
            for(size_t i = 0; i < m; ++i) out.tab[idx++] = float(F(si, pi, o + i));
            if(pi + 1 < kPhases)
                for(size_t i = 0; i < m; ++i) out.tab[idx++] = float(F(si, pi + 1, o + i) - F(si, pi, o + i));
            else
            {
                // last phase interpolates toward phase 0 shifted by one sample
                out.tab[idx++] = float(0.0 - F(si, pi, o));
                for(size_t i = 1; i < m; ++i) out.tab[idx++] = float(F(si, 0, o + i - 1) - F(si, pi, o + i));
            }
        }
        if(si + 1 < kScales)
        {
            for(unsigned pi = 0; pi < kPhases; ++pi)
            {
                for(size_t i = 0; i < m; ++i) out.tab[idx++] = float(F(si + 1, pi, o + i) - F(si, pi, o + i));
                if(pi + 1 < kPhases)
                {
                    for(size_t i = 0; i < m; ++i)
                        out.tab[idx++] = float((F(si + 1, pi + 1, o + i) - F(si + 1, pi, o + i))
                            - (F(si, pi + 1, o + i) - F(si, pi, o + i)));
                }
                else
                {
                    out.tab[idx++] = float((0.0 - F(si + 1, pi, o)) - (0.0 - F(si, pi, o)));
                    for(size_t i = 1; i < m; ++i)
                        out.tab[idx++] = float((F(si + 1, 0, o + i - 1) - F(si + 1, pi, o + i))
                            - (F(si, 0, o + i - 1) - F(si, pi, o + i)));
                }
            }
        }
        else
            idx += size_t{kPhases} * m * 2; // no scale deltas for the last scale: zeros
    }

    out.scaleBase = float(d.scaleBase);
    out.scaleRange = float(1.0 / (1.0 - d.scaleBase));
    for(unsigned i = 0; i < kScales; ++i) out.m[i] = (d.m[i] + 3u) & ~3u;
    out.filterOffset[0] = 0;
    for(unsigned i = 1; i < kScales; ++i)
        out.filterOffset[i] = out.filterOffset[i - 1] + out.m[i - 1] * 4u * kPhases;
}

double GaussCoeff(double idx)
{
    const double k = 0.5 + idx;
    if(k > 512.0) return 0.0;
    const double s = std::sin(kPi * 1.280 / 1024.0 * k);
    const double t = (std::cos(kPi * 2.000 / 1023.0 * k) - 1.0) * 0.50;
    const double u = (std::cos(kPi * 4.000 / 1023.0 * k) - 1.0) * 0.08;
    return s * (t + u + 1.0) / k;
}

void FillDeltas(CubicTable &t)
{
    for(unsigned pi = 0; pi + 1 < kPhases; ++pi)
        for(int k = 0; k < 4; ++k) t.phase[pi][4 + k] = t.phase[pi + 1][k] - t.phase[pi][k];
    const unsigned last = kPhases - 1;
    t.phase[last][4] = 0.0f - t.phase[last][0];
    for(int k = 1; k < 4; ++k) t.phase[last][4 + k] = t.phase[0][k - 1] - t.phase[last][k];
}

struct AllTables {
    BsincTable bsinc[3];
    CubicTable cubic[2];
    AllTables()
    {
        BuildBsinc(bsinc[0], 60, 11, 2);
        BuildBsinc(bsinc[1], 60, 23, 2);
        BuildBsinc(bsinc[2], 80, 47, 1);
        // Catmull-Rom spline
        for(unsigned pi = 0; pi < kPhases; ++pi)
        {
            const double mu = double(pi) / double(kPhases), mu2 = mu * mu, mu3 = mu * mu2;
            cubic[0].phase[pi][0] = float(-(1.0 / 3.0) * mu + 0.5 * mu2 - (1.0 / 6.0) * mu3);
            cubic[0].phase[pi][1] = float(1.0 - 0.5 * mu - mu2 + 0.5 * mu3);
            cubic[0].phase[pi][2] = float(mu + 0.5 * mu2 - 0.5 * mu3);
            cubic[0].phase[pi][3] = float(-(1.0 / 6.0) * mu + (1.0 / 6.0) * mu3);
        }
        FillDeltas(cubic[0]);
        // SNES-style gaussian
        const double indexScale = 512.0 / double(kPhases * 2);
        for(unsigned pi = 0; pi < kPhases; ++pi)
        {
            const double c0 = GaussCoeff(double(kPhases + pi) * indexScale);
            const double c1 = GaussCoeff(double(pi) * indexScale);
            const double c2 = GaussCoeff(double(kPhases - pi) * indexScale);
            const double c3 = GaussCoeff(double(kPhases * 2 - pi) * indexScale);
            const double norm = 1.0 / (c0 + c1 + c2 + c3);
            cubic[1].phase[pi][0] = float(c0 * norm);
            cubic[1].phase[pi][1] = float(c1 * norm);
            cubic[1].phase[pi][2] = float(c2 * norm);
            cubic[1].phase[pi][3] = float(c3 * norm);
        }
        FillDeltas(cubic[1]);
    }
};

const AllTables &Tables()
{
    static const AllTables t;
    return t;
}

// CubicFilter::CubicFilter, core/cubic_tables.cpp:109-128: the 256-step gaussian the reverb's
// modulated feedback taps interpolate with (getCoeff0..3 index one 513-entry array)
struct FineCubic {
    float f[kFineCubicSteps * 2 + 1]{};
    FineCubic()
    {
        const double indexScale = 512.0 / double(kFineCubicSteps * 2);
        for(unsigned i = 0; i < kFineCubicSteps / 2 + 1; ++i)
        {
            const double c0 = GaussCoeff(double(kFineCubicSteps + i) * indexScale);
            const double c1 = GaussCoeff(double(i) * indexScale);
            const double c2 = GaussCoeff(double(kFineCubicSteps - i) * indexScale);
            const double c3 = GaussCoeff(double(kFineCubicSteps * 2 - i) * indexScale);
            const double norm = 1.0 / (c0 + c1 + c2 + c3);
            f[kFineCubicSteps + i] = float(c0 * norm);
            f[i] = float(c1 * norm);
            f[kFineCubicSteps - i] = float(c2 * norm);
            f[kFineCubicSteps * 2 - i] = float(c3 * norm);
        }
    }
};

} // namespace

const BsincTable *GetBsincTable(int which)
{
    switch(which)
    {
    case 12: return &Tables().bsinc[0];
    case 24: return &Tables().bsinc[1];
    case 48: return &Tables().bsinc[2];
    }
    return nullptr;
}

const CubicTable *GetCubicTable(int which)
{
    if(which < 0 || which > 1) return nullptr;
    return &Tables().cubic[which];
}

const float *GetFineCubicFilter()
{
    static const FineCubic t;
    return t.f;
}

} // namespace oalgpu
