// PPhaseResampler (common/polyphase_resampler.cpp) restated: a polyphase Kaiser-windowed sinc in double precision
// (180 dB rejection, transition ending at the lower Nyquist).  The reference uses it where it is "called very
// infrequently": convolution responses at another rate (convolution.cpp:351-356) and HRTF data sets at another
// rate than the device (core/hrtf.cpp:539-560).  Pinned to the compiled reference at 1e-12 (tests/test_conv_multichannel.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace oalgpu {

inline double BesselI0(double x)
{   // sum_k ((x/2)^k / k!)^2 until the term stops mattering (:32-54)
    const double x2 = x / 2.0;
    double term = 1.0, sum = 1.0, last = 0.0;
    int k = 1;
    do {
        const double y = x2 / k;
        ++k;
        last = sum;
        term *= y * y;
        sum += term;
    } while(sum != last);
    return sum;
}

struct Polyphase {
    uint32_t p{1}, q{1}, m{1}, l{0};
    std::vector<double> f;

    void init(uint32_t srcRate, uint32_t dstRate)
    {   // :154-178
        uint32_t a = srcRate, b = dstRate;
        while(b) { const uint32_t r = a % b; a = b; b = r; }
        p = dstRate / a; q = srcRate / a;
        const double down = double(p > q ? p : q);
        const double cutoff = 0.47 / down, width = 0.03 / down;
        constexpr double rejection = 180.0, pi = 3.14159265358979323846;
        // Kaiser order (:102-108) and beta (:111-119) for r > 50 dB
        const uint32_t order = uint32_t(std::ceil((rejection - 7.95) / (2.285 * 2.0 * pi * width)));
        const double beta = 0.1102 * (rejection - 8.7);
        const double i0beta = BesselI0(beta);
        l = (order + 1u) / 2u;
        m = l * 2u + 1u;
        f.resize(m);
        for(uint32_t i = 0; i < m; ++i)
        {   // SincFilter (:134-139): w(k) 2 p f_t sinc(2 f_t x)
            const double x = double(i) - double(l), k = x / double(l);
            const double w = (k >= -1.0 && k <= 1.0) ? BesselI0(beta * std::sqrt(1.0 - k * k)) / i0beta : 0.0;
            const double sx = 2.0 * cutoff * x;
            const double sinc = std::abs(sx) < 1e-9 ? 1.0 : std::sin(pi * sx) / (pi * sx);
            f[i] = w * 2.0 * double(p) * cutoff * sinc;
        }
    }

    void process(const double *in, size_t nin, double *out, size_t nout) const
    {   // :182-249 (never in place here)
        size_t pos = l;                           // starts at l: the filter's delay is dropped
        for(size_t o = 0; o < nout; ++o)
        {
            size_t js = pos / p, jf = pos % p;
            pos += q;
            double acc = 0.0;
            if(jf < m)
            {
                size_t len = (m - jf - 1) / p + 1;
                if(js + 1 > nin)
                {
                    const size_t skip = std::min(js + 1 - nin, len);
                    jf += size_t{p} * skip; js -= skip; len -= skip;
                }
                if(len != 0 && js + 1 <= nin)
                {
                    const size_t cnt = std::min(js + 1, len);
                    for(size_t k = 0; k < cnt; ++k) { acc = acc + f[jf] * in[js - k]; jf += p; }   // newest input first
                }
            }
            out[o] = acc;
        }
    }
};


} // namespace oalgpu
