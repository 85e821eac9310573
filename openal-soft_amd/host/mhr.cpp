// ".mhr" v3 reader: header, per-field elevation/azimuth counts, s24le impulse responses,
// u8 quarter-sample delays; left-only sets are mirrored to the right ear.
// Format and limits: core/hrtf_loader.cpp:40-55 (limits), :583-721 (LoadHrtf03),
// :135-155 (MirrorLeftHrirs).
#include "mhr.hpp"

#include <cstring>

namespace oalgpu {
namespace {

class ByteReader {
    const uint8_t *mCur, *mEnd;
    bool mOk{true};
public:
    ByteReader(const void *p, size_t n) : mCur{static_cast<const uint8_t*>(p)}, mEnd{mCur + n} { }
    bool ok() const { return mOk; }
    uint32_t le(unsigned bytes)
    {
        if(size_t(mEnd - mCur) < bytes) { mOk = false; mCur = mEnd; return 0; }
        uint32_t v = 0;
        for(unsigned i = 0; i < bytes; ++i) v |= uint32_t{mCur[i]} << (8 * i);
        mCur += bytes;
        return v;
    }
    int32_t s24() { const uint32_t v = le(3); return int32_t((v ^ 0x800000u) - 0x800000u); }
    bool match(const char *magic, size_t n)
    {
        if(size_t(mEnd - mCur) < n || std::memcmp(mCur, magic, n) != 0) return false;
        mCur += n;
        return true;
    }
};

} // namespace

std::string ParseMhr(const void *data, size_t size, HrtfData &out)
{
    ByteReader rd{data, size};
    if(!rd.match("MinPHR03", 8)) return "not a MinPHR03 data set (only format v3 is supported)";
    out = HrtfData{};
    out.sampleRate = rd.le(4);
    const uint32_t chanType = rd.le(1);
    out.irSize = rd.le(1);
    const uint32_t fdCount = rd.le(1);
    if(!rd.ok()) return "premature end of file";
    if(chanType > 1) return "unsupported channel type";
    if(out.irSize < 8 || out.irSize > 128) return "unsupported HRIR size";
    if(fdCount < 1 || fdCount > 16) return "unsupported field count";

    for(uint32_t f = 0; f < fdCount; ++f)
    {
        const uint32_t distMm = rd.le(2);
        const uint32_t evCount = rd.le(1);
        if(!rd.ok()) return "premature end of file";
        if(distMm < 50 || distMm > 2500) return "unsupported field distance";
        if(evCount < 5 || evCount > 181) return "unsupported elevation count";
        const float dist = float(distMm) / 1000.0f;
        if(f > 0 && !(dist < out.fieldDistance.back())) return "field distances not descending";
        out.fieldDistance.push_back(dist);
        out.fieldEvCount.push_back(uint8_t(evCount));
        for(uint32_t e = 0; e < evCount; ++e)
        {
            const uint32_t az = rd.le(1);
            if(az < 1 || az > 255) return "unsupported azimuth count";
            out.elevAzCount.push_back(uint16_t(az));
        }
        if(!rd.ok()) return "premature end of file";
    }
    out.elevIrOffset.resize(out.elevAzCount.size());
    uint32_t total = 0;
    for(size_t e = 0; e < out.elevAzCount.size(); ++e)
    {
        out.elevIrOffset[e] = uint16_t(total);
        total += out.elevAzCount[e];
    }

    const uint32_t ears = chanType == 0 ? 1 : 2;
    out.coeffs.assign(size_t{total} * 128 * 2, 0.0f);
    out.delays.assign(size_t{total} * 2, 0);
    for(uint32_t ir = 0; ir < total; ++ir)
        for(uint32_t i = 0; i < out.irSize; ++i)
            for(uint32_t c = 0; c < ears; ++c)
                out.coeffs[(size_t{ir} * 128 + i) * 2 + c] = float(rd.s24()) / 8388608.0f;
    for(uint32_t ir = 0; ir < total; ++ir)
        for(uint32_t c = 0; c < ears; ++c)
        {
            const uint32_t d = rd.le(1);
            if(d > (63u << 2)) return "delay out of range";
            out.delays[size_t{ir} * 2 + c] = uint8_t(d);
        }
    if(!rd.ok()) return "premature end of file";

    if(ears == 1)
    {
        // The right ear at azimuth index j is the left ear at the mirrored index.
        for(size_t e = 0; e < out.elevAzCount.size(); ++e)
        {
            const size_t base = out.elevIrOffset[e], azc = out.elevAzCount[e];
            for(size_t j = 0; j < azc; ++j)
            {
                const size_t l = base + j, r = base + ((azc - j) % azc);
                for(size_t i = 0; i < 128; ++i)
                    out.coeffs[(r * 128 + i) * 2 + 1] = out.coeffs[(l * 128 + i) * 2 + 0];
                out.delays[r * 2 + 1] = out.delays[l * 2 + 0];
            }
        }
    }
    return {};
}

} // namespace oalgpu
