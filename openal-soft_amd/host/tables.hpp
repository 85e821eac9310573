// Host-side resampler tables (kernel operands).  See tables.cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace oalgpu {

struct BsincTable {            // BSincTable, core/bsinc_tables.h:11-16
    float scaleBase{}, scaleRange{};
    uint32_t m[16]{};
    uint32_t filterOffset[16]{};
    std::vector<float> tab;
};

struct CubicTable {            // CubicCoefficients[32], core/cubic_defs.h:10-13
    float phase[32][8]{};      // {mCoeffs[4], mDeltas[4]}
};

const BsincTable *GetBsincTable(int which);   // 12, 24, 48
const CubicTable *GetCubicTable(int which);   // 0 spline, 1 gaussian

// gCubicTable (CubicFilter, core/cubic_tables.h:22-40): 2*256+1 floats
constexpr unsigned kFineCubicSteps = 256;
const float *GetFineCubicFilter();

} // namespace oalgpu
