// EAX reverb, ReverbState::process and what it calls (alc/effects/reverb.cpp:1396-1883), as ONE
// launch of ONE workgroup per reverb instance.
//
// The network is a feedback structure: every 256-sample sub-block of the late reverb reads what
// the previous one wrote, and inside a sub-block four things are true recurrences -- the two
// dual-biquad sections (master band-pass, T60 damping) and the two all-pass stages (per-line
// comb all-pass, Gerzon vector all-pass with its scattering matrix).  Everything else (tap reads,
// cross-fades, cubic interpolation of the modulated feedback taps, reflections, scatters,
// delay-line writes) is data parallel over 4 lines x 256 samples.  So the parallelism on offer
// is: 4 lines, up to 256 samples in the parallel phases, early/late sections of consecutive
// sub-blocks, and the two pipelines while a parameter change cross-fades.  The workgroup has four
// wavefronts with fixed roles:
//     wave 0  early reflections of the current pipeline   (processEarly, :1558-1660)
//     wave 1  late reverb of the current pipeline         (processLate, :1696-1811)
//     wave 2  early reflections of the old pipeline        } only while fading
//     wave 3  late reverb of the old pipeline              }
// A late wave waits for its early wave through a progress word in LDS (the late input line taps
// may be zero samples behind).  Inside a wave: parallel phases run over the 64 lanes, the biquad
// recurrences run one line per lane (4 lanes) in the reference's operation order, the all-pass
// recurrences run in LDS windows in chunks of the shortest delay (within which they are
// parallel).  All arithmetic keeps the reference's order (-ffp-contract=off, FTZ), so the output
// is bit-identical to the CPU reference; the mix-out (MixOutPlain :637-656) is done last, by all
// 256 threads, adding the 8 (16 while fading) inputs per output sample in the reference's order.
//
// Delay lines stay in HBM (1.6 MB per instance, L2 resident); per launch the algorithmic traffic
// is ~100 KB, so this kernel is latency-bound by construction, not bandwidth-bound.
#include "dev_wave.hpp"
#include "reverb_dev.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr uint32_t kSub = 256;                   // MAX_UPDATE_SAMPLES, reverb.cpp:68
constexpr uint32_t kRow = kSub + 4;              // LDS row pitch of the 4-line scratch: the four biquad
                                                 // lanes hit different banks, rows stay 16-byte aligned
constexpr uint32_t kModFracOne = 1u << 24, kModFracMask = kModFracOne - 1u;   // :60-62
constexpr uint32_t kCubicBits = 8, kCubicSteps = 1u << kCubicBits, kCubicMask = kCubicSteps - 1u;

// The all-pass windows hold the delay history a block starts from plus everything the block
// adds (loaded once per launch; each sub-block writes its new samples back to HBM).
struct alignas(16) EarlyLds {
    float temp[4 * kRow];                                 // tempSamples[4][256]
    float eap[4 * (kRvMaxEarlyApOffset + kLine)];         // per-line all-pass windows
};
struct alignas(16) LateLds {
    float temp[4 * kRow];
    float vap[4 * (kRvMaxLateApOffset + kLine)];          // interleaved vector all-pass window
    uint32_t modDelays[kSub];
};
struct PipeLds { EarlyLds e; LateLds l; };

struct MixGain { float cur, step, tgt; uint32_t mode; };   // mode 0: silent, 1: fading, 2: constant gain

struct RvLds {
    PipeLds pipe[2];                             // [0] current pipeline, [1] old pipeline
    float cubic[kCubicSteps * 2 + 4];            // gCubicTable
    MixGain mix[2][8][OALGPU_MAX_AMBI_CHANNELS]; // [pipeline][early 0-3, late 4-7][target line]
    uint32_t progress[2];                        // samples of this block the early wave has finished
};

#define RV_STAMP(role, sub, k) do { if(L.stamps && lane == 0 && (sub) < 8u) \
    L.stamps[((role) * 8u + (sub)) * 8u + (k)] = __builtin_readcyclecounter(); } while(0)

__device__ __forceinline__ float Lerp(float a, float b, float mu) { return a + (b - a) * mu; }

// VectorPartialScatter, :1396-1405
__device__ __forceinline__ void PartialScatter(float (&o)[4], const float (&in)[4], float x, float y)
{
    const float o0 = x * in[0] + y * (in[1] + -in[2] + in[3]);
    const float o1 = x * in[1] + y * (-in[0] + in[2] + in[3]);
    const float o2 = x * in[2] + y * (in[0] + -in[1] + in[3]);
    const float o3 = x * in[3] + y * (-in[0] + -in[1] + -in[2]);
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}

// DualBiquad{f0,f1}.process in place over row[0..todo) -- BiquadFilter::dualProcess,
// core/filters/biquad.cpp:254-282.  One lane, serial in time.  The second section runs one sample
// behind the first, so that both sections' identical operation sequences pack into v_pk_mul_f32 /
// v_pk_add_f32 (.x = first section on sample i, .y = second section on sample i-1): 9 packed
// instructions per sample instead of 18 scalar ones, every product and sum still rounded on its
// own exactly as the reference's scalar code does.
__device__ __forceinline__ void DualBiquadSerial(float *row, uint32_t todo, const oalgpu_bq_coeffs &c0,
    const oalgpu_bq_coeffs &c1, float *z)
{
    f2 Z1 = {z[0], z[2]}, Z2 = {z[1], z[3]};
    const f2 B0 = {c0.b0, c1.b0}, B1 = {c0.b1, c1.b1}, B2 = {c0.b2, c1.b2}, A1 = {c0.a1, c1.a1}, A2 = {c0.a2, c1.a2};
    float yprev;
    {   // first section alone on sample 0
        const float x0 = row[0];
        const float y0 = x0 * B0.x + Z1.x;
        Z1.x = x0 * B1.x - y0 * A1.x + Z2.x;
        Z2.x = x0 * B2.x - y0 * A2.x;
        yprev = y0;
    }
#pragma unroll 8
    for(uint32_t i = 1; i < todo; ++i)
    {
        const f2 X = {row[i], yprev};
        const f2 Y = X * B0 + Z1;
        Z1 = X * B1 - Y * A1 + Z2;
        Z2 = X * B2 - Y * A2;
        row[i - 1] = Y.y;
        yprev = Y.x;
    }
    {   // second section alone on the last sample
        const float y1 = yprev * B0.y + Z1.y;
        Z1.y = yprev * B1.y - y1 * A1.y + Z2.y;
        Z2.y = yprev * B2.y - y1 * A2.y;
        row[todo - 1] = y1;
    }
    z[0] = Z1.x; z[1] = Z2.x; z[2] = Z1.y; z[3] = Z2.y;
}

// FAST mode: the same two sections over the four lines of the scratch as block scans -- one DPP row of 16 lanes per
// line, a run of 17 samples per lane (an odd pitch: the lanes of a row hit different LDS banks), the run-start states
// by a Kogge-Stone scan inside the row (ScanLinear2Row, dev_wave.hpp); the forced response as sum x[i] A^(16-i) Bv out
// of the loop that raises A to M = A^17, like the voice kernels' BiquadWaveScan.  The serial form is a chain of ~4
// dependent operations per sample on 4 of the wavefront's 64 lanes (10 K cycles per 256-sample sub-block and section
// pair); this is ~100 dependent operations per section.  Differs from the reference's loop by rounding only.
__device__ __forceinline__ void BiquadRowScan(float (&x)[kBqSeg], uint32_t cnt, const oalgpu_bq_coeffs &f, float &z1, float &z2,
    uint32_t lane, bool last)
{
    const float b0 = f.b0, b1 = f.b1, b2 = f.b2, a1 = f.a1, a2 = f.a2;
    S2 m0{1.0f, 0.0f}, m1{0.0f, 1.0f};
    S2 c{__builtin_fmaf(-a1, b0, b1), __builtin_fmaf(-a2, b0, b2)};
    S2 ea{0.0f, 0.0f}, eb{0.0f, 0.0f};
#pragma unroll
    for(int j = 0; j < kBqSeg; ++j)
    {
        const float xv = x[kBqSeg - 1 - j];
        if(j & 1) { eb.a = __builtin_fmaf(xv, c.a, eb.a); eb.b = __builtin_fmaf(xv, c.b, eb.b); }
        else { ea.a = __builtin_fmaf(xv, c.a, ea.a); ea.b = __builtin_fmaf(xv, c.b, ea.b); }
        c = S2{__builtin_fmaf(-a1, c.a, c.b), -a2 * c.a};
        m0 = S2{__builtin_fmaf(-a1, m0.a, m0.b), -a2 * m0.a};
        m1 = S2{__builtin_fmaf(-a1, m1.a, m1.b), -a2 * m1.a};
    }
    S2 e{ea.a + eb.a, ea.b + eb.b};
    {   // the row's first lane starts from the line's filter state
        const S2 mz = Mv2(m0, m1, S2{z1, z2});
        if((lane & 15u) == 0u) { e.a += mz.a; e.b += mz.b; }
    }
    S2 q0, q1;
    e = ScanLinear2Row(e, m0, m1, lane, q0, q1);
    // the run's start state: the end state of the lane before in the row (row_shr:1; a row's first lane keeps the line's state)
    S2 st;
    st.a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z1), __builtin_bit_cast(int, e.a), 0x111, 0xF, 0xF, false));
    st.b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z2), __builtin_bit_cast(int, e.b), 0x111, 0xF, 0xF, false));
    S2 zc = st;
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i)
    {
        const float xv = x[i];
        const float t1 = __builtin_fmaf(xv, b1, st.b);
        const float y = __builtin_fmaf(xv, b0, st.a);
        st.a = __builtin_fmaf(-y, a1, t1);
        st.b = __builtin_fmaf(xv, b2, -y * a2);
        x[i] = y;
        if(uint32_t(i) + 1u == cnt) zc = st;
    }
    if(last) { z1 = zc.a; z2 = zc.b; }       // (only the lane that holds the line's last sample keeps the new state)
}

// temp: the 4-line scratch (pitch kRow); c0 / c1: THIS LANE's line's coefficients (line = lane >> 4); z: that line's four
// state words {section 0: z1, z2; section 1: z1, z2}
__device__ __forceinline__ void DualBiquadRowsFast(float *temp, uint32_t todo, const oalgpu_bq_coeffs &c0, const oalgpu_bq_coeffs &c1,
    float *z, uint32_t lane)
{
    const uint32_t l = lane & 15u;
    float *row = temp + (lane >> 4) * kRow;
    const uint32_t begin = l * uint32_t(kBqSeg) < todo ? l * uint32_t(kBqSeg) : todo;
    const uint32_t cnt = (begin + uint32_t(kBqSeg) < todo) ? uint32_t(kBqSeg) : todo - begin;
    const bool last = todo != 0u && l == (todo - 1u) / uint32_t(kBqSeg);
    float x[kBqSeg];
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) x[i] = (uint32_t(i) < cnt) ? row[begin + i] : 0.0f;
    float s0 = z[0], s1 = z[1], s2 = z[2], s3 = z[3];
    BiquadRowScan(x, cnt, c0, s0, s1, lane, last);
    BiquadRowScan(x, cnt, c1, s2, s3, lane, last);
    WaveSync();                                  // every lane of the row has read the old state
    if(last) { z[0] = s0; z[1] = s1; z[2] = s2; z[3] = s3; }
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) if(uint32_t(i) < cnt) row[begin + i] = x[i];
}

// ---- early reflections: ReverbPipeline::processEarly, :1558-1660 -------------------------------
__device__ void EarlyWave(const RvLayout &L, const int p, EarlyLds &w, uint32_t *progress, const uint32_t lane)
{
    const oalgpu_reverb_pipeline &P = L.pipe[p];
    RvPipeState &S = L.state[p];
    const RvLines &ln = L.lines[p];
    const uint32_t n = L.n;
    const uint32_t mainMask = L.mainStride - 1u, eapMask = ln.eapStride - 1u, edMask = ln.edelayStride - 1u,
        liMask = ln.lateInStride - 1u;
    const float mixX = P.mix_x, mixY = P.mix_y, feed = P.early_ap_coeff, delayCoeff = P.early_coeff;
    float *earlyOut = L.earlyOut + size_t(p) * 4u * kLine;

    uint32_t offset = L.offset;
    const uint32_t role = p == L.current ? 0u : 2u;
    {   // all-pass history: win[k] = line sample at position L.offset - Offset[j] + k
        const uint32_t j = lane >> 4, l = lane & 15u;
        const uint32_t off = P.early_ap_offset[j];
        float *win = &w.eap[j * (kRvMaxEarlyApOffset + kLine)];
        const float *line = ln.eap + size_t{j} * ln.eapStride;
        for(uint32_t k = l; k < off; k += 16) win[k] = line[(offset - off + k) & eapMask];
    }
    for(uint32_t base = 0, sub = 0; base < n; ++sub)
    {
        const uint32_t todo = (n - base < kSub) ? n - base : kSub;
        const float fadeStep = 1.0f / float(todo);
        RV_STAMP(role, sub, 0);
        // the hand-over to the target taps happens after the first sub-block (:1579,1585)
        const float c0 = base ? P.early_delay_coeff[1] : P.early_delay_coeff[0];
        const float c1 = P.early_delay_coeff[1];
        {   // all 32 loads of the lane first, then the arithmetic (one memory round trip)
            float in0[4][4], in1[4][4];
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j)
            {
                const float *input = L.mainDelay + size_t{j} * L.mainStride;
                const uint32_t tap0 = offset - (base ? P.early_delay_tap[j][1] : P.early_delay_tap[j][0]);
                const uint32_t tap1 = offset - P.early_delay_tap[j][1];
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k)
                {
                    const uint32_t i = lane + 64u * k;
                    in0[j][k] = input[(tap0 + i) & mainMask];
                    in1[j][k] = input[(tap1 + i) & mainMask];
                }
            }
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j)
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k)
                {
                    const uint32_t i = lane + 64u * k;
                    if(i < todo) w.temp[j * kRow + i] = Lerp(in0[j][k] * c0, in1[j][k] * c1, fadeStep * float(i));
                }
        }
        WaveSync();
        RV_STAMP(role, sub, 1);
        if(L.fast) DualBiquadRowsFast(w.temp, todo, P.filter_lp, P.filter_hp, &S.z[lane >> 4][0], lane);
        else if(lane < 4)                                       // mFilter[j].process, :1611
            DualBiquadSerial(&w.temp[lane * kRow], todo, P.filter_lp, P.filter_hp, &S.z[lane][0]);
        WaveSync();
        RV_STAMP(role, sub, 2);

        // Allpass4::process, :1508-1540, in the LDS window of the line.  16 lanes per line; a chunk
        // of Offset[j] samples has no dependency inside it.
        {
            const uint32_t j = lane >> 4, l = lane & 15u;
            const uint32_t off = P.early_ap_offset[j];
            float *win = &w.eap[j * (kRvMaxEarlyApOffset + kLine)] + base;
            float *line = ln.eap + size_t{j} * ln.eapStride;
            float *row = &w.temp[j * kRow];
            for(uint32_t cb = 0; cb < todo; cb += off)
            {
                const uint32_t cnt = (todo - cb < off) ? todo - cb : off;
                for(uint32_t i = l; i < cnt; i += 16)
                {
                    const float x = row[cb + i];
                    const float y = win[cb + i] - feed * x;
                    win[off + cb + i] = x + feed * y;
                    row[cb + i] = y;
                }
                WaveSync();
            }
            for(uint32_t i = l; i < todo; i += 16) line[(offset + i) & eapMask] = win[off + i];
        }
        WaveSync();
        RV_STAMP(role, sub, 3);

        // DelayLineU::writeReflected, :340-365
        for(uint32_t i = lane; i < todo; i += 64)
        {
            const float s0 = w.temp[0 * kRow + i], s1 = w.temp[1 * kRow + i], s2 = w.temp[2 * kRow + i],
                s3 = w.temp[3 * kRow + i];
            const uint32_t o = (offset + i) & edMask;
            ln.edelay[0 * size_t{ln.edelayStride} + o] = (s0 - s1 - s2 - s3) * 0.5f;
            ln.edelay[1 * size_t{ln.edelayStride} + o] = (s1 - s0 - s2 - s3) * 0.5f;
            ln.edelay[2 * size_t{ln.edelayStride} + o] = (s2 - s0 - s1 - s3) * 0.5f;
            ln.edelay[3 * size_t{ln.edelayStride} + o] = (s3 - s0 - s1 - s2) * 0.5f;
        }
        // the taps below may land on samples this wave has just stored
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        WaveSync();
        RV_STAMP(role, sub, 4);
        {                                                       // :1619-1643
            float dl[4][4];
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j)
            {
                const float *buf = ln.edelay + size_t{j} * ln.edelayStride;
                const uint32_t tap = offset - P.early_offset[j];
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k) dl[j][k] = buf[(tap + lane + 64u * k) & edMask];
            }
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j)
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k)
                {
                    const uint32_t i = lane + 64u * k;
                    if(i < todo) earlyOut[j * kLine + base + i] = dl[j][k] * delayCoeff + w.temp[j * kRow + i];
                }
        }
        // VectorScatter (:1408-1423) into the late input line (:1649-1655)
        for(uint32_t i = lane; i < todo; i += 64)
        {
            const float v[4] = {w.temp[0 * kRow + i], w.temp[1 * kRow + i], w.temp[2 * kRow + i], w.temp[3 * kRow + i]};
            float o[4];
            PartialScatter(o, v, mixX, mixY);
            const uint32_t pos = (offset + i) & liMask;
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j) ln.lateIn[size_t{j} * ln.lateInStride + pos] = o[j];
        }
        base += todo;
        offset += todo;
        WaveSync();
        RV_STAMP(role, sub, 5);
        // publish: the late wave of this pipeline may now read late-input samples < base
        __hip_atomic_store(progress, base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ---- late reverb: ReverbPipeline::processLate, :1696-1811 --------------------------------------
__device__ void LateWave(const RvLayout &L, const int p, LateLds &w, uint32_t *progress, const float *cubic,
    const uint32_t lane)
{
    const oalgpu_reverb_pipeline &P = L.pipe[p];
    RvPipeState &S = L.state[p];
    const RvLines &ln = L.lines[p];
    const uint32_t n = L.n;
    const uint32_t ldMask = ln.ldelayStride - 1u, liMask = ln.lateInStride - 1u, vapMask = ln.vapStride - 1u;
    const float mixX = P.mix_x, mixY = P.mix_y, feed = P.late_ap_coeff;
    const float depth = P.mod_depth * float(kCubicSteps);
    const uint32_t step = P.mod_step;
    float *lateOut = L.lateOut + size_t(p) * 4u * kLine;
    uint32_t vapMax = 0;
    for(uint32_t c = 0; c < 4; ++c) vapMax = P.late_ap_offset[c] > vapMax ? P.late_ap_offset[c] : vapMax;

    uint32_t offset = L.offset;
    const uint32_t role = p == L.current ? 1u : 3u;
    {   // vector all-pass history: vap[k*4 + c] = interleaved line sample at L.offset - vapMax + k
        float4 *win4 = reinterpret_cast<float4*>(w.vap);
        const float4 *line4 = reinterpret_cast<const float4*>(ln.vap);
        for(uint32_t k = lane; k < vapMax; k += 64) win4[k] = line4[(offset - vapMax + k) & vapMask];
    }
    for(uint32_t base = 0, sub = 0; base < n; ++sub)
    {
        uint32_t todo = P.late_offset[0] < kSub ? P.late_offset[0] : kSub;
        todo = (n - base < todo) ? n - base : todo;
        RV_STAMP(role, sub, 0);

        // Modulation::calcDelays, :1662-1682
        for(uint32_t i = lane; i < todo; i += 64)
        {
            const uint32_t idx = L.modIndex[p] + (base + i) * step;
            const float x = float(idx & kModFracMask) * (1.0f / float(kModFracOne));
            const float lfo = !(idx & (kModFracOne >> 1))
                ? ((-16.0f * x * x) + (8.0f * x))
                : ((16.0f * x * x) + (-8.0f * x) + (-16.0f * x) + 8.0f);
            w.modDelays[i] = uint32_t((lfo + 1.0f) * depth);
        }
        WaveSync();
        // modulated feedback taps, cubic-interpolated (:1718-1747).  All four lines' 64 taps are requested before the first
        // is used: one round trip to the feedback lines per sub-block on the late line's chain instead of four.
        {
            float o[4][4][4];
            uint32_t doff[4];
#pragma unroll
            for(uint32_t k = 0; k < 4; ++k)
            {
                const uint32_t i = lane + 64u * k;
                const uint32_t idelay = w.modDelays[i < todo ? i : 0u];
                doff[k] = idelay & kCubicMask;
#pragma unroll
                for(uint32_t j = 0; j < 4; ++j)
                {
                    const float *input = ln.ldelay + size_t{j} * ln.ldelayStride;
                    const uint32_t delay = (offset - P.late_offset[j]) + i - (idelay >> kCubicBits);
#pragma unroll
                    for(uint32_t m = 0; m < 4; ++m) o[j][k][m] = input[(delay - m) & ldMask];
                }
            }
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j)
            {
                const float midGain = P.t60_mid_gain[j];
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k)
                {
                    const uint32_t i = lane + 64u * k;
                    const float out = o[j][k][0] * cubic[kCubicSteps + doff[k]] + o[j][k][1] * cubic[doff[k]]
                        + o[j][k][2] * cubic[kCubicSteps - doff[k]] + o[j][k][3] * cubic[kCubicSteps * 2u - doff[k]];
                    if(i < todo) w.temp[j * kRow + i] = out * midGain;
                }
            }
        }
        WaveSync();
        RV_STAMP(role, sub, 1);
        if(L.fast) DualBiquadRowsFast(w.temp, todo, P.t60_hf[lane >> 4], P.t60_lf[lane >> 4], &S.z[lane >> 4][4], lane);
        else if(lane < 4)                                       // mLate.T60[j].process, :1749
            DualBiquadSerial(&w.temp[lane * kRow], todo, P.t60_hf[lane], P.t60_lf[lane], &S.z[lane][4]);
        WaveSync();
        RV_STAMP(role, sub, 2);

        // the late input line must hold this pipeline's early output up to base + todo
        while(__hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < base + todo)
            __builtin_amdgcn_s_sleep(2);
        RV_STAMP(role, sub, 3);
        const float fadeStep = 1.0f / float(todo);
        {                                                       // :1753-1785
            float in0[4][4], in1[4][4];
            float densityStep[4];
            const float densityGain = P.late_density_gain;
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j)
            {
                const float *input = ln.lateIn + size_t{j} * ln.lateInStride;
                const uint32_t tap0 = offset - (base ? P.late_delay_tap[j][1] : P.late_delay_tap[j][0]);
                const uint32_t tap1 = offset - P.late_delay_tap[j][1];
                densityStep[j] = (tap0 != tap1) ? densityGain * fadeStep : 0.0f;
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k)
                {
                    const uint32_t i = lane + 64u * k;
                    in0[j][k] = input[(tap0 + i) & liMask];
                    in1[j][k] = input[(tap1 + i) & liMask];
                }
            }
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j)
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k)
                {
                    const uint32_t i = lane + 64u * k;
                    const float fadeCount = float(i);
                    const float fade0 = densityGain - densityStep[j] * fadeCount;
                    const float fade1 = densityStep[j] * fadeCount;
                    if(i < todo) w.temp[j * kRow + i] = in0[j][k] * fade0 + in1[j][k] * fade1 + w.temp[j * kRow + i];
                }
        }
        WaveSync();
        RV_STAMP(role, sub, 4);

        // VecAllpass::process, :1452-1503, in the LDS window.  Chunks of Offset[0] (the shortest
        // delay).
        {
            float4 *win4 = reinterpret_cast<float4*>(w.vap) + base;
            float *winf = w.vap + size_t{base} * 4u;
            const uint32_t minOff = P.late_ap_offset[0];
            for(uint32_t cb = 0; cb < todo;)
            {
                const uint32_t td = (todo - cb < minOff) ? todo - cb : minOff;
                for(uint32_t e = lane; e < td * 4u; e += 64)
                {
                    const uint32_t c = e & 3u, i = cb + (e >> 2);
                    const float input = w.temp[c * kRow + i];
                    const float out = winf[(vapMax + i - P.late_ap_offset[c]) * 4u + c] - feed * input;
                    winf[(vapMax + i) * 4u + c] = input + feed * out;
                    w.temp[c * kRow + i] = out;
                }
                WaveSync();
                for(uint32_t i = lane; i < td; i += 64)
                {
                    const float4 d = win4[vapMax + cb + i];
                    const float v[4] = {d.x, d.y, d.z, d.w};
                    float o[4];
                    PartialScatter(o, v, mixX, mixY);
                    win4[vapMax + cb + i] = make_float4(o[0], o[1], o[2], o[3]);
                }
                WaveSync();
                cb += td;
            }
            float4 *out4 = reinterpret_cast<float4*>(ln.vap);
            for(uint32_t i = lane; i < todo; i += 64) out4[(offset + i) & vapMask] = win4[vapMax + i];
        }
        RV_STAMP(role, sub, 5);
        // out for mixing (:1791-1797), then VectorScatterRev into the feedback lines (:1800-1806)
        for(uint32_t i = lane; i < todo; i += 64)
        {
            const float t0 = w.temp[0 * kRow + i], t1 = w.temp[1 * kRow + i], t2 = w.temp[2 * kRow + i],
                t3 = w.temp[3 * kRow + i];
            lateOut[0 * kLine + base + i] = t0; lateOut[1 * kLine + base + i] = t1;
            lateOut[2 * kLine + base + i] = t2; lateOut[3 * kLine + base + i] = t3;
            const float v[4] = {t3, t2, t1, t0};
            float o[4];
            PartialScatter(o, v, mixX, mixY);
            const uint32_t pos = (offset + i) & ldMask;
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j) ln.ldelay[size_t{j} * ln.ldelayStride + pos] = o[j];
        }
        // the next sub-block's feedback taps may land on samples this wave has just stored
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        WaveSync();
        RV_STAMP(role, sub, 6);
        base += todo;
        offset += todo;
    }
}

// One workgroup = one reverb instance.  `ticket` (null for a lone instance): several instances
// that add into the SAME target lines run side by side up to their mix-out, which they then
// perform one after the other in launch order -- instance `order` waits until the ticket word
// reaches its number -- so that the sums come out in the order a serial loop over the slots
// (alc/alu.cpp:2209-2257) would produce.
__device__ __forceinline__ void ReverbProcessBody(const RvLayout &L, RvLds &sm, uint32_t *ticket, uint32_t order, uint32_t count)
{
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t n = L.n;
    const int cur = L.current, old = !L.current;
    const bool oldRuns = (L.oldMode == 1 || L.oldMode == 2);
    if(L.stamps && t == 0) L.stamps[7 * 8 + 0] = __builtin_readcyclecounter();

    // B-Format -> A-Format into the main delay line (:1824-1838)
    {
        constexpr float B2A[4][4] = {{0.5f, 0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, -0.5f, 0.5f},
            {0.5f, 0.5f, -0.5f, -0.5f}, {0.5f, -0.5f, 0.5f, -0.5f}};
        const uint32_t mask = L.mainStride - 1u;
        for(uint32_t i = t; i < n; i += 256)
        {
            float in[4];
#pragma unroll
            for(int k = 0; k < 4; ++k) in[k] = L.wetIn[k * kLine + i];
#pragma unroll
            for(int c = 0; c < 4; ++c)
            {
                float tmp = 0.0f;
#pragma unroll
                for(int k = 0; k < 4; ++k) tmp = tmp + in[k] * B2A[c][k];
                L.mainDelay[size_t(c) * L.mainStride + ((L.offset + i) & mask)] = tmp;
            }
        }
    }
    if(L.oldMode == 3)
    {   // ReverbPipeline::clear, :550-564 (the scalar fields were cleared by the host)
        const RvLines &ln = L.lines[old];
        float4 *lines[5] = {reinterpret_cast<float4*>(ln.lateIn), reinterpret_cast<float4*>(ln.eap),
            reinterpret_cast<float4*>(ln.edelay), reinterpret_cast<float4*>(ln.vap), reinterpret_cast<float4*>(ln.ldelay)};
        const uint32_t counts[5] = {ln.lateInStride, ln.eapStride, ln.edelayStride, ln.vapStride, ln.ldelayStride};
        for(int k = 0; k < 5; ++k)
            for(uint32_t i = t; i < counts[k]; i += 256) lines[k][i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float *st = reinterpret_cast<float*>(&L.state[old]);
        for(uint32_t i = t; i < sizeof(RvPipeState) / sizeof(float); i += 256) st[i] = 0.0f;
    }
    if(t < 2) sm.progress[t] = 0u;
    for(uint32_t k = t; k < kCubicSteps * 2u + 1u; k += 256) sm.cubic[k] = L.cubic[k];
    __syncthreads();
    if(L.stamps && t == 0) L.stamps[7 * 8 + 1] = __builtin_readcyclecounter();

    {
        const int p = (wave >> 1) ? old : cur;
        if(wave < 2 || oldRuns)
        {
            if(!(wave & 1u)) EarlyWave(L, p, sm.pipe[wave >> 1].e, &sm.progress[wave >> 1], lane);
            else LateWave(L, p, sm.pipe[wave >> 1].l, &sm.progress[wave >> 1], sm.cubic, lane);
        }
    }
    __syncthreads();
    if(L.stamps && t == 0) L.stamps[7 * 8 + 2] = __builtin_readcyclecounter();
    // MixOutAmbiUp, :658-699 (devices above first order): the A-Format lines become B-Format rows here
    // (DoMixRow :619-634, with EarlyA2B / LateA2B), every row goes through its band splitter's
    // processHfScale (the in-place form, core/filters/splitter.cpp:98-131: one lane per row, the reference's operation order) and
    // the panning gains then pan AND upsample the rows.  The rows live in the pipelines' LDS, free by now.
    float *rowsLds[2] = {reinterpret_cast<float*>(&sm.pipe[0]), reinterpret_cast<float*>(&sm.pipe[1])};
    static_assert(sizeof(PipeLds) >= 8u * kLine * sizeof(float), "MixOutAmbiUp keeps its eight rows in a pipeline's LDS");
    if(L.upmix)
    {
        constexpr float kInvSqrt2 = 0.70710678118654752440f;
        constexpr float A2B[2][4][4] = {
            {{0.5f, 0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f, -0.5f}, {0.5f, -0.5f, -0.5f, 0.5f}, {0.5f, 0.5f, -0.5f, -0.5f}},
            {{0.5f, 0.5f, 0.5f, 0.5f}, {kInvSqrt2, -kInvSqrt2, 0.0f, 0.0f}, {0.0f, 0.0f, -kInvSqrt2, kInvSqrt2},
             {0.5f, 0.5f, -0.5f, -0.5f}}};
        const uint32_t npipes = oldRuns ? 2u : 1u;
        for(uint32_t q = 0; q < npipes; ++q)
        {
            const int p = q ? old : cur;
            const float *src[2] = {L.earlyOut + size_t(p) * 4u * kLine, L.lateOut + size_t(p) * 4u * kLine};
            for(uint32_t i = t; i < n; i += 256)
#pragma unroll
                for(int e = 0; e < 2; ++e)
                {
                    float in[4];
#pragma unroll
                    for(int k = 0; k < 4; ++k) in[k] = src[e][k * kLine + i];
#pragma unroll
                    for(int r = 0; r < 4; ++r)
                    {
                        float tmp = 0.0f;
#pragma unroll
                        for(int k = 0; k < 4; ++k)
                            if(fabsf(A2B[e][r][k]) > 0.00001f) tmp = tmp + in[k] * A2B[e][r][k];
                        rowsLds[q][(e * 4 + r) * kLine + i] = tmp;
                    }
                }
        }
        __syncthreads();
        if(t < 8u * npipes)
        {
            const uint32_t q = t >> 3, e = (t >> 2) & 1u, r = t & 3u;
            const int p = q ? old : cur;
            float *row = rowsLds[q] + (e * 4u + r) * kLine;
            float *st = L.state[p].split[e][r];
            const float hfscale = L.orderScale[r ? 1 : 0];
            const float apCoeff = L.splitCoeff, lpCoeff = L.splitCoeff * 0.5f + 0.5f;
            float lpZ1 = st[0], lpZ2 = st[1], apZ1 = st[2];
            for(uint32_t i = 0; i < n; ++i)
            {
                const float in = row[i];
                const float d0 = (in - lpZ1) * lpCoeff;
                const float lpY0 = lpZ1 + d0;
                lpZ1 = lpY0 + d0;
                const float d1 = (lpY0 - lpZ2) * lpCoeff;
                const float lpY1 = lpZ2 + d1;
                lpZ2 = lpY1 + d1;
                const float apY = in * apCoeff + apZ1;
                apZ1 = in - apY * apCoeff;
                row[i] = (apY - lpY1) * hfscale + lpY1;
            }
            st[0] = lpZ1; st[1] = lpZ2; st[2] = apZ1;
        }
        __syncthreads();
    }
    // FAST mode, several instances in the launch: an instance's eight (sixteen) inputs are summed per target line
    // BEFORE it takes its turn at the lines, and the turn is one add per output sample -- the instances' mix-outs, a
    // tenth of the kernel each, no longer queue behind each other.  (EXACT keeps the reference's term-by-term sum into
    // the line, which needs the line as the previous instance left it.)
    constexpr uint32_t kPreLines = 8;
    const bool preMix = ticket && L.fast && L.nlines <= kPreLines;
    float pre[kPreLines][4];
    auto takeTurn = [&]()
    {
        if(!ticket) return;
        if(t == 0)
            while(__hip_atomic_load(ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != order) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every wave drops what it may hold of the lines
    };
    if(!preMix) takeTurn();

    // MixOutPlain, :637-656: current pipeline first, then the old one (:1845,1878).  MixLine with
    // Counter == n (core/mixer/mixer_c.cpp:150-186): per (input, target line) the gain either
    // ramps over the whole block, or is constant, or the pair is skipped -- decided once here.
    {
        const float delta = 1.0f / float(n);
        const uint32_t npipes = oldRuns ? 2u : 1u;
        for(uint32_t e = t; e < npipes * 8u * L.nlines; e += 256)
        {
            const uint32_t q = e / (8u * L.nlines), r = e % (8u * L.nlines), j = r / L.nlines, c = r % L.nlines;
            const int p = q ? old : cur;
            const float curg = (j < 4) ? L.state[p].earlyCur[j][c] : L.state[p].lateCur[j - 4][c];
            const float tgt = (j < 4) ? L.pipe[p].early_gains_target[j][c] : L.pipe[p].late_gains_target[j - 4][c];
            const float step = (tgt - curg) * delta;
            MixGain g{curg, step, tgt, 0u};
            if(fabsf(step) > 1.1920928955078125e-07f) g.mode = 1u;
            else if(fabsf(tgt) > 0.00001f) g.mode = 2u;
            sm.mix[q][j][c] = g;
        }
        __syncthreads();
        float in[2][8][4];
#pragma unroll
        for(uint32_t q = 0; q < 2; ++q)
        {
            if(q >= npipes) break;
            const int p = q ? old : cur;
            const float *eo = L.earlyOut + size_t(p) * 4u * kLine, *lo = L.lateOut + size_t(p) * 4u * kLine;
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j)
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k)
                {
                    const uint32_t i = t + 256u * k;
                    if(L.upmix)
                    {
                        in[q][j][k] = rowsLds[q][j * kLine + (i < n ? i : 0u)];
                        in[q][4 + j][k] = rowsLds[q][(4u + j) * kLine + (i < n ? i : 0u)];
                        continue;
                    }
                    in[q][j][k] = eo[j * kLine + (i < n ? i : 0u)];
                    in[q][4 + j][k] = lo[j * kLine + (i < n ? i : 0u)];
                }
        }
        for(uint32_t c = 0; c < L.nlines; ++c)
        {
            float acc[4];
#pragma unroll
            for(uint32_t k = 0; k < 4; ++k) acc[k] = preMix ? 0.0f : L.outLines[c * kLine + ((t + 256u * k) < n ? t + 256u * k : 0u)];
#pragma unroll
            for(uint32_t q = 0; q < 2; ++q)
            {
                if(q >= npipes) break;
#pragma unroll
                for(uint32_t j = 0; j < 8; ++j)
                {
                    const MixGain g = sm.mix[q][j][c];
                    if(g.mode == 1u)
                    {
#pragma unroll
                        for(uint32_t k = 0; k < 4; ++k) acc[k] += in[q][j][k] * (g.cur + g.step * float(t + 256u * k));
                    }
                    else if(g.mode == 2u)
                    {
#pragma unroll
                        for(uint32_t k = 0; k < 4; ++k) acc[k] = acc[k] + in[q][j][k] * g.tgt;
                    }
                }
            }
            if(preMix)
            {
#pragma unroll
                for(uint32_t cc = 0; cc < kPreLines; ++cc)
                    if(cc == c) { pre[cc][0] = acc[0]; pre[cc][1] = acc[1]; pre[cc][2] = acc[2]; pre[cc][3] = acc[3]; }
                continue;
            }
#pragma unroll
            for(uint32_t k = 0; k < 4; ++k)
                if(t + 256u * k < n) L.outLines[c * kLine + t + 256u * k] = acc[k];
        }
        if(preMix)
        {
            takeTurn();
#pragma unroll
            for(uint32_t cc = 0; cc < kPreLines; ++cc)
            {
                if(cc >= L.nlines) break;
#pragma unroll
                for(uint32_t k = 0; k < 4; ++k)
                    if(t + 256u * k < n) L.outLines[cc * kLine + t + 256u * k] += pre[cc][k];
            }
        }
    }
    if(ticket) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this wave's line stores are out
    __syncthreads();
    if(ticket && t == 0)
        __hip_atomic_store(ticket, (order + 1u == count) ? 0u : order + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if(L.stamps && t == 0) L.stamps[7 * 8 + 3] = __builtin_readcyclecounter();
    // what process() leaves in the pipelines: Current = Target (MixLine with Counter == n), the
    // taps and the early coefficient handed over (:1579,1585,1759)
    for(int q = 0; q < (oldRuns ? 2 : 1); ++q)
    {
        const int p = q ? old : cur;
        oalgpu_reverb_pipeline &P = L.pipe[p];
        RvPipeState &S = L.state[p];
        for(uint32_t k = t; k < 4u * OALGPU_MAX_AMBI_CHANNELS; k += 256)
        {
            const uint32_t j = k / OALGPU_MAX_AMBI_CHANNELS, c = k % OALGPU_MAX_AMBI_CHANNELS;
            S.earlyCur[j][c] = P.early_gains_target[j][c];
            S.lateCur[j][c] = P.late_gains_target[j][c];
        }
        if(t < 4)
        {
            P.early_delay_tap[t][0] = P.early_delay_tap[t][1];
            P.late_delay_tap[t][0] = P.late_delay_tap[t][1];
        }
        if(t == 4) P.early_delay_coeff[0] = P.early_delay_coeff[1];
    }
}

__global__ void __launch_bounds__(256) ReverbProcessKernel(RvLayout L)
{
    __shared__ RvLds sm;
    ReverbProcessBody(L, sm, nullptr, 0u, 1u);
}

__global__ void __launch_bounds__(256) ReverbProcessBatchKernel(RvBatch B)
{
    __shared__ RvLds sm;
    ReverbProcessBody(B.r[blockIdx.x], sm, B.count > 1u ? B.ticket : nullptr, blockIdx.x, B.count);
}

// installs one pipeline's parameter block (passed by value) into device memory
__global__ void __launch_bounds__(256) ReverbInstallKernel(oalgpu_reverb_pipeline *dst, const oalgpu_reverb_pipeline src)
{
    const uint32_t *s = reinterpret_cast<const uint32_t*>(&src);
    uint32_t *d = reinterpret_cast<uint32_t*>(dst);
    for(uint32_t k = threadIdx.x; k < sizeof(oalgpu_reverb_pipeline) / 4u; k += blockDim.x) d[k] = s[k];
}

} // namespace

void LaunchReverbProcess(hipStream_t s, const RvLayout &L)
{ hipLaunchKernelGGL(ReverbProcessKernel, dim3(1), dim3(256), 0, s, L); }

void LaunchReverbBatch(hipStream_t s, const RvBatch &B)
{ hipLaunchKernelGGL(ReverbProcessBatchKernel, dim3(B.count), dim3(256), 0, s, B); }

void LaunchReverbInstall(hipStream_t s, oalgpu_reverb_pipeline *dst, const oalgpu_reverb_pipeline &src)
{ hipLaunchKernelGGL(ReverbInstallKernel, dim3(1), dim3(256), 0, s, dst, src); }

} // namespace oalgpu
