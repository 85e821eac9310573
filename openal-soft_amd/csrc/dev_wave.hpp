// Helpers of the wavefront-granular kernels (voice_wave.hip, post_wave.hip): packed-fp32 types,
// scalar-cache (constant address space) views of read-only HBM data, and the compiler-level
// fence used when the lanes of ONE wavefront exchange data through LDS.
#pragma once
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace oalgpu {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const f2 __attribute__((address_space(4))) cf2;        // scalar-cache (constant) loads
typedef const uint32_t __attribute__((address_space(4))) cu32;
typedef float f16 __attribute__((ext_vector_type(16)));
typedef const f16 __attribute__((address_space(4))) cf16;

// The wavefront kernels issue their LDS reads as single ds_read_b64 (2 LDS cycles per wave
// instruction, 256 B/clk).  Left to itself hipcc pairs them into ds_read2_b64, which the LDS
// services at half that rate and with 32-bank instead of 64-bank addressing -- measured on the
// resampler rows: 16 instead of 4 cycles per pair, SQ_LDS_BANK_CONFLICT = 1536 cycles per voice.
// The pairing is done by the SI load/store optimizer (disabled per kernel by this attribute;
// device pass only, the host pass does not know the feature) and by the IR load/store
// vectorizer (disabled for these translation units in the Makefile).
#if defined(__HIP_DEVICE_COMPILE__)
#define OALGPU_SINGLE_DS_OPS __attribute__((target("no-load-store-opt")))
#else
#define OALGPU_SINGLE_DS_OPS
#endif

__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { f2 r = {v, v}; return r; }

// Lanes of a wavefront exchange data through LDS in program order (the LDS executes one wave's
// operations in order); this only stops the COMPILER from moving memory operations across.
__device__ __forceinline__ void WaveSync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef const u4 __attribute__((address_space(4))) cu4;

__device__ __forceinline__ VoiceCtl LoadCtlScalar(const VoiceCtl *p)
{
    union { VoiceCtl c; u4 q[sizeof(VoiceCtl) / 16]; } u;
    cu4 *src = (cu4*)(uintptr_t)p;
#pragma unroll
    for(size_t k = 0; k < sizeof(VoiceCtl) / 16; ++k) u.q[k] = src[k];
    return u.c;
}


// ---- BandSplitter as a wavefront block scan ----------------------------------------------------
// BandSplitter::processHfScale / processScale (core/filters/splitter.cpp:65-97, :133-161) is a
// 3-state linear recurrence (lpZ1, lpZ2, apZ1); the wavefront runs it as a block scan -- each lane
// owns a run of samples, block start states by a 6-step Kogge-Stone scan over the lanes with powers
// of the run's transition matrix -- instead of n serial steps.  HFONLY: processHfScale (its first
// low-pass state update carries an extra coefficient, splitter.cpp:79, and there is no LF scale).
struct Sp3 { float a, b, c; };                              // (lpZ1, lpZ2, apZ1)

// one sample on state s; returns the output
template<bool HFONLY>
__device__ __forceinline__ float SplitStep(Sp3 &s, float x, float apCoeff, float lpCoeff, float hf, float lf)
{
    const float d0 = (x - s.a) * lpCoeff;
    const float lpY0 = s.a + d0;
    s.a = HFONLY ? __builtin_fmaf(d0, lpCoeff, lpY0) : lpY0 + d0;   // splitter.cpp:79 vs :146
    const float d1 = (lpY0 - s.b) * lpCoeff;
    const float lpY1 = s.b + d1;
    s.b = lpY1 + d1;
    const float apY = __builtin_fmaf(x, apCoeff, s.c);
    s.c = __builtin_fmaf(-apY, apCoeff, x);
    return __builtin_fmaf(apY - lpY1, hf, HFONLY ? lpY1 : lpY1 * lf);
}

struct Mat3 { Sp3 c0, c1, c2; };
__device__ __forceinline__ Sp3 MatVec3(const Mat3 &m, const Sp3 &v)
{
    Sp3 r;
    r.a = __builtin_fmaf(m.c2.a, v.c, __builtin_fmaf(m.c1.a, v.b, m.c0.a * v.a));
    r.b = __builtin_fmaf(m.c2.b, v.c, __builtin_fmaf(m.c1.b, v.b, m.c0.b * v.a));
    r.c = __builtin_fmaf(m.c2.c, v.c, __builtin_fmaf(m.c1.c, v.b, m.c0.c * v.a));
    return r;
}
__device__ __forceinline__ Mat3 MatMul3(const Mat3 &a, const Mat3 &b)
{ return Mat3{MatVec3(a, b.c0), MatVec3(a, b.c1), MatVec3(a, b.c2)}; }

// in place over buf[0..n); the lane stride (seg) is odd so the per-lane runs hit distinct banks
template<bool HFONLY>
__device__ __forceinline__ void SplitterScan(SplitterState &st, float *buf, uint32_t n, float hf, float lf, uint32_t lane)
{
    const float apCoeff = st.coeff, lpCoeff = st.coeff * 0.5f + 0.5f;
    const uint32_t seg = ((n + 63u) / 64u) | 1u;
    const uint32_t begin = lane * seg < n ? lane * seg : n;
    const uint32_t end = (begin + seg < n) ? begin + seg : n;
    Mat3 M{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for(uint32_t i = 0; i < seg; ++i)
    {
        SplitStep<HFONLY>(M.c0, 0.0f, apCoeff, lpCoeff, hf, lf); SplitStep<HFONLY>(M.c1, 0.0f, apCoeff, lpCoeff, hf, lf);
        SplitStep<HFONLY>(M.c2, 0.0f, apCoeff, lpCoeff, hf, lf);
    }
    Sp3 q{0, 0, 0};
    for(uint32_t i = begin; i < end; ++i) SplitStep<HFONLY>(q, buf[i], apCoeff, lpCoeff, hf, lf);
    Sp3 e = q;                                   // -> sum_{k<=l} M^(l-k) q_k
    Sp3 s0{st.lpZ1, st.lpZ2, st.apZ1};           // -> M^lane S_0
    Mat3 P = M;
#pragma unroll
    for(int step = 0; step < 6; ++step)
    {
        const int d = 1 << step;
        Sp3 o;
        o.a = __shfl_up(e.a, d); o.b = __shfl_up(e.b, d); o.c = __shfl_up(e.c, d);
        const Sp3 mo = MatVec3(P, o);
        if(int(lane) >= d) { e.a += mo.a; e.b += mo.b; e.c += mo.c; }
        const Sp3 ms = MatVec3(P, s0);
        if(lane & uint32_t(d)) s0 = ms;
        if(step < 5) P = MatMul3(P, P);
    }
    Sp3 prevE;
    prevE.a = __shfl_up(e.a, 1); prevE.b = __shfl_up(e.b, 1); prevE.c = __shfl_up(e.c, 1);
    Sp3 start = s0;
    if(lane > 0) { start.a += prevE.a; start.b += prevE.b; start.c += prevE.c; }
    for(uint32_t i = begin; i < end; ++i) buf[i] = SplitStep<HFONLY>(start, buf[i], apCoeff, lpCoeff, hf, lf);
    const int lastLane = int((n - 1u) / seg);
    st.lpZ1 = __shfl(start.a, lastLane); st.lpZ2 = __shfl(start.b, lastLane); st.apZ1 = __shfl(start.c, lastLane);
}

// The same scan of processHfScale with the structure of its recurrence used: the state (lpZ1, lpZ2, apZ1) evolves
// by a matrix that is lower triangular in (lpZ1, lpZ2) and decoupled in apZ1 -- [[p, 0, 0], [q, r, 0], [0, 0, s]] --
// so a run's transition and its powers are four numbers instead of nine ([[p,0],[q,r]]^2 = [[p^2,0],[q(p+r),r^2]]).
// Same arithmetic per sample (SplitStep), ~20 VGPRs instead of ~50: the HRTF post-process runs it beside the voice
// kernel, in the 32 registers per SIMD lane two voice wavefronts leave (post_wave.hip).
struct Tri3 { float p, q, r, s; };
__device__ __forceinline__ Sp3 TriVec(const Tri3 &m, const Sp3 &v)
{ return Sp3{m.p * v.a, __builtin_fmaf(m.q, v.a, m.r * v.b), m.s * v.c}; }
__device__ __forceinline__ void SplitterScanHfTri(SplitterState &st, float *buf, uint32_t n, float hf, uint32_t lane)
{
    const float apCoeff = st.coeff, lpCoeff = st.coeff * 0.5f + 0.5f;
    const uint32_t seg = ((n + 63u) / 64u) | 1u;
    const uint32_t begin = lane * seg < n ? lane * seg : n;
    const uint32_t end = (begin + seg < n) ? begin + seg : n;
    Sp3 ba{1.0f, 0.0f, 0.0f}, bb{0.0f, 1.0f, 0.0f}, bc{0.0f, 0.0f, 1.0f};
#pragma unroll 1
    for(uint32_t i = 0; i < seg; ++i)
    {
        SplitStep<true>(ba, 0.0f, apCoeff, lpCoeff, hf, 1.0f); SplitStep<true>(bb, 0.0f, apCoeff, lpCoeff, hf, 1.0f);
        SplitStep<true>(bc, 0.0f, apCoeff, lpCoeff, hf, 1.0f);
    }
    Tri3 P{ba.a, ba.b, bb.b, bc.c};
    Sp3 e{0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for(uint32_t i = begin; i < end; ++i) SplitStep<true>(e, buf[i], apCoeff, lpCoeff, hf, 1.0f);
    Sp3 s0{st.lpZ1, st.lpZ2, st.apZ1};
#pragma unroll 1
    for(int step = 0; step < 6; ++step)
    {
        const int d = 1 << step;
        Sp3 o;
        o.a = __shfl_up(e.a, d); o.b = __shfl_up(e.b, d); o.c = __shfl_up(e.c, d);
        const Sp3 mo = TriVec(P, o);
        if(int(lane) >= d) { e.a += mo.a; e.b += mo.b; e.c += mo.c; }
        const Sp3 ms = TriVec(P, s0);
        if(lane & uint32_t(d)) s0 = ms;
        P = Tri3{P.p * P.p, P.q * (P.p + P.r), P.r * P.r, P.s * P.s};
    }
    Sp3 start = s0;
    {
        const float pa = __shfl_up(e.a, 1), pb = __shfl_up(e.b, 1), pc = __shfl_up(e.c, 1);
        if(lane > 0) { start.a += pa; start.b += pb; start.c += pc; }
    }
#pragma unroll 1
    for(uint32_t i = begin; i < end; ++i) buf[i] = SplitStep<true>(start, buf[i], apCoeff, lpCoeff, hf, 1.0f);
    const int lastLane = int((n - 1u) / seg);
    st.lpZ1 = __shfl(start.a, lastLane); st.lpZ2 = __shfl(start.b, lastLane); st.apZ1 = __shfl(start.c, lastLane);
}

// The same scan once more for the one place where its latency is on a critical path (the fused HRTF post-process, post_wave.hip):
//   * the run's transition P = A^seg is data independent and the same for every channel: the HOST raises it (SplitterRunPowers,
//     api.hip) -- three of the five serial passes over a run gone;
//   * the scan over the lanes runs in the rows of 16 lanes by DPP row shifts with P, P^2, P^4, P^8, then three row carries with
//     P^((lane & 15) + 1) through v_readlane (the construction of ScanLinear2): no LDS round trips (__shfl_up is ds_bpermute).
__device__ __forceinline__ float DppRowShrF(float v, int d);
__device__ __forceinline__ Tri3 TriMul(const Tri3 &a, const Tri3 &b)       // powers of one matrix: the product commutes
{ return Tri3{a.p * b.p, __builtin_fmaf(a.q, b.p, a.r * b.q), a.r * b.r, a.s * b.s}; }
__device__ __forceinline__ void SplitterScanHfDpp(SplitterState &st, float *buf, uint32_t n, float hf, uint32_t lane, Tri3 P)
{
    const float apCoeff = st.coeff, lpCoeff = st.coeff * 0.5f + 0.5f;
    const uint32_t seg = ((n + 63u) / 64u) | 1u;
    const uint32_t begin = lane * seg < n ? lane * seg : n;
    const uint32_t end = (begin + seg < n) ? begin + seg : n;
    Sp3 e{0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for(uint32_t i = begin; i < end; ++i) SplitStep<true>(e, buf[i], apCoeff, lpCoeff, hf, 1.0f);
    const Sp3 s0{st.lpZ1, st.lpZ2, st.apZ1};
    {   // lane 0's run starts from the filter's state: the scan carries it along
        const Sp3 ms = TriVec(P, s0);
        if(lane == 0u) { e.a += ms.a; e.b += ms.b; e.c += ms.c; }
    }
    Tri3 pw = P, Q{1.0f, 0.0f, 1.0f, 1.0f};
    const uint32_t exp = (lane & 15u) + 1u;
#pragma unroll
    for(int step = 0; step < 4; ++step)
    {
        const int d = 1 << step;
        const Sp3 o{DppRowShrF(e.a, d), DppRowShrF(e.b, d), DppRowShrF(e.c, d)};     // 0 where the source lies in another row
        const Sp3 mo = TriVec(pw, o);
        e.a += mo.a; e.b += mo.b; e.c += mo.c;
        const Tri3 qn = TriMul(pw, Q);
        if(exp & uint32_t(d)) Q = qn;
        pw = TriMul(pw, pw);
    }
    if(exp == 16u) Q = pw;                              // P^16
#pragma unroll
    for(int r = 1; r < 4; ++r)
    {
        const Sp3 c{__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.a), 16 * r - 1)),
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.b), 16 * r - 1)),
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.c), 16 * r - 1))};
        const Sp3 mc = TriVec(Q, c);
        if((lane >> 4) == uint32_t(r)) { e.a += mc.a; e.b += mc.b; e.c += mc.c; }
    }
    // the run's start state: the end state of the lane before (wave_shr:1; lane 0 keeps the filter's state)
    Sp3 start;
    start.a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, s0.a), __builtin_bit_cast(int, e.a), 0x138, 0xF, 0xF, false));
    start.b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, s0.b), __builtin_bit_cast(int, e.b), 0x138, 0xF, 0xF, false));
    start.c = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, s0.c), __builtin_bit_cast(int, e.c), 0x138, 0xF, 0xF, false));
#pragma unroll 1
    for(uint32_t i = begin; i < end; ++i) buf[i] = SplitStep<true>(start, buf[i], apCoeff, lpCoeff, hf, 1.0f);
    const int lastLane = int((n - 1u) / seg);
    st.lpZ1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, start.a), lastLane));
    st.lpZ2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, start.b), lastLane));
    st.apZ1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, start.c), lastLane));
}

// ---- dual-ear FIR, packed over the ears -------------------------------------------------------
// acc[r] = (L,R) of output frame R*lane + r.  xw points at the x' entry of the lane's first
// frame; co16[b] = taps 8b..8b+7 as (Coeffs[j][0], Coeffs[j][1]) pairs, one s_load_dwordx16
// through the scalar cache each, fetched one block ahead of its use.  X is f2 (per-ear inputs,
// MixHrtf) or float (same input for both ears, MixDirectHrtf).
template<int R, int TAPS, typename X>
__device__ __forceinline__ void FirMainPk(f2 (&acc)[R], const X *xw, cf16 *co16)
{
    static_assert(TAPS % 8 == 0, "taps come in blocks of 8");
    X w[R + 3];
#pragma unroll
    for(int k = 0; k < R + 3; ++k) w[k] = xw[k - 3];
    // the four window entries block b+1 shifts in are read two blocks ahead of their first use
    X wn[4];
#pragma unroll
    for(int k = 0; k < 4; ++k) wn[k] = (TAPS > 4) ? xw[-4 - 3 + k] : w[k];
    f16 cnext = co16[0];
#pragma unroll
    for(int b8 = 0; b8 < TAPS / 8; ++b8)
    {
        const f16 c16 = cnext;
        if(b8 + 1 < TAPS / 8) cnext = co16[b8 + 1];
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
            const int b = 2 * b8 + h;
            X wnn[4];
#pragma unroll
            for(int k = 0; k < 4; ++k) wnn[k] = (b + 2 < TAPS / 4) ? xw[-4 * (b + 2) - 3 + k] : wn[k];
#pragma unroll
            for(int jj = 0; jj < 4; ++jj)
            {
                const f2 c = {c16[(4 * h + jj) * 2], c16[(4 * h + jj) * 2 + 1]};
#pragma unroll
                for(int r = 0; r < R; ++r)
                {
                    if constexpr(sizeof(X) == sizeof(f2)) acc[r] = pkfma(c, w[r + 3 - jj], acc[r]);
                    else acc[r] = pkfma(c, splat(w[r + 3 - jj]), acc[r]);
                }
            }
            if(b + 1 < TAPS / 4)
            {
#pragma unroll
                for(int k = R + 2; k >= 4; --k) w[k] = w[k - 4];
#pragma unroll
                for(int k = 0; k < 4; ++k) { w[k] = wn[k]; wn[k] = wnn[k]; }
            }
        }
    }
}

// ---- dual-ear FIR on the matrix pipe: the Toeplitz form of MixHrtf in split half precision -----
// core/mixer/hrtfbase.h:17-89 computes, per ear, y[m] = sum_{t < IrSize} h[t] * x'[m - t] over one update
// (x' = the delayed, gain-ramped input).  Cut the output into blocks of 16 frames, m = 16 n + i: then
//     Y[i][n] = sum_{k < 96} T[i][k] * X[k][n],   T[i][k] = h[i + 64 - k],   X[k][n] = x'[16 n - 64 + k]
// is a (16 x 96) x (96 x 16) product per 16 blocks (= 256 frames) and ear: three v_mfma_f32_16x16x32_f16
// (64 of the 96 K-rows carry taps).  gfx950 has no reduced-precision fp32 MFMA (no xf32), and the fp32-input
// form runs at the vector rate -- round 2 measured it slower than packed VALU FMAs -- but the f16 form runs
// at 16x that rate, so both operands are split into two halves each, v = hi + lo (hi = the value rounded
// toward zero to f16, lo = the remainder rounded to f16; a power-of-two scale per voice keeps both inside
// f16's range), and the product is taken as hi*hi + hi*lo + lo*hi: 22 of fp32's 24 significant bits per
// factor, products exact and accumulated in fp32 inside the instruction.  Measured against double precision
// the result is as close as the reference's own serial fp32 sum (4e-7 of the block's maximum;
// tests/test_tolerance_model.py).  9 MFMAs per tile and ear, 90 per voice (5 tiles: 4 x 256 frames + the
// 64-frame ring-out) = 1.4 K cycles of the matrix pipe against 8.7 K cycles of packed FMAs.
// Operands (the K slot of element idx of lane l is 8 (l >> 4) + idx for A and B alike):
//   A: lane l, row i = l & 15: T[i][32 c + 8 g + idx] = r[16 - i + 32 c + 8 g + idx] with r[u] = h[80 - u]
//      (zero outside u in [17, 80]): eight consecutive halves of the REVERSED response, read as five dwords
//      and shifted down by one half (v_alignbyte_b32) in the lanes whose offset is odd;
//   B: lane l, column j = l & 15: x'[16 (16 T + j) - 64 + 32 c + 8 g + idx]: one aligned ds_read_b128.
// acc[e][T]: lane l, element r = frame 16 (16 T + (l & 15)) + 4 (l >> 4) + r of ear e (T = 4: the ring-out,
// columns < 4).
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int kXhHalves = 64 + 1024 + 80;              // x' halves per ear and part: frames -64 .. 1103
constexpr int kXhDw = kXhHalves / 2;
constexpr int kHrHalves = 120;                         // r[u], u < 112, and the fifth dword of the last fragment
constexpr int kHrDw = kHrHalves / 2;
static_assert(kXhDw % 4 == 0 && kHrDw % 4 == 0, "16-byte rows");

// LDS images (WaveLds, wave_common.hpp): xh[ear][hi | lo][kXhDw] = x' * scale, two frames per dword;
// hr[ear][hi | lo][kHrDw] = r[u] * scale
// the power-of-two scale that brings a block whose largest magnitude has the bit pattern `maxBits` into
// [2^14, 2^15), and its inverse (both within 2^+-60, so that a product of two inverses stays finite)
__device__ __forceinline__ void HalfScale(uint32_t maxBits, float &scale, float &inv)
{
    int32_t sb = 268 - int32_t(maxBits >> 23);
    sb = sb < 67 ? 67 : (sb > 187 ? 187 : sb);
    scale = __builtin_bit_cast(float, uint32_t(sb) << 23);
    inv = __builtin_bit_cast(float, uint32_t(254 - sb) << 23);
}
// (v0, v1) -> the packed f16 pair of their leading halves and of the remainders
__device__ __forceinline__ void SplitHalf2(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    const auto h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    hi = __builtin_bit_cast(uint32_t, h);
    // (v - float(h) as fma(float(h), -1, v): the product is exact, and the compiler has v_fma_mix_f32 for the form)
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf(float(h[0]), -1.0f, v0), __builtin_fmaf(float(h[1]), -1.0f, v1)));
}
// maximum over the wavefront of non-negative floats given as bit patterns, by DPP (no LDS round trips)
__device__ __forceinline__ uint32_t WaveMaxBits(uint32_t v)
{
#define OALGPU_DPP_MAX(ctrl) do { const uint32_t o = uint32_t(__builtin_amdgcn_update_dpp(0, int(v), ctrl, 0xF, 0xF, true)); v = o > v ? o : v; } while(0)
    OALGPU_DPP_MAX(0xB1);     // quad_perm [1,0,3,2]
    OALGPU_DPP_MAX(0x4E);     // quad_perm [2,3,0,1]
    OALGPU_DPP_MAX(0x141);    // row_half_mirror
    OALGPU_DPP_MAX(0x140);    // row_mirror: every lane of a row of 16 holds the row's maximum
#undef OALGPU_DPP_MAX
    const uint32_t a = uint32_t(__builtin_amdgcn_readlane(int(v), 0)), b = uint32_t(__builtin_amdgcn_readlane(int(v), 16));
    const uint32_t c = uint32_t(__builtin_amdgcn_readlane(int(v), 32)), d = uint32_t(__builtin_amdgcn_readlane(int(v), 48));
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// ---- linear recurrences over the lanes of a wavefront (the biquad block scans of the voice kernels and of the EAX reverb) ----
constexpr int kBqSeg = 17;                    // ceil(1024 / 64) | 1
struct S2 { float a, b; };

__device__ __forceinline__ float BqStep(S2 &s, float x, float b0, float b1, float b2, float a1, float a2)
{
    const float y = __builtin_fmaf(x, b0, s.a);
    s.a = __builtin_fmaf(x, b1, __builtin_fmaf(-y, a1, s.b));
    s.b = __builtin_fmaf(x, b2, -y * a2);
    return y;
}
__device__ __forceinline__ S2 Mv2(S2 c0, S2 c1, S2 v)          // [c0 c1] * v
{ return S2{__builtin_fmaf(c1.a, v.b, c0.a * v.a), __builtin_fmaf(c1.b, v.b, c0.b * v.a)}; }

// E_l = sum_{k <= l} M^(l - k) e_k over the 64 lanes of a wavefront, M = [m0 m1] (columns): a Kogge-Stone scan with M,
// M^2, M^4, M^8 inside the rows of 16 lanes by DPP row shifts (no LDS round trips: ds_bpermute costs ~150 cycles a piece
// and the scan's twelve were a third of the filter's time), then three row carries: row r takes M^((lane & 15) + 1)
// times the finished total of row r - 1 (lane 16 r - 1, read with v_readlane).
__device__ __forceinline__ float DppRowShr(float v, int d)
{
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch(d)
    {
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); break;
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); break;
    }
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float DppRowShrF(float v, int d) { return DppRowShr(v, d); }
// the in-row part alone: E_l = sum over the lanes k <= l OF THE SAME ROW of 16 -- four independent scans per wavefront
// (the EAX reverb runs one of its four lines per row); q0/q1 return M^((lane & 15) + 1) for a caller that chains rows
__device__ __forceinline__ S2 ScanLinear2Row(S2 e, S2 m0, S2 m1, uint32_t lane, S2 &q0, S2 &q1)
{
    S2 p0 = m0, p1 = m1;                               // M^(2^step)
    q0 = S2{1.0f, 0.0f}; q1 = S2{0.0f, 1.0f};          // -> M^((lane & 15) + 1)
    const uint32_t exp = (lane & 15u) + 1u;
#pragma unroll
    for(int step = 0; step < 4; ++step)
    {
        const int d = 1 << step;
        const S2 o{DppRowShr(e.a, d), DppRowShr(e.b, d)};     // lanes whose source lies in another row receive 0
        const S2 mo = Mv2(p0, p1, o);
        e.a += mo.a; e.b += mo.b;
        const S2 r0 = Mv2(p0, p1, q0), r1 = Mv2(p0, p1, q1);
        if(exp & uint32_t(d)) { q0 = r0; q1 = r1; }
        const S2 n0 = Mv2(p0, p1, p0), n1 = Mv2(p0, p1, p1); p0 = n0; p1 = n1;
    }
    if(exp == 16u) { q0 = p0; q1 = p1; }               // M^16
    return e;
}
__device__ __forceinline__ S2 ScanLinear2(S2 e, S2 m0, S2 m1, uint32_t lane)
{
    S2 q0, q1;
    e = ScanLinear2Row(e, m0, m1, lane, q0, q1);
#pragma unroll
    for(int r = 1; r < 4; ++r)
    {
        const S2 c{__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.a), 16 * r - 1)),
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.b), 16 * r - 1))};
        const S2 mc = Mv2(q0, q1, c);
        if((lane >> 4) == uint32_t(r)) { e.a += mc.a; e.b += mc.b; }
    }
    return e;
}


// AHEAD: the tiles' B fragments are fetched one tile ahead of their use (24 more registers; the kernels that also hold
// line accumulators fetch them when they need them)
template<int TILES = 5, bool LOW8 = false, bool AHEAD = true>
__device__ __forceinline__ void FirMfmaH(f4 (&acc)[2][5], const uint32_t (&xh)[2][2][kXhDw], const uint32_t (&hr)[2][2][kHrDw],
    float inv, uint32_t lane)
{
    const uint32_t i = lane & 15u, g = lane >> 4;
    const uint32_t u0 = 16u - i + 8u * g, shift = (u0 & 1u) * 2u, ud = u0 >> 1;
    const f4 vinv = {inv, inv, inv, inv};
#pragma unroll
    for(int e = 0; e < 2; ++e)
    {   // one ear at a time: 6 A fragments live across the ear's five tiles, 6 B fragments per tile
        if constexpr (!AHEAD) __builtin_amdgcn_sched_barrier(0);      // (nor the next ear's fragments over this ear's tiles)
        h8 A[2][3];
#pragma unroll
        for(int s = 0; s < 2; ++s)
#pragma unroll
            for(int c = 0; c < 3; ++c)
            {   // eight halves from half offset u0 + 32 c: five dwords, shifted down by one half when u0 is odd
                const uint32_t *p = hr[e][s] + ud + 16 * c;
                const uint32_t q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
                const u4 q = {__builtin_amdgcn_alignbyte(q1, q0, shift), __builtin_amdgcn_alignbyte(q2, q1, shift),
                    __builtin_amdgcn_alignbyte(q3, q2, shift), __builtin_amdgcn_alignbyte(q4, q3, shift)};
                A[s][c] = __builtin_bit_cast(h8, q);
            }
        // the tiles' B fragments one tile ahead of their use; per tile three independent accumulator chains
        // (lo*hi, hi*lo, hi*hi: three MFMAs each) so that the nine issue back to back
        auto loadB = [&](h8 (&B)[2][3], int T)
        {
            const uint32_t j = (T == 4 && i > 3u) ? 3u : i;          // the ring-out tile has four columns
#pragma unroll
            for(int s = 0; s < 2; ++s)
            {
                const u4 *p = reinterpret_cast<const u4*>(xh[e][s]) + 32 * T + 2u * j + g;
#pragma unroll
                for(int c = 0; c < 3; ++c) B[s][c] = __builtin_bit_cast(h8, p[4 * c]);
            }
        };
        h8 B[2][3], Bn[AHEAD ? 2 : 1][AHEAD ? 3 : 1];
        loadB(B, 0);
#pragma unroll
        for(int T = 0; T < TILES; ++T)
        {
            if constexpr (AHEAD) { if(T + 1 < TILES) loadB(Bn, T + 1); }
            f4 ta = {0.0f, 0.0f, 0.0f, 0.0f}, tb = ta, tc = ta;
#pragma unroll
            for(int c = 0; c < 3; ++c)
            {
                ta = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1][c], B[0][c], ta, 0, 0, 0);
                tb = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0][c], B[1][c], tb, 0, 0, 0);
                tc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0][c], B[0][c], tc, 0, 0, 0);
            }
            const f4 sum = __builtin_elementwise_fma((ta + tb) + tc, vinv, acc[e][T]);
            if(!LOW8 || i < 8u) acc[e][T] = sum;      // LOW8: only the columns of frames 0..127 hold this product
            if constexpr (AHEAD)
            {
            if(T + 1 < TILES)
            {
#pragma unroll
                for(int s = 0; s < 2; ++s)
#pragma unroll
                    for(int c = 0; c < 3; ++c) B[s][c] = Bn[s][c];
            }
            }
            else if(T + 1 < TILES)
            {   // (and the scheduler must not pull the next tile's reads up over this tile's products either)
                __builtin_amdgcn_sched_barrier(0);
                loadB(B, T + 1);
            }
        }
    }
}

} // namespace oalgpu
