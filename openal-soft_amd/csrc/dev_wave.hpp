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

// ---- dual-ear FIR on the matrix pipe: the Toeplitz form of MixHrtf ------------------------------
// core/mixer/hrtfbase.h:17-89 computes, per ear, y[n] = sum_{j < IrSize} h[j] * x'[n - j] over one
// update (x' = the delayed, gain-ramped input).  Cut the output into blocks of 16 frames
// (n = 64 a + 16 b + r; a < 16 = matrix row, b < 4 = tile, r < 16 = matrix column) and the taps into
// j = 16 k + r - c (k < 5, c < 16): then
//     Y_b[a][r] = sum_{k, c} x'[64 a + 16 (b - k) + c] * h[16 k + r - c]
// is a dense (16 x 80) x (80 x 16) product per tile and ear -- 20 v_mfma_f32_16x16x4_f32 each, 64 of
// the 80 K-rows useful (the k = 0 and k = 4 Toeplitz tiles are complementary triangles).  The
// instruction is an exact k-ordered fp32 fma chain, so the result is fp32 like the packed-VALU form.
// Operands: lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]; with the K index of MFMA (k, t)
// chosen as c = 4 kk + t (kk = l >> 4) the four A fragments of one (b - k) are ONE ds_read_b128
// (x'[64 a + 16 (b - k) + 4 kk .. + 3]), 8 of them per ear, each reused by up to four tiles; the
// 20 B fragments per ear are h[16 k + (l & 15) - 4 kk - t] out of a zero-padded LDS copy of the HRIR.
// The 64-frame ring-out (frames 1024 + 16 b' + r) is a fifth tile whose rows are b' < 4:
//     Y_t[b'][r] = sum_{k > b', c} x'[1024 + 16 (b' - k) + c] * h[16 k + r - c]   (16 MFMAs per ear).
// acc[e][b]: lane l, element i = frame 64 (4 (l >> 4) + i) + 16 b + (l & 15) of ear e;
// acc[e][4]: lanes < 16, element i = frame 1024 + 16 i + l.
typedef float f4 __attribute__((ext_vector_type(4)));

template<int XS /* row stride of xp in floats */, int HPLEN>
__device__ __forceinline__ void FirMfma64(f4 (&acc)[2][5], const float *xp0, const float *xp1, const float *hp0,
    const float *hp1, uint32_t lane)
{
    const uint32_t a = lane & 15u, kk = lane >> 4;
#pragma unroll
    for(int e = 0; e < 2; ++e)
    {
        const float *xp = e ? xp1 : xp0;
        const float *hp = e ? hp1 : hp0;
        f4 A[8];
        const float *xa = xp + XS * a + 4u * kk;
#pragma unroll
        for(int d = 0; d < 8; ++d)                       // d = (b - k) + 4; rows shift by one from d = 4 on
            A[d] = *reinterpret_cast<const f4*>(xa + XS * (d >= 4 ? 1 : 0) + 16 * (d & 3));
        float B[5][4];
        const float *hb = hp + 16 + a - 4u * kk;         // (l & 15) doubles as the column r here
#pragma unroll
        for(int k = 0; k < 5; ++k)
#pragma unroll
            for(int t = 0; t < 4; ++t) B[k][t] = hb[16 * k - t];
        // the ring-out tile's inputs: row b' = l & 15 (< k), x'[1024 - 16 (k - b') + 4 kk ..]
        f4 T[4];
#pragma unroll
        for(int k = 1; k < 5; ++k)
        {
            const bool valid = a < uint32_t(k);
            const uint32_t i64 = valid ? 1088u - 16u * (uint32_t(k) - a) + 4u * kk : 1024u;
            const f4 v = *reinterpret_cast<const f4*>(xp + i64 + 4u * (i64 >> 6) /* = XS-strided rows */);
            T[k - 1] = valid ? v : f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        static_assert(XS == 68, "i64 + 4*(i64 >> 6) is the stride-68 row layout");
#pragma unroll
        for(int k = 0; k < 5; ++k)
#pragma unroll
            for(int t = 0; t < 4; ++t)
            {
#pragma unroll
                for(int b = 0; b < 4; ++b)
                    acc[e][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b - k + 4][t], B[k][t], acc[e][b], 0, 0, 0);
                if(k >= 1)
                    acc[e][4] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[k - 1][t], B[k][t], acc[e][4], 0, 0, 0);
            }
    }
}

// One wavefront's share of the same product when a 4-wavefront workgroup works on ONE voice
// (voice_block.hip): tile b (frames 64 a + 16 b + r) of both ears, and the k = b + 1 part of the
// ring-out tile.  Per ear the 20 + 4 MFMAs run as three independent accumulator chains (even /
// odd K steps of the main tile, ring-out), so none waits for the 40-cycle dependent latency.
// accM[e][p]: lane l, element i = partial p of frame 64 (4 (l >> 4) + i) + 16 b + (l & 15);
// accT[e]: lanes < 16, element i = this wave's part of frame 1024 + 16 i + l.
template<int XS>
__device__ __forceinline__ void FirMfmaTile(f4 (&accM)[2][2], f4 (&accT)[2], const float *xp0, const float *xp1,
    const float *hp0, const float *hp1, uint32_t lane, uint32_t b /* wave-uniform */)
{
    static_assert(XS == 68, "i64 + 4*(i64 >> 6) is the stride-68 row layout");
    const uint32_t a = lane & 15u, kk = lane >> 4;
    const uint32_t kt = b + 1u;                           // the ring-out part this wave computes
    const bool valid = a < kt;
    const uint32_t i64 = valid ? 1088u - 16u * (kt - a) + 4u * kk : 1024u;
#pragma unroll
    for(int e = 0; e < 2; ++e)
    {
        const float *xp = e ? xp1 : xp0;
        const float *hp = e ? hp1 : hp0;
        // every fragment of this ear is requested before the first MFMA (44 registers): one LDS round
        // trip per ear instead of one per K step
        f4 A[5];
        const float *xa = xp + XS * a + 4u * kk;
#pragma unroll
        for(int k = 0; k < 5; ++k)
        {
            const uint32_t d = b + 4u - uint32_t(k);      // (b - k) + 4
            A[k] = *reinterpret_cast<const f4*>(xa + (d >= 4u ? uint32_t(XS) : 0u) + 16u * (d & 3u));
        }
        f4 T = *reinterpret_cast<const f4*>(xp + i64 + 4u * (i64 >> 6));
        float B[5][4];
        const float *hb = hp + 16 + a - 4u * kk;
#pragma unroll
        for(int k = 0; k < 5; ++k)
#pragma unroll
            for(int t = 0; t < 4; ++t) B[k][t] = hb[16 * k - t];
        float Bt[4];                                      // = B[kt][.]: the ring-out part's taps (kt is wave-uniform)
        const float *hbt = hb + 16u * kt;
#pragma unroll
        for(int t = 0; t < 4; ++t) Bt[t] = hbt[-t];
        if(!valid) T = f4{0.0f, 0.0f, 0.0f, 0.0f};
        __builtin_amdgcn_sched_barrier(0);                // (the loads stay in front of the MFMAs)
#pragma unroll
        for(int k = 0; k < 5; ++k)
#pragma unroll
            for(int t = 0; t < 4; ++t)
            {
                accM[e][t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[k][t], B[k][t], accM[e][t & 1], 0, 0, 0);
                if(k == 2) accT[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t], Bt[t], accT[e], 0, 0, 0);
            }
        // one ear's fragment registers at a time: keep the other ear's loads behind this ear's MFMAs
        __builtin_amdgcn_sched_barrier(0);
    }
}

} // namespace oalgpu
