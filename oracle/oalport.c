/* oracle/oalport.c -- plain-C restatement of the reference's per-voice mixing path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the CPU oracle the GPU parity tests check
 * against when the compiled reference (oracle/_ref/liboalref.so) is not present; it is
 * itself pinned against that compiled reference by tests/test_oracle_pin.py and against
 * the golden vectors under tests/golden/ (generated from the compiled reference by
 * tests/golden/make_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product never links or calls it.
 *
 * Every function cites the reference file:line (relative to kcat/openal-soft @ 2026-08-21)
 * it restates.  Arithmetic is IEEE binary32 with FTZ+DAZ (core/fpu_ctrl.cpp:32-50) and no
 * FMA contraction (the reference is built without -mfma; this file is built with
 * -ffp-contract=off -mfpmath=sse).
 *
 * "simd" mode (default on) reproduces the summation order of the SSE/SSE2/SSE4.1 variants
 * the reference auto-selects on x86-64 (core/mixer/mixer_sse*.cpp); simd off reproduces
 * the *_C variants (core/mixer/mixer_c.cpp).
 */
#include "oalref.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

#define FRAC_BITS 16
#define FRAC_ONE (1u << FRAC_BITS)
#define FRAC_MASK (FRAC_ONE - 1u)
#define LINE OAL_BUFFER_LINE_SIZE
#define MAX_PAD OAL_MAX_RESAMPLER_PADDING
#define MAX_EDGE OAL_MAX_RESAMPLER_EDGE
#define HIST OAL_HRTF_HISTORY_LENGTH
#define HRIR_LEN OAL_HRIR_LENGTH
#define BSINC_SCALE_COUNT 16
#define BSINC_PHASE_COUNT 32
#define BSINC_PHASE_DIFF_BITS (FRAC_BITS - 5)
#define CUBIC_PHASE_COUNT 32
#define CUBIC_PHASE_DIFF_BITS (FRAC_BITS - 5)
#define GAIN_SILENCE_THRESHOLD 0.00001f /* core/mixer/defs.h:28 */
#define RESAMPLE_DATA_SIZE (LINE + 256 + MAX_PAD) /* DeviceBase::mResampleData, core/device.h:282 */
#define PI_D 3.14159265358979323846

static int g_simd = 1;

const char *oal_kind(void) { return "port"; }
void oal_set_simd(int enable) { g_simd = enable != 0; }

/* FPUCtl (core/fpu_ctrl.cpp:32-50): flush-to-zero + denormals-are-zero while mixing. */
static unsigned fpu_enter(void)
{
    unsigned old = _mm_getcsr();
    _mm_setcsr(old | 0x8000u /* FTZ */ | 0x0040u /* DAZ */);
    return old;
}
static void fpu_leave(unsigned old) { _mm_setcsr(old); }

static float lerpf(float a, float b, float mu) { return a + (b - a) * mu; } /* alnumeric.h:115 */
/* fastf2u: cvtss2si, round-to-nearest-even (alnumeric.h:163-189) */
static unsigned fastf2u(float f) { return (unsigned)_mm_cvt_ss2si(_mm_set_ss(f)); }
/* float2uint: truncation with clamping (alnumeric.h:223-240) */
static unsigned float2uint(float f)
{
    int conv_i; memcpy(&conv_i, &f, 4);
    unsigned mask = (unsigned)(conv_i >> 31) ^ 0xffffffffu;
    int shift = ((conv_i >> 23) & 0xff) - (127 + 23);
    if(shift < -23) return 0;
    if(shift > 8) return 0xffffffffu & mask;
    unsigned mant = ((unsigned)conv_i & 0x7fffffu) | 0x800000u;
    if(shift < 0) return (mant >> -shift) & mask;
    return (mant << shift) & mask;
}

/* ======================================================================== *
 * Tables
 * ======================================================================== */

/* ---- bsinc: core/bsinc_tables.cpp:22-371 ---- */
static double bessel_i0(double x) /* cyl_bessel_i, bsinc_tables.cpp:33-57 */
{
    const double x2 = x / 2.0;
    double term = 1.0, sum = 1.0, last_sum;
    int k = 1;
    do {
        const double y = x2 / k;
        ++k;
        last_sum = sum;
        term *= y * y;
        sum += term;
    } while(sum != last_sum);
    return sum;
}
static double sinc_(double x) /* bsinc_tables.cpp:64-69 */
{
    if(!(x > 2.2204460492503131e-16 || x < -2.2204460492503131e-16)) return 1.0;
    return sin(PI_D * x) / (PI_D * x);
}
static double kaiser_(double beta, double k, double besseli_0_beta) /* :85-90 */
{
    if(!(k >= -1.0 && k <= 1.0)) return 0.0;
    return bessel_i0(beta * sqrt(1.0 - k * k)) / besseli_0_beta;
}
static double kaiser_width(double rejection, double order) /* :95-101 */
{
    if(rejection > 21.19) return (rejection - 7.95) / (2.285 * PI_D * 2.0 * order);
    return 5.79 / (PI_D * 2.0) / order;
}
static double kaiser_beta(double rejection) /* :104-111 */
{
    if(rejection > 50.0) return 0.1102 * (rejection - 8.7);
    if(rejection >= 21.0) return 0.5842 * pow(rejection - 21.0, 0.4) + 0.07886 * (rejection - 21.0);
    return 0.0;
}
static double lerp_d(double a, double b, double t) /* std::lerp as libstdc++ 11 implements it */
{
    if((a <= 0 && b >= 0) || (a >= 0 && b <= 0)) return t * b + (1 - t) * a;
    if(t == 1) return b;
    const double x = a + t * (b - a);
    return ((t > 1) == (b > a)) ? (b < x ? x : b) : (b > x ? x : b);
}

typedef struct bsinc_hdr { /* BSincHeader, :114-145 */
    double beta, scaleBase, scaleLimit;
    double a[BSINC_SCALE_COUNT];
    unsigned m[BSINC_SCALE_COUNT];
    size_t total_size;
} bsinc_hdr;

static void bsinc_hdr_init(bsinc_hdr *h, double rejection, double order, double maxScale)
{
    h->beta = kaiser_beta(rejection);
    h->scaleBase = kaiser_width(rejection, order) / 2.0;
    h->scaleLimit = 1.0 / maxScale;
    h->total_size = 0;
    const double base_a = (order + 1.0) / 2.0;
    for(unsigned si = 0; si < BSINC_SCALE_COUNT; ++si)
    {
        const double scale = lerp_d(h->scaleBase, 1.0, (double)(si + 1u) / (double)BSINC_SCALE_COUNT);
        double a = base_a / scale;
        if(base_a * maxScale < a) a = base_a * maxScale;
        h->a[si] = a;
        unsigned a_ = (unsigned)a; /* truncation, then ceil fix-up */
        a_ += ((double)a_ != a) ? 1u : 0u;
        h->m[si] = a_ * 2u;
        h->total_size += 4u * BSINC_PHASE_COUNT * ((h->m[si] + 3u) & ~3u);
    }
}

typedef struct bsinc_tab {
    int ready;
    bsinc_hdr hdr;
    float *tab;
    oal_bsinc_table pub;
} bsinc_tab;
static bsinc_tab g_bsinc[3];

static void bsinc_build(bsinc_tab *t, double rejection, double order, double maxScale)
{
    bsinc_hdr *hdr = &t->hdr;
    bsinc_hdr_init(hdr, rejection, order, maxScale);
    const unsigned pts_max = (hdr->m[0] + 3u) & ~3u; /* BSincPointsMax, :155 */
    /* filter[si][pi][i] in f64 */
    double *filter = (double *)calloc((size_t)BSINC_SCALE_COUNT * BSINC_PHASE_COUNT * pts_max, sizeof(double));
#define FIL(si, pi, i) filter[((size_t)(si) * BSINC_PHASE_COUNT + (pi)) * pts_max + (i)]
    const double besseli_0_beta = bessel_i0(hdr->beta);
    for(unsigned si = 0; si < BSINC_SCALE_COUNT; ++si) /* :166-250 */
    {
        const double a = hdr->a[si];
        const unsigned m = hdr->m[si];
        const double l = floor(m * 0.5) - 1.0;
        const size_t o = (pts_max - m) / 2u;
        const double scale = lerp_d(hdr->scaleBase, 1.0, (double)(si + 1u) / (double)BSINC_SCALE_COUNT);
        const double max_cutoff = (0.5 - hdr->scaleBase) * scale;
        const double width = hdr->scaleBase * (hdr->scaleLimit > scale ? hdr->scaleLimit : scale);
        const double c2a = (scale - width) * 0.5;
        const double cutoff2 = (max_cutoff < c2a ? max_cutoff : c2a) * 2.0;
        for(unsigned pi = 0; pi < BSINC_PHASE_COUNT; ++pi)
        {
            const double phase = l + (double)pi / BSINC_PHASE_COUNT;
            for(unsigned i = 0; i < m; ++i)
            {
                const double x = (double)i - phase;
                FIL(si, pi, o + i) = kaiser_(hdr->beta, x / a, besseli_0_beta) * cutoff2 * sinc_(cutoff2 * x);
            }
        }
    }
    t->tab = (float *)calloc(hdr->total_size, sizeof(float));
    size_t idx = 0;
    for(unsigned si = 0; si < BSINC_SCALE_COUNT; ++si) /* :254-340 */
    {
        const size_t m = ((size_t)hdr->m[si] + 3u) & ~(size_t)3u;
        const size_t o = (pts_max - m) / 2u;
        for(unsigned pi = 0; pi < BSINC_PHASE_COUNT; ++pi)
        {
            for(size_t i = 0; i < m; ++i) t->tab[idx++] = (float)FIL(si, pi, o + i);
            if(pi < BSINC_PHASE_COUNT - 1)
            {
                for(size_t i = 0; i < m; ++i)
                    t->tab[idx++] = (float)(FIL(si, pi + 1, o + i) - FIL(si, pi, o + i));
            }
            else
            {
                t->tab[idx++] = (float)(0.0 - FIL(si, pi, o));
                for(size_t i = 1; i < m; ++i)
                    t->tab[idx++] = (float)(FIL(si, 0, o + i - 1) - FIL(si, pi, o + i));
            }
        }
        if(si < BSINC_SCALE_COUNT - 1)
        {
            for(unsigned pi = 0; pi < BSINC_PHASE_COUNT; ++pi)
            {
                for(size_t i = 0; i < m; ++i)
                    t->tab[idx++] = (float)(FIL(si + 1, pi, o + i) - FIL(si, pi, o + i));
                if(pi < BSINC_PHASE_COUNT - 1)
                {
                    for(size_t i = 0; i < m; ++i)
                        t->tab[idx++] = (float)((FIL(si + 1, pi + 1, o + i) - FIL(si + 1, pi, o + i))
                            - (FIL(si, pi + 1, o + i) - FIL(si, pi, o + i)));
                }
                else
                {
                    t->tab[idx++] = (float)((0.0 - FIL(si + 1, pi, o)) - (0.0 - FIL(si, pi, o)));
                    for(size_t i = 1; i < m; ++i)
                        t->tab[idx++] = (float)((FIL(si + 1, 0, o + i - 1) - FIL(si + 1, pi, o + i))
                            - (FIL(si, 0, o + i - 1) - FIL(si, pi, o + i)));
                }
            }
        }
        else
            idx += (size_t)BSINC_PHASE_COUNT * m * 2; /* zero-filled */
    }
#undef FIL
    free(filter);
    /* GenerateBSincTable, :352-365 */
    t->pub.scaleBase = (float)hdr->scaleBase;
    t->pub.scaleRange = (float)(1.0 / (1.0 - hdr->scaleBase));
    for(unsigned i = 0; i < BSINC_SCALE_COUNT; ++i) t->pub.m[i] = (hdr->m[i] + 3u) & ~3u;
    t->pub.filterOffset[0] = 0;
    for(unsigned i = 1; i < BSINC_SCALE_COUNT; ++i)
        t->pub.filterOffset[i] = t->pub.filterOffset[i - 1] + t->pub.m[i - 1] * 4u * BSINC_PHASE_COUNT;
    t->pub.tab = t->tab;
    t->pub.tablen = hdr->total_size;
    t->ready = 1;
}

static const oal_bsinc_table *bsinc_get(int which)
{
    int k = (which == 12) ? 0 : (which == 24) ? 1 : (which == 48) ? 2 : -1;
    if(k < 0) return NULL;
    if(!g_bsinc[k].ready)
    {
        /* bsinc_tables.cpp:151-157: {60dB,11th order,x2} {60,23,2} {80,47,1} */
        if(k == 0) bsinc_build(&g_bsinc[0], 60, 11, 2);
        if(k == 1) bsinc_build(&g_bsinc[1], 60, 23, 2);
        if(k == 2) bsinc_build(&g_bsinc[2], 80, 47, 1);
    }
    return &g_bsinc[k].pub;
}

int oal_bsinc_table_get(int which, oal_bsinc_table *out)
{
    const oal_bsinc_table *t = bsinc_get(which);
    if(!t) return -1;
    *out = *t;
    return 0;
}

/* ---- cubic: core/cubic_tables.cpp:24-106 ---- */
static float g_cubic[2][CUBIC_PHASE_COUNT][8];
static int g_cubic_ready;

static double gauss_coeff(double idx) /* GetCoeff, cubic_tables.cpp:26-34 */
{
    const double k = 0.5 + idx;
    if(k > 512.0) return 0.0;
    const double s = sin(PI_D * 1.280 / 1024.0 * k);
    const double t = (cos(PI_D * 2.000 / 1023.0 * k) - 1.0) * 0.50;
    const double u = (cos(PI_D * 4.000 / 1023.0 * k) - 1.0) * 0.08;
    return s * (t + u + 1.0) / k;
}

static void cubic_deltas(float (*tab)[8]) /* cubic_tables.cpp:58-71,92-104 */
{
    for(unsigned pi = 0; pi < CUBIC_PHASE_COUNT - 1; ++pi)
        for(int k = 0; k < 4; ++k) tab[pi][4 + k] = tab[pi + 1][k] - tab[pi][k];
    const unsigned pi = CUBIC_PHASE_COUNT - 1;
    tab[pi][4] = 0.0f - tab[pi][0];
    tab[pi][5] = tab[0][0] - tab[pi][1];
    tab[pi][6] = tab[0][1] - tab[pi][2];
    tab[pi][7] = tab[0][2] - tab[pi][3];
}

static void cubic_build(void)
{
    if(g_cubic_ready) return;
    /* SplineTable, cubic_tables.cpp:74-105 */
    const double third = 1.0 / 3.0, sixth = 1.0 / 6.0;
    for(unsigned pi = 0; pi < CUBIC_PHASE_COUNT; ++pi)
    {
        const double mu = (double)pi / (double)CUBIC_PHASE_COUNT;
        const double mu2 = mu * mu, mu3 = mu * mu2;
        g_cubic[0][pi][0] = (float)(-third * mu + 0.5 * mu2 - sixth * mu3);
        g_cubic[0][pi][1] = (float)(1.0 - 0.5 * mu - mu2 + 0.5 * mu3);
        g_cubic[0][pi][2] = (float)(mu + 0.5 * mu2 - 0.5 * mu3);
        g_cubic[0][pi][3] = (float)(-sixth * mu + sixth * mu3);
    }
    cubic_deltas(g_cubic[0]);
    /* GaussianTable, cubic_tables.cpp:39-72 */
    const double IndexScale = 512.0 / (double)(CUBIC_PHASE_COUNT * 2);
    for(unsigned pi = 0; pi < CUBIC_PHASE_COUNT; ++pi)
    {
        const double c0 = gauss_coeff((double)(CUBIC_PHASE_COUNT + pi) * IndexScale);
        const double c1 = gauss_coeff((double)pi * IndexScale);
        const double c2 = gauss_coeff((double)(CUBIC_PHASE_COUNT - pi) * IndexScale);
        const double c3 = gauss_coeff((double)(CUBIC_PHASE_COUNT * 2 - pi) * IndexScale);
        const double scale = 1.0 / (c0 + c1 + c2 + c3);
        g_cubic[1][pi][0] = (float)(c0 * scale);
        g_cubic[1][pi][1] = (float)(c1 * scale);
        g_cubic[1][pi][2] = (float)(c2 * scale);
        g_cubic[1][pi][3] = (float)(c3 * scale);
    }
    cubic_deltas(g_cubic[1]);
    g_cubic_ready = 1;
}

int oal_cubic_table_get(int which, float *out)
{
    cubic_build();
    memcpy(out, g_cubic[which ? 1 : 0], sizeof(g_cubic[0]));
    return 0;
}

/* ======================================================================== *
 * Resamplers
 * ======================================================================== */

/* BsincPrepare alc/alu.cpp:140-164; PrepareResampler :253-281; SelectResampler :167-238 */
void oal_prepare_resampler(int resampler, uint32_t increment, oal_interp_state *out)
{
    memset(out, 0, sizeof(*out));
    switch(resampler)
    {
    case OAL_RESAMPLER_POINT: out->kind = 0; return;
    case OAL_RESAMPLER_LINEAR: out->kind = 1; return;
    case OAL_RESAMPLER_SPLINE: out->kind = 2; out->table = 0; return;
    case OAL_RESAMPLER_GAUSSIAN: out->kind = 2; out->table = 1; return;
    default: break;
    }
    const int which = (resampler == OAL_RESAMPLER_FAST_BSINC12 || resampler == OAL_RESAMPLER_BSINC12) ? 12
        : (resampler == OAL_RESAMPLER_FAST_BSINC24 || resampler == OAL_RESAMPLER_BSINC24) ? 24 : 48;
    const int isfast = (resampler == OAL_RESAMPLER_FAST_BSINC12 || resampler == OAL_RESAMPLER_FAST_BSINC24
        || resampler == OAL_RESAMPLER_FAST_BSINC48);
    const oal_bsinc_table *table = bsinc_get(which);
    size_t si = BSINC_SCALE_COUNT - 1;
    float sf = 0.0f;
    if(increment > FRAC_ONE)
    {
        sf = (float)FRAC_ONE / (float)increment - table->scaleBase;
        sf = (float)BSINC_SCALE_COUNT * sf * table->scaleRange - 1.0f;
        if(!(sf > 0.0f)) sf = 0.0f; /* std::max(0.0f, x) */
        si = float2uint(sf);
        sf -= (float)si;
        sf = 1.0f - sqrtf(1.0f - sf * sf);
    }
    out->kind = (!isfast && increment > FRAC_ONE) ? 4 : 3;
    out->table = which;
    out->sf = sf;
    out->m = table->m[si];
    out->l = (out->m / 2u) - 1u;
    out->filter_offset = table->filterOffset[si];
}

/* SSE horizontal sum used by the bsinc/cubic SSE kernels: (l0+l3)+(l1+l2)
 * (core/mixer/mixer_sse.cpp:268-270). */
static float hsum_sse(const float r[4]) { return (r[0] + r[3]) + (r[1] + r[2]); }

static float do_cubic(const float (*filter)[8], const float *vals, unsigned frac, int order)
{
    /* do_cubic core/mixer/mixer_c.cpp:48-61; Resample_Cubic_SSE mixer_sse.cpp:199-230;
     * Resample_Cubic_SSE4 mixer_sse41.cpp:117-214 */
    const unsigned pi = frac >> CUBIC_PHASE_DIFF_BITS;
    const float pf = (float)(frac & ((1u << CUBIC_PHASE_DIFF_BITS) - 1u)) * (1.0f / (1u << CUBIC_PHASE_DIFF_BITS));
    float r[4];
    for(int k = 0; k < 4; ++k) r[k] = (filter[pi][k] + pf * filter[pi][4 + k]) * vals[k];
    if(order == 0) return r[0] + r[1] + r[2] + r[3];           /* C: left to right */
    if(order == 1) return (r[0] + r[1]) + (r[2] + r[3]);       /* SSE4/SSE2 main loop (transpose) */
    return hsum_sse(r);                                        /* SSE, and SSE4 tail */
}

void oal_resample(int resampler, uint32_t increment, const float *src, size_t srclen,
    uint32_t frac, float *dst, size_t n)
{
    (void)srclen;
    oal_interp_state st;
    oal_prepare_resampler(resampler, increment, &st);
    const unsigned csr = fpu_enter();
    size_t pos;
    switch(st.kind)
    {
    case 0: /* Resample_Point_C mixer_c.cpp:190-192 */
        pos = MAX_EDGE;
        for(size_t i = 0; i < n; ++i)
        {
            dst[i] = src[pos];
            frac += increment; pos += frac >> FRAC_BITS; frac &= FRAC_MASK;
        }
        break;
    case 1: /* Resample_Linear_C mixer_c.cpp:194-196 (SSE2/SSE4 variants are value-identical) */
        pos = MAX_EDGE;
        for(size_t i = 0; i < n; ++i)
        {
            dst[i] = lerpf(src[pos], src[pos + 1], (float)frac * (1.0f / FRAC_ONE));
            frac += increment; pos += frac >> FRAC_BITS; frac &= FRAC_MASK;
        }
        break;
    case 2: /* Resample_Cubic_* */
        cubic_build();
        pos = MAX_EDGE - 1;
        for(size_t i = 0; i < n; ++i)
        {
            const int order = !g_simd ? 0 : (i < (n & ~(size_t)3)) ? 1 : 2;
            dst[i] = do_cubic(g_cubic[st.table], src + pos, frac, order);
            frac += increment; pos += frac >> FRAC_BITS; frac &= FRAC_MASK;
        }
        break;
    case 3: /* Resample_FastBSinc_C mixer_c.cpp:63-82,205-212; _SSE mixer_sse.cpp:232-277 */
    case 4: /* Resample_BSinc_C mixer_c.cpp:84-105,214-221; _SSE mixer_sse.cpp:279-329 */
        {
            const oal_bsinc_table *table = bsinc_get(st.table);
            const float *filter = table->tab + st.filter_offset;
            const size_t m = st.m;
            const float sf = st.sf;
            pos = MAX_EDGE - st.l;
            for(size_t i = 0; i < n; ++i)
            {
                const unsigned pi = frac >> BSINC_PHASE_DIFF_BITS;
                const float pf = (float)(frac & ((1u << BSINC_PHASE_DIFF_BITS) - 1u))
                    * (1.0f / (1u << BSINC_PHASE_DIFF_BITS));
                const float *fil = filter + 2 * (size_t)pi * m;
                const float *phd = fil + m;
                const float *scd = fil + BSINC_PHASE_COUNT * 2 * m;
                const float *spd = scd + m;
                float r4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                float r = 0.0f;
                for(size_t j = 0; j < m; ++j)
                {
                    float f;
                    if(st.kind == 3) f = fil[j] + pf * phd[j];
                    else if(g_simd) f = (fil[j] + sf * scd[j]) + pf * (phd[j] + sf * spd[j]);
                    else f = fil[j] + sf * scd[j] + pf * (phd[j] + sf * spd[j]);
                    if(g_simd) r4[j & 3] = r4[j & 3] + f * src[pos + j];
                    else r += f * src[pos + j];
                }
                dst[i] = g_simd ? hsum_sse(r4) : r;
                frac += increment; pos += frac >> FRAC_BITS; frac &= FRAC_MASK;
            }
        }
        break;
    }
    fpu_leave(csr);
}

/* ======================================================================== *
 * Mix  (MixLine core/mixer/mixer_c.cpp:150-186; Mix_C :247-268; the SSE variant
 * core/mixer/mixer_sse.cpp:94-195,355-385 produces identical values: its vector
 * step counter holds the same exact integers and vmadd is mul-then-add)
 * ======================================================================== */
static void mix_line(const float *in, size_t n, float *dst, float *CurrentGain, float TargetGain,
    float delta, size_t fade_len, size_t Counter)
{
    const float step = (TargetGain - *CurrentGain) * delta;
    size_t pos = 0;
    if(fabsf(step) > 1.1920928955078125e-07f /* FLT_EPSILON */)
    {
        const float gain = *CurrentGain;
        float step_count = 0.0f;
        for(; pos < fade_len; ++pos)
        {
            dst[pos] += in[pos] * (gain + step * step_count);
            step_count += 1.0f;
        }
        if(fade_len < Counter)
        {
            *CurrentGain = gain + step * step_count;
            return;
        }
    }
    *CurrentGain = TargetGain;
    if(!(fabsf(TargetGain) > GAIN_SILENCE_THRESHOLD)) return;
    for(; pos < n; ++pos) dst[pos] = dst[pos] + in[pos] * TargetGain;
}

static void mix_lines(const float *in, size_t n, float *out, size_t nlines, float *cur,
    const float *tgt, size_t counter, size_t outpos)
{
    const float delta = (counter > 0) ? 1.0f / (float)counter : 0.0f;
    const size_t fade_len = counter < n ? counter : n;
    for(size_t c = 0; c < nlines; ++c)
        mix_line(in, n, out + c * LINE + outpos, &cur[c], tgt[c], delta, fade_len, counter);
}

void oal_mix(const float *in, size_t n, float *out, size_t nlines, float *cur_gains,
    const float *target_gains, size_t counter, size_t outpos)
{
    const unsigned csr = fpu_enter();
    mix_lines(in, n, out, nlines, cur_gains, target_gains, counter, outpos);
    fpu_leave(csr);
}

void oal_mix_one(const float *in, size_t n, float *out, float *cur_gain, float target_gain, size_t counter)
{
    const unsigned csr = fpu_enter();
    const float delta = (counter > 0) ? 1.0f / (float)counter : 0.0f;
    mix_line(in, n, out, cur_gain, target_gain, delta, counter < n ? counter : n, counter);
    fpu_leave(csr);
}

/* ======================================================================== *
 * HRTF mixers  (core/mixer/hrtfbase.h:17-133)
 * ======================================================================== */
/* ApplyCoeffs mixer_c.cpp:139-148; the SSE variant (mixer_sse.cpp:40-92) rounds an odd
 * IrSize up to even and is otherwise value-identical (mul then add per element). */
static void apply_coeffs(float *values /* f32x2[] */, size_t irsize, const float *coeffs, float left, float right)
{
    if(g_simd) irsize = (irsize + 1) & ~(size_t)1;
    for(size_t j = 0; j < irsize; ++j)
    {
        values[j * 2 + 0] = values[j * 2 + 0] + coeffs[j * 2 + 0] * left;
        values[j * 2 + 1] = values[j * 2 + 1] + coeffs[j * 2 + 1] * right;
    }
}

static void mix_hrtf(const float *in, float *accum, size_t irsize, const float *coeffs,
    const uint32_t delay[2], float gain, float gainstep, size_t n)
{   /* MixHrtfBase hrtfbase.h:17-42 */
    size_t ldelay = HIST - delay[0], rdelay = HIST - delay[1];
    float stepcount = 0.0f;
    for(size_t i = 0; i < n; ++i)
    {
        const float g = gain + gainstep * stepcount;
        const float left = in[ldelay++] * g;
        const float right = in[rdelay++] * g;
        apply_coeffs(accum + i * 2, irsize, coeffs, left, right);
        stepcount += 1.0f;
    }
}

static void mix_hrtf_blend(const float *in, float *accum, size_t irsize, const float *oldcoeffs,
    const uint32_t olddelay[2], float oldgain, const float *newcoeffs, const uint32_t newdelay[2],
    float newgainstep, size_t n)
{   /* MixHrtfBlendBase hrtfbase.h:44-89 */
    const float oldGainStep = oldgain / (float)n;
    if(oldgain > GAIN_SILENCE_THRESHOLD)
    {
        size_t ldelay = HIST - olddelay[0], rdelay = HIST - olddelay[1];
        float stepcount = (float)n;
        for(size_t i = 0; i < n; ++i)
        {
            const float g = oldGainStep * stepcount;
            const float left = in[ldelay++] * g;
            const float right = in[rdelay++] * g;
            apply_coeffs(accum + i * 2, irsize, oldcoeffs, left, right);
            stepcount -= 1.0f;
        }
    }
    if(newgainstep * (float)n > GAIN_SILENCE_THRESHOLD)
    {
        size_t ldelay = HIST + 1 - newdelay[0], rdelay = HIST + 1 - newdelay[1];
        float stepcount = 1.0f;
        for(size_t i = 1; i < n; ++i)
        {
            const float g = newgainstep * stepcount;
            const float left = in[ldelay++] * g;
            const float right = in[rdelay++] * g;
            apply_coeffs(accum + i * 2, irsize, newcoeffs, left, right);
            stepcount += 1.0f;
        }
    }
}

void oal_mix_hrtf(const float *in, float *accum, uint32_t irsize, const float *coeffs,
    const uint32_t delay[2], float gain, float gainstep, size_t n)
{
    const unsigned csr = fpu_enter();
    mix_hrtf(in, accum, irsize, coeffs, delay, gain, gainstep, n);
    fpu_leave(csr);
}

void oal_mix_hrtf_blend(const float *in, float *accum, uint32_t irsize, const float *oldcoeffs,
    const uint32_t olddelay[2], float oldgain, const float *newcoeffs, const uint32_t newdelay[2],
    float newgainstep, size_t n)
{
    const unsigned csr = fpu_enter();
    mix_hrtf_blend(in, accum, irsize, oldcoeffs, olddelay, oldgain, newcoeffs, newdelay, newgainstep, n);
    fpu_leave(csr);
}

/* ---- BandSplitter core/filters/splitter.cpp ---- */
void oal_splitter_init(oal_splitter *s, float f0norm) /* splitter.cpp:14-26 */
{
    const float w = 3.14159265358979323846f * 2.0f * (f0norm < 0.49f ? f0norm : 0.49f);
    const float cw = cosf(w);
    if(cw > 1.1920928955078125e-07f) s->coeff = (sinf(w) - 1.0f) / cw;
    else s->coeff = cw * -0.5f;
    s->lp_z1 = s->lp_z2 = s->ap_z1 = 0.0f;
}

static void splitter_hfscale(oal_splitter *s, const float *in, float *out, size_t n, float hfscale)
{   /* processHfScale(input, output, hfscale) splitter.cpp:65-97 -- note lp_z1 update at :79 */
    const float ap_coeff = s->coeff;
    const float lp_coeff = s->coeff * 0.5f + 0.5f;
    float lp_z1 = s->lp_z1, lp_z2 = s->lp_z2, ap_z1 = s->ap_z1;
    for(size_t i = 0; i < n; ++i)
    {
        const float x = in[i];
        const float d0 = (x - lp_z1) * lp_coeff;
        const float lp_y0 = lp_z1 + d0;
        lp_z1 = lp_y0 + d0 * lp_coeff;
        const float d1 = (lp_y0 - lp_z2) * lp_coeff;
        const float lp_y1 = lp_z2 + d1;
        lp_z2 = lp_y1 + d1;
        const float ap_y = x * ap_coeff + ap_z1;
        ap_z1 = x - ap_y * ap_coeff;
        out[i] = (ap_y - lp_y1) * hfscale + lp_y1;
    }
    s->lp_z1 = lp_z1; s->lp_z2 = lp_z2; s->ap_z1 = ap_z1;
}

void oal_splitter_process_hfscale(oal_splitter *s, const float *in, float *out, size_t n, float hfscale)
{
    const unsigned csr = fpu_enter();
    splitter_hfscale(s, in, out, n, hfscale);
    fpu_leave(csr);
}

static void splitter_scale(oal_splitter *s, float *samples, size_t n, float hfscale, float lfscale)
{   /* processScale splitter.cpp:133-161 */
    const float ap_coeff = s->coeff;
    const float lp_coeff = s->coeff * 0.5f + 0.5f;
    float lp_z1 = s->lp_z1, lp_z2 = s->lp_z2, ap_z1 = s->ap_z1;
    for(size_t i = 0; i < n; ++i)
    {
        const float x = samples[i];
        const float d0 = (x - lp_z1) * lp_coeff;
        const float lp_y0 = lp_z1 + d0;
        lp_z1 = lp_y0 + d0;
        const float d1 = (lp_y0 - lp_z2) * lp_coeff;
        const float lp_y1 = lp_z2 + d1;
        lp_z2 = lp_y1 + d1;
        const float ap_y = x * ap_coeff + ap_z1;
        ap_z1 = x - ap_y * ap_coeff;
        samples[i] = (ap_y - lp_y1) * hfscale + lp_y1 * lfscale;
    }
    s->lp_z1 = lp_z1; s->lp_z2 = lp_z2; s->ap_z1 = ap_z1;
}

void oal_splitter_process_scale(oal_splitter *s, float *samples, size_t n, float hfscale, float lfscale)
{
    const unsigned csr = fpu_enter();
    splitter_scale(s, samples, n, hfscale, lfscale);
    fpu_leave(csr);
}

static void mix_direct_hrtf(float *left, float *right, const float *in, size_t nch, float *accum,
    oal_splitter *splitters, const float *hfscales, const float *chan_coeffs, size_t irsize, size_t n)
{   /* MixDirectHrtfBase hrtfbase.h:91-133 */
    float temp[LINE];
    for(size_t c = 0; c < nch; ++c)
    {
        splitter_hfscale(&splitters[c], in + c * LINE, temp, n, hfscales[c]);
        const float *coeffs = chan_coeffs + c * HRIR_LEN * 2;
        for(size_t i = 0; i < n; ++i)
            apply_coeffs(accum + i * 2, irsize, coeffs, temp[i], temp[i]);
    }
    for(size_t i = 0; i < n; ++i)
    {
        left[i] = left[i] + accum[i * 2 + 0];
        right[i] = right[i] + accum[i * 2 + 1];
    }
    /* :130-132: move the HrirLength in-progress frames to the front, clear the rest */
    memmove(accum, accum + n * 2, sizeof(float) * 2 * HRIR_LEN);
    memset(accum + HRIR_LEN * 2, 0, sizeof(float) * 2 * n);
}

void oal_mix_direct_hrtf(float *left, float *right, const float *in, size_t nch, float *accum,
    oal_splitter *splitters, const float *hfscales, const float *chan_coeffs, size_t irsize, size_t n)
{
    const unsigned csr = fpu_enter();
    mix_direct_hrtf(left, right, in, nch, accum, splitters, hfscales, chan_coeffs, irsize, n);
    fpu_leave(csr);
}

/* ======================================================================== *
 * BiquadInterpFilter  (core/filters/biquad.{h,cpp})
 * ======================================================================== */
#define INTERP_STEPS 8       /* biquad.cpp:25 */
#define SAMPLES_PER_STEP 32  /* biquad.cpp:30 */

void oal_biquad_reset(oal_biquad *f) /* default state, biquad.h:44-50,138-139 */
{
    memset(f, 0, sizeof(*f));
    f->b0 = 1.0f; f->tb0 = 1.0f;
    f->counter = -1;
}

void oal_biquad_clear(oal_biquad *f) /* BiquadInterpFilter::clear biquad.h:152-157 */
{
    f->z1 = f->z2 = 0.0f;
    f->b0 = f->tb0; f->b1 = f->tb1; f->b2 = f->tb2; f->a1 = f->ta1; f->a2 = f->ta2;
    f->counter = 0;
}

static int check_set(float *dst, float value) /* biquad.cpp:38-43 */
{
    const int is_diff = !(fabsf(value - *dst) <= 0.015625f);
    *dst = value;
    return is_diff;
}

void oal_biquad_set_params_from_slope(oal_biquad *f, int type, float f0norm, float gain, float slope)
{
    /* setParamsFromSlope biquad.h:172-177 -> rcpQFromSlope :64-65 -> setParams biquad.cpp:131-149
     * -> SetParams :48-129 */
    if(!(gain > 0.001f)) gain = 0.001f; /* std::max(gain, 0.001f) */
    const float rcpQ = sqrtf((gain + 1.0f / gain) * (1.0f / slope - 1.0f) + 2.0f);
    if(!(gain > 0.00001f)) gain = 0.00001f;
    const float w0 = 3.14159265358979323846f * 2.0f * (f0norm < 0.49f ? f0norm : 0.49f);
    const float sin_w0 = sinf(w0);
    const float cos_w0 = cosf(w0);
    const float alpha = sin_w0 / 2.0f * rcpQ;
    float a[3] = {1.0f, 0.0f, 0.0f}, b[3] = {1.0f, 0.0f, 0.0f};
    float sqrtgain_alpha_2;
    switch(type)
    {
    case OAL_BIQUAD_HIGHSHELF:
        sqrtgain_alpha_2 = 2.0f * sqrtf(gain) * alpha;
        b[0] = gain * ((gain + 1.0f) + (gain - 1.0f) * cos_w0 + sqrtgain_alpha_2);
        b[1] = -2.0f * gain * ((gain - 1.0f) + (gain + 1.0f) * cos_w0);
        b[2] = gain * ((gain + 1.0f) + (gain - 1.0f) * cos_w0 - sqrtgain_alpha_2);
        a[0] = (gain + 1.0f) - (gain - 1.0f) * cos_w0 + sqrtgain_alpha_2;
        a[1] = 2.0f * ((gain - 1.0f) - (gain + 1.0f) * cos_w0);
        a[2] = (gain + 1.0f) - (gain - 1.0f) * cos_w0 - sqrtgain_alpha_2;
        break;
    case OAL_BIQUAD_LOWSHELF:
        sqrtgain_alpha_2 = 2.0f * sqrtf(gain) * alpha;
        b[0] = gain * ((gain + 1.0f) - (gain - 1.0f) * cos_w0 + sqrtgain_alpha_2);
        b[1] = 2.0f * gain * ((gain - 1.0f) - (gain + 1.0f) * cos_w0);
        b[2] = gain * ((gain + 1.0f) - (gain - 1.0f) * cos_w0 - sqrtgain_alpha_2);
        a[0] = (gain + 1.0f) + (gain - 1.0f) * cos_w0 + sqrtgain_alpha_2;
        a[1] = -2.0f * ((gain - 1.0f) + (gain + 1.0f) * cos_w0);
        a[2] = (gain + 1.0f) + (gain - 1.0f) * cos_w0 - sqrtgain_alpha_2;
        break;
    case OAL_BIQUAD_PEAKING:
        b[0] = 1.0f + alpha * gain; b[1] = -2.0f * cos_w0; b[2] = 1.0f - alpha * gain;
        a[0] = 1.0f + alpha / gain; a[1] = -2.0f * cos_w0; a[2] = 1.0f - alpha / gain;
        break;
    case OAL_BIQUAD_LOWPASS:
        b[0] = (1.0f - cos_w0) / 2.0f; b[1] = 1.0f - cos_w0; b[2] = (1.0f - cos_w0) / 2.0f;
        a[0] = 1.0f + alpha; a[1] = -2.0f * cos_w0; a[2] = 1.0f - alpha;
        break;
    case OAL_BIQUAD_HIGHPASS:
        b[0] = (1.0f + cos_w0) / 2.0f; b[1] = -(1.0f + cos_w0); b[2] = (1.0f + cos_w0) / 2.0f;
        a[0] = 1.0f + alpha; a[1] = -2.0f * cos_w0; a[2] = 1.0f - alpha;
        break;
    case OAL_BIQUAD_BANDPASS:
        b[0] = alpha; b[1] = 0.0f; b[2] = -alpha;
        a[0] = 1.0f + alpha; a[1] = -2.0f * cos_w0; a[2] = 1.0f - alpha;
        break;
    }
    int is_diff = check_set(&f->tb0, b[0] / a[0]);
    is_diff |= check_set(&f->tb1, b[1] / a[0]);
    is_diff |= check_set(&f->tb2, b[2] / a[0]);
    is_diff |= check_set(&f->ta1, a[1] / a[0]);
    is_diff |= check_set(&f->ta2, a[2] / a[0]);
    /* BiquadInterpFilter::setParams biquad.cpp:131-149 */
    if(!is_diff)
    {
        if(f->counter <= 0)
        {
            f->counter = 0;
            f->b0 = f->tb0; f->b1 = f->tb1; f->b2 = f->tb2; f->a1 = f->ta1; f->a2 = f->ta2;
        }
    }
    else if(f->counter >= 0)
        f->counter = INTERP_STEPS * SAMPLES_PER_STEP;
    else
    {
        f->counter = 0;
        f->b0 = f->tb0; f->b1 = f->tb1; f->b2 = f->tb2; f->a1 = f->ta1; f->a2 = f->ta2;
    }
}

static void biquad_dual_raw(oal_biquad *f0, oal_biquad *f1, const float *src, float *dst, size_t n)
{   /* BiquadFilter::dualProcess biquad.cpp:254-282 */
    float z01 = f0->z1, z02 = f0->z2, z11 = f1->z1, z12 = f1->z2;
    for(size_t i = 0; i < n; ++i)
    {
        const float x0 = src[i];
        const float y0 = x0 * f0->b0 + z01;
        z01 = x0 * f0->b1 - y0 * f0->a1 + z02;
        z02 = x0 * f0->b2 - y0 * f0->a2;
        const float x1 = y0;
        const float y1 = x1 * f1->b0 + z11;
        z11 = x1 * f1->b1 - y1 * f1->a1 + z12;
        z12 = x1 * f1->b2 - y1 * f1->a2;
        dst[i] = y1;
    }
    f0->z1 = z01; f0->z2 = z02; f1->z1 = z11; f1->z2 = z12;
}

static void biquad_lerp_coeffs(oal_biquad *f, float a)
{
    f->b0 = lerpf(f->b0, f->tb0, a); f->b1 = lerpf(f->b1, f->tb1, a); f->b2 = lerpf(f->b2, f->tb2, a);
    f->a1 = lerpf(f->a1, f->ta1, a); f->a2 = lerpf(f->a2, f->ta2, a);
}
static void biquad_snap(oal_biquad *f)
{ f->counter = 0; f->b0 = f->tb0; f->b1 = f->tb1; f->b2 = f->tb2; f->a1 = f->ta1; f->a2 = f->ta2; }

static void biquad_dual_interp(oal_biquad *f0, oal_biquad *f1, const float *src, float *dst, size_t n)
{   /* BiquadInterpFilter::dualProcess biquad.cpp:284-343 */
    const int maxcounter = f0->counter > f1->counter ? f0->counter : f1->counter;
    if(maxcounter > 0)
    {
        int counter = maxcounter / SAMPLES_PER_STEP;
        size_t steprem = (size_t)(SAMPLES_PER_STEP - (maxcounter & (SAMPLES_PER_STEP - 1)));
        while(counter > 0)
        {
            const size_t td = steprem < n ? steprem : n;
            biquad_dual_raw(f0, f1, src, dst, td);
            steprem -= td;
            if(steprem)
            {
                steprem = SAMPLES_PER_STEP - steprem;
                f0->counter = (counter * SAMPLES_PER_STEP) | (int)steprem;
                f1->counter = f0->counter;
                return;
            }
            src += td; dst += td; n -= td;
            steprem = SAMPLES_PER_STEP;
            --counter;
            if(!counter)
            {
                biquad_snap(f0); biquad_snap(f1);
                break;
            }
            const float a = 1.0f / (float)(counter + 1);
            biquad_lerp_coeffs(f0, a);
            biquad_lerp_coeffs(f1, a);
            if(n == 0)
            {
                f0->counter = counter * SAMPLES_PER_STEP;
                f1->counter = f0->counter;
                return;
            }
        }
    }
    biquad_dual_raw(f0, f1, src, dst, n);
}

void oal_biquad_dual_process(oal_biquad *f0, oal_biquad *f1, const float *src, float *dst, size_t n)
{
    const unsigned csr = fpu_enter();
    biquad_dual_interp(f0, f1, src, dst, n);
    fpu_leave(csr);
}

/* ======================================================================== *
 * HRTF data set  (core/hrtf_loader.cpp:583-721 v3 reader, :135-155 mirror;
 * core/hrtf.cpp:167-260 getCoeffs)
 * ======================================================================== */
typedef struct hrtf_store {
    uint32_t sample_rate, ir_size, num_fields, num_elevs, num_irs;
    float *field_distance; uint8_t *field_evcount;
    uint16_t *elev_azcount, *elev_iroffset;
    float *coeffs;   /* num_irs x 128 x 2 */
    uint8_t *delays; /* num_irs x 2 */
} hrtf_store;
static hrtf_store g_hrtf;

static void hrtf_free(void)
{
    free(g_hrtf.field_distance); free(g_hrtf.field_evcount); free(g_hrtf.elev_azcount);
    free(g_hrtf.elev_iroffset); free(g_hrtf.coeffs); free(g_hrtf.delays);
    memset(&g_hrtf, 0, sizeof(g_hrtf));
}

int oal_hrtf_load(const char *path)
{
    FILE *fp = fopen(path, "rb");
    if(!fp) return -1;
    fseek(fp, 0, SEEK_END);
    long len = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    uint8_t *d = (uint8_t *)malloc((size_t)len);
    if(fread(d, 1, (size_t)len, fp) != (size_t)len) { fclose(fp); free(d); return -2; }
    fclose(fp);
    size_t p = 0;
#define NEED(k) do { if(p + (k) > (size_t)len) { free(d); hrtf_free(); return -3; } } while(0)
    NEED(8);
    if(memcmp(d, "MinPHR03", 8) != 0) { free(d); return -4; } /* only format v3 is restated */
    p = 8;
    hrtf_free();
    NEED(7);
    const uint32_t rate = d[p] | (d[p + 1] << 8) | (d[p + 2] << 16) | ((uint32_t)d[p + 3] << 24); p += 4;
    const unsigned channelType = d[p++];
    const unsigned irSize = d[p++];
    const unsigned fdCount = d[p++];
    if(channelType > 1 || irSize < 8 || irSize > HRIR_LEN || fdCount < 1 || fdCount > 16) { free(d); return -5; }
    g_hrtf.sample_rate = rate; g_hrtf.ir_size = irSize; g_hrtf.num_fields = fdCount;
    g_hrtf.field_distance = (float *)calloc(fdCount, sizeof(float));
    g_hrtf.field_evcount = (uint8_t *)calloc(fdCount, 1);
    g_hrtf.elev_azcount = (uint16_t *)calloc(16 * 181, sizeof(uint16_t));
    g_hrtf.elev_iroffset = (uint16_t *)calloc(16 * 181, sizeof(uint16_t));
    size_t nelev = 0;
    for(unsigned f = 0; f < fdCount; ++f)
    {
        NEED(3);
        const unsigned distance = d[p] | (d[p + 1] << 8); p += 2;
        const unsigned evCount = d[p++];
        if(distance < 50 || distance > 2500 || evCount < 5 || evCount > 181) { free(d); hrtf_free(); return -6; }
        g_hrtf.field_distance[f] = (float)distance / 1000.0f;
        g_hrtf.field_evcount[f] = (uint8_t)evCount;
        NEED(evCount);
        for(unsigned e = 0; e < evCount; ++e) g_hrtf.elev_azcount[nelev + e] = d[p++];
        nelev += evCount;
    }
    g_hrtf.num_elevs = (uint32_t)nelev;
    g_hrtf.elev_iroffset[0] = 0;
    for(size_t e = 1; e < nelev; ++e)
        g_hrtf.elev_iroffset[e] = (uint16_t)(g_hrtf.elev_azcount[e - 1] + g_hrtf.elev_iroffset[e - 1]);
    const size_t irTotal = (size_t)g_hrtf.elev_azcount[nelev - 1] + g_hrtf.elev_iroffset[nelev - 1];
    g_hrtf.num_irs = (uint32_t)irTotal;
    g_hrtf.coeffs = (float *)calloc(irTotal * HRIR_LEN * 2, sizeof(float));
    g_hrtf.delays = (uint8_t *)calloc(irTotal * 2, 1);
    const unsigned nch = channelType == 0 ? 1 : 2;
    NEED(irTotal * irSize * nch * 3 + irTotal * nch);
    for(size_t ir = 0; ir < irTotal; ++ir)
        for(unsigned i = 0; i < irSize; ++i)
            for(unsigned c = 0; c < nch; ++c)
            {
                int32_t v = d[p] | (d[p + 1] << 8) | (d[p + 2] << 16); p += 3;
                v = (v ^ 0x800000) - 0x800000; /* fixsign<24> hrtf_loader.cpp:156-166 */
                g_hrtf.coeffs[(ir * HRIR_LEN + i) * 2 + c] = (float)v / 8388608.0f;
            }
    for(size_t ir = 0; ir < irTotal; ++ir)
        for(unsigned c = 0; c < nch; ++c)
        {
            g_hrtf.delays[ir * 2 + c] = d[p++];
            if(g_hrtf.delays[ir * 2 + c] > (63u << 2)) { free(d); hrtf_free(); return -7; }
        }
    if(channelType == 0)
    {   /* MirrorLeftHrirs hrtf_loader.cpp:135-155 */
        for(size_t e = 0; e < nelev; ++e)
        {
            const size_t evoffset = g_hrtf.elev_iroffset[e], azcount = g_hrtf.elev_azcount[e];
            for(size_t j = 0; j < azcount; ++j)
            {
                const size_t lidx = evoffset + j, ridx = evoffset + ((azcount - j) % azcount);
                for(unsigned i = 0; i < HRIR_LEN; ++i)
                    g_hrtf.coeffs[(ridx * HRIR_LEN + i) * 2 + 1] = g_hrtf.coeffs[(lidx * HRIR_LEN + i) * 2 + 0];
                g_hrtf.delays[ridx * 2 + 1] = g_hrtf.delays[lidx * 2 + 0];
            }
        }
    }
#undef NEED
    free(d);
    return 0;
}

int oal_hrtf_info_get(oal_hrtf_info *out)
{
    if(!g_hrtf.coeffs) return -1;
    out->sample_rate = g_hrtf.sample_rate; out->ir_size = g_hrtf.ir_size;
    out->num_fields = g_hrtf.num_fields; out->num_elevs = g_hrtf.num_elevs; out->num_irs = g_hrtf.num_irs;
    return 0;
}

int oal_hrtf_raw(float *field_distance, uint8_t *field_evcount, uint16_t *elev_azcount,
    uint16_t *elev_iroffset, float *coeffs, uint8_t *delays)
{
    if(!g_hrtf.coeffs) return -1;
    memcpy(field_distance, g_hrtf.field_distance, g_hrtf.num_fields * sizeof(float));
    memcpy(field_evcount, g_hrtf.field_evcount, g_hrtf.num_fields);
    memcpy(elev_azcount, g_hrtf.elev_azcount, g_hrtf.num_elevs * 2);
    memcpy(elev_iroffset, g_hrtf.elev_iroffset, g_hrtf.num_elevs * 2);
    memcpy(coeffs, g_hrtf.coeffs, (size_t)g_hrtf.num_irs * HRIR_LEN * 2 * sizeof(float));
    memcpy(delays, g_hrtf.delays, (size_t)g_hrtf.num_irs * 2);
    return 0;
}

typedef struct idx_blend { unsigned idx; float blend; } idx_blend;
static idx_blend calc_ev_index(unsigned evcount, float ev) /* hrtf.cpp:167-173 */
{
    ev = (0.318309886183790671538f * ev + 0.5f) * (float)(evcount - 1);
    const unsigned idx = float2uint(ev);
    idx_blend r = {idx < evcount - 1u ? idx : evcount - 1u, ev - (float)idx};
    return r;
}
static idx_blend calc_az_index(unsigned azcount, float az) /* hrtf.cpp:178-184 */
{
    az = (0.318309886183790671538f * 0.5f * az + 1.0f) * (float)azcount;
    const unsigned idx = float2uint(az);
    idx_blend r = {idx % azcount, az - (float)idx};
    return r;
}

static void hrtf_get_coeffs(float elevation, float azimuth, float distance, float spread,
    float *coeffs, uint32_t delays[2])
{   /* HrtfStore::getCoeffs hrtf.cpp:192-260 */
    const float dirfact = 1.0f - (0.318309886183790671538f / 2.0f * spread);
    size_t ebase = 0, field = 0;
    for(; field + 1 < g_hrtf.num_fields; ++field)
    {
        if(distance >= g_hrtf.field_distance[field]) break;
        ebase += g_hrtf.field_evcount[field];
    }
    const unsigned evcount = g_hrtf.field_evcount[field];
    const idx_blend elev0 = calc_ev_index(evcount, elevation);
    const size_t elev1_idx = (elev0.idx + 1u < evcount - 1u) ? elev0.idx + 1u : evcount - 1u;
    const size_t ir0offset = g_hrtf.elev_iroffset[ebase + elev0.idx];
    const size_t ir1offset = g_hrtf.elev_iroffset[ebase + elev1_idx];
    const unsigned azc0 = g_hrtf.elev_azcount[ebase + elev0.idx];
    const unsigned azc1 = g_hrtf.elev_azcount[ebase + elev1_idx];
    const idx_blend az0 = calc_az_index(azc0, azimuth);
    const idx_blend az1 = calc_az_index(azc1, azimuth);
    const size_t idx[4] = {ir0offset + az0.idx, ir0offset + ((az0.idx + 1) % azc0),
        ir1offset + az1.idx, ir1offset + ((az1.idx + 1) % azc1)};
    const float blend[4] = {(1.0f - elev0.blend) * (1.0f - az0.blend) * dirfact,
        (1.0f - elev0.blend) * (az0.blend) * dirfact,
        (elev0.blend) * (1.0f - az1.blend) * dirfact,
        (elev0.blend) * (az1.blend) * dirfact};
    float d = (float)g_hrtf.delays[idx[0] * 2] * blend[0] + (float)g_hrtf.delays[idx[1] * 2] * blend[1]
        + (float)g_hrtf.delays[idx[2] * 2] * blend[2] + (float)g_hrtf.delays[idx[3] * 2] * blend[3];
    delays[0] = fastf2u(d * (1.0f / 4.0f));
    d = (float)g_hrtf.delays[idx[0] * 2 + 1] * blend[0] + (float)g_hrtf.delays[idx[1] * 2 + 1] * blend[1]
        + (float)g_hrtf.delays[idx[2] * 2 + 1] * blend[2] + (float)g_hrtf.delays[idx[3] * 2 + 1] * blend[3];
    delays[1] = fastf2u(d * (1.0f / 4.0f));
    const float PassthruCoeff = (float)(1.0 / 1.41421356237309504880); /* hrtf.cpp:81 */
    coeffs[0] = PassthruCoeff * (1.0f - dirfact);
    coeffs[1] = PassthruCoeff * (1.0f - dirfact);
    for(size_t i = 2; i < HRIR_LEN * 2; ++i) coeffs[i] = 0.0f;
    for(int c = 0; c < 4; ++c)
    {
        const float *src = g_hrtf.coeffs + idx[c] * HRIR_LEN * 2;
        const float mult = blend[c];
        for(size_t i = 0; i < HRIR_LEN * 2; ++i) coeffs[i] = src[i] * mult + coeffs[i];
    }
}

void oal_hrtf_get_coeffs(float elevation, float azimuth, float distance, float spread,
    float *coeffs, uint32_t delays[2])
{
    const unsigned csr = fpu_enter();
    hrtf_get_coeffs(elevation, azimuth, distance, spread, coeffs, delays);
    fpu_leave(csr);
}

/* ======================================================================== *
 * Scene level: Voice::mix  (core/voice.cpp)
 * ======================================================================== */
typedef struct buffer_item { /* VoiceBufferItem core/voice.h:84-98 */
    void *data; int fmt; uint32_t frame_step, sample_len, loop_start, loop_end;
    int view;   /* data points into another item's storage (one channel of an interleaved buffer) */
} buffer_item;

typedef struct send_params { /* SendParams core/voice.h:73-82 */
    oal_biquad lp, hp;
    float cur[OAL_MAX_AMBI_CHANNELS], tgt[OAL_MAX_AMBI_CHANNELS];
} send_params;

/* NfcFilter (core/filters/nfc.h): sections of order 1..4, coefficients as the reference names
 * them; c[o] = {a0, a1.., b1..} with 2*o+1 entries used */
typedef struct nfc_filter {
    float base_gain[5];
    float a[5][5], b[5][5];   /* a[o][0..o], b[o][1..o] */
    float z[5][4];
} nfc_filter;

typedef struct voice { /* Voice core/voice.h:176-270 (mono, static sources) */
    int play_state;
    int position; unsigned position_frac;
    int cur_buffer, loop_buffer; /* -1 = null */
    unsigned step; int resampler;
    int is_fading, has_hrtf;
    int direct_filter_active; int send_filter_active[OAL_MAX_SENDS]; int send_slot[OAL_MAX_SENDS];
    float prev_samples[MAX_PAD];
    oal_biquad lp, hp;
    float hrtf_old_coeffs[HRIR_LEN * 2]; uint32_t hrtf_old_delay[2]; float hrtf_old_gain;
    float hrtf_tgt_coeffs[HRIR_LEN * 2]; uint32_t hrtf_tgt_delay[2]; float hrtf_tgt_gain;
    float hrtf_history[HIST];
    float gains_cur[OAL_MAX_OUTPUT_CHANNELS], gains_tgt[OAL_MAX_OUTPUT_CHANNELS];
    send_params send[OAL_MAX_SENDS];
    /* ChannelData::mAmbiSplitter / mAmbiHFScale / mAmbiLFScale + VoiceFlag::IsAmbisonic */
    int is_ambisonic; oal_splitter ambi_splitter; float ambi_hf_scale, ambi_lf_scale;
    int has_nfc; nfc_filter nfc;   /* VoiceFlag::HasNfc, DirectParams::NFCtrlFilter */
} voice;

struct oal_scene {
    oal_device_desc desc;
    uint32_t irsize;
    buffer_item *buffers; size_t nbuffers;
    voice *voices; size_t nvoices;
    float *mix;   /* (dry+real) lines */
    float *wet;   /* slots x wet_channels lines */
    float accum[(LINE + HRIR_LEN) * 2];
    /* device scratch, core/device.h:279-288 */
    float resample_data[RESAMPLE_DATA_SIZE];
    float sample_data[LINE + 256];
    float filtered[LINE];
    float extra[LINE + HIST];
    /* DirectHrtfState */
    oal_splitter *dsplit; float *dhfscale; float *dcoeffs; uint32_t dirsize;
    /* DeviceBase::mNFCtrlFilter, NumChannelsPerOrder */
    nfc_filter dev_nfc; uint32_t chans_per_order[5];
};

oal_scene *oal_scene_create(const oal_device_desc *desc)
{
    if(desc->hrtf && !g_hrtf.coeffs) return NULL;
    oal_scene *s = (oal_scene *)calloc(1, sizeof(*s));
    s->desc = *desc;
    s->irsize = desc->hrtf ? g_hrtf.ir_size : 0;
    s->mix = (float *)calloc((size_t)(desc->num_dry_channels + desc->num_real_channels) * LINE, sizeof(float));
    s->wet = (float *)calloc((size_t)(desc->num_slots * desc->wet_channels + 1) * LINE, sizeof(float));
    s->dsplit = (oal_splitter *)calloc(desc->num_dry_channels + 1, sizeof(oal_splitter));
    s->dhfscale = (float *)calloc(desc->num_dry_channels + 1, sizeof(float));
    s->dcoeffs = (float *)calloc((size_t)(desc->num_dry_channels + 1) * HRIR_LEN * 2, sizeof(float));
    s->dirsize = s->irsize;
    return s;
}

void oal_scene_destroy(oal_scene *s)
{
    if(!s) return;
    for(size_t i = 0; i < s->nbuffers; ++i) if(!s->buffers[i].view) free(s->buffers[i].data);
    free(s->buffers); free(s->voices); free(s->mix); free(s->wet);
    free(s->dsplit); free(s->dhfscale); free(s->dcoeffs);
    free(s);
}

static const size_t g_fmt_bytes[7] = {1, 2, 4, 4, 8, 1, 1};

int oal_scene_add_buffer(oal_scene *s, const void *data, int fmt_type, uint32_t frame_step,
    uint32_t sample_len, uint32_t loop_start, uint32_t loop_end)
{
    if(fmt_type < 0 || fmt_type > OAL_FMT_ALAW) return -1;
    s->buffers = (buffer_item *)realloc(s->buffers, (s->nbuffers + 1) * sizeof(buffer_item));
    buffer_item *b = &s->buffers[s->nbuffers];
    const size_t nbytes = (size_t)sample_len * frame_step * g_fmt_bytes[fmt_type];
    b->data = malloc(nbytes + 16);
    memcpy(b->data, data, nbytes);
    b->fmt = fmt_type; b->frame_step = frame_step; b->sample_len = sample_len;
    b->loop_start = loop_start; b->loop_end = loop_end;
    b->view = 0;
    return (int)s->nbuffers++;
}

/* channel `channel` of an interleaved buffer as a buffer of its own (same frames and loop points) */
static int add_buffer_view(oal_scene *s, int buffer, uint32_t channel)
{
    s->buffers = (buffer_item *)realloc(s->buffers, (s->nbuffers + 1) * sizeof(buffer_item));
    buffer_item *b = &s->buffers[s->nbuffers];
    *b = s->buffers[buffer];
    b->data = (char*)b->data + (size_t)channel * g_fmt_bytes[b->fmt];
    b->view = 1;
    return (int)s->nbuffers++;
}

int oal_scene_add_voice(oal_scene *s, const oal_voice_desc *desc)
{
    s->voices = (voice *)realloc(s->voices, (s->nvoices + 1) * sizeof(voice));
    voice *v = &s->voices[s->nvoices];
    memset(v, 0, sizeof(*v));
    /* Voice::prepare voice.cpp:1235-1397: cleared history/params, mStep = 0 */
    oal_biquad_reset(&v->lp); oal_biquad_reset(&v->hp);
    for(int i = 0; i < OAL_MAX_SENDS; ++i)
    {
        oal_biquad_reset(&v->send[i].lp); oal_biquad_reset(&v->send[i].hp);
        v->send_slot[i] = -1;
    }
    v->play_state = OAL_VOICE_PLAYING;
    v->position = desc->position; v->position_frac = desc->position_frac;
    v->cur_buffer = desc->buffer;
    v->loop_buffer = desc->looping ? desc->buffer : -1;
    v->has_hrtf = s->desc.hrtf;
    v->nfc = s->dev_nfc;   /* chandata.mDryParams.NFCtrlFilter = device->mNFCtrlFilter, voice.cpp:1391 */
    return (int)s->nvoices++;
}

static void set_filter_pair(oal_biquad *lp, oal_biquad *hp, const oal_filter_params *fp)
{   /* alc/alu.cpp:1630-1633,1648-1651 */
    oal_biquad_set_params_from_slope(lp, OAL_BIQUAD_HIGHSHELF, fp->hf_norm, fp->gain_hf, 1.0f);
    oal_biquad_set_params_from_slope(hp, OAL_BIQUAD_LOWSHELF, fp->lf_norm, fp->gain_lf, 1.0f);
}

int oal_scene_set_voice_params(oal_scene *s, int vi, const oal_voice_params *p)
{
    voice *v = &s->voices[vi];
    const unsigned csr = fpu_enter();
    v->step = p->step;
    v->resampler = p->resampler;
    for(unsigned i = 0; i < s->desc.num_aux_sends; ++i) v->send_slot[i] = p->send_slot[i];
    if(s->desc.hrtf)
    {
        hrtf_get_coeffs(p->hrtf_ev, p->hrtf_az, p->hrtf_dist, p->hrtf_spread, v->hrtf_tgt_coeffs, v->hrtf_tgt_delay);
        v->hrtf_tgt_gain = p->hrtf_gain;
        v->has_hrtf = 1;
    }
    else
        memcpy(v->gains_tgt, p->dry_gains, sizeof(v->gains_tgt));
    for(unsigned i = 0; i < s->desc.num_aux_sends; ++i)
        memcpy(v->send[i].tgt, p->send_gains[i], sizeof(v->send[i].tgt));
    v->direct_filter_active = p->direct_filter.active != 0;
    set_filter_pair(&v->lp, &v->hp, &p->direct_filter);
    for(unsigned i = 0; i < s->desc.num_aux_sends; ++i)
    {
        v->send_filter_active[i] = p->send_filter[i].active != 0;
        set_filter_pair(&v->send[i].lp, &v->send[i].hp, &p->send_filter[i]);
    }
    fpu_leave(csr);
    return 0;
}

/* ---- NfcFilter, core/filters/nfc.cpp:56-288 ---- */
static const float NFC_B1[1] = {1.0f};
static const float NFC_B2[2] = {3.0f, 3.0f};
static const float NFC_B3[3] = {3.6778f, 6.4595f, 2.3222f};
static const float NFC_B4[4] = {4.2076f, 11.4877f, 5.7924f, 9.1401f};

/* the section gains g_1 (second-order part) and g_0 (remaining part) and the derived
 * coefficients for order o and angular frequency w; out[1..o] (NfcFilterCreateN / AdjustN) */
static float nfc_design(int o, float w, float *out)
{
    const float r = 0.5f * w;
    if(o == 1)
    {
        const float b_00 = NFC_B1[0] * r;
        const float g_0 = 1.0f + b_00;
        out[1] = 2.0f * b_00 / g_0;
        return g_0;
    }
    if(o == 2)
    {
        const float b_10 = NFC_B2[0] * r, b_11 = NFC_B2[1] * (r * r);
        const float g_1 = 1.0f + b_10 + b_11;
        out[1] = (2.0f * b_10 + 4.0f * b_11) / g_1;
        out[2] = 4.0f * b_11 / g_1;
        return g_1;
    }
    if(o == 3)
    {
        const float b_10 = NFC_B3[0] * r, b_11 = NFC_B3[1] * (r * r), b_00 = NFC_B3[2] * r;
        const float g_1 = 1.0f + b_10 + b_11, g_0 = 1.0f + b_00;
        out[1] = (2.0f * b_10 + 4.0f * b_11) / g_1;
        out[2] = 4.0f * b_11 / g_1;
        out[3] = 2.0f * b_00 / g_0;
        return g_1 * g_0;
    }
    {
        const float b_10 = NFC_B4[0] * r, b_11 = NFC_B4[1] * (r * r), b_00 = NFC_B4[2] * r, b_01 = NFC_B4[3] * (r * r);
        const float g_1 = 1.0f + b_10 + b_11, g_0 = 1.0f + b_00 + b_01;
        out[1] = (2.0f * b_10 + 4.0f * b_11) / g_1;
        out[2] = 4.0f * b_11 / g_1;
        out[3] = (2.0f * b_00 + 4.0f * b_01) / g_0;
        out[4] = 4.0f * b_01 / g_0;
        return g_1 * g_0;
    }
}

static void nfc_init(nfc_filter *f, float w1) /* NfcFilter::init :205-211 */
{
    memset(f, 0, sizeof(*f));
    for(int o = 1; o <= 4; ++o)
    {
        const float g = nfc_design(o, w1, f->a[o]);
        f->base_gain[o] = 1.0f / g;
        f->a[o][0] = 1.0f;
        for(int k = 1; k <= o; ++k) f->b[o][k] = f->a[o][k];
    }
}

static void nfc_adjust(nfc_filter *f, float w0) /* NfcFilter::adjust :213-219 */
{
    for(int o = 1; o <= 4; ++o)
    {
        const float g = nfc_design(o, w0, f->b[o]);
        f->a[o][0] = f->base_gain[o] * g;
    }
}

static void nfc_process(nfc_filter *f, int o, const float *src, float *dst, size_t n) /* :222-288 */
{
    const float *a = f->a[o], *b = f->b[o];
    float *z = f->z[o];
    for(size_t i = 0; i < n; ++i)
    {
        const float in = src[i];
        if(o == 1)
        {
            const float y = in * a[0] - a[1] * z[0];
            dst[i] = y + b[1] * z[0];
            z[0] += y;
            continue;
        }
        const float y0 = in * a[0] - a[1] * z[0] - a[2] * z[1];
        const float out0 = y0 + b[1] * z[0] + b[2] * z[1];
        z[1] += z[0];
        z[0] += y0;
        if(o == 2) { dst[i] = out0; continue; }
        if(o == 3)
        {
            const float y1 = out0 - a[3] * z[2];
            dst[i] = y1 + b[3] * z[2];
            z[2] += y1;
            continue;
        }
        const float y1 = out0 - a[3] * z[2] - a[4] * z[3];
        dst[i] = y1 + b[3] * z[2] + b[4] * z[3];
        z[3] += z[2];
        z[2] += y1;
    }
}

int oal_scene_set_nfc(oal_scene *s, float w1, const uint32_t channels_per_order[5])
{
    nfc_init(&s->dev_nfc, w1);
    memcpy(s->chans_per_order, channels_per_order, sizeof(s->chans_per_order));
    return 0;
}

int oal_scene_set_voice_nfc(oal_scene *s, int vi, float w0)
{
    const unsigned csr = fpu_enter();
    nfc_adjust(&s->voices[vi].nfc, w0);
    s->voices[vi].has_nfc = 1;
    fpu_leave(csr);
    return 0;
}

int oal_scene_add_voice_multi(oal_scene *s, const oal_voice_desc *desc, uint32_t num_channels)
{   /* one mono voice per channel, in channel order: DoMix (voice.cpp:934-984) walks the channels
     * of a voice in that order too, so the buses accumulate identically */
    if(num_channels != s->buffers[desc->buffer].frame_step) return -1;
    int first = -1;
    for(uint32_t c = 0; c < num_channels; ++c)
    {
        oal_voice_desc d = *desc;
        d.buffer = add_buffer_view(s, desc->buffer, c);
        const int vi = oal_scene_add_voice(s, &d);
        if(c == 0) first = vi;
    }
    return first;
}

int oal_scene_set_channel_params(oal_scene *s, int voice, uint32_t channel, const oal_voice_params *p)
{ return oal_scene_set_voice_params(s, voice + (int)channel, p); }

int oal_scene_set_channel_ambi_scale(oal_scene *s, int voice, uint32_t channel, float xover_norm,
    float hf_scale, float lf_scale)
{
    struct voice *v = &s->voices[voice + (int)channel];
    oal_splitter_init(&v->ambi_splitter, xover_norm);
    v->ambi_hf_scale = hf_scale; v->ambi_lf_scale = lf_scale;
    v->is_ambisonic = 1;
    return 0;
}

int oal_scene_set_voice_state(oal_scene *s, int vi, int vstate)
{
    s->voices[vi].play_state = vstate;
    return 0;
}

/* SampleInfo<T>::to_float core/fmt_traits.h:91-139 */
static const int16_t muLawTable_seg[8] = {0, 132, 396, 924, 1980, 4092, 8316, 16764};
static int16_t mulaw_decode(uint8_t u)
{   /* closed form of muLawDecompressionTable, fmt_traits.h:12-45 */
    u = (uint8_t)~u;
    const int sign = u & 0x80, exponent = (u >> 4) & 7, mantissa = u & 15;
    const int mag = muLawTable_seg[exponent] + (mantissa << (exponent + 3));
    return (int16_t)(sign ? -mag : mag);
}
static int16_t alaw_decode(uint8_t a)
{   /* closed form of aLawDecompressionTable, fmt_traits.h:47-80 */
    a ^= 0x55;
    const int sign = a & 0x80, exponent = (a >> 4) & 7, mantissa = a & 15;
    int mag = (exponent == 0) ? ((mantissa << 4) + 8) : (((mantissa << 4) + 0x108) << (exponent - 1));
    return (int16_t)(sign ? mag : -mag);
}

static float load_sample(const buffer_item *b, size_t idx)
{
    switch(b->fmt)
    {
    case OAL_FMT_UBYTE: return ((float)((const uint8_t *)b->data)[idx] - 128.0f) * (1.0f / 128.0f);
    case OAL_FMT_SHORT: return (float)((const int16_t *)b->data)[idx] * (1.0f / 32768.0f);
    case OAL_FMT_INT: return (float)((const int32_t *)b->data)[idx] * (1.0f / 2147483648.0f);
    case OAL_FMT_FLOAT: return ((const float *)b->data)[idx];
    case OAL_FMT_DOUBLE: return (float)((const double *)b->data)[idx];
    case OAL_FMT_MULAW: return (float)mulaw_decode(((const uint8_t *)b->data)[idx]) * (1.0f / 32768.0f);
    case OAL_FMT_ALAW: return (float)alaw_decode(((const uint8_t *)b->data)[idx]) * (1.0f / 32768.0f);
    }
    return 0.0f;
}

/* LoadSamples<T> voice.cpp:270-286 (channel 0 of an interleaved frame) */
static void load_samples(float *dst, size_t n, const buffer_item *b, size_t srcOffset)
{
    for(size_t i = 0; i < n; ++i) dst[i] = load_sample(b, (srcOffset + i) * b->frame_step);
}

/* LoadBufferStatic voice.cpp:500-544 */
static void load_buffer_static(const buffer_item *b, int looping, size_t dataPosInt, float *dst, size_t n)
{
    if(!looping)
    {
        float lastSample = 0.0f;
        if(b->sample_len > dataPosInt)
        {
            const size_t buffer_remaining = b->sample_len - dataPosInt;
            const size_t remaining = n < buffer_remaining ? n : buffer_remaining;
            load_samples(dst, remaining, b, dataPosInt);
            lastSample = dst[remaining - 1];
            dst += remaining; n -= remaining;
        }
        for(size_t i = 0; i < n; ++i) dst[i] = lastSample;
    }
    else
    {
        const size_t loopStart = b->loop_start, loopEnd = b->loop_end;
        const size_t intPos = (dataPosInt < loopEnd) ? dataPosInt
            : (((dataPosInt - loopStart) % (loopEnd - loopStart)) + loopStart);
        size_t remaining = loopEnd - intPos; if(n < remaining) remaining = n;
        load_samples(dst, remaining, b, intPos);
        dst += remaining; n -= remaining;
        const size_t loopSize = loopEnd - loopStart;
        while(n > 0)
        {
            const size_t toFill = n < loopSize ? n : loopSize;
            load_samples(dst, toFill, b, loopStart);
            dst += toFill; n -= toFill;
        }
    }
}

/* CalculateBufferSize voice.cpp:600-640 */
static void calc_buffer_size(unsigned fracPos, unsigned increment, unsigned dstRemaining,
    unsigned *dst, unsigned *src)
{
    const unsigned SrcSizeMax = RESAMPLE_DATA_SIZE - MAX_EDGE;
    const unsigned ext = increment <= FRAC_ONE;
    const uint64_t srcSize64 = (((uint64_t)(dstRemaining - ext) * increment + fracPos) >> FRAC_BITS) + ext + MAX_EDGE;
    if(srcSize64 <= SrcSizeMax) { *dst = dstRemaining; *src = (unsigned)srcSize64; return; }
    const uint64_t dstSize64 = (((uint64_t)(SrcSizeMax - MAX_EDGE) << FRAC_BITS) - fracPos) / increment;
    if(dstSize64 < dstRemaining) { *dst = (unsigned)dstSize64 & ~3u; *src = SrcSizeMax; return; }
    *dst = dstRemaining; *src = SrcSizeMax;
}

static int add_sat(int a, int b)
{
    long long r = (long long)a + b;
    if(r > 2147483647LL) r = 2147483647LL;
    if(r < -2147483648LL) r = -2147483648LL;
    return (int)r;
}

/* LoadResampledSamples voice.cpp:642-824 (one real channel, static sources) */
static void load_resampled(oal_scene *s, voice *v, int vstate, int intBufferPos, unsigned fracBufferPos,
    unsigned increment, unsigned samplesToLoad, unsigned samplesToMix, int bufferListItem,
    int bufferLoopItem, float *mixing)
{
    float *rdata = s->resample_data;
    memcpy(rdata, v->prev_samples, sizeof(v->prev_samples));
    float *srcBuffer = rdata + MAX_EDGE;
    int intPos = intBufferPos;
    unsigned fracPos = fracBufferPos;
    for(unsigned samplesLoaded = 0; samplesLoaded < samplesToLoad;)
    {
        unsigned bdst, bsrc;
        calc_buffer_size(fracPos, increment, samplesToLoad - samplesLoaded, &bdst, &bsrc);
        unsigned srcSampleDelay = 0;
        if(intPos < 0)
        {
            srcSampleDelay = (unsigned)(-intPos);
            if(srcSampleDelay >= bsrc)
            {
                memset(mixing + samplesLoaded, 0, sizeof(float) * bdst);
                memset(srcBuffer, 0, sizeof(float) * bsrc);
                samplesLoaded += bdst;
                if(samplesLoaded < samplesToLoad)
                {
                    fracPos += bdst * increment;
                    const unsigned srcOffset = fracPos >> FRAC_BITS;
                    fracPos &= FRAC_MASK;
                    intPos = add_sat(intPos, (int)srcOffset);
                }
                continue;
            }
            memset(srcBuffer, 0, sizeof(float) * srcSampleDelay);
        }
        if(bufferListItem < 0)
        {   /* voice.cpp:704-719: hold the history sample nearest zero */
            const unsigned avail = bsrc < MAX_EDGE ? bsrc : MAX_EDGE;
            const unsigned tofill = bsrc > MAX_EDGE ? bsrc : MAX_EDGE;
            unsigned best = 0;
            for(unsigned i = 1; i < avail; ++i)
                if(fabsf(srcBuffer[i]) < fabsf(srcBuffer[best])) best = i;
            for(unsigned i = best + 1; i < tofill; ++i) srcBuffer[i] = srcBuffer[best];
        }
        else
        {
            const unsigned uintPos = intPos < 0 ? 0u : (unsigned)intPos;
            load_buffer_static(&s->buffers[bufferListItem], bufferLoopItem >= 0, uintPos,
                srcBuffer + srcSampleDelay, bsrc - srcSampleDelay);
        }
        if(increment == FRAC_ONE && fracPos == 0)
            memcpy(mixing + samplesLoaded, srcBuffer, sizeof(float) * bdst);
        else
            oal_resample(v->resampler, increment, rdata, RESAMPLE_DATA_SIZE, fracPos, mixing + samplesLoaded, bdst);
        if(vstate == OAL_VOICE_PLAYING)
        {
            const unsigned loadEnd = samplesLoaded + bdst;
            if(samplesToMix > samplesLoaded && samplesToMix <= loadEnd)
            {
                const size_t dstOffset = samplesToMix - samplesLoaded;
                const size_t srcOffset = (dstOffset * increment + fracPos) >> FRAC_BITS;
                memcpy(v->prev_samples, rdata + srcOffset, sizeof(v->prev_samples));
            }
        }
        samplesLoaded += bdst;
        if(samplesLoaded < samplesToLoad)
        {
            fracPos += bdst * increment;
            const unsigned srcOffset = fracPos >> FRAC_BITS;
            fracPos &= FRAC_MASK;
            if(intPos < 0) intPos += (int)srcOffset;
            else intPos = add_sat(intPos, (int)srcOffset);
            memmove(rdata, rdata + srcOffset, sizeof(float) * MAX_PAD);
        }
    }
}

/* DoFilters voice.cpp:255-267 */
static const float *do_filters(oal_biquad *lp, oal_biquad *hp, float *dst, const float *src, size_t n, int active)
{
    if(active) { biquad_dual_interp(lp, hp, src, dst, n); return dst; }
    oal_biquad_clear(lp); oal_biquad_clear(hp);
    return src;
}

/* DoHrtfMix voice.cpp:827-902 */
static void do_hrtf_mix(oal_scene *s, voice *v, const float *samples, size_t n, float targetGain,
    size_t counter, size_t outPos, int isPlaying)
{
    float *HrtfSamples = s->extra;
    memcpy(HrtfSamples, v->hrtf_history, sizeof(v->hrtf_history));
    memcpy(HrtfSamples + HIST, samples, sizeof(float) * n);
    if(isPlaying) memcpy(v->hrtf_history, HrtfSamples + n, sizeof(v->hrtf_history));
    size_t fademix = 0;
    if(counter && outPos == 0)
    {
        fademix = n < counter ? n : counter;
        float gain = targetGain;
        if(counter > fademix)
        {
            const float a = (float)fademix / (float)counter;
            gain = lerpf(v->hrtf_old_gain, targetGain, a);
        }
        mix_hrtf_blend(HrtfSamples, s->accum + outPos * 2, s->irsize, v->hrtf_old_coeffs, v->hrtf_old_delay,
            v->hrtf_old_gain, v->hrtf_tgt_coeffs, v->hrtf_tgt_delay, gain / (float)fademix, fademix);
        memcpy(v->hrtf_old_coeffs, v->hrtf_tgt_coeffs, sizeof(v->hrtf_old_coeffs));
        v->hrtf_old_delay[0] = v->hrtf_tgt_delay[0]; v->hrtf_old_delay[1] = v->hrtf_tgt_delay[1];
        v->hrtf_old_gain = gain;
        outPos += fademix;
    }
    if(fademix < n)
    {
        const size_t todo = n - fademix;
        float gain = targetGain;
        if(counter > n)
        {
            const float a = (float)todo / (float)(counter - fademix);
            gain = lerpf(v->hrtf_old_gain, targetGain, a);
        }
        mix_hrtf(HrtfSamples + fademix, s->accum + outPos * 2, s->irsize, v->hrtf_tgt_coeffs, v->hrtf_tgt_delay,
            v->hrtf_old_gain, (gain - v->hrtf_old_gain) / (float)todo, todo);
        v->hrtf_old_gain = gain;
    }
}

/* Voice::mix voice.cpp:988-1233 (static mono voice, no delayed start) */
static void voice_mix(oal_scene *s, voice *v, int vstate, unsigned samplesToDo)
{
    const unsigned numSends = s->desc.num_aux_sends;
    int bufPosInt = v->position;
    unsigned bufPosFrac = v->position_frac;
    int bufferListItem = v->cur_buffer;
    int bufferLoopItem = v->loop_buffer;
    const unsigned increment = v->step;
    if(increment < 1)
    {
        if(vstate == OAL_VOICE_STOPPING) v->play_state = OAL_VOICE_STOPPED;
        return;
    }
    if(bufferLoopItem >= 0) /* :1015-1019 */
    {
        if(bufPosInt >= 0 && (unsigned)bufPosInt >= s->buffers[bufferListItem].loop_end)
            bufferLoopItem = -1;
    }
    const unsigned outPos = 0;
    const unsigned samplesToMix = samplesToDo - outPos;
    const unsigned samplesToLoad = samplesToMix;
    float *mixing = s->sample_data;

    load_resampled(s, v, vstate, bufPosInt, bufPosFrac, increment, samplesToLoad, samplesToMix,
        bufferListItem, bufferLoopItem, mixing);

    if(v->is_ambisonic) /* :1082-1091 */
        splitter_scale(&v->ambi_splitter, mixing, samplesToMix, v->ambi_hf_scale, v->ambi_lf_scale);

    const unsigned counter = v->is_fading ? (samplesToMix < 64u ? samplesToMix : 64u) : 0u; /* :1093 */
    if(!counter)
    {   /* :1094-1112 */
        if(!v->has_hrtf) memcpy(v->gains_cur, v->gains_tgt, sizeof(v->gains_cur));
        else
        {
            memcpy(v->hrtf_old_coeffs, v->hrtf_tgt_coeffs, sizeof(v->hrtf_old_coeffs));
            v->hrtf_old_delay[0] = v->hrtf_tgt_delay[0]; v->hrtf_old_delay[1] = v->hrtf_tgt_delay[1];
            v->hrtf_old_gain = v->hrtf_tgt_gain;
        }
        for(unsigned i = 0; i < numSends; ++i)
            if(v->send_slot[i] >= 0) memcpy(v->send[i].cur, v->send[i].tgt, sizeof(v->send[i].cur));
    }

    /* DoMix voice.cpp:934-984 */
    {
        static const float SilentCoeffs[OAL_MAX_OUTPUT_CHANNELS] = {0};
        const float *samples = do_filters(&v->lp, &v->hp, s->filtered, mixing, samplesToMix, v->direct_filter_active);
        if(v->has_hrtf)
        {
            const float targetGain = v->hrtf_tgt_gain * (float)(vstate == OAL_VOICE_PLAYING);
            do_hrtf_mix(s, v, samples, samplesToMix, targetGain, counter, outPos, vstate == OAL_VOICE_PLAYING);
        }
        else
        {
            const float *tg = (vstate == OAL_VOICE_PLAYING) ? v->gains_tgt : SilentCoeffs;
            if(v->has_nfc)
            {   /* DoNfcMix voice.cpp:904-932 */
                mix_lines(samples, samplesToMix, s->mix, 1, v->gains_cur, tg, counter, outPos);
                size_t line = 1;
                for(int order = 1; order <= 4 && s->chans_per_order[order]; ++order)
                {
                    const size_t cnt = s->chans_per_order[order];
                    nfc_process(&v->nfc, order, samples, s->extra, samplesToMix);
                    mix_lines(s->extra, samplesToMix, s->mix + line * LINE, cnt, v->gains_cur + line, tg + line,
                        counter, outPos);
                    line += cnt;
                }
            }
            else
            mix_lines(samples, samplesToMix, s->mix, s->desc.num_dry_channels, v->gains_cur, tg, counter, outPos);
        }
        for(unsigned i = 0; i < numSends; ++i)
        {
            if(v->send_slot[i] < 0) continue;
            const float *ws = do_filters(&v->send[i].lp, &v->send[i].hp, s->filtered, mixing, samplesToMix,
                v->send_filter_active[i]);
            const float *tg = (vstate == OAL_VOICE_PLAYING) ? v->send[i].tgt : SilentCoeffs;
            mix_lines(ws, samplesToMix, s->wet + (size_t)v->send_slot[i] * s->desc.wet_channels * LINE,
                s->desc.wet_channels, v->send[i].cur, tg, counter, outPos);
        }
    }
    v->is_fading = 1;
    if(vstate == OAL_VOICE_STOPPING) { v->play_state = OAL_VOICE_STOPPED; return; }

    /* position update :1126-1154 */
    bufPosFrac += increment * samplesToMix;
    const unsigned samplesDone = bufPosFrac >> FRAC_BITS;
    bufPosInt = add_sat(bufPosInt, (int)samplesDone);
    bufPosFrac &= FRAC_MASK;
    if(bufferListItem >= 0 && bufPosInt > 0)
    {
        const buffer_item *b = &s->buffers[bufferListItem];
        if(bufferLoopItem >= 0)
        {
            unsigned DataPosUInt = (unsigned)bufPosInt;
            if(DataPosUInt >= b->loop_end)
            {
                DataPosUInt = ((DataPosUInt - b->loop_start) % (b->loop_end - b->loop_start)) + b->loop_start;
                bufPosInt = (int)DataPosUInt;
            }
        }
        else if((unsigned)bufPosInt >= b->sample_len)
            bufferListItem = -1;
    }
    v->position = bufPosInt;
    v->position_frac = bufPosFrac;
    v->cur_buffer = bufferListItem;
    if(bufferListItem < 0)
    {
        v->loop_buffer = -1;
        v->play_state = OAL_VOICE_STOPPING; /* :1224-1232 */
    }
}

int oal_scene_mix(oal_scene *s, uint32_t samples_to_do, int post_process)
{
    const unsigned csr = fpu_enter();
    const size_t nlines = s->desc.num_dry_channels + s->desc.num_real_channels;
    memset(s->mix, 0, nlines * LINE * sizeof(float));                                   /* alu.cpp:2417 */
    memset(s->wet, 0, (size_t)s->desc.num_slots * s->desc.wet_channels * LINE * sizeof(float)); /* :2196 */
    for(size_t i = 0; i < s->nvoices; ++i)
    {
        voice *v = &s->voices[i];
        const int vstate = v->play_state;
        if(vstate == OAL_VOICE_PLAYING || vstate == OAL_VOICE_STOPPING)
            voice_mix(s, v, vstate, samples_to_do);
    }
    if(post_process && s->desc.hrtf)
    {   /* DeviceBase::Process(HrtfPostProcess) alu.cpp:289-298 */
        float *left = s->mix + (size_t)s->desc.num_dry_channels * LINE;
        float *right = left + LINE;
        mix_direct_hrtf(left, right, s->mix, s->desc.num_dry_channels, s->accum, s->dsplit, s->dhfscale,
            s->dcoeffs, s->dirsize, samples_to_do);
    }
    fpu_leave(csr);
    return 0;
}

int oal_scene_post_process(oal_scene *s, uint32_t samples_to_do)
{   /* DeviceBase::Process(HrtfPostProcess) alu.cpp:289-298 */
    if(!s->desc.hrtf) return -1;
    const unsigned csr = fpu_enter();
    float *left = s->mix + (size_t)s->desc.num_dry_channels * LINE;
    float *right = left + LINE;
    mix_direct_hrtf(left, right, s->mix, s->desc.num_dry_channels, s->accum, s->dsplit, s->dhfscale,
        s->dcoeffs, s->dirsize, samples_to_do);
    fpu_leave(csr);
    return 0;
}

const float *oal_scene_dry(oal_scene *s) { return s->mix; }
const float *oal_scene_wet(oal_scene *s, int slot) { return s->wet + (size_t)slot * s->desc.wet_channels * LINE; }
const float *oal_scene_hrtf_accum(oal_scene *s) { return s->accum; }

int oal_scene_voice_state(oal_scene *s, int vi, oal_voice_state *out)
{
    const voice *v = &s->voices[vi];
    memset(out, 0, sizeof(*out));
    out->play_state = v->play_state;
    out->position = v->position;
    out->position_frac = v->position_frac;
    out->has_buffer = v->cur_buffer >= 0;
    out->fading = v->is_fading;
    memcpy(out->prev_samples, v->prev_samples, sizeof(out->prev_samples));
    memcpy(out->dry_current, v->gains_cur, sizeof(out->dry_current));
    out->hrtf_old_gain = v->hrtf_old_gain;
    out->hrtf_old_delay[0] = v->hrtf_old_delay[0]; out->hrtf_old_delay[1] = v->hrtf_old_delay[1];
    memcpy(out->hrtf_history, v->hrtf_history, sizeof(out->hrtf_history));
    out->direct_lp = v->lp; out->direct_hp = v->hp;
    for(int i = 0; i < OAL_MAX_SENDS; ++i)
    {
        memcpy(out->send_current[i], v->send[i].cur, sizeof(out->send_current[i]));
        out->send_lp[i] = v->send[i].lp; out->send_hp[i] = v->send[i].hp;
    }
    return 0;
}

int oal_scene_set_direct_hrtf(oal_scene *s, const float *chan_coeffs, const float *hfscales,
    float xover_norm, uint32_t irsize)
{
    if(!s->desc.hrtf) return -1;
    s->dirsize = irsize;
    for(unsigned c = 0; c < s->desc.num_dry_channels; ++c)
    {
        oal_splitter_init(&s->dsplit[c], xover_norm);
        s->dhfscale[c] = hfscales[c];
    }
    memcpy(s->dcoeffs, chan_coeffs, (size_t)s->desc.num_dry_channels * HRIR_LEN * 2 * sizeof(float));
    return 0;
}

/* ======================================================================== *
 * Convolution reverb  (ConvolutionState, alc/effects/convolution.cpp:253-716)
 *
 * The reference splits the impulse response into 128-tap segments: segment 0 is
 * applied in the time domain (apply_fir :205-250, over mInput :260), segments
 * 1.. in the frequency domain (256-point pffft, spectra pre-scaled and
 * z-reordered at setup :411-470, accumulated per completed 128-sample input
 * block :672-709), whose result is overlap-added one block later (:644-652,
 * :701-707).  The sum of all of it is the linear convolution of the wet input
 * with the impulse response, sample for sample; this restatement keeps segment 0
 * as the float dot product of apply_fir's scalar variant and evaluates the other
 * segments as one double-precision dot product per output sample, so it agrees
 * with the compiled reference to float rounding (pinned with a tolerance in
 * tests/test_oracle_pin.py, not bit-for-bit: the FFT order is not restated).
 * update (:474-621) for a mono response: the channel is panned straight ahead,
 * Target[i] = AmbiMap[i].Scale * coeffs[AmbiMap[i].Index] * slot gain
 * (ComputePanGains core/mixer.cpp:103-112) with an identity map; process ends
 * with MixSamples(chan buffer, out, Current, Target, Counter = samplesToDo)
 * (NormalMix :298-304).  Same-rate responses only (the polyphase resampler of
 * :356-362,419-426 is a setup step this restatement leaves to the reference).
 * ======================================================================== */
#define CONV_SEG 128
struct oal_conv {
    unsigned nlines, irlen;
    float *ir;              /* device-rate impulse response */
    float *hist;            /* the last irlen input samples, oldest first */
    float cur[OAL_MAX_AMBI_CHANNELS], tgt[OAL_MAX_AMBI_CHANNELS];
};

/* CalcAmbiCoeffs(y, z, x) up to first order (core/ambidefs.h: ACN order, N3D scaling);
 * the callers in scope pan into first-order buses. */
void oal_calc_direction_coeffs(const float dir[3], float spread, float *out25)
{
    const float y = -dir[0], z = dir[1], x = -dir[2];   /* CalcDirectionCoeffs, core/mixer.h:68-73 */
    memset(out25, 0, 25 * sizeof(float));
    out25[0] = 1.0f;
    out25[1] = 1.732050808f * y;
    out25[2] = 1.732050808f * z;
    out25[3] = 1.732050808f * x;
    if(spread > 0.0f)
    {   /* core/mixer.cpp:20-62 */
        const float ca = cosf(spread * 0.5f);
        const float scale = sqrtf(1.0f + 0.318309886183790671538f * 0.5f * spread);
        const float zh1 = scale * 0.5f * (ca + 1.0f);
        out25[0] *= scale; out25[1] *= zh1; out25[2] *= zh1; out25[3] *= zh1;
    }
}

oal_conv *oal_conv_create(uint32_t sample_rate, uint32_t num_out_lines, const float *ir,
    uint32_t ir_len, uint32_t ir_rate)
{
    if(sample_rate != ir_rate || ir_len < 1 || num_out_lines > OAL_MAX_AMBI_CHANNELS) return NULL;
    oal_conv *c = (oal_conv*)calloc(1, sizeof(*c));
    c->nlines = num_out_lines;
    c->irlen = ir_len;
    c->ir = (float*)malloc(ir_len * sizeof(float));
    memcpy(c->ir, ir, ir_len * sizeof(float));
    c->hist = (float*)calloc(ir_len, sizeof(float));
    return c;
}

void oal_conv_update(oal_conv *c, float slot_gain)
{
    static const float front[3] = {0.0f, 0.0f, -1.0f};  /* MonoMap, convolution.cpp:153 */
    float coeffs[25];
    oal_calc_direction_coeffs(front, 0.0f, coeffs);
    memset(c->tgt, 0, sizeof(c->tgt));
    for(unsigned i = 0; i < c->nlines; ++i) c->tgt[i] = 1.0f * coeffs[i] * slot_gain;
}

void oal_conv_process(oal_conv *c, const float *wet_in, float *out_lines, uint32_t n)
{
    const unsigned csr = fpu_enter();
    float buf[LINE];
    const unsigned L = c->irlen;
    /* timeline = [hist (L samples) | wet_in (n samples)] */
    float *tl = (float*)malloc((L + n) * sizeof(float));
    memcpy(tl, c->hist, L * sizeof(float));
    memcpy(tl + L, wet_in, n * sizeof(float));
    for(uint32_t i = 0; i < n; ++i)
    {
        const float *x = tl + L + i;                 /* x[0] = current sample, x[-k] = k samples ago */
        float fir = 0.0f;                            /* apply_fir, convolution.cpp:240-248 */
        const unsigned first = L < CONV_SEG ? L : CONV_SEG;
        for(unsigned j = CONV_SEG - first; j < CONV_SEG; ++j)   /* filter[j] = ir[127-j], input[j] = x[j-127] */
            fir += x[(int)j - (CONV_SEG - 1)] * c->ir[CONV_SEG - 1 - j];
        double tail = 0.0;
        for(unsigned k = CONV_SEG; k < L; ++k) tail += (double)c->ir[k] * (double)x[-(int)k];
        buf[i] = fir + (float)tail;
    }
    memcpy(c->hist, tl + n, L * sizeof(float));
    free(tl);
    mix_lines(buf, n, out_lines, c->nlines, c->cur, c->tgt, n, 0);   /* NormalMix :298-304 */
    fpu_leave(csr);
}

void oal_conv_destroy(oal_conv *c)
{
    if(!c) return;
    free(c->ir); free(c->hist); free(c);
}

/* ======================================================================== *
 * EAX reverb: ReverbState::process and what it calls, alc/effects/reverb.cpp.
 * The parameter side (ReverbState::update) is NOT restated here: the block the compiled
 * reference computed (oal_reverb_get_params) -- or the product host's -- is installed with
 * oal_reverb_set_params, following which fields update() touches (:1263-1350).
 * ======================================================================== */
#define RV_LINES 4
#define RV_MAX_UPDATE 256            /* MAX_UPDATE_SAMPLES, reverb.cpp:68 */
#define RV_MOD_FRACBITS 24
#define RV_MOD_FRACONE (1u << RV_MOD_FRACBITS)
#define RV_MOD_FRACMASK (RV_MOD_FRACONE - 1u)
#define RV_CUBIC_BITS 8              /* CubicFilter::sTableBits, core/cubic_tables.h:23-25 */
#define RV_CUBIC_STEPS (1u << RV_CUBIC_BITS)
#define RV_CUBIC_MASK (RV_CUBIC_STEPS - 1u)
enum { RV_DEVICE_CLEAR, RV_START_FADE, RV_FADING, RV_CLEANUP, RV_NORMAL };   /* reverb.cpp:589-595 */

static float g_rv_cubic[RV_CUBIC_STEPS * 2 + 1];
static int g_rv_cubic_ready;
static void rv_cubic_build(void) /* CubicFilter::CubicFilter, core/cubic_tables.cpp:109-128 */
{
    if(g_rv_cubic_ready) return;
    const double index_scale = 512.0 / (double)(RV_CUBIC_STEPS * 2);
    for(unsigned i = 0; i < RV_CUBIC_STEPS / 2 + 1; ++i)
    {
        const double c0 = gauss_coeff((double)(RV_CUBIC_STEPS + i) * index_scale);
        const double c1 = gauss_coeff((double)i * index_scale);
        const double c2 = gauss_coeff((double)(RV_CUBIC_STEPS - i) * index_scale);
        const double c3 = gauss_coeff((double)(RV_CUBIC_STEPS * 2 - i) * index_scale);
        const double scale = 1.0 / (c0 + c1 + c2 + c3);
        g_rv_cubic[RV_CUBIC_STEPS + i] = (float)(c0 * scale);
        g_rv_cubic[i] = (float)(c1 * scale);
        g_rv_cubic[RV_CUBIC_STEPS - i] = (float)(c2 * scale);
        g_rv_cubic[RV_CUBIC_STEPS * 2 - i] = (float)(c3 * scale);
    }
    g_rv_cubic_ready = 1;
}
int oal_reverb_cubic_table(float *out513)
{
    rv_cubic_build();
    memcpy(out513, g_rv_cubic, sizeof(g_rv_cubic));
    return 0;
}

typedef struct rv_line { float *buf; size_t stride; } rv_line;     /* DelayLineU / DelayLineI: 4 lines */
typedef struct rv_bq { oal_bq c; float z1, z2; } rv_bq;            /* BiquadFilter */
typedef struct rv_gains { float cur[OAL_MAX_AMBI_CHANNELS], tgt[OAL_MAX_AMBI_CHANNELS]; } rv_gains;

typedef struct rv_pipeline {                                       /* ReverbPipeline, reverb.cpp:502-565 */
    rv_bq lp[RV_LINES], hp[RV_LINES];
    rv_line late_in;
    size_t early_tap[RV_LINES][2]; float early_coeff[2];
    size_t late_tap[RV_LINES][2];
    float mix_x, mix_y;
    rv_line eap; float eap_coeff; size_t eap_off[RV_LINES];        /* mEarly.Allpass */
    rv_line edelay; size_t e_off[RV_LINES]; float e_coeff;         /* mEarly.Delay/Offset/Coeff */
    rv_gains egains[RV_LINES];
    rv_line ldelay; size_t l_off[RV_LINES]; float density_gain;    /* mLate */
    float mid_gain[RV_LINES]; rv_bq t60hf[RV_LINES], t60lf[RV_LINES];
    unsigned mod_index, mod_step; float mod_depth;
    unsigned mod_delays[RV_MAX_UPDATE];
    rv_line vap; float vap_coeff; size_t vap_off[RV_LINES];        /* mLate.VecAp (interleaved line) */
    rv_gains lgains[RV_LINES];
    size_t fade_count;
} rv_pipeline;

struct oal_reverb {
    unsigned nlines;
    float *samples; size_t total;
    uint32_t lengths[11];
    int state, current;
    rv_line main_delay;
    rv_pipeline pipe[2];
    size_t offset;
    float temp[RV_LINES][RV_MAX_UPDATE];
    float early[RV_LINES][LINE], late[RV_LINES][LINE];
};

static unsigned rv_next_pow2(unsigned v) /* NextPowerOf2, common/alnumeric.h */
{
    if(v > 0) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; }
    return v + 1;
}
static size_t rv_line_length(float length, float frequency, unsigned extra) /* calcLineLength :273-284 */
{ return (size_t)rv_next_pow2(float2uint(ceilf(length * frequency)) + extra) * RV_LINES; }

static const float RV_EARLY_TAP[4] = {0.000000e+0f, 1.010676e-3f, 2.126553e-3f, 3.358580e-3f};
static const float RV_EARLY_AP[4] = {4.854840e-4f, 5.360178e-4f, 5.918117e-4f, 6.534130e-4f};
static const float RV_EARLY_LINE[4] = {2.992520e-3f, 5.456575e-3f, 7.688329e-3f, 9.709681e-3f};
static const float RV_LATE_AP[4] = {8.091400e-4f, 1.019453e-3f, 1.407968e-3f, 1.618280e-3f};
static const float RV_LATE_LINE[4] = {9.709681e-3f, 1.223343e-2f, 1.689561e-2f, 1.941936e-2f};

static void rv_pipeline_clear(rv_pipeline *p) /* ReverbPipeline::clear :550-564 */
{
    memset(p->late_in.buf, 0, p->late_in.stride * RV_LINES * sizeof(float));
    for(int j = 0; j < RV_LINES; ++j) { p->lp[j].z1 = p->lp[j].z2 = p->hp[j].z1 = p->hp[j].z2 = 0.0f; }
    memset(p->early_tap, 0, sizeof(p->early_tap));
    p->early_coeff[0] = p->early_coeff[1] = 0.0f;
    memset(p->late_tap, 0, sizeof(p->late_tap));
    memset(p->eap.buf, 0, p->eap.stride * RV_LINES * sizeof(float));
    memset(p->edelay.buf, 0, p->edelay.stride * RV_LINES * sizeof(float));
    memset(p->egains, 0, sizeof(p->egains));
    memset(p->vap.buf, 0, p->vap.stride * RV_LINES * sizeof(float));
    memset(p->ldelay.buf, 0, p->ldelay.stride * RV_LINES * sizeof(float));
    for(int j = 0; j < RV_LINES; ++j) { p->t60hf[j].z1 = p->t60hf[j].z2 = p->t60lf[j].z1 = p->t60lf[j].z2 = 0.0f; }
    p->mod_index = 0u; p->mod_step = 1u; p->mod_depth = 0.0f;
    memset(p->lgains, 0, sizeof(p->lgains));
}

oal_reverb *oal_reverb_create(uint32_t sample_rate, uint32_t num_out_lines)
{
    if(num_out_lines > OAL_MAX_AMBI_CHANNELS) return NULL;
    rv_cubic_build();
    oal_reverb *r = (oal_reverb*)calloc(1, sizeof(*r));
    r->nlines = num_out_lines;
    /* allocLines, reverb.cpp:728-820 */
    const float frequency = (float)sample_rate;
    const float multiplier = fmaxf(1.0f, cbrtf(1.0f * 1000.0f));
    const float max_mod_delay = 4.0f * 0.05f / 2.0f;
    const unsigned late_vecap_extra = float2uint(ceilf(RV_LATE_AP[0] * multiplier * frequency));
    const float late_diff_avg = (RV_LATE_LINE[3] - RV_LATE_LINE[0]) / 4.0f;
    size_t k = 0;
    r->lengths[k++] = (uint32_t)rv_line_length(0.3f + RV_EARLY_TAP[3] * multiplier, frequency, LINE);
    for(int p = 0; p < 2; ++p)
    {
        r->lengths[k++] = (uint32_t)rv_line_length(0.1f + late_diff_avg * multiplier, frequency, LINE);
        r->lengths[k++] = (uint32_t)rv_line_length(RV_EARLY_AP[3] * multiplier, frequency, 0);
        r->lengths[k++] = (uint32_t)rv_line_length(RV_EARLY_LINE[3] * multiplier, frequency, RV_MAX_UPDATE);
        r->lengths[k++] = (uint32_t)rv_line_length(RV_LATE_AP[3] * multiplier, frequency, late_vecap_extra);
        r->lengths[k++] = (uint32_t)rv_line_length(RV_LATE_LINE[3] * multiplier + max_mod_delay, frequency, 4);
    }
    for(k = 0; k < 11; ++k) r->total += r->lengths[k];
    r->samples = (float*)calloc(r->total, sizeof(float));
    float *at = r->samples;
    k = 0;
#define RV_TAKE(line) do { (line).buf = at; (line).stride = r->lengths[k] / RV_LINES; at += r->lengths[k++]; } while(0)
    RV_TAKE(r->main_delay);
    for(int p = 0; p < 2; ++p)
    {
        RV_TAKE(r->pipe[p].late_in); RV_TAKE(r->pipe[p].eap); RV_TAKE(r->pipe[p].edelay);
        RV_TAKE(r->pipe[p].vap); RV_TAKE(r->pipe[p].ldelay);
    }
#undef RV_TAKE
    /* deviceUpdate :829-833; constructor defaults of ReverbPipeline (:526-535) and BiquadFilter */
    for(int p = 0; p < 2; ++p)
    {
        rv_pipeline *P = &r->pipe[p];
        rv_pipeline_clear(P);
        P->mix_x = 1.0f; P->mix_y = 0.0f; P->fade_count = 1;
        for(int j = 0; j < RV_LINES; ++j)
            P->lp[j].c.b0 = P->hp[j].c.b0 = P->t60hf[j].c.b0 = P->t60lf[j].c.b0 = 1.0f;
    }
    r->state = RV_DEVICE_CLEAR;
    r->current = 0;
    return r;
}

void oal_reverb_destroy(oal_reverb *r) { if(r) { free(r->samples); free(r); } }

int oal_reverb_update(oal_reverb *r, const oal_reverb_props *props, float slot_gain)
{ (void)r; (void)props; (void)slot_gain; return -1; }

int oal_reverb_get_params(oal_reverb *r, oal_reverb_params *out) { (void)r; (void)out; return -1; }

int oal_reverb_line_lengths(oal_reverb *r, uint32_t *lengths11)
{
    memcpy(lengths11, r->lengths, sizeof(r->lengths));
    return (int)r->total;
}

/* What ReverbState::update writes (:1263-1350).  A toggled current_pipeline means the update was
 * a full one: the state becomes StartFade/Normal and the old pipeline's early coefficient target
 * drops to 0 (:1275-1279).  Only the current pipeline's parameters are replaced; everything
 * process() itself mutates (taps[.][0], gains.Current, the old pipeline's targets and fade count)
 * is left alone. */
int oal_reverb_set_params(oal_reverb *r, const oal_reverb_params *q)
{
    if(q->current_pipeline != r->current)
    {
        r->state = q->pipeline_state;
        r->current = q->current_pipeline;
        r->pipe[!r->current].early_coeff[1] = 0.0f;
    }
    rv_pipeline *P = &r->pipe[r->current];
    const oal_reverb_pipeline *s = &q->pipe[r->current];
    for(int j = 0; j < RV_LINES; ++j)
    {
        P->lp[j].c = s->filter_lp; P->hp[j].c = s->filter_hp;
        P->early_tap[j][1] = s->early_delay_tap[j][1];
        P->late_tap[j][1] = s->late_delay_tap[j][1];
        P->eap_off[j] = s->early_ap_offset[j];
        P->e_off[j] = s->early_offset[j];
        P->l_off[j] = s->late_offset[j];
        P->vap_off[j] = s->late_ap_offset[j];
        P->mid_gain[j] = s->t60_mid_gain[j];
        P->t60hf[j].c = s->t60_hf[j]; P->t60lf[j].c = s->t60_lf[j];
        memcpy(P->egains[j].tgt, s->early_gains_target[j], sizeof(P->egains[j].tgt));
        memcpy(P->lgains[j].tgt, s->late_gains_target[j], sizeof(P->lgains[j].tgt));
    }
    P->early_coeff[1] = s->early_delay_coeff[1];
    P->mix_x = s->mix_x; P->mix_y = s->mix_y;
    P->eap_coeff = s->early_ap_coeff; P->e_coeff = s->early_coeff;
    P->density_gain = s->late_density_gain;
    P->mod_step = s->mod_step; P->mod_depth = s->mod_depth;
    P->vap_coeff = s->late_ap_coeff;
    P->fade_count = s->fade_sample_count;
    return 0;
}

static void rv_dual_biquad(rv_bq *f0, rv_bq *f1, const float *src, float *dst, size_t n)
{   /* DualBiquad{f0,f1}.process -> BiquadFilter::dualProcess, core/filters/biquad.cpp:254-282 */
    float z01 = f0->z1, z02 = f0->z2, z11 = f1->z1, z12 = f1->z2;
    for(size_t i = 0; i < n; ++i)
    {
        const float x0 = src[i];
        const float y0 = x0 * f0->c.b0 + z01;
        z01 = x0 * f0->c.b1 - y0 * f0->c.a1 + z02;
        z02 = x0 * f0->c.b2 - y0 * f0->c.a2;
        const float y1 = y0 * f1->c.b0 + z11;
        z11 = y0 * f1->c.b1 - y1 * f1->c.a1 + z12;
        z12 = y0 * f1->c.b2 - y1 * f1->c.a2;
        dst[i] = y1;
    }
    f0->z1 = z01; f0->z2 = z02; f1->z1 = z11; f1->z2 = z12;
}

static void rv_partial_scatter(float out[4], const float in[4], float x, float y) /* :1396-1405 */
{
    const float o0 = x * in[0] + y * (in[1] + -in[2] + in[3]);
    const float o1 = x * in[1] + y * (-in[0] + in[2] + in[3]);
    const float o2 = x * in[2] + y * (in[0] + -in[1] + in[3]);
    const float o3 = x * in[3] + y * (-in[0] + -in[1] + -in[2]);
    out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3;
}

static void rv_line_write(const rv_line *l, size_t offset, size_t c, const float *in, size_t n) /* DelayLineU::write :312-326 */
{
    float *out = l->buf + c * l->stride;
    for(size_t i = 0; i < n; ++i) out[(offset + i) & (l->stride - 1)] = in[i];
}

static void rv_allpass4(rv_pipeline *P, float (*samples)[RV_MAX_UPDATE], size_t offset, size_t todo)
{   /* Allpass4::process :1508-1540 */
    const float feed = P->eap_coeff;
    const size_t mask = P->eap.stride - 1;
    for(size_t j = 0; j < RV_LINES; ++j)
    {
        float *buf = P->eap.buf + j * P->eap.stride;
        size_t dst = offset, tap = offset - P->eap_off[j];
        for(size_t i = 0; i < todo; ++i)
        {
            const float x = samples[j][i];
            const float y = buf[(tap++) & mask] - feed * x;
            buf[(dst++) & mask] = x + feed * y;
            samples[j][i] = y;
        }
    }
}

static void rv_vec_allpass(rv_pipeline *P, float (*samples)[RV_MAX_UPDATE], size_t offset, float xc, float yc,
    size_t todo)
{   /* VecAllpass::process :1452-1503 */
    const size_t mask = P->vap.stride - 1;
    float *buf = P->vap.buf;
    const float feed = P->vap_coeff;
    for(size_t base = 0; base < todo;)
    {
        size_t vo[RV_LINES], maxoff;
        for(int c = 0; c < RV_LINES; ++c) vo[c] = (offset - P->vap_off[c]) & mask;
        offset &= mask;
        maxoff = offset;
        for(int c = 0; c < RV_LINES; ++c) if(vo[c] > maxoff) maxoff = vo[c];
        size_t td = mask + 1 - maxoff;
        if(todo - base < td) td = todo - base;
        if(P->vap_off[0] < td) td = P->vap_off[0];
        for(int c = 0; c < RV_LINES; ++c)
        {
            size_t out_off = vo[c], in_off = offset;
            for(size_t i = 0; i < td; ++i)
            {
                const float input = samples[c][base + i];
                const float out = buf[(out_off++) * RV_LINES + (size_t)c] - feed * input;
                buf[(in_off++) * RV_LINES + (size_t)c] = input + feed * out;
                samples[c][base + i] = out;
            }
        }
        for(size_t j = 0; j < td; ++j)
        {
            float *d = buf + (offset + j) * RV_LINES;
            rv_partial_scatter(d, d, xc, yc);
        }
        offset += td;
        base += td;
    }
}

static void rv_process_early(oal_reverb *r, rv_pipeline *P, size_t offset, size_t samplesToDo)
{   /* ReverbPipeline::processEarly :1558-1660 */
    const rv_line *in_delay = &r->main_delay;
    const size_t inmask = in_delay->stride - 1;
    for(size_t base = 0; base < samplesToDo;)
    {
        const size_t todo = (samplesToDo - base < RV_MAX_UPDATE) ? samplesToDo - base : RV_MAX_UPDATE;
        const float fadeStep = 1.0f / (float)todo;
        const float c0 = P->early_coeff[0], c1 = P->early_coeff[1];
        P->early_coeff[0] = P->early_coeff[1];
        for(size_t j = 0; j < RV_LINES; ++j)
        {
            const float *input = in_delay->buf + j * in_delay->stride;
            size_t tap0 = offset - P->early_tap[j][0], tap1 = offset - P->early_tap[j][1];
            P->early_tap[j][0] = P->early_tap[j][1];
            float fadeCount = 0.0f;
            for(size_t i = 0; i < todo; ++i)
            {
                const float in0 = input[(tap0++) & inmask], in1 = input[(tap1++) & inmask];
                r->temp[j][i] = lerpf(in0 * c0, in1 * c1, fadeStep * fadeCount);
                fadeCount += 1.0f;
            }
            rv_dual_biquad(&P->lp[j], &P->hp[j], r->temp[j], r->temp[j], todo);
        }
        rv_allpass4(P, r->temp, offset, todo);
        /* DelayLineU::writeReflected :340-365 */
        {
            const size_t mask = P->edelay.stride - 1;
            float *b = P->edelay.buf;
            const size_t st = P->edelay.stride;
            for(size_t i = 0; i < todo; ++i)
            {
                const float s0 = r->temp[0][i], s1 = r->temp[1][i], s2 = r->temp[2][i], s3 = r->temp[3][i];
                const size_t o = (offset + i) & mask;
                b[0 * st + o] = (s0 - s1 - s2 - s3) * 0.5f;
                b[1 * st + o] = (s1 - s0 - s2 - s3) * 0.5f;
                b[2 * st + o] = (s2 - s0 - s1 - s3) * 0.5f;
                b[3 * st + o] = (s3 - s0 - s1 - s2) * 0.5f;
            }
        }
        for(size_t j = 0; j < RV_LINES; ++j)
        {
            const float *buf = P->edelay.buf + j * P->edelay.stride;
            const size_t mask = P->edelay.stride - 1;
            size_t tap = offset - P->e_off[j];
            for(size_t i = 0; i < todo; ++i)
                r->early[j][base + i] = buf[(tap++) & mask] * P->e_coeff + r->temp[j][i];
        }
        for(size_t i = 0; i < todo; ++i)                      /* VectorScatter :1408-1423 */
        {
            float v[4] = {r->temp[0][i], r->temp[1][i], r->temp[2][i], r->temp[3][i]};
            rv_partial_scatter(v, v, P->mix_x, P->mix_y);
            r->temp[0][i] = v[0]; r->temp[1][i] = v[1]; r->temp[2][i] = v[2]; r->temp[3][i] = v[3];
        }
        for(size_t j = 0; j < RV_LINES; ++j) rv_line_write(&P->late_in, offset, j, r->temp[j], todo);
        base += todo;
        offset += todo;
    }
}

static void rv_calc_delays(rv_pipeline *P, size_t todo) /* Modulation::calcDelays :1662-1682 */
{
    unsigned idx = P->mod_index;
    const unsigned step = P->mod_step;
    const float depth = P->mod_depth * (float)RV_CUBIC_STEPS;
    for(size_t i = 0; i < todo; ++i)
    {
        const float x = (float)(idx & RV_MOD_FRACMASK) * (1.0f / (float)RV_MOD_FRACONE);
        const float lfo = !(idx & (RV_MOD_FRACONE >> 1))
            ? ((-16.0f * x * x) + (8.0f * x))
            : ((16.0f * x * x) + (-8.0f * x) + (-16.0f * x) + 8.0f);
        idx += step;
        P->mod_delays[i] = float2uint((lfo + 1.0f) * depth);
    }
    P->mod_index = idx;
}

static void rv_process_late(oal_reverb *r, rv_pipeline *P, size_t offset, size_t samplesToDo)
{   /* ReverbPipeline::processLate :1696-1811 */
    for(size_t base = 0; base < samplesToDo;)
    {
        size_t todo = P->l_off[0] < RV_MAX_UPDATE ? P->l_off[0] : RV_MAX_UPDATE;
        if(samplesToDo - base < todo) todo = samplesToDo - base;
        rv_calc_delays(P, todo);
        for(size_t j = 0; j < RV_LINES; ++j)
        {
            const float *input = P->ldelay.buf + j * P->ldelay.stride;
            const size_t mask = P->ldelay.stride - 1;
            const float midGain = P->mid_gain[j];
            size_t tap = offset - P->l_off[j];
            for(size_t i = 0; i < todo; ++i)
            {
                const size_t idelay = P->mod_delays[i];
                const size_t delay = tap - (idelay >> RV_CUBIC_BITS);
                const size_t doff = idelay & RV_CUBIC_MASK;
                ++tap;
                const float out0 = input[(delay) & mask], out1 = input[(delay - 1) & mask];
                const float out2 = input[(delay - 2) & mask], out3 = input[(delay - 3) & mask];
                const float out = out0 * g_rv_cubic[RV_CUBIC_STEPS + doff] + out1 * g_rv_cubic[doff]
                    + out2 * g_rv_cubic[RV_CUBIC_STEPS - doff] + out3 * g_rv_cubic[RV_CUBIC_STEPS * 2 - doff];
                r->temp[j][i] = out * midGain;
            }
            rv_dual_biquad(&P->t60hf[j], &P->t60lf[j], r->temp[j], r->temp[j], todo);
        }
        const float fadeStep = 1.0f / (float)todo;
        for(size_t j = 0; j < RV_LINES; ++j)
        {
            const float *input = P->late_in.buf + j * P->late_in.stride;
            const size_t mask = P->late_in.stride - 1;
            size_t tap0 = offset - P->late_tap[j][0], tap1 = offset - P->late_tap[j][1];
            P->late_tap[j][0] = P->late_tap[j][1];
            const float densityGain = P->density_gain;
            const float densityStep = (tap0 != tap1) ? densityGain * fadeStep : 0.0f;
            float fadeCount = 0.0f;
            for(size_t i = 0; i < todo; ++i)
            {
                const float fade0 = densityGain - densityStep * fadeCount;
                const float fade1 = densityStep * fadeCount;
                fadeCount += 1.0f;
                r->temp[j][i] = input[(tap0++) & mask] * fade0 + input[(tap1++) & mask] * fade1 + r->temp[j][i];
            }
        }
        rv_vec_allpass(P, r->temp, offset, P->mix_x, P->mix_y, todo);
        for(size_t j = 0; j < RV_LINES; ++j) memcpy(&r->late[j][base], r->temp[j], todo * sizeof(float));
        for(size_t i = 0; i < todo; ++i)                      /* VectorScatterRev :1428-1443 */
        {
            const float v[4] = {r->temp[3][i], r->temp[2][i], r->temp[1][i], r->temp[0][i]};
            float o[4];
            rv_partial_scatter(o, v, P->mix_x, P->mix_y);
            r->temp[0][i] = o[0]; r->temp[1][i] = o[1]; r->temp[2][i] = o[2]; r->temp[3][i] = o[3];
        }
        for(size_t j = 0; j < RV_LINES; ++j) rv_line_write(&P->ldelay, offset, j, r->temp[j], todo);
        base += todo;
        offset += todo;
    }
}

static void rv_mix_out(oal_reverb *r, rv_pipeline *P, float *out_lines, size_t todo) /* MixOutPlain :637-656 */
{
    for(int j = 0; j < RV_LINES; ++j)
        mix_lines(r->early[j], todo, out_lines, r->nlines, P->egains[j].cur, P->egains[j].tgt, todo, 0);
    for(int j = 0; j < RV_LINES; ++j)
        mix_lines(r->late[j], todo, out_lines, r->nlines, P->lgains[j].cur, P->lgains[j].tgt, todo, 0);
}

void oal_reverb_process(oal_reverb *r, const float *wet_in, float *out_lines, uint32_t n)
{   /* ReverbState::process :1813-1883 */
    static const float B2A[4][4] = {{0.5f, 0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, -0.5f, 0.5f},
        {0.5f, 0.5f, -0.5f, -0.5f}, {0.5f, -0.5f, 0.5f, -0.5f}};      /* :91-97 */
    const unsigned csr = fpu_enter();
    const size_t offset = r->offset;
    rv_pipeline *oldp = &r->pipe[!r->current], *curp = &r->pipe[r->current];
    float tmp[LINE];
    for(size_t c = 0; c < RV_LINES; ++c)
    {
        for(size_t i = 0; i < n; ++i) tmp[i] = 0.0f;
        for(size_t k = 0; k < 4; ++k)
        {
            const float gain = B2A[c][k];
            for(size_t i = 0; i < n; ++i) tmp[i] = tmp[i] + wet_in[k * LINE + i] * gain;
        }
        rv_line_write(&r->main_delay, offset, c, tmp, n);
    }
    if(r->state < RV_FADING) r->state = RV_FADING;
    rv_process_early(r, curp, offset, n);
    rv_process_late(r, curp, offset, n);
    rv_mix_out(r, curp, out_lines, n);
    if(r->state != RV_NORMAL)
    {
        if(r->state == RV_CLEANUP)
        {
            rv_pipeline_clear(oldp);
            r->state = RV_NORMAL;
        }
        else
        {
            if(n >= oldp->fade_count)
            {
                for(int j = 0; j < RV_LINES; ++j)
                {
                    memset(oldp->egains[j].tgt, 0, sizeof(oldp->egains[j].tgt));
                    memset(oldp->lgains[j].tgt, 0, sizeof(oldp->lgains[j].tgt));
                }
                oldp->fade_count = 0;
                r->state = RV_CLEANUP;
            }
            else
                oldp->fade_count -= n;
            rv_process_early(r, oldp, offset, n);
            rv_process_late(r, oldp, offset, n);
            rv_mix_out(r, oldp, out_lines, n);
        }
    }
    r->offset = offset + n;
    fpu_leave(csr);
}
