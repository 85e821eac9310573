/* Hand-written stand-in for config_simd.h.in: x86-64 with SSE..SSE4.1, as the
 * reference's CMake configures on this host. */
#ifndef ORACLE_REFCFG_CONFIG_SIMD_H
#define ORACLE_REFCFG_CONFIG_SIMD_H
#define HAVE_SSE 1
#define HAVE_SSE2 1
#define HAVE_SSE3 1
#define HAVE_SSE4_1 1
#define HAVE_SSE_INTRINSICS 1
#define HAVE_NEON 0
#endif
