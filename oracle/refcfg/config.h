/* Hand-written stand-in for the header the reference's CMake would generate
 * from config.h.in (Linux x86-64, no optional deps).  TEST INFRASTRUCTURE:
 * used only to compile the reference sources in place into oracle/_ref/. */
#ifndef ORACLE_REFCFG_CONFIG_H
#define ORACLE_REFCFG_CONFIG_H
#define FORCE_ALIGN
#define HAVE_DLFCN_H
#define HAVE_CPUID_H
#define HAVE_GCC_GET_CPUID
#define HAVE_PTHREAD_SETSCHEDPARAM
#define HAVE_PTHREAD_SETNAME_NP
#define HAVE_CXXMODULES 0
#define HAVE_DYNLOAD 0
#define HAVE_RTKIT 0
#define ALSOFT_UWP 0
#define ALSOFT_EAX 0
#endif
