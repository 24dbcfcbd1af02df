// gmr_ik_prof.h -- diagnostic phase timer shared by the IK kernels (GMR_IK_PROFILE builds only; never in the
// shipped kernels): s_memtime stamps accumulated per phase, written to a buffer no other code reads.
#pragma once
#include <hip/hip_runtime.h>

namespace gmr {

enum { PH_PRE, PH_FK, PH_ERR, PH_JLOG, PH_PAIRS, PH_CVEC, PH_HACC, PH_KBUILD, PH_CHOL, PH_SUBST, PH_RATIO,
       PH_MULT, PH_INTEG, PH_IO, PH_NFACT, PH_NSOLVE, PH_TICKS, PH_REALTIME, PH_COUNT };
#ifdef GMR_IK_PROFILE
struct Prof {
  unsigned long long acc[PH_COUNT];
  unsigned long long t0;
  __device__ __forceinline__ void begin() { t0 = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void end(int ph) { acc[ph] += __builtin_amdgcn_s_memtime() - t0; }
  __device__ __forceinline__ void count(int ph) { acc[ph] += 1; }
};
#define PROF_BEGIN(p) (p).begin()
#define PROF_END(p, ph) (p).end(ph)
#define PROF_COUNT(p, ph) (p).count(ph)
#else
struct Prof {};
#define PROF_BEGIN(p)
#define PROF_END(p, ph)
#define PROF_COUNT(p, ph)
#endif

}  // namespace gmr
