// Dispatch table of tssep_gemm_f32 (gemm.hip, gemm_bf16x3*.hip): which kernel takes which request, and the
// split count a weight gradient should be launched with.  ONE place decides; the launcher, the plan query
// (tssep_gemm_plan) and the split query (tssep_gemm_wgrad_splits) all walk the same candidates, and a caller may
// name a kernel itself (tssep_gemm_f32_on: the A/B tools and the shape sweep time every candidate that way).
//
// Production builds carry NO run-time switch: `gemm_switches()` is a constant, nothing in the library reads the
// environment, nothing global is mutated (include/tssep_hip.h, conventions).  The experiment build (`make exp`,
// -DTSSEP_GEMM_EXP, selected with TSSEP_HIP_LIB) re-reads the historical TSSEP_GEMM_* variables on every call, so
// the alternating A/B scripts under tools/ keep working against that library.
#pragma once
#include <stdint.h>
#include <stdlib.h>

namespace gemm_detail {

struct GemmSwitches {
  int tall, big, big_p, big_p320, stream, nt_w160, wide, xcol;      // row x row family
  int tn, tn_big, tn_p320, tn_w160, tn_h160, tn_tall, tn_xc; // weight-gradient family (tn_tall: 4 = the tuned rule)
  int remap_wide, f32_rows;                         // store variants
  int hack;                                         // timing probes (experiment build only)
};

#ifdef TSSEP_GEMM_EXP
inline int env_int_(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && e[0] ? atoi(e) : dflt;
}
inline GemmSwitches gemm_switches() {
  GemmSwitches s;
  s.tall = env_int_("TSSEP_GEMM_TALL", 1);
  s.big = env_int_("TSSEP_GEMM_BIG", 1);
  s.big_p = env_int_("TSSEP_GEMM_BIG_P", 1);
  s.big_p320 = env_int_("TSSEP_GEMM_BIG_P320", 1);
  s.stream = env_int_("TSSEP_GEMM_STREAM", 1);
  s.nt_w160 = env_int_("TSSEP_GEMM_NT_W160", 1);
  s.wide = env_int_("TSSEP_GEMM_WIDE", 1);
  s.xcol = env_int_("TSSEP_GEMM_XCOL", 1);
  s.tn = env_int_("TSSEP_GEMM_TN", 1);
  s.tn_big = env_int_("TSSEP_GEMM_TN_BIG", 1);
  s.tn_p320 = env_int_("TSSEP_GEMM_TN_P320", 1);
  s.tn_w160 = env_int_("TSSEP_GEMM_TN_W160", 1);
  s.tn_h160 = env_int_("TSSEP_GEMM_TN_H160", 1);
  s.tn_tall = env_int_("TSSEP_GEMM_TN_TALL", 4);
  s.tn_xc = env_int_("TSSEP_GEMM_TN_XC", 1);
  s.remap_wide = env_int_("TSSEP_GEMM_REMAP_WIDE", 1);
  s.f32_rows = env_int_("TSSEP_GEMM_F32_ROWS", 1);
  s.hack = env_int_("TSSEP_GEMM_HACK", 0);
  return s;
}
#else
constexpr GemmSwitches gemm_switches() { return GemmSwitches{1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 4, 1, 1, 1, 0}; }
#endif

// How a launcher is being driven: launch (stream), or answer "would you take this?" without launching.
struct GemmCall {
  void* stream;
  bool dry;          // plan only: validity checks, no launch
  int32_t force;     // TSSEP_GEMM_AUTO or the one kernel the caller asked for
  int32_t chosen;    // out: the kernel that took (or would take) the request
};

// candidate `kid` is tried when the caller forced it, or -- in automatic mode -- when the rule wants it
inline bool gemm_try(const GemmCall& c, int32_t kid, bool rule_wants) {
  return c.force == 0 /* TSSEP_GEMM_AUTO */ ? rule_wants : c.force == kid;
}

}  // namespace gemm_detail
