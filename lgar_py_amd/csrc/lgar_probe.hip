// lgar_probe.hip -- vector-ALU issue-rate probe for gfx950 (the compute-side roof of the LGAR path).
//
// The LGAR column update is bound by VALU issue (≈10^3 flop per algorithmic byte), and most of its instructions are the
// v_log_f32 / v_exp_f32 pairs of the 121-node Geff trapezoid.  bench.py reports achieved transcendentals/s against the
// rate THIS chip sustains, measured here: every wave runs `iters` iterations of 64 inline-asm instructions of one kind
// on 8 independent register chains (or one dependent chain for the latency ops), with a chosen number of resident waves
// per SIMD (set by the launch's dynamic LDS size: 160 KiB / (4 k) per one-wave workgroup => k waves per SIMD).
#include <hip/hip_runtime.h>

#include "../../include/lgar.h"

namespace lgar {

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void *probe_lds_base() {
  extern __shared__ char probe_lds[];
  return probe_lds;
}

template <int OP> __device__ __forceinline__ void probe_body(float (&v)[8], f32x2 (&p)[8], double (&d)[8], float c) {
  if (OP == LGAR_PROBE_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_LOG) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_SQRT) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : );
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_CMP) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(c) : "vcc");
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_EXP_DEP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_FMA_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(c));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_FMA64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_MUL64) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_ADD64) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_RCP64) {
#define X(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_CNDMASK_SGPR) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(v[i]) : "v"(c) : "s10", "s11");
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_BFI) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(v[i]) : "v"(c));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_CMP_CNDMASK) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc");
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  } else if (OP == LGAR_PROBE_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_READLANE) {
#define X(i) asm volatile("v_readlane_b32 s10, %0, 3" : : "v"(v[i]) : "s10");
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_DS_READ) {
    float *lp = (float *)probe_lds_base();
#define X(i) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"((unsigned)(threadIdx.x * 4 + (i) * 256)));
    REP64(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)");
    (void)lp;
  } else if (OP == LGAR_PROBE_MIN) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_LDEXP64) {
#define X(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(v[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_FREXP_EXP64) {
#define X(i) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(v[i]) : "v"(d[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_RNDNE64) {
#define X(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_CVT_I32_F64) {
#define X(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(v[i]) : "v"(d[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_CVT_F64_I32) {
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(v[i]));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    REP64(X)
#undef X
  } else if (OP == LGAR_PROBE_GEFF_MIX) {
    // the instruction stream of one iteration of the lean fp32 Geff loop (lgar_device.hpp geff<float>: two trapezoid
    // nodes): 4 v_log + 4 v_exp + 3 v_pk_fma + 5 v_pk_mul + 2 v_pk_add, in program order with its dependences; four node
    // pairs = 72 instructions per probe iteration
#define NODEPAIR(a, b)                                                                                 \
    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[a]) : "v"(p[b]));                                 \
    asm volatile("v_log_f32 %0, %1" : "=v"(v[a]) : "v"(p[a].x));                                          \
    asm volatile("v_log_f32 %0, %1" : "=v"(v[b]) : "v"(p[a].y));                                          \
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[b]) : "v"(p[a]));                                     \
    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[a]) : "v"(p[b]));                                     \
    asm volatile("v_exp_f32 %0, %0" : "+v"(v[a]));                                                        \
    asm volatile("v_exp_f32 %0, %0" : "+v"(v[b]));                                                        \
    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[a]) : "v"(p[b]));                                 \
    asm volatile("v_log_f32 %0, %0" : "+v"(v[a]));                                                        \
    asm volatile("v_log_f32 %0, %0" : "+v"(v[b]));                                                        \
    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[a]) : "v"(p[b]));                                     \
    asm volatile("v_exp_f32 %0, %0" : "+v"(v[a]));                                                        \
    asm volatile("v_exp_f32 %0, %0" : "+v"(v[b]));                                                        \
    asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[a]));                                                 \
    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[a]) : "v"(p[b]));                                 \
    asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[a]));                                                 \
    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[a]) : "v"(p[b]));                                     \
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[b + 1]) : "v"(p[a]));
    NODEPAIR(0, 2)
    NODEPAIR(4, 6)
    NODEPAIR(0, 2)
    NODEPAIR(4, 6)
#undef NODEPAIR
  }
}

template <int OP> __global__ __launch_bounds__(64) void lgar_probe_kernel(int iters, float *sink) {
  extern __shared__ char probe_lds[];  // only sizes the occupancy
  float v[8];
  f32x2 p[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    v[i] = 1.0f + 1e-3f * float(threadIdx.x + i);
    p[i].x = v[i];
    p[i].y = v[i] + 0.5f;
    d[i] = 1.0 + 1e-9 * double(threadIdx.x + i);
  }
  const float c = 1.0f + 1e-7f * float(blockIdx.x & 7);
  for (int it = 0; it < iters; it++) probe_body<OP>(v, p, d, c);
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += v[i] + p[i].x + p[i].y + (float)d[i];
  if (s == 123.456f) sink[blockIdx.x * 64 + threadIdx.x] = s + probe_lds[threadIdx.x];  // never true: keeps the chains live
}

}  // namespace lgar

using namespace lgar;

extern "C" int32_t lgar_valu_probe_insts(int32_t op) { return op == LGAR_PROBE_GEFF_MIX ? 72 : 64; }

extern "C" int32_t lgar_valu_probe(int32_t op, int32_t n_workgroups, int32_t lds_bytes_per_workgroup, int32_t iters,
                                   void *sink, void *stream) {
  if (n_workgroups <= 0 || iters <= 0 || lds_bytes_per_workgroup < 0 || lds_bytes_per_workgroup > 160 * 1024 || !sink)
    return LGAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(n_workgroups), b(64);
  const size_t lds = (size_t)lds_bytes_per_workgroup;
#define CASE(OP)                                                                                        \
  case OP:                                                                                              \
    if (lds > 64 * 1024)                                                                                \
      (void)hipFuncSetAttribute((const void *)lgar_probe_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(lgar_probe_kernel<OP>, g, b, lds, st, iters, (float *)sink);                     \
    break;
  switch (op) {
    CASE(LGAR_PROBE_EXP) CASE(LGAR_PROBE_LOG) CASE(LGAR_PROBE_RCP) CASE(LGAR_PROBE_SQRT) CASE(LGAR_PROBE_FMA)
    CASE(LGAR_PROBE_MUL) CASE(LGAR_PROBE_PK_FMA) CASE(LGAR_PROBE_PK_MUL) CASE(LGAR_PROBE_CNDMASK) CASE(LGAR_PROBE_CMP)
    CASE(LGAR_PROBE_EXP_DEP) CASE(LGAR_PROBE_FMA_DEP) CASE(LGAR_PROBE_FMA64) CASE(LGAR_PROBE_MUL64)
    CASE(LGAR_PROBE_ADD64) CASE(LGAR_PROBE_RCP64) CASE(LGAR_PROBE_GEFF_MIX) CASE(LGAR_PROBE_CNDMASK_SGPR)
    CASE(LGAR_PROBE_BFI) CASE(LGAR_PROBE_CMP_CNDMASK) CASE(LGAR_PROBE_ADD) CASE(LGAR_PROBE_READLANE)
    CASE(LGAR_PROBE_DS_READ) CASE(LGAR_PROBE_MIN) CASE(LGAR_PROBE_LDEXP64) CASE(LGAR_PROBE_FREXP_EXP64)
    CASE(LGAR_PROBE_RNDNE64) CASE(LGAR_PROBE_CVT_I32_F64) CASE(LGAR_PROBE_CVT_F64_I32) CASE(LGAR_PROBE_ADD_U32)
    default: return LGAR_E_ARG;
  }
#undef CASE
  return hipGetLastError() == hipSuccess ? 0 : LGAR_E_LAUNCH;
}
