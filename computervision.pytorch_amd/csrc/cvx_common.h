// Shared device/host helpers for the MI355X (gfx950 / CDNA4) detection engine.
// Wave = 64 lanes everywhere; fp16 storage, fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

#define CVX_WAVE 64

// ---- error plumbing (C-ABI functions return int status, message via cvx_last_error) ----------
void cvx_set_error(const std::string& msg);
#define CVX_FAIL(msg)                                                         \
  do {                                                                        \
    cvx_set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + (msg)); \
    return -1;                                                                \
  } while (0)
#define CVX_HIP(call)                                                         \
  do {                                                                        \
    hipError_t e__ = (call);                                                  \
    if (e__ != hipSuccess) CVX_FAIL(std::string(#call) + " -> " + hipGetErrorString(e__)); \
  } while (0)
#define CVX_CHECK(cond, msg)                                                  \
  do {                                                                        \
    if (!(cond)) CVX_FAIL(std::string("check failed: ") + #cond + " : " + (msg)); \
  } while (0)
#define CVX_TRY(call)                                                         \
  do {                                                                        \
    int rc__ = (call);                                                        \
    if (rc__ != 0) return rc__;                                               \
  } while (0)

static inline int cvx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Tuning knobs (tile thresholds, ring depths, occupancy caps ...) are environment variables ONLY in a -DCVX_TUNING build;
// the release library ignores the environment entirely, so a stray variable can change neither numerics nor speed.
#ifdef CVX_TUNING
#include <cstdlib>
static inline int cvx_tune_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
static inline bool cvx_tune_set(const char* name) { return getenv(name) != nullptr; }
#else
static inline int cvx_tune_int(const char*, int dflt) { return dflt; }
static inline bool cvx_tune_set(const char*) { return false; }
#endif

// Opt-in to more than 64 KB of dynamic LDS for a kernel.  The attribute is per device: one bit per device ordinal in the
// caller's static mask, so a process that drives several GPUs (one engine each) opts in on each of them.
inline int cvx_lds_optin(const void* kernel, int bytes, unsigned long long* done_mask) {
  int dev = 0;
  CVX_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (*done_mask & bit) return 0;
  CVX_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  *done_mask |= bit;
  return 0;
}

// ---- small device helpers ---------------------------------------------------------------------
// v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the ~10-instruction IEEE division: the elementwise BN/SiLU passes are
// VALU-limited on the big layers, and every consumer rounds the result to fp16 anyway.
__device__ __forceinline__ float cvx_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float cvx_silu(float x) { return x * cvx_sigmoid(x); }
// d silu / dx
__device__ __forceinline__ float cvx_silu_grad(float x) {
  float s = cvx_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}
// sum over the 16 lanes that share (lane >> 4), every lane gets the total.  Four DPP adds (quad xor 1, quad xor 2,
// half-row mirror, row mirror) instead of four ds_bpermute round trips through the LDS pipe.
__device__ __forceinline__ float cvx_dpp_add(float v, float moved) { return v + moved; }
template <int CTRL>
__device__ __forceinline__ float cvx_dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float cvx_wave_sum16(float v) {
  v += cvx_dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += cvx_dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += cvx_dpp_mov<0x141>(v);  // row_half_mirror
  v += cvx_dpp_mov<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ float cvx_wave_sum64(float v) {
  v = cvx_wave_sum16(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// Deterministic cross-workgroup sums: every logical value is TWO 64-bit integer words that only ever see integer
// atomic adds, so the total does not depend on arrival order.  Word 0 holds round(v * 2^6) (range +-1.4e17: a sum of
// squares of fp16-bounded values over 1e7 rows fits), word 1 the exact remainder in units of 2^-40 (|remainder| <=
// 2^33 per add: no overflow below 2^30 adds).  One word at 2^-30 -- the first version -- wrapped silently once a sum of
// squares passed 8.6e9 (stem at batch 32 with pre-BN RMS > 51).
#define CVX_FIX_WORDS 2
__device__ __forceinline__ void cvx_fix_atomic_add(long long* slab, long long idx, float v) {
  const float s = v * 64.0f;                       // exact (power of two)
  const float c = rintf(s);                        // |s| >= 2^24: s is an integer already, c == s
  const long long coarse = __float2ll_rn(c);
  const long long fine = __float2ll_rn((s - c) * 17179869184.0f);  // (s - c) in [-0.5, 0.5], exact; * 2^34
  unsigned long long* d = reinterpret_cast<unsigned long long*>(slab + idx * CVX_FIX_WORDS);
  atomicAdd(d, (unsigned long long)coarse);
  if (fine != 0) atomicAdd(d + 1, (unsigned long long)fine);
}
__device__ __forceinline__ double cvx_fix_to_double(long long coarse, long long fine) {
  return (double)coarse * (1.0 / 64.0) + (double)fine * (1.0 / 1099511627776.0);
}

__device__ __forceinline__ float cvx_wave_max64(float v) {
  for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
