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
// Deterministic cross-workgroup sums: partial sums are converted to 64-bit fixed point (2^-30 resolution,
// +-8.6e9 range) and added with integer atomics, so the total does not depend on arrival order.
#define CVX_FIX_SCALE 1073741824.0f
__device__ __forceinline__ void cvx_fix_atomic_add(long long* dst, float v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__float2ll_rn(v * CVX_FIX_SCALE));
}
__device__ __forceinline__ double cvx_fix_to_double(long long v) { return (double)v * (1.0 / 1073741824.0); }

__device__ __forceinline__ float cvx_wave_max64(float v) {
  for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
