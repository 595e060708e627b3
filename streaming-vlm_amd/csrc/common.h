// Shared device helpers for the svlm HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits; all conversions are explicit

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16, 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define SVLM_OK 0
#define SVLM_EINVAL (-22)
#define SVLM_ELAUNCH (-5)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even, the rounding torch uses for fp32 -> bf16: the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
// (one instruction per TWO values; the integer-arithmetic form costs five per value and these helpers sit on the
// critical path of every RoPE-on-load and every epilogue)
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  hw_bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float rbf(float f) { return __uint_as_float(((unsigned)f2bf(f)) << 16); }  // round through bf16
__device__ __forceinline__ float lo_bf(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_bf(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

__device__ __forceinline__ void unpack8(const u32x4_t& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = lo_bf(v[i]); f[2 * i + 1] = hi_bf(v[i]); }
}
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
  return v;
}

// Wave-wide reductions on the VALU only: DPP quad_perm / row mirrors inside each 16-lane row, then
// v_permlane16/32_swap across rows (a __shfl_xor butterfly is six ds_bpermute round trips through the LDS crossbar).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum(float v) {      // all 16 lanes of a row end up with the row total
  v += dpp_f<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);    // row_half_mirror
  v += dpp_f<0x140>(v);    // row_mirror
  return v;
}
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum(v);
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
  v = row_max(v);
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// Cross-row reductions without the LDS crossbar: __shfl_xor lowers to ds_bpermute (~100 cycles of latency on the
// critical path of a one-wave-per-SIMD attention loop); v_permlane16/32_swap are plain VALU.
// permlaneNN_swap(x, x) returns {x with odd/upper part replaced by the even/lower part, x with even/lower part
// replaced by the odd/upper part}, so op(r[0], r[1]) is the xor-16 / xor-32 butterfly step for max and sum.
__device__ __forceinline__ float xor16_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Workgroup barrier for LDS hand-offs that leaves global loads IN FLIGHT.  __syncthreads() carries a release fence
// for which hipcc emits s_waitcnt vmcnt(0) (gfx950 counts loads and stores on one counter), draining a software
// prefetch ring at every K-tile; here only this wave's LDS traffic is waited for.  The asm memory clobber keeps
// the compiler from moving LDS/global accesses across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// activation codes shared with include/svlm.h
#define SVLM_ACT_NONE 0
#define SVLM_ACT_QUICK_GELU 1
#define SVLM_ACT_GELU_ERF 2
#define SVLM_ACT_SILU 3
#define SVLM_ACT_SWIGLU 4      // GEMM only: W = [gate rows; up rows], C[m, n] = silu(gate_n . a_m) * (up_n . a_m)

// y is the bf16-rounded linear output (as float); returns the activation with the eager
// module's intermediate roundings (each torch op rounds to bf16).
__device__ __forceinline__ float apply_act(float y, int act) {
  if (act == SVLM_ACT_QUICK_GELU) {          // x * sigmoid(1.702 * x): three roundings
    float t = rbf(1.702f * y);
    float s = rbf(1.0f / (1.0f + expf(-t)));
    return rbf(y * s);
  } else if (act == SVLM_ACT_GELU_ERF) {     // nn.GELU(): one rounding
    return rbf(0.5f * y * (1.0f + erff(y * 0.70710678118654752440f)));
  } else if (act == SVLM_ACT_SILU) {         // F.silu: one rounding
    return rbf(y / (1.0f + expf(-y)));
  }
  return y;
}

// ---- Philox4x32-10 (Salmon et al., SC'11): counter-based, no state to carry between launches
__device__ __forceinline__ unsigned svlm_philox_first(unsigned c0, unsigned c1, unsigned k0, unsigned k1) {
  unsigned c2 = 0u, c3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}
// uniform in (0, 1): 23 random bits + 1/2, exactly representable in fp32, never 0 or 1
__device__ __forceinline__ float svlm_philox_uniform(const unsigned* __restrict__ rng, unsigned step, unsigned idx) {
  return ((float)(svlm_philox_first(idx, step, rng[0], rng[1]) >> 9) + 0.5f) * (1.0f / 8388608.0f);
}
// standard Gumbel noise g = -log(E), E ~ Exp(1).  The draw that WINS an argmax over a vocabulary of ~1.5e5 sits in the upper tail
// (g ~ 12, E ~ 6e-6): E = -log1p(-v) with v = (x + 1/2) * 2^-32 keeps all 32 random bits where E is small (v is exact in fp32 for
// x < 2^24), so that tail is resolved to ~1e-10 instead of the 6e-8 steps of a 24-bit uniform; v = 1 (x rounds up to 2^32,
// probability 2^-25) gives g = -inf, a token that simply does not win.
__device__ __forceinline__ float svlm_gumbel_noise(const unsigned* __restrict__ rng, unsigned step, unsigned idx) {
  const float v = ((float)svlm_philox_first(idx, step, rng[0], rng[1]) + 0.5f) * (1.0f / 4294967296.0f);
  return -logf(-log1pf(-v));
}

// The product library takes NO behaviour from the environment.  The tuning switches of the launchers (tile shapes, split counts,
// kernel variants, the timing-only DIAG builds of the decode attention) read it through svlm_env(), which returns nullptr unless the
// library is the DIAGNOSTIC build (-DSVLM_TUNING: tools/build_diag_lib.py -> streaming-vlm_amd/build/libsvlm_hip_diag.so, loaded
// through SVLM_LIB_PATH by the tools that sweep those switches).
#include <stdlib.h>
#ifdef SVLM_TUNING
static inline const char* svlm_env(const char* name) { return getenv(name); }
#else
static inline const char* svlm_env(const char*) { return nullptr; }
#endif

void svlm_set_error(const char* fmt, ...);
#define SVLM_CHECK_ARG(cond, ...)                    \
  do {                                               \
    if (!(cond)) {                                   \
      svlm_set_error(__VA_ARGS__);                   \
      return SVLM_EINVAL;                            \
    }                                                \
  } while (0)
int svlm_check_launch(const char* what);
