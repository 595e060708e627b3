// Persistent decode-layer tail (T = 1): everything of one decoder layer that follows the attention, and the QKV projection of the
// NEXT layer, as ONE launch of one 4-wave workgroup per CU:
//
//   x'  = x + W_o . attn                                   (qwen2/language_forward.py:161,196)       phase O
//   h   = silu(W_g . n2(x')) * (W_u . n2(x'))              (Qwen2MLP, :198-201)                       phase GU
//   x'' = x' + W_d . h                                     (:201-202)                                 phase DOWN
//   [q|k|v] = W_qkv' . n1'(x'') + b'  -> q buffer, k / v straight into the new token's pool slot      phase QKV
//                                                          (next layer, :183,80-82 + streaming_cache.py:72-73)
//
// replacing the four launches  o_proj GEMV / gate-up GEMV / down_proj GEMV / next layer's QKV GEMV  of the per-op decode step
// (decode_fused.hip, gemv.hip).  Between those four ops every CU needs the WHOLE output vector of the op before (all-to-all seams),
// which is what a kernel boundary gives for free -- and what costs each launch its ~1.5 us boundary plus the ~2 us until its first
// weight bytes arrive, with 4-11 us of streaming per op.  Here the weight stream never stops at a seam:
//
//   * wave roles (cdna_hip_programming.md section 5.6, "1 loader + 3 consumers"): waves 0-2 of a workgroup are CONSUMERS, wave 3 is
//     the GATHERER.  A consumer walks one flattened list of weight batches over all four phases (a batch = NL 16-byte loads per
//     lane, straight into VGPRs, non-temporal) through a ring of three register buffers: while it waits at a seam, the next
//     phase's first three batches are already in flight or landed (3 waves x 3 x NL KB per CU).
//   * hand-offs are 8-byte {tag, 2 x bf16} granules (Guideline 16, form R2: the data IS the flag): a producer publishes each pair of
//     outputs with ONE relaxed agent-scope 8-byte store, the gatherer of EVERY workgroup sweeps the vector with relaxed agent-scope
//     loads until every tag is set, normalises it where the next op wants RMSNorm, and leaves it in LDS behind a workgroup barrier.
//     No fence, no flag, no ordering assumption; only the gatherer ever polls (vmcnt retires in order: a poll issued behind a
//     consumer's weight loads would wait for them).  Every (layer, seam) has its own granule block, zeroed once per decode step
//     (svlm_dec_tail_reset, a memset node at the head of the step's graph), so a tag is written once per step and never reused.
//   * every spin is bounded: a gatherer that gives up raises status[0] (sticky, checked by the host at the end of the chunk), stops
//     polling for the rest of the launch and lets its workgroup run to the end on whatever it has, so the grid always drains; later
//     launches see the status word and do not poll at all.
//   * results do not depend on placement: any workgroup may land on any CU (or two on one), rows are assigned by blockIdx only.
//
// Arithmetic is the per-op kernels' arithmetic: the same fp32 FMA order per row (8 elements per lane per 512-element chunk, chunks
// ascending, wave_sum), the same bf16 rounding points, the same 4 x 64-thread partial sums inside the RMSNorm; O / GU / QKV rows are
// bit-identical to svlm_gemv_bf16 / svlm_dec_gate_up / svlm_dec_qkv on the same inputs, DOWN differs in the fp32 summation order of
// its K split (3 waves here, 4 there).
#include "common.h"

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

#define TAIL_TAG 1ull
#define TAIL_SPIN_LIMIT (1u << 17)          // sweeps of one chunk before a gatherer gives up (~0.1-0.3 s)
#define TAIL_ERR_TIMEOUT 1

extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];

struct DecTailParams {
  const bf16_t* attn;      // [qd]   attention output of this layer (written by the launch before)
  bf16_t* x;               // [H]    residual stream: read as x, rewritten as x''
  // the four weight matrices by PHASE (0 o_proj [H][ld], 1 gate rows then up rows [2I][ld], 2 down_proj [H][ld], 3 the next layer's
  // QKV [qd + 2 kd][ld]) + entry 4, the filler behind the last batch (0 bytes: every load out of range).  Indexed with the
  // wave-uniform phase straight out of the kernel-argument segment: a select chain over four pointer / size / pitch triples kept in
  // SGPRs is what hipcc 7.2 got wrong once the loop spilled SGPRs (a null base in the in-loop QKV issue).
  const bf16_t* mat[5];
  unsigned mbytes[5];
  unsigned ld2[5];         // row pitch in bytes
  const bf16_t* ln2;       // [H]
  const bf16_t* ln1n;      // [H]    next layer's input norm, or null: no QKV phase (last layer)
  const bf16_t* qkv_b;     // [qd + 2 kd]
  bf16_t* q_out;           // [qd]
  bf16_t* k_planes;        // next layer's K planes [Hkv][n_slots][D]
  bf16_t* v_planes;
  const int* slot_of;
  const int* len_dev;      // device-side index of the new token (or null: len_host)
  u64* g_x1;               // [H / 2]  granules of x'
  u64* g_h;                // [I / 2]  granules of h
  u64* g_x2;               // [H / 2]  granules of x''
  int* status;
  u64* stamps;             // optional [grid][2][16] wall-clock stamps (tools/dec_tail_bench.py), else null
  int H, I, qd, kd, D, n_slots, len_host;
  float eps;
};

typedef __bf16 hw_bf2 __attribute__((ext_vector_type(2)));

// Weights and granules are read through buffer descriptors: a 32-bit lane offset plus a scalar row offset (no 64-bit address
// arithmetic on the VALU), and the range check makes every out-of-range slot -- rows behind the end of a matrix, K chunks behind the
// end of a wave's slice, the filler batches behind the last real one -- a load that returns zero WITHOUT touching memory.
#define TAIL_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((const void*)(ptr)), 0, (int)(bytes), 0x00020000)
#define TAIL_AUX_NT 2                       // streamed once: non-temporal
#define TAIL_AUX_SC1 16                     // agent-coherent: hand-off granules
#define TAIL_OOB 0x7ffff000u                // an offset behind every descriptor's range (matrices are < 2 GiB: host check)

__device__ __forceinline__ void publish(u64* g, int idx, unsigned payload) {
  __hip_atomic_store((gu64*)(g + idx), (TAIL_TAG << 32) | (u64)payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc += w . x over the 8 elements of one 16-byte piece: four v_dot2c_f32_bf16 on the packed pairs (the fp32-unpack + FMA form of
// the per-op kernels costs 16 VALU issues per piece, and one wave per SIMD issues one VALU instruction per 4 cycles: a consumer
// wave would be issue-bound at a third of the stream rate)
__device__ __forceinline__ void dot8(const u32x4_t& w, const u32x4_t& x, float& acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned wi = w[i], xi = x[i];     // scalars first: __builtin_bit_cast of a vector-element lvalue reads element 0 (hipcc 7.2)
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_bf2, wi), __builtin_bit_cast(hw_bf2, xi), acc, false);
  }
}

template <int NL>
struct TailBuf {
  u32x4_t v[NL];
  unsigned ex;             // per-lane epilogue operand requested with the batch (residual x[n] / bias[n])
};

// wave-uniform description of a consumer wave's share of the four phases
struct TailWork {
  int c0, c1, c2, c3;      // batches per phase (named, not an array: a run-time index would put the struct in scratch)
  __device__ __forceinline__ int cnt(int ph) const { return ph == 0 ? c0 : ph == 1 ? c1 : ph == 2 ? c2 : c3; }
  int o_b0, gu_u0, d_u0, q_b0;
  int nkb, kc0, kc1, nkc;  // DOWN: batches per row pair, this wave's chunk range [kc0, kc1) of the nkc chunks of a row
};

// Requests one batch: NL 16-byte loads + the epilogue operand.  The branches only pick the matrix and the NL scalar offsets; the loads
// themselves are ONE unconditional block (loads under branches end in register copies at the join, which wait for the data, and a
// path with fewer loads would make hipcc's wait-count bookkeeping assume it at every later wait).  phase == 4 is the filler behind
// the last real batch: out-of-range loads, no traffic.
template <int NC, int RB>
__device__ __forceinline__ void tail_issue(const DecTailParams& p, const TailWork& w, int phase, int j, TailBuf<NC * RB>& b, int lane) {
  constexpr int NL = NC * RB, NCD = NL / 2;
  const bf16_t* mat = p.mat[phase];
  const unsigned bytes = p.mbytes[phase], ld2 = p.ld2[phase];
  unsigned so[NL];
  const bf16_t* exa = p.x;
#pragma unroll
  for (int i = 0; i < NL; ++i) so[i] = 0u;       // phase 4: an empty range, every load returns zero without traffic
  if (phase == 2) {
    const int u = j / w.nkb, kb = j - u * w.nkb;
    const int cb = w.kc0 + kb * NCD;
    const unsigned r0 = (unsigned)(2 * (w.d_u0 + u)) * ld2;
    const unsigned r1 = r0 + ld2;
#pragma unroll
    for (int c = 0; c < NCD; ++c) {
      const bool in = cb + c < w.kc1;
      const unsigned ko = (unsigned)(cb + c) * 1024u;
      so[c] = in ? r0 + ko : TAIL_OOB;
      so[NCD + c] = in ? r1 + ko : TAIL_OOB;
    }
  } else if (phase < 4) {
    unsigned ro[RB];
    if (phase == 1) {
      if constexpr (RB == 4) {
        const unsigned n0 = 2u * (unsigned)(w.gu_u0 + j);
        ro[0] = n0 * ld2;
        ro[1] = ro[0] + ld2;
        ro[RB / 2] = ((unsigned)p.I + n0) * ld2;
        ro[RB / 2 + 1] = ro[RB / 2] + ld2;
      } else {
        const unsigned n0 = 2u * (unsigned)w.gu_u0 + (unsigned)j;
        ro[0] = n0 * ld2;
        ro[RB - 1] = ((unsigned)p.I + n0) * ld2;
      }
    } else {
      // O and QKV: RB consecutive rows; the lane's epilogue operand is x[n] / bias[n]
      const int N = phase == 0 ? p.H : p.qd + 2 * p.kd;
      const int row0 = ((phase == 0 ? w.o_b0 : w.q_b0) + j) * RB;
#pragma unroll
      for (int r = 0; r < RB; ++r) ro[r] = row0 + r < N ? (unsigned)(row0 + r) * ld2 : TAIL_OOB;
      exa = (phase == 0 ? p.x : p.qkv_b) + min(row0 + min(lane, RB - 1), N - 1);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
      for (int c = 0; c < NC; ++c) so[r * NC + c] = ro[r] == TAIL_OOB ? TAIL_OOB : ro[r] + c * 1024u;
    }
  }
  const __amdgpu_buffer_rsrc_t rs = TAIL_RSRC(mat, bytes);
  const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
  for (int i = 0; i < NL; ++i) b.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + so[i], 0, TAIL_AUX_NT);   // (the range check covers the VGPR offset only)
  b.ex = (unsigned)*(__attribute__((address_space(1))) const bf16_t*)exa;
}

// cross-batch state of a consumer wave
struct TailState {
  float d0, d1;            // DOWN accumulators of the current row pair
  unsigned pend;           // RB == 2: the even h value of a granule, waiting for the odd one
};

// The activation rows in LDS are zero-padded to whole 512-element chunks, so no product needs a mask.
template <int NC, int RB>
__device__ __forceinline__ void tail_consume(const DecTailParams& p, const TailWork& w, int phase, int j, const TailBuf<NC * RB>& b,
                                             TailState& st, int lane, int cw, const bf16_t* s_in, float* s_red, int slot) {
  constexpr int NL = NC * RB, NCD = NL / 2;
  if (phase == 2) {
    const int u = j / w.nkb, kb = j - u * w.nkb;
    const int cb = w.kc0 + kb * NCD;
    if (kb == 0) { st.d0 = 0.f; st.d1 = 0.f; }
#pragma unroll
    for (int c = 0; c < NCD; ++c) {
      const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(s_in + min(cb + c, w.nkc - 1) * 512 + lane * 8);   // weights of a chunk >= kc1 are zero
      dot8(b.v[c], xv, st.d0);
      dot8(b.v[NCD + c], xv, st.d1);
    }
    if (kb == w.nkb - 1) {
      const float s0 = wave_sum(st.d0), s1 = wave_sum(st.d1);
      if (lane == 0) {
        s_red[cw * 64 + 2 * u] = s0;
        s_red[cw * 64 + 2 * u + 1] = s1;
      }
    }
    return;
  }
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(s_in + c * 512 + lane * 8);
#pragma unroll
    for (int r = 0; r < RB; ++r) dot8(b.v[r * NC + c], xv, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = wave_sum(acc[r]);
  if (phase == 1) {
    // h[n] = bf16( silu(bf16 g) * bf16 u )   (decode_fused.hip:dec_gate_up_kernel)
    if constexpr (RB == 4) {
      const unsigned h0 = f2bf(apply_act(rbf(acc[0]), SVLM_ACT_SILU) * rbf(acc[RB / 2]));
      const unsigned h1 = f2bf(apply_act(rbf(acc[1]), SVLM_ACT_SILU) * rbf(acc[RB / 2 + 1]));
      if (lane == 0) publish(p.g_h, w.gu_u0 + j, h0 | (h1 << 16));
    } else {
      const unsigned h0 = f2bf(apply_act(rbf(acc[0]), SVLM_ACT_SILU) * rbf(acc[RB - 1]));
      if (j & 1) {
        if (lane == 0) publish(p.g_h, w.gu_u0 + (j >> 1), st.pend | (h0 << 16));
      } else {
        st.pend = h0;
      }
    }
    return;
  }
  // O and QKV: lane r finishes row r of the batch
  float mine = acc[0];
#pragma unroll
  for (int r = 1; r < RB; ++r) mine = lane == r ? acc[r] : mine;
  if (phase == 0) {
    // x'[n] = bf16( x[n] + bf16(o[n]) )      (gemv.hip residual epilogue)
    const int row0 = (w.o_b0 + j) * RB;
    const unsigned v = f2bf(rbf(mine) + bf2f((bf16_t)b.ex));
    const unsigned hi = (unsigned)__shfl_down((int)v, 1, 64);
    const int n = row0 + lane;
    if (lane < RB && (lane & 1) == 0 && n < p.H) publish(p.g_x1, n >> 1, v | (hi << 16));
  } else {
    // [q|k|v][n] = bf16(acc + b[n]); k and v go straight into the new token's pool slot   (decode_fused.hip:dec_qkv_kernel)
    const int N = p.qd + 2 * p.kd;
    const int n = (w.q_b0 + j) * RB + lane;
    if (lane < RB && n < N) {
      const bf16_t v = f2bf(mine + bf2f((bf16_t)b.ex));
      if (n < p.qd) {
        p.q_out[n] = v;
      } else {
        const int jj0 = n - p.qd;
        const int jj = jj0 < p.kd ? jj0 : jj0 - p.kd;
        bf16_t* plane = jj0 < p.kd ? p.k_planes : p.v_planes;
        plane[((size_t)(jj / p.D) * p.n_slots + slot) * p.D + jj % p.D] = v;
      }
    }
  }
}

// ---------------------------------------------------------------- gatherer
// All `n` granules of one vector (n even) into LDS, one u32 = two bf16 per granule.  Polling costs fabric bandwidth the weight
// stream needs (256 gatherers sweeping 8 KB each per microsecond are 2-3 TB/s of coherent reads), so the wait is done on 64
// SENTINEL granule pairs spread over the vector -- one 16-byte load per lane per poll -- and the full sweep (16 loads of two
// granules per lane per pass, every tag checked, missing ones re-read) only starts once all sentinels have arrived.
// Returns false once the spin bound is exhausted (what is there is copied, the caller flags the launch).
__device__ __forceinline__ bool tail_gather(__amdgpu_buffer_rsrc_t rs, int n, unsigned* dst, int lane, bool poll) {
  const int npairs = n >> 1;
  bool ok_all = true;
  unsigned spins = 0;
  if (poll) {
    const unsigned soff = (unsigned)min((int)(((long long)lane * npairs) >> 6), npairs - 1) * 16u;
    for (;;) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, soff, 0, TAIL_AUX_SC1);
      if (__all(v[1] == (unsigned)TAIL_TAG && v[3] == (unsigned)TAIL_TAG)) break;
      if (++spins > TAIL_SPIN_LIMIT) { poll = false; ok_all = false; break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  for (int base = 0; base < npairs; base += 1024) {
    u32x4_t v[16];
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int pi = base + k * 64 + lane;
        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)pi * 16u, 0, TAIL_AUX_SC1);      // out of range: zeros, no traffic
        ok &= pi >= npairs || (v[k][1] == (unsigned)TAIL_TAG && v[k][3] == (unsigned)TAIL_TAG);
      }
      if (__all(ok) || !poll) break;
      if (++spins > TAIL_SPIN_LIMIT) { poll = false; ok_all = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int pi = base + k * 64 + lane;
      if (pi < npairs) {
        u32x2_t pr = {v[k][0], v[k][2]};
        *reinterpret_cast<u32x2_t*>(dst + 2 * pi) = pr;
      }
    }
  }
  return ok_all;
}

// RMSNorm of the row in s_raw (bf16[H]) -> s_out = bf16( gain * bf16(x * rsqrt(mean x^2 + eps)) ), by ONE wave, with the partial sums
// of decode_fused.hip:stage_x (256 virtual threads = 4 per lane, one wave_sum per 64 of them, the four added in order).
template <int XC>
__device__ __forceinline__ void tail_norm(const bf16_t* s_raw, const u32x4_t (&gain)[XC][4], int H, float eps, bf16_t* s_out, int lane) {
  float red[4];
#pragma unroll
  for (int vt = 0; vt < 4; ++vt) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      const int c = (i * 256 + vt * 64 + lane) * 8;
      if (c < H) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(s_raw + c), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) ss += f[q] * f[q];
      }
    }
    red[vt] = wave_sum(ss);
  }
  const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
#pragma unroll
  for (int vt = 0; vt < 4; ++vt) {
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      const int c = (i * 256 + vt * 64 + lane) * 8;
      if (c < H) {
        float f[8], gq[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(s_raw + c), f);
        unpack8(gain[i][vt], gq);
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = gq[q] * rbf(f[q] * r);
        *reinterpret_cast<u32x4_t*>(s_out + c) = pack8(f);
      }
    }
  }
}

template <int XC>
__device__ __forceinline__ void tail_load_gain(const bf16_t* g, int H, u32x4_t (&gain)[XC][4], int lane) {
#pragma unroll
  for (int vt = 0; vt < 4; ++vt) {
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      const int c = (i * 256 + vt * 64 + lane) * 8;
      gain[i][vt] = *reinterpret_cast<const u32x4_t*>(g + (c < H ? c : 0));
    }
  }
}

__device__ __forceinline__ void tail_zero(bf16_t* s, int from, int to, int lane) {      // [from, to), both multiples of 8
  const u32x4_t z = {0u, 0u, 0u, 0u};
  for (int c = from + lane * 8; c < to; c += 512) *reinterpret_cast<u32x4_t*>(s + c) = z;
}

// wall-clock stamps exist in the STAMPS instantiation only (tools/dec_tail_bench.py): a store under a branch in the consumer's loop
// costs every later wait its exact count
#define TAIL_STAMP(role, i)                                                                                     \
  do {                                                                                                          \
    if constexpr (STAMPS) {                                                                                     \
      if (p.stamps && lane == 0) p.stamps[((size_t)blockIdx.x * 2 + (role)) * 16 + (i)] = wall_clock64();        \
    }                                                                                                           \
  } while (0)

// blocked share of `n` items for part `i` of `parts`
__device__ __forceinline__ int share_lo(int n, int i, int parts) { return (int)(((long long)n * i) / parts); }

template <int NC, int RB, bool STAMPS>
__global__ __launch_bounds__(256) void dec_tail_kernel(const DecTailParams p) {
  constexpr int NL = NC * RB, NCD = NL / 2;
  constexpr int XC = (NC * 512 + 2047) / 2048;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x;
  const int nkc = (p.I + 511) / 512;
  // ---- LDS carve-up (every offset a multiple of 16); activation rows are padded with zeros to whole chunks
  bf16_t* s_a = reinterpret_cast<bf16_t*>(tail_smem);                  // attn (phase O), later the normalised x'' (phase QKV)
  bf16_t* s_xs1 = s_a + NC * 512;                                      // n2(x')
  bf16_t* s_xr1 = s_xs1 + NC * 512;                                    // x' raw (residual of DOWN)
  bf16_t* s_h = s_xr1 + NC * 512;                                      // h
  float* s_red = reinterpret_cast<float*>(s_h + nkc * 512);            // [3][64] DOWN partial sums
  // ---- DOWN rows of this workgroup: row pairs [du0, du1)
  const int du0 = share_lo(p.H / 2, blockIdx.x, G), du1 = share_lo(p.H / 2, blockIdx.x + 1, G);

  if (wave == 3) {
    // ================================================================ gatherer
    bool poll = p.status[0] == 0, failed = false;                       // an earlier launch gave up: do not poll at all
    TAIL_STAMP(0, 0);
    const __amdgpu_buffer_rsrc_t rs_x1 = TAIL_RSRC(p.g_x1, (p.H / 2) * 8), rs_h = TAIL_RSRC(p.g_h, (p.I / 2) * 8),
                                 rs_x2 = TAIL_RSRC(p.g_x2, (p.H / 2) * 8);
    // G0: the attention row (a plain input), the zero pads, both norm gains
    for (int c = lane * 8; c < p.qd; c += 512) *reinterpret_cast<u32x4_t*>(s_a + c) = *reinterpret_cast<const u32x4_t*>(p.attn + c);
    tail_zero(s_a, p.qd, NC * 512, lane);
    tail_zero(s_xs1, p.H, NC * 512, lane);
    tail_zero(s_h, p.I, nkc * 512, lane);
    u32x4_t gain[XC][4];
    tail_load_gain<XC>(p.ln2, p.H, gain, lane);
    lds_barrier();                                                      // B0
    TAIL_STAMP(0, 1);
    // G1: x' -> raw + normalised
    if (!tail_gather(rs_x1, p.H / 2, reinterpret_cast<unsigned*>(s_xr1), lane, poll)) { poll = false; failed = true; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TAIL_STAMP(0, 2);
    tail_norm<XC>(s_xr1, gain, p.H, p.eps, s_xs1, lane);
    if (p.ln1n) tail_load_gain<XC>(p.ln1n, p.H, gain, lane);
    lds_barrier();                                                      // B1
    TAIL_STAMP(0, 3);
    // G2: h
    if (!tail_gather(rs_h, p.I / 2, reinterpret_cast<unsigned*>(s_h), lane, poll)) { poll = false; failed = true; }
    TAIL_STAMP(0, 4);
    lds_barrier();                                                      // B2
    TAIL_STAMP(0, 5);
    lds_barrier();                                                      // B3 (consumers: DOWN partial sums are in LDS)
    TAIL_STAMP(0, 6);
    // G3: x'' -> normalised with the next layer's input gain (raw copy parked in the h region, which DOWN has finished with)
    if (p.ln1n) {
      bf16_t* s_xr2 = s_h;
      if (!tail_gather(rs_x2, p.H / 2, reinterpret_cast<unsigned*>(s_xr2), lane, poll)) { poll = false; failed = true; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TAIL_STAMP(0, 7);
      tail_zero(s_a, p.H, NC * 512, lane);
      tail_norm<XC>(s_xr2, gain, p.H, p.eps, s_a, lane);
    }
    lds_barrier();                                                      // B4
    TAIL_STAMP(0, 8);
    if (failed && lane == 0) atomicOr(p.status, TAIL_ERR_TIMEOUT);
    return;
  }

  // ================================================================ consumers
  const int cw = wave;
  const int gw = blockIdx.x * 3 + cw, CW = G * 3;
  TailWork w;
  {
    const int nbO = (p.H + RB - 1) / RB;
    w.o_b0 = share_lo(nbO, gw, CW);
    w.c0 = share_lo(nbO, gw + 1, CW) - w.o_b0;
    const int nuG = p.I / 2;
    w.gu_u0 = share_lo(nuG, gw, CW);
    w.c1 = (share_lo(nuG, gw + 1, CW) - w.gu_u0) * (RB == 4 ? 1 : 2);
    w.nkc = nkc;
    w.kc0 = share_lo(nkc, cw, 3);
    w.kc1 = share_lo(nkc, cw + 1, 3);
    w.nkb = (w.kc1 - w.kc0 + NCD - 1) / NCD;
    w.d_u0 = du0;
    w.c2 = (du1 - du0) * w.nkb;
    const int nbQ = p.ln1n ? (p.qd + 2 * p.kd + RB - 1) / RB : 0;
    w.q_b0 = share_lo(nbQ, gw, CW);
    w.c3 = share_lo(nbQ, gw + 1, CW) - w.q_b0;
  }
  // the new token's index: requested first, used (to look the slot up) only behind the first three batches, so that neither of the
  // two dependent round trips delays the head of the weight stream
  int len = p.len_host;
  if (p.len_dev) len = *(__attribute__((address_space(1))) const int*)p.len_dev;
  TailState st;
  st.d0 = st.d1 = 0.f;
  st.pend = 0u;
  TailBuf<NL> b0, b1, b2;
  // issue cursor (ip, ij) and consume cursor (cp, cj) over the flattened (phase, batch) list
  int ip = 0, ij = 0, cp = 0, cj = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (ip < 4 && w.cnt(ip) == 0) ++ip;
    if (cp < 4 && w.cnt(cp) == 0) ++cp;
  }
  int nbar = 0;       // workgroup barriers passed: B0 B1 B2 | B3 B4
  // no inner loops anywhere in the consumer's main loop: hipcc flushes vmcnt in front of a loop that stores and does not load
#define TAIL_ADVANCE(ph, jj)                         \
  do {                                               \
    ++jj;                                            \
    if (jj >= w.cnt(ph)) {                           \
      jj = 0;                                        \
      ++ph;                                          \
      if (ph < 4 && w.cnt(ph) == 0) ++ph;            \
      if (ph < 4 && w.cnt(ph) == 0) ++ph;            \
      if (ph < 4 && w.cnt(ph) == 0) ++ph;            \
    }                                                \
  } while (0)
#define TAIL_ISSUE(buf)                                                              \
  do {                                                                               \
    tail_issue<NC, RB>(p, w, ip, ij, buf, lane);                                     \
    if (ip < 4) TAIL_ADVANCE(ip, ij);                                                \
  } while (0)
  // the x'' epilogue of DOWN, by consumer wave 0 right behind B3: x''[n] = bf16( x'[n] + bf16(sum of the three K slices) )
#define TAIL_ONE_BARRIER(need)                                                                               \
  if (nbar < (need)) {                                                                                       \
    if (cw == 0) TAIL_STAMP(1, 2 * nbar + 1);                                                                \
    lds_barrier();                                                                                           \
    ++nbar;                                                                                                  \
    if (cw == 0) TAIL_STAMP(1, 2 * nbar);                                                                    \
    if (nbar == 4 && cw == 0) {                                                                              \
      const int nrows = 2 * (du1 - du0);                                                                     \
      const int rr = lane < nrows ? lane : 0;                                                                \
      const int n = 2 * du0 + rr;                                                                            \
      const float s = s_red[rr] + s_red[64 + rr] + s_red[128 + rr];                                          \
      const unsigned v = f2bf(rbf(s) + bf2f(s_xr1[n]));                                                      \
      const unsigned hi = (unsigned)__shfl_down((int)v, 1, 64);                                              \
      if (lane < nrows) {                                                                                    \
        p.x[n] = (bf16_t)v;                                                                                  \
        if ((lane & 1) == 0) publish(p.g_x2, n >> 1, v | (hi << 16));                                        \
      }                                                                                                      \
    }                                                                                                        \
  }
#define TAIL_BARRIERS(need)                                                                                  \
  do {                                                                                                       \
    TAIL_ONE_BARRIER(need) TAIL_ONE_BARRIER(need) TAIL_ONE_BARRIER(need) TAIL_ONE_BARRIER(need) TAIL_ONE_BARRIER(need) \
  } while (0)
#define TAIL_STEP(buf)                                                                                       \
  {                                                                                                          \
    if (cp >= 4) break;                                                                                      \
    TAIL_BARRIERS(cp == 3 ? 5 : cp + 1);                                                                     \
    const bf16_t* s_in = cp == 0 ? s_a : cp == 1 ? s_xs1 : cp == 2 ? s_h : s_a;                              \
    tail_consume<NC, RB>(p, w, cp, cj, buf, st, lane, cw, s_in, s_red, slot);                                \
    asm volatile("" ::"v"(buf.ex)); /* keeps the operand's register allocated until here in every phase: freed while its load is    \
                                       pending, its next writer would have to wait for the whole batch */                          \
    TAIL_ADVANCE(cp, cj);                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    TAIL_ISSUE(buf);                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }
  if (cw == 0) TAIL_STAMP(1, 0);
  // DOWN partial sums of a wave without a K slice (short rows) must read as zero
  s_red[cw * 64 + lane] = 0.f;
  TAIL_ISSUE(b0);
  TAIL_ISSUE(b1);
  TAIL_ISSUE(b2);
  __builtin_amdgcn_sched_barrier(0);
  const int slot = p.slot_of[len];
  __builtin_amdgcn_sched_barrier(0);
  for (;;) {
    TAIL_STEP(b0)
    TAIL_STEP(b1)
    TAIL_STEP(b2)
  }
  TAIL_BARRIERS(5);
  if (cw == 0) TAIL_STAMP(1, 11);
#undef TAIL_STEP
#undef TAIL_BARRIERS
#undef TAIL_ONE_BARRIER
#undef TAIL_ISSUE
#undef TAIL_ADVANCE
}

// ================================================================ host side
extern "C" int svlm_device_cus();

static inline long long tail_layer_bytes(int H, int I) { return (((long long)(H / 2) * 2 + I / 2) * 8 + 255) / 256 * 256; }

// bytes of granule workspace for `n_layers` layers: [status block 256 B][layer 0][layer 1]...
extern "C" long long svlm_dec_tail_ws_bytes(int H, int I, int n_layers) {
  if (H <= 0 || I <= 0 || n_layers <= 0 || H % 2 || I % 2) return SVLM_EINVAL;
  return 256 + tail_layer_bytes(H, I) * n_layers;
}

// Zero every granule of the step (NOT the status block: a give-up stays visible until the host clears it).
extern "C" int svlm_dec_tail_reset(void* ws, int H, int I, int n_layers, void* stream) {
  SVLM_CHECK_ARG(ws != nullptr && svlm_dec_tail_ws_bytes(H, I, n_layers) > 0, "svlm_dec_tail_reset: bad arguments");
  if (hipMemsetAsync((char*)ws + 256, 0, (size_t)(tail_layer_bytes(H, I) * n_layers), (hipStream_t)stream) != hipSuccess) {
    svlm_set_error("svlm_dec_tail_reset: hipMemsetAsync failed");
    return SVLM_ELAUNCH;
  }
  return SVLM_OK;
}

extern "C" int svlm_dec_tail(const void* attn, void* x, const void* o_w, int ld_o, const void* ln2, const void* gu_w, int ld_gu,
                             const void* down_w, int ld_down, const void* ln1_next, const void* qkv_w_next, int ld_qkv,
                             const void* qkv_b_next, void* q_out, void* k_planes_next, void* v_planes_next, const int* slot_of,
                             const int* len_dev, int len_host, int H, int I, int qd, int kd, int D, int n_slots, float eps, void* ws,
                             int layer, int n_layers, int grid, void* stamps, void* stream) {
  SVLM_CHECK_ARG(H > 0 && I > 0 && H % 8 == 0 && I % 8 == 0 && qd > 0 && qd % 8 == 0, "svlm_dec_tail: bad H=%d I=%d qd=%d", H, I, qd);
  SVLM_CHECK_ARG(ld_o >= qd && ld_gu >= H && ld_down >= I && ld_o % 8 == 0 && ld_gu % 8 == 0 && ld_down % 8 == 0,
                 "svlm_dec_tail: bad leading dimensions %d %d %d", ld_o, ld_gu, ld_down);
  SVLM_CHECK_ARG(attn && x && o_w && ln2 && gu_w && down_w && ws, "svlm_dec_tail: null operand");
  SVLM_CHECK_ARG(layer >= 0 && layer < n_layers, "svlm_dec_tail: layer %d outside [0, %d)", layer, n_layers);
  const bool qkv = ln1_next != nullptr;
  if (qkv) {
    SVLM_CHECK_ARG(qkv_w_next && qkv_b_next && q_out && k_planes_next && v_planes_next && slot_of, "svlm_dec_tail: null QKV operand");
    SVLM_CHECK_ARG(kd > 0 && D > 0 && kd % D == 0 && n_slots > 0 && ld_qkv >= H && ld_qkv % 8 == 0, "svlm_dec_tail: bad kd=%d D=%d ld_qkv=%d", kd, D, ld_qkv);
  }
  // every workgroup waits for every other one's outputs: all of them must be resident together, and one per CU always can be
  static int n_cus = 0;
  if (n_cus == 0) n_cus = svlm_device_cus();
  SVLM_CHECK_ARG(n_cus > 0, "svlm_dec_tail: no HIP device");
  if (grid <= 0) grid = n_cus;
  SVLM_CHECK_ARG(grid <= n_cus, "svlm_dec_tail: grid %d exceeds the %d CUs of the device (workgroups could not all be resident)", grid, n_cus);
  // a workgroup's DOWN rows are summed through a [3][64] LDS block
  SVLM_CHECK_ARG(2 * ((H / 2 + grid - 1) / grid) <= 64, "svlm_dec_tail: %d rows per workgroup at grid %d", 2 * ((H / 2 + grid - 1) / grid), grid);
  const int inw = (qd > H ? qd : H);
  const int nc = (inw + 511) / 512, nkc = (I + 511) / 512;
  SVLM_CHECK_ARG(H <= nkc * 512, "svlm_dec_tail: H=%d beyond the padded intermediate row %d", H, nkc * 512);
  SVLM_CHECK_ARG((long long)2 * I * ld_gu * 2 < (1ll << 31) && (long long)H * ld_down * 2 < (1ll << 31) && (long long)H * ld_o * 2 < (1ll << 31) &&
                     (!qkv || (long long)(qd + 2 * kd) * ld_qkv * 2 < (1ll << 31)), "svlm_dec_tail: weight matrix beyond 2 GiB");
  const size_t lds = (size_t)(3 * nc * 512 + nkc * 512) * 2 + 3 * 64 * 4;
  SVLM_CHECK_ARG(lds <= 64 * 1024, "svlm_dec_tail: %zu bytes of LDS", lds);
  DecTailParams p;
  p.attn = (const bf16_t*)attn; p.x = (bf16_t*)x; p.ln2 = (const bf16_t*)ln2;
  p.ln1n = (const bf16_t*)ln1_next; p.qkv_b = (const bf16_t*)qkv_b_next;
  p.mat[0] = (const bf16_t*)o_w; p.mbytes[0] = (unsigned)H * ld_o * 2u; p.ld2[0] = ld_o * 2u;
  p.mat[1] = (const bf16_t*)gu_w; p.mbytes[1] = 2u * I * ld_gu * 2u; p.ld2[1] = ld_gu * 2u;
  p.mat[2] = (const bf16_t*)down_w; p.mbytes[2] = (unsigned)H * ld_down * 2u; p.ld2[2] = ld_down * 2u;
  p.mat[3] = qkv ? (const bf16_t*)qkv_w_next : (const bf16_t*)o_w; p.mbytes[3] = qkv ? (unsigned)(qd + 2 * kd) * ld_qkv * 2u : 0u; p.ld2[3] = qkv ? ld_qkv * 2u : 0u;
  p.mat[4] = (const bf16_t*)o_w; p.mbytes[4] = 0u; p.ld2[4] = 0u;
  p.q_out = (bf16_t*)q_out; p.k_planes = (bf16_t*)k_planes_next; p.v_planes = (bf16_t*)v_planes_next;
  p.slot_of = slot_of; p.len_dev = len_dev; p.len_host = len_host;
  char* base = (char*)ws + 256 + tail_layer_bytes(H, I) * layer;
  p.g_x1 = (u64*)base;
  p.g_h = p.g_x1 + H / 2;
  p.g_x2 = p.g_h + I / 2;
  p.status = (int*)ws;
  p.stamps = (u64*)stamps;
  p.H = H; p.I = I; p.qd = qd; p.kd = kd; p.D = D; p.n_slots = n_slots; p.eps = eps;
  if (!qkv) { p.slot_of = (const int*)ws; p.len_dev = nullptr; p.len_host = 0; }      // any readable int (status[0]); never used
  hipStream_t s = (hipStream_t)stream;
  if (stamps) {
    SVLM_CHECK_ARG(nc == 3, "svlm_dec_tail: the stamped build exists for hidden sizes of 1025..1536 only");
    dec_tail_kernel<3, 4, true><<<grid, 256, lds, s>>>(p);
  } else if (nc <= 1) dec_tail_kernel<1, 4, false><<<grid, 256, lds, s>>>(p);
  else if (nc <= 2) dec_tail_kernel<2, 4, false><<<grid, 256, lds, s>>>(p);
  else if (nc <= 3) dec_tail_kernel<3, 4, false><<<grid, 256, lds, s>>>(p);
  else if (nc <= 4) dec_tail_kernel<4, 4, false><<<grid, 256, lds, s>>>(p);
  else if (nc <= 7) dec_tail_kernel<7, 2, false><<<grid, 256, lds, s>>>(p);
  else {
    svlm_set_error("svlm_dec_tail: hidden size %d beyond the built variants (<= 3584)", inw);
    return SVLM_EINVAL;
  }
  return svlm_check_launch("svlm_dec_tail");
}
