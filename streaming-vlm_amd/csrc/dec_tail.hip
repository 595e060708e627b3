// Persistent decode-layer tail (T = 1): everything of one decoder layer that follows the attention, and the QKV projection of the
// NEXT layer, as ONE launch of one 16-wave workgroup per CU:
//
//   x'  = x + W_o . attn                                   (qwen2/language_forward.py:161,196)       phase O
//   h   = silu(W_g . n2(x')) * (W_u . n2(x'))              (Qwen2MLP, :198-201)                       phase GU
//   x'' = x' + W_d . h                                     (:201-202)                                 phase DOWN
//   [q|k|v] = W_qkv' . n1'(x'') + b'  -> q buffer, k / v straight into the new token's pool slot      phase QKV
//                                                          (next layer, :183,80-82 + streaming_cache.py:72-73)
//
// replacing the four launches  o_proj GEMV / gate-up GEMV / down_proj GEMV / next layer's QKV GEMV  of the per-op decode step
// (decode_fused.hip, gemv.hip).  Between those ops every CU needs the WHOLE output vector of the op before (all-to-all seams), which
// is what a kernel boundary gives for free -- and what costs each launch its ~1.5 us boundary plus the ~2 us until its first weight
// bytes arrive.  The per-op kernels already stream at the HBM rate once they run; what a single launch can win is those fixed costs.
//
// The whole layer in flight at once.  A Qwen2-VL-2B layer's tail weighs 93.6 MB = 366 KB per CU, and a CU's register file holds
// 512 KB: with 16 waves per CU, every consumer wave requests ALL its gate/up and down_proj rows (26 loads of 16 B per lane = 104
// VGPRs) in its first microsecond, straight into registers, non-temporal, through range-checked buffer descriptors.  HBM then
// streams the layer without a pause from the first microsecond to the last byte, whatever the seams do meanwhile; the phases
// consume their registers as the data and the hand-offs arrive (vmcnt retires in order: gate/up first, then down_proj).  There is
// no ring and no load issued from inside a loop.  (Round 3's first form of this kernel ran 3 consumer waves per CU over a ring of
// three register batches -- 30 MB in flight per chip: HBM idled at every seam, 41 us per layer against 24 for the per-op launches;
// its in-kernel timeline, profiles/r03_dec_tail_vs_per_op.json, is what led here.)
//
//   * wave roles: waves 0-14 of a workgroup are CONSUMERS, wave 15 is the GATHERER.  The gatherer computes the workgroup's six
//     o_proj rows itself (18 loads, consumed before it ever polls), then sweeps each seam's vector, normalises it where the next op
//     wants RMSNorm, and leaves it in LDS behind a workgroup barrier.  Only the gatherer polls: a poll issued behind a consumer's
//     weight loads would wait for all of them.
//   * hand-offs are 8-byte {tag, 2 x bf16} granules (Guideline 16, form R2: the data IS the flag): a producer publishes each pair of
//     outputs with ONE relaxed agent-scope 8-byte store; every gatherer waits on 64 sentinel pairs spread over the vector (one
//     16-byte load per lane per poll: 256 gatherers sweeping whole vectors would cost the weight stream terabytes per second), then
//     sweeps it with 16-byte sc1 buffer loads until every tag is set.  No fence, no flag, no ordering assumption.  Every (layer, seam)
//     has its own granule block, zeroed once per decode step (svlm_dec_tail_reset, a memset node at the head of the step's graph).
//   * every spin is bounded: a gatherer that gives up raises status[0] (sticky, checked by the host at the end of the chunk), stops
//     polling for the rest of the launch and lets its workgroup run to the end on whatever it has, so the grid always drains; later
//     launches see the status word and do not poll at all.
//   * results do not depend on placement: any workgroup may land on any CU, rows are assigned by blockIdx only.
//
// Shapes: the layer must fit -- per consumer wave at most SG gate/up row pairs and SD down_proj (row, 512-column) pieces, for the
// instantiated (NC, SG, SD): Qwen2-VL-2B (1536 / 8960) at one workgroup per CU, and small test widths.  Larger models (the 7B's
// 466 MB per layer) do not, and their per-op kernels already run at 80 % of the HBM peak: svlm_dec_tail refuses them.
//
// Arithmetic: the per-op kernels' rounding points and RMSNorm partial sums; products summed pairwise with v_dot2c_f32_bf16 (fp32
// accumulation), so results agree with svlm_gemv_bf16 / svlm_dec_gate_up / svlm_dec_qkv to fp32 summation order.
#include "common.h"

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

#define TAIL_TAG 1ull
#define TAIL_SPIN_LIMIT (1u << 17)          // polls of one wait before a gatherer gives up (~0.1-0.3 s)
#define TAIL_ERR_TIMEOUT 1
#define TAIL_CW 15                          // consumer waves per workgroup (wave 15 is the gatherer)
#define TAIL_OR 8                           // most o_proj / down_proj rows of one workgroup

extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];

struct DecTailParams {
  const bf16_t* attn;      // [qd]   attention output of this layer (written by the launch before)
  bf16_t* x;               // [H]    residual stream: read as x, rewritten as x''
  const bf16_t* o_w;       // [H][ld_o]
  const bf16_t* ln2;       // [H]
  const bf16_t* gu_w;      // [2I][ld_gu]   gate rows, then up rows
  const bf16_t* down_w;    // [H][ld_down]
  const bf16_t* ln1n;      // [H]    next layer's input norm, or null: no QKV phase (last layer)
  const bf16_t* qkv_w;     // [qd + 2 kd][ld_qkv]
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
  int ld_o, ld_gu, ld_down, ld_qkv;
  int H, I, qd, kd, D, n_slots, len_host;
  float eps;
};

typedef __bf16 hw_bf2 __attribute__((ext_vector_type(2)));

// Weights and granules are read through buffer descriptors: a 32-bit offset per lane, no 64-bit address arithmetic on the VALU, and
// the range check (it covers the VGPR offset, not the scalar one) makes every slot a wave does not own -- rows behind the end of
// its share -- a load that returns zero WITHOUT touching memory.
#define TAIL_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((const void*)(ptr)), 0, (int)(bytes), 0x00020000)
#define TAIL_AUX_NT 2                       // streamed once: non-temporal
#define TAIL_AUX_SC1 16                     // agent-coherent: hand-off granules
#define TAIL_OOB 0x7ffff000u                // an offset behind every descriptor's range (matrices are < 2 GiB: host check)

__device__ __forceinline__ void publish(u64* g, int idx, unsigned payload) {
  __hip_atomic_store((gu64*)(g + idx), (TAIL_TAG << 32) | (u64)payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc += w . x over the 8 elements of one 16-byte piece: four v_dot2c_f32_bf16 on the packed pairs
__device__ __forceinline__ void dot8(const u32x4_t& w, const u32x4_t& x, float& acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned wi = w[i], xi = x[i];     // scalars first: __builtin_bit_cast of a vector-element lvalue reads element 0 (hipcc 7.2)
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_bf2, wi), __builtin_bit_cast(hw_bf2, xi), acc, false);
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

// wall-clock stamps exist in the STAMPS instantiation only (tools/dec_tail_bench.py)
#define TAIL_STAMP(role, i)                                                                                     \
  do {                                                                                                          \
    if constexpr (STAMPS) {                                                                                     \
      if (p.stamps && lane == 0) p.stamps[((size_t)blockIdx.x * 2 + (role)) * 16 + (i)] = wall_clock64();        \
    }                                                                                                           \
  } while (0)

// blocked share of `n` items for part `i` of `parts`
__device__ __forceinline__ int share_lo(int n, int i, int parts) { return (int)(((long long)n * i) / parts); }

template <int NC, int SG, int SD, int OR, bool STAMPS>
__global__ __launch_bounds__(1024) void dec_tail_kernel(const DecTailParams p) {
  constexpr int XC = (NC * 512 + 2047) / 2048;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x, b = blockIdx.x;
  const int nkc = (p.I + 511) / 512;
  // ---- LDS carve-up (every offset a multiple of 16); activation rows are padded with zeros to whole 512-element chunks
  bf16_t* s_xs1 = reinterpret_cast<bf16_t*>(tail_smem);                // n2(x')
  bf16_t* s_xr1 = s_xs1 + NC * 512;                                    // x' raw (residual of DOWN)
  bf16_t* s_xs2 = s_xr1 + NC * 512;                                    // n1'(x'')
  bf16_t* s_h = s_xs2 + NC * 512;                                      // h (later: x'' raw)
  float* s_red = reinterpret_cast<float*>(s_h + nkc * 512);            // [TAIL_OR][16] DOWN partial sums, one column per consumer
  bf16_t* s_hown = reinterpret_cast<bf16_t*>(s_red + TAIL_OR * 16);    // [64] this workgroup's own h values
  // ---- this workgroup's share: o_proj / down_proj rows [r0, r1) (whole pairs), h values [h0, h1) (whole pairs), QKV rows [q0, q1)
  const int r0 = 2 * share_lo(p.H / 2, b, G), r1 = 2 * share_lo(p.H / 2, b + 1, G), nrow = r1 - r0;
  const int h0 = 2 * share_lo(p.I / 2, b, G), h1 = 2 * share_lo(p.I / 2, b + 1, G);
  const int nq = p.ln1n ? p.qd + 2 * p.kd : 0;
  const int q0 = share_lo(nq, b, G), q1 = share_lo(nq, b + 1, G);
  const unsigned voff = (unsigned)lane * 16u;
  float* s_opart = reinterpret_cast<float*>(s_hown + 64);              // [OR * NC] o_proj partial sums, one per (row, chunk) piece

  // ---- phase O, first half, by ALL 16 waves: piece f = row * NC + chunk of the workgroup's o_proj rows goes to wave f (pieces 16 ..
  // to the gatherer, which has registers to spare): ONE 16-byte load per lane, the FIRST request this wave makes.  The memory system
  // serves requests in arrival order: asked for by one wave (18 loads deep, beside 3840 waves asking for 26 each) the o_proj rows
  // landed with the average byte of the layer -- x' left 12.8 us into the launch; spread over every wave's first slot they land in
  // the first microsecond.
  const __amdgpu_buffer_rsrc_t rs_o = TAIL_RSRC(p.o_w, (unsigned)p.H * p.ld_o * 2u);
  auto o_off = [&](int f) -> unsigned {
    const int row = f / NC, ch = f - row * NC;
    return f < OR * NC && row < nrow ? (unsigned)(r0 + row) * (unsigned)p.ld_o * 2u + (unsigned)ch * 1024u : TAIL_OOB;
  };
  auto o_attn = [&](int f) -> u32x4_t {
    const int k = (f % NC) * 512 + lane * 8;
    return f < OR * NC && k < p.qd ? *reinterpret_cast<const u32x4_t*>(p.attn + k) : u32x4_t{0u, 0u, 0u, 0u};
  };
  const u32x4_t ow0 = __builtin_amdgcn_raw_buffer_load_b128(rs_o, voff + o_off(wave), 0, TAIL_AUX_NT);
  const u32x4_t oa0 = o_attn(wave);
  // ... and a barrier (no wait: the loads stay in flight) before anybody requests more: a CU hands its requests to the memory system
  // in issue order at ~1 KB per 16 clocks, so one wave's 26 weight loads in front of another wave's o_proj piece delay it by microseconds
  asm volatile("s_barrier" ::: "memory");
  if (wave == TAIL_CW) {
    // ================================================================ gatherer
    bool poll = p.status[0] == 0, failed = false;                       // an earlier launch gave up: do not poll at all
    TAIL_STAMP(0, 0);
    const __amdgpu_buffer_rsrc_t rs_x1 = TAIL_RSRC(p.g_x1, (p.H / 2) * 8), rs_h = TAIL_RSRC(p.g_h, (p.I / 2) * 8),
                                 rs_x2 = TAIL_RSRC(p.g_x2, (p.H / 2) * 8);
    // ---- phase O: every wave of the workgroup (this one included) requested one or two (row, 512-column) pieces of the workgroup's
    // o_proj rows as its FIRST loads (tail_o_piece above): the memory system serves requests in arrival order, so these land in the
    // first microsecond of the launch instead of with the average byte of the layer.  Here: the sum of each row's pieces, the residual
    u32x4_t gain[XC][4];                                                // requested now: behind the consumers' bursts they would arrive with the last weight
    tail_load_gain<XC>(p.ln2, p.H, gain, lane);
    constexpr int NX = OR * NC > 16 ? OR * NC - 16 : 0;                 // pieces behind the sixteenth
    u32x4_t owx[NX > 0 ? NX : 1], oax[NX > 0 ? NX : 1];
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      owx[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, voff + o_off(16 + t), 0, TAIL_AUX_NT);
      oax[t] = o_attn(16 + t);
    }
    const unsigned short xres = p.x[r0 + (lane < nrow ? lane : 0)];        // lane r finishes row r
    tail_zero(s_xs1, p.H, NC * 512, lane);
    tail_zero(s_xs2, p.H, NC * 512, lane);
    tail_zero(s_h, p.I, nkc * 512, lane);
    {
      float a = 0.f;
      dot8(ow0, oa0, a);
      a = wave_sum(a);
      if (lane == 0 && wave < OR * NC) s_opart[wave] = a;
#pragma unroll
      for (int t = 0; t < NX; ++t) {
        float ax = 0.f;
        dot8(owx[t], oax[t], ax);
        ax = wave_sum(ax);
        if (lane == 0) s_opart[16 + t] = ax;
      }
    }
    lds_barrier();                                                      // B0: every piece's partial sum is in LDS
    {
      const int rr = lane < nrow ? lane : 0;
      float o = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) o += s_opart[rr * NC + c];
      // x'[n] = bf16( x[n] + bf16(o[n]) )      (gemv.hip residual epilogue)
      const unsigned v = f2bf(rbf(o) + bf2f((bf16_t)xres));
      const unsigned hi = (unsigned)__shfl_down((int)v, 1, 64);
      if (lane < nrow && (lane & 1) == 0) publish(p.g_x1, (r0 + lane) >> 1, v | (hi << 16));
    }
    TAIL_STAMP(0, 1);
    // ---- seam 1: x' -> raw + normalised
    if (!tail_gather(rs_x1, p.H / 2, reinterpret_cast<unsigned*>(s_xr1), lane, poll)) { poll = false; failed = true; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TAIL_STAMP(0, 2);
    tail_norm<XC>(s_xr1, gain, p.H, p.eps, s_xs1, lane);
    if (p.ln1n) tail_load_gain<XC>(p.ln1n, p.H, gain, lane);
    lds_barrier();                                                      // B1: n2(x') is in LDS
    TAIL_STAMP(0, 3);
    lds_barrier();                                                      // Bh: the workgroup's own h values are in LDS (consumer 0 publishes)
    // ---- seam 2: h
    if (!tail_gather(rs_h, p.I / 2, reinterpret_cast<unsigned*>(s_h), lane, poll)) { poll = false; failed = true; }
    TAIL_STAMP(0, 4);
    lds_barrier();                                                      // B2: h is in LDS
    TAIL_STAMP(0, 5);
    lds_barrier();                                                      // B3: DOWN partial sums are in LDS (consumer 0 publishes x'')
    TAIL_STAMP(0, 6);
    // ---- seam 3: x'' -> normalised with the next layer's input gain (raw copy parked in the h region, which DOWN has finished with)
    if (p.ln1n) {
      bf16_t* s_xr2 = s_h;
      if (!tail_gather(rs_x2, p.H / 2, reinterpret_cast<unsigned*>(s_xr2), lane, poll)) { poll = false; failed = true; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TAIL_STAMP(0, 7);
      tail_norm<XC>(s_xr2, gain, p.H, p.eps, s_xs2, lane);
    }
    lds_barrier();                                                      // B4: n1'(x'') is in LDS
    TAIL_STAMP(0, 8);
    if (failed && lane == 0) atomicOr(p.status, TAIL_ERR_TIMEOUT);
    return;
  }

  // ================================================================ consumers
  const int cw = wave;
  if (cw == 0) TAIL_STAMP(1, 0);
  // the new token's index first: it arrives ahead of the weights (vmcnt retires in order), so the slot lookup behind them never
  // waits for more than this one load
  int len = p.len_host;
  if (p.len_dev) len = *p.len_dev;
  const __amdgpu_buffer_rsrc_t rs_gu = TAIL_RSRC(p.gu_w, 2u * (unsigned)p.I * p.ld_gu * 2u), rs_d = TAIL_RSRC(p.down_w, (unsigned)p.H * p.ld_down * 2u);
  // ---- everything this wave will ever multiply, requested now: SG gate/up row pairs, then SD down_proj pieces
  const int hb = h0 + cw * SG;                        // first h value of this wave
  u32x4_t wg[SG][NC], wu[SG][NC], wd[SD];
  const int fb = cw * SD;                             // first (row, chunk) piece of this wave: piece f = row * nkc + chunk
  const int rowA = fb / nkc;                          // its pieces lie in local rows rowA and rowA + 1 (SD < nkc, or nkc == 1 == SD)
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const bool own = hb + s < h1;
    const unsigned og = own ? (unsigned)(hb + s) * (unsigned)p.ld_gu * 2u : TAIL_OOB;
    const unsigned ou = own ? (unsigned)(p.I + hb + s) * (unsigned)p.ld_gu * 2u : TAIL_OOB;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      wg[s][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_gu, voff + og + c * 1024u, 0, TAIL_AUX_NT);
      wu[s][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_gu, voff + ou + c * 1024u, 0, TAIL_AUX_NT);
    }
  }
  const int slot = p.slot_of ? p.slot_of[len] : 0;
  // DOWN partial sums of the rows this wave has no piece of must read as zero
  if (lane < TAIL_OR) s_red[lane * 16 + cw] = 0.f;
  {    // this wave's o_proj piece: its load was the first one requested, everything above is still in flight
    float a = 0.f;
    dot8(ow0, oa0, a);
    a = wave_sum(a);
    if (lane == 0 && cw < OR * NC) s_opart[cw] = a;
  }
  if (cw == 0) TAIL_STAMP(1, 1);
  lds_barrier();                                                        // B0: the o_proj pieces' partial sums are in LDS
  lds_barrier();                                                        // B1
  if (cw == 0) TAIL_STAMP(1, 2);
  // ---- the down_proj pieces are requested HERE, not with gate/up: a CU's memory pipeline is a queue -- a poll issued behind N bytes of
  // weight requests comes back after N / 25 GB/s -- so each seam's polls should queue behind exactly the weights the NEXT phase needs
  // (they arrive together), never behind more.  With everything requested at launch the x' seam waited for all 366 KB: B1 at 16 us.
#pragma unroll
  for (int s = 0; s < SD; ++s) {
    const int f = fb + s, row = f / nkc, ch = f - row * nkc;
    const unsigned od = row < nrow ? (unsigned)(r0 + row) * (unsigned)p.ld_down * 2u + (unsigned)ch * 1024u : TAIL_OOB;
    wd[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, voff + od, 0, TAIL_AUX_NT);
  }

  // ---- phase GU: h[n] = bf16( silu(bf16 g) * bf16 u )   (decode_fused.hip:dec_gate_up_kernel)
  {
    float ag[SG], au[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) ag[s] = au[s] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(s_xs1 + c * 512 + lane * 8);
#pragma unroll
      for (int s = 0; s < SG; ++s) { dot8(wg[s][c], xv, ag[s]); dot8(wu[s][c], xv, au[s]); }
    }
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const float g = wave_sum(ag[s]), u = wave_sum(au[s]);
      const bf16_t hv = f2bf(apply_act(rbf(g), SVLM_ACT_SILU) * rbf(u));
      if (lane == 0 && hb + s < h1) s_hown[hb + s - h0] = hv;
    }
  }
  // ---- the next layer's QKV row of this wave: requested into the registers gate/up has just released
  const __amdgpu_buffer_rsrc_t rs_q = TAIL_RSRC(p.qkv_w ? p.qkv_w : p.o_w, nq ? (unsigned)nq * p.ld_qkv * 2u : 0u);
  const int qrow = q0 + cw;
  const bool qown = qrow < q1;
  u32x4_t wq[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
    wq[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_q, voff + (qown ? (unsigned)qrow * (unsigned)p.ld_qkv * 2u + c * 1024u : TAIL_OOB), 0, TAIL_AUX_NT);
  unsigned short qbias = 0;
  if (qown) qbias = p.qkv_b[qrow];
  if (cw == 0) TAIL_STAMP(1, 3);
  lds_barrier();                                                        // Bh
  if (cw == 0) {
    const int j = lane < (h1 - h0) / 2 ? lane : 0;
    const unsigned pay = (unsigned)s_hown[2 * j] | ((unsigned)s_hown[2 * j + 1] << 16);
    if (lane < (h1 - h0) / 2) publish(p.g_h, h0 / 2 + lane, pay);
    TAIL_STAMP(1, 4);
  }
  lds_barrier();                                                        // B2
  if (cw == 0) TAIL_STAMP(1, 5);
  // ---- phase DOWN: this wave's pieces of (at most) two rows
  {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int s = 0; s < SD; ++s) {
      const int f = fb + s, row = f / nkc, ch = f - row * nkc;
      const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(s_h + ch * 512 + lane * 8);
      float d = 0.f;
      dot8(wd[s], xv, d);                              // pieces this wave does not own are zeros (out-of-range loads)
      if (row == rowA) a0 += d; else a1 += d;
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) {
      if (rowA < TAIL_OR) s_red[rowA * 16 + cw] = a0;
      if (rowA + 1 < TAIL_OR) s_red[(rowA + 1) * 16 + cw] = a1;
    }
  }
  if (cw == 0) TAIL_STAMP(1, 6);
  lds_barrier();                                                        // B3
  if (cw == 0) {
    // x''[n] = bf16( x'[n] + bf16(sum of the waves' pieces, in wave order) )
    const int rr = lane < nrow ? lane : 0;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < TAIL_CW; ++w) sum += s_red[rr * 16 + w];
    const unsigned v = f2bf(rbf(sum) + bf2f(s_xr1[r0 + rr]));
    const unsigned hi = (unsigned)__shfl_down((int)v, 1, 64);
    if (lane < nrow) {
      p.x[r0 + lane] = (bf16_t)v;
      if ((lane & 1) == 0) publish(p.g_x2, (r0 + lane) >> 1, v | (hi << 16));
    }
    TAIL_STAMP(1, 7);
  }
  lds_barrier();                                                        // B4
  if (cw == 0) TAIL_STAMP(1, 8);
  // ---- phase QKV: [q|k|v][n] = bf16(acc + b[n]); k and v go straight into the new token's pool slot   (decode_fused.hip:dec_qkv_kernel)
  {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) dot8(wq[c], *reinterpret_cast<const u32x4_t*>(s_xs2 + c * 512 + lane * 8), acc);
    acc = wave_sum(acc);
    if (lane == 0 && qown) {
      const bf16_t v = f2bf(acc + bf2f((bf16_t)qbias));
      if (qrow < p.qd) {
        p.q_out[qrow] = v;
      } else {
        const int jj0 = qrow - p.qd;
        const int jj = jj0 < p.kd ? jj0 : jj0 - p.kd;
        bf16_t* plane = jj0 < p.kd ? p.k_planes : p.v_planes;
        plane[((size_t)(jj / p.D) * p.n_slots + slot) * p.D + jj % p.D] = v;
      }
    }
  }
  if (cw == 0) TAIL_STAMP(1, 9);
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

// 1 when svlm_dec_tail has a build for this layer geometry at `grid` workgroups (0 = one per CU), else 0 (the layer's weights do not
// fit the register file: use the per-op entry points); negative on bad arguments.  Host arithmetic only.
extern "C" int svlm_dec_tail_supported(int H, int I, int qd, int kd, int grid) {
  if (H <= 0 || I <= 0 || qd <= 0 || kd < 0 || H % 8 || I % 8 || qd % 8) return SVLM_EINVAL;
  if (grid <= 0) grid = 256;
  const int inw = qd > H ? qd : H, nc = (inw + 511) / 512, nkc = (I + 511) / 512;
  const int nrow = 2 * ((H / 2 + grid - 1) / grid), nh = 2 * ((I / 2 + grid - 1) / grid), nqr = (qd + 2 * kd + grid - 1) / grid;
  const int sg = (nh + TAIL_CW - 1) / TAIL_CW, sd = (nrow * nkc + TAIL_CW - 1) / TAIL_CW;
  if (nrow > (nc == 3 ? 6 : 4) || nh > 64 || nqr > TAIL_CW || H > nkc * 512) return 0;
  if (sd >= nkc && !(nkc == 1 && sd == 1)) return 0;                    // a wave's pieces must lie in at most two rows
  const bool fits = (nc == 3 && sg <= 3 && sd <= 8) || (nc <= 2 && sg <= 1 && sd <= 1);
  const size_t lds = (size_t)(3 * nc * 512 + nkc * 512) * 2 + TAIL_OR * 16 * 4 + 128 + TAIL_OR * 8 * 4;
  return fits && lds <= 64 * 1024 ? 1 : 0;
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
  SVLM_CHECK_ARG(svlm_dec_tail_supported(H, I, qd, qkv ? kd : 0, grid) == 1,
                 "svlm_dec_tail: no build for H=%d I=%d qd=%d at %d workgroups (the layer's weights must fit the register file: Qwen2-VL-2B class); "
                 "use the per-op entry points", H, I, qd, grid);
  SVLM_CHECK_ARG((long long)2 * I * ld_gu * 2 < (1ll << 31) && (long long)H * ld_down * 2 < (1ll << 31) && (long long)H * ld_o * 2 < (1ll << 31) &&
                     (!qkv || (long long)(qd + 2 * kd) * ld_qkv * 2 < (1ll << 31)), "svlm_dec_tail: weight matrix beyond 2 GiB");
  const int inw = (qd > H ? qd : H);
  const int nc = (inw + 511) / 512, nkc = (I + 511) / 512;
  const size_t lds = (size_t)(3 * nc * 512 + nkc * 512) * 2 + TAIL_OR * 16 * 4 + 128 + TAIL_OR * 8 * 4;
  DecTailParams p;
  p.attn = (const bf16_t*)attn; p.x = (bf16_t*)x; p.o_w = (const bf16_t*)o_w; p.ln2 = (const bf16_t*)ln2;
  p.gu_w = (const bf16_t*)gu_w; p.down_w = (const bf16_t*)down_w;
  p.ln1n = (const bf16_t*)ln1_next; p.qkv_w = (const bf16_t*)qkv_w_next; p.qkv_b = (const bf16_t*)qkv_b_next;
  p.q_out = (bf16_t*)q_out; p.k_planes = (bf16_t*)k_planes_next; p.v_planes = (bf16_t*)v_planes_next;
  p.slot_of = slot_of; p.len_dev = len_dev; p.len_host = len_host;
  char* base = (char*)ws + 256 + tail_layer_bytes(H, I) * layer;
  p.g_x1 = (u64*)base;
  p.g_h = p.g_x1 + H / 2;
  p.g_x2 = p.g_h + I / 2;
  p.status = (int*)ws;
  p.stamps = (u64*)stamps;
  p.ld_o = ld_o; p.ld_gu = ld_gu; p.ld_down = ld_down; p.ld_qkv = ld_qkv;
  p.H = H; p.I = I; p.qd = qd; p.kd = kd; p.D = D; p.n_slots = n_slots; p.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  if (nc == 3) {
    if (stamps) dec_tail_kernel<3, 3, 8, 6, true><<<grid, 1024, lds, s>>>(p);
    else dec_tail_kernel<3, 3, 8, 6, false><<<grid, 1024, lds, s>>>(p);
  } else if (nc == 2) {
    SVLM_CHECK_ARG(!stamps, "svlm_dec_tail: the stamped build exists for hidden sizes of 1025..1536 only");
    dec_tail_kernel<2, 1, 1, 4, false><<<grid, 1024, lds, s>>>(p);
  } else {
    SVLM_CHECK_ARG(!stamps, "svlm_dec_tail: the stamped build exists for hidden sizes of 1025..1536 only");
    dec_tail_kernel<1, 1, 1, 4, false><<<grid, 1024, lds, s>>>(p);
  }
  return svlm_check_launch("svlm_dec_tail");
}
