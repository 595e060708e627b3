// Single-query (decode-step) GQA attention over the slot-mapped KV pool with RoPE-ON-LOAD.
//
// Reference semantics (qwen2/language_forward.py:66-166, shrink mode): keys are cached
// un-rotated (:95-97); on every forward M-RoPE is applied to the query (right-aligned last
// position, :44-53) and to ALL cached keys (:55-63) in the activation dtype
// (x*cos -> bf16, rotate_half(x)*sin -> bf16, sum -> bf16), then causal flash attention
// (:148) with fp32 scores/softmax, P rounded to bf16 before P.V, fp32 accumulation.
// The reference materialises repeat_kv (:107-108) and the rotated K; here one workgroup
// streams each K/V row ONCE for all G query heads of its kv head and rotates in registers.
//
// HBM/latency-bound (2*L*Hkv*D*2 B of K/V per layer-step, ~2 MB at the BASELINE window).
// Split-KV (flash-decoding): grid = (n_splits, Hkv); a workgroup owns up to 64 consecutive
// logical keys = 4 steps of 16 rows (16 lanes x 16 B per 256-B row, 4 rows per wave-load).
// ALL of a workgroup's loads (slot indices, then K, V, cos, sin rows of every step) are issued
// before the first use, so a workgroup costs two dependent memory latencies, not eight.
// Wave shuffles reduce the dots (xor 1,2,4,8) and merge the online-softmax states (xor 16,32);
// the four waves merge through LDS; a second small kernel merges the splits with the splits
// spread over waves and the loads unrolled.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define DA_D 128
#define DA_GMAX 8
#define DA_MAX_STEPS 4     // chunk <= 64

__device__ __forceinline__ void rope8(const float (&x)[8], const float (&xp)[8], const float (&c)[8], const float (&s)[8],
                                      bool upper, float (&out)[8]) {
  // out = bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)); rotate_half = (-x2, x1): lower half pairs with -partner
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float rot = upper ? xp[i] : -xp[i];
    out[i] = rbf(rbf(x[i] * c[i]) + rbf(rot * s[i]));
  }
}

// Split kernel on the matrix cores.  The G query heads of one kv head form the (padded-to-16) column block of two
// small products per 16-key tile -- exactly the prefill kernel's scheme with "queries" = heads:
//   S^T[key][head] = K_rot[key][:] . Q_rot[head][:]     (4 x v_mfma_f32_16x16x32_bf16, K rows from LDS)
//   O^T[d][head]   = V^T[d][key] . P^T[key][head]       (8 x MFMA, V^T via ds_read_b64_tr_b16, half the k-slots used)
// All 256 threads first stage the chunk: every thread rotates up to four 16-B pieces of un-rotated K (RoPE-on-load:
// partner half through a DPP row rotate, cos/sin rows from the table) into LDS and copies V; then wave w owns the
// 16-key tile w.  The VALU work per workgroup drops ~4x against one-dot-product-per-lane-group.
typedef short v4s_da_t __attribute__((ext_vector_type(4)));
#define DA_KLD 136   // K / Q LDS row stride (bf16): 272 B, conflict-free ds_read_b128 fragments
#define DA_VLD 144   // V LDS row stride (bf16): 288 B, conflict-free ds_read_b64_tr_b16
#define DA_OLD 132   // merge-buffer row stride (fp32): 528 B -- at 512 B the 16-B accumulator stores of the G head rows all hit the same banks (4.3 conflict cycles per LDS instruction, profiles/pmc_waits.json)

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) { return __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xF, 0xF, true); }

// Diagnostic build only (tools/decode_attn_stamps.py): wall-clock stamps of the phases of the bounded-window pair, 8 per workgroup
#ifdef SVLM_TUNING
__device__ unsigned long long* da_stamps = nullptr;
extern "C" int svlm_diag_set_da_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(da_stamps), &p, sizeof(p)) == hipSuccess ? SVLM_OK : SVLM_ELAUNCH;
}
#define DA_STAMP(wg, i)                                                                        \
  do {                                                                                         \
    if (threadIdx.x == 0 && da_stamps != nullptr) da_stamps[(size_t)(wg) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define DA_STAMP(wg, i) do { } while (0)
#endif

template <int G>
__global__ __launch_bounds__(256) void decode_attn_split_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k_planes, const bf16_t* __restrict__ v_planes,
    const int* __restrict__ slot_of, const bf16_t* __restrict__ rope_cs, const int* __restrict__ len_dev, int len_add,
    float* __restrict__ ws_m, float* __restrict__ ws_l, float* __restrict__ ws_acc, int Hq, int Hkv, int n_slots,
    int chunk, float scale, int max_len, const bf16_t* __restrict__ k_lin, const bf16_t* __restrict__ v_lin, int lin_rows,
    const int* __restrict__ lin_len_dev) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * DA_KLD * 2 + 64 * DA_VLD * 2 + 16 * DA_KLD * 2 + 512];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(lds);
  bf16_t* Vs = Ks + 64 * DA_KLD;
  bf16_t* Qs = Vs + 64 * DA_VLD;
  float* Om = reinterpret_cast<float*>(lds);                     // [4][16][DA_OLD] fp32 (33 KB of the 35 KB of Ks + Vs), reuses them after the tiles are consumed
  float* Mm = reinterpret_cast<float*>(lds + 64 * DA_KLD * 2 + 64 * DA_VLD * 2 + 16 * DA_KLD * 2);   // [4][16]
  float* Lm = Mm + 64;

  const int start = blockIdx.x * chunk;
  const int kvh = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int srow = tid >> 4, c = tid & 15;      // staging: 16 lanes per 256-B row
  const int wg_id = blockIdx.y * gridDim.x + blockIdx.x;
  (void)wg_id;
  DA_STAMP(wg_id, 0);
  const bool upper = c >= 8;
  const int fc = (c & 7) * 8;
  const bf16_t* kp = k_planes + (size_t)kvh * n_slots * DA_D + c * 8;
  const bf16_t* vp = v_planes + (size_t)kvh * n_slots * DA_D + c * 8;

  // ---- slot lookups do not wait for the length (stale entries are valid slots; masked later)
  int rows[DA_MAX_STEPS], slots[DA_MAX_STEPS];
#pragma unroll
  for (int it = 0; it < DA_MAX_STEPS; ++it) {
    rows[it] = min(start + it * 16 + srow, max_len - 1);
    slots[it] = slot_of[rows[it]];
  }
  u32x4_t qraw = u32x4_t{0, 0, 0, 0};
  if (srow < G) qraw = *reinterpret_cast<const u32x4_t*>(q + (size_t)(kvh * G + srow) * DA_D + c * 8);
  // ---- nor do the rows of the linear planes (header of svlm_decode_attn_lin): tile (start / 16 + it) of the kv head, chunk c of key
  // srow at [c >> 2][(c & 3) * 16 + srow][8] -- the 256 threads of one step read one whole 4 KB tile.  Requested before the lengths are
  // known (the planes are allocated to lin_rows; whether the rows are VALID is decided below), so the common case is ONE memory
  // latency deep: no slot table, no cos / sin rows, no rotation
  u32x4_t kraw[DA_MAX_STEPS], vraw[DA_MAX_STEPS], craw[DA_MAX_STEPS], sraw[DA_MAX_STEPS];
  if (k_lin != nullptr) {
    const int n_lt = lin_rows >> 4;
#pragma unroll
    for (int it = 0; it < DA_MAX_STEPS; ++it) {
      if (it * 16 >= chunk) break;
      const int lt = min((start >> 4) + it, n_lt - 1);
      kraw[it] = *reinterpret_cast<const u32x4_t*>(k_lin + ((size_t)kvh * n_lt + lt) * 2048 + (c >> 2) * 512 + ((c & 3) * 16 + srow) * 8);
      vraw[it] = *reinterpret_cast<const u32x4_t*>(v_lin + ((size_t)kvh * lin_rows + lt * 16 + srow) * DA_D + c * 8);
    }
  }
  // (the planes' state is read NEXT TO the length, not behind the branch on it: one scalar round trip, not two)
  const int lin_r = (k_lin != nullptr && lin_len_dev != nullptr) ? lin_len_dev[0] : 0;
  const int lin_f = (k_lin != nullptr && lin_len_dev != nullptr) ? lin_len_dev[1] : 0;
  const int L = (len_dev ? *len_dev : 0) + len_add;
  if (start >= L) return;
  DA_STAMP(wg_id, 1);
  const int n_rows = min(chunk, L - start);
  const int n_steps = (n_rows + 15) >> 4;        // workgroup-uniform
  const bf16_t* csq = rope_cs + (size_t)(L - 1) * DA_D;      // (behind the length, but not on the critical path: measured with row 0 instead)
  const u32x4_t qc = *reinterpret_cast<const u32x4_t*>(csq + fc);
  const u32x4_t qs = *reinterpret_cast<const u32x4_t*>(csq + 64 + fc);
  // lin_state = {rows rotated, fresh rows present}: rows below lin_state[0] are rotated in the linear planes; with lin_state[1] the rows
  // above it (appended by svlm_dec_qkv_lin since the prefill) are there too, UN-rotated.  A key range wholly below lin_state[0] keeps the
  // rows requested above as they are (wg_lin); the range that holds appended rows keeps them and rotates those (wg_fresh: their cos /
  // sin rows are requested only now, behind the length, but they are a few rows of a table every layer reads); without the fresh rows
  // (a host edit of the logical order since the prefill, a decode step that does not maintain the planes) it takes the pool path
  const int lin_rot = min(lin_r, L);
  const bool lin_fresh = lin_f != 0;
  const bool wg_lin = start + chunk <= lin_rot;             // workgroup-uniform; implies n_rows == chunk
  const bool wg_fresh = !wg_lin && lin_fresh;               // workgroup-uniform
  if (wg_fresh) {
#pragma unroll
    for (int it = 0; it < DA_MAX_STEPS; ++it) {
      if (it >= n_steps) break;
      craw[it] = sraw[it] = u32x4_t{0, 0, 0, 0};
      if (start + it * 16 + srow >= lin_rot) {
        const bf16_t* csr = rope_cs + (size_t)rows[it] * DA_D;
        craw[it] = *reinterpret_cast<const u32x4_t*>(csr + fc);
        sraw[it] = *reinterpret_cast<const u32x4_t*>(csr + 64 + fc);
      }
    }
  } else if (!wg_lin) {
#pragma unroll
    for (int it = 0; it < DA_MAX_STEPS; ++it) {
      if (it >= n_steps) break;
      kraw[it] = *reinterpret_cast<const u32x4_t*>(kp + (size_t)slots[it] * DA_D);
      vraw[it] = *reinterpret_cast<const u32x4_t*>(vp + (size_t)slots[it] * DA_D);
      const bf16_t* csr = rope_cs + (size_t)rows[it] * DA_D;
      craw[it] = *reinterpret_cast<const u32x4_t*>(csr + fc);
      sraw[it] = *reinterpret_cast<const u32x4_t*>(csr + 64 + fc);
    }
  }

  // ---- stage the rotated query block (rows >= G are zero) ...
  {
    u32x4_t outq = u32x4_t{0, 0, 0, 0};
    u32x4_t rp;
#pragma unroll
    for (int i = 0; i < 4; ++i) rp[i] = dpp_u<0x128>(qraw[i]);     // row_ror:8 == partner half of the same row
    if (srow < G) {
      float x[8], xp[8], cc[8], sn[8], o[8];
      unpack8(qraw, x); unpack8(rp, xp); unpack8(qc, cc); unpack8(qs, sn);
      rope8(x, xp, cc, sn, upper, o);
      outq = pack8(o);
    }
    *reinterpret_cast<u32x4_t*>(Qs + srow * DA_KLD + c * 8) = outq;
  }
  // ---- ... and the chunk's keys (rotated) and values; rows past the end are zero
  if (wg_lin) {
#pragma unroll
    for (int it = 0; it < DA_MAX_STEPS; ++it) {
      if (it >= n_steps) break;
      const int lrow = it * 16 + srow;
      *reinterpret_cast<u32x4_t*>(Ks + lrow * DA_KLD + c * 8) = kraw[it];
      *reinterpret_cast<u32x4_t*>(Vs + lrow * DA_VLD + c * 8) = vraw[it];
    }
  } else
#pragma unroll
  for (int it = 0; it < DA_MAX_STEPS; ++it) {
    if (it >= n_steps) break;
    const int lrow = it * 16 + srow;
    u32x4_t kp4;
#pragma unroll
    for (int i = 0; i < 4; ++i) kp4[i] = dpp_u<0x128>(kraw[it][i]);
    float x[8], xp[8], cc[8], sn[8], o[8];
    unpack8(kraw[it], x); unpack8(kp4, xp); unpack8(craw[it], cc); unpack8(sraw[it], sn);
    rope8(x, xp, cc, sn, upper, o);
    const bool ok = lrow < n_rows;
    const bool rotated = wg_fresh && start + lrow < lin_rot;        // (this row came rotated; its cos / sin registers hold nothing)
    const u32x4_t z = u32x4_t{0, 0, 0, 0};
    *reinterpret_cast<u32x4_t*>(Ks + lrow * DA_KLD + c * 8) = ok ? (rotated ? kraw[it] : pack8(o)) : z;
    *reinterpret_cast<u32x4_t*>(Vs + lrow * DA_VLD + c * 8) = ok ? vraw[it] : z;
  }
  DA_STAMP(wg_id, 2);
  __syncthreads();
  DA_STAMP(wg_id, 3);

  // ---- wave w: 16-key tile w
  const int fr = lane & 15, fq = lane >> 4;
  float m_w = -1e30f, l_w = 0.f;
  f32x4_t oacc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (wave < n_steps) {
    f32x4_t sacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(Ks + (wave * 16 + fr) * DA_KLD + ks * 32 + fq * 8);
      const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(Qs + fr * DA_KLD + ks * 32 + fq * 8);
      sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, sacc, 0, 0, 0);
    }
    float sc[4], mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = wave * 16 + fq * 4 + r < n_rows;
      sc[r] = ok ? sacc[r] * scale : -1e30f;
      mx = fmaxf(mx, sc[r]);
    }
    m_w = xor32_max(xor16_max(mx));
    float p[8], rs = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = sc[r] > -1e29f ? __expf(sc[r] - m_w) : 0.f;
      p[r + 4] = 0.f;
      rs += p[r];
    }
    l_w = xor32_sum(xor16_sum(rs));
    u32x4_t pk = pack8(p);
    const bf16x8_t pb = *reinterpret_cast<bf16x8_t*>(&pk);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const bf16_t* vr = Vs + (wave * 16 + fq * 4 + (fr >> 2)) * DA_VLD + dt * 16 + (fr & 3) * 4;
      const v4s_da_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_da_t*)(vr));
      const bf16x8_t a = bf16x8_t{lo[0], lo[1], lo[2], lo[3], lo[0], lo[1], lo[2], lo[3]};   // upper k-slots meet P == 0
      oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb, oacc[dt], 0, 0, 0);
    }
  }
  DA_STAMP(wg_id, 4);
  __syncthreads();          // every wave is done with Ks / Vs: the region becomes the merge buffer
  if (fr < G) {
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4_t*>(Om + ((wave * 16 + fr) * DA_OLD) + dt * 16 + fq * 4) = oacc[dt];
    if (fq == 0) { Mm[wave * 16 + fr] = m_w; Lm[wave * 16 + fr] = l_w; }
  }
  __syncthreads();
  DA_STAMP(wg_id, 5);
  const size_t part = (size_t)blockIdx.x * Hq;
  if (tid < G * 32) {               // one pass: thread (head g, four columns) -- the four waves' weights once per thread, 16-B LDS reads and stores
    const int g = tid >> 5, d = (tid & 31) * 4;
    float mn = Mm[g];
#pragma unroll
    for (int w = 1; w < 4; ++w) mn = fmaxf(mn, Mm[w * 16 + g]);
    float l = 0.f;
    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float e = __expf(Mm[w * 16 + g] - mn);
      l += Lm[w * 16 + g] * e;
      const f32x4_t o = *reinterpret_cast<const f32x4_t*>(Om + (w * 16 + g) * DA_OLD + d);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] += o[i] * e;
    }
    const int hq = kvh * G + g;
    *reinterpret_cast<f32x4_t*>(ws_acc + (part + hq) * DA_D + d) = a;
    if (d == 0) { ws_m[part + hq] = mn; ws_l[part + hq] = l; }
  }
  DA_STAMP(wg_id, 6);
}

// One workgroup per q head; wave w merges splits w, w+4, ... with 4 loads in flight per lane, then the
// waves merge through LDS.  Lane owns d = 2*lane, 2*lane+1.
// Long-cache variant: a workgroup owns up to DA_LONG_MAX consecutive keys and walks them in passes of 64 with the same
// staging + MFMA tile as above; each wave keeps a running (m, l, O) over its 16-key tile of every pass (online softmax),
// so a 32k-key cache needs 128-171 partials per head instead of 512-683 and every workgroup streams 64-128 KB.
// The K / V / cos / sin rows of pass p+1 are in flight (register double buffer, unconditional clamped loads) while pass p
// is rotated, staged and multiplied; the slot indices of the whole range are fetched once into LDS.
#define DA_LONG_MAX 512
#ifdef SVLM_TUNING           // the multi-pass kernel and its DIAG variants live in the diagnostic build only (launch_split below)
struct DaPass {
  u32x4_t k[4], v[4], c[4], s[4];
};

// DIAG (tools only, SVLM_DA_DIAG): 1 = no rotation arithmetic, 2 = no cos/sin loads, 4 = no MFMA work, 8 = no LDS staging -- timing-only
// builds that split the per-pass cost; outputs are wrong by construction.
template <int G, int DIAG = 0>
__global__ __launch_bounds__(256) void decode_attn_long_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k_planes, const bf16_t* __restrict__ v_planes,
    const int* __restrict__ slot_of, const bf16_t* __restrict__ rope_cs, const int* __restrict__ len_dev, int len_add,
    float* __restrict__ ws_m, float* __restrict__ ws_l, float* __restrict__ ws_acc, int Hq, int Hkv, int n_slots,
    int chunk, float scale, int max_len) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * DA_KLD * 2 + 64 * DA_VLD * 2 + 16 * DA_KLD * 2 + 512 + DA_LONG_MAX * 4];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(lds);
  bf16_t* Vs = Ks + 64 * DA_KLD;
  bf16_t* Qs = Vs + 64 * DA_VLD;
  float* Om = reinterpret_cast<float*>(lds);
  float* Mm = reinterpret_cast<float*>(lds + 64 * DA_KLD * 2 + 64 * DA_VLD * 2 + 16 * DA_KLD * 2);
  float* Lm = Mm + 64;
  int* slots_s = reinterpret_cast<int*>(lds + 64 * DA_KLD * 2 + 64 * DA_VLD * 2 + 16 * DA_KLD * 2 + 512);

  const int start = blockIdx.x * chunk;
  const int kvh = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int srow = tid >> 4, c = tid & 15;
  const bool upper = c >= 8;
  const int fc = (c & 7) * 8;
  const bf16_t* kp = k_planes + (size_t)kvh * n_slots * DA_D + c * 8;
  const bf16_t* vp = v_planes + (size_t)kvh * n_slots * DA_D + c * 8;

  for (int i = tid; i < chunk; i += 256) slots_s[i] = slot_of[min(start + i, max_len - 1)];
  u32x4_t qraw = u32x4_t{0, 0, 0, 0};
  if (srow < G) qraw = *reinterpret_cast<const u32x4_t*>(q + (size_t)(kvh * G + srow) * DA_D + c * 8);
  const int L = (len_dev ? *len_dev : 0) + len_add;
  if (start >= L) return;                                   // workgroup-uniform
  const int n_rows = min(chunk, L - start);
  const int n_pass = (n_rows + 63) >> 6;
  {
    const bf16_t* csq = rope_cs + (size_t)(L - 1) * DA_D;
    const u32x4_t qc = *reinterpret_cast<const u32x4_t*>(csq + fc);
    const u32x4_t qs = *reinterpret_cast<const u32x4_t*>(csq + 64 + fc);
    u32x4_t outq = u32x4_t{0, 0, 0, 0};
    u32x4_t rp;
#pragma unroll
    for (int i = 0; i < 4; ++i) rp[i] = dpp_u<0x128>(qraw[i]);
    if (srow < G) {
      float x[8], xp[8], cc[8], sn[8], o[8];
      unpack8(qraw, x); unpack8(rp, xp); unpack8(qc, cc); unpack8(qs, sn);
      rope8(x, xp, cc, sn, upper, o);
      outq = pack8(o);
    }
    *reinterpret_cast<u32x4_t*>(Qs + srow * DA_KLD + c * 8) = outq;
  }
  __syncthreads();                                          // slots_s and Qs visible

  auto load_pass = [&](int p, DaPass& b) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lr = min(p * 64 + it * 16 + srow, n_rows - 1);      // clamped: rows past the end are zeroed when staged
      const int slot = slots_s[lr];
      b.k[it] = *reinterpret_cast<const u32x4_t*>(kp + (size_t)slot * DA_D);
      b.v[it] = *reinterpret_cast<const u32x4_t*>(vp + (size_t)slot * DA_D);
      if constexpr (DIAG & 2) {
        b.c[it] = b.k[it]; b.s[it] = b.v[it];
      } else {
        const bf16_t* csr = rope_cs + (size_t)(start + lr) * DA_D;
        b.c[it] = *reinterpret_cast<const u32x4_t*>(csr + fc);
        b.s[it] = *reinterpret_cast<const u32x4_t*>(csr + 64 + fc);
      }
    }
  };
  auto stage = [&](int p, const DaPass& b) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lrow = it * 16 + srow;
      u32x4_t kp4;
#pragma unroll
      for (int i = 0; i < 4; ++i) kp4[i] = dpp_u<0x128>(b.k[it][i]);
      const bool ok = p * 64 + lrow < n_rows;
      const u32x4_t z = u32x4_t{0, 0, 0, 0};
      u32x4_t kr;
      if constexpr (DIAG & 1) {
        kr = b.k[it] ^ b.c[it] ^ b.s[it] ^ kp4;
      } else {
        float x[8], xp[8], cc[8], sn[8], o[8];
        unpack8(b.k[it], x); unpack8(kp4, xp); unpack8(b.c[it], cc); unpack8(b.s[it], sn);
        rope8(x, xp, cc, sn, upper, o);
        kr = pack8(o);
      }
      if constexpr (DIAG & 8) {
        asm volatile("" ::"v"(kr), "v"(b.v[it]));
      } else {
        *reinterpret_cast<u32x4_t*>(Ks + lrow * DA_KLD + c * 8) = ok ? kr : z;
        *reinterpret_cast<u32x4_t*>(Vs + lrow * DA_VLD + c * 8) = ok ? b.v[it] : z;
      }
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  float m_w = -1e30f, l_w = 0.f;
  f32x4_t oacc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int p) {
    const int base = p * 64 + wave * 16;                    // this wave's tile of the pass
    if (base >= n_rows) return;                             // wave-uniform
    if constexpr (DIAG & 4) return;
    f32x4_t sacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(Ks + (wave * 16 + fr) * DA_KLD + ks * 32 + fq * 8);
      const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(Qs + fr * DA_KLD + ks * 32 + fq * 8);
      sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, sacc, 0, 0, 0);
    }
    float sc[4], mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = base + fq * 4 + r < n_rows;
      sc[r] = ok ? sacc[r] * scale : -1e30f;
      mx = fmaxf(mx, sc[r]);
    }
    const float m_new = fmaxf(m_w, xor32_max(xor16_max(mx)));
    const float alpha = __expf(m_w - m_new);
    float p8[8], rs = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p8[r] = sc[r] > -1e29f ? __expf(sc[r] - m_new) : 0.f;
      p8[r + 4] = 0.f;
      rs += p8[r];
    }
    l_w = l_w * alpha + xor32_sum(xor16_sum(rs));
    m_w = m_new;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
    u32x4_t pk = pack8(p8);
    const bf16x8_t pb = *reinterpret_cast<bf16x8_t*>(&pk);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const bf16_t* vr = Vs + (wave * 16 + fq * 4 + (fr >> 2)) * DA_VLD + dt * 16 + (fr & 3) * 4;
      const v4s_da_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_da_t*)(vr));
      const bf16x8_t a = bf16x8_t{lo[0], lo[1], lo[2], lo[3], lo[0], lo[1], lo[2], lo[3]};
      oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb, oacc[dt], 0, 0, 0);
    }
  };

  DaPass A, B;
  load_pass(0, A);
  for (int p = 0; p < n_pass; p += 2) {
    load_pass(min(p + 1, n_pass - 1), B);                   // unconditional: a redundant reload at the end costs nothing
    __builtin_amdgcn_sched_barrier(0);
    stage(p, A);
    lds_barrier();
    compute(p);
    lds_barrier();
    if (p + 1 >= n_pass) break;
    load_pass(min(p + 2, n_pass - 1), A);
    __builtin_amdgcn_sched_barrier(0);
    stage(p + 1, B);
    lds_barrier();
    compute(p + 1);
    lds_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing redundant loads must not outlive the wave's registers
  if (fr < G) {
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4_t*>(Om + ((wave * 16 + fr) * DA_D) + dt * 16 + fq * 4) = oacc[dt];
    if (fq == 0) { Mm[wave * 16 + fr] = m_w; Lm[wave * 16 + fr] = l_w; }
  }
  __syncthreads();
  const size_t part = (size_t)blockIdx.x * Hq;
  for (int idx = tid; idx < G * DA_D; idx += 256) {
    const int g = idx / DA_D, d = idx % DA_D;
    float mn = Mm[g];
#pragma unroll
    for (int w = 1; w < 4; ++w) mn = fmaxf(mn, Mm[w * 16 + g]);
    float l = 0.f, a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float e = __expf(Mm[w * 16 + g] - mn);
      l += Lm[w * 16 + g] * e;
      a += Om[(w * 16 + g) * DA_D + d] * e;
    }
    const int hq = kvh * G + g;
    ws_acc[(part + hq) * DA_D + d] = a;
    if (d == 0) { ws_m[part + hq] = mn; ws_l[part + hq] = l; }
  }
}
#endif       // SVLM_TUNING

// Streaming variant for long caches (chunk % 64 == 0): the multi-pass kernel above meets at two workgroup barriers per 64-key pass,
// which leaves it latency-bound (built with no arithmetic at all it still needs 23 of its 30 us at 32k keys x 4 kv heads:
// tools/decode_attn_sweep.py with SVLM_DA_DIAG).  Here every WAVE is its own pipeline over 16-key tiles (wave w takes tiles w, w+4,
// ... of the workgroup's key range) with NO barrier in the loop:
//   * K rows are loaded straight into the MFMA A-operand layout: lane (key = lane & 15, fq = lane >> 4) takes the four 16-B pieces
//     d = 32 ks + 8 fq .. + 7 (ks = 0..3) of its key's row -- a key's rotation partner d +- 64 is piece ks +- 2 of the SAME lane, so
//     RoPE-on-load is lane-local arithmetic on registers, with the cos / sin pieces loaded in the same pattern (whole 256-B rows
//     through a second LDS slab were measured too: same speed at 32k keys, 4-10 % slower at 131k);
//   * V rows go through a 4.5 KB LDS slab PRIVATE to the wave (ds_write_b128, then the transposing ds_read_b64_tr_b16 of the
//     kernel above): LDS operations of one wave execute in order, no barrier is needed;
//   * the loads of the wave's NEXT tile (K, cos/sin, V: 12 KB per wave) are in flight while the current one is rotated and
//     multiplied; 12 resident waves per CU keep ~140 KB per CU in flight.
// Everything after the loop (4-wave merge, fp32 partials for decode_attn_combine_kernel) is the multi-pass kernel's.
// Measured (MI355X, tools/decode_attn_sweep.py, split + combine): 4 kv heads x 131k keys 71 us = 3.8 TB/s of K/V; 4 x 32k 25.8 us
// (multi-pass kernel 30.6); 2 x 32k 17.4 us (one-pass kernel 21.0).
#define DA_STREAM_TPW 8          // tiles per wave: chunk <= 16 * 4 * 8 = 512 keys per workgroup
struct DaTile {
  u32x4_t k[4], cs[4], v[4];     // K pieces ks = 0..3; cos pieces 0,1 and sin pieces 0,1; V rows 4 i + (lane >> 4), chunk lane & 15
};

template <int G, int DIAG = 0, bool WLIN = false>       // WLIN: launched with linear planes (below); DIAG bits (diagnostic build only): 1 = no cos/sin loads, no rotation; 2 = loads only, no arithmetic at all; 4 = cos/sin rows from a 16 KB table
__global__ __launch_bounds__(256, 2) void decode_attn_stream_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k_planes, const bf16_t* __restrict__ v_planes,
    const int* __restrict__ slot_of, const bf16_t* __restrict__ rope_cs, const int* __restrict__ len_dev, int len_add,
    float* __restrict__ ws_m, float* __restrict__ ws_l, float* __restrict__ ws_acc, int Hq, int Hkv, int n_slots,
    int chunk, float scale, int max_len, const bf16_t* __restrict__ k_lin, const bf16_t* __restrict__ v_lin, int lin_rows,
    const int* __restrict__ lin_len_dev) {
  // LDS: [4][16][DA_OLD] fp32 merge buffer (33 KB), whose first 18 KB double as the four waves' V slabs during the loop; Q block; m / l
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 16 * DA_OLD * 4 + 16 * DA_KLD * 2 + 512];
  float* Om = reinterpret_cast<float*>(lds);
  bf16_t* Qs = reinterpret_cast<bf16_t*>(lds + 4 * 16 * DA_OLD * 4);
  float* Mm = reinterpret_cast<float*>(lds + 4 * 16 * DA_OLD * 4 + 16 * DA_KLD * 2);
  float* Lm = Mm + 64;

  // the LAST key range is dispatched FIRST: it is the one that holds the rows appended since the prefill (pool path below, slower per tile)
  const int sp = gridDim.x - 1 - blockIdx.x;
  const int start = sp * chunk;
  const int kvh = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  bf16_t* Vw = reinterpret_cast<bf16_t*>(lds) + wave * 16 * DA_VLD;       // this wave's V slab: 16 rows x 288 B
  const bf16_t* kp = k_planes + (size_t)kvh * n_slots * DA_D;
  const bf16_t* vp = v_planes + (size_t)kvh * n_slots * DA_D;

  // linear planes: the 16-key tile r of kv head h is 4 KB at k_lin + (h * lin_rows / 16 + r) * 2048, laid out as the four operand loads
  // of this kernel: [ks][lane][8] (lane = fq * 16 + key) -- every load instruction reads 1 KB of consecutive bytes; V rows are row-major
  auto load_lin = [&](int t, DaTile& b) {
    const int rt = min((start >> 4) + t * 4 + wave, (lin_rows >> 4) - 1);      // (clamped: the first tile is requested before the lengths are known)
    const bf16_t* kt = k_lin + ((size_t)kvh * (lin_rows >> 4) + rt) * 2048 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b.k[ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kt + ks * 512));
    const bf16_t* vt = v_lin + ((size_t)kvh * lin_rows + rt * 16 + fq) * DA_D + fr * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) b.v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vt + (size_t)i * 4 * DA_D));
  };
  DaTile A, B;
  // WLIN: the wave's first tile is requested NOW -- the planes are allocated to lin_rows, whether the rows are valid is decided below --
  // so the query staging (two dependent loads, a rotation, a barrier: ~1.5 us) runs underneath the first K/V bytes
  if constexpr (WLIN) load_lin(0, A);
  // rotated query block -> Qs (rows >= G are zero), exactly as the kernels above
  {
    const int srow = tid >> 4, c = tid & 15;
    u32x4_t qraw = u32x4_t{0, 0, 0, 0};
    if (srow < G) qraw = *reinterpret_cast<const u32x4_t*>(q + (size_t)(kvh * G + srow) * DA_D + c * 8);
    const int Lq = (len_dev ? *len_dev : 0) + len_add;
    const bf16_t* csq = rope_cs + (size_t)(max(Lq, 1) - 1) * DA_D;
    const u32x4_t qc = *reinterpret_cast<const u32x4_t*>(csq + (c & 7) * 8);
    const u32x4_t qs = *reinterpret_cast<const u32x4_t*>(csq + 64 + (c & 7) * 8);
    u32x4_t outq = u32x4_t{0, 0, 0, 0};
    u32x4_t rp;
#pragma unroll
    for (int i = 0; i < 4; ++i) rp[i] = dpp_u<0x128>(qraw[i]);
    if (srow < G) {
      float x[8], xp[8], cc[8], sn[8], o[8];
      unpack8(qraw, x); unpack8(rp, xp); unpack8(qc, cc); unpack8(qs, sn);
      rope8(x, xp, cc, sn, c >= 8, o);
      outq = pack8(o);
    }
    *reinterpret_cast<u32x4_t*>(Qs + srow * DA_KLD + c * 8) = outq;
  }
  const int lin_r = WLIN ? lin_len_dev[0] : 0, lin_f = WLIN ? lin_len_dev[1] : 0;      // (next to the length: one scalar round trip)
  const int L = (len_dev ? *len_dev : 0) + len_add;
  if (start >= L) {                                         // workgroup-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the requested tile must have landed before the wave's registers are released)
    return;
  }
  const int n_rows = min(chunk, L - start);
  const int n_tiles = (n_rows + 15) >> 4;                   // of the workgroup; wave w owns tiles w, w + 4, ...
  // WLIN: rows [0, lin_len) of the cache also exist ROTATED, in logical order, in the linear planes the prefill left behind (header of
  // svlm_decode_attn_lin): tiles that lie there are streamed from them (no slot table, no cos/sin rows, no rotation).  Without
  // linear planes (!WLIN) the kernel is the pipelined pool path alone.
  const int lin_len = min(lin_r, L);
  const bool lin_fresh = lin_f != 0;                        // rows above lin_len are in the planes too, un-rotated (svlm_dec_qkv_lin)
  lds_barrier();                                            // Qs visible (LDS only: the tile requested above stays in flight)
  bf16x8_t qf[4];                                           // this lane's B fragments of the query block: constant over the loop
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(Qs + fr * DA_KLD + ks * 32 + fq * 8);

  auto load_tile = [&](int t, int slot_own, DaTile& b) {
    // K in operand layout: piece ks of row `slot_own`; cos / sin pieces of the key's logical row
    const bf16_t* krow = kp + (size_t)slot_own * DA_D + fq * 8;
    int lrow = min(start + (t * 4 + wave) * 16 + fr, max_len - 1);
    if constexpr ((DIAG & 8) != 0) lrow = (lrow >> 4) & 63;  // timing only: the 16 keys of a tile share one cos/sin row (one cache line per load instruction)
    if constexpr ((DIAG & 4) != 0) lrow &= 63;               // timing only: every cos/sin row from a 16 KB table (what an L1-resident table would cost)
    const bf16_t* csr = rope_cs + (size_t)lrow * DA_D + fq * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b.k[ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(krow + ks * 32));
    if constexpr ((DIAG & 1) == 0) {
      b.cs[0] = *reinterpret_cast<const u32x4_t*>(csr);
      b.cs[1] = *reinterpret_cast<const u32x4_t*>(csr + 32);
      b.cs[2] = *reinterpret_cast<const u32x4_t*>(csr + 64);
      b.cs[3] = *reinterpret_cast<const u32x4_t*>(csr + 96);
    }
    // V rows for the slab: row 4 i + fq, 16-B chunk fr; that row's slot sits in lane 4 i + fq
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vslot = __shfl(slot_own, 4 * i + fq, 64);
      b.v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vp + (size_t)vslot * DA_D + fr * 8));
    }
  };

  float m_w = -1e30f, l_w = 0.f;
  f32x4_t oacc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](auto mode_c, int t, const DaTile& b) {
    constexpr int MODE = decltype(mode_c)::value;           // 0: rotate every key (pool rows); 1: keys come rotated (linear planes);
    constexpr bool LIN = MODE == 1;                         // 2: linear planes, keys at or above lin_len are rotated here
    const int base = (t * 4 + wave) * 16;                   // first key of the tile, relative to `start`
    if constexpr ((DIAG & 2) != 0) {                        // timing only: every loaded register is consumed, nothing else happens
      unsigned x = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) x ^= b.k[i][0] ^ b.k[i][1] ^ b.k[i][2] ^ b.k[i][3] ^ b.v[i][0] ^ b.v[i][1] ^ b.v[i][2] ^ b.v[i][3];
      if constexpr ((DIAG & 1) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x ^= b.cs[i][0] ^ b.cs[i][1] ^ b.cs[i][2] ^ b.cs[i][3];
      }
      oacc[0][0] += __uint_as_float(x & 0x3F800000u);
      l_w = 1.f; m_w = 0.f;
      return;
    }
    // ---- RoPE on registers: out(d) = bf(bf(x c) + bf(rot s)), rot = -x(d + 64) for d < 64, x(d - 64) above
    bf16x8_t kf[4];
    if constexpr ((DIAG & 1) != 0 || LIN) {
#pragma unroll
      for (int h = 0; h < 4; ++h) { u32x4_t kk = b.k[h]; kf[h] = *reinterpret_cast<bf16x8_t*>(&kk); }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float xl[8], xu[8], cc[8], sn[8], ol[8], ou[8];
        unpack8(b.k[h], xl); unpack8(b.k[h + 2], xu); unpack8(b.cs[h], cc); unpack8(b.cs[2 + h], sn);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ol[i] = rbf(rbf(xl[i] * cc[i]) + rbf(-xu[i] * sn[i]));
          ou[i] = rbf(rbf(xu[i] * cc[i]) + rbf(xl[i] * sn[i]));
        }
        u32x4_t pl = pack8(ol), pu = pack8(ou);
        if constexpr (MODE == 2) {                          // this lane's key (row base + fr) came rotated
          if (start + base + fr < lin_len) { pl = b.k[h]; pu = b.k[h + 2]; }
        }
        kf[h] = *reinterpret_cast<bf16x8_t*>(&pl);
        kf[h + 2] = *reinterpret_cast<bf16x8_t*>(&pu);
      }
    }
    // ---- V to the wave's slab (LDS operations of one wave are executed in order: no barrier)
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(Vw + (4 * i + fq) * DA_VLD + fr * 8) = b.v[i];
    if (base + 16 > n_rows) {       // (wave-uniform) the tile that crosses the end of the cache: rows behind it come from clamped slots and
                                    // may hold anything (a pool need not be zero-filled); their P is 0, but 0 x Inf / NaN is NaN in the MFMA
      const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (base + 4 * i + fq >= n_rows) *reinterpret_cast<u32x4_t*>(Vw + (4 * i + fq) * DA_VLD + fr * 8) = z;
    }
    f32x4_t sacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], sacc, 0, 0, 0);
    float sc[4], mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = base + fq * 4 + r < n_rows;
      sc[r] = ok ? sacc[r] * scale : -1e30f;
      mx = fmaxf(mx, sc[r]);
    }
    const float m_new = fmaxf(m_w, xor32_max(xor16_max(mx)));
    const float alpha = __expf(m_w - m_new);
    float p8[8], rs = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p8[r] = sc[r] > -1e29f ? __expf(sc[r] - m_new) : 0.f;
      p8[r + 4] = 0.f;
      rs += p8[r];
    }
    l_w = l_w * alpha + xor32_sum(xor16_sum(rs));
    m_w = m_new;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
    u32x4_t pk = pack8(p8);
    const bf16x8_t pb = *reinterpret_cast<bf16x8_t*>(&pk);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const bf16_t* vr = Vw + (fq * 4 + (fr >> 2)) * DA_VLD + dt * 16 + (fr & 3) * 4;
      const v4s_da_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_da_t*)(vr));
      const bf16x8_t a = bf16x8_t{lo[0], lo[1], lo[2], lo[3], lo[0], lo[1], lo[2], lo[3]};   // upper k-slots meet P == 0
      oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb, oacc[dt], 0, 0, 0);
    }
  };

  // software pipeline, two named tile buffers; a wave's tile index runs 0 .. nt - 1 (workgroup tile 4 t + wave)
  const int nt = n_tiles > wave ? (n_tiles - wave + 3) >> 2 : 0;   // tiles of this wave
  if constexpr (WLIN) {
    // the wave's tiles that lie wholly below lin_len (a prefix of them: all but the one or two that hold the rows appended since the
    // prefill) are streamed from the linear planes, pipelined (tile 0 is already on its way); the rest one at a time
    const int ntl = min(nt, max(0, ((lin_len - start) >> 4) - wave + 3) >> 2);
#pragma unroll
    for (int t = 0; t < DA_STREAM_TPW; t += 2) {
      if (t >= ntl) break;                                  // wave-uniform
      if (t + 1 < DA_STREAM_TPW) load_lin(min(t + 1, ntl - 1), B);      // redundant reload at the end: harmless
      __builtin_amdgcn_sched_barrier(0);
      compute(std::integral_constant<int, 1>{}, t, A);
      if (t + 1 >= ntl) break;
      if (t + 2 < DA_STREAM_TPW) load_lin(min(t + 2, ntl - 1), A);
      __builtin_amdgcn_sched_barrier(0);
      compute(std::integral_constant<int, 1>{}, t + 1, B);
    }
    for (int t = ntl; t < nt; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a trailing redundant load of the loop above may still be writing A
      if (lin_fresh) {                                      // (kernel-uniform) the appended rows are in the planes: rotate them here
        load_lin(t, A);
        const bf16_t* csr = rope_cs + (size_t)min(start + (t * 4 + wave) * 16 + fr, max_len - 1) * DA_D + fq * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) A.cs[i] = *reinterpret_cast<const u32x4_t*>(csr + i * 32);
        compute(std::integral_constant<int, 2>{}, t, A);
      } else {                                              // planes not maintained since the prefill: pool rows
        load_tile(t, slot_of[min(start + (t * 4 + wave) * 16 + fr, max_len - 1)], A);
        compute(std::integral_constant<int, 0>{}, t, A);
      }
    }
  } else {
    // slot of key (lane & 15) in each of the wave's tiles (stale / clamped entries are valid slots; masked later)
    int slot_t[DA_STREAM_TPW];
#pragma unroll
    for (int t = 0; t < DA_STREAM_TPW; ++t) slot_t[t] = slot_of[min(start + (t * 4 + wave) * 16 + fr, max_len - 1)];
    if (nt > 0) load_tile(0, slot_t[0], A);
#pragma unroll
    for (int t = 0; t < DA_STREAM_TPW; t += 2) {
      if (t >= nt) break;                                   // wave-uniform
      if (t + 1 < DA_STREAM_TPW) load_tile(min(t + 1, nt - 1), slot_t[t + 1 < DA_STREAM_TPW ? t + 1 : t], B);      // redundant reload at the end: harmless
      __builtin_amdgcn_sched_barrier(0);
      compute(std::integral_constant<int, 0>{}, t, A);
      if (t + 1 >= nt) break;
      if (t + 2 < DA_STREAM_TPW) load_tile(min(t + 2, nt - 1), slot_t[t + 2 < DA_STREAM_TPW ? t + 2 : t], A);
      __builtin_amdgcn_sched_barrier(0);
      compute(std::integral_constant<int, 0>{}, t + 1, B);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // trailing redundant loads must not outlive the registers
  __syncthreads();                                          // every wave is done with its V slab: the region becomes the merge buffer
  if (fr < G) {
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4_t*>(Om + ((wave * 16 + fr) * DA_OLD) + dt * 16 + fq * 4) = oacc[dt];
    if (fq == 0) { Mm[wave * 16 + fr] = m_w; Lm[wave * 16 + fr] = l_w; }
  }
  __syncthreads();
  const size_t part = (size_t)sp * Hq;
  if (tid < G * 32) {               // thread (head g, four columns), as in the bounded-window kernel
    const int g = tid >> 5, d = (tid & 31) * 4;
    float mn = Mm[g];
#pragma unroll
    for (int w = 1; w < 4; ++w) mn = fmaxf(mn, Mm[w * 16 + g]);
    float l = 0.f;
    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float e = __expf(Mm[w * 16 + g] - mn);
      l += Lm[w * 16 + g] * e;
      const f32x4_t o = *reinterpret_cast<const f32x4_t*>(Om + (w * 16 + g) * DA_OLD + d);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] += o[i] * e;
    }
    const int hq = kvh * G + g;
    *reinterpret_cast<f32x4_t*>(ws_acc + (part + hq) * DA_D + d) = a;
    if (d == 0) { ws_m[part + hq] = mn; ws_l[part + hq] = l; }
  }
}

// Merge of the splits.  grid = (Hq, DS): a workgroup owns 128/DS output columns of one head, so a long cache's partials
// (3 MB at 32k keys) are pulled by DS x Hq CUs instead of Hq.  Inside a wave the 64 lanes are 64/CW split sub-slots x CW
// column pairs (CW = 64/DS): wave w, sub-slot j covers splits (w*SUB + j) + NW*SUB*u, in batches of DA_CB whose loads are ALL
// issued before anything is consumed (the partials were written by other CUs: every dependent round trip is ~1-2 us of
// L2/fabric latency, and the kernel is nothing but such round trips).  43 splits (2k keys) on 4 waves = one batch.
#ifndef DA_CB
#define DA_CB 12
#endif
template <int NW, int DS>
__global__ __launch_bounds__(NW * 64) void decode_attn_combine_kernel(const float* __restrict__ ws_m, const float* __restrict__ ws_l,
                                                                      const float* __restrict__ ws_acc, const int* __restrict__ len_dev,
                                                                      int len_add, bf16_t* __restrict__ out, int Hq, int chunk, int ns_max) {
  constexpr int CW = 64 / DS;            // lanes across the workgroup's columns (2 columns each)
  constexpr int SUB = DS;                // split sub-slots per wave
  constexpr int STRIDE = NW * SUB;       // splits covered per step of u
  const int hq = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane % CW, sub = lane / CW;
  const int cwg = 4096 + blockIdx.y * gridDim.x + blockIdx.x;      // (stamp rows of the combine sit behind the split's)
  (void)cwg;
  DA_STAMP(cwg, 0);
  const int col = blockIdx.y * (DA_D / DS) + 2 * cl;
  const int first = wave * SUB + sub;
  // The first batch does not wait for the length: slots beyond the live splits are valid memory of the workspace (clamped to
  // its last slot) whose stale contents are skipped below, so the partial loads and the length load are ONE round trip, not two.
  float m_[DA_CB], l_[DA_CB];
  float2 v_[DA_CB];
#pragma unroll
  for (int u = 0; u < DA_CB; ++u) {
    const int i0 = min(first + STRIDE * u, ns_max - 1);
    const size_t p = (size_t)i0 * Hq + hq;
    m_[u] = ws_m[p];
    l_[u] = ws_l[p];
    v_[u] = *reinterpret_cast<const float2*>(ws_acc + p * DA_D + col);
  }
  const int L = (len_dev ? *len_dev : 0) + len_add;
  const int ns = (L + chunk - 1) / chunk;
  // Online merge: the running maximum of a wave comes from the m values it already holds in registers (one DPP/permlane
  // reduction per batch), not from a second pass over ws_m -- that pass was a dependent round trip to another CU's partials
  // behind the length, as long as the whole first batch.  The waves' (m, l, acc) triples meet in LDS below.
  float mrun = -1e30f, l = 0.f, a0 = 0.f, a1 = 0.f;
  for (int base = 0;;) {
    float bm = -1e30f;
#pragma unroll
    for (int u = 0; u < DA_CB; ++u)
      if (base + first + STRIDE * u < ns) bm = fmaxf(bm, m_[u]);
    const float mnew = fmaxf(mrun, wave_max(bm));
    const float alpha = __expf(mrun - mnew);
    l *= alpha; a0 *= alpha; a1 *= alpha;
    mrun = mnew;
#pragma unroll
    for (int u = 0; u < DA_CB; ++u) {
      if (base + first + STRIDE * u < ns) {
        const float e = __expf(m_[u] - mrun);
        l += l_[u] * e;
        a0 += v_[u].x * e;
        a1 += v_[u].y * e;
      }
    }
    base += STRIDE * DA_CB;
    if (base >= ns) break;
#pragma unroll
    for (int u = 0; u < DA_CB; ++u) {
      const int i0 = min(base + first + STRIDE * u, ns - 1);
      const size_t p = (size_t)i0 * Hq + hq;
      m_[u] = ws_m[p];
      l_[u] = ws_l[p];
      v_[u] = *reinterpret_cast<const float2*>(ws_acc + p * DA_D + col);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // fold the sub-slots of the wave (lanes cl, cl + CW, ...; they share mrun), then the waves through LDS
  if constexpr (DS >= 2) { l += __shfl_xor(l, 32, 64); a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64); }
  if constexpr (DS >= 4) { l += __shfl_xor(l, 16, 64); a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64); }
  __shared__ float sm[NW];
  __shared__ float sl[NW];
  __shared__ float sa[NW][DA_D / DS];
  if (lane == 0) { sm[wave] = mrun; sl[wave] = l; }
  if (sub == 0) {
    sa[wave][2 * cl] = a0;
    sa[wave][2 * cl + 1] = a1;
  }
  DA_STAMP(cwg, 1);
  __syncthreads();
  DA_STAMP(cwg, 2);
  if (threadIdx.x < DA_D / DS) {
    const int d = threadIdx.x;
    float mt = sm[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mt = fmaxf(mt, sm[w]);
    float lt = 0.f, at = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float e = __expf(sm[w] - mt);          // a wave without a live split holds (-1e30, 0, 0): e underflows to 0
      lt += sl[w] * e;
      at += sa[w][d] * e;
    }
    out[(size_t)hq * DA_D + blockIdx.y * (DA_D / DS) + d] = f2bf(at / lt);
  }
  DA_STAMP(cwg, 3);
}

// ws layout: [max_splits*Hq] m | [max_splits*Hq] l | [max_splits*Hq*128] acc   (floats)
extern "C" long long svlm_decode_attn_ws_bytes(int Hq, int max_len, int chunk) {
  if (Hq <= 0 || max_len <= 0 || chunk <= 0) return SVLM_EINVAL;
  const long long ns = (max_len + chunk - 1) / chunk;
  return ns * Hq * (2 + DA_D) * (long long)sizeof(float);
}

template <int G>
static void launch_split(dim3 grid, hipStream_t st, const bf16_t* q, const bf16_t* kp, const bf16_t* vp, const int* slot_of,
                         const bf16_t* cs, const int* len_dev, int len_add, float* ws_m, float* ws_l, float* ws_acc, int Hq, int Hkv,
                         int n_slots, int chunk, float scale, int max_len, const bf16_t* k_lin, const bf16_t* v_lin, int lin_rows, const int* lin_len) {
#ifdef SVLM_TUNING
  // diagnostic build only: the multi-pass long-cache kernel the streaming kernel replaced, and its timing-only DIAG variants
  static const int diag = svlm_env("SVLM_DA_DIAG") ? atoi(svlm_env("SVLM_DA_DIAG")) : 0;
  static const bool stream_k = svlm_env("SVLM_DA_STREAM") == nullptr || atoi(svlm_env("SVLM_DA_STREAM")) != 0;
  if (chunk > 16 * DA_MAX_STEPS && !stream_k) {
    if constexpr (G == 6 || G == 7) {
      switch (diag) {
        case 1: decode_attn_long_kernel<G, 1><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len); return;
        case 2: decode_attn_long_kernel<G, 2><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len); return;
        case 3: decode_attn_long_kernel<G, 3><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len); return;
        case 4: decode_attn_long_kernel<G, 4><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len); return;
        case 7: decode_attn_long_kernel<G, 7><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len); return;
        case 15: decode_attn_long_kernel<G, 15><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len); return;
        default: break;
      }
    }
    decode_attn_long_kernel<G><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len);
    return;
  }
#endif
  // caches beyond 16 * DA_MAX_STEPS keys per workgroup: the barrier-free streaming kernel (chunk <= 64 * DA_STREAM_TPW = DA_LONG_MAX,
  // checked by the caller); the bounded windows: the split kernel
#ifdef SVLM_TUNING
  if (chunk > 16 * DA_MAX_STEPS && diag != 0) {
    if constexpr (G == 6 || G == 7) {
      if (diag == 1) decode_attn_stream_kernel<G, 1><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
      else if (diag == 2) decode_attn_stream_kernel<G, 2><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
      else if (diag == 3) decode_attn_stream_kernel<G, 3><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
      else if (diag == 8) decode_attn_stream_kernel<G, 8><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
      else if (diag == 4) decode_attn_stream_kernel<G, 4><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
      else decode_attn_stream_kernel<G, 6><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
      return;
    }
  }
#endif
  if (chunk > 16 * DA_MAX_STEPS && k_lin != nullptr)
    decode_attn_stream_kernel<G, 0, true><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
  else if (chunk > 16 * DA_MAX_STEPS)
    decode_attn_stream_kernel<G><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
  else
    decode_attn_split_kernel<G><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, k_lin, v_lin, lin_rows, lin_len);
}

// The same attention with the LINEAR PLANES of the layer beside the pool.  The reference rotates every cached key in every forward
// (language_forward.py:55-63); between two evictions the positions of the cached rows do not change, so the rotated keys the prefill
// of a chunk has to produce anyway (rope_gather_kernel, flash_attn.hip) are kept -- svlm_prefill_attn_ropeload_lin -- and every decode
// step of the chunk streams them instead of rotating the pool rows again:
//   k_lin  (Hkv, lin_rows / 16, 4, 64, 8) bf16: 16-key tiles of ROTATED keys in logical order, each tile laid out as the four MFMA
//          operand loads of the decode kernels ([ks][lane = fq * 16 + key][8]: one load instruction = 1 KB of consecutive bytes)
//   v_lin  (Hkv, lin_rows, 128) bf16: values in logical order
//   lin_len_dev[2] = {R, F}: rows [0, R) of both are valid, keys ROTATED (written by the prefill; the host lowers R when it edits the
//          logical order); F != 0: the rows appended since (svlm_dec_qkv_lin writes them) follow at their logical rows, keys UN-rotated
// A key range below R needs no slot table, no cos / sin rows and no rotation; the range with the appended rows (at most max_new_tokens
// of them) rotates those while it stages them -- or, with F == 0 (a host edit since the prefill, a decode step that does not maintain
// the planes), takes the pool path.  Same bits every way: the rotation arithmetic of all producers is the same expression.
extern "C" int svlm_decode_attn_lin(const void* q, const void* k_planes, const void* v_planes, const int* slot_of,
                                    const void* rope_cs, const int* len_dev, int len_add, const void* k_lin, const void* v_lin,
                                    int lin_rows, const int* lin_len_dev, void* out, void* ws,
                                    int Hq, int Hkv, int D, int n_slots, int max_len, int chunk, float scale, void* stream) {
  SVLM_CHECK_ARG((k_lin == nullptr) == (v_lin == nullptr) && (k_lin == nullptr) == (lin_len_dev == nullptr),
                 "svlm_decode_attn_lin: k_lin, v_lin and lin_len_dev come together");
  SVLM_CHECK_ARG(k_lin == nullptr || (lin_rows > 0 && lin_rows % 16 == 0 && lin_rows >= max_len),
                 "svlm_decode_attn_lin: lin_rows=%d must be a multiple of 16 and >= max_len=%d", lin_rows, max_len);
  SVLM_CHECK_ARG(D == DA_D, "svlm_decode_attn_ropeload: head_dim %d unsupported (128 only)", D);
  SVLM_CHECK_ARG(Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= DA_GMAX, "svlm_decode_attn_ropeload: Hq=%d Hkv=%d (group must be <= %d)", Hq, Hkv, DA_GMAX);
  SVLM_CHECK_ARG(chunk > 0 && max_len > 0 && n_slots > 0 &&
                     ((chunk % 16 == 0 && chunk <= 16 * DA_MAX_STEPS) || (chunk % 64 == 0 && chunk <= DA_LONG_MAX)),
                 "svlm_decode_attn_ropeload: chunk=%d must be a multiple of 16 up to %d or a multiple of 64 up to %d", chunk,
                 16 * DA_MAX_STEPS, DA_LONG_MAX);
  SVLM_CHECK_ARG(len_dev != nullptr || (len_add > 0 && len_add <= max_len), "svlm_decode_attn_ropeload: length %d outside (0, %d]", len_add, max_len);
  const int ns = (max_len + chunk - 1) / chunk;
  float* ws_m = (float*)ws;
  float* ws_l = ws_m + (size_t)ns * Hq;
  float* ws_acc = ws_l + (size_t)ns * Hq;
  dim3 grid(ns, Hkv);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *qq = (const bf16_t*)q, *kp = (const bf16_t*)k_planes, *vp = (const bf16_t*)v_planes, *cs = (const bf16_t*)rope_cs;
#define SVLM_DA_CASE(GG) \
  case GG: launch_split<GG>(grid, st, qq, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len, (const bf16_t*)k_lin, (const bf16_t*)v_lin, lin_rows, lin_len_dev); break;
  switch (Hq / Hkv) {
    SVLM_DA_CASE(1) SVLM_DA_CASE(2) SVLM_DA_CASE(3) SVLM_DA_CASE(4) SVLM_DA_CASE(5) SVLM_DA_CASE(6) SVLM_DA_CASE(7) SVLM_DA_CASE(8)
  }
#undef SVLM_DA_CASE
  int rc = svlm_check_launch("svlm_decode_attn_ropeload(split)");
  if (rc) return rc;
  static const int force_ds = svlm_env("SVLM_DA_COMBINE_DS") ? atoi(svlm_env("SVLM_DA_COMBINE_DS")) : 0;
  if (force_ds < 0) return SVLM_OK;           // diagnostic build only: time the split kernel alone
  const int ns_max = (max_len + chunk - 1) / chunk;
  // measured on MI355X (tools/decode_attn_sweep.py): column halves pay from ~64 splits (7B @ window 4096: 13.8 -> 12.3 us),
  // column quarters on 16 waves from ~200 (32k keys: 26.3 -> 22.8 us); below that the extra workgroups only add latency
  if (force_ds == 4 || (force_ds == 0 && ns_max > 192)) {
    decode_attn_combine_kernel<16, 4><<<dim3(Hq, 4), 1024, 0, st>>>(ws_m, ws_l, ws_acc, len_dev, len_add, (bf16_t*)out, Hq, chunk, ns_max);
  } else if (force_ds == 2 || (force_ds == 0 && ns_max > 64)) {
    decode_attn_combine_kernel<4, 2><<<dim3(Hq, 2), 256, 0, st>>>(ws_m, ws_l, ws_acc, len_dev, len_add, (bf16_t*)out, Hq, chunk, ns_max);
  } else {                     // bounded windows: up to two batches per wave on 4 waves
    decode_attn_combine_kernel<4, 1><<<Hq, 256, 0, st>>>(ws_m, ws_l, ws_acc, len_dev, len_add, (bf16_t*)out, Hq, chunk, ns_max);
  }
  return svlm_check_launch("svlm_decode_attn_ropeload(combine)");
}

extern "C" int svlm_decode_attn_ropeload(const void* q, const void* k_planes, const void* v_planes, const int* slot_of,
                                         const void* rope_cs, const int* len_dev, int len_add, void* out, void* ws,
                                         int Hq, int Hkv, int D, int n_slots, int max_len, int chunk, float scale, void* stream) {
  return svlm_decode_attn_lin(q, k_planes, v_planes, slot_of, rope_cs, len_dev, len_add, nullptr, nullptr, 0, nullptr, out, ws, Hq, Hkv, D,
                              n_slots, max_len, chunk, scale, stream);
}
