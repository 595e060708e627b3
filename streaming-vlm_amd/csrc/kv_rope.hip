// KV-pool data movement and the M-RoPE cos/sin table.
//
// Pool layout (one allocation per stream, owned by the caller):
//     pool[layer][kv(0=K,1=V)][Hkv][n_slots][D]   bf16, keys stored UN-ROTATED
// A "plane" is one [n_slots][D] array.  `slot_of[i]` maps logical token i to its slot, so
// eviction (reference: torch.index_select into fresh tensors, inference.py:54-59) and the
// assistant-text move (torch.cat of 4 slices, inference.py:106-107) are edits of `slot_of`;
// bytes only move on append and on defragmentation.
#include "common.h"

// ---------------------------------------------------------------- M-RoPE table
// Row i: [cos(f=0..half) | sin(f=0..half)] in bf16, f-th frequency driven by the axis the
// mrope section assigns to f (t: [0,sec_t), h: [sec_t, sec_t+sec_h), w: rest).  Restates
// Qwen2VLRotaryEmbedding.forward (fp32 angle = pos * inv_freq, cos/sin cast to the activation
// dtype) + the section select of apply_multimodal_rotary_pos_emb (qwen2/language_forward.py:43-60,271).
// inv_freq comes from the host so that it is bit-identical to torch's table.
__global__ void mrope_table_kernel(const int* __restrict__ pos3, int pos_stride, const float* __restrict__ posf3,
                                   const float* __restrict__ inv_freq, bf16_t* __restrict__ cs, int start, int count,
                                   int half, int sec_t, int sec_h) {
  const int total = count * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = start + i / half, f = i % half;
    const int axis = f < sec_t ? 0 : (f < sec_t + sec_h ? 1 : 2);
    const float p = posf3 ? posf3[(size_t)axis * pos_stride + r] : (float)pos3[(size_t)axis * pos_stride + r];
    const float ang = p * inv_freq[f];
    cs[(size_t)r * 2 * half + f] = f2bf(cosf(ang));
    cs[(size_t)r * 2 * half + half + f] = f2bf(sinf(ang));
  }
}

extern "C" int svlm_mrope_table(const int* pos3, const float* posf3, int pos_stride, const float* inv_freq, void* cs,
                                int start, int count, int head_dim, int sec_t, int sec_h, int sec_w, void* stream) {
  SVLM_CHECK_ARG((pos3 != nullptr) != (posf3 != nullptr), "svlm_mrope_table: pass exactly one of pos3 / posf3");
  SVLM_CHECK_ARG(head_dim > 0 && head_dim % 2 == 0 && sec_t + sec_h + sec_w == head_dim / 2,
                 "svlm_mrope_table: mrope sections %d+%d+%d != head_dim/2=%d", sec_t, sec_h, sec_w, head_dim / 2);
  SVLM_CHECK_ARG(start >= 0 && count >= 0 && start + count <= pos_stride, "svlm_mrope_table: rows [%d,%d) exceed stride %d", start, start + count, pos_stride);
  if (count == 0) return SVLM_OK;
  const int total = count * (head_dim / 2);
  int grid = (total + 255) / 256;
  grid = grid > 2048 ? 2048 : grid;
  mrope_table_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(pos3, pos_stride, posf3, inv_freq, (bf16_t*)cs, start, count, head_dim / 2, sec_t, sec_h);
  return svlm_check_launch("svlm_mrope_table");
}

// ---------------------------------------------------------------- append
// Writes T new un-rotated K rows and V rows (reference: StreamingCache.update = torch.cat,
// generate/streaming_cache.py:72-73) of one layer to the slots of logical rows base..base+T-1.
// base = *len_dev (device scalar, for graph replay) or `start`.
__global__ void kv_append_kernel(const bf16_t* __restrict__ k_new, int k_stride, const bf16_t* __restrict__ v_new, int v_stride,
                                 bf16_t* __restrict__ k_plane, bf16_t* __restrict__ v_plane, const int* __restrict__ slot_of,
                                 const int* __restrict__ len_dev, int start, int T, int Hkv, int D, int n_slots) {
  const int base = len_dev ? *len_dev : start;
  const int cpr = D / 8;                      // 16-B chunks per (token, head)
  const int total = T * Hkv * cpr;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % cpr, h = (i / cpr) % Hkv, t = i / (cpr * Hkv);
    const int slot = slot_of[base + t];
    const size_t dst = ((size_t)h * n_slots + slot) * D + c * 8;
    *reinterpret_cast<u32x4_t*>(k_plane + dst) = *reinterpret_cast<const u32x4_t*>(k_new + (size_t)t * k_stride + h * D + c * 8);
    *reinterpret_cast<u32x4_t*>(v_plane + dst) = *reinterpret_cast<const u32x4_t*>(v_new + (size_t)t * v_stride + h * D + c * 8);
  }
}

extern "C" int svlm_kv_append(const void* k_new, int k_stride, const void* v_new, int v_stride, void* k_planes, void* v_planes,
                              const int* slot_of, const int* len_dev, int start, int T, int Hkv, int D, int n_slots, void* stream) {
  SVLM_CHECK_ARG(T >= 0 && Hkv > 0 && D > 0 && D % 8 == 0 && n_slots > 0, "svlm_kv_append: bad shape T=%d Hkv=%d D=%d n_slots=%d", T, Hkv, D, n_slots);
  SVLM_CHECK_ARG(k_stride % 8 == 0 && v_stride % 8 == 0 && k_stride >= Hkv * D && v_stride >= Hkv * D, "svlm_kv_append: bad strides %d %d", k_stride, v_stride);
  if (T == 0) return SVLM_OK;
  const int total = T * Hkv * (D / 8);
  int grid = (total + 255) / 256;
  grid = grid > 2048 ? 2048 : grid;
  kv_append_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)k_new, k_stride, (const bf16_t*)v_new, v_stride, (bf16_t*)k_planes,
                                                          (bf16_t*)v_planes, slot_of, len_dev, start, T, Hkv, D, n_slots);
  return svlm_check_launch("svlm_kv_append");
}

// ---------------------------------------------------------------- in-place row moves (defragmentation)
// For every plane of the pool: row src[i] -> row dst[i].  Destinations must be free slots that
// are not sources of this call (the host allocator guarantees it), so all moves are independent.
__global__ void kv_move_rows_kernel(bf16_t* __restrict__ pool, long long n_planes, int n_slots, int D,
                                    const int* __restrict__ src, const int* __restrict__ dst, int n) {
  const int cpr = D / 8;
  const long long total = n_planes * n * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpr);
    const int m = (int)((i / cpr) % n);
    const long long p = i / ((long long)cpr * n);
    bf16_t* plane = pool + (size_t)p * n_slots * D;
    *reinterpret_cast<u32x4_t*>(plane + (size_t)dst[m] * D + c * 8) = *reinterpret_cast<const u32x4_t*>(plane + (size_t)src[m] * D + c * 8);
  }
}

extern "C" int svlm_kv_move_rows(void* pool, long long n_planes, int n_slots, int D, const int* src, const int* dst, int n, void* stream) {
  SVLM_CHECK_ARG(n_planes > 0 && n_slots > 0 && D > 0 && D % 8 == 0 && n >= 0, "svlm_kv_move_rows: bad shape planes=%lld n_slots=%d D=%d n=%d", n_planes, n_slots, D, n);
  if (n == 0) return SVLM_OK;
  const long long total = n_planes * n * (D / 8);
  long long grid = (total + 255) / 256;
  grid = grid > 8192 ? 8192 : grid;
  kv_move_rows_kernel<<<(int)grid, 256, 0, (hipStream_t)stream>>>((bf16_t*)pool, n_planes, n_slots, D, src, dst, n);
  return svlm_check_launch("svlm_kv_move_rows");
}

// ---------------------------------------------------------------- gather to the reference's dense layout
// out[h][i][:] = plane[h][slot_of[i]][:], i < L.  Used by the StreamingCache-compatible views
// ((1, Hkv, L, D) tensors, inference.py:54-55) and by tests; not on the hot path.
__global__ void kv_gather_kernel(const bf16_t* __restrict__ planes, const int* __restrict__ slot_of, bf16_t* __restrict__ out,
                                 int L, int Hkv, int D, int n_slots) {
  const int cpr = D / 8;
  const long long total = (long long)Hkv * L * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpr);
    const int t = (int)((i / cpr) % L);
    const int h = (int)(i / ((long long)cpr * L));
    *reinterpret_cast<u32x4_t*>(out + ((size_t)h * L + t) * D + c * 8) =
        *reinterpret_cast<const u32x4_t*>(planes + ((size_t)h * n_slots + slot_of[t]) * D + c * 8);
  }
}

extern "C" int svlm_kv_gather(const void* planes, const int* slot_of, void* out, int L, int Hkv, int D, int n_slots, void* stream) {
  SVLM_CHECK_ARG(L >= 0 && Hkv > 0 && D > 0 && D % 8 == 0 && n_slots > 0, "svlm_kv_gather: bad shape");
  if (L == 0) return SVLM_OK;
  const long long total = (long long)Hkv * L * (D / 8);
  long long grid = (total + 255) / 256;
  grid = grid > 4096 ? 4096 : grid;
  kv_gather_kernel<<<(int)grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)planes, slot_of, (bf16_t*)out, L, Hkv, D, n_slots);
  return svlm_check_launch("svlm_kv_gather");
}
