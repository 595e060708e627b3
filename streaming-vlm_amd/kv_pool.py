"""Paged, slot-mapped KV pool with in-place eviction -- the replacement for the reference's
``StreamingCache`` (generate/streaming_cache.py:6-74: a list of per-layer tensors re-allocated by
``torch.cat`` on every step) and for ``prune_id_and_kv_cache`` / ``resort_id_and_kv`` /
``contiguous_id_and_kv`` (inference.py:50-68,100-108: index_select / cat into fresh tensors).

Device layout: ``pool[layer][kv][Hkv][n_slots][D]`` bf16 (keys un-rotated) sized once for the
bounded stream; ``slot_of[i]`` (int32) maps logical token i to its slot.  Slots are handed out in
pages of ``page_tokens`` consecutive rows so a chunk's rows stay contiguous; eviction and the
assistant-text move edit ``slot_of`` on the host (O(L) int32) and free whole pages when their
last live row goes; bytes move only on append and when ``defragment`` packs sparse pages.

Beside the pool the cache holds its LINEAR PLANES ``lin[layer][kv][Hkv][lin_rows][D]``: the rotated keys (in the decode kernels'
operand layout) and the values of rows ``[0, lin_valid)`` in logical order, as the prefill of the current chunk gathered them
(``svlm_prefill_attn_ropeload_lin``).  The reference rotates every cached key in every forward (qwen2/language_forward.py:55-63);
the positions of cached rows only change when the host edits the logical order, so the decode steps of a chunk stream that copy
(``svlm_decode_attn_lin``); the rows appended since are written there too by the decode step's QKV launch (``svlm_dec_qkv_lin``, keys
un-rotated: the attention rotates those few while it stages them), so a decode step reads no slot table at all.  The state the kernels
see is ``lin_len_dev = {rows rotated, appended rows follow}``; every edit of the logical order lowers the first and clears the second.
"""
from __future__ import annotations

from collections import deque
from typing import List, Optional

import numpy as np
import torch


class _LayerPlanes:
    """`past_key_values.key_cache` / `.value_cache` of the reference's cache object (a list of per-layer
    (1, Hkv, L, D) tensors, generate/streaming_cache.py:6-28): reads gather the layer in logical order, assignment
    (inference.py:58-59 writes the pruned tensors back this way) stores the rows through the slot table."""

    def __init__(self, pool: "KVPool", which: int):
        self._pool, self._which = pool, which

    def __len__(self):
        return self._pool.n_layers

    def __getitem__(self, layer: int):
        return self._pool.layer_kv(layer)[self._which]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __setitem__(self, layer: int, rows):
        self._pool._assign_plane(layer, self._which, rows)


class KVPool:
    _next_serial = 0          # pools are numbered: a captured decode graph is keyed by the serial of the pool whose buffers it holds

    def __init__(self, n_layers: int, n_kv_heads: int, head_dim: int, max_len: int, device, ops, page_tokens: int = 16,
                 slack: float = 1.0, linear_planes: bool = True):
        """max_len: the largest logical length the stream ever reaches (sink + window + one chunk).
        linear_planes=False: no rotated copy (saves max_len rows of K and V per layer; every decode step then rotates the pool rows)."""
        self.n_layers, self.Hkv, self.D = n_layers, n_kv_heads, head_dim
        self.P = page_tokens
        self.max_len = int(max_len)
        n_pages = int(np.ceil(max_len * (1.0 + slack) / page_tokens)) + 2
        self.n_pages = n_pages
        self.n_slots = n_pages * page_tokens
        self.device = device
        self.ops = ops
        self.pool = torch.zeros((n_layers, 2, n_kv_heads, self.n_slots, head_dim), dtype=torch.bfloat16, device=device)
        self.slot_of = np.zeros(self.max_len, dtype=np.int32)         # logical -> slot
        self.slot_of_dev = torch.zeros(self.max_len, dtype=torch.int32, device=device)
        self.length = 0           # rows holding data (== past_key_values.get_seq_length())
        self.reserved = 0         # rows with a slot assigned (>= length)
        self.page_live = np.zeros(n_pages, dtype=np.int32)
        self.free_pages = deque(range(n_pages))
        self.open_page = -1
        self.open_fill = page_tokens
        self._dirty_from = 0
        dev_is_gpu = torch.device(device).type == "cuda"
        self._stage = [torch.empty(self.max_len + 2, dtype=torch.int32).pin_memory() for _ in range(4)] if dev_is_gpu else None
        self._stage_ev = [None] * 4       # the event behind each buffer's last copy: waited for before the buffer is written again
        self._stage_i = 0
        self.stats = dict(moved_rows=0, defrags=0, evicted_rows=0)
        KVPool._next_serial += 1
        self.serial = KVPool._next_serial
        self.key_cache, self.value_cache = _LayerPlanes(self, 0), _LayerPlanes(self, 1)
        # pos_mode="append": the M-RoPE position of every cached row, edited in lockstep with slot_of (float64 holds the
        # int positions of Qwen2-VL and the float32 ones of Qwen2.5-VL exactly)
        self.pos_rows = np.zeros((3, self.max_len), dtype=np.float64)
        # linear planes (module docstring): rows never read above lin_valid, so no initialisation
        self.lin_rows = -(-self.max_len // 16) * 16
        self.lin = torch.empty((n_layers, 2, n_kv_heads, self.lin_rows, head_dim), dtype=torch.bfloat16, device=device) if linear_planes else None
        self.lin_len_dev = torch.zeros(2, dtype=torch.int32, device=device)      # {rows rotated, appended rows follow}
        self.lin_valid = 0        # host mirror of lin_len_dev[0]
        self.lin_fresh = False    # host mirror of lin_len_dev[1]: rows [lin_valid, length) are in the planes too (keys un-rotated)
        self._lin_dirty = False   # lowered / cleared on the host since the device copy was written
        self._half = {}           # layer -> (which, rows): one plane assigned, waiting for its partner
        self._upd_rows = 0        # rows of an update() pass that has not reached the last layer yet

    # ------------------------------------------------------------------ linear planes
    def lin_args(self):
        """(planes, lin_len_dev) for ops.prefill_attn / ops.decode_attn, or None without linear planes."""
        return None if self.lin is None else (self.lin, self.lin_len_dev)

    def lin_written(self, L: int):
        """The prefill's gather launches have left rows [0, L) of every layer in the linear planes (and L in *lin_len_dev)."""
        if self.lin is not None:
            self.lin_valid, self.lin_fresh, self._lin_dirty = int(L), True, False

    def _lin_touch(self, first_row: int):
        """Logical rows >= first_row no longer are what the linear planes hold (nor do rows appended from now on go there in order)."""
        if self.lin is not None and (first_row < self.lin_valid or self.lin_fresh):
            self.lin_valid, self.lin_fresh, self._lin_dirty = max(min(int(first_row), self.lin_valid), 0), False, True

    def lin_appends_off(self):
        """The decode steps that follow do not write their rows into the linear planes (svlm_dec_tail's QKV): rows above lin_valid
        are read from the pool."""
        self._lin_touch(self.lin_valid)

    # ------------------------------------------------------------------ allocation
    def _take_slot(self) -> int:
        if self.open_fill >= self.P:
            if not self.free_pages:
                raise MemoryError("KV pool out of pages")
            self.open_page = self.free_pages.popleft()
            self.open_fill = 0
        s = self.open_page * self.P + self.open_fill
        self.open_fill += 1
        self.page_live[self.open_page] += 1
        return s

    def _free_slot(self, s: int):
        pg = s // self.P
        self.page_live[pg] -= 1
        if self.page_live[pg] == 0 and pg != self.open_page:
            self.free_pages.append(pg)
        elif self.page_live[pg] == 0 and pg == self.open_page:
            # nothing live on the open page: restart it from row 0 instead of leaking its tail
            self.open_fill = 0

    def free_slots_available(self) -> int:
        return len(self.free_pages) * self.P + (self.P - self.open_fill)

    def reserve(self, n: int):
        """Assign slots to logical rows [reserved, reserved + n)."""
        if self.reserved + n > self.max_len:
            raise MemoryError(f"KV length {self.reserved + n} exceeds the pool's max_len {self.max_len}")
        if self.free_slots_available() < n:
            self.defragment()
            if self.free_slots_available() < n:
                raise MemoryError("KV pool out of slots even after defragmentation")
        self._dirty_from = min(self._dirty_from, self.reserved)
        for i in range(n):
            self.slot_of[self.reserved + i] = self._take_slot()
        self.reserved += n

    def commit(self, new_length: int):
        assert new_length <= self.reserved
        self.length = new_length

    def release_reserved(self):
        """Give back rows reserved beyond `length` (unused decode slots after EOS / end of chunk)."""
        for i in range(self.length, self.reserved):
            self._free_slot(int(self.slot_of[i]))
        self.reserved = self.length

    # ------------------------------------------------------------------ structural edits (no data movement)
    def prune(self, start: int, end: int):
        """Delete the CLOSED logical interval [start, end] (reference: prune_id_and_kv_cache)."""
        assert self.reserved == self.length, "release_reserved() before editing the cache"
        assert 0 <= start <= end < self.length, (start, end, self.length)
        for s in self.slot_of[start:end + 1]:
            self._free_slot(int(s))
        n = end - start + 1
        self.slot_of[start:self.length - n] = self.slot_of[end + 1:self.length].copy()
        self.pos_rows[:, start:self.length - n] = self.pos_rows[:, end + 1:self.length].copy()
        self.length -= n
        self.reserved = self.length
        self._dirty_from = min(self._dirty_from, start)
        self._lin_touch(start)
        self.stats["evicted_rows"] += n

    def move(self, src_s: int, src_e: int, dst: int):
        """Move rows [src_s, src_e] to directly after row dst (reference: resort_id_and_kv)."""
        assert self.reserved == self.length
        assert dst < src_s <= src_e < self.length
        so = self.slot_of
        seg = so[src_s:src_e + 1].copy()
        mid = so[dst + 1:src_s].copy()
        so[dst + 1:dst + 1 + seg.size] = seg
        so[dst + 1 + seg.size:src_e + 1] = mid
        pr = self.pos_rows
        pseg, pmid = pr[:, src_s:src_e + 1].copy(), pr[:, dst + 1:src_s].copy()
        pr[:, dst + 1:dst + 1 + seg.size] = pseg
        pr[:, dst + 1 + seg.size:src_e + 1] = pmid
        self._dirty_from = min(self._dirty_from, dst + 1)
        self._lin_touch(dst + 1)

    def truncate(self, new_length: int):
        assert self.reserved == self.length and 0 <= new_length <= self.length
        if new_length < self.length:
            self.prune(new_length, self.length - 1)

    # ------------------------------------------------------------------ device sync / defragmentation
    def sync_device(self):
        """Upload the changed tail of slot_of (a few KB, once per chunk, stream-ordered)."""
        lo, hi = self._dirty_from, self.reserved
        if lo < hi:
            if self._stage is not None:
                # page-locked staging, asynchronous copy: a copy from pageable memory makes the host wait for everything enqueued in
                # front of it (the look-ahead ViT's tail that is meant to run underneath the host's turnaround).  Four buffers in turn:
                # a generate() call syncs the table at most a few times and ends with a host sync, so a buffer's last copy has been read
                st = self._stage_next()[:hi - lo]
                st.copy_(torch.from_numpy(self.slot_of[lo:hi]))
                self.slot_of_dev[lo:hi].copy_(st, non_blocking=True)
                self._stage_done()
            else:
                self.slot_of_dev[lo:hi].copy_(torch.from_numpy(self.slot_of[lo:hi].copy()))
        self._dirty_from = self.max_len
        if self._lin_dirty:
            if self._stage is not None:
                st = self._stage_next()[:2]
                st[0], st[1] = self.lin_valid, int(self.lin_fresh)
                self.lin_len_dev.copy_(st, non_blocking=True)
                self._stage_done()
            else:
                self.lin_len_dev.copy_(torch.tensor([self.lin_valid, int(self.lin_fresh)], dtype=torch.int32))
            self._lin_dirty = False

    def _stage_next(self):
        self._stage_i = (self._stage_i + 1) % len(self._stage)
        ev = self._stage_ev[self._stage_i]
        if ev is not None:
            ev.synchronize()              # (long done in the streaming loop: four syncs and a host sync on the tokens ago)
        return self._stage[self._stage_i]

    def _stage_done(self):
        ev = torch.cuda.Event()
        ev.record()
        self._stage_ev[self._stage_i] = ev

    def fragmentation(self) -> float:
        """Fraction of slots of non-free pages that hold no live row."""
        used_pages = self.n_pages - len(self.free_pages)
        if used_pages == 0:
            return 0.0
        return 1.0 - float(self.reserved) / float(used_pages * self.P)

    def defragment(self):
        """Pack the live rows of the sparsest pages into fresh pages, in place (svlm_kv_move_rows).
        Sources and destinations are disjoint slot sets, so one launch moves every layer's rows."""
        assert self.reserved == self.length
        if self.length == 0:
            return 0
        so = self.slot_of[:self.length]
        pages = so // self.P
        live = self.page_live.copy()
        # candidates: closed pages that are at most half full, sparsest first
        cand = [int(p) for p in np.argsort(live) if 0 < live[p] <= self.P // 2 and p != self.open_page]
        src_idx: List[int] = []
        budget = self.free_slots_available()
        for p in cand:
            rows = np.flatnonzero(pages == p)
            if len(src_idx) + rows.size > budget:
                break
            src_idx.extend(int(r) for r in rows)
        if not src_idx:
            return 0
        src_idx.sort()
        src = so[src_idx].copy()
        dst = np.empty_like(src)
        freed_after = []
        for k, li in enumerate(src_idx):
            dst[k] = self._take_slot()
            so[li] = dst[k]
        for s in src:
            freed_after.append(int(s))
        # launch the move BEFORE recycling the source pages
        self.ops.kv_move_rows(self.pool, torch.from_numpy(src).to(self.device), torch.from_numpy(dst).to(self.device))
        for s in freed_after:
            self._free_slot(s)
        self._dirty_from = 0
        self.stats["moved_rows"] += len(src_idx)
        self.stats["defrags"] += 1
        return len(src_idx)

    # ------------------------------------------------------------------ reference-compatible views
    def get_seq_length(self) -> int:
        return self.length

    def _rows2d(self, t):
        """(1, Hkv, n, D) or (Hkv, n, D) -> (n, Hkv*D) bf16 rows on the pool's device."""
        t = t.reshape(self.Hkv, -1, self.D) if t.dim() == 4 else t
        if t.dim() != 3 or t.shape[0] != self.Hkv or t.shape[2] != self.D:
            raise ValueError(f"expected (1, {self.Hkv}, L, {self.D}) rows, got {tuple(t.shape)}")
        return t.to(device=self.device, dtype=torch.bfloat16).permute(1, 0, 2).reshape(t.shape[1], self.Hkv * self.D).contiguous()

    def _assign_plane(self, layer: int, which: int, rows):
        """`cache.key_cache[i] = k` / `cache.value_cache[i] = v`: a layer's K and V are written together once both
        halves are there (svlm_kv_append stores a K row and a V row per slot).  The first assignment whose length
        differs from the cache's re-sizes the logical sequence (drops or adds TAIL rows): the reference always
        rewrites every layer in full, so which physical rows survive is immaterial."""
        if not 0 <= layer < self.n_layers:
            raise IndexError(layer)
        r = self._rows2d(rows)
        other = self._half.pop(layer, None)
        if other is None:
            self._half[layer] = (which, r)
            return
        if other[0] == which or other[1].shape[0] != r.shape[0]:
            raise ValueError(f"layer {layer}: key_cache[i] and value_cache[i] must be assigned as a pair of equal length")
        k, v = (other[1], r) if which == 1 else (r, other[1])
        n = k.shape[0]
        self._lin_touch(0)        # the layer's rows are rewritten behind the linear planes' back
        if n != self.length:
            self.release_reserved()
            if n < self.length:
                self.truncate(n)
            else:
                self.reserve(n - self.length)
                self.commit(n)
        if n:
            self.sync_device()
            self.ops.kv_append(k, v, self.pool, layer, self.slot_of_dev, 0, n)

    def update(self, key_states, value_states, layer_idx: int, cache_kwargs=None):
        """`StreamingCache.update` (generate/streaming_cache.py:30-74): append T new rows to layer `layer_idx` and
        return that layer's full (1, Hkv, L + T, D) K and V.  Slots are taken when layer 0 arrives, the logical
        length grows when the last layer has been written."""
        k, v = self._rows2d(key_states), self._rows2d(value_states)
        T = k.shape[0]
        if layer_idx == 0:
            if self._upd_rows:
                raise RuntimeError("update(): the previous pass did not reach the last layer")
            self._lin_touch(self.length)      # rows appended here exist in the pool only
            self.release_reserved()
            self.reserve(T)
            self.sync_device()
            self._upd_rows = T
        if T != self._upd_rows or v.shape[0] != T:
            raise ValueError(f"update(): layer {layer_idx} brings {T} rows, layer 0 brought {self._upd_rows}")
        self.ops.kv_append(k, v, self.pool, layer_idx, self.slot_of_dev, self.length, T)
        n = self.length + T
        out = tuple(self.ops.kv_gather(self.pool, layer_idx, w, self.slot_of_dev, n).unsqueeze(0) for w in (0, 1))
        if layer_idx == self.n_layers - 1:
            self.commit(n)
            self._upd_rows = 0
        return out

    def layer_kv(self, layer: int):
        """Dense (1, Hkv, L, D) K and V in logical order, like the tensors the reference's cache holds."""
        if self._half:
            raise RuntimeError(f"layers {sorted(self._half)}: only one of key_cache[i] / value_cache[i] has been assigned")
        self.sync_device()
        k = self.ops.kv_gather(self.pool, layer, 0, self.slot_of_dev, self.length)
        v = self.ops.kv_gather(self.pool, layer, 1, self.slot_of_dev, self.length)
        return k.unsqueeze(0), v.unsqueeze(0)

    def __iter__(self):
        for i in range(self.n_layers):
            yield self.layer_kv(i)

    def __len__(self):
        return self.n_layers
