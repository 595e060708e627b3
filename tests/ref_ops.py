"""TEST-ONLY ops backend: the ``ops.HipOps`` interface implemented with eager CPU torch (oracle math).

It exists so that the HOST logic of the product (KV pool / slot table, eviction policies, position
ids, prompt building, generate loop, device-state feedback) can be exercised on a machine without a
GPU.  It is never importable from the product package; the product has no CPU path.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from oracle import model as om

BF16 = torch.bfloat16


def _act(y, act):
    if act == 1:
        return om.quick_gelu(y)
    if act == 2:
        return F.gelu(y)
    if act == 3:
        return F.silu(y)
    return y


class RefOps:
    name = "ref-cpu"

    def gemm(self, A, W, bias=None, residual=None, out=None, act=0):
        if act == 4:          # ACT_SWIGLU: W = [gate; up]
            gu = F.linear(A, W, bias)
            I = gu.shape[1] // 2
            y = F.silu(gu[:, :I]) * gu[:, I:]
            if out is None:
                return y
            out.copy_(y)
            return out
        y = _act(F.linear(A, W, bias), act)
        if residual is not None:
            y = residual + y
        if out is None:
            return y
        out.copy_(y)
        return out

    def gemm_norm(self, A, W, norm_w, eps, out, out_norm, bias=None, residual=None, act=0, norm_b=None):
        self.gemm(A, W, bias=bias, residual=residual, out=out, act=act)
        if norm_b is None:
            self.rmsnorm(out, norm_w, eps, out=out_norm)
        else:
            self.layernorm(out, norm_w, norm_b, eps, out=out_norm)
        return out, out_norm

    def gemv(self, x, W, bias=None, residual=None, out=None, out_f32=None, act=0):
        y = _act(F.linear(x.reshape(1, -1), W, bias)[0], act)
        if residual is not None:
            y = residual.reshape(-1) + y
        if out is not None:
            out.reshape(-1).copy_(y)
        if out_f32 is not None:
            out_f32.copy_(y.float())
        return out if out is not None else out_f32

    def prefetch(self, t, n_wgs=256):
        pass

    def rmsnorm(self, x, w, eps, out=None):
        y = om.rms_norm(x, w, eps)
        if out is None:
            return y
        out.copy_(y.reshape(out.shape))
        return out

    def layernorm(self, x, w, b, eps, out=None):
        y = F.layer_norm(x, (x.shape[-1],), w, b, eps)
        if out is None:
            return y
        out.copy_(y)
        return out

    def add(self, a, b, out=None):
        y = a + b
        if out is None:
            return y
        out.copy_(y)
        return out

    def silu_mul(self, gu, out=None):
        I = gu.shape[1] // 2
        y = F.silu(gu[:, :I]) * gu[:, I:]
        if out is None:
            return y
        out.copy_(y)
        return out

    def patchify_u8(self, frames, patch=14, temporal=2, merge=2, out=None):
        from streaming_vlm_amd.synthetic import patchify
        pix, grid = patchify(frames, patch, temporal, merge)
        pix = pix.to(torch.bfloat16)
        if out is not None:
            out.copy_(pix)
            pix = out
        return pix, grid

    def gather_rows(self, table, alt, idx, out, idx_off=None):
        off = int(idx_off[0]) if idx_off is not None else 0
        rows = out.shape[0]
        for r in range(rows):
            i = int(idx[off + r])
            out[r] = table[i] if i >= 0 else alt[-1 - i]
        return out

    def vit_rope(self, qkv, cosT, sinT, H, d):
        N = qkv.shape[0]
        v = qkv.view(N, 3, H, d)
        cos = torch.cat([cosT, cosT], -1).unsqueeze(1)
        sin = torch.cat([sinT, sinT], -1).unsqueeze(1)
        for which in (0, 1):
            x = v[:, which].float()
            v[:, which] = (x * cos + om.rotate_half(x) * sin).to(qkv.dtype)
        return qkv

    def vit_attn(self, qkv, n_seq, seq_len, H, d, scale, out=None):
        N = qkv.shape[0]
        v = qkv.view(N, 3, H, d)
        res = torch.empty((N, H * d), dtype=qkv.dtype)
        for s in range(n_seq):
            sl = slice(s * seq_len, (s + 1) * seq_len)
            q, k, vv = (v[sl, j].transpose(0, 1) for j in range(3))
            res[sl] = om.flash_attention(q, k, vv, None, scale).transpose(0, 1).reshape(seq_len, H * d)
        if out is None:
            return res
        out.copy_(res)
        return out

    def mrope_table(self, pos3, inv_freq, rope_cs, start, count, sections):
        D = rope_cs.shape[1]
        half = D // 2
        p = pos3[:, start:start + count].float()
        axis = torch.tensor([0] * sections[0] + [1] * sections[1] + [2] * sections[2])
        ang = p[axis, :].t() * inv_freq.unsqueeze(0)          # (count, half)
        rope_cs[start:start + count, :half] = ang.cos().to(BF16)
        rope_cs[start:start + count, half:] = ang.sin().to(BF16)

    def kv_append(self, k_new, v_new, pool, layer, slot_of, start, T, len_dev=None):
        base = int(len_dev[0]) if len_dev is not None else start
        _, _, Hkv, n_slots, D = pool.shape
        for t in range(T):
            s = int(slot_of[base + t])
            pool[layer, 0, :, s] = k_new[t].view(Hkv, D)
            pool[layer, 1, :, s] = v_new[t].view(Hkv, D)

    def kv_move_rows(self, pool, src, dst):
        pool[:, :, :, dst.long()] = pool[:, :, :, src.long()]

    def kv_gather(self, pool, layer, which, slot_of, L):
        return pool[layer, which][:, slot_of[:L].long()].contiguous()

    def decode_attn_ws(self, Hq, max_len, chunk, device):
        return torch.zeros(1, dtype=torch.float32)

    def _rot(self, x, rope_cs, rows):
        """x (H, n, D) un-rotated, rope rows (n,) -> eager bf16 rope (x*cos + rotate_half(x)*sin)."""
        half = rope_cs.shape[1] // 2
        cs = rope_cs[rows]
        cos = torch.cat([cs[:, :half], cs[:, :half]], -1)
        sin = torch.cat([cs[:, half:], cs[:, half:]], -1)
        return om.apply_rope(x, cos, sin)

    def _attend(self, q, pool, layer, slot_of, rope_cs, T, L, Hq, scale):
        _, _, Hkv, n_slots, D = pool.shape
        sl = slot_of[:L].long()
        K = pool[layer, 0][:, sl]
        V = pool[layer, 1][:, sl]
        rows = torch.arange(L)
        Kr = self._rot(K, rope_cs, rows).repeat_interleave(Hq // Hkv, dim=0)
        Vr = V.repeat_interleave(Hq // Hkv, dim=0)
        qr = self._rot(q, rope_cs, rows[L - T:])
        return om.flash_attention(qr, Kr, Vr, L - T, scale)

    # ---- linear planes (include/svlm.h: svlm_decode_attn_lin): 16-key tiles of rotated keys, chunk c of key r at [c >> 2][(c & 3) * 16 + r][8]
    @staticmethod
    def lin_k_tiles(Kr, lin_rows):
        """Kr (Hkv, L, 128) rotated keys in logical order -> (Hkv, lin_rows, 128) storage of the K plane (rows >= L zero)."""
        Hkv, L, D = Kr.shape
        pad = torch.zeros((Hkv, lin_rows, D), dtype=Kr.dtype)
        pad[:, :L] = Kr
        t = pad.view(Hkv, lin_rows // 16, 16, 4, 4, 8)            # [h][tile][r][ks][fq][8]
        return t.permute(0, 1, 3, 4, 2, 5).reshape(Hkv, lin_rows, D).contiguous()       # [h][tile][ks][fq][r][8]

    @staticmethod
    def lin_k_rows(plane, L):
        """inverse of lin_k_tiles: the K plane's storage -> (Hkv, L, 128) rotated keys."""
        Hkv, lin_rows, D = plane.shape
        t = plane.view(Hkv, lin_rows // 16, 4, 4, 16, 8).permute(0, 1, 4, 2, 3, 5)
        return t.reshape(Hkv, lin_rows, D)[:, :L]

    def decode_attn(self, q, pool, layer, slot_of, rope_cs, out, ws, Hq, max_len, chunk, scale, length=0, len_dev=None, lin=None):
        L = (int(len_dev[0]) if len_dev is not None else 0) + length
        D = pool.shape[-1]
        if lin is not None:
            # the rows the kernel would stream from the linear planes must be what the pool path produces: this is the check of the
            # host's bookkeeping (every edit of the logical order has to lower *lin_len)
            planes, lin_len = lin
            n = min(int(lin_len[0]), L)
            if n:
                sl = slot_of[:n].long()
                Kr = self._rot(pool[layer, 0][:, sl], rope_cs, torch.arange(n))
                assert torch.equal(self.lin_k_rows(planes[layer, 0], n), Kr), f"stale rotated keys in the linear planes (layer {layer}, {n} rows)"
                assert torch.equal(planes[layer, 1][:, :n], pool[layer, 1][:, sl]), f"stale values in the linear planes (layer {layer}, {n} rows)"
            if int(lin_len[1]) and L > n:       # the appended rows the kernel would take from the planes (keys un-rotated)
                sl = slot_of[n:L].long()
                assert torch.equal(self.lin_k_rows(planes[layer, 0], L)[:, n:], pool[layer, 0][:, sl]), f"appended keys missing from the linear planes (layer {layer}, rows {n}..{L})"
                assert torch.equal(planes[layer, 1][:, n:L], pool[layer, 1][:, sl]), f"appended values missing from the linear planes (layer {layer}, rows {n}..{L})"
        o = self._attend(q.view(Hq, 1, D), pool, layer, slot_of, rope_cs, 1, L, Hq, scale)
        out.copy_(o.reshape(out.shape))
        return out

    def prefill_attn(self, q, pool, layer, slot_of, rope_cs, out, T, L, Hq, scale, k_new=None, v_new=None, lin=None):
        D = pool.shape[-1]
        if k_new is not None:
            self.kv_append(k_new, v_new, pool, layer, slot_of, L - T, T)
        if lin is not None:
            planes, lin_len = lin
            sl = slot_of[:L].long()
            planes[layer, 0] = self.lin_k_tiles(self._rot(pool[layer, 0][:, sl], rope_cs, torch.arange(L)), planes.shape[3])
            planes[layer, 1][:, :L] = pool[layer, 1][:, sl]
            lin_len[0], lin_len[1] = L, 1
        o = self._attend(q[:T].reshape(T, Hq, D).transpose(0, 1), pool, layer, slot_of, rope_cs, T, L, Hq, scale)
        out[:T] = o.transpose(0, 1).reshape(T, Hq * D)
        return out

    def quant_rows_fp8(self, x, q=None, scale=None):
        qv, s = om.quant_rows_fp8(x)
        qv, s = qv.to(torch.float8_e4m3fn), s.reshape(-1)
        if q is not None:
            q.copy_(qv); scale.copy_(s)
            return q, scale
        return qv, s

    def gemm_fp8(self, A8, a_scale, W8, w_scale, bias=None, residual=None, out=None, act=0, norm_w=None, norm_b=None, eps=1e-6, out_norm=None,
                 out_norm_q=None):
        y = (A8.float() @ W8.float().t()) * (a_scale.reshape(-1, 1) * w_scale.reshape(1, -1))
        if bias is not None:
            y = y + bias.float()
        y = _act(y.to(torch.bfloat16), act)
        if residual is not None:
            y = y + residual
        if out is None:
            out = torch.empty_like(y)
        out.copy_(y)
        if norm_w is not None:
            if norm_b is not None:
                out_norm.copy_(F.layer_norm(out, (out.shape[-1],), norm_w, norm_b, eps))
            else:
                out_norm.copy_(om.rms_norm(out, norm_w, eps))
            if out_norm_q is not None:
                self.quant_rows_fp8(out_norm, out_norm_q[0], out_norm_q[1])
        return out

    def mark_seen(self, ids, n, seen):
        seen[ids[:n].long()] = 1

    def sampling_ws(self, V, device):
        return torch.zeros(4, dtype=torch.float32)

    def dec_qkv(self, x, ln_w, eps, W, bias, q_out, pool, layer, slot_of, qd, kd, length=0, len_dev=None, lin=None):
        y = F.linear(om.rms_norm(x.reshape(1, -1), ln_w, eps), W, bias)[0]
        q_out[:qd] = y[:qd]
        self.kv_append(y[qd:qd + kd].reshape(1, -1), y[qd + kd:].reshape(1, -1), pool, layer, slot_of, length, 1, len_dev=len_dev)
        if lin is not None:          # the appended row also goes to the linear planes: V as it is, K un-rotated in its tile position
            planes = lin[0]
            row = int(len_dev[0]) if len_dev is not None else length
            Hkv, D = planes.shape[2], planes.shape[4]
            k = y[qd:qd + kd].reshape(Hkv, D)
            t, r = row // 16, row % 16
            planes[layer, 0].view(Hkv, -1, 4, 4, 16, 8)[:, t, :, :, r, :] = k.view(Hkv, 4, 4, 8)
            planes[layer, 1][:, row] = y[qd + kd:].reshape(Hkv, D)

    def dec_gate_up(self, x, ln_w, eps, W, h):
        y = F.linear(om.rms_norm(x.reshape(1, -1), ln_w, eps), W)
        I = W.shape[0] // 2
        h.copy_((F.silu(y[:, :I]) * y[:, I:])[0])

    # persistent layer tail (csrc/dec_tail.hip) as the four ops it fuses
    def dec_tail_ws(self, H, I, n_layers, device):
        return torch.zeros(32, dtype=torch.int64)

    def dec_tail_reset(self, ws, H, I, n_layers):
        pass

    def dec_tail(self, attn, x, o_w, ln2, gu_w, down_w, eps, ws, layer, n_layers, nxt=None, grid=0, stamps=None):
        self.gemv(attn, o_w, residual=x, out=x)
        h = torch.zeros(gu_w.shape[0] // 2, dtype=x.dtype)
        self.dec_gate_up(x, ln2, eps, gu_w, h)
        self.gemv(h, down_w, residual=x, out=x)
        if nxt is not None:
            ln1, qkv_w, qkv_b, q_out, pool, li, slot_of, qd, kd, length, len_dev = nxt
            self.dec_qkv(x, ln1, eps, qkv_w, qkv_b, q_out, pool, li, slot_of, qd, kd, length=length, len_dev=len_dev)

    def dec_lm_head(self, x, ln_w, eps, W, logits, seen, penalty, suppress, ws, temperature=None, rng=None, state=None):
        logits.copy_(F.linear(om.rms_norm(x.reshape(1, -1), ln_w, eps), W)[0].float())
        self._pending = (logits, seen, penalty, suppress, temperature, rng)

    def argmax_finish(self, ws, V, seen, tok_buf, state, advance_kv):
        logits, seen_, penalty, suppress, temperature, rng = self._pending
        if rng is not None:
            return self.penalty_sample(logits, seen_, penalty, suppress, temperature, 0, 1.0, rng, tok_buf, state, advance_kv, ws)
        self.penalty_argmax(logits, seen_, penalty, suppress, tok_buf, state, advance_kv, ws)

    def penalty_sample(self, logits, seen, penalty, suppress, temperature, top_k, top_p, rng, tok_buf, state, advance_kv, ws=None):
        """CPU stand-in of svlm_penalty_sample: the oracle's processors + a multinomial draw seeded by (rng, token index)."""
        from oracle.generate import warp_scores
        sc = logits.clone()
        if seen is not None:
            m = seen.bool()
            sc[m] = torch.where(sc[m] < 0, sc[m] * penalty, sc[m] / penalty)
        if suppress is not None:
            sc[suppress.long()] = float("-inf")
        cur = int(state[1]) + 1
        g = torch.Generator().manual_seed((int(rng[0]) & 0xFFFFFFFF) * 1000003 + (int(rng[1]) & 0xFFFFFFFF) * 7919 + cur)
        tok = int(torch.multinomial(torch.softmax(warp_scores(sc, temperature, top_k, top_p), dim=-1), 1, generator=g))
        tok_buf[cur] = tok
        state[1] = cur
        state[0] += advance_kv
        if seen is not None:
            seen[tok] = 1

    def penalty_argmax(self, logits, seen, penalty, suppress, tok_buf, state, advance_kv, ws=None):
        sc = logits.clone()
        if seen is not None:
            m = seen.bool()
            sc[m] = torch.where(sc[m] < 0, sc[m] * penalty, sc[m] / penalty)
        if suppress is not None:
            sc[suppress.long()] = float("-inf")
        tok = int(torch.argmax(sc))
        cur = int(state[1]) + 1
        tok_buf[cur] = tok
        state[1] = cur
        state[0] += advance_kv
        if seen is not None:
            seen[tok] = 1
