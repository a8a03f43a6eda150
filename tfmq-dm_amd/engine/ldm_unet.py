"""Latent-diffusion UNet with SpatialTransformer blocks (Stable Diffusion v1 family) lowered to the HIP
kernels.  Dataflow of `UNetModel.forward` (ldm/modules/diffusionmodules/openaimodel.py:744-780) with
`QuantResBlock` / `QuantBasicTransformerBlock` / `cross_attn_forward` (quant/quant_block.py:178-299):

  ResBlock            GN32(1e-5)+SiLU+quant -> conv3x3 (+emb row) -> GN+SiLU+quant -> conv3x3 (+skip / 1x1 FP skip)
  SpatialTransformer  GN(1e-6)+quant -> proj_in 1x1 -> tokens -> transformer block -> quant -> proj_out 1x1 (+x)
  transformer block   LN+quant -> fused q|k|v GEMM -> flash attention (8 heads) -> quant -> to_out (+x)
                      LN+quant -> to_q ; context -> quant -> fused k|v GEMM -> cross attention -> to_out (+x)
                      LN+quant -> ff.net.0.proj -> GEGLU+quant -> ff.net.2 (+x)
Token tensors [B,T,C] are NHWC images with W = 1, so every Linear over tokens is the same implicit-GEMM
kernel as the 1x1 convs.  Attention matmuls stay un-quantised (their quantizers are never enabled,
SURVEY §0 fact 2)."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from .. import ops
from .._lib import TfmqError
from .ddim_unet import UnitReached, StopAt, DdimUNetEngine, LayerQ, _Layer, _tape


def _n_children(sd, prefix):
    idx = set()
    for k in sd:
        if k.startswith(prefix + "."):
            head = k[len(prefix) + 1:].split(".")[0]
            if head.isdigit():
                idx.add(int(head))
    return len(idx)


def ldm_resblock_paths(sd) -> List[str]:
    out = []
    for grp in ("input_blocks", "middle_block", "output_blocks"):
        if grp == "middle_block":
            out += [f"{grp}.{j}" for j in range(_n_children(sd, grp)) if f"{grp}.{j}.in_layers.0.weight" in sd]
            continue
        for i in range(_n_children(sd, grp)):
            out += [f"{grp}.{i}.{j}" for j in range(_n_children(sd, f"{grp}.{i}")) if f"{grp}.{i}.{j}.in_layers.0.weight" in sd]
    return out


class LdmUNetEngine(DdimUNetEngine):
    """cfg: model_channels, num_heads (head dim = channels // num_heads), in_channels."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: dict, device="cuda:0"):
        # token Linears become 1x1 convs (same GEMM kernel); the TIB Linears stay small GEMVs
        sd2 = {}
        for k, v in sd.items():
            if k.endswith(".weight") and v.dim() == 2 and not (k.startswith("time_embed") or ".emb_layers." in k):
                v = v.reshape(v.shape[0], v.shape[1], 1, 1)
            elif k.endswith(".weight") and v.dim() == 3:          # Conv1d of an AttentionBlock (kernel size 1)
                v = v.reshape(v.shape[0], v.shape[1], 1, 1)
            sd2[k] = v
        # AttentionBlock (QKVAttentionLegacy, openaimodel.py:372-405): the qkv projection's output channels are ordered
        # (head, {q,k,v}, c).  Re-order its rows once to ({q,k,v}, head, c): q | k | v become column slices of one
        # buffer, exactly the operand layout of the attention kernels.  Same mathematics.
        for k in [k for k in sd2 if k.endswith(".qkv.weight")]:
            Cc = sd2[k].shape[1]
            nhc = cfg.get("num_head_channels", -1)
            heads = cfg["num_heads"] if nhc in (-1, None) else Cc // nhc
            d = Cc // heads
            perm = torch.arange(3 * Cc).reshape(heads, 3, d).permute(1, 0, 2).reshape(-1).to(sd2[k].device)
            sd2[k] = sd2[k][perm].contiguous()
            if k[:-6] + "bias" in sd2:
                sd2[k[:-6] + "bias"] = sd2[k[:-6] + "bias"][perm].contiguous()
        super().__init__(sd2, dict(cfg), device)
        self.res_names = ldm_resblock_paths(self.sd)
        self._ctx_pad = None
        self._pair_half = False
        self.fuse_q8 = os.environ.get("TFMQ_NO_Q8") is None

    def tib_layout(self):
        return ("time_embed.0", "time_embed.2", [r + ".emb_layers.1" for r in self.res_names],
                self.cfg["model_channels"], True)

    # ------------------------------------------------------------------ prepare: add fused k|v of the cross attention
    def prepare(self, wq=None, qtable=None, step=None, attn_q=None):
        super().prepare(wq, qtable, step, attn_q)
        wq = wq or {}
        self.fused_kv: Dict[str, _Layer] = {}
        # self-attention q|k|v: the base class fuses names ending in ".q"; here the layers are to_q/to_k/to_v
        for n in list(self.layers):
            if not n.endswith(".to_q"):
                continue
            p = n[:-5]
            names3 = [p + s for s in (".to_q", ".to_k", ".to_v")]
            ls = [self.layers[x] for x in names3]
            if p.endswith("attn1"):
                f = self._fuse(ls, [wq.get(x) for x in names3])
                if f is not None:
                    self.fused_qkv[p] = f
            else:
                f = self._fuse(ls[1:], [wq.get(x) for x in names3[1:]])
                if f is not None:
                    self.fused_kv[p] = f

        # GEGLU projection with the value/gate rows interleaved per 128-column tile: the GEMM epilogue applies
        # value * gelu(gate) and the consumer's 8-bit quantizer, the [B,T,2*inner] fp32 tensor never reaches HBM
        self.geglu_fused: Dict[str, ops.PackedW4] = {}
        for n in list(self.layers):
            if not n.endswith(".ff.net.0.proj"):
                continue
            l0, l2 = self.layers[n], self.layers.get(n[:-len("0.proj")] + "2")
            q = wq.get(n)
            if l0.kind != "w4a8" or l2 is None or l2.kind != "w4a8" or q is None or l0.wide or l2.wide:
                continue
            w, b = self.sd[n + ".weight"], self.sd.get(n + ".bias")
            inner = w.shape[0] // 2
            if inner % 64:
                continue
            perm = ops.geglu_perm(inner, self.dev)
            a = None if q.alpha is None else q.alpha.to(self.dev)[perm].contiguous()
            self.geglu_fused[n] = ops.pack_w4(w[perm].contiguous(), q.delta.to(self.dev).reshape(-1)[perm].contiguous(),
                                              q.zp.to(self.dev).reshape(-1)[perm].contiguous(), a,
                                              None if b is None else b[perm].contiguous())

    def _fuse(self, ls, qs):
        kinds = {l.kind for l in ls}
        if len(kinds) != 1 or any(l.wide for l in ls):
            return None
        if ls[0].kind == "w4a8":
            ids = [q.qid for q in qs]
            if not all(bool(torch.equal(self.qtable[:, ids[0]], self.qtable[:, i])) for i in ids[1:]):
                return None
            bias = None if ls[0].p.bias is None else torch.cat([l.p.bias for l in ls])
            pk = ops.PackedW4(torch.cat([l.p.packed for l in ls]), torch.cat([l.p.wmeta for l in ls]),
                              torch.cat([l.p.wscale for l in ls]), bias, sum(l.p.cout for l in ls), ls[0].p.cin, 1, 1)
            f = _Layer("w4a8", pk, ls[0].aq)
            f.sibling_qids = tuple(ids[1:])
            return f
        pf = ls[0].p
        ws = None if pf.wscale is None else torch.cat([l.p.wscale for l in ls])
        bias = None if pf.bias is None else torch.cat([l.p.bias for l in ls])
        return _Layer(ls[0].kind, ops.PackedF16(torch.cat([l.p.w16 for l in ls]), bias, sum(l.p.cout for l in ls), pf.cin, 1, 1, ws), None,
                      torch.cat([l.w32 for l in ls]) if self.exact_fp else None)

    # ------------------------------------------------------------------ helpers
    def _quant_in(self, layer: _Layer, x: torch.Tensor, siblings=()):
        """fp32 tensor -> what `layer` consumes (int8 under its activation quantizer, or fp32)."""
        if layer.kind != "w4a8":
            return x
        if self.calib is not None:
            self._observe(layer.aq, x, siblings)
        return ops.quantize_act(x, layer.aq)

    def _ln(self, name, x, layer: _Layer):
        g, b = self.sd[name + ".weight"], self.sd[name + ".bias"]
        if layer.kind == "w4a8" and self.calib is None:
            return ops.layernorm(x, g, b, 1e-5, layer.aq)[0]
        yf = ops.layernorm(x, g, b, 1e-5, None)[1]
        tp = _tape()
        if tp is not None and tp.depends(x):
            tp.rec([x], [yf], lambda gouts: [ops.layernorm_bwd(x.contiguous(), gouts[0].reshape(x.shape).contiguous(), g, 1e-5)])
        return self._quant_in(layer, yf, getattr(layer, "sibling_qids", ()))

    def _tok(self, layer: _Layer, x, **kw):
        """Linear over tokens: x [B,T,C] (int8 or fp32) -> fp32 [B,T,N]."""
        B, T, Cc = x.shape
        kw.setdefault("want_stats", False)
        if "residual" in kw and kw["residual"] is not None:
            kw["residual"] = kw["residual"].reshape(B, T, 1, -1)
        y = layer.run(x.reshape(B, T, 1, Cc), **kw)
        return y.reshape(B, T, -1)

    # ------------------------------------------------------------------ blocks
    def _res(self, p, x1, x2, rowadd_kw, out_aq=None):
        """out_aq: the block's output is consumed only by that activation quantizer (an up-sampling conv follows): the
        last conv's epilogue writes the int8 bins instead of fp32 (TFMQ_OUT_Q8)."""
        L = self.layers
        cin, cout = L[p + ".in_layers.2"], L[p + ".out_layers.3"]
        has_skip = (p + ".skip_connection") in L
        if x2 is not None and not has_skip:
            raise TfmqError(f"{p}: concatenated input without skip_connection")
        half = has_skip and self._fp_conv_half_ok(L[p + ".skip_connection"])
        virt = has_skip and self._virtual_cat_ok(L[p + ".skip_connection"], x1, x2)     # the skip conv reads its two sources itself
        h, xcat = self._gn(p + ".in_layers.0", x1, x2, True, cin,
                           want_cat=has_skip and not virt and (x2 is not None or (half and x1.dtype != torch.float16)), eps=1e-5,
                           half=half, half_main=True)
        h = cin.run(h, pad=(1, 1, 1, 1), **rowadd_kw, **self._o16())
        h, _ = self._gn(p + ".out_layers.0", h, None, True, cout, eps=1e-5, half_main=True)
        if virt:
            sc = L[p + ".skip_connection"].run(x1, x2=x2, want_stats=False, **self._o16())
        else:
            sc = L[p + ".skip_connection"].run(xcat if xcat is not None else x1, want_stats=False, **self._o16()) if has_skip else x1
        if out_aq is not None and cout.kind == "w4a8" and not cout.wide:
            return cout.run(h, pad=(1, 1, 1, 1), residual=sc, want_stats=False, out_q8=out_aq)
        return cout.run(h, pad=(1, 1, 1, 1), residual=sc, **self._o16())

    def _attention(self, p, xq_src, ctx, x_res, self_attn: bool):
        """one CrossAttention + residual; xq_src: LN output already in to_q's input form."""
        L = self.layers
        heads = self.cfg["num_heads"]
        to_out = L[p + ".to_out.0"]
        f = self.fused_qkv.get(p) if self_attn else None
        aq_on = p in self.attn_q          # cross_attn_forward with use_aq (quant_block.py:226-243)
        if self.exact_fp or aq_on:
            f = None if f is None or f.kind == "w4a8" else f      # (the w4a8 fused projection would write fp16 attention operands)
        if f is not None and f.kind != "w4a8" and self.calib is None and not self.exact_fp and not aq_on:
            # FP / weight-only state (calibration data passes, FP sampling of the calibration set): the same fp16-operand
            # attention, fed by the un-quantised fused projection
            B, T, Cin = xq_src.shape
            Cc = f.p.cout // 3
            d = Cc // heads
            if (ops.attention_f16_ok(d, T) and (2 * Cc) % 128 == 0 and T % 4 == 0 and xq_src.dtype == torch.float32
                    and ops.f16_dma_ok(Cin, 1, 1)):
                y16, vt = ops.conv2d_f16(ops.to_half(xq_src.reshape(B, T, 1, Cin)), f.p, out_f16=True, t_col0=2 * Cc)
                y16 = y16.reshape(B, T, 3 * Cc)
                o, _ = ops.attention_f16(y16[..., :Cc], y16[..., Cc:2 * Cc], vt, heads, float(d ** -0.5))
                return self._tok(to_out, self._quant_in(to_out, o), residual=x_res, **self._o16())
        if f is not None and f.kind == "w4a8":
            # main path: the projection GEMM writes q | k as fp16 rows and v as fp16 V^T, the attention kernel
            # copies them tile by tile (no fp32 round trip, no conversion, no transposition)
            B, T, Cin = xq_src.shape
            Cc = f.p.cout // 3
            d = Cc // heads
            if ops.attention_f16_ok(d, T) and (2 * Cc) % 128 == 0 and T % 4 == 0:
                y16, vt = ops.conv2d_w4a8(xq_src.reshape(B, T, 1, Cin), f.p, f.aq, out_f16=True, t_col0=2 * Cc)
                y16 = y16.reshape(B, T, 3 * Cc)
                aq = to_out.aq if to_out.kind == "w4a8" else None
                if aq is not None and self.calib is None:
                    _, o = ops.attention_f16(y16[..., :Cc], y16[..., Cc:2 * Cc], vt, heads, float(d ** -0.5), aq, want_f32=False)
                else:
                    o, _ = ops.attention_f16(y16[..., :Cc], y16[..., Cc:2 * Cc], vt, heads, float(d ** -0.5))
                    o = self._quant_in(to_out, o)
                return self._tok(to_out, o, residual=x_res, **self._o16())
        if (not self_attn) and self.calib is None and self._ctx_pad is not None and not self.exact_fp and not aq_on:
            # cross attention on fp16 operands: the context is stored padded to a multiple of 8 tokens (padding masked
            # in the kernel), to_q writes fp16 rows, to_k fp16 rows, to_v its fp16 transpose
            lq, lk, lv = L[p + ".to_q"], L[p + ".to_k"], L[p + ".to_v"]
            cpad, n_ctx = self._ctx_pad
            B, T, Cin = xq_src.shape
            Cc = lq.p.cout
            d = Cc // heads
            if lq.kind == lk.kind == lv.kind == "w4a8" and not (lq.wide or lk.wide or lv.wide) and ops.attention_f16_ok(d, cpad.shape[1]):
                q16 = ops.conv2d_w4a8(xq_src.reshape(B, T, 1, Cin), lq.p, lq.aq, out_f16=True).reshape(B, T, Cc)
                return self._cross_core(p, q16, x_res)
        return self._attention_rest(p, xq_src, ctx, x_res, self_attn)

    def _cross_ok(self, p) -> bool:
        """The fp16-operand cross attention of the sampling path (context padded to a multiple of 8 keys) applies to block p."""
        L = self.layers
        lq, lk, lv = L[p + ".to_q"], L[p + ".to_k"], L[p + ".to_v"]
        if self.calib is not None or self._ctx_pad is None or self.exact_fp or p in self.attn_q:
            return False
        d = lq.p.cout // self.cfg["num_heads"]
        return (lq.kind == lk.kind == lv.kind == "w4a8" and not (lq.wide or lk.wide or lv.wide) and ops.attention_f16_ok(d, self._ctx_pad[0].shape[1]))

    def _cross_core(self, p, q16, x_res, defer_out=False):
        """k | v projections of the padded context + the fp16 attention (+ to_out and the residual unless defer_out: then the int8 bins
        of to_out's quantizer are returned).  q16: fp16 [B, T, C] queries."""
        L = self.layers
        heads = self.cfg["num_heads"]
        lq, lk, lv, to_out = L[p + ".to_q"], L[p + ".to_k"], L[p + ".to_v"], L[p + ".to_out.0"]
        cpad, n_ctx = self._ctx_pad
        B, T, Cc = q16.shape
        d = Cc // heads
        Bc, Lp, Dc = cpad.shape
        fkv = self.fused_kv.get(p)
        if fkv is not None and fkv.kind == "w4a8" and Cc % 128 == 0 and os.environ.get("TFMQ_FUSED_KV", "1") != "0":
            # to_k and to_v quantise the same context with quantizers that agree at every step (prepare()): one quantise pass and
            # one GEMM write k as fp16 rows and v as its fp16 transpose (the transposed region starts at a multiple of 128 channels)
            kv, vt = ops.conv2d_w4a8(ops.quantize_act(cpad, fkv.aq).reshape(Bc, Lp, 1, Dc), fkv.p, fkv.aq, out_f16=True, t_col0=Cc)
            k16 = kv.reshape(Bc, Lp, 2 * Cc)[..., :Cc]
        else:
            k16 = ops.conv2d_w4a8(ops.quantize_act(cpad, lk.aq).reshape(Bc, Lp, 1, Dc), lk.p, lk.aq, out_f16=True).reshape(Bc, Lp, Cc)
            _, vt = ops.conv2d_w4a8(ops.quantize_act(cpad, lv.aq).reshape(Bc, Lp, 1, Dc), lv.p, lv.aq, out_f16=True, t_col0=0)
        aq = to_out.aq if to_out.kind == "w4a8" else None
        if aq is not None:
            _, o = ops.attention_f16(q16, k16, vt, heads, float(d ** -0.5), aq, want_f32=False, n_keys=n_ctx)
        else:
            o, _ = ops.attention_f16(q16, k16, vt, heads, float(d ** -0.5), n_keys=n_ctx)
        if defer_out:
            return o
        return self._tok(to_out, o, residual=x_res, **self._o16())

    def _attention_rest(self, p, xq_src, ctx, x_res, self_attn: bool):
        """_attention's generic forms (fp32 operands, calibration observation, exact / quantised attention)."""
        L = self.layers
        heads = self.cfg["num_heads"]
        to_out = L[p + ".to_out.0"]
        aq_on = p in self.attn_q
        if self_attn and p in self.fused_qkv:
            qkv = self._tok(self.fused_qkv[p], xq_src)
            Cc = qkv.shape[-1] // 3
            q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
        else:
            q = self._tok(L[p + ".to_q"], xq_src)
            Cc = q.shape[-1]
            if self_attn:
                k = self._tok(L[p + ".to_k"], xq_src)
                v = self._tok(L[p + ".to_v"], xq_src)
            elif p in self.fused_kv:
                f = self.fused_kv[p]
                kv = self._tok(f, self._quant_in(f, ctx, getattr(f, "sibling_qids", ())))
                k, v = kv[..., :Cc], kv[..., Cc:]
            else:
                k = self._tok(L[p + ".to_k"], self._quant_in(L[p + ".to_k"], ctx))
                v = self._tok(L[p + ".to_v"], self._quant_in(L[p + ".to_v"], ctx))
        d = Cc // heads
        aq = to_out.aq if to_out.kind == "w4a8" else None
        if aq_on:
            o = self._quant_in(to_out, self._attention_quantised(p, q, k, v, heads, float(d ** -0.5)))
        elif self.exact_fp:
            o, oq = self._attention_exact(q, k, v, heads, float(d ** -0.5), aq if self.calib is None else None)
            o = oq if oq is not None else self._quant_in(to_out, o)
        elif aq is not None and self.calib is None:
            _, o = ops.attention(q, k, v, heads, float(d ** -0.5), aq, want_f32=False)
        else:
            o, _ = ops.attention(q, k, v, heads, float(d ** -0.5))
            o = self._quant_in(to_out, o)
        return self._tok(to_out, o, residual=x_res, **self._o16())

    def _attn2_single_token(self, p, ctx, x):
        """Cross attention over a context of ONE token (the class embedding of the class-conditional LDM): the softmax over one key
        is exactly 1, so every query's attention output is to_v(context) of its batch item, to_out of it is one row per batch item
        and the block adds that row to every token -- what CrossAttention.forward (ldm/modules/attention.py:168-194) computes, without
        norm2 / to_q / to_k / the attention kernel, whose values cannot reach the output.  (Activation calibration runs the whole
        block: the quantizers of to_q / to_k still see their inputs.)"""
        L = self.layers
        lv, to_out = L[p + ".to_v"], L[p + ".to_out.0"]
        B, T, Cc = x.shape
        v = self._tok(lv, self._quant_in(lv, ctx))                 # [B, 1, C] fp32
        r = self._tok(to_out, self._quant_in(to_out, v))           # [B, 1, C] fp32: to_out(v) + bias
        return ops.row_broadcast_add(x, r.reshape(B, Cc))

    @staticmethod
    def _dup(h: torch.Tensor) -> torch.Tensor:
        """[B, ...] -> [2B, ...] = cat(h, h): the guidance pair of `_forward(pair_prefix=True)` from its shared prefix (device copies)."""
        B = h.shape[0]
        out = ops._alloc(2 * B, *h.shape[1:], dtype=h.dtype, device=h.device)
        out[:B].copy_(h)
        out[B:].copy_(h)
        st = getattr(h, "_tfmq_stats", None)
        if st is not None:      # the producing epilogue's GroupNorm statistics ([image][segment] rows, taken before the fp16 rounding) travel along
            out._tfmq_stats = (LdmUNetEngine._dup(st[0]), st[1])
        return out

    def _chain_ok(self, p, x, taps) -> bool:
        """SpatialTransformer p (depth 1) at token width 320 in the fp16-stream sampling state: its Linears around the two attentions run as
        row chains (tfmq_row_chain) -- [norm + proj_in + norm1 + q|k|v] and [attn1.to_out + residual + norm2 + attn2.to_q]."""
        L = self.layers
        B, H, W, Cc = x.shape
        T = H * W
        tb = p + ".transformer_blocks.0"
        f = self.fused_qkv.get(tb + ".attn1")
        pin = L[p + ".proj_in"]
        heads = self.cfg["num_heads"]
        return (self._h16 and self.calib is None and taps is None and not self.exact_fp and x.dtype == torch.float16 and ops.row_chain_ok(Cc, B * T, T, True) and ops.chain_tokens_ok(B * T)
                and _n_children(self.sd, p + ".transformer_blocks") == 1 and pin.kind == "w4a8" and not pin.wide and pin.p.cout == Cc
                and f is not None and f.kind == "w4a8" and not f.wide and f.p.cout == 3 * Cc and (tb + ".attn1") not in self.attn_q
                and ops.attention_f16_ok(Cc // heads, T) and getattr(x, "_tfmq_stats", None) is not None and T % x._tfmq_stats[1] == 0
                and L[tb + ".attn1.to_out.0"].kind == "w4a8" and not L[tb + ".attn1.to_out.0"].wide)

    def _tblock_chained(self, sp, x_img, ctx, out_aq=None, final_ok=False):
        """_st's norm / proj_in and the transformer block of SpatialTransformer sp with the row chains (see _chain_ok).  Returns the block's
        output tokens (fp16, or proj_out's int8 bins with out_aq) -- what _tblock returns -- and False; or, with final_ok and proj_out fused
        behind the feed-forward, the SpatialTransformer's OUTPUT image (proj_out + x_img, with its GroupNorm statistics) and True.
        Bit-identical to the separate launches."""
        L = self.layers
        B, H, W, Cc = x_img.shape
        T, heads = H * W, self.cfg["num_heads"]
        p = sp + ".transformer_blocks.0"
        pin, f = L[sp + ".proj_in"], self.fused_qkv[p + ".attn1"]
        ab = ops.gn_affine_from_stats(x_img, self.sd[sp + ".norm.weight"], self.sd[sp + ".norm.bias"], 1e-6)
        pre = ops.row_chain(x_img.reshape(B * T, Cc), T, [dict(pw=pin.p, aq=pin.aq, ln=True), dict(pw=f.p, aq=f.aq, t_col0=2 * Cc)], gn=ab,
                            ln=(self.sd[p + ".norm1.weight"], self.sd[p + ".norm1.bias"], 1e-5))
        h, y16, vt = pre[0][0].reshape(B, T, Cc), pre[1][0].reshape(B, T, 3 * Cc), pre[1][1]
        to_out = L[p + ".attn1.to_out.0"]
        _, o = ops.attention_f16(y16[..., :Cc], y16[..., Cc:2 * Cc], vt, heads, float((Cc // heads) ** -0.5), to_out.aq, want_f32=False)
        pend = self._pair_half
        lq = L[p + ".attn2.to_q"]
        single = ctx is not None and ctx.shape[1] == 1 and os.environ.get("TFMQ_SINGLE_CTX_TOKEN", "1") != "0"
        if (not single) and self._cross_ok(p + ".attn2") and lq.p.cout == Cc:
            mid = ops.row_chain(o.reshape(B * T, Cc), T, [dict(pw=to_out.p, aq=to_out.aq, residual=h.reshape(B * T, Cc), ln=True), dict(pw=lq.p, aq=lq.aq)],
                                ln=(self.sd[p + ".norm2.weight"], self.sd[p + ".norm2.bias"], 1e-5))
            x, q16 = mid[0][0].reshape(B, T, Cc), mid[1][0].reshape(B, T, Cc)
            if pend:      # the guidance pair parts here (norm2 / to_q saw the shared tensor: per-item arithmetic, same bits)
                x, q16, self._pair_half = self._dup(x), self._dup(q16), False
            to_out2, pout = L[p + ".attn2.to_out.0"], L[sp + ".proj_out"]
            ff0, ff2 = L[p + ".ff.net.0.proj"], L[p + ".ff.net.2"]
            gp = self.geglu_fused.get(p + ".ff.net.0.proj")
            Bx = x.shape[0]
            if (os.environ.get("TFMQ_FF_CHAIN", "1") != "0" and to_out2.kind == "w4a8" and not to_out2.wide and gp is not None and ff2.kind == "w4a8" and not ff2.wide
                    and ops.ff_fused_ok(Cc, gp.cout // 2, gp, ff2.p) and (Bx * T) % 256 == 0 and to_out2.p.cout == Cc):
                # round 4: attn2.to_out + residual, norm3, the feed-forward and (when the tokens feed nothing else) proj_out + the SpatialTransformer's
                # input + the next GroupNorm's statistics as ONE launch (tfmq_ff_fused with the Linears in front / behind)
                o2 = self._cross_core(p + ".attn2", q16, x, defer_out=True)
                pre = dict(xq=o2.reshape(Bx, T, Cc), pw=to_out2.p, aq=to_out2.aq, residual=x)
                post = None
                if final_ok and pout.kind == "w4a8" and not pout.wide and self.fuse_q8 and pout.p.cout == Cc:
                    xin = x_img if x_img.shape[0] == Bx else self._dup(x_img)
                    post = dict(pw=pout.p, residual=xin.reshape(Bx, T, Cc), stats=True, hw=T)
                _, y = ops.ff_fused(None, self.sd[p + ".norm3.weight"], self.sd[p + ".norm3.bias"], 1e-5, ff0.aq, gp, ff2.aq, ff2.p,
                                    out_q8=pout.aq if (post is not None or out_aq is not None) else None, pre=pre, post=post)
                if post is not None:
                    y4 = y.reshape(Bx, H, W, Cc)
                    if getattr(y, "_tfmq_stats", None) is not None:      # (the attribute lives on the tensor object, not on its views)
                        y4._tfmq_stats = y._tfmq_stats
                    return y4, True
                return y, False
            x = self._cross_core(p + ".attn2", q16, x)
        else:
            x = self._tok(to_out, o, residual=h, **self._o16())
            x = self._tblock_after_attn1(p, x, ctx)
        return self._ff(p, x, out_aq), False

    def _tblock(self, p, x, ctx, out_aq=None):
        L = self.layers
        q1 = self.fused_qkv.get(p + ".attn1", L[p + ".attn1.to_q"])
        x = self._attention(p + ".attn1", self._ln(p + ".norm1", x, q1), None, x, True)
        x = self._tblock_after_attn1(p, x, ctx)
        return self._ff(p, x, out_aq)

    def _tblock_after_attn1(self, p, x, ctx):
        L = self.layers
        # pair_prefix: everything up to here saw only (x, t) -- identical for the two members of a guidance pair -- and ran once per
        # pair; the first cross attention is where the members part (norm2 / to_q's quantizer still see the shared tensor)
        pend = self._pair_half
        if (ctx is not None and ctx.shape[1] == 1 and self.calib is None and x.shape[-1] % 8 == 0 and not self.exact_fp and (p + ".attn2") not in self.attn_q
                and os.environ.get("TFMQ_SINGLE_CTX_TOKEN", "1") != "0"):
            if pend:
                x, self._pair_half = self._dup(x), False
            x = self._attn2_single_token(p + ".attn2", ctx, x)
        else:
            xq2 = self._ln(p + ".norm2", x, L[p + ".attn2.to_q"])
            if pend:
                x, xq2, self._pair_half = self._dup(x), self._dup(xq2), False
            x = self._attention(p + ".attn2", xq2, ctx, x, False)
        return x

    def _ff(self, p, x, out_aq=None):
        L = self.layers
        ff0, ff2 = L[p + ".ff.net.0.proj"], L[p + ".ff.net.2"]
        gp = self.geglu_fused.get(p + ".ff.net.0.proj")
        if gp is not None and self.calib is None:
            if (x.dtype == torch.float16 and ff2.kind == "w4a8" and not ff2.wide and ops.ff_fused_ok(x.shape[-1], gp.cout // 2, gp, ff2.p)
                    and (out_aq is not None or self._h16) and ops.chain_tokens_ok(x.numel() // x.shape[-1])):
                # round 4: norm3 -> ff.net.0.proj -> GEGLU -> quantise -> ff.net.2 (+ x) as ONE launch, a token per lane; the GEGLU bins
                # go from the accumulators into the second GEMM's MFMA operand.  Bit-identical to the three launches below.
                return ops.ff_fused(x, self.sd[p + ".norm3.weight"], self.sd[p + ".norm3.bias"], 1e-5, ff0.aq, gp, ff2.aq, ff2.p, out_q8=out_aq)
            xq = self._ln(p + ".norm3", x, ff0)
            B, T, Cc = xq.shape
            # (exact_fp: the diagnostics mode keeps the 5e-7 GELU -- the consumer-sized form moves bins)
            g = ops.conv2d_w4a8(xq.reshape(B, T, 1, Cc), gp, ff0.aq, geglu_oq=ff2.aq, geglu_exact=self.exact_fp).reshape(B, T, -1)
            if out_aq is not None:      # tokens feed only proj_out's quantizer: int8 straight from the epilogue
                return self._tok(ff2, g, residual=x, out_q8=out_aq)
            return self._tok(ff2, g, residual=x, **self._o16())
        h = self._tok(ff0, self._ln(p + ".norm3", x, ff0))
        if ff2.kind == "w4a8" and self.calib is None:
            g = ops.geglu(h, ff2.aq)[0]
        else:
            gg = ops.geglu(h, None)[1]
            tp = _tape()
            if tp is not None and tp.depends(h):
                tp.rec([h], [gg], lambda gouts: [ops.geglu_bwd(h.contiguous(), gouts[0].reshape(gg.shape).contiguous())])
            g = self._quant_in(ff2, gg)
        return self._tok(ff2, g, residual=x, **self._o16())

    def _st(self, p, x, ctx, taps=None, out_aq=None):
        """taps (reconstruction data capture): every QuantLayer / QuantBasicTransformerBlock of the SpatialTransformer is a
        reconstruction unit of its own (recon_model walks norm, proj_in, transformer_blocks, proj_out)."""
        L = self.layers
        B, H, W, Cc = x.shape
        pin, pout = L[p + ".proj_in"], L[p + ".proj_out"]
        if self._chain_ok(p, x, taps):
            fuse_q = pout.kind == "w4a8" and not pout.wide and self.fuse_q8
            tok, final = self._tblock_chained(p, x, ctx, out_aq=pout.aq if fuse_q else None, final_ok=out_aq is None and self._h16)
            if final:
                return tok
            if tok.shape[0] != B:       # pair_prefix: the guidance pair parted inside this transformer; the residual is the shared tensor
                x, B = self._dup(x), tok.shape[0]
            h = tok.reshape(B, H, W, -1) if tok.dtype == torch.int8 else self._quant_in(pout, tok.reshape(B, H, W, -1))
            if out_aq is not None and pout.kind == "w4a8" and not pout.wide:
                return pout.run(h, residual=x, want_stats=False, out_q8=out_aq)
            return pout.run(h, residual=x, **self._o16())
        h_in, _ = self._gn(p + ".norm", x, None, False, pin, eps=1e-6)
        h = pin.run(h_in, want_stats=False, **self._o16())
        if taps is not None:
            taps[p + ".proj_in"] = (h_in, h)
        tok = h.reshape(B, H * W, h.shape[-1])
        nblk = _n_children(self.sd, p + ".transformer_blocks")
        fuse_q = pout.kind == "w4a8" and not pout.wide and self.calib is None and taps is None and self.fuse_q8
        for i in range(nblk):
            name = f"{p}.transformer_blocks.{i}"
            tin = tok
            tok = self._tblock(name, tok, ctx, out_aq=pout.aq if (fuse_q and i == nblk - 1) else None)
            if taps is not None:
                taps[name] = ((tin, ctx), tok)
        if tok.shape[0] != B:       # pair_prefix: the guidance pair parted inside this transformer; the residual is the shared tensor
            x, B = self._dup(x), tok.shape[0]
        h = tok.reshape(B, H, W, -1) if tok.dtype == torch.int8 else self._quant_in(pout, tok.reshape(B, H, W, -1))
        if taps is not None and getattr(taps, "name", None) == p + ".proj_out":
            # Fisher tape (engine/fisher.py): d(out + x) / d out = I, so the fused launch's output stands for the layer's own output
            y = pout.run(h, residual=x, **self._o16())
            taps[p + ".proj_out"] = (h, y)
            return y
        if taps is not None:
            # the layer's own output (its reconstruction target) excludes the residual; the data path stays the fused
            # launch, so a tapped forward is bit-identical to the sampling forward
            taps[p + ".proj_out"] = (h, pout.run(h, want_stats=False))
        if out_aq is not None and pout.kind == "w4a8" and not pout.wide:
            return pout.run(h, residual=x, want_stats=False, out_q8=out_aq)
        return pout.run(h, residual=x, **self._o16())

    def _attn_block(self, p, x, taps=None):
        """AttentionBlock._forward (openaimodel.py:317-326): un-quantised (Conv1d is not a QuantLayer type)."""
        L = self.layers
        B, H, W, Cc = x.shape
        qkv_l, po = L[p + ".qkv"], L[p + ".proj_out"]
        nhc = self.cfg.get("num_head_channels", -1)
        heads = self.cfg["num_heads"] if nhc in (-1, None) else Cc // nhc
        d = Cc // heads
        T = H * W
        # the matmul seams of QKVAttentionLegacy as reconstruction units (QuantQKMatMul / QuantSMVMatMul, quant_block.py:303-354): their inputs
        # and outputs exist only in the explicit form below
        seam = taps is not None and getattr(taps, "stop", None) in (p + ".attention.qkv_matmul", p + ".attention.smv_matmul")
        if (self.calib is None and not self.exact_fp and not seam and (p + ".attention") not in self.attn_q and qkv_l.kind != "w4a8" and self._fp_conv_half_ok(qkv_l) and ops.attention_f16_ok(d, T)
                and T % 4 == 0 and os.environ.get("TFMQ_ATTNBLOCK_F16", "1") != "0"):
            # fp16 operands end to end: the GroupNorm writes fp16, the qkv conv writes q | k as fp16 rows and v as fp16 V^T, the flash
            # kernel copies them tile by tile -- the values the fp32-operand kernel below rounds to on load (same products), without
            # the fp32 q / k / v round trip (the pattern of the SpatialTransformer's FP state in _attention)
            hn, _ = self._gn(p + ".norm", x, None, False, qkv_l, eps=1e-5, half_main=True)
            if hn.dtype == torch.float16:
                if (2 * Cc) % 128 == 0:
                    y16, vt = ops.conv2d_f16(hn.reshape(B, T, 1, Cc), qkv_l.p, out_f16=True, t_col0=2 * Cc)
                    y16 = y16.reshape(B, T, 3 * Cc)
                else:
                    # the transposed region of a launch starts at a multiple of 128 output channels: two launches on the weight
                    # rows of q | k and of v (C = 672 of the CelebA UNet)
                    sp = getattr(qkv_l, "_qk_v", None)
                    if sp is None:
                        sp = qkv_l._qk_v = (ops.slice_f16_rows(qkv_l.p, 0, 2 * Cc), ops.slice_f16_rows(qkv_l.p, 2 * Cc, 3 * Cc))
                    y16 = ops.conv2d_f16(hn.reshape(B, T, 1, Cc), sp[0], out_f16=True).reshape(B, T, 2 * Cc)
                    _, vt = ops.conv2d_f16(hn.reshape(B, T, 1, Cc), sp[1], out_f16=True, t_col0=0)
                o, _ = ops.attention_f16(y16[..., :Cc], y16[..., Cc:2 * Cc], vt, heads, float(d ** -0.5))
                return po.run(o.reshape(B, H, W, Cc), residual=x, want_stats=True, **self._o16())
        else:
            hn, _ = self._gn(p + ".norm", x, None, False, qkv_l, eps=1e-5)
        qkv = qkv_l.run(hn, want_stats=False).reshape(B, H * W, 3 * Cc)
        if seam:
            self._attn_seam_taps(p, qkv, heads, taps)          # (ends the forward: the StopAt dictionary raises at the requested seam)
        # q*s . k*s with s = d^-1/4 (QKMatMul) == (q . k) * d^-1/2
        if (p + ".attention") in self.attn_q:      # QuantQKMatMul / QuantSMVMatMul with use_aq: the quantizers see q, k scaled by d^-1/4 (quant_block.py:318-323)
            o = self._attention_quantised(p + ".attention", qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], heads, 1.0, pre=float(d ** -0.25))
        elif self.exact_fp:
            o, _ = self._attention_exact(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], heads, float(d ** -0.5), None)
        else:
            o, _ = ops.attention(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], heads, float(d ** -0.5))
        return po.run(o.reshape(B, H, W, Cc), residual=x, want_stats=True, **self._o16())

    def _attn_seam_taps(self, p, qkv, heads: int, taps):
        """taps of the two matmul modules of an AttentionBlock: `<p>.attention.qkv_matmul` = ((q, k), weight) with weight = (q s)(k s)^T,
        s = d^-1/4 (QuantQKMatMul.forward, quant_block.py:313-328), and `<p>.attention.smv_matmul` = ((softmax(weight), v), a)
        (QuantSMVMatMul.forward :345-354) -- under the block's live matmul quantizers when the engine was prepared with them.  Layouts:
        q, k, v [B, T, heads d] (head-major channels), weight [B, heads, T, T], a [B, T, heads d]."""
        B, T, C3 = qkv.shape
        Cc = C3 // 3
        d = Cc // heads
        pre = float(d ** -0.25)
        q, k, v = (qkv[..., i * Cc:(i + 1) * Cc].contiguous() for i in range(3))
        cfg = self.attn_q.get(p + ".attention") or {}            # (one of the two modules may be live alone: q, k or v, w)
        sel = {w: ops.qsel(self.qtable, cfg[w], self.step) for w in ("q", "k", "v", "w") if w in cfg}

        def scaled(x):
            xs = ops._alloc_like(x)
            xs.zero_()
            ops.axpy(xs, x, pre)
            return xs
        qs, ks = scaled(q), scaled(k)
        if "q" in sel and "k" in sel:
            qs, ks = ops.fake_quant_sel(qs, sel["q"], 256), ops.fake_quant_sel(ks, sel["k"], 256)
        S = ops._alloc(B, heads, T, T, dtype=torch.float32, device=qkv.device)
        ops.gemm_strided(qs, 0, Cc, 1, T * Cc, ks, 0, 1, Cc, T * Cc, S, 0, T, heads * T * T, T, T, d, B, hsa=d, hsb=d, hsc=T * T, heads=heads)
        taps[p + ".attention.qkv_matmul"] = ((q, k), S)
        P = ops.softmax_rows(S, 1.0)
        Ph, vh = P, v
        if "v" in sel:
            vh = ops.fake_quant_sel(v, sel["v"], 256)
        if "w" in sel:
            Ph = ops.fake_quant_sel(P, sel["w"], cfg["w_level"])
        o = ops._alloc(B, T, Cc, dtype=torch.float32, device=qkv.device)
        ops.gemm_strided(Ph, 0, T, 1, heads * T * T, vh, 0, Cc, 1, T * Cc, o, 0, Cc, T * Cc, T, d, T, B, hsa=T * T, hsb=d, hsc=d, heads=heads)
        taps[p + ".attention.smv_matmul"] = ((P, v), o)

    def _seq(self, p, h, skip, ctx, rowadd, taps):
        L = self.layers
        nchild = _n_children(self.sd, p)
        for j in range(nchild):
            q = f"{p}.{j}"
            hin = h
            # the next child is an up-sampling conv on 8-bit activations: this child's output feeds only its quantizer
            nxt = L.get(f"{p}.{j + 1}.conv") if j + 1 < nchild else None
            out_aq = nxt.aq if (nxt is not None and nxt.kind == "w4a8" and not nxt.wide and self.calib is None and taps is None
                                and self.fuse_q8) else None
            if (q + ".in_layers.0.weight") in self.sd:
                h = self._res(q, h, skip if j == 0 else None, rowadd(q), out_aq=out_aq)
                if taps is not None:
                    taps[q] = ((hin, skip) if (j == 0 and skip is not None) else hin, h)
            elif (q + ".transformer_blocks.0.norm1.weight") in self.sd:
                h = self._st(q, h, ctx, taps, out_aq=out_aq)
                if taps is not None:
                    taps[q] = (hin, h)
            elif (q + ".qkv") in L:
                h = self._attn_block(q, h, taps)
                if taps is not None:
                    taps[q] = (hin, h)
            elif (q + ".op") in L:
                dl = L[q + ".op"]
                hd = ops.to_half(h) if (h.dtype == torch.float32 and self._fp_conv_half_ok(dl)) else h
                h = dl.run(hd, stride=2, pad=(1, 1, 1, 1), **self._o16())
            elif (q + ".conv") in L:
                up = L[q + ".conv"]
                hq = h if h.dtype == torch.int8 else self._quant_in(up, h)
                h = up.run(hq, pad=(1, 1, 1, 1), up2x=True, **self._o16())
                if taps is not None and hq.dtype == torch.float32:
                    taps[q + ".conv"] = (ops.upsample2x(hq), h)
            elif q in L:
                h = L[q].run(h, pad=(1, 1, 1, 1), **self._o16())
            else:
                raise TfmqError(f"LdmUNetEngine: unknown child {q}")
        return h

    # ------------------------------------------------------------------ forward
    def forward(self, *a, **k):
        """See _forward.  Outside activation calibration the conv / linear tile shapes are measured once per shape
        (ops.autotuned) and reused -- the output does not depend on them."""
        if not hasattr(self, "tiles"):
            self.tiles = {}
        try:
            with ops.autotuned(self.tiles if self.calib is None else None):
                return self._forward(*a, **k)
        except UnitReached:
            return None

    def _forward(self, x: torch.Tensor, t: Optional[torch.Tensor] = None, context: Optional[torch.Tensor] = None,
                taps: Optional[dict] = None, pair_prefix: bool = False) -> torch.Tensor:
        """x: fp32 NHWC latents [B,H,W,C]; t: [B] timesteps (or None -> per-step TIB table); context: fp32 [B,L,D].

        pair_prefix (classifier-free guidance, ldm/models/diffusion/ddim.py:180-186: x_in = cat([x] * 2), t_in = cat([t] * 2),
        c_in = cat([uc, c])): `x` holds the B latents ONCE, `context` the 2B rows cat(uc, c); returns eps of the 2B-item batch.
        The two members of a pair share (x, t), so every tensor in front of the first cross attention -- conv_in, the first
        ResBlock, the first SpatialTransformer's norm / proj_in / norm1 / self attention / norm2 -- is the same for both: it is
        computed once per pair and copied where the members part.  Per-item arithmetic is batch independent, so the result
        equals the forward of the materialised 2B batch bit for bit (tests/test_engine_ldm_gpu.py)."""
        if not self.prepared:
            raise TfmqError("LdmUNetEngine.forward before prepare()")
        if context is None and any(k.endswith(".attn2.to_k.weight") for k in self.sd):
            raise TfmqError("LdmUNetEngine: SpatialTransformer UNets need a context tensor")
        L = self.layers
        self._h16 = self.stream_f16 and self.calib is None and taps is None and self._stream_f16_possible()
        self._pair_half = False
        if pair_prefix:
            if context is None or context.shape[0] != 2 * x.shape[0] or self.calib is not None or taps is not None:
                raise TfmqError("LdmUNetEngine: pair_prefix needs B latents with 2B context rows, outside calibration / taps")
            if t is not None and (t.shape[0] != 2 * x.shape[0] or not bool(torch.equal(t[:x.shape[0]], t[x.shape[0]:]))):
                raise TfmqError("LdmUNetEngine: pair_prefix needs t = cat([t] * 2)")
            self._pair_half = True
        if t is not None:
            projs = dict(zip(self.res_names, self.tib(t)))
            if taps is not None:
                taps["__temb__"] = self._last_temb

            def rowadd(p):
                return dict(rowadd=projs[p])
        else:
            if self.tib_table is None:
                raise TfmqError("forward(t=None) needs build_tib_table() first")

            def rowadd(p):
                o = self.tib_off[p]
                return dict(rowadd=self.tib_table[0, o:], rowadd_ld=0, rowadd_step=self.step,
                            rowadd_step_stride=self.tib_table.shape[1])
        ctx = None if context is None else context.contiguous()
        self._ctx_pad = None
        if ctx is not None and self.calib is None:
            Bc, Lc, Dc = ctx.shape
            Lp = (Lc + 7) // 8 * 8
            cpad = ctx
            if Lp != Lc:      # e.g. 77 CLIP tokens -> 80 rows; the 3 zero rows are masked keys in the attention kernel
                cpad = ops._alloc(Bc, Lp, Dc, dtype=torch.float32, device=ctx.device)
                cpad.zero_()
                cpad[:, :Lc].copy_(ctx)
            self._ctx_pad = (cpad, Lc)
        hs = []
        h = x
        for i in range(_n_children(self.sd, "input_blocks")):
            was_half = self._pair_half
            h = self._seq(f"input_blocks.{i}", h, None, ctx, rowadd, taps)
            if was_half and not self._pair_half:        # the pair parted inside this block: the skips taken before it are shared
                hs = [self._dup(e) for e in hs]
            hs.append(h)
        if self._pair_half:
            raise TfmqError("LdmUNetEngine: pair_prefix found no cross attention in the down path")
        h = self._seq("middle_block", h, None, ctx, rowadd, taps)
        for i in range(_n_children(self.sd, "output_blocks")):
            h = self._seq(f"output_blocks.{i}", h, hs.pop(), ctx, rowadd, taps)
        h, _ = self._gn("out.0", h, None, True, None, eps=1e-5, half=self._fp_conv_half_ok(L["out.2"]))
        return L["out.2"].run(h, pad=(1, 1, 1, 1), want_stats=False)
