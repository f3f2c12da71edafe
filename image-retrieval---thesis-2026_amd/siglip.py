"""SigLIP (so400m) vision and text towers and the dual encoder the reference evaluates as "MedSigLIP".

Mirrors (paths into /root/reference):
  MedSigLIP wrapper: backbone = AutoModel.from_pretrained("google/medsiglip-448").vision_model      model.py:536-634
  zero-shot path: model.get_text_features / get_image_features / logit_scale                          eval_medsiglip.py:164-211
The reference takes both towers from `transformers` (a hub download).  Here they are own modules that reproduce the
transformers SigLIP parameter names (`vision_model.embeddings.patch_embedding`, `...encoder.layers.N.self_attn.q_proj`,
`...head.attention.in_proj_weight`, `text_model.embeddings.token_embedding`, `text_model.head`, `logit_scale`, ...) so that
MedSigLIP / SigLIP checkpoints load unchanged, and whose CUDA fp32 inference path runs on libmirx only:
  patch embedding  = mirx_patchify_nchw + MFMA Linear (three bf16 terms: the pixel range is unknown)
  LayerNorm        = mirx_layernorm
  q / k / v        = ONE packed MFMA Linear (two fp16 terms: its input is a LayerNorm output, bounded)
  attention        = vision: flash attention on the matrix pipe at head_dim 72 (mirx_attention_qkv_f32_split2h: two fp16
                     terms behind the provable q / k / v bounds; _split3 when a bound is not finite);
                     text (64 tokens, key-padding mask) and the pooling head (1 probe query): mirx_attention_small
  out_proj / fc2   = MFMA Linear with the residual added in the epilogue;  fc1 = MFMA Linear + tanh-GELU epilogue
Other inputs (CPU tensors, autograd, output_attentions=True for the reference's rollout explainer, model.py:546-551) take
the plain torch path of the same modules.  transformers is NOT imported by this module; tests cross-check against
transformers.SiglipModel built from a local config.
"""
import ctypes
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import model as _m

MEDSIGLIP_VISION = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                        image_size=448, patch_size=14)
MEDSIGLIP_TEXT = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                      vocab_size=32000, max_position_embeddings=64, projection_size=1152)
LN_EPS = 1e-6


def _fast(x):
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


class TowerOutput(SimpleNamespace):
    """last_hidden_state / pooler_output / attentions, attribute and index access like the transformers outputs."""

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output)[i]


class _PackedRows:
    """Several Linear layers that read the same input, seen as ONE Linear (weights stacked along the output axis):
    duck-typed for mirx.model._linear_s3 / _linear_h2 (in_features, out_features, weight, bias)."""

    def __init__(self, parts):
        self.parts = parts                       # callables -> (weight, bias)
        self._key = None
        self.__dict__["_cache"] = {}

    def refresh(self):
        wb = [p() for p in self.parts]
        key = tuple((w.data_ptr(), w._version, None if b is None else b._version) for w, b in wb)
        if key != self._key:
            self.weight = torch.cat([w.detach() for w, _ in wb], 0)
            self.bias = None if wb[0][1] is None else torch.cat([b.detach() for _, b in wb], 0)
            self.in_features, self.out_features = self.weight.shape[1], self.weight.shape[0]
            self._key = key
        return self


class _Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = dim, heads, dim // heads
        self.scale = self.head_dim ** -0.5
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.q_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)
        self.__dict__["_packed"] = _PackedRows([lambda: (self.q_proj.weight, self.q_proj.bias),
                                                lambda: (self.k_proj.weight, self.k_proj.bias),
                                                lambda: (self.v_proj.weight, self.v_proj.bias)])

    def forward(self, x, key_mask=None, output_attentions=False):
        """x [B, N, C]; key_mask [B, N] (1 = attend) or None.  -> (context before out_proj [B, N, C], probs or None)"""
        b, n, c = x.shape
        q = self.q_proj(x).view(b, n, self.num_heads, self.head_dim).transpose(1, 2)
        k = self.k_proj(x).view(b, n, self.num_heads, self.head_dim).transpose(1, 2)
        v = self.v_proj(x).view(b, n, self.num_heads, self.head_dim).transpose(1, 2)
        att = torch.matmul(q, k.transpose(2, 3)) * self.scale
        if key_mask is not None:
            att = att.masked_fill(~key_mask.bool()[:, None, None, :], torch.finfo(att.dtype).min)
        att = F.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        ctx = torch.matmul(att, v).transpose(1, 2).reshape(b, n, c)
        return ctx, (att if output_attentions else None)


class _MLP(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        if _fast(x):
            return _m._linear_auto(self.fc2, _m._linear_auto(self.fc1, x, act=2))
        return self.fc2(F.gelu(self.fc1(x), approximate="tanh"))


class _EncoderLayer(nn.Module):
    def __init__(self, dim, hidden, heads):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(dim, eps=LN_EPS)
        self.self_attn = _Attention(dim, heads)
        self.layer_norm2 = nn.LayerNorm(dim, eps=LN_EPS)
        self.mlp = _MLP(dim, hidden)

    def forward(self, x, key_mask=None, output_attentions=False):
        if _fast(x) and not output_attentions and self.self_attn.head_dim in (16, 32, 64, 72) and x.shape[-1] % 16 == 0:
            return self._forward_mirx(x, key_mask), None
        ctx, att = self.self_attn(self.layer_norm1(x), key_mask, output_attentions)
        x = x + self.self_attn.out_proj(ctx)
        return x + self.mlp(self.layer_norm2(x)), att

    def _forward_mirx(self, x, key_mask):
        sa, mlp = self.self_attn, self.mlp
        b, n, c = x.shape
        x = x.contiguous()
        lib = _lib.load()
        b1, b2 = _m._layernorm_bound(self.layer_norm1), _m._layernorm_bound(self.layer_norm2)
        pk = sa._packed.refresh()
        pk.__dict__["_mirx_cfg"] = _m._cfg(sa)                   # the packed view runs under its attention module's configuration
        bctx, bh = _m._linear_out_bound(self.layer_norm1, sa.v_proj), _m._linear_out_bound(self.layer_norm2, mlp.fc1)
        # big batches: every Linear on the DMA-fed kernel, its input handed over as terms rows by the producer
        terms = _m._linear_terms_ok(self, b * n, (pk, sa.out_proj, mlp.fc1, mlp.fc2), (b1, b2, bctx, bh))
        if terms:
            h1t, s1 = _m._layernorm_terms(self.layer_norm1, x, b1)
            qkv = _m._linear_terms(pk, h1t, s1, (b, n))                                                   # [b, n, 3c]
        else:
            h1 = _m._layernorm(self.layer_norm1, x)
            qkv = _m._linear_h2(pk, h1, b1) if _m._linear_h2_ok(pk, h1, b1) else _m._linear_s3(pk, h1)      # [b, n, 3c]
        bqk = max(_m._linear_out_bound(self.layer_norm1, sa.q_proj), _m._linear_out_bound(self.layer_norm1, sa.k_proj))
        bv = _m._linear_out_bound(self.layer_norm1, sa.v_proj)
        flash = key_mask is None and n >= 32 and sa.head_dim in (32, 64, 72, 96) and b <= 65535
        ct = None
        if terms and flash and _m._cfg(self).attention_two_fp16 and 0.0 < bqk < 3.0e4:
            # the attention kernel hands its output to the projection as terms rows too (|ctx| <= bctx)
            ct, sc = (torch.empty if c % 32 == 0 else torch.zeros)((b * n, (c + 31) // 32 * 64), dtype=torch.float16, device=x.device), _m._terms_scale(bctx)
            with torch.cuda.device(x.device):
                _lib.check(lib.mirx_attention_qkv_f32_split2h_terms(_m._ptr(qkv), b, n, sa.num_heads, sa.head_dim, float(sa.scale),
                                                                    bqk, bv, sc, _m._ptr(ct), _m._stream(x.device)),
                           "mirx_attention_qkv_f32_split2h_terms")
        ctx = torch.empty((b, n, c), dtype=torch.float32, device=x.device) if ct is None else None
        with torch.cuda.device(x.device):
            st = _m._stream(x.device)
            if ct is not None:
                pass
            elif flash and _m._cfg(self).attention_two_fp16 and 0.0 < bqk < 3.0e4 and 0.0 < bv < 3.0e4:
                # q, k, v come out of a LayerNorm-fed Linear: provable bounds -> two fp16 terms per operand
                _lib.check(lib.mirx_attention_qkv_f32_split2h(_m._ptr(qkv), b, n, sa.num_heads, sa.head_dim, float(sa.scale),
                                                              bqk, bv, _m._ptr(ctx), st), "mirx_attention_qkv_f32_split2h")
            elif flash:
                _lib.check(lib.mirx_attention_qkv_f32_split3(_m._ptr(qkv), b, n, sa.num_heads, sa.head_dim, float(sa.scale),
                                                             _m._ptr(ctx), st), "mirx_attention_qkv_f32_split3")
            else:
                km = None if key_mask is None else key_mask.to(device=x.device, dtype=torch.uint8).contiguous()
                base = qkv.data_ptr()
                _lib.check(lib.mirx_attention_small(ctypes.c_void_p(base), 3 * c, ctypes.c_void_p(base + 4 * c),
                                                    ctypes.c_void_p(base + 8 * c), 3 * c,
                                                    _m._ptr(km) if km is not None else None, b, sa.num_heads, sa.head_dim,
                                                    n, n, float(sa.scale), _m._ptr(ctx), st), "mirx_attention_small")
        # the context is a softmax-weighted average of V rows: bounded like the V rows of the packed projection; |gelu(v)| <= |v|
        if terms:
            if ct is None:
                ct, sc = _m._rows_to_terms(ctx, bctx)
            x = _m._linear_terms(sa.out_proj, ct, sc, (b, n), res=x)
            h2t, s2 = _m._layernorm_terms(self.layer_norm2, x, b2)
            hidt, sh = _m._linear_terms(mlp.fc1, h2t, s2, (b, n), act=2, terms_bound=bh)
            return _m._linear_terms(mlp.fc2, hidt, sh, (b, n), res=x, out=x)
        x = (_m._linear_h2(sa.out_proj, ctx, bv, res=x) if _m._linear_h2_ok(sa.out_proj, ctx, bv)
             else _m._linear_s3(sa.out_proj, ctx, res=x))
        h2 = _m._layernorm(self.layer_norm2, x)
        hid = (_m._linear_h2(mlp.fc1, h2, b2, act=2) if _m._linear_h2_ok(mlp.fc1, h2, b2) else _m._linear_s3(mlp.fc1, h2, act=2))
        bh = _m._linear_out_bound(self.layer_norm2, mlp.fc1)               # |gelu(v)| <= |v|
        if _m._linear_h2_ok(mlp.fc2, hid, bh):
            return _m._linear_h2(mlp.fc2, hid, bh, res=x, out=x)
        return _m._linear_s3(mlp.fc2, hid, res=x, out=x)


class _Encoder(nn.Module):
    def __init__(self, dim, hidden, heads, depth):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(dim, hidden, heads) for _ in range(depth)])

    def forward(self, x, key_mask=None, output_attentions=False):
        atts = [] if output_attentions else None
        for layer in self.layers:
            x, a = layer(x, key_mask, output_attentions)
            if output_attentions:
                atts.append(a)
        return x, (tuple(atts) if output_attentions else None)


class _VisionEmbeddings(nn.Module):
    def __init__(self, dim, image_size, patch, channels=3):
        super().__init__()
        self.patch_size = patch
        self.patch_embedding = nn.Conv2d(channels, dim, kernel_size=patch, stride=patch, padding="valid")
        self.num_positions = (image_size // patch) ** 2
        self.position_embedding = nn.Embedding(self.num_positions, dim)

    def forward(self, pixel_values):
        if _fast(pixel_values):
            tok = _m._conv_patch_tokens(self.patch_embedding, pixel_values)                 # [B * gh * gw, dim]
            x = tok.view(pixel_values.shape[0], -1, tok.shape[-1])
        else:
            x = self.patch_embedding(pixel_values).flatten(2).transpose(1, 2)
        if x.shape[1] != self.num_positions:
            raise ValueError(f"expected {self.num_positions} patches, got {x.shape[1]} (resize the image to the tower's size)")
        return x + self.position_embedding.weight[None]


class _PoolingHead(nn.Module):
    """transformers SiglipMultiheadAttentionPoolingHead: one learned probe attends over all tokens
    (nn.MultiheadAttention parameter names), then LayerNorm + MLP with a skip."""

    def __init__(self, dim, hidden, heads):
        super().__init__()
        self.probe = nn.Parameter(torch.randn(1, 1, dim))
        self.attention = nn.MultiheadAttention(dim, heads, batch_first=True)
        self.layernorm = nn.LayerNorm(dim, eps=LN_EPS)
        self.mlp = _MLP(dim, hidden)
        c = dim
        self.__dict__["_kv"] = _PackedRows([lambda: (self.attention.in_proj_weight[c:], self.attention.in_proj_bias[c:])])

    def _probe_query(self):
        """in_proj_q(probe): constant for given weights, computed once per weight version."""
        at, c = self.attention, self.probe.shape[-1]
        key = (self.probe._version, at.in_proj_weight._version, at.in_proj_bias._version, self.probe.device)
        cached = self.__dict__.get("_q")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key, F.linear(self.probe[0], at.in_proj_weight[:c], at.in_proj_bias[:c]))
            self.__dict__["_q"] = cached
        return cached[1]

    def forward(self, hidden, in_bound=None):
        b, n, c = hidden.shape
        at = self.attention
        dh = c // at.num_heads
        if _fast(hidden) and dh in (16, 32, 64, 72) and c % 16 == 0:
            lib = _lib.load()
            kvp = self._kv.refresh()
            hidden = hidden.contiguous()
            kv = (_m._linear_h2(kvp, hidden, in_bound) if in_bound is not None and _m._linear_h2_ok(kvp, hidden, in_bound)
                  else _m._linear_s3(kvp, hidden))                                             # [b, n, 2c]
            q = self._probe_query().expand(b, c).contiguous()                                   # [b, c]
            ctx = torch.empty((b, 1, c), dtype=torch.float32, device=hidden.device)
            with torch.cuda.device(hidden.device):
                base = kv.data_ptr()
                _lib.check(lib.mirx_attention_small(_m._ptr(q), c, ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * c), 2 * c,
                                                    None, b, at.num_heads, dh, 1, n, float(dh) ** -0.5, _m._ptr(ctx),
                                                    _m._stream(hidden.device)), "mirx_attention_small")
            x = _m._linear_auto(at.out_proj, ctx)
        else:
            x = at(self.probe.repeat(b, 1, 1), hidden, hidden, need_weights=False)[0]
        x = x + self.mlp(_m._layernorm(self.layernorm, x))
        return x[:, 0]


class SiglipVisionTower(nn.Module):
    """transformers SiglipVisionTransformer surface: forward(pixel_values, output_attentions, return_dict) ->
    .last_hidden_state / .pooler_output / .attentions; attributes embeddings / encoder / post_layernorm / head / config."""

    def __init__(self, hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                 image_size=448, patch_size=14, num_channels=3):
        super().__init__()
        self.config = SimpleNamespace(hidden_size=hidden_size, intermediate_size=intermediate_size,
                                      num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                      image_size=image_size, patch_size=patch_size, num_channels=num_channels,
                                      layer_norm_eps=LN_EPS, _attn_implementation="eager")
        self.embeddings = _VisionEmbeddings(hidden_size, image_size, patch_size, num_channels)
        self.encoder = _Encoder(hidden_size, intermediate_size, num_attention_heads, num_hidden_layers)
        self.post_layernorm = nn.LayerNorm(hidden_size, eps=LN_EPS)
        self.head = _PoolingHead(hidden_size, intermediate_size, num_attention_heads)

    def forward(self, pixel_values=None, output_attentions=False, return_dict=True, **_):
        x = self.embeddings(pixel_values)
        x, atts = self.encoder(x, None, output_attentions)
        x = _m._layernorm(self.post_layernorm, x)
        pooled = self.head(x, _m._layernorm_bound(self.post_layernorm))
        return TowerOutput(last_hidden_state=x, pooler_output=pooled, attentions=atts)


class SiglipTextTower(nn.Module):
    """transformers SiglipTextModel surface: token + position embeddings, the same encoder WITHOUT a causal mask, final
    LayerNorm, the LAST position pooled (a padding position when the prompt is shorter: SigLIP was trained that way),
    Linear head.  `attention_mask` [B, N] (1 = token) hides padded keys, as the reference passes it
    (eval_medsiglip.py:164-176)."""

    def __init__(self, hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                 vocab_size=32000, max_position_embeddings=64, projection_size=None):
        super().__init__()
        projection_size = projection_size or hidden_size
        self.config = SimpleNamespace(hidden_size=hidden_size, intermediate_size=intermediate_size,
                                      num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                      vocab_size=vocab_size, max_position_embeddings=max_position_embeddings,
                                      projection_size=projection_size, layer_norm_eps=LN_EPS)
        self.embeddings = nn.Module()
        self.embeddings.token_embedding = nn.Embedding(vocab_size, hidden_size)
        self.embeddings.position_embedding = nn.Embedding(max_position_embeddings, hidden_size)
        self.encoder = _Encoder(hidden_size, intermediate_size, num_attention_heads, num_hidden_layers)
        self.final_layer_norm = nn.LayerNorm(hidden_size, eps=LN_EPS)
        self.head = nn.Linear(hidden_size, projection_size)

    def forward(self, input_ids=None, attention_mask=None, output_attentions=False, return_dict=True, **_):
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        n = input_ids.shape[1]
        if n > self.config.max_position_embeddings:
            raise ValueError(f"Sequence length must be at most max_position_embeddings ({n} > "
                             f"{self.config.max_position_embeddings})")
        x = self.embeddings.token_embedding(input_ids) + self.embeddings.position_embedding.weight[None, :n]
        x, atts = self.encoder(x, attention_mask, output_attentions)
        x = _m._layernorm(self.final_layer_norm, x)
        return TowerOutput(last_hidden_state=x, pooler_output=_m._linear_auto(self.head, x[:, -1, :].contiguous()), attentions=atts)


class SiglipDualEncoder(nn.Module):
    """transformers SiglipModel surface used by eval_medsiglip.py: get_text_features(input_ids, attention_mask),
    get_image_features(pixel_values), logit_scale / logit_bias, forward(...) -> logits_per_image / text_embeds /
    image_embeds."""

    def __init__(self, vision_config=None, text_config=None):
        super().__init__()
        self.vision_model = SiglipVisionTower(**(vision_config or MEDSIGLIP_VISION))
        self.text_model = SiglipTextTower(**(text_config or MEDSIGLIP_TEXT))
        self.logit_scale = nn.Parameter(torch.tensor([math.log(10.0)]))
        self.logit_bias = nn.Parameter(torch.tensor([-10.0]))

    def get_text_features(self, input_ids, attention_mask=None, **_):
        return self.text_model(input_ids=input_ids, attention_mask=attention_mask).pooler_output

    def get_image_features(self, pixel_values, **_):
        return self.vision_model(pixel_values=pixel_values).pooler_output

    def forward(self, input_ids=None, pixel_values=None, attention_mask=None, **_):
        image_embeds = text_embeds = None
        if pixel_values is not None:
            image_embeds = F.normalize(self.get_image_features(pixel_values), dim=-1)
        if input_ids is not None:
            text_embeds = F.normalize(self.get_text_features(input_ids, attention_mask), dim=-1)
        logits = None
        if image_embeds is not None and text_embeds is not None:
            logits = (text_embeds @ image_embeds.t()) * self.logit_scale.exp() + self.logit_bias
        return SimpleNamespace(image_embeds=image_embeds, text_embeds=text_embeds, logits_per_text=logits,
                               logits_per_image=None if logits is None else logits.t())
