"""fp32 CPU restatement of the (E)VQ-VAE quantise + decode path (oracle / test infrastructure only).

quant : models/lib/quantizer.py:35-64, models/vq_vae_emotion.py:221-252 (emotion-sliced codebook)
decode: models/vq_vae_vocaset.py:35-43,245-258, models/vq_vae_emotion.py:33-41,335-352,
        models/lib/base_models.py:37-87,138-174,286-301, models/utils/base_model_util.py:81-94
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .weights import PRESETS, VQ_HEADS, VQ_HIDDEN, VQ_LAYERS


def quant(w, preset, z, emo=None):
    """z [B, L*G, c] -> (z_q [B, c, L*G], idx [B*L*G, 1] int64).  d = sum z^2 + sum e^2 - 2 z e^T,
    first-min argmin, gather (quantizer.py:36-50, :63).  EVQ: codebook slice by argmax(one_hot)."""
    p = PRESETS[preset]
    E = w["quantize.embedding.weight"]
    B = z.shape[0]
    zq = torch.empty_like(z)
    idx = torch.empty(B, z.shape[1], dtype=torch.int64)
    for b in range(B):
        Eb = E
        if p["n_books"] > 1:
            pos = int(torch.argmax(emo[b]))                       # vq_vae_emotion.py:223
            Eb = E[pos * 256:(pos + 1) * 256]
        zf = z[b].reshape(-1, p["c"])
        d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(Eb ** 2, dim=1) \
            - 2 * torch.matmul(zf, Eb.t())
        i = torch.argmin(d, dim=1)
        idx[b] = i
        zq[b] = z[b] + (Eb[i] - z[b])       # straight-through form, quantizer.py:56 (rounds like the reference)
    return zq.permute(0, 2, 1).contiguous(), idx.reshape(-1, 1)


def quant_stats(w, preset, z, emo=None, beta=0.25):
    """The rest of VectorQuantizer.forward's tuple for a B = 1 call (quantizer.py:46-61; EVQ vq_vae_emotion.py:232-249):
    min_encodings [R, 256] one-hot, loss = beta * mean((z_q - z)^2) + mean((z_q - z)^2), perplexity =
    exp(-sum(e_mean * log(e_mean + 1e-10))), e_mean = mean(min_encodings, dim=0).  -> (loss, perplexity, min_encodings)."""
    p = PRESETS[preset]
    E = w["quantize.embedding.weight"]
    Eb = E
    if p["n_books"] > 1:
        pos = int(torch.argmax(emo.reshape(-1, emo.shape[-1])[0]))
        Eb = E[pos * 256:(pos + 1) * 256]
    _, idx = quant(w, preset, z, emo)
    me = torch.zeros(idx.shape[0], 256)
    me.scatter_(1, idx, 1)
    z_q = torch.matmul(me, Eb).view(z.shape)
    loss = beta * torch.mean((z_q - z) ** 2) + torch.mean((z_q - z) ** 2)
    e_mean = torch.mean(me, dim=0)
    return loss, torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10))), me


def gelu_tanh(x):
    """models/utils/base_model_util.py:81-94."""
    cdf = 0.5 * (1.0 + torch.tanh((np.sqrt(2 / np.pi) * (x + 0.044715 * torch.pow(x, 3)))))
    return x * cdf


def decode_clip(w, preset, zq, pe_index=0, trace=None):
    """zq [c, L*G] (one clip of quant()'s output) -> vertices offsets [L, V3].

    pe_index: the reference indexes the positional table by *batch* position
    (models/lib/base_models.py:300); bs = 1 usage == pe[0] for every clip."""
    p = PRESETS[preset]
    G, c, d, H = p["G"], p["c"], VQ_HIDDEN, VQ_HEADS
    x = zq.t().reshape(-1, G * c)                                  # vq_vae_vocaset.py:37-39 -> [L, G*c]
    L = x.shape[0]
    if p["vq_pre"]:                                                # vq_vae_emotion.py:338-340
        x = F.linear(x, w["decoder.decoder_linear_embedding_pre.net.weight"],
                     w["decoder.decoder_linear_embedding_pre.net.bias"])
    xc = x.t().unsqueeze(0)                                        # [1, 1024, L]
    xc = F.conv1d(F.pad(xc, (2, 2), mode="replicate"), w["decoder.expander.0.0.weight"],
                  w["decoder.expander.0.0.bias"])
    xc = F.leaky_relu(xc, 0.2)
    xc = F.instance_norm(xc, eps=1e-5)                             # affine=False
    h = xc[0].t()                                                  # [L, 1024]
    if trace is not None:
        trace["expander"] = h.clone()
    h = F.linear(h, w["decoder.decoder_linear_embedding.net.weight"],
                 w["decoder.decoder_linear_embedding.net.bias"])
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe = torch.zeros(d)
    pe[0::2] = torch.sin(pe_index * div)
    pe[1::2] = torch.cos(pe_index * div)
    h = h + pe                                                     # base_models.py:300
    hd = d // H
    for l in range(VQ_LAYERS):
        a = f"decoder.decoder_transformer.net.{2 * l}.fn."
        m = f"decoder.decoder_transformer.net.{2 * l + 1}.fn."
        x = F.layer_norm(h, (d,), w[a + "norm.weight"], w[a + "norm.bias"], 1e-5)
        qkv = F.linear(x, w[a + "fn.to_qkv.weight"])               # "b n (qkv h d) -> qkv b h n d"
        q, k, v = [t.view(L, H, hd).transpose(0, 1) for t in qkv.split(d, dim=1)]
        s = torch.bmm(q, k.transpose(1, 2)) * (d ** -0.5)          # scale = hidden^-0.5 (:144)
        o = torch.bmm(torch.softmax(s, dim=-1), v).transpose(0, 1).reshape(L, d)
        h = h + F.linear(o, w[a + "fn.to_out.weight"], w[a + "fn.to_out.bias"])
        x = F.layer_norm(h, (d,), w[m + "norm.weight"], w[m + "norm.bias"], 1e-5)
        x = gelu_tanh(F.linear(x, w[m + "fn.l1.weight"], w[m + "fn.l1.bias"]))
        h = h + F.linear(x, w[m + "fn.l2.weight"], w[m + "fn.l2.bias"])
    return F.linear(h, w["decoder.vertice_map_reverse.weight"],
                    w.get("decoder.vertice_map_reverse.bias"))


def _transformer(w, prefix, h):
    """6 pre-LN blocks of models/lib/base_models.py:177-227 (attention scale hidden^-0.5, tanh-GELU MLP)."""
    d, H = VQ_HIDDEN, VQ_HEADS
    L, hd = h.shape[0], d // H
    for l in range(VQ_LAYERS):
        a = f"{prefix}.net.{2 * l}.fn."
        m = f"{prefix}.net.{2 * l + 1}.fn."
        x = F.layer_norm(h, (d,), w[a + "norm.weight"], w[a + "norm.bias"], 1e-5)
        q, k, v = [t.view(L, H, hd).transpose(0, 1) for t in F.linear(x, w[a + "fn.to_qkv.weight"]).split(d, dim=1)]
        o = torch.bmm(torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (d ** -0.5), dim=-1), v).transpose(0, 1).reshape(L, d)
        h = h + F.linear(o, w[a + "fn.to_out.weight"], w[a + "fn.to_out.bias"])
        x = F.layer_norm(h, (d,), w[m + "norm.weight"], w[m + "norm.bias"], 1e-5)
        h = h + F.linear(gelu_tanh(F.linear(x, w[m + "fn.l1.weight"], w[m + "fn.l1.bias"])), w[m + "fn.l2.weight"], w[m + "fn.l2.bias"])
    return h


def encode_clip(w, preset, x, emo=None):
    """x [L, V3] (vertices minus template) -> latent [L*G, c]
    (VQAutoEncoder.encode + TransformerEncoder.forward: models/vq_vae_vocaset.py:23-28,134-191,
    models/vq_vae_emotion.py:20-26,130-196; pe[0] as in decode)."""
    p = PRESETS[preset]
    d = VQ_HIDDEN
    h = F.leaky_relu(F.linear(x, w["encoder.vertice_mapping.0.weight"], w["encoder.vertice_mapping.0.bias"]), 0.2)
    if p["n_books"] > 1:
        h = h + F.leaky_relu(F.linear(emo, w["encoder.emotion_mapping.0.weight"], w["encoder.emotion_mapping.0.bias"]), 0.2)
    xc = F.conv1d(F.pad(h.t().unsqueeze(0), (2, 2), mode="replicate"), w["encoder.squasher.0.0.weight"], w["encoder.squasher.0.0.bias"])
    h = F.instance_norm(F.leaky_relu(xc, 0.2), eps=1e-5)[0].t()
    h = F.linear(h, w["encoder.encoder_linear_embedding.net.weight"], w["encoder.encoder_linear_embedding.net.bias"])
    pe = torch.zeros(d)
    pe[1::2] = 1.0
    h = _transformer(w, "encoder.encoder_transformer", h + pe)
    if p["vq_pre"]:
        h = F.linear(h, w["encoder.encoder_linear_embedding_post.net.weight"], w["encoder.encoder_linear_embedding_post.net.bias"])
    return h.reshape(-1, p["c"])


def encode(w, preset, x, emo=None):
    return torch.stack([encode_clip(w, preset, x[b], None if emo is None else emo[b]) for b in range(x.shape[0])])


def decode(w, preset, zq):
    """zq [B, c, L*G] -> [B, L, V3]; every clip decoded as a bs = 1 reference call (pe[0])."""
    return torch.stack([decode_clip(w, preset, zq[b]) for b in range(zq.shape[0])])
