"""Deterministic, name-keyed synthetic weights (oracle / test infrastructure).

There are no checkpoints in this environment, so every parity case uses random-init
weights.  To make the weights reproducible on any box *without* depending on module
construction order, each tensor is drawn from its own `torch.Generator` seeded by
crc32(name) ^ seed.  Names and shapes are the reference's state-dict names (SURVEY.md section 5;
the compatibility surface: models/fdm_vocaset.py:17-51, models/fdm_vqvae_mead.py:17-53,
models/vq_vae_vocaset.py:194-243, models/vq_vae_emotion.py:279-333, transformers HubertModel).
tests/golden/make_golden.py loads these dicts into the *reference* modules with
strict=True, which also pins the name/shape surface.
"""
import math
import zlib

import torch

# ---------------------------------------------------------------------------------------------
# presets (values from models/fdm_vocaset.py:9,45, models/fdm_vqvae_mead.py:9,46, models/fdm.py:10,
# models/utils/config.py:4-20,44-60,64-80)
# ---------------------------------------------------------------------------------------------
PRESETS = {
    "vocaset": dict(d=1024, n_head=8, n_layers=8, ffn=2048, G=16, c=64, n_style=8, n_emo=0,
                    audio_in=1024, pair=1, pe="periodic", period=30, K=256, n_books=1,
                    V3=15069, vq_pre=False, vq_out_bias=True, latent_mish=True),
    "mead": dict(d=512, n_head=4, n_layers=8, ffn=1024, G=8, c=64, n_style=25, n_emo=7,
                 audio_in=2048, pair=2, pe="sinus", period=30, K=256, n_books=7,
                 V3=15069, vq_pre=True, vq_out_bias=False, latent_mish=True),
    # BIWI denoiser semantics are build-defined: struct='Dec', regroup x8.
    "biwi": dict(d=1024, n_head=4, n_layers=8, ffn=2048, G=8, c=128, n_style=6, n_emo=0,
                 audio_in=1536, pair=2, pe="sinus", period=25, K=256, n_books=1,
                 V3=70110, vq_pre=True, vq_out_bias=False, latent_mish=False),
}

# tiny presets: same structure, small dims -> fixtures can carry everything, CPU tests are fast
# (the reference hard-codes G = 16 / 8 in forward, so tiny presets shrink c instead of G)
PRESETS["vocaset_tiny"] = dict(PRESETS["vocaset"], d=256, n_head=2, n_layers=2, ffn=512, G=16, c=16,
                               audio_in=1024, V3=96)
PRESETS["mead_tiny"] = dict(PRESETS["mead"], d=256, n_head=2, n_layers=2, ffn=512, G=8, c=32,
                            audio_in=2048, V3=96)

VQ_HIDDEN = 1024       # models/utils/config.py: hidden_size
VQ_LAYERS = 6          # num_hidden_layers
VQ_HEADS = 8           # num_attention_heads
VQ_FFN = 1536          # intermediate_size


def fdm_shapes(preset):
    """Denoiser parameters excluding `audio_encoder.*` (models/fdm_vocaset.py:20-51)."""
    p = PRESETS[preset]
    d, ffn = p["d"], p["ffn"]
    s = {
        "audio_extract.0.weight": (d, p["audio_in"]), "audio_extract.0.bias": (d,),
        "audio_extract.2.weight": (d, d), "audio_extract.2.bias": (d,),
        "time_embedd.0.weight": (d, 1000), "time_embedd.0.bias": (d,),
        "style_embedd.weight": (d, p["n_style"]), "style_embedd.bias": (d,),
        "latent_encoder.0.weight": (d, d), "latent_encoder.0.bias": (d,),
        "latent_decoder.weight": (d, d), "latent_decoder.bias": (d,),
    }
    if p["n_emo"]:
        s["emotion_embedd.weight"] = (d, p["n_emo"])
        s["emotion_embedd.bias"] = (d,)
    for l in range(p["n_layers"]):
        pre = f"transformer_decoder.layers.{l}."
        s[pre + "self_attn.in_proj_weight"] = (3 * d, d)
        s[pre + "self_attn.in_proj_bias"] = (3 * d,)
        s[pre + "self_attn.out_proj.weight"] = (d, d)
        s[pre + "self_attn.out_proj.bias"] = (d,)
        s[pre + "multihead_attn.in_proj_weight"] = (3 * d, d)
        s[pre + "multihead_attn.in_proj_bias"] = (3 * d,)
        s[pre + "multihead_attn.out_proj.weight"] = (d, d)
        s[pre + "multihead_attn.out_proj.bias"] = (d,)
        s[pre + "linear1.weight"] = (ffn, d)
        s[pre + "linear1.bias"] = (ffn,)
        s[pre + "linear2.weight"] = (d, ffn)
        s[pre + "linear2.bias"] = (d,)
        for n in ("norm1", "norm2", "norm3"):
            s[pre + n + ".weight"] = (d,)
            s[pre + n + ".bias"] = (d,)
    return s


def hubert_shapes(n_layers=24, hidden=1024, ffn=4096, conv_dim=512):
    """transformers HubertModel (feat_extract_norm='layer', stable layer norm) parameter names."""
    kern = (10, 3, 3, 3, 3, 2, 2)
    s = {"masked_spec_embed": (hidden,)}
    for i, k in enumerate(kern):
        pre = f"feature_extractor.conv_layers.{i}."
        s[pre + "conv.weight"] = (conv_dim, 1 if i == 0 else conv_dim, k)
        s[pre + "conv.bias"] = (conv_dim,)
        s[pre + "layer_norm.weight"] = (conv_dim,)
        s[pre + "layer_norm.bias"] = (conv_dim,)
    s["feature_projection.layer_norm.weight"] = (conv_dim,)
    s["feature_projection.layer_norm.bias"] = (conv_dim,)
    s["feature_projection.projection.weight"] = (hidden, conv_dim)
    s["feature_projection.projection.bias"] = (hidden,)
    s["encoder.pos_conv_embed.conv.bias"] = (hidden,)
    s["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, 128)
    s["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (hidden, hidden // 16, 128)
    s["encoder.layer_norm.weight"] = (hidden,)
    s["encoder.layer_norm.bias"] = (hidden,)
    for l in range(n_layers):
        pre = f"encoder.layers.{l}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[pre + f"attention.{n}.weight"] = (hidden, hidden)
            s[pre + f"attention.{n}.bias"] = (hidden,)
        s[pre + "layer_norm.weight"] = (hidden,)
        s[pre + "layer_norm.bias"] = (hidden,)
        s[pre + "feed_forward.intermediate_dense.weight"] = (ffn, hidden)
        s[pre + "feed_forward.intermediate_dense.bias"] = (ffn,)
        s[pre + "feed_forward.output_dense.weight"] = (hidden, ffn)
        s[pre + "feed_forward.output_dense.bias"] = (hidden,)
        s[pre + "final_layer_norm.weight"] = (hidden,)
        s[pre + "final_layer_norm.bias"] = (hidden,)
    return s


def wav2vec_shapes(n_layers=12, hidden=768, ffn=3072, conv_dim=512):
    """transformers Wav2Vec2Model (base: feat_extract_norm='group', no conv bias, post-LN encoder) parameter names."""
    s = hubert_shapes(n_layers, hidden, ffn, conv_dim)
    for i in range(7):
        pre = f"feature_extractor.conv_layers.{i}."
        del s[pre + "conv.bias"]
        if i > 0:
            del s[pre + "layer_norm.weight"]
            del s[pre + "layer_norm.bias"]
    return s


def vq_shapes(preset, hidden=VQ_HIDDEN, n_layers=VQ_LAYERS, ffn=VQ_FFN):
    """Quantizer + decoder parameters (encoder.* is training-only, out of scope)."""
    p = PRESETS[preset]
    s = {"quantize.embedding.weight": (p["K"] * p["n_books"], p["c"]),
         "decoder.expander.0.0.weight": (hidden, hidden, 5),
         "decoder.expander.0.0.bias": (hidden,),
         "decoder.decoder_linear_embedding.net.weight": (hidden, hidden),
         "decoder.decoder_linear_embedding.net.bias": (hidden,),
         "decoder.vertice_map_reverse.weight": (p["V3"], hidden)}
    if p["vq_out_bias"]:
        s["decoder.vertice_map_reverse.bias"] = (p["V3"],)
    if p["vq_pre"]:
        s["decoder.decoder_linear_embedding_pre.net.weight"] = (hidden, p["G"] * p["c"])
        s["decoder.decoder_linear_embedding_pre.net.bias"] = (hidden,)
    for l in range(n_layers):
        a = f"decoder.decoder_transformer.net.{2 * l}.fn."
        m = f"decoder.decoder_transformer.net.{2 * l + 1}.fn."
        s[a + "norm.weight"] = (hidden,)
        s[a + "norm.bias"] = (hidden,)
        s[a + "fn.to_qkv.weight"] = (3 * hidden, hidden)
        s[a + "fn.to_out.weight"] = (hidden, hidden)
        s[a + "fn.to_out.bias"] = (hidden,)
        s[m + "norm.weight"] = (hidden,)
        s[m + "norm.bias"] = (hidden,)
        s[m + "fn.l1.weight"] = (ffn, hidden)
        s[m + "fn.l1.bias"] = (ffn,)
        s[m + "fn.l2.weight"] = (hidden, ffn)
        s[m + "fn.l2.bias"] = (hidden,)
    return s


def vq_encoder_shapes(preset, hidden=VQ_HIDDEN, n_layers=VQ_LAYERS, ffn=VQ_FFN):
    """encoder.* parameters (models/vq_vae_vocaset.py:134-191, models/vq_vae_emotion.py:130-196, models/vq_vae.py)."""
    p = PRESETS[preset]
    s = {"encoder.vertice_mapping.0.weight": (hidden, p["V3"]), "encoder.vertice_mapping.0.bias": (hidden,),
         "encoder.squasher.0.0.weight": (hidden, hidden, 5), "encoder.squasher.0.0.bias": (hidden,),
         "encoder.encoder_linear_embedding.net.weight": (hidden, hidden),
         "encoder.encoder_linear_embedding.net.bias": (hidden,)}
    if p["n_books"] > 1:
        s["encoder.emotion_mapping.0.weight"] = (hidden, 7)
        s["encoder.emotion_mapping.0.bias"] = (hidden,)
    if p["vq_pre"]:
        s["encoder.encoder_linear_embedding_post.net.weight"] = (p["G"] * p["c"], hidden)
        s["encoder.encoder_linear_embedding_post.net.bias"] = (p["G"] * p["c"],)
    for l in range(n_layers):
        a = f"encoder.encoder_transformer.net.{2 * l}.fn."
        m = f"encoder.encoder_transformer.net.{2 * l + 1}.fn."
        s[a + "norm.weight"] = (hidden,)
        s[a + "norm.bias"] = (hidden,)
        s[a + "fn.to_qkv.weight"] = (3 * hidden, hidden)
        s[a + "fn.to_out.weight"] = (hidden, hidden)
        s[a + "fn.to_out.bias"] = (hidden,)
        s[m + "norm.weight"] = (hidden,)
        s[m + "norm.bias"] = (hidden,)
        s[m + "fn.l1.weight"] = (ffn, hidden)
        s[m + "fn.l1.bias"] = (ffn,)
        s[m + "fn.l2.weight"] = (hidden, ffn)
        s[m + "fn.l2.bias"] = (hidden,)
    return s


def _is_norm(name):
    return ("norm" in name) and ("to_" not in name)


_MADE = {}      # (shapes, seed, prefix) -> tensors: the suites ask for the same few weight sets dozens of times (HuBERT-large is 315 M values)


def make_weights(shapes, seed=0, prefix=""):
    """name -> fp32 tensor.  Matrices ~ N(0, 1/fan_in) so activations stay O(1); biases N(0, 0.02);
    LayerNorm weights 1 + N(0, 0.1); codebooks U(-1/K, 1/K) as in models/lib/quantizer.py:33.
    (Memoised per process: callers get a fresh dict over shared, read-only tensors.)"""
    key = (tuple((k, tuple(v)) for k, v in shapes.items()), seed, prefix)
    if key in _MADE:
        return dict(_MADE[key])
    out = _make_weights(shapes, seed, prefix)
    _MADE[key] = out
    return dict(out)


def _make_weights(shapes, seed, prefix):
    out = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        if name.endswith("embedding.weight") and name.startswith("quantize"):
            k = shape[0]
            k = 256 if k % 256 == 0 else k
            w = (torch.rand(shape, generator=g) * 2 - 1) / k
        elif name.endswith("original0"):
            w = 1.0 + 0.25 * torch.rand(shape, generator=g)
        elif _is_norm(name) and name.endswith("weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif _is_norm(name) and name.endswith("bias"):
            w = 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("bias") or name.endswith("in_proj_bias") or len(shape) == 1:
            w = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            w = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        out[prefix + name] = w.float().contiguous()
    return out


def make_fdm_weights(preset, seed=0):
    return make_weights(fdm_shapes(preset), seed)


def make_hubert_weights(n_layers=24, seed=0, prefix=""):
    return make_weights(hubert_shapes(n_layers), seed, prefix)


def make_wav2vec_weights(n_layers=12, seed=0, prefix=""):
    return make_weights(wav2vec_shapes(n_layers), seed, prefix)


def make_vq_weights(preset, seed=0, encoder=False):
    w = make_weights(vq_shapes(preset), seed)
    if encoder:
        w.update(make_weights(vq_encoder_shapes(preset), seed))
    return w


def synth_inputs(preset, B, L, seed=1, audio_frames=None):
    """Seeded synthetic inputs: hubert features, x_T, one-hots.

    Returns a dict of CPU tensors; `hub` stands in for HubertModel(audio).last_hidden_state
    ([B, N, 1024]) so that denoiser cases do not depend on the audio encoder."""
    p = PRESETS[preset]
    g = torch.Generator().manual_seed(seed)
    n = audio_frames if audio_frames is not None else L * p["pair"]
    hub = torch.randn(B, n, 1024, generator=g)
    x = torch.randn(B, L * p["G"], p["c"], generator=g)
    sid = torch.eye(p["n_style"])[torch.arange(B) % p["n_style"]]
    out = dict(hub=hub, x=x, style=sid)
    if p["n_emo"]:
        out["emo"] = torch.eye(p["n_emo"])[torch.arange(B) % p["n_emo"]]
    return out
