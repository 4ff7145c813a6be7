"""fp32 CPU restatement of the wav2vec2-base audio encoder used by the BIWI denoiser (oracle / test
infrastructure only).

Restates models/wav2vec.py:69-143 (forward override: conv features -> drop last frame if odd (:100-101) ->
feature projection -> encoder) over the published wav2vec 2.0 BASE architecture implemented by the third-party
`transformers` package (pinned ==4.32.0; 5.15.0 here): Wav2Vec2FeatureEncoder with feat_extract_norm='group'
(GroupNorm(512, 512) on the first conv only, no conv bias), Wav2Vec2FeatureProjection, Wav2Vec2Encoder
(LayerNorm before the stack, post-LN layers).  Pinned against the reference run in the build container
(tests/golden/wav2vec.npz)."""
import torch
import torch.nn.functional as F

from .hubert_oracle import CONV_STRIDE, POS_GROUPS, POS_K, pos_conv_weight

N_HEAD, EPS = 12, 1e-5


def feature_extractor(w, wav, pre=""):
    h = wav.view(1, 1, -1)
    for i, s in enumerate(CONV_STRIDE):
        p = f"{pre}feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, w[p + "conv.weight"], None, stride=s)
        if i == 0:
            h = F.group_norm(h, h.shape[1], w[p + "layer_norm.weight"], w[p + "layer_norm.bias"], EPS)
        h = F.gelu(h)
    return h[0].transpose(0, 1)


def encoder_layer(w, p, h):
    """Wav2Vec2EncoderLayer.forward (post-LN)."""
    n, d = h.shape
    hd = d // N_HEAD
    q = F.linear(h, w[p + "attention.q_proj.weight"], w[p + "attention.q_proj.bias"]).view(n, N_HEAD, hd).transpose(0, 1)
    k = F.linear(h, w[p + "attention.k_proj.weight"], w[p + "attention.k_proj.bias"]).view(n, N_HEAD, hd).transpose(0, 1)
    v = F.linear(h, w[p + "attention.v_proj.weight"], w[p + "attention.v_proj.bias"]).view(n, N_HEAD, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2)) * (hd ** -0.5)
    o = torch.bmm(torch.softmax(s, dim=-1), v).transpose(0, 1).reshape(n, d)
    h = h + F.linear(o, w[p + "attention.out_proj.weight"], w[p + "attention.out_proj.bias"])
    h = F.layer_norm(h, (d,), w[p + "layer_norm.weight"], w[p + "layer_norm.bias"], EPS)
    f = F.linear(F.gelu(F.linear(h, w[p + "feed_forward.intermediate_dense.weight"], w[p + "feed_forward.intermediate_dense.bias"])),
                 w[p + "feed_forward.output_dense.weight"], w[p + "feed_forward.output_dense.bias"])
    return F.layer_norm(h + f, (d,), w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], EPS)


def wav2vec_forward_clip(w, wav, n_layers=12, pre=""):
    """wav [n] fp32 (processor-normalised) -> last_hidden_state [N, 768]."""
    f = feature_extractor(w, wav, pre)
    if f.shape[0] % 2 != 0:                       # models/wav2vec.py:100-101
        f = f[:-1]
    x = F.layer_norm(f, (f.shape[1],), w[pre + "feature_projection.layer_norm.weight"],
                     w[pre + "feature_projection.layer_norm.bias"], EPS)
    h = F.linear(x, w[pre + "feature_projection.projection.weight"], w[pre + "feature_projection.projection.bias"])
    pc = F.conv1d(h.t().unsqueeze(0), pos_conv_weight(w, pre), w[pre + "encoder.pos_conv_embed.conv.bias"],
                  padding=POS_K // 2, groups=POS_GROUPS)[0, :, :-1]
    h = h + F.gelu(pc).t()
    h = F.layer_norm(h, (h.shape[1],), w[pre + "encoder.layer_norm.weight"], w[pre + "encoder.layer_norm.bias"], EPS)
    for l in range(n_layers):
        h = encoder_layer(w, f"{pre}encoder.layers.{l}.", h)
    return h


def wav2vec_forward(w, wav, n_layers=12, pre=""):
    return torch.stack([wav2vec_forward_clip(w, wav[b], n_layers, pre) for b in range(wav.shape[0])])
