"""fp32 CPU restatement of the HuBERT-large audio encoder (oracle / test infrastructure only).

Restates models/hubert.py:75-146 (the reference's forward override) on top of the published
HuBERT architecture implemented by the third-party `transformers` package (pinned ==4.32.0 in
/root/reference/requirements.txt:255; 5.15.0 in this image -- same op graph for the
feat_extract_norm='layer' / do_stable_layer_norm=True configuration):
  HubertFeatureEncoder -> HubertFeatureProjection -> HubertEncoderStableLayerNorm.
Parity is pinned against the reference run in the build container (tests/golden/hubert_*.npz).
"""
import math

import torch
import torch.nn.functional as F

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
N_HEAD = 16
POS_K = 128
POS_GROUPS = 16
EPS = 1e-5


def num_frames(n_samples):
    """Frames after the conv stack, then the reference's even crop (models/hubert.py:95-96)."""
    n = n_samples
    for k, s in zip(CONV_KERNEL, CONV_STRIDE):
        n = (n - k) // s + 1
    return n - (n % 2)


def feature_extractor(w, wav, pre=""):
    """wav [n] -> [N', 512]: 7 x {Conv1d -> LayerNorm(channels) -> GELU(erf)} (HubertLayerNormConvLayer)."""
    h = wav.view(1, 1, -1)
    for i, s in enumerate(CONV_STRIDE):
        p = f"{pre}feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, w[p + "conv.weight"], w[p + "conv.bias"], stride=s)
        h = F.layer_norm(h.transpose(1, 2), (h.shape[1],), w[p + "layer_norm.weight"],
                         w[p + "layer_norm.bias"], EPS).transpose(1, 2)
        h = F.gelu(h)
    return h[0].transpose(0, 1)


def pos_conv_weight(w, pre=""):
    """weight_norm(dim=2): W = g * v / ||v||, norm over dims (0, 1) per kernel tap."""
    g = w[pre + "encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v = w[pre + "encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    return g * v / v.norm(2, dim=(0, 1), keepdim=True)


def encoder_layer(w, p, h):
    """HubertEncoderLayerStableLayerNorm.forward (pre-LN attention + pre-LN FFN)."""
    n, d = h.shape
    hd = d // N_HEAD
    x = F.layer_norm(h, (d,), w[p + "layer_norm.weight"], w[p + "layer_norm.bias"], EPS)
    q = F.linear(x, w[p + "attention.q_proj.weight"], w[p + "attention.q_proj.bias"])
    k = F.linear(x, w[p + "attention.k_proj.weight"], w[p + "attention.k_proj.bias"])
    v = F.linear(x, w[p + "attention.v_proj.weight"], w[p + "attention.v_proj.bias"])
    q = q.view(n, N_HEAD, hd).transpose(0, 1)
    k = k.view(n, N_HEAD, hd).transpose(0, 1)
    v = v.view(n, N_HEAD, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2)) * (hd ** -0.5)
    o = torch.bmm(torch.softmax(s, dim=-1), v).transpose(0, 1).reshape(n, d)
    h = h + F.linear(o, w[p + "attention.out_proj.weight"], w[p + "attention.out_proj.bias"])
    x = F.layer_norm(h, (d,), w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], EPS)
    x = F.gelu(F.linear(x, w[p + "feed_forward.intermediate_dense.weight"],
                        w[p + "feed_forward.intermediate_dense.bias"]))
    return h + F.linear(x, w[p + "feed_forward.output_dense.weight"],
                        w[p + "feed_forward.output_dense.bias"])


def linear_interpolation(features, input_fps, output_fps, output_len=None):
    """models/hubert.py:62-69 restated without F.interpolate: align_corners linear resampling of [B, T, C] along T.
    src = t * (T-1)/(T_out-1); i0 = floor(src); y = (1-l)*x[i0] + l*x[min(i0+1, T-1)], all in fp32."""
    B, T, C = features.shape
    if output_len is None:
        output_len = int(T / float(input_fps) * output_fps)
    scale = torch.tensor((T - 1) / (output_len - 1) if output_len > 1 else 0.0, dtype=torch.float32)
    src = scale * torch.arange(output_len, dtype=torch.float32)
    i0 = src.floor().long().clamp(max=T - 1)
    i1 = (i0 + 1).clamp(max=T - 1)
    l1 = (src - i0.float()).view(1, -1, 1)
    return (1.0 - l1) * features[:, i0] + l1 * features[:, i1]


def hubert_forward_clip(w, wav, n_layers=24, pre="", trace=None, frame_num=None, interp_fps=None):
    """wav [n] fp32 (already processor-normalised) -> last_hidden_state [N, 1024].
    frame_num: models/hubert.py:97-98 (crop of the conv features).  interp_fps=(in, out): build-defined optional
    branch: resample the conv features with linear_interpolation instead of the even crop."""
    f = feature_extractor(w, wav, pre)
    if interp_fps:
        f = linear_interpolation(f.unsqueeze(0), interp_fps[0], interp_fps[1], output_len=frame_num)[0]
    else:
        if f.shape[0] % 2 != 0:                       # models/hubert.py:95-96
            f = f[:-1]
        if frame_num and f.shape[0] > frame_num * 2:  # models/hubert.py:97-98
            f = f[: frame_num * 2]
    if trace is not None:
        trace["conv"] = f.clone()
    x = F.layer_norm(f, (f.shape[1],), w[pre + "feature_projection.layer_norm.weight"],
                     w[pre + "feature_projection.layer_norm.bias"], EPS)
    h = F.linear(x, w[pre + "feature_projection.projection.weight"],
                 w[pre + "feature_projection.projection.bias"])
    pc = F.conv1d(h.t().unsqueeze(0), pos_conv_weight(w, pre), w[pre + "encoder.pos_conv_embed.conv.bias"],
                  padding=POS_K // 2, groups=POS_GROUPS)[0, :, :-1]
    h = h + F.gelu(pc).t()
    if trace is not None:
        trace["posconv"] = h.clone()
    for l in range(n_layers):
        h = encoder_layer(w, f"{pre}encoder.layers.{l}.", h)
    return F.layer_norm(h, (h.shape[1],), w[pre + "encoder.layer_norm.weight"],
                        w[pre + "encoder.layer_norm.bias"], EPS)


def hubert_forward(w, wav, n_layers=24, pre="", frame_num=None, interp_fps=None):
    """wav [B, n] -> [B, N, 1024]; clips are independent."""
    return torch.stack([hubert_forward_clip(w, wav[b], n_layers, pre, frame_num=frame_num, interp_fps=interp_fps)
                        for b in range(wav.shape[0])])


def processor_normalize(wav, pad_seconds=0.0, sr=16000):
    """Wav2Vec2 feature-extractor default: (x - mean) / sqrt(var + 1e-7) (demo/demo_vocaset.py:84-85),
    then the demo's optional zero pad (demo/demo_vocaset.py:90)."""
    x = (wav - wav.mean()) / torch.sqrt(wav.var(unbiased=False) + 1e-7)
    if pad_seconds:
        x = torch.cat([x, torch.zeros(int(pad_seconds * sr))])
    return x
