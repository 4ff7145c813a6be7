"""Import harness for the *reference* (read-only, /root/reference) -- container-only.

Used exclusively by tests/golden/make_golden.py to generate golden vectors.  Nothing
in here (and nothing from /root/reference) travels to the GPU box: the -m gpu tests,
smoke() and bench.py never import this module.

The shims follow SURVEY.md section 8c:
  (1) import transformers before installing stub modules,
  (2) stub modules for dead imports of the lucidrains trainer code,
  (3) HubertModel.from_pretrained -> random-init HuBERT-large config (eager attention),
  (4) a `str` second positional argument to HubertModel.forward is ignored (a17b),
  (5) sys.argv is reset before the argparse-as-config helpers run.
"""
import sys
import types

REF = "/root/reference"


def install(hubert_layers=24):
    import torch
    import transformers  # noqa: F401  (must precede the stubs)
    from transformers import HubertConfig

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ident = lambda *a, **k: None
    stub("einops_exts", check_shape=ident, rearrange_many=ident)
    stub("rotary_embedding_torch", RotaryEmbedding=object)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", ToTensor=object, ToPILImage=object,
                         Compose=object, Resize=object, CenterCrop=object,
                         RandomHorizontalFlip=object, Lambda=object)
    tv.utils = stub("torchvision.utils")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    stub("video_diffusion_pytorch.text", tokenize=ident, bert_embed=ident, BERT_MODEL_DIM=768)
    sys.argv = ["x"]

    import models.hubert as ref_hubert

    cfg_kwargs = dict(hidden_size=1024, num_hidden_layers=hubert_layers, num_attention_heads=16,
                      intermediate_size=4096, feat_extract_norm="layer", conv_bias=True,
                      do_stable_layer_norm=True, feat_proj_layer_norm=True,
                      attn_implementation="eager")

    def _from_pretrained(cls, *a, **k):
        return cls(HubertConfig(**cfg_kwargs))

    ref_hubert.HubertModel.from_pretrained = classmethod(_from_pretrained)
    _orig_fwd = ref_hubert.HubertModel.forward

    def _fwd(self, input_values, attention_mask=None, *a, **k):
        if isinstance(attention_mask, str):
            attention_mask = None
        return _orig_fwd(self, input_values, attention_mask, *a, **k)

    ref_hubert.HubertModel.forward = _fwd
    return ref_hubert
