"""Dataset presets of the FDM path (one dataclass per dataset, SURVEY.md section 5 'Config / flags').

Numbers come from the reference constructors and argparse defaults:
  VOCASET  models/fdm_vocaset.py:9,20-51,63     models/utils/config.py:64-80
  3D-MEAD  models/fdm_vqvae_mead.py:9,20-53,73-74  models/utils/config.py:4-20
  BIWI     models/fdm.py:10-48 (struct='Dec' + regroup x8: build-defined)  config.py:44-60
"""
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class Preset:
    name: str
    d: int            # feature_dim
    n_head: int
    n_layers: int
    ffn: int          # dim_feedforward = 2 * feature_dim
    G: int            # face_quan_num: latent vectors per frame
    c: int            # zquant_dim
    n_style: int
    n_emo: int
    audio_in: int     # input width of audio_extract.0
    pair: int         # HuBERT frames folded per latent frame
    pe: str           # 'periodic' | 'sinus'
    period: int       # ALiBi period
    K: int = 256      # codes per codebook
    n_books: int = 1  # emotion-sliced codebooks
    V3: int = 15069
    vq_pre: bool = False
    vq_out_bias: bool = True
    latent_mish: bool = True
    style_mish: bool = False
    max_len: int = 600     # init_biased_mask(max_seq_len=600), models/fdm_vocaset.py:44

    @property
    def head_dim(self):
        return self.d // self.n_head


VOCASET = Preset("vocaset", 1024, 8, 8, 2048, 16, 64, 8, 0, 1024, 1, "periodic", 30)
MEAD = Preset("mead", 512, 4, 8, 1024, 8, 64, 25, 7, 2048, 2, "sinus", 30, n_books=7, vq_pre=True, vq_out_bias=False)
BIWI = Preset("biwi", 1024, 4, 8, 2048, 8, 128, 6, 0, 1536, 2, "sinus", 25, V3=70110, vq_pre=True, vq_out_bias=False,
              latent_mish=False, style_mish=True)
# small structural twins used by tests (head_dim stays 128)
VOCASET_TINY = replace(VOCASET, name="vocaset_tiny", d=256, n_head=2, n_layers=2, ffn=512, c=16, V3=96)
MEAD_TINY = replace(MEAD, name="mead_tiny", d=256, n_head=2, n_layers=2, ffn=512, c=32, V3=96)

PRESETS = {p.name: p for p in (VOCASET, MEAD, BIWI, VOCASET_TINY, MEAD_TINY)}

VQ_HIDDEN, VQ_LAYERS, VQ_HEADS, VQ_FFN = 1024, 6, 8, 1536   # models/utils/config.py defaults


def get(name):
    if isinstance(name, Preset):
        return name
    return PRESETS[name]
