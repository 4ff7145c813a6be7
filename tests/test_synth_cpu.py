"""CPU: product-side synthetic generator == oracle-side generator (same names, shapes, values),
and the product presets agree with the oracle's."""
import torch

from fdm_amd import presets, synth
from oracle import weights as W


def test_presets_agree():
    for name, op in W.PRESETS.items():
        pp = presets.get(name)
        for k in ("d", "n_head", "n_layers", "ffn", "G", "c", "n_style", "n_emo", "audio_in", "pair", "pe", "period",
                  "K", "n_books", "V3", "vq_pre", "vq_out_bias", "latent_mish"):
            assert getattr(pp, k) == op[k], (name, k)


def test_generators_agree():
    for preset in ("vocaset_tiny", "mead_tiny", "mead"):
        a, b = synth.make_fdm_weights(preset), W.make_fdm_weights(preset)
        assert a.keys() == b.keys()
        assert all(torch.equal(a[k], b[k]) for k in a)
        ia, ib = synth.synth_inputs(preset, 2, 9, seed=3), W.synth_inputs(preset, 2, 9, seed=3)
        assert all(torch.equal(ia[k], ib[k]) for k in ia)
    a, b = synth.make_hubert_weights(1), W.make_hubert_weights(1)
    assert all(torch.equal(a[k], b[k]) for k in a) and a.keys() == b.keys()
    a, b = synth.make_wav2vec_weights(1), W.make_wav2vec_weights(1)
    assert all(torch.equal(a[k], b[k]) for k in a) and a.keys() == b.keys()
    a, b = synth.make_vq_weights("mead"), W.make_vq_weights("mead")
    assert all(torch.equal(a[k], b[k]) for k in a) and a.keys() == b.keys()
