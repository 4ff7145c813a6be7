"""CPU: the oracle restatement against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Tolerance 5e-5 max-abs on O(4) outputs (measured <= 1.5e-5);
schedule buffers, masks, PE tables and VQ indices are bit-exact."""
import numpy as np
import pytest
import torch

from oracle import fdm_oracle as FO
from oracle import hubert_oracle as HO
from oracle import vq_oracle as VO
from oracle import weights as W

TOL = 5e-5


def mad(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def test_schedule_bit_exact(golden):
    g = golden("schedule")
    ours = FO.schedule_buffers(1000)
    for k, v in ours.items():
        assert np.array_equal(v.numpy(), g[k]), k
    # known answers
    assert abs(float(ours["betas"][0]) - 4.12842237e-05) < 1e-12
    assert abs(float(ours["betas"][999]) - 0.999899983) < 1e-7
    assert abs(float(ours["sqrt_recip_alphas_cumprod"][999]) - 64166.3125) < 0.5
    assert abs(float(ours["posterior_log_variance_clipped"][0]) + 46.0517006) < 1e-5
    for steps in (3, 50, 100, 250):
        assert np.array_equal(np.array(FO.ddim_time_pairs(steps)), g[f"ddim_pairs_{steps}"])
    assert FO.ddim_time_pairs(50)[0] == (999, 979) and FO.ddim_time_pairs(50)[-1] == (19, -1)


def test_masks_and_pe_bit_exact(golden):
    g = golden("masks")
    rows = g["rows"].tolist()
    for (h, per) in ((8, 30), (4, 30), (4, 25), (2, 30)):
        m = FO.biased_mask(h, 600, per)
        assert np.array_equal(m[:, rows, :].numpy(), g[f"mask_{h}_{per}"])
    m = FO.biased_mask(8, 600, 30)
    assert float(m[0, 35, 5]) == -0.5 and float(m[0, 35, 6]) == 0.0 and float(m[0, 35, 36]) == float("-inf")
    ppe = FO.positional_table(1024, "periodic", 30, 630)
    assert np.array_equal(ppe[rows[:8] + [629]].numpy(), g["ppe_1024_rows"])
    pe = FO.positional_table(512, "sinus", 30, 600)
    assert np.array_equal(pe[rows].numpy(), g["pe_512_rows"])


@pytest.mark.parametrize("preset", ["vocaset_tiny", "mead_tiny", "vocaset", "mead"])
def test_fdm_step(golden, preset):
    g = golden(f"fdm_step_{preset}")
    w = W.make_fdm_weights(preset)
    for (L, t) in g["cases"].tolist():
        inp = W.synth_inputs(preset, 1, L, seed=100 + L)
        emo = inp["emo"][0] if "emo" in inp else None
        for folded in (False, True):
            out = FO.fdm_forward_clip(w, preset, inp["hub"][0], t, inp["x"][0], inp["style"][0], emo, folded)
            assert mad(out, g[f"x0_L{L}_t{t}"]) < TOL, (preset, L, t, folded)


@pytest.mark.parametrize("preset", ["vocaset_tiny", "mead"])
def test_chains(golden, preset):
    g = golden(f"chains_{preset}")
    w = W.make_fdm_weights(preset)
    L = int(g["L"])
    inp = W.synth_inputs(preset, 1, L, seed=7)
    emo = inp.get("emo")
    den = lambda x, t: FO.fdm_forward(w, preset, inp["hub"], t, x, inp["style"], emo, folded=True)
    for name in ("lo", "hi"):
        rec = []
        FO.p_sample_loop(den, inp["x"].clone(), torch.from_numpy(g[f"ddpm_{name}_noise"]),
                         g[f"ddpm_{name}_t"].tolist(), record=rec)
        assert mad(torch.stack(rec), g[f"ddpm_{name}_steps"]) < TOL
    if "ddim_3_final" in g:
        for steps in (3, 50):
            out = FO.ddim_sample(den, inp["x"].clone(), steps)
            assert mad(out, g[f"ddim_{steps}_final"]) < TOL


def test_cfg_mix(golden):
    g = golden("cfg_mead")
    w = W.make_fdm_weights("mead")
    inp = W.synth_inputs("mead", 1, int(g["L"]), seed=55)
    out = FO.fdm_forward_cfg(w, "mead", inp["hub"], int(g["t"]), inp["x"], inp["style"], inp["emo"], 2.5)[0]
    assert mad(out, g["mix"]) < TOL
    assert mad(FO.cfg_mix(torch.from_numpy(g["cond"]), torch.from_numpy(g["uncond"]), 2.5), g["mix"]) < 1e-6


def test_hubert(golden):
    g = golden("hubert")
    gen = torch.Generator().manual_seed(12)
    wav = HO.processor_normalize(torch.randn(32000, generator=gen) * 0.1)
    assert HO.num_frames(32000) == 98 and HO.num_frames(32080) == 100 and HO.num_frames(160000) == 498
    w2 = W.make_hubert_weights(2)
    tr = {}
    out = HO.hubert_forward_clip(w2, wav, 2, trace=tr)
    assert mad(tr["conv"], g["conv_2s"]) < 1e-5
    assert mad(out, g["out_L2_2s"]) < TOL
    w24 = W.make_hubert_weights(24)
    assert mad(HO.hubert_forward_clip(w24, wav, 24), g["out_L24_2s"]) < TOL


def test_audio_misc(golden):
    g = golden("audio_misc")
    assert mad(HO.processor_normalize(torch.from_numpy(g["wav"])), g["normalized"]) < 1e-6
    out = FO.adain(torch.from_numpy(g["adain_c"]), torch.from_numpy(g["adain_s"]))
    assert mad(out, g["adain_out"]) < 1e-6


def vq_case(preset, L, e):
    p = W.PRESETS[preset]
    w = W.make_vq_weights(preset)
    E = w["quantize.embedding.weight"]
    gen = torch.Generator().manual_seed(40 + L)
    z = torch.randn(1, L * p["G"], p["c"], generator=gen) * (1.5 / 256)
    base = e * 256 if p["n_books"] > 1 else 0
    z[0, 0] = E[base + 17]
    if z.shape[1] > 2:
        z[0, 1] = 0.5 * (E[base + 3] + E[base + 200])
        z[0, 2] = E[base + 255]
    emo = torch.eye(7)[e].unsqueeze(0) if p["n_books"] > 1 else None
    return w, z, emo


@pytest.mark.parametrize("preset,L,e", [("vocaset", 2, 0), ("vocaset", 5, 0), ("vocaset", 100, 0),
                                        ("mead", 5, 0), ("mead", 5, 6), ("mead", 12, 3), ("biwi", 5, 0)])
def test_vq(golden, preset, L, e):
    g = golden("vq")
    w, z, emo = vq_case(preset, L, e)
    zq, idx = VO.quant(w, preset, z, emo)
    key = f"{preset}_L{L}_e{e}"
    assert np.array_equal(idx.numpy().astype(np.int16), g[key + "_idx"])          # index path: bit-exact
    assert idx[0, 0] == 17 and (L * W.PRESETS[preset]["G"] < 3 or idx[2, 0] == 255)
    assert abs(float(zq.double().sum()) - float(g[key + "_zq_sum"])) < 1e-9
    dec = VO.decode(w, preset, zq)[0]
    if key + "_dec" in g:
        assert mad(dec, g[key + "_dec"]) < TOL
    else:
        assert mad(dec[:, ::16], g[key + "_dec_cols16"]) < TOL


def test_cfg1_e2e_hoisted_and_folded(golden):
    """cfg-1 (1 clip x 100 frames, DDIM 50, HuBERT-large) -- the reference ran it as written
    (HuBERT inside the loop); the oracle runs hoisted + folded and must agree."""
    g = golden("cfg1_e2e")
    gen = torch.Generator().manual_seed(1)
    wav = HO.processor_normalize(torch.randn(32080, generator=gen) * 0.1).unsqueeze(0)
    xT = torch.randn(1, 1600, 64, generator=gen)
    sid = torch.eye(8)[2:3]
    wd = W.make_fdm_weights("vocaset")
    hub = HO.hubert_forward(W.make_hubert_weights(24), wav, 24)
    den = lambda x, t: FO.fdm_forward(wd, "vocaset", hub, t, x, sid, None, folded=True)
    out = FO.ddim_sample(den, xT.clone(), 50)
    assert mad(out, g["final"]) < TOL


def test_wav2vec2_base(golden):
    """BIWI audio encoder (models/wav2vec.py) -- SURVEY.md section 8f rank 2."""
    from oracle import wav2vec_oracle as WO
    g = golden("wav2vec")
    gen = torch.Generator().manual_seed(22)
    wav = HO.processor_normalize(torch.randn(32000, generator=gen) * 0.1)
    assert mad(WO.wav2vec_forward_clip(W.make_wav2vec_weights(2), wav, 2), g["out_L2_2s"]) < TOL
    assert mad(WO.wav2vec_forward_clip(W.make_wav2vec_weights(12), wav, 12), g["out_L12_2s"]) < TOL


STATS_CASES = [("vocaset", 5, 0), ("vocaset", 100, 0), ("mead", 5, 0), ("mead", 5, 5), ("mead", 100, 0), ("mead", 100, 5), ("biwi", 5, 0), ("biwi", 100, 0)]


def vq_stats_case(preset, L, e):
    p = W.PRESETS[preset]
    gen = torch.Generator().manual_seed(140 + L)
    z = torch.randn(1, L * p["G"], p["c"], generator=gen) * (1.5 / 256)
    emo = torch.eye(7)[e].unsqueeze(0) if p["n_books"] > 1 else None
    return W.make_vq_weights(preset), z, emo


@pytest.mark.parametrize("preset,L,e", STATS_CASES)
def test_vq_quant_full_tuple(golden, preset, L, e):
    """emb_loss, perplexity and min_encodings of the reference's quant() (quantizer.py:46-61) as the reference returned them."""
    g = golden("vq_stats")
    w, z, emo = vq_stats_case(preset, L, e)
    key = f"{preset}_L{L}_e{e}"
    loss, perp, me = VO.quant_stats(w, preset, z, emo)
    assert float(loss) == pytest.approx(float(g[key + "_loss"]), rel=1e-6)
    assert float(perp) == pytest.approx(float(g[key + "_perplexity"]), rel=1e-6)
    assert np.array_equal(me.sum(0).numpy().astype(np.int32), g[key + "_hist"])
    assert np.array_equal(me.argmax(1).numpy().astype(np.int16), g[key + "_idx"][:, 0])


@pytest.mark.parametrize("preset", ["vocaset", "mead", "biwi"])
def test_vq_encode_round_trip(golden, preset):
    """VQ-VAE encoder: encode -> quant -> decode as the reference's stage-1 round trip."""
    g = golden("vq_encode")
    p = W.PRESETS[preset]
    w = W.make_vq_weights(preset, encoder=True)
    x = torch.randn(1, 10, p["V3"], generator=torch.Generator().manual_seed(60)) * 0.3
    emo = torch.eye(7)[5].unsqueeze(0) if p["n_books"] > 1 else None
    h = VO.encode(w, preset, x, emo)
    assert mad(h[0], g[f"{preset}_h"]) < TOL
    zq, idx = VO.quant(w, preset, h, emo)
    assert np.array_equal(idx.numpy().astype(np.int16), g[f"{preset}_idx"])
    assert mad(VO.decode(w, preset, zq)[0][:, ::16], g[f"{preset}_dec_cols16"]) < TOL


def test_hubert_frame_num_and_linear_interpolation(golden):
    """models/hubert.py:97-98 (frame_num crop before the encoder) and :62-69 (linear_interpolation)."""
    g = golden("hubert_frames")
    gen = torch.Generator().manual_seed(12)
    wav = HO.processor_normalize(torch.randn(32000, generator=gen) * 0.1)
    out = HO.hubert_forward_clip(W.make_hubert_weights(2), wav, 2, frame_num=20)
    assert out.shape == (40, 1024) and mad(out, g["out_L2_2s_fn20"]) < TOL
    for key in [k for k in g.files if k.startswith("interp_") and k.endswith("_x")]:
        _, T, To, C, _ = key.split("_")
        y = HO.linear_interpolation(torch.from_numpy(g[key]), 50, 30, output_len=int(To))
        assert mad(y, g[key[:-1] + "y"]) < 1e-6
    assert HO.linear_interpolation(torch.zeros(1, 100, 4), 50, 30).shape[1] == 60


@pytest.mark.parametrize("dataset", ["vocaset", "BIWI"])
def test_metrics_oracle_reproduces_the_numbers_the_reference_printed(dataset):
    """computer_metrix.py main(): the reference itself was run on this seeded dataset (make_golden.py g13)."""
    from oracle import metrics_oracle as MO
    from tests.metrics_data import golden_metrics, sequences
    rec = golden_metrics()[dataset]
    templates, seqs = sequences(rec)
    order = [(a, b) for a in rec["subjects"] for b in rec["sentences"]]
    gts = np.concatenate([seqs[k][0] for k in order])
    prs = np.concatenate([seqs[k][1] for k in order])
    assert gts.shape[0] == rec["frame_number"]
    fd = [MO.fdd(seqs[k][0], seqs[k][1], templates[k[0]], rec["upper"]) for k in order]
    got = dict(mean_vertex_error=MO.mean_vertex_error(gts, prs), lip_vertex_error=MO.max_vertex_error(gts, prs, rec["mouth"]),
               fdd=sum(fd) / len(fd), abs_fdd=sum(abs(v) for v in fd) / len(fd),
               diversity=float(np.mean([MO.mean_vertex_error(seqs[k][0], seqs[k][1]) for k in order])))
    for k, v in got.items():      # the reference prints 5 significant digits
        assert abs(v - rec[k]) <= 6e-5 * abs(rec[k]), (k, v, rec[k])
