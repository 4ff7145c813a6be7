"""CPU: the drop-in class surface -- import paths, constructor signatures, method names and state-dict
keys/shapes equal the reference's (tests/golden/state_keys.json was dumped from the reference modules)."""
import inspect
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "face-diffusion-model_amd", "dropin")


@pytest.fixture(scope="module")
def ref_keys():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))


@pytest.fixture(scope="module")
def dropin():
    sys.path.insert(0, DROPIN)
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[m]
    yield
    sys.path.remove(DROPIN)


def shapes(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def test_vocaset_diffusion_state_dict_matches_reference(dropin, ref_keys):
    from models.fdm_vocaset import FDM
    from video_diffusion_pytorch.diffusion_BIWI_encoder_decoder import GaussianDiffusion
    m = FDM(feature_dim=1024)
    d = GaussianDiffusion(m, timesteps=1000, loss_type="l2")
    assert shapes(d) == ref_keys["diffusion_vocaset"]
    # reference zero-initialises latent_decoder (models/fdm_vocaset.py:50-51)
    assert float(m.latent_decoder.weight.abs().max()) == 0.0
    assert [p for p in inspect.signature(FDM.__init__).parameters][1:5] == ["feature_dim", "n_head", "num_layers", "struct"]
    assert [p for p in inspect.signature(FDM.forward).parameters][1:] == ["audio", "t", "vertice", "id_one_hot"]
    for name in ("sample", "ddim_sample", "p_sample_loop", "p_sample", "p_mean_variance", "q_posterior",
                 "predict_noise_from_start", "q_sample"):
        assert hasattr(d, name)
    assert [p for p in inspect.signature(d.ddim_sample).parameters][:4] == ["audio", "latent_motion_shape", "id_one_hot", "steps"]
    # a checkpoint in the reference's layout loads ('model' key, strict=False as samples/sample_diffusion_vocaset.py:94-97)
    sd = {k: torch.zeros(v) for k, v in ref_keys["diffusion_vocaset"].items()}
    res = d.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert m._plan_stale
    # no CPU fallback
    from fdm_amd._lib import FdmError
    with pytest.raises(FdmError):
        m(torch.zeros(1, 4000), torch.zeros(1, dtype=torch.long), torch.zeros(1, 160, 64), torch.eye(8)[:1])


def test_mead_and_vq_state_dicts(dropin, ref_keys):
    from models.fdm_vqvae_mead import FDM
    from models.utils.config import biwi_vq_vae_args, vocaset_vq_vae_args, vq_vae_args
    from models.vq_vae_emotion import VQAutoEncoder as V2
    from models.vq_vae_vocaset import VQAutoEncoder as V1
    from models.vq_vae import VQAutoEncoder as V3
    assert shapes(FDM(feature_dim=512, audio_encoder=False)) == {k: v for k, v in ref_keys["fdm_mead"].items() if not k.startswith("audio_encoder.")}
    assert [p for p in inspect.signature(FDM.forward).parameters][1:6] == ["audio", "t", "vertice", "emotion_one_hot", "id_one_hot"]
    for name, V, args in (("vq_vocaset", V1, vocaset_vq_vae_args()), ("vq_mead", V2, vq_vae_args()), ("vq_biwi", V3, biwi_vq_vae_args())):
        ae = V(args)
        assert shapes(ae) == ref_keys[name], name                                        # encoder.* included
        ae.load_state_dict({k: torch.zeros(v) for k, v in ref_keys[name].items()})      # strict


def test_schedule_helpers_and_cli_flags(dropin):
    from fdm_amd import pipeline, schedule
    import numpy as np
    assert schedule.ddim_time_pairs(100)[0] == (999, 989) and schedule.ddim_time_pairs(100)[-1] == (9, -1)
    x = pipeline.processor_normalize(np.arange(100, dtype=np.float32), pad_seconds=1.0)
    assert x.shape == (16100,) and abs(float(x[:100].mean())) < 1e-6 and float(x[100:].max()) == 0.0


def test_metric_and_resampling_dropins_import_with_the_reference_names(dropin):
    """computer_metrix.py (main, compute_diversity), metric/metric.py arithmetic, models/hubert.py linear_interpolation."""
    import computer_metrix as cm
    from metric import metric as mm
    from models.hubert import HubertModel, linear_interpolation
    from models.wav2vec import linear_interpolation as li2
    assert callable(cm.main) and callable(cm.compute_diversity) and callable(mm.mead_vertex_metrics)
    assert list(inspect.signature(linear_interpolation).parameters) == ["features", "input_fps", "output_fps", "output_len"]
    assert li2 is linear_interpolation
    assert "frame_num" in inspect.signature(HubertModel.forward).parameters
    from fdm_amd._lib import FdmError
    with pytest.raises(FdmError):      # no CPU fallback
        linear_interpolation(torch.zeros(1, 4, 4), 50, 30)


def test_sampler_entry_points_and_file_names_match_what_the_metrics_read(dropin):
    """The three named samplers exist (samples/sample_diffusion_{vocaset,mead,biwi}.py) and write the reference's file names:
    VOCASET `<file>_condition_<conditioning subject>` (sample_diffusion_vocaset.py:61-62,86-88), MEAD / BIWI `<file>`
    (sample_diffusion_mead.py:86, sample_diffusion_biwi.py:78) -- the names computer_metrix.py:69-74,171-174 (and the
    evaluation drop-in, fdm_amd.metrics.pred_name) read back."""
    import importlib.util
    from fdm_amd import metrics
    sd = os.path.join(DROPIN, "samples")
    for ds in ("vocaset", "mead", "biwi"):
        src = open(os.path.join(sd, f"sample_diffusion_{ds}.py")).read()
        assert f'main("{ds}")' in src
    spec = importlib.util.spec_from_file_location("sample_diffusion", os.path.join(sd, "sample_diffusion.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    conds = mod.CONDITION_SUBJECTS["vocaset"]
    assert len(conds) == 8 and len(set(conds)) == 8 and all(c.startswith("FaceTalk_") and c.endswith("_TA") for c in conds)
    subject, sentence = "FaceTalk_170809_00138_TA", "sentence21"
    for i, c in enumerate(conds):
        assert mod.save_name("vocaset", f"{subject}_{sentence}.wav", i) == metrics.pred_name(subject, sentence, c)
    assert mod.save_name("biwi", "F1_e37.wav", 0) == metrics.pred_name("F1", "e37") == "F1_e37"
    assert mod.save_name("mead", "M003_front_angry_level_3_001.wav", 0) == "M003_front_angry_level_3_001"
    assert metrics.pred_name("F2", "e38", "F2", model="fdm") == "fdm_F2_e38_condition_F2"      # computer_metrix.py:69-71
    assert mod.SHIPPED_DDIM_STEPS == {"vocaset": 100, "biwi": 50, "mead": None}
