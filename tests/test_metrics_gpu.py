"""GPU: evaluation metrics and the audio-feature resampling / frame_num crop, through the
C ABI, against the oracle and against the numbers the reference's own computer_metrix.py printed (metrics.json)."""
import numpy as np
import pytest
import torch

from fdm_amd import metrics, ops
from fdm_amd._lib import F32, FdmError
from fdm_amd.hubert import HubertPlan
from fdm_amd.modules import HubertModel, linear_interpolation
from oracle import hubert_oracle as HO
from oracle import metrics_oracle as MO
from oracle import weights as W
from tests.metrics_data import golden_metrics, sequences, write_dataset

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


@pytest.mark.parametrize("F_,V,R", [(1, 50, 7), (13, 6172, 300), (40, 23370, 1500), (300, 5023, 5023)])
def test_vertex_error_and_motion_std_vs_oracle(F_, V, R):
    tmpl, ss = MO.synth_sequences(F_ + V, 1, F_, V)
    gt, pred = ss[0]
    region = None if R == V else sorted(np.random.RandomState(R).choice(V, R, replace=False).tolist())
    r = metrics.vertex_error(gt, pred, region, DEV)
    sq = MO.region_sq_err(gt, pred, region)
    # per-frame maxima are bit-identical to numpy's float32 arithmetic
    assert np.array_equal(r["frame_max"].cpu().numpy(), sq.max(axis=1))
    assert rel(r["max"], MO.max_vertex_error(gt, pred, region)) < 1e-6
    assert rel(r["mean_sq"], MO.mean_sq_error(gt, pred, region)) < 1e-5
    if region is None:
        assert rel(r["mean_dist"], MO.mean_vertex_error(gt, pred)) < 1e-5
    reg = list(range(V)) if region is None else region
    if F_ > 1:
        assert rel(metrics.motion_std(gt, tmpl, region, DEV), MO.motion_std(gt, tmpl, reg)) < 2e-5
        assert rel(metrics.upper_face_dynamics_deviation(gt, pred, tmpl, region, DEV), MO.fdd(gt, pred, tmpl, reg)) < 2e-3


def test_metric_argument_errors():
    z = np.zeros((3, 10, 3), np.float32)
    with pytest.raises(FdmError):
        metrics.vertex_error(z, z[:2], None, DEV)
    with pytest.raises(FdmError):
        metrics.vertex_error(z, z, [10], DEV)
    with pytest.raises(FdmError):
        metrics.motion_std(z, np.zeros(9, np.float32), [0], DEV)


@pytest.mark.parametrize("dataset", ["vocaset", "BIWI"])
def test_evaluate_matches_the_numbers_the_reference_printed(dataset, tmp_path):
    """Same files, same flags as computer_metrix.py; the expected values were printed by the reference itself."""
    rec = golden_metrics()[dataset]
    write_dataset(rec, dataset, str(tmp_path))
    subj = " ".join(rec["subjects"])
    res = metrics.evaluate(str(tmp_path / "pred"), str(tmp_path / "gt"), str(tmp_path / "regions"), str(tmp_path / "templates.pkl"),
                           train_subjects=subj, dataset=dataset, device=DEV, verbose=False)
    assert res["frames"] == rec["frame_number"]
    for k in ("mean_vertex_error", "lip_vertex_error", "fdd", "abs_fdd"):      # 5 printed digits
        assert rel(res[k], rec[k]) < 6e-5, (k, res[k], rec[k])
    div = metrics.diversity(str(tmp_path / "pred"), subj, subj, dataset, DEV, verbose=False)
    assert rel(div, rec["diversity"]) < 6e-5


def test_mead_vertex_metrics_vs_oracle():
    _, ss = MO.synth_sequences(3, 1, 25, 5023)
    gt, pred = ss[0]
    rs = np.random.RandomState(0)
    face, lip, emo = [sorted(rs.choice(5023, n, replace=False).tolist()) for n in (1800, 250, 600)]
    m = metrics.mead_vertex_metrics(gt, pred, face, lip, emo, DEV)
    assert rel(m["FVE"], MO.max_vertex_error(gt, pred, face)) < 1e-6 and rel(m["LVE"], MO.max_vertex_error(gt, pred, lip)) < 1e-6
    assert rel(m["EME"], MO.mean_sq_error(gt, pred, emo)) < 1e-5 and rel(m["ALL"], MO.max_vertex_error(gt, pred)) < 1e-6


def test_linear_interpolation_vs_reference_golden(golden):
    g = golden("hubert_frames")
    for key in [k for k in g.files if k.startswith("interp_") and k.endswith("_x")]:
        To = int(key.split("_")[2])
        y = linear_interpolation(torch.from_numpy(g[key]).to(DEV), 50, 30, output_len=To)
        assert float((y.cpu() - torch.from_numpy(g[key[:-1] + "y"])).abs().max()) < 1e-6
    assert linear_interpolation(torch.zeros(1, 100, 4, device=DEV), 50, 30).shape == (1, 60, 4)
    with pytest.raises(FdmError):
        linear_interpolation(torch.zeros(1, 4, 4), 50, 30)


def test_hubert_frame_num_crops_before_the_encoder(golden):
    """models/hubert.py:97-98: the crop precedes the (non-causal) encoder, so it is not a slice of the uncropped output."""
    g = golden("hubert_frames")
    gen = torch.Generator().manual_seed(12)
    wav = HO.processor_normalize(torch.randn(32000, generator=gen) * 0.1)
    w2 = W.make_hubert_weights(2)
    plan = HubertPlan(w2, 2, F32, DEV)
    out = plan.forward(wav, frame_num=20)
    assert out.shape == (1, 40, 1024)
    assert float((out[0].cpu() - torch.from_numpy(g["out_L2_2s_fn20"])).abs().max()) < 1e-4
    assert float((plan.forward(wav)[0, :40] - out[0]).abs().max()) > 1e-3
    # drop-in class surface
    m = HubertModel(n_layers=2, dtype="fp32")
    m.load_state_dict(w2)
    o2 = m(wav.unsqueeze(0).to(DEV), "vocaset", frame_num=20).last_hidden_state
    assert torch.equal(o2, out)
    # build-defined optional 50 -> 30 fps branch against the oracle
    oi = plan.forward(wav, frame_num=59, interp_fps=(50, 30))
    ref = HO.hubert_forward_clip(w2, wav, 2, frame_num=59, interp_fps=(50, 30))
    assert oi.shape == (1, 59, 1024) and float((oi[0].cpu() - ref).abs().max()) < 1e-4
    assert plan.forward(wav, interp_fps=(50, 30)).shape[1] == int(99 / 50.0 * 30)
