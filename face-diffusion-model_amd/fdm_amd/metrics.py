"""On-device evaluation metrics of the reference: lip / face vertex error,
mean vertex error, emotion mean error, upper-face dynamics deviation (FDD) and diversity.

Mirrors computer_metrix.py:6-136 (BIWI / VOCASET evaluation, `main`), :139-194 (`compute_diversity`) and the
vertex-error block of metric/metric.py:115-138 (3D-MEAD; its ground truth needs the FLAME model, which is
outside this path, so only the arithmetic on vertex arrays is provided).  The reductions run in
libfdm_hip.so (fdm_op_vertex_err, fdm_op_motion_std); numpy/pickle are used for file IO only."""
import argparse
import os
import pickle

import numpy as np
import torch

from . import ops
from ._lib import FdmError


def _dev(a, device, dtype=torch.float32):
    t = torch.as_tensor(np.ascontiguousarray(a)) if not torch.is_tensor(a) else a
    return t.to(device=device, dtype=dtype).contiguous()


def _region(region, V, device):
    if region is None:
        return None, V
    r = _dev(np.asarray(region, dtype=np.int64), device, torch.int32)
    if r.numel() == 0 or int(r.min()) < 0 or int(r.max()) >= V:
        raise FdmError("vertex region is empty or out of range")
    return r, r.numel()


def vertex_error(gt, pred, region=None, device="cuda:0"):
    """gt, pred [F, V, 3] -> dict(max=mean_f max_r d2, mean_sq=mean d2, mean_dist=mean |d|, frame_max=[F] tensor),
    d2 = |gt - pred|^2 over the region's vertices (all vertices when region is None)."""
    g, p = _dev(gt, device), _dev(pred, device)
    if g.shape != p.shape or g.dim() != 3 or g.shape[2] != 3:
        raise FdmError(f"vertex_error expects two [F, V, 3] arrays, got {tuple(g.shape)} and {tuple(p.shape)}")
    F, V = g.shape[0], g.shape[1]
    r, R = _region(region, V, device)
    fmax = torch.empty(F, device=device)
    fsum = torch.empty(2 * F, device=device, dtype=torch.float64)
    out = torch.empty(3, device=device, dtype=torch.float64)
    ops.vertex_err(g, p, r, R, F, V, fmax, fsum, out)
    o = out.cpu()
    return dict(max=float(o[0]), mean_sq=float(o[1]), mean_dist=float(o[2]), frame_max=fmax)


def motion_std(verts, template, region, device="cuda:0"):
    """mean over the region of the per-vertex std over frames of |verts - template|^2 (computer_metrix.py:95-105)."""
    v = _dev(verts, device)
    F, V = v.shape[0], v.shape[1]
    t = _dev(np.asarray(template).reshape(-1, 3) if not torch.is_tensor(template) else template.reshape(-1, 3), device)
    if t.shape[0] != V:
        raise FdmError(f"template has {t.shape[0]} vertices, sequence has {V}")
    r, R = _region(region, V, device)
    partial = torch.empty(2 * min(F, 64) * R, device=device, dtype=torch.float64)
    out = torch.empty(1, device=device, dtype=torch.float64)
    ops.motion_std(v, t, r, R, F, V, partial, out)
    return float(out.cpu()[0])


def lip_vertex_error(gt, pred, mouth_map, device="cuda:0"):
    """'Lip Vertex Error' (computer_metrix.py:121-127): mean over frames of the max squared lip-vertex error."""
    return vertex_error(gt, pred, mouth_map, device)["max"]


def mean_vertex_error(gt, pred, device="cuda:0"):
    """'Mean Vertex Error' (computer_metrix.py:118-119)."""
    return vertex_error(gt, pred, None, device)["mean_dist"]


def upper_face_dynamics_deviation(gt, pred, template, upper_map, device="cuda:0"):
    """Per-sequence FDD term (computer_metrix.py:95-107): std-of-motion(gt) - std-of-motion(pred)."""
    return motion_std(gt, template, upper_map, device) - motion_std(pred, template, upper_map, device)


def load_regions(region_path, dataset):
    """Vertex index lists as computer_metrix.py:23-55 reads them."""
    if dataset == "BIWI":
        with open(os.path.join(region_path, "lve.txt")) as f:
            mouth = [int(i) for i in f.read().split(", ")]
        with open(os.path.join(region_path, "fdd.txt")) as f:
            upper = [int(i) for i in f.read().split(", ")]
        return mouth, upper, 23370, ["e" + str(i).zfill(2) for i in range(37, 41)]
    with open(os.path.join(region_path, "weighted_mouth_mask.txt")) as f:
        mouth = [i for i, v in enumerate(float(line.strip()) for line in f if line) if v > 0.1]
    with open(os.path.join(region_path, "forehead_mask.txt")) as f:
        upper = [i for i, v in enumerate(float(line.strip()) for line in f if line) if v > 0.4]
    return mouth, upper, 6172, [str(i) for i in range(46, 51)]


def pred_name(subject, sentence, condition=None, model=""):
    """Basename of a prediction file as computer_metrix.py reads it: `<subject>_<sentence>` (:74, what the MEAD / BIWI samplers
    write), `<subject>_<sentence>_condition_<conditioning subject>` (:171-174, what samples/sample_diffusion_vocaset.py:86-88
    writes), with `<model>_` in front when a model tag is given (:69-71)."""
    name = subject + "_" + sentence + (("_condition_" + condition) if condition is not None else "")
    return (model + "_" + name) if model != "" else name


def evaluate(pred_path, gt_path, region_path, templates_path, train_subjects="F2 F3 F4 M3 M4 M5", model="", dataset="BIWI",
             device="cuda:0", verbose=True):
    """computer_metrix.py `main` (:6-136): same files, same naming, same printed lines; returns the numbers."""
    mouth, upper, nv, sentences = load_regions(region_path, dataset)
    with open(templates_path, "rb") as fin:
        templates = pickle.load(fin, encoding="latin1")
    say = print if verbose else (lambda *a, **k: None)
    gts, preds, fdd = [], [], []
    for subject in train_subjects.split(" "):
        for sentence in sentences:
            gt = np.load(os.path.join(gt_path, subject + "_" + sentence + ".npy")).reshape(-1, nv, 3)
            name = pred_name(subject, sentence, subject, model) if model != "" else pred_name(subject, sentence)
            pred = np.load(os.path.join(pred_path, name + ".npy")).reshape(-1, nv, 3)
            n = min(gt.shape[0], pred.shape[0])
            gt, pred = _dev(gt[:n], device), _dev(pred[:n], device)
            say(tuple(pred.shape))
            fdd.append(upper_face_dynamics_deviation(gt, pred, templates[subject], upper, device))
            say(f"{subject}_{sentence}")
            say("FDD: {:.4e}".format(fdd[-1]), "FDD: {:.4e}".format(sum(fdd) / len(fdd)))
            gts.append(gt)
            preds.append(pred)
    gt_all, pred_all = torch.cat(gts), torch.cat(preds)
    say("Frame Number: {}".format(gt_all.shape[0]))
    say(tuple(gt_all.shape))
    res = dict(frames=int(gt_all.shape[0]),
               mean_vertex_error=vertex_error(gt_all, pred_all, None, device)["mean_dist"],
               lip_vertex_error=vertex_error(gt_all, pred_all, mouth, device)["max"],
               fdd=sum(fdd) / len(fdd), abs_fdd=sum(abs(x) for x in fdd) / len(fdd))
    say("Mean Vertex Error: {:.4e}".format(res["mean_vertex_error"]))
    say("Lip Vertex Error: {:.4e}".format(res["lip_vertex_error"]))
    say("FDD: {:.4e}".format(res["fdd"]))
    say("ABS FDD: {:.4e}".format(res["abs_fdd"]))
    return res


def diversity(pred_path, train_subjects, test_subjects, dataset="BIWI", device="cuda:0", verbose=True, sentences=None, nr_vertices=None):
    """computer_metrix.py `compute_diversity` (:139-194): mean pairwise vertex distance between the predictions of one
    test sequence under different conditioning subjects.  `sentences` / `nr_vertices` override the two tables the reference
    hard-codes per dataset (:155-162), e.g. VOCASET's 5023-vertex meshes and `sentenceNN` names."""
    nv = nr_vertices or (23370 if dataset == "BIWI" else 6172)
    sentences = sentences or (["e" + str(i).zfill(2) for i in range(37, 41)] if dataset == "BIWI" else [str(i) for i in range(46, 51)])
    say = print if verbose else (lambda *a, **k: None)
    total, num = 0.0, 0
    for subject in test_subjects.split(" "):
        for sentence in sentences:
            say(subject, sentence)
            seqs = []
            for cond in train_subjects.split(" "):
                fp = os.path.join(pred_path, pred_name(subject, sentence, cond) + ".npy")
                if os.path.exists(fp):
                    seqs.append(_dev(np.load(fp).reshape(-1, nv, 3), device))
            n = len(seqs)
            if n < 2:
                continue
            d = 0.0
            for i in range(n - 1):
                for j in range(i + 1, n):
                    d += vertex_error(seqs[i], seqs[j], None, device)["mean_dist"]
            d /= (n - 1) * n / 2
            say(d)
            total += d
            num += 1
    if num == 0:
        raise FdmError("diversity: no test sequence has two or more conditioned predictions")
    say("Diversity: {:.4e}".format(total / num))
    return total / num


def mead_vertex_metrics(gt, pred, face_vertex, lip_vertex, emotion_vertex, device="cuda:0"):
    """The vertex-error block of metric/metric.py:115-138 on [F, 5023, 3] arrays: FVE, LVE (max over the region),
    EME (mean over the region) and the all-vertex error."""
    return dict(FVE=vertex_error(gt, pred, face_vertex, device)["max"], LVE=vertex_error(gt, pred, lip_vertex, device)["max"],
                EME=vertex_error(gt, pred, emotion_vertex, device)["mean_sq"], ALL=vertex_error(gt, pred, None, device)["max"])


def main(argv=None):
    """CLI of computer_metrix.py (:7-17 and :140-150), then `main()` and `compute_diversity()` as its __main__ does."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--train_subjects", type=str, default="F2 F3 F4 M3 M4 M5")
    ap.add_argument("--test_subjects", type=str, default="F1 F5 F6 F7 F8 M1 M2 M6")
    ap.add_argument("--pred_path", type=str, default="/data/WX/fdm/checkpoints/diffusion_Encoder_Decoder/result")
    ap.add_argument("--gt_path", type=str, default="/data/WX/BIWI_dataset/vertices_npy")
    ap.add_argument("--region_path", type=str, default="/data/WX/BIWI_dataset/regions/")
    ap.add_argument("--templates_path", type=str, default="/data/WX/BIWI_dataset/templates.pkl")
    ap.add_argument("--model", type=str, default="")
    ap.add_argument("--num_sample", type=str)
    ap.add_argument("--dataset", type=str, default="BIWI")
    ap.add_argument("--device", type=str, default="cuda:0")
    a = ap.parse_args(argv)
    evaluate(a.pred_path, a.gt_path, a.region_path, a.templates_path, a.train_subjects, a.model, a.dataset, a.device)
    try:
        diversity(a.pred_path, a.train_subjects, a.test_subjects, a.dataset, a.device)
    except FdmError as e:
        print(e)
