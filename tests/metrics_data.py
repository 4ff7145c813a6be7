"""Rebuilds the seeded synthetic evaluation dataset of tests/golden/make_golden.py::g13_metrics (same seeds, same file
layout as the reference's computer_metrix.py expects) so the numbers in tests/golden/metrics.json -- printed by the
reference itself -- can be checked without committing the vertex arrays."""
import json
import os
import pickle

import numpy as np

from oracle import metrics_oracle as MO

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_metrics():
    return json.load(open(os.path.join(HERE, "golden", "metrics.json")))


def sequences(rec):
    """rec = one dataset entry of metrics.json -> (templates {subject: [V*3]}, {(subject, sentence): (gt, pred)})."""
    templates, seqs = {}, {}
    for si, subj in enumerate(rec["subjects"]):
        tmpl, ss = MO.synth_sequences(rec["seed"] + si, len(rec["sentences"]), rec["frames"], rec["nv"])
        templates[subj] = tmpl.reshape(-1)
        for sent, pair in zip(rec["sentences"], ss):
            seqs[(subj, sent)] = pair
    return templates, seqs


def write_dataset(rec, dataset, root):
    """Writes gt/, pred/, regions/ and templates.pkl under root exactly as g13_metrics did."""
    templates, seqs = sequences(rec)
    for d in ("gt", "pred", "regions"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    s0, s1 = rec["subjects"][0], rec["subjects"][1]
    for (subj, sent), (gt, pred) in seqs.items():
        np.save(os.path.join(root, "gt", f"{subj}_{sent}.npy"), gt.reshape(gt.shape[0], -1))
        np.save(os.path.join(root, "pred", f"{subj}_{sent}.npy"), pred.reshape(pred.shape[0], -1))
        np.save(os.path.join(root, "pred", f"{subj}_{sent}_condition_{s0}.npy"), pred.reshape(pred.shape[0], -1))
        np.save(os.path.join(root, "pred", f"{subj}_{sent}_condition_{s1}.npy"), gt.reshape(gt.shape[0], -1))
    with open(os.path.join(root, "templates.pkl"), "wb") as f:
        pickle.dump(templates, f)
    nv = rec["nv"]
    if dataset == "BIWI":
        open(os.path.join(root, "regions", "lve.txt"), "w").write(", ".join(map(str, rec["mouth"])))
        open(os.path.join(root, "regions", "fdd.txt"), "w").write(", ".join(map(str, rec["upper"])))
    else:
        mm = np.zeros(nv); mm[rec["mouth"]] = 0.5
        um = np.zeros(nv); um[rec["upper"]] = 0.9
        open(os.path.join(root, "regions", "weighted_mouth_mask.txt"), "w").write("\n".join(f"{v:.3f}" for v in mm) + "\n")
        open(os.path.join(root, "regions", "forehead_mask.txt"), "w").write("\n".join(f"{v:.3f}" for v in um) + "\n")
    return templates, seqs
