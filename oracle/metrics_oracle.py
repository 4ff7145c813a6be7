"""ORACLE (test infrastructure, never imported by the product): numpy restatement of the reference's evaluation
metrics -- computer_metrix.py:84-136 (main), :139-194 (compute_diversity) and metric/metric.py:115-138.

Same float32 arithmetic and numpy reductions as the reference, written with fancy indexing instead of the
reference's per-vertex list comprehensions (identical values, same dtype)."""
import numpy as np


def region_sq_err(gt, pred, region=None):
    """[F, V, 3] x2 -> [F, R] squared vertex error (computer_metrix.py:121-123: np.square, sum over axis 2)."""
    g = gt if region is None else gt[:, region, :]
    p = pred if region is None else pred[:, region, :]
    return np.sum(np.square(g - p), axis=2)


def max_vertex_error(gt, pred, region=None):
    """'Lip Vertex Error' / FVE / all-vertex error: max over the region per frame, mean over frames
    (computer_metrix.py:121-127, metric/metric.py:115-128)."""
    return float(np.mean(np.max(region_sq_err(gt, pred, region), axis=1)))


def mean_sq_error(gt, pred, region=None):
    """'Emotion Mean Error' (metric/metric.py:130-133): mean over the region per frame, mean over frames."""
    return float(np.mean(np.mean(region_sq_err(gt, pred, region), axis=1)))


def mean_vertex_error(gt, pred):
    """'Mean Vertex Error' (computer_metrix.py:118-119); also the pairwise term of compute_diversity (:180)."""
    return float(np.mean(np.linalg.norm(gt - pred, axis=2)))


def motion_std(verts, template, region):
    """computer_metrix.py:95-99: mean over the region of std over frames of the squared motion magnitude."""
    motion = verts - template.reshape(1, -1, 3)
    l2 = np.sum(np.square(motion[:, region, :]), axis=2)
    return float(np.mean(np.std(l2, axis=0)))


def fdd(gt, pred, template, region):
    """One sequence's FDD term (computer_metrix.py:95-107)."""
    return motion_std(gt, template, region) - motion_std(pred, template, region)


def synth_sequences(seed, n_seq, frames, nv, scale=0.01):
    """Seeded synthetic (gt, pred) vertex sequences and a template (legacy RandomState: platform-stable)."""
    rs = np.random.RandomState(seed)
    template = rs.standard_normal((nv, 3)).astype(np.float32) * 0.1
    out = []
    for i in range(n_seq):
        f = frames + (i % 3)
        drift = np.cumsum(rs.standard_normal((f, nv, 3)).astype(np.float32) * scale * 0.1, axis=0)
        gt = (template[None] + drift).astype(np.float32)
        pred = (gt + rs.standard_normal((f + 1, nv, 3)).astype(np.float32)[:f] * scale).astype(np.float32)
        out.append((gt, pred))
    return template, out
