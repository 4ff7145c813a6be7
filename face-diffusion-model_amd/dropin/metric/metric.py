"""Vertex-error arithmetic of /root/reference metric/metric.py:115-138 (FVE / LVE / EME / all-vertex error) on the
MI355X.  The reference script derives its ground truth from FLAME parameters (FLAME_PyTorch: not part of the sampling
path and absent here); these functions take vertex arrays [F, 5023, 3] instead."""
from fdm_amd.metrics import mead_vertex_metrics, motion_std, vertex_error  # noqa: F401
