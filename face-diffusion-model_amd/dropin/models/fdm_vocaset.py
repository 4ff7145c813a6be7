"""Replaces /root/reference models/fdm_vocaset.py (FDM :8-91)."""
from fdm_amd.modules import FDM  # noqa: F401
from fdm_amd.schedule import alibi_slopes, positional_table  # noqa: F401
