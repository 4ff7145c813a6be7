"""Replaces /root/reference models/fdm.py (BIWI FDM :9-99; build-defined semantics)."""
from fdm_amd.modules import FDMBiwi as FDM  # noqa: F401
