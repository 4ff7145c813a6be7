"""Replaces /root/reference models/fdm_vqvae_mead.py (FDM :8-104)."""
from fdm_amd.modules import FDMMead as FDM  # noqa: F401
