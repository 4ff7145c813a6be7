"""Replaces /root/reference models/vq_vae.py (VQAutoEncoder.quant / .decode)."""
from fdm_amd.modules import VQAutoEncoder  # noqa: F401
