"""Replaces /root/reference models/vq_vae_vocaset.py (VQAutoEncoder.quant / .decode)."""
from fdm_amd.modules import VQAutoEncoder  # noqa: F401
