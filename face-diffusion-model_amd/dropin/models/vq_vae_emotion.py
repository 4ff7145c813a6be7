"""Replaces /root/reference models/vq_vae_emotion.py (VQAutoEncoder.quant / .decode)."""
from fdm_amd.modules import VQAutoEncoder  # noqa: F401
