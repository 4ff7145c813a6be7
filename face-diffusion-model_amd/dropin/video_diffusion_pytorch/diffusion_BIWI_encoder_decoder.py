"""Replaces the live part of /root/reference video_diffusion_pytorch/diffusion_BIWI_encoder_decoder.py
(GaussianDiffusion :549-761; the Unet3D / Trainer / gif dataset remainder is dead code there)."""
from fdm_amd.modules import GaussianDiffusion  # noqa: F401
from fdm_amd.schedule import cosine_beta_schedule  # noqa: F401
