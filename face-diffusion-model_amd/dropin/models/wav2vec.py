"""Replaces /root/reference models/wav2vec.py (Wav2Vec2Model :69-143, the BIWI audio encoder; linear_interpolation :61-67)."""
from fdm_amd.modules import Wav2Vec2Model, linear_interpolation  # noqa: F401
