"""Replaces /root/reference models/wav2vec.py (Wav2Vec2Model :69-143, the BIWI audio encoder)."""
from fdm_amd.modules import Wav2Vec2Model  # noqa: F401
