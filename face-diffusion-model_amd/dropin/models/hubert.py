"""Replaces /root/reference models/hubert.py (HubertModel :72-146, linear_interpolation :62-69)."""
from fdm_amd.modules import HubertModel, linear_interpolation  # noqa: F401
