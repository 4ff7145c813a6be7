"""Replaces /root/reference models/hubert.py (HubertModel :72-146)."""
from fdm_amd.modules import HubertModel  # noqa: F401
