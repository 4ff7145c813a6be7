"""Replaces /root/reference utiles/classifierfree.py (ClassifierFreeSampleModel :8-21)."""
from fdm_amd.modules import ClassifierFreeSampleModel  # noqa: F401
