"""Single-wav inference CLI -- replaces /root/reference demo/demo_3d_mead.py (flags :109-121, output layout :106)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _path  # noqa: F401,E402
from fdm_amd.pipeline import demo_main  # noqa: E402

if __name__ == "__main__":
    demo_main("mead")
