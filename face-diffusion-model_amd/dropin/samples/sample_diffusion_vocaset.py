"""Entry point of the reference's samples/sample_diffusion_vocaset.py:22-88 (same file name; output names and shipped schedule in sample_diffusion.py)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sample_diffusion import main  # noqa: E402

if __name__ == "__main__":
    main("vocaset")
