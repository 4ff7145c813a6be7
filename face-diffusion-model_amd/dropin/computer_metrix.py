"""Replaces /root/reference computer_metrix.py (main :6-136, compute_diversity :139-194): same flags, same files,
same printed lines; the reductions run on the MI355X through libfdm_hip.so (fdm_amd.metrics)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _path  # noqa: F401,E402
from fdm_amd.metrics import diversity as compute_diversity, evaluate, main  # noqa: E402,F401

if __name__ == "__main__":
    main()
