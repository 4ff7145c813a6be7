"""Drop-in import paths: `from models.fdm_vocaset import FDM` etc. resolve to the MI355X-native classes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _path  # noqa: F401,E402
