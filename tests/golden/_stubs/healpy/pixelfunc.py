import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", "..")))
from oracle.healpix import nest2ring, ring2nest  # noqa: E402,F401
