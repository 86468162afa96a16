"""Stand-in for healpy==1.15.2 (absent in this image), used ONLY by tests/golden/make_golden.py.
Routes `hp.pixelfunc.ring2nest/nest2ring` (reference hp_shifting.py:329,333) to the oracle's
restatement of the published HEALPix algorithm (oracle/healpix.py; parity vs healpy unpinned)."""
from . import pixelfunc  # noqa: F401
