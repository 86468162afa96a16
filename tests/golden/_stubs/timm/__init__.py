"""Stand-in for timm==0.4.12 (absent in this image), used ONLY by tests/golden/make_golden.py to
import the reference.  Own code; contributes no arithmetic when drop rates are 0 / eval mode."""
