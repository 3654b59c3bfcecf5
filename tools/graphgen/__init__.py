"""Compatibility shim: the graph generator library is product code now (circom-witnesscalc_amd/graphgen);
what stays here is pywriter.py, the independent pure-Python `.bin` writer the tests compare the product's bytes with."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import cwc_import

cwc_import.load()
