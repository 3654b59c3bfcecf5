"""Alias of circom_witnesscalc_amd.graphgen.circuits (moved into the package)."""
import sys

import circom_witnesscalc_amd.graphgen.circuits as _m

sys.modules[__name__] = _m
