"""pywriter.py: the independent pure-Python `.bin` writer the tests compare the product's bytes with (test infrastructure).
The graph generator library itself is product code: circom-witnesscalc_amd/graphgen."""
