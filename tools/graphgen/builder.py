"""The product's Builder (circom_witnesscalc_amd.graphgen.builder) next to the independent Python writer."""
from circom_witnesscalc_amd.graphgen.builder import *  # noqa: F401,F403
from circom_witnesscalc_amd.graphgen.builder import Builder, R, graph_stats, write_bin  # noqa: F401
from .pywriter import encode_node, serialize_graph  # noqa: F401
