"""Graph generator library: `builder` -- symbolic graph builder on top of the C-ABI producer
(gwb_builder_*, the product's `.bin` writer); `circuits` -- circuit-shaped generators (Poseidon, SMT verifier, EdDSA ladder,
SHA-256, bigint long division, the reference's small test circuits, fuzz DAGs)."""
from . import builder, circuits  # noqa: F401
from .builder import Builder, write_bin  # noqa: F401
