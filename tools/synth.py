"""Synthetic input sets for bench.py and the parity tests, exactly as SURVEY.md 8(d) words them: every scalar uniform
in [0, r) from a counter-based generator (SplitMix64 of seed and element id, element id = set_index * n_inputs + k),
rejection-sampled below r; sha256-class graphs get uniform bits.  A value depends only on (seed, global set index, k),
so any shard of a global batch can be generated on its own rank."""
import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_R_LIMBS = np.array([(R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
_G = np.uint64(0x9E3779B97F4A7C15)


def _splitmix(x):
    with np.errstate(over="ignore"):
        z = x + _G
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _lt_r(limbs):
    """limbs uint64 [..., 4] little-endian -> value < r"""
    lt = np.zeros(limbs.shape[:-1], dtype=bool)
    eq = np.ones(limbs.shape[:-1], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (limbs[..., i] < _R_LIMBS[i])
        eq &= limbs[..., i] == _R_LIMBS[i]
    return lt


def synth_inputs(kind, n_inputs, batch, seed, first_set=0, first_row=None):
    """-> uint8 [batch, n_inputs, 32] canonical little-endian rows of global sets first_set .. first_set + batch - 1
    (slot 0 = 1).  kind "field": uniform in [0, r); "bits": uniform {0, 1}.  first_row replaces global set 0 when this
    shard holds it (the reference's own input file)."""
    sets = np.arange(first_set, first_set + batch, dtype=np.uint64)
    elem = sets[:, None] * np.uint64(n_inputs) + np.arange(n_inputs, dtype=np.uint64)[None, :]      # element id
    base = _splitmix(np.uint64(seed) ^ _splitmix(elem))                                              # one stream per element
    limbs = np.empty((batch, n_inputs, 4), dtype=np.uint64)
    if kind == "bits":
        limbs[:] = 0
        limbs[..., 0] = base & np.uint64(1)
    else:
        flat, fbase = limbs.reshape(-1, 4), base.reshape(-1)
        idx = np.arange(flat.shape[0])          # elements still to draw
        attempt = 0
        while idx.size:
            with np.errstate(over="ignore"):
                cand = np.stack([_splitmix(fbase[idx] + np.uint64(4 * attempt + j + 1) * _G) for j in range(4)], axis=-1)
            cand[:, 3] &= np.uint64((1 << 62) - 1)   # 254 bits in all: about three draws in four are below r
            flat[idx] = cand
            idx = idx[~_lt_r(cand)]
            attempt += 1
    rows = limbs.view(np.uint8).reshape(batch, n_inputs, 32).copy()
    rows[:, 0, :] = 0
    rows[:, 0, 0] = 1
    if first_row is not None and first_set == 0 and batch > 0:
        rows[0] = first_row
    return rows
