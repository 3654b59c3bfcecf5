"""ctypes binding of oracle/libcwc_oracle.so (ORACLE -- test infrastructure, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERR = {0: "ok", 1: "shl overflow (reference panics)", 2: "bit op result == r (reference panics)",
       3: "unimplemented op (reference panics)", 4: "bad .bin", 5: "index out of range"}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libcwc_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcwc_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_graph_load.restype = ctypes.c_void_p
        L.orc_graph_load.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int)]
        L.orc_graph_free.argtypes = [ctypes.c_void_p]
        L.orc_graph_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.orc_graph_witness.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_evaluate.restype = ctypes.c_int
        L.orc_evaluate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.orc_evaluate_batch.restype = ctypes.c_int
        L.orc_evaluate_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.c_void_p, ctypes.c_void_p]
        L.orc_evaluate_batch_threads.restype = ctypes.c_int
        L.orc_evaluate_batch_threads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        L.orc_eval_op.restype = ctypes.c_int
        L.orc_eval_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                  ctypes.c_char_p]
        L.orc_wtns_from_witness.restype = ctypes.c_uint64
        L.orc_wtns_from_witness.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        _LIB = L
    return _LIB


def to32(x):
    return int(x).to_bytes(32, "little")


def ints_to_array(rows):
    """list[list[int]] -> uint8 array [B, n, 32] (canonical little-endian)."""
    b = len(rows)
    n = len(rows[0]) if b else 0
    buf = bytearray(b * n * 32)
    p = 0
    for r in rows:
        for v in r:
            buf[p:p + 32] = int(v).to_bytes(32, "little")
            p += 32
    return np.frombuffer(bytes(buf), dtype=np.uint8).reshape(b, n, 32).copy()


def array_to_ints(a):
    a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
    return [int.from_bytes(a[i].tobytes(), "little") for i in range(a.shape[0])]


class Graph:
    def __init__(self, data):
        err = ctypes.c_int(0)
        self._h = lib().orc_graph_load(bytes(data), len(data), ctypes.byref(err))
        if not self._h:
            raise ValueError("oracle: cannot load graph: %s" % ERR.get(err.value, err.value))
        info = (ctypes.c_uint64 * 4)()
        lib().orc_graph_info(self._h, info)
        self.n_nodes, self.n_witness, self.n_inputs, self.n_op = [int(x) for x in info]

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:  # (module globals are None-d at interpreter exit)
            _LIB.orc_graph_free(self._h)
            self._h = None

    def witness_signals(self):
        out = np.zeros(self.n_witness, dtype=np.uint32)
        lib().orc_graph_witness(self._h, out.ctypes.data)
        return out

    def evaluate_batch(self, inputs):
        """inputs: uint8 [B, n_in, 32] canonical LE -> (witness uint8 [B, W, 32], status int32 [B])."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
        b, n_in = inputs.shape[0], inputs.shape[1]
        out = np.zeros((b, self.n_witness, 32), dtype=np.uint8)
        status = np.zeros(b, dtype=np.int32)
        lib().orc_evaluate_batch(self._h, inputs.ctypes.data, n_in, b, out.ctypes.data, status.ctypes.data)
        return out, status

    def time_batch(self, inputs):
        """Evaluate without keeping outputs per set separately (cpu_baseline leg)."""
        import time
        inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
        b, n_in = inputs.shape[0], inputs.shape[1]
        out = np.zeros((b, self.n_witness, 32), dtype=np.uint8)
        status = np.zeros(b, dtype=np.int32)
        t0 = time.perf_counter()
        lib().orc_evaluate_batch(self._h, inputs.ctypes.data, n_in, b, out.ctypes.data, status.ctypes.data)
        return time.perf_counter() - t0, out, status


def time_batch_threads(graph, inputs, n_threads):
    """The batch over `n_threads` host threads (contiguous shares): seconds, witness, status."""
    import time
    inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
    b, n_in = inputs.shape[0], inputs.shape[1]
    out = np.zeros((b, graph.n_witness, 32), dtype=np.uint8)
    status = np.zeros(b, dtype=np.int32)
    t0 = time.perf_counter()
    lib().orc_evaluate_batch_threads(graph._h, inputs.ctypes.data, n_in, b, out.ctypes.data, status.ctypes.data, n_threads)
    return time.perf_counter() - t0, out, status


def eval_op(kind, op, a, b=None, c=None):
    """kind: 'Uno'|'Duo'|'Tres'; op: wire code; operands canonical ints. Returns int or raises."""
    k = {"Uno": 2, "Duo": 3, "Tres": 4}[kind]
    out = ctypes.create_string_buffer(32)
    rc = lib().orc_eval_op(k, op, to32(a), None if b is None else to32(b), None if c is None else to32(c), out)
    if rc:
        raise ArithmeticError(ERR.get(rc, rc))
    return int.from_bytes(out.raw, "little")


def wtns_from_witness(w):
    """w: uint8 [W, 32] -> bytes."""
    w = np.ascontiguousarray(w, dtype=np.uint8)
    n = w.shape[0]
    out = np.zeros(76 + 32 * n, dtype=np.uint8)
    ln = lib().orc_wtns_from_witness(w.ctypes.data, n, out.ctypes.data)
    return out[:ln].tobytes()
