"""MI355X-native calc-witness path of circom-witnesscalc -- Python host-side mirror of the C-ABI.

The product is `libcircom_witnesscalc_amd.so` (HIP kernels for gfx950 + C++ host, see csrc/ and
include/*.h).  This module binds it with ctypes and mirrors the reference's Rust-level API names
(reference src/lib.rs:114-247): `calc_witness`, `wtns_from_witness`, `deserialize_inputs`-equivalent
`inputs_from_json`, plus the additive batch API.  torch is used only to hold device buffers.

There is no CPU evaluation path: every witness value is computed by the HIP kernels; without the
shared library or without a HIP device the calls raise.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CWC_LIB_PATH") or os.path.join(_HERE, "libcircom_witnesscalc_amd.so")
_lib = None


class GwStatus(ctypes.Structure):
    _fields_ = [("code", ctypes.c_int), ("error_msg", ctypes.c_void_p)]


class GraphInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("n_nodes", "n_op", "n_input_nodes", "n_const", "n_inputs",
                                                "n_witness", "depth", "algorithmic_bytes_per_set")]


class Timing(ctypes.Structure):
    _fields_ = [("tile_width", ctypes.c_uint32), ("n_launches", ctypes.c_uint32), ("n_bundles", ctypes.c_uint64),
                ("n_slots", ctypes.c_uint64), ("interp_ms", ctypes.c_float), ("pack_ms", ctypes.c_float),
                ("divider", ctypes.c_uint32), ("streams", ctypes.c_uint32)]


class ProgramStats(ctypes.Structure):
    _fields_ = [("tile_width", ctypes.c_uint32), ("divider", ctypes.c_uint32), ("streams", ctypes.c_uint32), ("n_classes", ctypes.c_uint32),
                ("n_bundles", ctypes.c_uint64), ("n_fused_nodes", ctypes.c_uint64), ("class_bundles", ctypes.c_uint64 * 16),
                ("class_nodes", ctypes.c_uint64 * 16), ("model_wave_cycles", ctypes.c_double), ("lanes_active_mean", ctypes.c_double),
                ("values_per_bundle_mean", ctypes.c_double), ("chain_floor_cycles", ctypes.c_double), ("n_scan_steps", ctypes.c_uint64), ("n_conv_products", ctypes.c_uint64)]


CLASS_NAMES = ["INPUT", "MUL", "LIN", "DIV", "CMPZ", "CMPS", "BIT", "IDIVMOD", "TERN", "DIVREQ", "DIVGET", "MULQ", "SYNC", "MULF", "SCAN"]


class E2eStats(ctypes.Structure):
    _fields_ = [("n_sets", ctypes.c_size_t), ("sub_batch", ctypes.c_size_t), ("parse_threads", ctypes.c_uint32), ("write_threads", ctypes.c_uint32),
                ("parse_seconds", ctypes.c_double), ("wait_for_drain_seconds", ctypes.c_double), ("total_seconds", ctypes.c_double),
                ("witness_bytes", ctypes.c_uint64), ("failed_sets", ctypes.c_uint64)]


class Handoff(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("form", ctypes.c_uint32), ("hip_stream", ctypes.c_void_p), ("done_event", ctypes.c_void_p)]


FORM_CANONICAL, FORM_MONTGOMERY = 0, 1


class WitnessCalcError(RuntimeError):
    pass


def pinned_rows(shape):
    """uint8 array in pinned host memory (gwb_host_alloc); freed when the array and its views are gone."""
    import weakref
    n = int(np.prod(shape))
    ptr = lib().gwb_host_alloc(n)
    if not ptr:
        raise WitnessCalcError("gwb_host_alloc(%d) failed" % n)
    buf = (ctypes.c_uint8 * max(n, 1)).from_address(ptr)
    weakref.finalize(buf, lib().gwb_host_free, ptr)
    return np.frombuffer(buf, dtype=np.uint8, count=n).reshape(shape)


EXPORTED_SYMBOLS = [
    "gw_calc_witness", "gwb_graph_load", "gwb_graph_free", "gwb_graph_info", "gwb_graph_serialize",
    "gwb_inputs_from_json", "gwb_set_tile_width", "gwb_calc_witness_batch_device", "gwb_calc_witness_batch_host",
    "gwb_last_timing", "gwb_wtns_size", "gwb_wtns_from_witness", "gwb_graph_export", "gwb_graph_import",
    "gwb_free_status", "gwb_profile_classes", "gwb_pick_tile_width", "gwb_inputs_from_json_batch", "gwb_wtns_save_batch",
    "gwb_host_alloc", "gwb_host_free", "gwb_timing_history", "gwb_calc_witness_batch_handoff", "gwb_ubench_modmul", "gwb_graph_pick_tile_width", "gwb_graph_broadcast",
    "gwb_builder_new", "gwb_builder_free", "gwb_builder_input", "gwb_builder_constant", "gwb_builder_uno", "gwb_builder_duo", "gwb_builder_tres",
    "gwb_builder_witness", "gwb_builder_input_signal", "gwb_builder_node_count", "gwb_builder_finish",
    "gwb_ubench_modmul_block", "gwb_program_stats", "gwb_calc_witness_json_to_wtns", "gwb_model_class_cycles",
    "gwb_graphgen_bigint_class", "gwb_graphgen_rsa_long_div_class", "gwb_graph_op_histogram",
    "gwb_kernel_source_hash", "gwb_rccl_unique_id", "gwb_rccl_comm_init", "gwb_rccl_comm_ranks", "gwb_rccl_comm_destroy",
]


def build(verbose=False, diag=False):
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950).  diag=True also builds the diagnostic library
    (stamped interpreter instances for gwb_profile_classes; the class-profile / calibration tools load it via CWC_LIB_PATH)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"] + (["all", "diag"] if diag else [])
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


DIAG_LIB_PATH = os.path.join(_HERE, "libcircom_witnesscalc_amd_diag.so")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WitnessCalcError("HIP extension %s is missing: run __graft_entry__.build() "
                                   "(there is no CPU fallback)" % LIB_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64 (soname libamdhip64.so.7).  Loading
        # torch first makes this library's NEEDED libamdhip64.so.7 resolve to that already-loaded copy; the
        # other order would map a second runtime (system ROCm) next to torch's and torch then sees no GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, u32p = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)
        stp = ctypes.POINTER(GwStatus)
        L.gw_calc_witness.restype = ctypes.c_int
        L.gw_calc_witness.argtypes = [ctypes.c_char_p, vp, sz, ctypes.POINTER(vp), ctypes.POINTER(sz), stp]
        L.gwb_graph_load.restype = ctypes.c_int
        L.gwb_graph_load.argtypes = [vp, sz, ctypes.POINTER(vp), stp]
        L.gwb_graph_free.argtypes = [vp]
        L.gwb_graph_info.argtypes = [vp, ctypes.POINTER(GraphInfo)]
        L.gwb_graph_serialize.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz), stp]
        L.gwb_inputs_from_json.argtypes = [vp, ctypes.c_char_p, vp, stp]
        L.gwb_set_tile_width.argtypes = [vp, ctypes.c_uint32]
        L.gwb_calc_witness_batch_device.argtypes = [vp, vp, sz, vp, vp, vp, stp]
        L.gwb_calc_witness_batch_host.argtypes = [vp, vp, sz, vp, vp, stp]
        L.gwb_last_timing.argtypes = [vp, ctypes.POINTER(Timing)]
        L.gwb_host_alloc.restype = ctypes.c_void_p
        L.gwb_host_alloc.argtypes = [sz]
        L.gwb_host_free.restype = None
        L.gwb_host_free.argtypes = [ctypes.c_void_p]
        L.gwb_timing_history.restype = ctypes.c_int
        L.gwb_timing_history.argtypes = [ctypes.c_void_p, sz, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(sz)]
        L.gwb_wtns_size.restype = sz
        L.gwb_wtns_size.argtypes = [sz]
        L.gwb_wtns_from_witness.argtypes = [vp, sz, vp]
        L.gwb_graph_export.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(vp), ctypes.POINTER(sz), stp]
        L.gwb_graph_import.argtypes = [vp, sz, ctypes.POINTER(vp), stp]
        L.gwb_free_status.argtypes = [stp]
        if hasattr(L, "gwb_graphgen_bigint_class"):  # (absent from an older build loaded through CWC_LIB_PATH for a same-box A/B)
            L.gwb_graphgen_bigint_class.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp), ctypes.POINTER(sz), stp]
            L.gwb_graphgen_rsa_long_div_class.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(sz), stp]
            L.gwb_graph_op_histogram.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), sz]
        L.gwb_profile_classes.argtypes = [vp, vp, sz, vp, vp, vp, stp]
        L.gwb_inputs_from_json_batch.argtypes = [vp, ctypes.c_char_p, sz, vp, sz, ctypes.POINTER(sz), stp]
        L.gwb_wtns_save_batch.argtypes = [vp, sz, sz, ctypes.c_char_p, stp]
        L.gwb_calc_witness_batch_handoff.argtypes = [vp, vp, sz, vp, vp, ctypes.POINTER(Handoff), stp]
        L.gwb_ubench_modmul.restype = ctypes.c_double
        L.gwb_ubench_modmul.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        L.gwb_ubench_modmul_block.restype = ctypes.c_double
        L.gwb_model_class_cycles.restype = ctypes.c_double
        L.gwb_kernel_source_hash.restype = ctypes.c_char_p
        L.gwb_kernel_source_hash.argtypes = []
        L.gwb_model_class_cycles.argtypes = [ctypes.c_uint32]
        L.gwb_ubench_modmul_block.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        L.gwb_program_stats.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(ProgramStats)]
        L.gwb_calc_witness_json_to_wtns.argtypes = [vp, ctypes.c_char_p, sz, ctypes.c_char_p, sz, ctypes.POINTER(sz), vp, sz, ctypes.POINTER(E2eStats), stp]
        L.gwb_graph_pick_tile_width.restype = ctypes.c_uint32
        L.gwb_graph_pick_tile_width.argtypes = [vp, sz]
        L.gwb_pick_tile_width.restype = ctypes.c_uint32
        L.gwb_pick_tile_width.argtypes = [sz]
        u32 = ctypes.c_uint32
        L.gwb_builder_new.restype = vp
        L.gwb_builder_new.argtypes = []
        L.gwb_builder_free.restype = None
        L.gwb_builder_free.argtypes = [vp]
        for name, args in (("input", [u32]), ("constant", [ctypes.c_char_p, sz]), ("uno", [u32, u32]), ("duo", [u32, u32, u32]), ("tres", [u32, u32, u32, u32])):
            f = getattr(L, "gwb_builder_" + name)
            f.restype = u32
            f.argtypes = [vp] + args
        L.gwb_builder_witness.argtypes = [vp, u32]
        L.gwb_builder_input_signal.argtypes = [vp, ctypes.c_char_p, u32, u32]
        L.gwb_builder_node_count.restype = ctypes.c_uint64
        L.gwb_builder_node_count.argtypes = [vp]
        L.gwb_builder_finish.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz), stp]
        _lib = L
    return _lib


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def _take_status(st):
    msg = ctypes.string_at(st.error_msg).decode("utf-8", "replace") if st.error_msg else ""
    lib().gwb_free_status(ctypes.byref(st))
    return msg


def _check(rc, st):
    msg = _take_status(st)
    if rc != 0:
        raise WitnessCalcError(msg or "call failed")


def wtns_save_batch(witness, path_pattern):
    """One `.wtns` file per input set; witness uint8 [B, W, 32]; path_pattern with one %lu (gwb_wtns_save_batch)."""
    witness = np.ascontiguousarray(witness, dtype=np.uint8)
    st = GwStatus()
    rc = lib().gwb_wtns_save_batch(witness.ctypes.data, witness.shape[1], witness.shape[0], path_pattern.encode(), ctypes.byref(st))
    _check(rc, st)


def ubench_modmul(waves_per_simd=4, iters=2000, block=False):
    """Chip-wide one-lane Montgomery products per second (gwb_ubench_modmul; block: the interpreter's own multiplier,
    gwb_ubench_modmul_block, at most two waves per SIMD)."""
    if block:
        return float(lib().gwb_ubench_modmul_block(waves_per_simd, iters))
    return float(lib().gwb_ubench_modmul(waves_per_simd, iters))


def kernel_source_hash():
    """SHA-256 of the kernel sources the loaded library's device code was built from (gwb_kernel_source_hash)."""
    return lib().gwb_kernel_source_hash().decode()


def model_cycles():
    """The cost model's lone-wave cycles per bundle class as the library loaded them ({class name: cycles};
    gwb_model_class_cycles): built-in, CWC_MODEL_CYCLES, or the calibration file of tools/gpu_calibrate.py."""
    return {n: float(lib().gwb_model_class_cycles(c)) for c, n in enumerate(CLASS_NAMES)}


def graphgen_native(kind, **kw):
    """`.bin` bytes of one of BASELINE config 5's class graphs from the native generators (gwb_graphgen_*): kind "bigint" (k, n_bits,
    rounds) or "rsa" (n, k, muls, range_checks) -- the same bytes as graphgen.circuits.build_bigint_class / build_rsa_long_div_class
    write, without ten million nodes passing through Python."""
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), GwStatus()
    L = lib()
    if kind == "bigint":
        rc = L.gwb_graphgen_bigint_class(int(kw.get("k", 8)), int(kw.get("n_bits", 64)), int(kw.get("rounds", 4)), ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    elif kind == "rsa":
        rc = L.gwb_graphgen_rsa_long_div_class(int(kw.get("n", 121)), int(kw.get("k", 17)), int(kw.get("muls", 2)), 1 if kw.get("range_checks", True) else 0,
                                               ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
    else:
        raise ValueError(kind)
    _check(rc, st)
    data = ctypes.string_at(out.value, n.value)
    _libc.free(out)
    return data


def pick_tile_width(batch):
    """Input sets per wavefront the library uses for a batch of this size (gwb_pick_tile_width)."""
    return int(lib().gwb_pick_tile_width(batch))


def calc_witness_wtns(inputs_json, graph_data):
    """gw_calc_witness (reference src/lib.rs:44-111): JSON text + `.bin` bytes -> `.wtns` bytes."""
    if isinstance(inputs_json, str):
        inputs_json = inputs_json.encode("utf-8")
    graph_data = bytes(graph_data)
    out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), GwStatus()
    rc = lib().gw_calc_witness(inputs_json, graph_data, len(graph_data), ctypes.byref(out), ctypes.byref(n),
                               ctypes.byref(st))
    _check(rc, st)
    data = ctypes.string_at(out.value, n.value)
    _libc.free(out)
    return data


def calc_witness(inputs_json, graph_data):
    """calc_witness (reference src/lib.rs:125-136): -> list[int] witness values."""
    w = calc_witness_wtns(inputs_json, graph_data)
    body = w[76:]
    return [int.from_bytes(body[i:i + 32], "little") for i in range(0, len(body), 32)]


def wtns_from_witness(witness):
    """wtns_from_witness (reference src/lib.rs:114-123). witness: list[int] or uint8 array [W, 32]."""
    if not isinstance(witness, np.ndarray):
        witness = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in witness), dtype=np.uint8)
    witness = np.ascontiguousarray(witness, dtype=np.uint8).reshape(-1, 32)
    n = witness.shape[0]
    out = np.zeros(lib().gwb_wtns_size(n), dtype=np.uint8)
    if lib().gwb_wtns_from_witness(witness.ctypes.data, n, out.ctypes.data) != 0:
        raise WitnessCalcError("gwb_wtns_from_witness failed")
    return out.tobytes()


class Graph:
    """A parsed + compiled graph (deserialize_witnesscalc_graph, reference src/storage.rs:214-249), reusable
    across calls -- the reference re-parses the `.bin` on every calc_witness (src/lib.rs:129)."""

    def __init__(self, graph_data=None, _handle=None):
        self._h = ctypes.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            graph_data = bytes(graph_data)
            st = GwStatus()
            rc = lib().gwb_graph_load(graph_data, len(graph_data), ctypes.byref(self._h), ctypes.byref(st))
            _check(rc, st)
        info = GraphInfo()
        lib().gwb_graph_info(self._h, ctypes.byref(info))
        for name, _ in GraphInfo._fields_:
            setattr(self, name, int(getattr(info, name)))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.gwb_graph_free(self._h)
            self._h = None

    __del__ = close

    @classmethod
    def from_blob(cls, blob):
        """gwb_graph_import: build a replica from a compiled-program blob (what ranks receive over RCCL)."""
        blob = bytes(blob)
        h, st = ctypes.c_void_p(), GwStatus()
        rc = lib().gwb_graph_import(blob, len(blob), ctypes.byref(h), ctypes.byref(st))
        _check(rc, st)
        return cls(_handle=h)

    def op_histogram(self):
        """{operation name: nodes} of the graph as loaded (gwb_graph_op_histogram)"""
        h = (ctypes.c_uint64 * 24)()
        if lib().gwb_graph_op_histogram(self._h, h, 24) != 0:
            raise WitnessCalcError("gwb_graph_op_histogram failed")
        names = ["Mul", "Div", "Add", "Sub", "Pow", "Idiv", "Mod", "Eq", "Neq", "Lt", "Gt", "Leq", "Geq", "Land", "Lor", "Shl", "Shr", "Bor", "Band", "Bxor",
                 "Neg", "TernCond", "Input", "Const"]
        return dict(sorted(((nm, int(h[i])) for i, nm in enumerate(names) if h[i]), key=lambda kv: -kv[1]))

    def export_blob(self, tile_width):
        out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), GwStatus()
        rc = lib().gwb_graph_export(self._h, tile_width, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
        _check(rc, st)
        data = ctypes.string_at(out.value, n.value)
        _libc.free(out)
        return data

    def serialize(self):
        """serialize_witnesscalc_graph (reference src/storage.rs:137-183)."""
        out, n, st = ctypes.c_void_p(), ctypes.c_size_t(), GwStatus()
        rc = lib().gwb_graph_serialize(self._h, ctypes.byref(out), ctypes.byref(n), ctypes.byref(st))
        _check(rc, st)
        data = ctypes.string_at(out.value, n.value)
        _libc.free(out)
        return data

    def pick_tile_width(self, batch):
        """Program key the cost model chooses for this graph at this batch size (gwb_graph_pick_tile_width)."""
        key = int(lib().gwb_graph_pick_tile_width(self._h, batch))
        if key == 0:
            raise WitnessCalcError("gwb_graph_pick_tile_width failed")
        return key

    def set_tile_width(self, t):
        if lib().gwb_set_tile_width(self._h, t) != 0:
            raise WitnessCalcError("tile width must be 0 or a power of two in 1..64")

    def inputs_from_json(self, inputs_json):
        """-> uint8 [n_inputs, 32] row (slot 0 = 1); reference src/lib.rs:195-247, 154-181."""
        if isinstance(inputs_json, str):
            inputs_json = inputs_json.encode("utf-8")
        row = np.zeros((self.n_inputs, 32), dtype=np.uint8)
        st = GwStatus()
        rc = lib().gwb_inputs_from_json(self._h, inputs_json, row.ctypes.data, ctypes.byref(st))
        _check(rc, st)
        return row

    def inputs_from_json_batch(self, text):
        """JSON array of input objects or NDJSON -> uint8 [B, n_inputs, 32] (gwb_inputs_from_json_batch)."""
        if isinstance(text, str):
            text = text.encode("utf-8")
        n = ctypes.c_size_t()
        st = GwStatus()
        lib().gwb_inputs_from_json_batch(self._h, text, len(text), None, 0, ctypes.byref(n), ctypes.byref(st))
        msg = _take_status(st)
        if "rows buffer too small" not in msg and msg:
            raise WitnessCalcError(msg)
        rows = np.zeros((n.value, self.n_inputs, 32), dtype=np.uint8)
        st = GwStatus()
        rc = lib().gwb_inputs_from_json_batch(self._h, text, len(text), rows.ctypes.data, n.value, ctypes.byref(n), ctypes.byref(st))
        _check(rc, st)
        return rows

    def json_to_wtns(self, text, path_pattern, first_index=0, with_status=True):
        """End to end, streaming (gwb_calc_witness_json_to_wtns): JSON array / NDJSON text -> one `.wtns` file per input set
        (path_pattern with one %lu); a set with a non-zero status word gets no file.  Returns (per-set status uint32 [B],
        stats dict); with_status=False passes no status buffer: a failed set then raises."""
        if isinstance(text, str):
            text = text.encode("utf-8")
        n = ctypes.c_size_t()
        cap = max(1, text.count(b"\n") + 1, text.count(b"{"))
        status = np.zeros(cap, dtype=np.uint32)
        es, st = E2eStats(), GwStatus()
        rc = lib().gwb_calc_witness_json_to_wtns(self._h, text, len(text), path_pattern.encode(), first_index, ctypes.byref(n),
                                                  status.ctypes.data if with_status else None, cap if with_status else 0,
                                                  ctypes.byref(es), ctypes.byref(st))
        _check(rc, st)
        return status[:n.value], {k: getattr(es, k) for k, _ in E2eStats._fields_}

    def calc_witness_batch(self, inputs, out=None):
        """Host buffers: inputs uint8 [B, n_inputs, 32] -> (witness uint8 [B, W, 32], status uint32 [B]).
        `out`: optional contiguous uint8 [B, W, 32] to fill (e.g. from pinned_rows, which skips the staging copy)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
        b = inputs.shape[0]
        assert inputs.shape[1:] == (self.n_inputs, 32), inputs.shape
        if out is None:
            wit = np.empty((b, self.n_witness, 32), dtype=np.uint8)
        else:
            wit = out
            assert wit.dtype == np.uint8 and wit.shape == (b, self.n_witness, 32) and wit.flags["C_CONTIGUOUS"]
        status = np.zeros(b, dtype=np.uint32)
        st = GwStatus()
        rc = lib().gwb_calc_witness_batch_host(self._h, inputs.ctypes.data, b, wit.ctypes.data, status.ctypes.data,
                                               ctypes.byref(st))
        _check(rc, st)
        return wit, status

    def calc_witness_batch_device(self, d_inputs, d_witness, d_status, stream=None, montgomery=False, done_event=None):
        """Device-resident: torch uint8 cuda tensors [B, n_inputs, 32] -> [B, W, 32], int32/uint32 [B].
        Asynchronous on `stream` (torch.cuda.Stream) or the current torch stream.  montgomery / done_event
        (torch.cuda.Event, recorded behind the call's last kernel): the prover hand-off of gwb_calc_witness_batch_handoff."""
        import torch
        b = d_inputs.shape[0]
        assert d_inputs.is_cuda and d_witness.is_cuda and d_status.is_cuda
        assert d_inputs.is_contiguous() and d_witness.is_contiguous() and d_status.is_contiguous()
        assert tuple(d_inputs.shape[1:]) == (self.n_inputs, 32) and tuple(d_witness.shape) == (b, self.n_witness, 32)
        s = stream if stream is not None else torch.cuda.current_stream()
        st = GwStatus()
        if montgomery or done_event is not None:
            if done_event is not None:
                done_event.record(s)  # (creates the underlying hipEvent_t; the library records it again behind its kernels)
            h = Handoff(ctypes.sizeof(Handoff), FORM_MONTGOMERY if montgomery else FORM_CANONICAL, s.cuda_stream,
                        done_event.cuda_event if done_event is not None else None)
            rc = lib().gwb_calc_witness_batch_handoff(self._h, d_inputs.data_ptr(), b, d_witness.data_ptr(), d_status.data_ptr(),
                                                      ctypes.byref(h), ctypes.byref(st))
        else:
            rc = lib().gwb_calc_witness_batch_device(self._h, d_inputs.data_ptr(), b, d_witness.data_ptr(),
                                                     d_status.data_ptr(), s.cuda_stream, ctypes.byref(st))
        _check(rc, st)

    def timing_history(self, max_launches):
        """(interp_ms, pack_ms) float32 arrays of the most recent launches on this handle, oldest first (synchronizes
        on their events); for timing a run of asynchronous calls without a synchronization inside it."""
        a = np.zeros(max_launches, dtype=np.float32)
        b = np.zeros(max_launches, dtype=np.float32)
        n = ctypes.c_size_t(0)
        if lib().gwb_timing_history(self._h, max_launches, a.ctypes.data, b.ctypes.data, ctypes.byref(n)) != 0:
            raise WitnessCalcError("gwb_timing_history failed")
        return a[:n.value], b[:n.value]

    def profile_classes(self, d_inputs, d_witness, d_status):
        """Diagnostic stamped build: {class: (cycles, 0, 0, bundles)} over sampled waves, plus "_sections":
        {"MUL" / "LIN": (top + staged-operand wait, LDS reads + previous bundle's stores, staging issue, arithmetic, ring write, bundles)}."""
        out = np.zeros(96, dtype=np.uint64)
        st = GwStatus()
        rc = lib().gwb_profile_classes(self._h, d_inputs.data_ptr(), d_inputs.shape[0], d_witness.data_ptr(),
                                       d_status.data_ptr(), out.ctypes.data, ctypes.byref(st))
        _check(rc, st)
        names = ["INPUT", "MUL", "LIN", "DIV", "CMPZ", "CMPS", "BIT", "IDIVMOD", "TERN", "DIVREQ", "DIVGET", "MULQ"]
        res = {n: tuple(int(x) for x in out[4 * i:4 * i + 4]) for i, n in enumerate(names)}
        res["MULF"] = tuple(int(x) for x in out[64:68])
        res["SCAN"] = tuple(int(x) for x in out[68:72])
        res["_scan_kinds"] = {k: tuple(int(x) for x in out[72 + 4 * i:76 + 4 * i]) for i, k in enumerate(("carry", "division", "convolution", "borrow", "comparison"))}
        res["_sections"] = {"MUL": tuple(int(x) for x in out[48:54]), "LIN": tuple(int(x) for x in out[56:62])}
        # the issue part of section 1 (reads, record refill, previous stores ISSUED; the rest of the section is the wait for the LDS reads)
        res["_issue_part"] = {"MUL": int(out[92]), "LIN": int(out[93])}
        n = int(out[63])
        res["_waves"] = {"n": n, "max_cycles": int(out[54]), "min_cycles": (1 << 40) - int(out[55]) if n else 0,
                         "mean_cycles": int(out[62]) // n if n else 0}
        return res

    def program_stats(self, key=0):
        """Statistics of a compiled program (0: the one the last batch call used): gwb_program_stats."""
        ps = ProgramStats()
        if lib().gwb_program_stats(self._h, key, ctypes.byref(ps)) != 0:
            raise WitnessCalcError("gwb_program_stats: no such program")
        n = ps.n_classes
        return {"tile_width": ps.tile_width, "divider": ps.divider, "streams": ps.streams, "n_bundles": ps.n_bundles,
                "n_fused_nodes": ps.n_fused_nodes, "n_scan_steps": ps.n_scan_steps, "n_conv_products": ps.n_conv_products, "chain_floor_cycles": ps.chain_floor_cycles, "model_wave_cycles": ps.model_wave_cycles, "lanes_active_mean": ps.lanes_active_mean, "values_per_bundle_mean": ps.values_per_bundle_mean,
                "class_bundles": {CLASS_NAMES[c]: int(ps.class_bundles[c]) for c in range(n) if ps.class_bundles[c]},
                "class_nodes": {CLASS_NAMES[c]: int(ps.class_nodes[c]) for c in range(n) if ps.class_nodes[c]}}

    def last_timing(self):
        t = Timing()
        if lib().gwb_last_timing(self._h, ctypes.byref(t)) != 0:
            raise WitnessCalcError("gwb_last_timing failed")
        return {n: getattr(t, n) for n, _ in Timing._fields_}


from . import graphgen  # noqa: E402,F401  (graph generator library on top of the C-ABI producer)
