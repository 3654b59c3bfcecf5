"""Multi-GPU plumbing for batched witness generation: one process per GPU, input sets sharded
contiguously, the compiled graph program broadcast once from rank 0 (RCCL over xGMI when the backend is
"nccl"; "gloo" in CPU tests).  There is no steady-state collective: every input set is independent
(reference graph::evaluate touches only its own `values`, src/graph.rs:371-382)."""
import numpy as np


def shard_range(batch, rank, world_size):
    """Contiguous shard [lo, hi) of `batch` input sets for `rank`: sizes differ by at most one."""
    base, rem = divmod(batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(blob, src=0, device=None):
    """Broadcast a bytes object from `src` to every rank with torch.distributed; returns bytes.
    `blob` is only read on `src`.  `device`: torch device of the staging tensor ("cuda:N" for the
    nccl/RCCL backend so that the payload moves GPU to GPU, "cpu" for gloo)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    size = int(n.item())
    if rank == src:
        buf = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
    else:
        buf = torch.empty(size, dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return buf.cpu().numpy().tobytes()


def broadcast_graph(pkg, graph_data, tile_width=0, src=0, device=None, batch_per_rank=None):
    """Rank `src` parses + compiles `graph_data` (`.bin` bytes) and broadcasts the compiled program; the other ranks
    import it (gwb_graph_import: checksummed + validated) instead of re-parsing.  tile_width = 0: rank `src` asks the
    library's cost model for the program of a `batch_per_rank`-set shard (every rank gets the same shard size up to one
    set), so the replicas run what a single-GPU call of that size would run.  Returns a pkg.Graph."""
    import torch.distributed as dist
    rank = dist.get_rank()
    g = None
    blob = b""
    if rank == src:
        g = pkg.Graph(graph_data)
        if not tile_width:
            assert batch_per_rank, "tile_width = 0 needs the shard size"
            tile_width = g.pick_tile_width(batch_per_rank)
        g.set_tile_width(tile_width)
        blob = g.export_blob(tile_width)
    blob = broadcast_blob(blob, src=src, device=device)
    if rank != src:
        g = pkg.Graph.from_blob(blob)
    return g


class RcclComm:
    """A RCCL communicator of this process group's ranks, made for the library's own collective (gwb_graph_broadcast takes an
    ncclComm_t; torch does not hand out the one it uses).  rank `src` draws the unique id, torch.distributed carries its 128
    bytes to the other ranks, every rank joins -- through the library's own gwb_rccl_unique_id / gwb_rccl_comm_init (the
    128-byte id goes to ncclCommInitRank by value: done in C, not through ctypes).  RCCL is the copy torch has already
    loaded (librccl.so in torch/lib), resolved through the process's global symbols."""

    @staticmethod
    def load_library():
        """the RCCL copy torch itself uses first (by path: dlopen then returns that instance and makes its symbols global, which
        is where gwb_graph_broadcast looks ncclBroadcast up), then the system's; None when there is none"""
        import ctypes
        import os
        import torch
        for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"):
            try:
                cand = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
                for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommCount", "ncclCommDestroy", "ncclBroadcast", "ncclGetErrorString"):
                    getattr(cand, sym)
                return cand
            except (OSError, AttributeError):
                continue
        return None

    def __init__(self, device, src=0, library=None, pkg=None):
        import ctypes
        import torch
        import torch.distributed as dist
        self._ct = ctypes
        L = library or RcclComm.load_library()
        if L is None:
            raise RuntimeError("RCCL not found (librccl.so)")
        self.L = L
        self._native = None
        if pkg is not None:  # the library's own helpers (the id by value in C)
            N = pkg.lib()
            N.gwb_rccl_unique_id.argtypes = [ctypes.c_void_p, ctypes.POINTER(pkg.GwStatus)]
            N.gwb_rccl_comm_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(pkg.GwStatus)]
            N.gwb_rccl_comm_ranks.argtypes = [ctypes.c_void_p]
            N.gwb_rccl_comm_destroy.argtypes = [ctypes.c_void_p]
            N.gwb_rccl_comm_destroy.restype = None
            rank, world = dist.get_rank(), dist.get_world_size()
            raw = ctypes.create_string_buffer(128)

            def check(rc, st, what):
                msg = ctypes.string_at(st.error_msg).decode("utf-8", "replace") if st.error_msg else ""
                N.gwb_free_status(ctypes.byref(st))
                if rc != 0:
                    raise RuntimeError("%s: %s" % (what, msg))
            # Every rank enters the broadcast whatever happened on the source: its failure travels as a flag byte behind the id and
            # all ranks raise together (a source that raised first would leave the others waiting in the collective for good).
            src_error = None
            if rank == src:
                st = pkg.GwStatus()
                try:
                    check(N.gwb_rccl_unique_id(raw, ctypes.byref(st)), st, "gwb_rccl_unique_id")
                except RuntimeError as e:
                    src_error = str(e)
            t = torch.frombuffer(bytearray(raw.raw + (b"\x00" if src_error is None else b"\x01")), dtype=torch.uint8).clone().to(device)
            dist.broadcast(t, src=src)
            got = t.cpu().numpy().tobytes()
            if got[128]:
                raise RuntimeError(src_error or "rank %d could not draw a RCCL unique id" % src)
            raw = ctypes.create_string_buffer(got[:128], 128)
            torch.cuda.set_device(device)
            self.comm = ctypes.c_void_p()
            st = pkg.GwStatus()
            check(N.gwb_rccl_comm_init(raw, world, rank, ctypes.byref(self.comm), ctypes.byref(st)), st, "gwb_rccl_comm_init")
            self.n_ranks = N.gwb_rccl_comm_ranks(self.comm)
            self._native = N
            return

        class UniqueId(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_ubyte * 128)]  # (c_char would cut the id at its first zero byte when read back)
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        L.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = UniqueId()
        src_failed = rank == src and L.ncclGetUniqueId(ctypes.byref(uid)) != 0  # (signalled through the broadcast: see above)
        t = torch.frombuffer(bytearray((bytes(uid.internal) if rank == src else bytes(128)) + (b"\x01" if src_failed else b"\x00")), dtype=torch.uint8).clone().to(device)
        dist.broadcast(t, src=src)
        raw = t.cpu().numpy().tobytes()
        if raw[128]:
            raise RuntimeError("ncclGetUniqueId failed on rank %d" % src)
        ctypes.memmove(ctypes.byref(uid), raw[:128], 128)
        self.comm = ctypes.c_void_p()
        torch.cuda.set_device(device)
        rc = L.ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank)
        if rc != 0:
            L.ncclGetErrorString.restype = ctypes.c_char_p
            raise RuntimeError("ncclCommInitRank failed: %s" % L.ncclGetErrorString(rc).decode())
        n = ctypes.c_int(0)
        L.ncclCommCount(self.comm, ctypes.byref(n))
        self.n_ranks = n.value

    def close(self):
        if getattr(self, "comm", None) and self.comm.value:
            if self._native is not None:
                self._native.gwb_rccl_comm_destroy(self.comm)
            else:
                self.L.ncclCommDestroy(self.comm)
            self.comm = self._ct.c_void_p()


def broadcast_graph_rccl(pkg, graph_data, tile_width=0, src=0, device=None, batch_per_rank=0, comm=None):
    """The same exchange through the C-ABI collective gwb_graph_broadcast (RCCL over xGMI): rank `src` loads the graph, the
    library asks its cost model (tile_width = 0), exports the program and broadcasts it on a RCCL communicator, every other
    rank imports it.  Returns (pkg.Graph, number of ranks of the communicator the broadcast ran on)."""
    import ctypes
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    own = comm is None
    if own:
        # every rank must take the same road: whether RCCL can be driven from here is agreed on first (a rank that could not
        # would otherwise leave the others waiting in ncclCommInitRank); if not, torch.distributed carries the program
        lib_ = RcclComm.load_library()
        flag = torch.tensor([1 if lib_ is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            return broadcast_graph(pkg, graph_data, tile_width, src=src, device=device, batch_per_rank=batch_per_rank), 0
        comm = RcclComm(device, src, library=lib_, pkg=pkg)
    try:
        g = pkg.Graph(graph_data) if rank == src else None
        out, st = ctypes.c_void_p(), pkg.GwStatus()
        L = pkg.lib()
        L.gwb_graph_broadcast.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(pkg.GwStatus)]
        stream = torch.cuda.current_stream(device)
        rc = L.gwb_graph_broadcast(g._h if g is not None else None, tile_width, batch_per_rank or 0, src, rank, comm.comm, stream.cuda_stream, ctypes.byref(out),
                                   ctypes.byref(st))
        msg = ctypes.string_at(st.error_msg).decode("utf-8", "replace") if st.error_msg else ""
        L.gwb_free_status(ctypes.byref(st))
        if rc != 0:
            raise pkg.WitnessCalcError(msg or "gwb_graph_broadcast failed")
        if rank != src:
            g = pkg.Graph(_handle=out)
        elif not tile_width:
            tile_width = g.pick_tile_width(batch_per_rank)
        if rank == src:
            g.set_tile_width(tile_width)
        return g, comm.n_ranks
    finally:
        if own:
            comm.close()


def set_checksums(d_witness, chunk=256):
    """64-bit checksum per input set of device witness rows [B, W, 32] (torch uint8 cuda/cpu tensor): the row's int64
    words times fixed odd weights, summed with wrap-around.  Independent of how the batch was sharded, so the list of
    checksums of a job is the same on 1 GPU and on 8 (bench.py hashes that list)."""
    import torch
    b, w = d_witness.shape[0], d_witness.shape[1]
    words = d_witness.view(torch.int64).reshape(b, w * 4)
    k = torch.arange(1, w * 4 + 1, dtype=torch.int64, device=d_witness.device)
    weights = (k * -7046029254386353131) | 1   # (0x9E3779B97F4A7C15 as int64; odd multipliers)
    out = torch.empty(b, dtype=torch.int64, device=d_witness.device)
    for lo in range(0, b, chunk):
        out[lo:lo + chunk] = (words[lo:lo + chunk] * weights).sum(dim=1)
    return out
