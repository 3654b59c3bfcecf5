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
