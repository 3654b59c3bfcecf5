"""world_size-2 CPU (gloo) test of the multi-GPU plumbing: shard arithmetic and the one-off broadcast of
the compiled program blob.  No data-path collective exists (input sets are independent)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cwc_import
    pkg = cwc_import.load()
    from circom_witnesscalc_amd import dist as cdist
    from tools.graphgen import circuits as C
    import program_emulator as pe
    from oracle import model
    blob = b""
    data = C.build_gadgets().to_bin()
    if rank == 0:
        g = pkg.Graph(data)
        blob = g.export_blob(4)
    got = cdist.broadcast_blob(blob, src=0, device="cpu")
    # every rank can decode the same program and evaluates ITS shard of a 10-set batch identically to the model
    b = pe.Blob(got)
    nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
    lo, hi = cdist.shard_range(10, rank, world)
    ok = True
    for s in range(lo, hi):
        row = [1] + [(s * 1000003 + k) ** 3 % model.M for k in range(6)]
        ok &= pe.run(b, row)[0] == model.evaluate(nodes, row, wit)
    dist.barrier()
    q.put((rank, len(got), b.T, lo, hi, ok))
    dist.destroy_process_group()


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    import cwc_import
    cwc_import.load()
    from circom_witnesscalc_amd.dist import shard_range
    for batch in (0, 1, 7, 8, 1024, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_broadcast_program_blob_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] > 0 and res[0][2] == res[1][2] == 4
    assert (res[0][3], res[0][4], res[1][3], res[1][4]) == (0, 5, 5, 10)
    assert res[0][5] and res[1][5]
