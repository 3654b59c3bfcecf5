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
    import cwc_import
    C = cwc_import.load().graphgen.circuits
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


def _worker_config4(rank, world, port, q):
    """BASELINE config 4's flow at toy size on the emulator: one global batch of counter-generated sets, contiguous
    shards, rank 0 asks the cost model for the program of a shard and broadcasts it (checksummed blob), every rank
    evaluates its shard and contributes per-set checksums; the hash of the gathered list must equal the single-process one."""
    import hashlib
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cwc_import
    pkg = cwc_import.load()
    from circom_witnesscalc_amd import dist as cdist
    import cwc_import
    C = cwc_import.load().graphgen.circuits
    from tools.synth import synth_inputs
    import program_emulator as pe
    data = C.build_gadgets().to_bin()
    total, per = 14, 7
    key, blob = 0, b""
    if rank == 0:
        g = pkg.Graph(data)
        key = g.pick_tile_width(per)          # host-only: the cost model needs no device
        blob = g.export_blob(key)
    got = cdist.broadcast_blob(blob, src=0, device="cpu")
    prog = pe.Blob(got)

    def evaluate(lo, hi):
        rows = synth_inputs("field", prog.n_inputs, hi - lo, 0xC1C00004, lo)
        wit = np.zeros((hi - lo, prog.n_witness, 32), dtype=np.uint8)
        for s in range(hi - lo):
            vals, status = pe.run(prog, [int.from_bytes(rows[s, k].tobytes(), "little") for k in range(prog.n_inputs)])
            if status == 0:
                wit[s] = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)
        return cdist.set_checksums(torch.from_numpy(wit))
    lo, hi = cdist.shard_range(total, rank, world)
    cs = evaluate(lo, hi)
    parts = [torch.empty(cdist.shard_range(total, r, world)[1] - cdist.shard_range(total, r, world)[0], dtype=torch.int64) for r in range(world)]
    dist.all_gather(parts, cs)
    digest = hashlib.sha256(torch.cat(parts).numpy().tobytes()).hexdigest()
    single = hashlib.sha256(evaluate(0, total).numpy().tobytes()).hexdigest() if rank == 0 else None
    dist.barrier()
    q.put((rank, digest, single, prog.T, lo, hi))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_config4_flow_world2_digest_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_config4, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == res[0][2]                      # both ranks see the same digest = the single-process digest
    assert (res[0][4], res[0][5], res[1][4], res[1][5]) == (0, 7, 7, 14) and res[0][3] == res[1][3]


@pytest.mark.timeout(300)
def test_bench_self_launch_two_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher around it must start its two ranks itself (a child
    torch.distributed.run) and print ONE line that says n_gpus == 2: the 8-GPU scaling run cannot silently record one
    GPU.  --dry-run: gloo + the program emulator, no GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["group_ranks"] == 2 and j["dry_run"] is True
    assert j["matches_single_gpu_digest"] is True and j["sets_with_error_status"] == 0
    assert "x2" in j["config"]["parallelism"]


@pytest.mark.timeout(420)
def test_bench_self_launch_eight_ranks_dry_run():
    """The launch the driver makes at round end -- `python bench.py --gpus 8` -- rehearsed on CPU with all eight ranks: one line,
    n_gpus == 8, a process group of eight, eight contiguous shards whose gathered checksums hash to the single-process digest,
    and the line's self-validation fields (the N = 1 step of the same run, efficiency_vs_n1) computed.  gloo + the program
    emulator: the launch path and the shard arithmetic, not a measurement."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--dry-run", "--batch-per-gpu", "3"],
                       capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["group_ranks"] == 8 and j["dry_run"] is True
    assert j["matches_single_gpu_digest"] is True and j["sets_with_error_status"] == 0
    assert "x8" in j["config"]["parallelism"] and j["efficiency_vs_n1"] is not None and j["n1_ms_per_step_same_run"] > 0


@pytest.mark.timeout(300)
def test_bench_rank_dying_before_the_broadcast_fails_the_job():
    """A rank that dies in front of the program broadcast must end the job with a non-zero exit code -- the launcher ends the
    ranks that wait in the collective, and the watchdog around the broadcast ends them by itself if the launcher does not --
    within the watchdog's time, and no JSON line may be printed."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_DRYRUN_DIE_RANK="1", BENCH_BCAST_TIMEOUT="20", OMP_NUM_THREADS="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0, r.stdout
    assert time.time() - t0 < 150
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_rejects_world_size_mismatch():
    """a launcher with fewer ranks than --gpus says (or none, with RANK set) is an error, not a silent one-GPU run"""
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE 1 != --gpus 2" in r.stderr


def test_bench_watchdog_exits_nonzero():
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(0.3, 'test block'):\n"
            "    time.sleep(30)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "watchdog: test block did not finish" in r.stderr
