#!/usr/bin/env python3
"""Benchmark of the calc-witness hot path on MI355X (see BASELINE.json / SURVEY.md 8(d)).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path (interpreter + witness pack kernels) over one batch of synthetic input
sets already resident in HBM: BASELINE config 2 -- authV2-class graph, 1024 input sets per GPU (weak scaling:
every rank evaluates its own 1024 sets; the compiled graph program is broadcast once from rank 0 over RCCL,
there is no data-path collective).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def synth_inputs(graph_kind, n_inputs, batch, seed, first_row=None):
    """Synthetic input sets, canonical 32-byte LE rows: uniform values below 2^253 (< r) for authV2-class,
    uniform bits for sha256; slot 0 = 1.  Counter-based: depends only on (seed, batch)."""
    rng = np.random.default_rng(seed)
    rows = np.frombuffer(rng.bytes(batch * n_inputs * 32), dtype=np.uint8).reshape(batch, n_inputs, 32).copy()
    if graph_kind == "sha256":
        rows[:, :, 1:] = 0
        rows[:, :, 0] &= 1
    else:
        rows[:, :, 31] &= 0x1F
    rows[:, 0, :] = 0
    rows[:, 0, 0] = 1
    if first_row is not None:
        rows[0] = first_row
    return rows


def main():
    # Everything except the final JSON line goes to stderr (RCCL prints a version banner on stdout at init).
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=1024)
    ap.add_argument("--graph", choices=["authv2", "sha256"], default="authv2")
    ap.add_argument("--tile-width", type=int, default=0, help="0 = library heuristic")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="input sets timed on one host core (0 = skip)")
    ap.add_argument("--extra-batch", type=int, default=8192,
                    help="also report (outside `value`) the throughput at this per-GPU batch, N=1 only; 0 = skip")
    ap.add_argument("--host-path", type=int, default=1,
                    help="also report (outside `value`) the PCIe-inclusive rate of the host-buffer entry point, N=1 only")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    # under torch.distributed.run (RANK set) the distributed path is taken even with a single rank, so that the
    # RCCL init / program broadcast / import code is exercised on 1-GPU boxes too
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group("nccl", device_id=dev)

    import cwc_import
    pkg = cwc_import.load()
    if not os.path.exists(pkg.LIB_PATH):
        if rank == 0:
            pkg.build()
        if distributed:
            dist.barrier()
    from circom_witnesscalc_amd import dist as cdist
    from tools.graphgen import circuits as C
    from tools.graphgen.builder import graph_stats

    B = args.batch_per_gpu
    # ---- graph: generated on rank 0, compiled there, program broadcast over RCCL (xGMI) ----
    data = None
    stats = None
    if rank == 0:
        builder = C.build_authv2_class() if args.graph == "authv2" else C.build_sha256(512)
        nodes, wit, _ = builder.finalize()
        stats = graph_stats(nodes, wit)
        data = builder.to_bin()
    if distributed:
        tile = args.tile_width or pkg.pick_tile_width(B)  # same on every rank
        g = cdist.broadcast_graph(pkg, data, tile, src=0, device=dev)
    else:
        g = pkg.Graph(data)
        g.set_tile_width(args.tile_width)

    first_row = None
    if rank == 0 and args.graph == "authv2":  # set 0 = the reference's own input file through the JSON path
        first_row = g.inputs_from_json(open(os.path.join(ROOT, "tests", "golden", "circuit9_authV2_inputs.json")).read())
    rows = synth_inputs(args.graph, g.n_inputs, B, 0xC1C00002 + rank, first_row)
    d_in = torch.from_numpy(rows).to(dev)
    d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)

    def step():
        g.calc_witness_batch_device(d_in, d_out, d_st)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if os.environ.get("BENCH_SYNC_EACH_STEP"):  # diagnostic: the host waits for every step (exposes launch latency)
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    bad_sets = int((d_st != 0).sum().item())
    # HIP events the library recorded on the launch stream around each kernel of the timed steps (read after the run:
    # no synchronization inside the timed region)
    tm = g.last_timing()
    interp_ms, pack_ms = g.timing_history(args.steps * tm["n_launches"])
    assert len(interp_ms) == min(args.steps * tm["n_launches"], 256)
    interp_ms, pack_ms = interp_ms * tm["n_launches"], pack_ms * tm["n_launches"]  # per step

    if rank == 0:
        total_sets = world * B * args.steps
        value = total_sets / elapsed
        avg_interp_s = float(np.mean(interp_ms)) * 1e-3
        alg_bytes = g.algorithmic_bytes_per_set * B  # per launch
        achieved = alg_bytes / avg_interp_s / 1e9
        out = {
            "metric": "witnesses/sec", "value": value, "unit": "witnesses/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "authV2-class graph, %d input sets per GPU (BASELINE config 2)" % B if args.graph == "authv2"
                       else "sha256_512 graph, %d input sets per GPU (BASELINE config 3)" % B,
                       "graph": "generated (tools/graphgen): real circom graphs cannot be built offline",
                       "n_nodes": g.n_nodes, "n_op": g.n_op, "n_witness": g.n_witness, "depth": g.depth,
                       "op_histogram": stats["hist"], "batch_per_gpu": B, "tile_width": tm["tile_width"],
                       "interpreter_waves_per_divider_wave": tm["divider"],
                       "bundles": tm["n_bundles"], "slots": tm["n_slots"], "sets_with_error_status": bad_sets,
                       "parallelism": "batch shards x%d, program broadcast over RCCL" % world if distributed
                       else "single process, 1 GPU"},
            "field_ops_per_sec": value * g.n_op,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": committed_traffic(args.graph, B, tm["tile_width"]), "kernel": "interp_kernel<T=%d>" % tm["tile_width"],
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_interp_s * 1e3,
                         "pack_kernel_avg_ms": float(np.mean(pack_ms))},
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(data, rows, d_out, min(args.cpu_sample, B))
        if world == 1 and args.host_path:
            out["pcie_inclusive"] = host_path_point(pkg, g, rows, d_out)
        if world == 1 and args.extra_batch > B:
            out["large_batch"] = large_batch_point(pkg, g, args.graph, args.extra_batch, dev)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def host_path_point(pkg, g, rows, d_out):
    """Informational only (never `value`): the same batch through gwb_calc_witness_batch_host -- input rows from host
    memory, witness rows back into a pinned host buffer (gwb_host_alloc) and into a reused pageable one."""
    res = {"unit": "witnesses/s", "witness_bytes": int(rows.shape[0]) * g.n_witness * 32}
    want = d_out.cpu().numpy()
    for name, out in (("pinned_destination", pkg.pinned_rows(want.shape)), ("pageable_destination", np.zeros_like(want))):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            wit, st = g.calc_witness_batch(rows, out=out)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        res[name] = {"value": rows.shape[0] / best, "ms_per_step": best * 1e3, "matches_device_path": bool(np.array_equal(wit, want))}
    return res


def large_batch_point(pkg, g, graph_kind, batch, dev):
    """Informational only (never part of `value`): the same kernel at the per-GPU batch of BASELINE config 4
    (65536 sets over 8 GPUs = 8192 per GPU), device-resident, library-chosen tile width."""
    rows = synth_inputs(graph_kind, g.n_inputs, batch, 0xC1C00004)
    d_in = torch.from_numpy(rows).to(dev)
    d_out = torch.empty((batch, g.n_witness, 32), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(batch, dtype=torch.int32, device=dev)
    g.set_tile_width(0)
    g.calc_witness_batch_device(d_in, d_out, d_st)
    torch.cuda.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        g.calc_witness_batch_device(d_in, d_out, d_st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = g.last_timing()
    achieved = g.algorithmic_bytes_per_set * batch / (tm["interp_ms"] * 1e-3) / 1e9
    return {"batch_per_gpu": batch, "value": batch / dt, "unit": "witnesses/s", "ms_per_step": dt * 1e3,
            "tile_width": tm["tile_width"], "interpreter_waves_per_divider_wave": tm["divider"], "launches": tm["n_launches"],
            "roofline_frac": achieved / HBM_PEAK_GBS, "sets_with_error_status": int((d_st != 0).sum().item())}


def committed_traffic(graph_kind, batch, tile_width):
    """HBM bytes per interpreter launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_summary.json,
    collected on this same command line), or None when no committed profile matches this configuration."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))):
        try:
            j = json.load(open(f))
        except (OSError, ValueError):
            continue
        c = j.get("config", {})
        if c.get("graph") == graph_kind and c.get("batch_per_gpu") == batch and c.get("tile_width") == tile_width:
            best = j["kernels"]["interp"]["hbm_bytes_per_launch_corrected"]
    return best


def cpu_baseline(graph_data, rows, d_out, n):
    """The oracle's C restatement of the reference evaluate() (sequential, scalar 4x64 Montgomery, one core),
    timed on a bounded sample of the same input sets; also cross-checks those sets against the GPU witnesses."""
    from oracle import cbind
    og = cbind.Graph(graph_data)
    og.evaluate_batch(rows[:2])  # warm caches / page in
    t, want, st = og.time_batch(rows[:n])
    got = d_out[:n].cpu().numpy()
    ok = st == 0
    out = {"value": n / t, "unit": "witnesses/s", "cores": 1, "kind": "port",
           "sample": "first %d input sets of rank 0's batch, graph parsed once outside the timed window (B1 of BASELINE.md)" % n,
           "seconds": t, "matches_gpu": bool(np.array_equal(got[ok], want[ok]))}
    # reported beside it: the reference really re-parses the .bin per call (lib.rs:129) ...
    t0 = time.perf_counter()
    for _ in range(3):
        cbind.Graph(graph_data)
    t_parse = (time.perf_counter() - t0) / 3
    out["including_parse_per_call"] = {"value": 1.0 / (t / n + t_parse), "unit": "witnesses/s", "parse_seconds": t_parse}
    # ... and a courtesy upper bound: the same loop on every host core (the reference is single-threaded)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if cores > 1:
        tt, want_mt, st_mt = cbind.time_batch_threads(og, rows[:n], cores)
        out["all_cores"] = {"value": n / tt, "unit": "witnesses/s", "cores": cores, "seconds": tt,
                            "matches_one_core": bool(np.array_equal(want_mt, want) and np.array_equal(st_mt, st))}
    return out


if __name__ == "__main__":
    main()
