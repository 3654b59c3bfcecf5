#!/usr/bin/env python3
"""Benchmark of the calc-witness hot path on MI355X (see BASELINE.json / SURVEY.md 8(d)).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path (interpreter + witness pack kernels) over one batch of synthetic input sets already
resident in HBM.  `value` is BASELINE config 2 -- authV2-class graph, 1024 input sets per GPU, weak scaling: every rank
evaluates its own shard of a global batch of 1024 x N counter-generated sets; rank 0 compiles, asks the cost model for
the program of a 1024-set shard and broadcasts that program once over RCCL (no data-path collective).  Prints ONE JSON
line on rank 0.  Outside `value`, the same line carries (N = 1 unless said otherwise):

  roofline.compute   the ceiling that binds (instruction issue): modmul-equivalents/s against the chip's one-lane
                     Montgomery product rate measured in this run (gwb_ubench_modmul)
  cpu_baseline       the oracle's C port of the reference evaluate() on one host core, same input sets, byte-compared
  config3            BASELINE config 3: sha256_512 graph, 4096 sets, every digest checked against hashlib
  config4_per_gpu    BASELINE config 4's per-GPU share: 8192 authV2-class sets, sampled sets against the oracle
  config4            (every N) the config-4 job itself: a global batch of 8192 x N sets, contiguous shards, per-set checksums
                     of all ranks hashed and compared with the same job recomputed on rank 0 alone
  config5            BASELINE config 5 at its named size on the builder's first generator: the 10.5 M-node bigint / long_div-class graph (64-bit
                     limbs x 32, division by ONE limb), 32 sets on this GPU (3 timed steps, 4 sets against the oracle = its CPU baseline sample)
                     and, in `all_256_sets_on_one_gpu`, the whole 256-set batch on this one GPU
  config5_rsa        the same on the class BASELINE names: the zk-email RSA / long_div-class graph, 121-bit registers x 17, long_div by the
                     17-register modulus (circom-bigint's witness hints), 310 chained modular multiplications = 10.0 M nodes
  json_front_end     sets/s of the batched NDJSON -> rows front-end
  e2e_json_to_wtns   NDJSON -> `.wtns` files on tmpfs through the streaming pipeline (parse | kernels | D2H slices | writers)
  pcie_inclusive     the host-buffer entry point (never `value`)

`--config 3` / `--config 4` / `--config 5` make one of those the timed `value` instead; CWC_GRAPH_BIN=<file.bin> runs a
real graph (synthetic inputs by its input count) in place of the generated authV2-class one.

`--gpus N` with N > 1 and no RANK in the environment starts the N ranks itself: a CHILD `python -m torch.distributed.run
--nproc-per-node N bench.py ...` spawned before anything touches a GPU, its JSON line relayed (never an exec).  `--dry-run`
runs the same multi-rank flow on CPU (gloo, the program emulator of tests/ instead of the kernels, a toy graph): what the
world-size-2 test of the launch path uses.
"""
import argparse
import hashlib
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
SHADER_CLOCK_GHZ = 2.4  # MI355X peak engine clock (same guide): what the per-class cycle measurements of the cost model are in
SEED = 0xC1C00000      # + config number
# One-lane Montgomery products a node costs: Mul and Fr::new 1; a conversion out of Montgomery form is the
# reduction half alone (0.5); Div = safegcd inversion (round 3: 52.2 k lone-wave cycles against 1.44 k for a product,
# profiles/r03_inv_bench.txt; round 2 counted 53 for the 73.5 k-cycle inversion) + 2 products.
MODMUL_EQ = {"Mul": 1.0, "Input": 1.0, "Div": 38.0, "Lt": 1.0, "Gt": 1.0, "Leq": 1.0, "Geq": 1.0, "Shl": 2.0, "Shr": 2.0,
             "Band": 2.0, "Bor": 2.0, "Bxor": 2.0, "Idiv": 6.0, "Mod": 6.0}


CPU_SAMPLE = 1024  # --cpu-sample (0: no CPU baselines anywhere in the line)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def modmul_equivalents(hist, n_witness):
    return sum(MODMUL_EQ.get(k, 0.0) * v for k, v in hist.items()) + 0.5 * n_witness


class Workload:
    """graph bytes + what is needed to make inputs for it and to check its witnesses"""

    def __init__(self, kind):
        import cwc_import
        pkg = cwc_import.load()
        C = pkg.graphgen.circuits
        graph_stats = pkg.graphgen.builder.graph_stats
        self.kind = kind
        path = os.environ.get("CWC_GRAPH_BIN") if kind == "authv2" else None
        if path:
            from oracle import model
            self.data = open(path, "rb").read()
            nodes, wit, self.inputs = model.deserialize_witnesscalc_graph(self.data)
            self.stats = graph_stats(nodes, wit)
            self.name = "graph file %s" % os.path.basename(path)
            self.source = "CWC_GRAPH_BIN"
        else:
            self.name = {"authv2": "authV2-class graph", "sha256": "sha256_512 graph", "bigint": "bigint / long_div-class graph (64-bit limbs x 32, division by one limb)",
                         "rsa": "zk-email RSA / long_div-class graph (121-bit registers x 17, long_div by the 17-register modulus)"}[kind]
            if kind in ("bigint", "rsa"):
                # BASELINE config 5's two class graphs at the named size, ten million nodes each, from the native generators (gwb_graphgen_*:
                # the same bytes as the Python generator library writes, tests/test_host_formats.py; ~2 s instead of ~20 s).  bigint: 32 limbs
                # x 4000 rounds of multiply / long-divide by one limb / compare; rsa: 310 chained modular multiplications = 18 RSA-65537
                # exponentiations as circom-bigint's witness hints compute them.  Statistics come from the loaded handle (stats_from).
                self.data = pkg.graphgen_native("bigint", k=32, n_bits=64, rounds=4000) if kind == "bigint" else pkg.graphgen_native("rsa", n=121, k=17, muls=310)
                self.inputs, self.stats = None, None
                self.source = "generated natively (gwb_graphgen_*; specification: circom-witnesscalc_amd/graphgen/circuits.py, byte-equal): the reference front-end cannot compile such circuits"
            else:
                builder = C.build_authv2_class() if kind == "authv2" else C.build_sha256(512)
                nodes, wit, self.inputs = builder.finalize()
                self.stats = graph_stats(nodes, wit)
                self.data = builder.to_bin()
                self.source = "generated (circom-witnesscalc_amd/graphgen, written through the C-ABI producer gwb_builder_*): real circom graphs cannot be built offline"
        self.input_kind = "bits" if kind == "sha256" else "field"

    def stats_from(self, g):
        """statistics of a natively generated graph from its loaded handle (no Python node list exists)"""
        if self.stats is None:
            self.stats = {"N": g.n_nodes, "N_op": g.n_op, "W": g.n_witness, "depth": g.depth, "hist": g.op_histogram()}
        return self.stats

    def first_row(self, g):
        if self.kind == "authv2" and self.source != "CWC_GRAPH_BIN":  # global set 0 = the reference's own input file through the JSON path
            return g.inputs_from_json(open(os.path.join(ROOT, "tests", "golden", "circuit9_authV2_inputs.json")).read())
        return None


def make_inputs(wl, g, batch, config, first_set=0):
    from tools.synth import synth_inputs
    return synth_inputs(wl.input_kind, g.n_inputs, batch, SEED + config, first_set, wl.first_row(g) if first_set == 0 else None)


def run_steps(g, d_in, d_out, d_st, steps, warmup, distributed):
    for _ in range(warmup):
        g.calc_witness_batch_device(d_in, d_out, d_st)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.calc_witness_batch_device(d_in, d_out, d_st)
        if os.environ.get("BENCH_SYNC_EACH_STEP"):  # diagnostic: the host waits for every step (exposes launch latency)
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def kernel_times(g, steps):
    """HIP events the library recorded on the launch stream around each kernel of the timed steps (read after the run:
    no synchronization inside the timed region) -> per-step interpreter / pack milliseconds"""
    tm = g.last_timing()
    interp_ms, pack_ms = g.timing_history(steps * tm["n_launches"])
    assert len(interp_ms) == min(steps * tm["n_launches"], 256)
    return tm, interp_ms * tm["n_launches"], pack_ms * tm["n_launches"]


class Watchdog:
    """Ends the process (exit code 3, message on stderr) when the guarded block has not finished after `seconds`: a rank that
    never arrives at a collective (RCCL communicator, gwb_graph_broadcast) must fail the job, not hang it -- the launcher then
    ends the other ranks.  Never re-execs anything."""

    def __init__(self, seconds, what):
        import threading
        self.t = threading.Timer(seconds, self.fire)
        self.t.daemon = True
        self.what, self.seconds = what, seconds

    def fire(self):
        try:
            os.write(2, ("bench.py: watchdog: %s did not finish within %.0f s (rank %s); exiting\n" % (self.what, self.seconds, os.environ.get("RANK", "0"))).encode())
        finally:
            os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def self_launch(n, argv, real_stdout):
    """`python bench.py --gpus N` (N > 1, no RANK): the ranks are started by a child `python -m torch.distributed.run` -- the
    same command line the driver would use -- and its stdout (rank 0's one JSON line) is relayed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    out, _ = p.communicate()
    line = None
    for ln in out.decode("utf-8", "replace").splitlines():  # (the launcher may print warnings of its own on stdout)
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            log(ln)
    if line is not None:
        os.write(real_stdout, (line + "\n").encode())
    elif p.returncode == 0:
        log("bench.py: the ranks exited 0 without a JSON line")
        return 4
    return p.returncode


def dry_run(args, real_stdout):
    """CPU rehearsal of the distributed bench (world-size-N flow without a GPU): gloo process group, a toy graph (the gadget
    graph: every op class), rank 0 compiles the cost model's program for a shard and broadcasts the blob, every rank
    evaluates ITS shard of one global batch on the program emulator (tests/program_emulator.py -- the test suite's model
    of the interpreter; the product has no CPU path), per-set checksums are gathered and hashed, rank 0 recomputes the job
    alone.  The line has the shape of the real one; its numbers are not measurements of anything."""
    import cwc_import
    pkg = cwc_import.load()
    from circom_witnesscalc_amd import dist as cdist
    from tools.synth import synth_inputs
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import program_emulator as pe
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if distributed:
        dist.init_process_group("gloo")
    per = args.batch_per_gpu or 6
    total = per * world
    blob, key = b"", 0
    if rank == 0:
        g = pkg.Graph(pkg.graphgen.circuits.build_gadgets().to_bin())
        key = g.pick_tile_width(per)   # host-only: the cost model needs no device
        blob = g.export_blob(key)
    if os.environ.get("BENCH_DRYRUN_DIE_RANK") == str(rank):  # (test hook: a rank that dies in front of the collective must fail the job, not hang it)
        log("bench.py --dry-run: rank %d exits before the broadcast (BENCH_DRYRUN_DIE_RANK)" % rank)
        os._exit(7)
    if distributed:
        with Watchdog(float(os.environ.get("BENCH_BCAST_TIMEOUT", "120")), "program broadcast (gloo)"):
            blob = cdist.broadcast_blob(blob, src=0, device="cpu")
    prog = pe.Blob(blob)

    def evaluate(lo, hi):
        rows = synth_inputs("field", prog.n_inputs, hi - lo, SEED + 4, lo)
        wit = np.zeros((hi - lo, prog.n_witness, 32), dtype=np.uint8)
        bad = 0
        for k in range(hi - lo):
            vals, status = pe.run(prog, [int.from_bytes(rows[k, q].tobytes(), "little") for q in range(prog.n_inputs)])
            bad += status != 0
            if status == 0:
                wit[k] = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)
        return cdist.set_checksums(torch.from_numpy(wit)), bad
    lo, hi = cdist.shard_range(total, rank, world)
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        cs, bad = evaluate(lo, hi)
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        te = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        parts = [torch.empty(cdist.shard_range(total, r, world)[1] - cdist.shard_range(total, r, world)[0], dtype=torch.int64) for r in range(world)]
        dist.all_gather(parts, cs)
        cs_all = torch.cat(parts)
        group_ranks = dist.get_world_size()
    else:
        cs_all, group_ranks = cs, 1
    n1_ms = None
    if distributed:  # the N = 1 step of the same run, as the real line has it: rank 0 alone repeats its shard while the others wait
        if rank == 0:
            t1 = time.perf_counter()
            for _ in range(max(1, args.steps)):
                evaluate(lo, hi)
            n1_ms = (time.perf_counter() - t1) / max(1, args.steps) * 1e3
        dist.barrier()
    if rank == 0:
        digest = hashlib.sha256(cs_all.numpy().tobytes()).hexdigest()
        single = hashlib.sha256(evaluate(0, total)[0].numpy().tobytes()).hexdigest()
        steps = max(1, args.steps)
        out = {"metric": "witnesses/sec", "value": total * steps / elapsed, "unit": "witnesses/s", "n_gpus": world, "steps": steps, "warmup": 0,
               "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "python int",
               "data": "synthetic", "dry_run": True,
               "config": {"workload": "DRY RUN on CPU: gadget graph, %d sets per rank on the program emulator (launch-path rehearsal, not a measurement)" % per,
                          "tile_width": prog.T, "parallelism": "contiguous shards of one global batch x%d, program blob broadcast over gloo" % world},
               "rccl_ranks": None, "group_ranks": group_ranks, "program_key": key, "n1_ms_per_step_same_run": n1_ms,
               "efficiency_vs_n1": (n1_ms / (elapsed / steps * 1e3)) if n1_ms else None, "digest_of_set_checksums": digest, "single_gpu_digest": single,
               "matches_single_gpu_digest": digest == single, "sets_with_error_status": int(bad)}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    # Everything except the final JSON line goes to stderr (RCCL prints a version banner on stdout at init).
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=2, help="BASELINE config timed as `value`")
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="0 = the config's own (1024 / 4096 / 8192 / 32)")
    ap.add_argument("--graph", choices=["authv2", "sha256"], default=None, help="(old spelling of --config 2 / 3)")
    ap.add_argument("--tile-width", type=int, default=0, help="0 = the library's cost model")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="input sets timed on one host core (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="0 = only the timed metric (no sub-records)")
    ap.add_argument("--extra-batch", type=int, default=None, help=argparse.SUPPRESS)   # (round-1 flags, still accepted)
    ap.add_argument("--host-path", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-selfcheck", action="store_true", help="fail (exit 5) unless the committed PMC summary behind roofline.traffic is of this library's kernel sources "
                                                                 "(the comparison itself always runs: roofline.traffic_kernel_hash_matches)")
    ap.add_argument("--config5-graph", choices=["rsa", "bigint"], default="rsa", help="--config 5: the class BASELINE names (zk-email RSA / long_div: 121-bit registers x 17) or the "
                                                                                       "builder's first generator (64-bit limbs x 32, division by one limb)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the multi-rank flow: gloo, program emulator, toy graph (no GPU, no timing claim)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # One command for the N-GPU run: start the ranks as a CHILD launcher (nothing in this process has touched a GPU:
        # `import torch` does not initialise one) and relay its line.  Never an exec.
        sys.exit(self_launch(args.gpus, sys.argv[1:], real_stdout))
    if args.dry_run:
        return dry_run(args, real_stdout)
    if args.graph == "sha256" and args.config == 2:
        args.config = 3
    if args.extra_batch == 0 and args.host_path == 0:
        args.extras = 0
    global CPU_SAMPLE
    CPU_SAMPLE = args.cpu_sample
    cfg = args.config
    kind = "sha256" if cfg == 3 else args.config5_graph if cfg == 5 else "authv2"
    B = args.batch_per_gpu or {2: 1024, 3: 4096, 4: 8192, 5: 32}[cfg]
    if cfg == 5:
        args.extras = 0  # (the sub-records belong to the default run)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d: launch with torch.distributed.run --nproc-per-node == --gpus (or plain `python bench.py --gpus N`, which does that itself)" % (world, args.gpus)
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

    # ---- graph: built on rank 0, compiled there, the cost model's program for a B-set shard broadcast over RCCL (xGMI) ----
    wl = Workload(kind) if rank == 0 else None
    rccl_ranks = None
    if distributed:
        # the program goes out through the C-ABI collective (gwb_graph_broadcast on a RCCL communicator of this job's ranks);
        # a rank that never arrives must not hang the job: the watchdog ends this process (non-zero) and the launcher the rest
        with Watchdog(float(os.environ.get("BENCH_BCAST_TIMEOUT", "120")) + (600 if cfg == 5 else 0), "RCCL communicator / gwb_graph_broadcast"):
            g, rccl_ranks = cdist.broadcast_graph_rccl(pkg, wl.data if rank == 0 else None, args.tile_width, src=0, device=dev, batch_per_rank=B)
    else:
        g = pkg.Graph(wl.data)
        g.set_tile_width(args.tile_width)

    # ---- inputs: this rank's contiguous shard of ONE global batch of B x world counter-generated sets ----
    from tools.synth import synth_inputs
    lo, hi = cdist.shard_range(B * world, rank, world)
    first_row = wl.first_row(g) if rank == 0 else None
    rows = synth_inputs("bits" if kind == "sha256" else "field", g.n_inputs, hi - lo, SEED + cfg, lo, first_row)
    d_in = torch.from_numpy(rows).to(dev)
    d_out = torch.empty((hi - lo, g.n_witness, 32), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(hi - lo, dtype=torch.int32, device=dev)

    elapsed = run_steps(g, d_in, d_out, d_st, args.steps, args.warmup, distributed)
    per_rank_ms, n1_ms = None, None
    if distributed:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_t = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(all_t, te)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in all_t]
        elapsed = max(float(x.item()) for x in all_t)
        # the N = 1 step of the same run: rank 0 alone repeats its shard while the other ranks wait at the barrier
        if rank == 0:
            n1_ms = run_steps(g, d_in, d_out, d_st, args.steps, 1, False) / args.steps * 1e3
        dist.barrier()
    bad_sets = int((d_st != 0).sum().item())
    tm, interp_ms, pack_ms = kernel_times(g, args.steps)

    out = None
    if rank == 0:
        value = world * B * args.steps / elapsed
        avg_interp_s = float(np.mean(interp_ms)) * 1e-3
        alg_bytes = g.algorithmic_bytes_per_set * B  # per launch
        achieved = alg_bytes / avg_interp_s / 1e9
        traffic, traffic_source = committed_traffic(kind, B, tm["tile_width"], pkg.kernel_source_hash())
        if args.pmc_selfcheck and PMC_CHECK["same_kernels"] is not True:
            log("bench.py --pmc-selfcheck: %s" % traffic_source)
            sys.exit(5)
        wl.stats_from(g)
        eq_per_set = modmul_equivalents(wl.stats["hist"], g.n_witness)
        # the ceiling that binds: one-lane Montgomery products per second, chip-wide, with the multiplier the interpreter's
        # full-width bundles use (fr_mul_wave, 322 issue slots; its pinned accumulators allow two waves per SIMD) -- beside it
        # the generic 380-slot multiplier at four waves per SIMD (round 2's denominator, which flattered the fraction)
        peak_blk2, peak_blk1 = pkg.ubench_modmul(2, 1000, block=True), pkg.ubench_modmul(1, 1000, block=True)
        peak4 = pkg.ubench_modmul(4, 1000)
        peak = max(peak_blk2, peak_blk1, peak4)
        PEAK["modmul_per_s"] = peak or None
        ps = g.program_stats()
        out = {
            "metric": "witnesses/sec", "value": value, "unit": "witnesses/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s, %d input sets per GPU (BASELINE config %d)" % (wl.name, B, cfg),
                       "graph": wl.source, "inputs": "uniform in [0, r) by rejection (bits for sha256), SplitMix64 of seed 0xC1C0000%d and element id = "
                       "global set index * n_inputs + k; set 0 = the reference's input file" % cfg,
                       "n_nodes": g.n_nodes, "n_op": g.n_op, "n_witness": g.n_witness, "depth": g.depth,
                       "op_histogram": wl.stats["hist"], "batch_per_gpu": B, "tile_width": tm["tile_width"],
                       "interpreter_waves_per_divider_wave": tm["divider"], "streams_per_tile": tm["streams"],
                       "bundles": tm["n_bundles"], "slots": tm["n_slots"], "sets_with_error_status": bad_sets,
                       "parallelism": "contiguous shards of one global batch x%d, cost-model program broadcast over RCCL" % world if distributed
                       else "single process, 1 GPU"},
            "field_ops_per_sec": value * g.n_op,
            "rccl_ranks": rccl_ranks, "per_rank_ms_per_step": per_rank_ms, "n1_ms_per_step_same_run": n1_ms,
            "efficiency_vs_n1": ((n1_ms / (elapsed / args.steps * 1e3)) if n1_ms else None),
            "scaling_note": "weak scaling: every rank evaluates its own 1/N shard of one global batch of B x N sets; efficiency_vs_n1 = "
                            "the step time of rank 0 alone on its shard (same run, other ranks idle) over the slowest rank's step time with "
                            "all ranks busy.  No scaling curve has been measured by the builder: the GPU boxes of this pool have one GPU." if distributed else None,
            # What binds this kernel is instruction issue on lone wavefronts along the graph's dependency chain, not memory: `bound`,
            # `achieved`, `peak`, `frac` are that ceiling (modmul-equivalents/s against the chip's one-lane Montgomery product rate measured
            # in this run); SURVEY 8(d)'s algorithmic-byte model against the HBM peak is the `hbm` block beside it, with the counter traffic.
            "roofline": {"bound": "valu_issue", "unit": "modmul-equivalents/s", "achieved": eq_per_set * B / avg_interp_s, "peak": peak or None,
                         "frac": (eq_per_set * B / avg_interp_s / peak) if peak else None,
                         "traffic": traffic, "traffic_source": traffic_source, "traffic_kernel_hash_matches": PMC_CHECK["same_kernels"],
                         "binding": "valu_issue: lone-wave instruction issue along the graph's dependency chain; the hbm block is SURVEY 8(d)'s "
                                    "algorithmic-byte model, which this kernel does not run into (operands are forwarded on chip: measured traffic is a "
                                    "fraction of the algorithmic bytes)",
                         "kernel": "interp_kernel<T=%d>" % tm["tile_width"], "avg_launch_ms": avg_interp_s * 1e3, "pack_kernel_avg_ms": float(np.mean(pack_ms)),
                         "binding_resource": "valu_issue",
                         "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": alg_bytes,
                                 # the same algorithmic bytes over the whole step (interpreter + pack kernel), and what the counters say the
                                 # memory system really moved over the interpreter's time
                                 "frac_step": alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                 "hbm_measured_frac": (traffic / avg_interp_s / 1e9 / HBM_PEAK_GBS) if traffic else None},
                         "lanes_active_mean": ps["lanes_active_mean"], "values_per_bundle_mean": ps["values_per_bundle_mean"],
                         "lanes_active_note": "of a wavefront's 64 lanes, the mean number that hold work of a node of the graph (the four "
                                              "lanes that share a narrow bundle's product all count), and the mean number of field elements a "
                                              "bundle produces (64 = one per lane); both weighted by the modelled time of the bundles (program "
                                              "statistics).  What a lone wave pays per bundle does not depend on either, which is why "
                                              "frac is what it is",
                         "program": {"class_bundles": ps["class_bundles"], "class_nodes": ps["class_nodes"], "fused_nodes": ps["n_fused_nodes"],
                                     "scan_steps": ps["n_scan_steps"], "model_wave_cycles": ps["model_wave_cycles"]},
                         # the floor of this execution model (one wave walks the whole graph): the compiled graph's longest dependent
                         # chain priced at the BEST measured latency of each operation on a lone wavefront -- arithmetic only, no bundle
                         # front end (four-lane product 704 cycles, addition 290, inversion 52.2 k; compile.cc) -- against the interpreter's
                         # launch time at the shader clock
                         "chain": {"floor_cycles": ps["chain_floor_cycles"], "clock_ghz": SHADER_CLOCK_GHZ,
                                   "floor_ms": ps["chain_floor_cycles"] / (SHADER_CLOCK_GHZ * 1e6),
                                   "achieved_over_floor": ps["chain_floor_cycles"] / (SHADER_CLOCK_GHZ * 1e6) / (avg_interp_s * 1e3),
                                   "model_over_floor": ps["model_wave_cycles"] / ps["chain_floor_cycles"] if ps["chain_floor_cycles"] else None},
                         "compute": {"unit": "modmul-equivalents/s", "achieved": eq_per_set * B / avg_interp_s,
                                     "peak": peak or None, "frac": (eq_per_set * B / avg_interp_s / peak) if peak else None,
                                     "peak_source": "measured in this run, chip-wide one-lane Montgomery products/s: the best of the interpreter's own "
                                                    "multiplier (fr_mul_wave, 322 issue slots) at 2 and 1 waves per SIMD and the generic 380-slot "
                                                    "multiplier at 4 waves per SIMD",
                                     "peak_interpreter_multiplier_2_waves_per_simd": peak_blk2 or None,
                                     "peak_interpreter_multiplier_1_wave_per_simd": peak_blk1 or None,
                                     "peak_generic_multiplier_4_waves_per_simd": peak4 or None,
                                     "modmul_equivalents_per_set": eq_per_set}},
        }
        if world == 1 and args.cpu_sample > 0:
            t0 = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(wl.data, rows, d_out, min(args.cpu_sample, B))
            log("cpu_baseline: %.1f s" % (time.perf_counter() - t0))
    if args.extras:
        extras(pkg, cdist, wl, g, cfg, rows, d_out, dev, rank, world, distributed, out)
        if rank == 0 and isinstance(out.get("config4"), dict) and "matches_single_gpu_digest" in out["config4"]:
            out["matches_single_gpu_digest"] = out["config4"]["matches_single_gpu_digest"]
    if rank == 0:
        flatten_roofline(out, cfg)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


SCHEMA = 6  # (advisor, round 5) 5: roofline.bound / frac = the VALU-issue ceiling, `--config 5` = the RSA-class graph; 6: + the flat scalars below


def flatten_roofline(out, cfg=2):
    """The driver's record keeps only the SCALARS of `roofline`: SURVEY 8(d)'s own number (algorithmic bytes / interpreter time / 8 TB/s) and
    the other figures a reader needs therefore also sit there as flat keys, next to the nested blocks they come from (tools/design_table.py
    reads these keys)."""
    rf = out.get("roofline")
    if not isinstance(rf, dict):
        return
    out["schema"] = SCHEMA
    hbm, chain, comp = rf.get("hbm", {}), rf.get("chain", {}), rf.get("compute", {})
    rf["hbm_frac"] = hbm.get("frac")                      # algorithmic bytes per launch / the interpreter's launch time / 8 TB/s
    rf["hbm_frac_step"] = hbm.get("frac_step")            # ... over the whole step (interpreter + pack)
    rf["hbm_achieved_gbs"] = hbm.get("achieved")
    rf["hbm_measured_frac"] = hbm.get("hbm_measured_frac")  # counter traffic / launch time / 8 TB/s
    rf["traffic_over_algorithmic"] = (rf["traffic"] / hbm["algorithmic_bytes_per_launch"]) if rf.get("traffic") and hbm.get("algorithmic_bytes_per_launch") else None
    rf["chain_achieved_over_floor"] = chain.get("achieved_over_floor")
    rf["chain_floor_ms"] = chain.get("floor_ms")
    pmc = PMC_CHECK.get("interp")
    eq = comp.get("modmul_equivalents_per_set")
    batch = out.get("config", {}).get("batch_per_gpu")
    # of the vector instructions the counters saw per launch (divider waves included), the share the arithmetic alone needs: one 322-slot
    # product per 64 lane-products
    # (configurations 2 and 4: field products are 322-slot Montgomery products there; bit and limb graphs multiply small integers)
    rf["valu_useful_issue_frac"] = (eq * batch / 64.0 * 322.0 / pmc["sq_insts_valu_per_launch"]) if cfg in (2, 4) and pmc and eq and batch and pmc.get("sq_insts_valu_per_launch") else None
    rf["sq_wait_any_frac"] = (pmc["sq_wait_any_per_launch"] / pmc["sq_wave_cycles_per_launch"]) if pmc and pmc.get("sq_wave_cycles_per_launch") else None
    for name in ("config3", "config4_per_gpu", "config5", "config5_rsa"):
        sub = out.get(name)
        if not isinstance(sub, dict) or "value" not in sub:
            continue
        rf[name + "_value"] = sub["value"]
        rf[name + "_ms_per_step"] = sub.get("ms_per_step")
        c = sub.get("compute")
        rf[name + "_compute_frac"] = c.get("frac") if isinstance(c, dict) else None
        rf[name + "_hbm_frac"] = sub.get("roofline_frac")
        if sub.get("traffic") is not None:
            rf[name + "_traffic"] = sub["traffic"]
            rf[name + "_traffic_over_algorithmic"] = sub.get("traffic_over_algorithmic")
    ss = out.get("single_shot")
    if isinstance(ss, dict):
        rf["single_shot_warm_ms"] = ss.get("warm_call_ms_median_of_last_5")
        rf["single_shot_first_call_ms"] = ss.get("first_call_ms")


def extras(pkg, cdist, wl, g, cfg, rows, d_out, dev, rank, world, distributed, out):
    """sub-records outside `value` (every rank takes part in config4, the rest is rank 0 of a single-GPU run)"""
    t_all = time.perf_counter()
    if world == 1 and rank == 0:
        if cfg == 2:
            out["pcie_inclusive"] = host_path_point(pkg, g, rows, d_out)
            out["json_front_end"] = json_front_end_point(wl, g)
            if wl.source != "CWC_GRAPH_BIN":
                out["single_shot"] = single_shot_point(pkg, wl)
            t0 = time.perf_counter()
            out["e2e_json_to_wtns"] = e2e_json_to_wtns_point(wl, g)
            log("e2e_json_to_wtns: %.1f s" % (time.perf_counter() - t0))
        del d_out
        torch.cuda.empty_cache()
        if cfg != 3:
            t0 = time.perf_counter()
            out["config3"] = config3_point(pkg, dev, cpu_sample=min(256, CPU_SAMPLE))
            log("config3: %.1f s" % (time.perf_counter() - t0))
        if cfg != 4 and wl.kind == "authv2":
            t0 = time.perf_counter()
            out["config4_per_gpu"] = config4_per_gpu_point(pkg, wl, dev)
            log("config4_per_gpu: %.1f s" % (time.perf_counter() - t0))
        if cfg != 5 and os.environ.get("BENCH_SKIP_CONFIG5") is None:
            t0 = time.perf_counter()
            try:
                out["config5"] = config5_point(pkg, dev, "bigint")
            except Exception as e:  # (a sub-record: its failure is reported, not fatal)
                out["config5"] = {"error": "%s: %s" % (type(e).__name__, e)}
            log("config5: %.1f s" % (time.perf_counter() - t0))
            torch.cuda.empty_cache()
            t0 = time.perf_counter()
            try:   # the class BASELINE names: zk-email RSA / long_div (121-bit registers x 17, multi-register divisor)
                out["config5_rsa"] = config5_point(pkg, dev, "rsa")
            except Exception as e:
                out["config5_rsa"] = {"error": "%s: %s" % (type(e).__name__, e)}
            log("config5_rsa: %.1f s" % (time.perf_counter() - t0))
    if wl is None or wl.kind == "authv2" or world > 1:
        t0 = time.perf_counter()
        rec = config4_job(pkg, cdist, wl, dev, rank, world, distributed)
        if rank == 0:
            out["config4"] = rec
            log("config4 job: %.1f s" % (time.perf_counter() - t0))
    if rank == 0:
        out["extras_seconds"] = time.perf_counter() - t_all


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


def single_shot_point(pkg, wl, shots=12):
    """Informational only (never `value`): the reference's own entry point gw_calc_witness (one input set per call, graph
    image and inputs JSON in, `.wtns` bytes out; reference lib.rs:125-136) on the reference's authV2 input file.  The first
    call on a graph parses and compiles it (a quick program; the search for the best one runs in the background and the
    calls switch over when it is done); later calls find the handle by the image's bytes.  The on-disk program cache is off
    for this record (CWC_PROGRAM_CACHE=0 in its child process) so that the first call is a real one."""
    import subprocess
    code = (
        "import sys, time, json, os\n"
        "sys.path.insert(0, %r)\n"
        "import cwc_import\n"
        "pkg = cwc_import.load()\n"
        "from oracle import cbind\n"
        "data = pkg.graphgen.circuits.build_authv2_class().to_bin()\n"
        "js = open(os.path.join(%r, 'tests', 'golden', 'circuit9_authV2_inputs.json')).read()\n"
        "ms = []\n"
        "for i in range(%d):\n"
        "    t0 = time.perf_counter(); w = pkg.calc_witness_wtns(js, data); ms.append((time.perf_counter() - t0) * 1e3)\n"
        "    if i < 6: time.sleep(0.4)\n"
        "import numpy as np\n"
        "og = cbind.Graph(data)\n"
        "row = np.asarray(pkg.Graph(data).inputs_from_json(js), dtype=np.uint8).reshape(1, og.n_inputs, 32)\n"
        "want, st = og.evaluate_batch(row)\n"
        "ok = bool(st[0] == 0 and w == cbind.wtns_from_witness(want[0]))\n"
        "print(json.dumps({'ms': ms, 'ok': ok, 'bytes': len(w)}))\n") % (ROOT, ROOT, shots)
    env = dict(os.environ, CWC_PROGRAM_CACHE="0")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # (a sub-record: its failure is reported, not fatal)
        return {"error": "%s: %s" % (type(e).__name__, e)}
    warm = sorted(rec["ms"][-5:])
    return {"unit": "ms per gw_calc_witness call", "first_call_ms": rec["ms"][0], "warm_call_ms_median_of_last_5": warm[2], "calls_ms": [round(x, 2) for x in rec["ms"]],
            "wtns_bytes": rec["bytes"], "matches_oracle_wtns": rec["ok"], "note": "own process; first call includes the device runtime's start-up"}


def rows_to_ndjson(g_inputs, rows):
    """input rows -> NDJSON text with the graph's own signal names (decimal strings, as the reference's input files)"""
    lines = []
    for r in rows:
        vals = [int.from_bytes(r[k].tobytes(), "little") for k in range(r.shape[0])]
        obj = {name: [str(v) for v in vals[off:off + n]] for name, (off, n) in g_inputs.items()}
        lines.append(json.dumps(obj))
    return "\n".join(lines)


def json_front_end_point(wl, g, n=4096):
    """NDJSON of n input objects -> packed rows (gwb_inputs_from_json_batch, host threads)."""
    from tools.synth import synth_inputs
    src = synth_inputs("field", g.n_inputs, n, SEED + 8)
    text = rows_to_ndjson(wl.inputs, src).encode()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        got = g.inputs_from_json_batch(text)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return {"value": n / best, "unit": "input sets/s", "sets": n, "text_bytes": len(text), "seconds": best,
            "threads": os.environ.get("CWC_PARSE_THREADS") or "all cores (%d)" % (os.cpu_count() or 1), "matches_source_rows": bool(np.array_equal(got, src))}


def e2e_json_to_wtns_point(wl, g, n=8192):
    """SURVEY 8(f) f3 end to end: NDJSON of n input objects -> `.wtns` files on a memory file system, through the streaming
    pipeline of gwb_calc_witness_json_to_wtns (parse threads | upload + kernels | device-to-host slices | writer threads).
    The files of a round are deleted before the next one (8192 x 2.4 MB at once would not be polite to /dev/shm); 32 of them are
    byte-compared with the oracle's `.wtns` first."""
    import shutil
    import tempfile
    from oracle import cbind
    from tools.synth import synth_inputs
    per_round = 4096
    src = synth_inputs("field", g.n_inputs, per_round, SEED + 9)
    text = rows_to_ndjson(wl.inputs, src).encode()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="cwc_e2e_", dir=base)
    try:
        pat = os.path.join(d, "w_%06lu.wtns")
        total_s, rounds, stats, ok_files, bad_status = 0.0, 0, None, True, 0
        og = cbind.Graph(wl.data)
        for r in range(max(1, n // per_round) + 1):  # (round 0 warms the handle up: workspace, program, pinned buffers)
            t0 = time.perf_counter()
            st, stats = g.json_to_wtns(text, pat, first_index=0)
            dt = time.perf_counter() - t0
            bad_status += int((st != 0).sum())
            if r == 0:
                sample = [0, 1, 2, 3, 511, 512, 513, 1023, 1024, 1500, 2046, 2047, 4095] + list(range(3000, 3019))
                want, wst = og.evaluate_batch(src[sample])
                for k, s_ in enumerate(sample):
                    got = open(pat % s_, "rb").read()
                    ok_files = ok_files and got[76:] == want[k].tobytes() and len(got) == 76 + g.n_witness * 32 and not wst[k]
            else:
                total_s += dt
                rounds += 1
            for f in os.listdir(d):
                os.unlink(os.path.join(d, f))
        sets = rounds * per_round
        return {"value": sets / total_s, "unit": "witnesses/s", "sets": sets, "seconds": total_s, "rounds_of": per_round,
                "file_system": "tmpfs (/dev/shm)" if base else "temporary directory", "witness_bytes_per_set": g.n_witness * 32,
                "link_rate_GBs": sets * g.n_witness * 32 / total_s / 1e9, "sub_batch": stats["sub_batch"], "parse_threads": stats["parse_threads"],
                "write_threads": stats["write_threads"], "parse_seconds_last_round": stats["parse_seconds"],
                "wait_for_drain_seconds_last_round": stats["wait_for_drain_seconds"], "sets_with_error_status": bad_status,
                "files_checked_against_oracle": 32, "matches_oracle_files": bool(ok_files)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def timed_batch(g, d_in, d_out, d_st, steps=3, key=0):
    g.set_tile_width(key)
    g.calc_witness_batch_device(d_in, d_out, d_st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.calc_witness_batch_device(d_in, d_out, d_st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm, interp_ms, pack_ms = kernel_times(g, steps)
    return dt, tm, float(np.mean(interp_ms)), float(np.mean(pack_ms))


def compute_block(hist, n_witness, batch, interp_ms):
    """modmul-equivalents/s of a sub-record's interpreter launches against the chip's one-lane Montgomery product rate measured in this
    run (main: PEAK) -- the ceiling that binds, beside the algorithmic-byte figure that stopped telling how far a kernel is from the machine"""
    eq = modmul_equivalents(hist, n_witness)
    ach = eq * batch / (interp_ms * 1e-3)
    peak = PEAK["modmul_per_s"]
    return {"unit": "modmul-equivalents/s", "achieved": ach, "peak": peak, "frac": (ach / peak) if peak else None, "modmul_equivalents_per_set": eq,
            "peak_source": "roofline.compute of this line (measured in this run)"}


def config3_point(pkg, dev, batch=4096, cpu_sample=256):
    """BASELINE config 3: sha256_512, 4096 sets on one GPU; EVERY set's 256 output bits against hashlib (an anchor outside
    this repository's arithmetic), 8 sets against the oracle as whole witnesses."""
    from oracle import cbind
    wl = Workload("sha256")
    g = pkg.Graph(wl.data)
    rows = make_inputs(wl, g, batch, 3)
    d_in = torch.from_numpy(rows).to(dev)
    d_out = torch.empty((batch, g.n_witness, 32), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(batch, dtype=torch.int32, device=dev)
    dt, tm, interp_ms, pack_ms = timed_batch(g, d_in, d_out, d_st)
    # witness layout of the generated graph: [1, out[256], in[512], ...]; bits are MSB first per byte as circomlib's Sha256
    wit_bits = d_out[:, 1:257, 0].cpu().numpy()
    clean = int(d_out[:, 1:257, 1:].max().item()) == 0
    msgs = np.packbits(rows[:, 1:513, 0], axis=1)
    want_bits = np.unpackbits(np.frombuffer(b"".join(hashlib.sha256(m.tobytes()).digest() for m in msgs), dtype=np.uint8).reshape(batch, 32), axis=1)
    ok = int((wit_bits == want_bits).all(axis=1).sum()) if clean else 0
    og = cbind.Graph(wl.data)
    want, st = og.evaluate_batch(rows[:8])
    cpu = None
    if cpu_sample > 0:  # the same port of evaluate() on this graph: one pinned core and every core, on the first sets of the batch
        cpu = cpu_baseline(wl.data, rows, d_out, min(cpu_sample, batch))
        cpu["gpu_over_one_core"] = (batch / dt) / cpu["value"]
        if "all_cores" in cpu:
            cpu["gpu_over_all_cores"] = (batch / dt) / cpu["all_cores"]["value"]
    return {"workload": "sha256_512 graph, %d input sets, 1 GPU (BASELINE config 3)" % batch, "value": batch / dt, "unit": "witnesses/s",
            "cpu_baseline": cpu, "compute": compute_block(wl.stats["hist"], g.n_witness, batch, interp_ms),
            "ms_per_step": dt * 1e3, "interp_kernel_ms": interp_ms, "pack_kernel_ms": pack_ms, "tile_width": tm["tile_width"],
            "n_op": g.n_op, "n_witness": g.n_witness, "bundles": tm["n_bundles"],
            "sets_with_error_status": int((d_st != 0).sum().item()),
            "roofline_frac": g.algorithmic_bytes_per_set * batch / (interp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            **sub_traffic(pkg, "sha256", batch, tm["tile_width"], g.algorithmic_bytes_per_set * batch, interp_ms),
            "matches_hashlib": bool(ok == batch), "sets_checked_against_hashlib": batch,
            "matches_oracle": bool(np.array_equal(d_out[:8].cpu().numpy(), want) and not st.any()), "sets_checked_against_oracle": 8}


def config4_per_gpu_point(pkg, wl, dev, batch=8192):
    """BASELINE config 4's per-GPU share (65536 sets over 8 GPUs): 8192 authV2-class sets on this GPU, library-chosen
    program; 8 sampled sets against the oracle as whole witnesses, error status of every set."""
    from oracle import cbind
    g = pkg.Graph(wl.data)
    rows = make_inputs(wl, g, batch, 4)
    d_in = torch.from_numpy(rows).to(dev)
    d_out = torch.empty((batch, g.n_witness, 32), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(batch, dtype=torch.int32, device=dev)
    dt, tm, interp_ms, pack_ms = timed_batch(g, d_in, d_out, d_st)
    sample = [0, 1, batch // 3, batch // 2, batch - 2049, batch - 2048, batch - 2, batch - 1]
    og = cbind.Graph(wl.data)
    want, st = og.evaluate_batch(rows[sample])
    got = d_out[torch.tensor(sample, device=dev)].cpu().numpy()
    achieved = g.algorithmic_bytes_per_set * batch / (interp_ms * 1e-3) / 1e9
    return {"workload": "%s, %d input sets on one GPU (per-GPU share of BASELINE config 4)" % (wl.name, batch), "value": batch / dt,
            "unit": "witnesses/s", "ms_per_step": dt * 1e3, "interp_kernel_ms": interp_ms, "pack_kernel_ms": pack_ms,
            "tile_width": tm["tile_width"], "interpreter_waves_per_divider_wave": tm["divider"], "launches": tm["n_launches"],
            "roofline_frac": achieved / HBM_PEAK_GBS, "compute": compute_block(wl.stats_from(g)["hist"], g.n_witness, batch, interp_ms),
            **sub_traffic(pkg, wl.kind, batch, tm["tile_width"], g.algorithmic_bytes_per_set * batch, interp_ms),
            "sets_with_error_status": int((d_st != 0).sum().item()),
            "matches_oracle": bool(np.array_equal(got, want) and not st.any()), "sets_checked_against_oracle": len(sample)}


def config5_point(pkg, dev, kind="bigint", batch=32, steps=3, cpu_sample=4, batch_one_gpu=256):
    """BASELINE config 5 at its named size on one of its two class graphs -- "bigint": 64-bit limbs x 32, long division by ONE limb (the
    builder's first generator); "rsa": the class BASELINE names, zk-email RSA / long_div: 121-bit registers x 17, long_div by the
    17-register modulus, 310 chained modular multiplications -- ten million nodes each, 32 input sets on one GPU (256 over 8),
    library-chosen program.  `cpu_sample` sets against the oracle as whole witnesses -- the same sets are the CPU baseline's sample (one
    pinned core, then every core).  `all_256_sets_on_one_gpu`: the whole config-5 batch on THIS GPU (the 32-set shard occupies 32 of
    the chip's 1024 SIMDs, one wavefront per set: splitting 256 sets over 8 GPUs buys latency, not throughput)."""
    t0 = time.perf_counter()
    wl = Workload(kind)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    g = pkg.Graph(wl.data)
    t_load = time.perf_counter() - t0
    wl.stats_from(g)
    rows = make_inputs(wl, g, batch_one_gpu, 5)
    t0 = time.perf_counter()
    key = g.pick_tile_width(batch)
    g.set_tile_width(key)
    blob_len = len(g.export_blob(key))   # (compiles the chosen program on the host; the blob is what a multi-GPU job broadcasts)
    t_compile = time.perf_counter() - t0
    d_in = torch.from_numpy(rows).to(dev)
    d_out = torch.empty((batch_one_gpu, g.n_witness, 32), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(batch_one_gpu, dtype=torch.int32, device=dev)
    dt, tm, interp_ms, pack_ms = timed_batch(g, d_in[:batch], d_out[:batch], d_st[:batch], steps=steps, key=key)
    cpu = None
    if cpu_sample > 0 and CPU_SAMPLE > 0:
        cpu = cpu_baseline(wl.data, rows[:batch], d_out[:batch], min(cpu_sample, batch), parse_reps=1, all_cores_sets=batch)
        cpu["gpu_over_one_core"] = (batch / dt) / cpu["value"]
        if "all_cores" in cpu:
            cpu["gpu_over_all_cores"] = (batch / dt) / cpu["all_cores"]["value"]
    ps = g.program_stats()
    bad = int((d_st[:batch] != 0).sum().item())
    # the whole batch of config 5 on this one GPU
    one_gpu = None
    if batch_one_gpu > batch:
        try:
            key_all = g.pick_tile_width(batch_one_gpu)
            dt2, tm2, interp2, pack2 = timed_batch(g, d_in, d_out, d_st, steps=steps, key=key_all)
            one_gpu = {"value": batch_one_gpu / dt2, "unit": "witnesses/s", "sets": batch_one_gpu, "ms_per_step": dt2 * 1e3, "interp_kernel_ms": interp2, "pack_kernel_ms": pack2,
                       "tile_width": tm2["tile_width"], "bundles": tm2["n_bundles"], "sets_with_error_status": int((d_st != 0).sum().item()),
                       "step_time_over_32_set_step": dt2 / dt, "compute": compute_block(wl.stats["hist"], g.n_witness, batch_one_gpu, interp2),
                       "note": "all 256 sets of BASELINE config 5 on one GPU: %.2f x the 32-set step's time for 8 x the sets" % (dt2 / dt)}
            if CPU_SAMPLE > 0:  # (advisor, round 5) sets beyond the first 32 -- which only this run evaluates -- against the oracle as whole witnesses
                from oracle import cbind
                sample = [batch + 8, batch_one_gpu - 1]
                want, wst = cbind.Graph(wl.data).evaluate_batch(rows[sample])
                got = d_out[torch.tensor(sample, device=dev)].cpu().numpy()
                one_gpu["matches_oracle"] = bool(np.array_equal(got, want) and not wst.any())
                one_gpu["sets_checked_against_oracle"] = sample
        except Exception as e:  # (a sub-record: its failure is reported, not fatal)
            one_gpu = {"error": "%s: %s" % (type(e).__name__, e)}
    return {"workload": "%s, %d nodes, %d input sets on one GPU (BASELINE config 5: 256 sets over 8 GPUs)" % (wl.name, g.n_nodes, batch),
            "graph": wl.source, "value": batch / dt, "unit": "witnesses/s", "ms_per_step": dt * 1e3, "steps": steps, "interp_kernel_ms": interp_ms, "pack_kernel_ms": pack_ms,
            "n_nodes": g.n_nodes, "n_op": g.n_op, "n_witness": g.n_witness, "depth": g.depth, "op_histogram": wl.stats["hist"], "tile_width": tm["tile_width"],
            "bundles": tm["n_bundles"], "slots": tm["n_slots"], "generate_seconds": t_gen, "load_seconds": t_load, "compile_and_export_seconds": t_compile,
            "program_bytes": blob_len, "class_bundles": ps["class_bundles"], "scan_steps": ps["n_scan_steps"], "lanes_active_mean": ps["lanes_active_mean"],
            "field_ops_per_sec": batch / dt * g.n_op, "nodes_per_sec": batch / dt * g.n_nodes, "sets_with_error_status": bad,
            "compute": compute_block(wl.stats["hist"], g.n_witness, batch, interp_ms),
            "hbm_model_frac": g.algorithmic_bytes_per_set * batch / (interp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "roofline_frac": g.algorithmic_bytes_per_set * batch / (interp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            **sub_traffic(pkg, kind, batch, tm["tile_width"], g.algorithmic_bytes_per_set * batch, interp_ms),
            "all_256_sets_on_one_gpu": one_gpu,
            "cpu_baseline": cpu, "matches_oracle": (cpu or {}).get("matches_gpu"), "sets_checked_against_oracle": min(cpu_sample, batch) if cpu else 0}


def config4_job(pkg, cdist, wl, dev, rank, world, distributed, per_gpu=8192):
    """BASELINE config 4 as a job: ONE global batch of per_gpu x N counter-generated authV2-class sets, contiguous shards
    (shard_range), rank 0 asks the cost model for the program of a shard and broadcasts it, every rank evaluates its
    shard and returns the 64-bit checksums of its witness rows; rank 0 hashes the list in set order and compares it with
    the same job recomputed on rank 0 alone, shard by shard (the N = 1 digest).  No data-path collective."""
    total = per_gpu * world
    if distributed:
        if rank == 0 and wl.kind != "authv2":
            wl = Workload("authv2")
        g, _ = cdist.broadcast_graph_rccl(pkg, wl.data if rank == 0 else None, 0, src=0, device=dev, batch_per_rank=per_gpu)
    else:
        if wl.kind != "authv2":
            wl = Workload("authv2")
        g = pkg.Graph(wl.data)
        g.set_tile_width(g.pick_tile_width(per_gpu))
    from tools.synth import synth_inputs
    first_row = wl.first_row(g) if rank == 0 else None

    def shard(lo, hi, steps):
        rows = synth_inputs("field", g.n_inputs, hi - lo, SEED + 4, lo, first_row if lo == 0 else None)
        d_in = torch.from_numpy(rows).to(dev)
        d_out = torch.empty((hi - lo, g.n_witness, 32), dtype=torch.uint8, device=dev)
        d_st = torch.zeros(hi - lo, dtype=torch.int32, device=dev)
        dt = run_steps(g, d_in, d_out, d_st, steps, 1, distributed and steps > 0) if steps else None
        if not steps:
            g.calc_witness_batch_device(d_in, d_out, d_st)
            torch.cuda.synchronize()
        cs = cdist.set_checksums(d_out)
        return dt, cs, int((d_st != 0).sum().item()), g.last_timing()

    lo, hi = cdist.shard_range(total, rank, world)
    steps = 3
    dt, cs, bad, tm = shard(lo, hi, steps)
    if distributed:
        te = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dt = float(te.item())
        parts = [torch.empty(cdist.shard_range(total, r, world)[1] - cdist.shard_range(total, r, world)[0], dtype=torch.int64, device=dev)
                 for r in range(world)]
        dist.all_gather(parts, cs)   # (64 KiB per rank: bookkeeping of the check, not a data-path exchange)
        cs_all = torch.cat(parts)
        tb = torch.tensor([bad], dtype=torch.int64, device=dev)
        dist.all_reduce(tb)
        bad = int(tb.item())
    else:
        cs_all = cs
    if rank != 0:
        return None
    digest = hashlib.sha256(cs_all.cpu().numpy().tobytes()).hexdigest()
    rec = {"workload": "authV2-class graph, one global batch of %d sets = %d per GPU x %d GPU(s) (BASELINE config 4)" % (total, per_gpu, world),
           "value": total * steps / dt, "unit": "witnesses/s", "n_gpus": world, "ms_per_step": dt / steps * 1e3, "scaling": "weak",
           "tile_width": tm["tile_width"], "interpreter_waves_per_divider_wave": tm["divider"], "sets_with_error_status": bad,
           "digest_of_set_checksums": digest}
    if world > 1:  # the N = 1 reference of the same job, on this GPU alone
        ref = []
        for r in range(world):
            a, b = cdist.shard_range(total, r, world)
            ref.append(shard(a, b, 0)[1])
        rec["single_gpu_digest"] = hashlib.sha256(torch.cat(ref).cpu().numpy().tobytes()).hexdigest()
        rec["matches_single_gpu_digest"] = rec["single_gpu_digest"] == digest
    return rec


def find_pmc_summary(graph_kind, batch, tile_width):
    """(summary, path) of the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_summary*.json, tools/profile_summary.py) whose
    configuration is this graph / batch / tile width, or (None, None)"""
    import glob
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary*.json"))):
        try:
            j = json.load(open(f))
        except (OSError, ValueError):
            continue
        c = j.get("config", {})
        if c.get("graph") == graph_kind and c.get("batch_per_gpu") == batch and c.get("tile_width") == tile_width:
            best = (j, f)
    return best


def committed_traffic(graph_kind, batch, tile_width, kernel_hash=None):
    """HBM bytes per interpreter launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_summary.json, collected
    on this same command line by tools/collect_evidence.sh) -- a cross-reference, not measured in this run -- or None when
    no committed profile matches this configuration."""
    j, f = find_pmc_summary(graph_kind, batch, tile_width)
    if j is None:
        return None, "none: no committed PMC profile matches this graph / batch / tile width"
    best = j["kernels"]["interp"]["hbm_bytes_per_launch_corrected"]
    PMC_CHECK["interp"] = j["kernels"]["interp"]
    src = "committed rocprofv3 --pmc passes of this command line (%s), FETCH_SIZE x2 gfx950 correction; not re-measured in this run" % os.path.relpath(f, ROOT)
    profiled = j.get("kernel_source_hash")
    same = None if not profiled or not kernel_hash else profiled == kernel_hash
    src += "; profiled kernels = this library's" if same else "; the profile is of OTHER kernel sources than this library's" if same is False else "; kernel sources of the profile not recorded"
    PMC_CHECK["same_kernels"] = same
    return best, src


def sub_traffic(pkg, graph_kind, batch, tile_width, alg_bytes_per_launch, interp_ms):
    """the counter traffic of a sub-record's interpreter launch from its own committed PMC summary (tools/collect_evidence.sh runs the
    passes on `bench.py --config N`): keys to merge into the sub-record"""
    j, f = find_pmc_summary(graph_kind, batch, tile_width)
    if j is None:
        return {"traffic": None, "traffic_source": "none: no committed PMC profile matches this graph / batch / tile width"}
    k = j["kernels"]["interp"]
    t = k["hbm_bytes_per_launch_corrected"]
    profiled = j.get("kernel_source_hash")
    return {"traffic": t, "traffic_over_algorithmic": t / alg_bytes_per_launch if alg_bytes_per_launch else None,
            "hbm_measured_frac": t / (interp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic_source": os.path.relpath(f, ROOT),
            "traffic_kernel_hash_matches": (profiled == pkg.kernel_source_hash()) if profiled else None,
            "profiled_avg_launch_ms": k.get("avg_duration_ms"),
            "sq_wait_any_frac": (k["sq_wait_any_per_launch"] / k["sq_wave_cycles_per_launch"]) if k.get("sq_wave_cycles_per_launch") else None}


PEAK = {"modmul_per_s": None}  # chip-wide one-lane Montgomery products per second measured in this run (main): the denominator of every `compute` block
PMC_CHECK = {"same_kernels": None, "interp": None}  # --pmc-selfcheck: the committed PMC summary must be of the kernels this library was built from


def cpu_baseline(graph_data, rows, d_out, n, parse_reps=3, all_cores_sets=None):
    """The oracle's C restatement of the reference evaluate() (sequential, scalar 4x64 Montgomery, one core),
    timed on a bounded sample of the same input sets; also cross-checks those sets against the GPU witnesses."""
    from oracle import cbind
    og = cbind.Graph(graph_data)
    og.evaluate_batch(rows[:2])  # warm caches / page in
    aff = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    pinned = None
    if aff:  # the timed thread stays on ONE core of the allowed set (the last one: core 0 takes the interrupts)
        try:
            os.sched_setaffinity(0, {aff[-1]})
            pinned = aff[-1]
        except OSError:
            pinned = None
    try:
        t, want, st = og.time_batch(rows[:n])
    finally:
        if pinned is not None:
            os.sched_setaffinity(0, set(aff))
    got = d_out[:n].cpu().numpy() if d_out is not None else None
    ok = st == 0
    out = {"value": n / t, "unit": "witnesses/s", "cores": 1, "kind": "port",
           "affinity": ("one thread pinned to CPU %d of the process's %d allowed CPUs" % (pinned, len(aff))) if pinned is not None else "one thread, not pinned",
           "sample": "first %d input sets of rank 0's batch, graph parsed once outside the timed window (B1 of BASELINE.md)" % n,
           "seconds": t, "matches_gpu": bool(np.array_equal(got[ok], want[ok])) if got is not None else None}
    # reported beside it: the reference really re-parses the .bin per call (lib.rs:129) ...
    t0 = time.perf_counter()
    for _ in range(parse_reps):
        cbind.Graph(graph_data)
    t_parse = (time.perf_counter() - t0) / parse_reps
    out["including_parse_per_call"] = {"value": 1.0 / (t / n + t_parse), "unit": "witnesses/s", "parse_seconds": t_parse}
    # ... and a courtesy upper bound: the same loop on every host core (the reference is single-threaded)
    cores = len(aff) if aff else (os.cpu_count() or 1)
    if cores > 1:
        m = min(all_cores_sets or n, rows.shape[0])  # (graphs that take a core a third of a second per set: more sets than the one-core sample)
        tt, want_mt, st_mt = cbind.time_batch_threads(og, rows[:m], cores)
        out["all_cores"] = {"value": m / tt, "unit": "witnesses/s", "cores": cores, "threads_with_work": min(m, cores), "sets": m, "seconds": tt,
                            "matches_one_core": bool(np.array_equal(want_mt[:n], want) and np.array_equal(st_mt[:n], st))}
    return out


if __name__ == "__main__":
    main()
