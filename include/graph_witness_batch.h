/* Additive batch C-ABI: many independent input sets for one graph, evaluated on an MI355X.
 *
 * The reference has no batch entry point: calc_witness (src/lib.rs:125-136) re-parses the graph and
 * evaluates one input set per call.  These functions split that call into its stages so that the graph
 * is parsed/compiled once (replaces storage.rs:214-249 per call), inputs and witnesses stay resident in
 * HBM (replaces graph::evaluate, src/graph.rs:367-391, per set) and `.wtns` framing (src/lib.rs:114-123)
 * is applied per set on demand.  The single-shot symbol gw_calc_witness is unchanged.
 *
 * Plain pointers and sizes only.  All functions return 0 on success, 1 on failure with status filled as
 * in graph_witness.h (status may be NULL).
 */
#ifndef CWC_AMD_GRAPH_WITNESS_BATCH_H
#define CWC_AMD_GRAPH_WITNESS_BATCH_H

#include <stddef.h>
#include <stdint.h>

#include "graph_witness.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gwb_graph gwb_graph_t;

typedef struct {
  uint64_t n_nodes;          /* nodes in the .bin */
  uint64_t n_op;             /* Op + UnoOp + TresOp nodes (field-ops per input set) */
  uint64_t n_input_nodes;    /* Input nodes */
  uint64_t n_const;          /* constant nodes */
  uint64_t n_inputs;         /* length of the inputs buffer (slot 0 = 1), reference src/lib.rs:138-152 */
  uint64_t n_witness;        /* witness elements per set */
  uint64_t depth;            /* dependency levels */
  uint64_t algorithmic_bytes_per_set; /* 32*[sum_ops(arity+1) + 2*n_input_nodes + 2*n_witness] */
} gwb_graph_info_t;

typedef struct {
  uint32_t tile_width;       /* input sets per wavefront used by the last call */
  uint32_t n_launches;       /* interpreter launches of the last call (chunks) */
  uint64_t n_bundles;        /* wave-instruction bundles per tile */
  uint64_t n_slots;          /* value slots per set */
  float interp_ms;           /* HIP-event time of the interpreter kernel(s), on the launch stream */
  float pack_ms;             /* HIP-event time of the witness pack kernel(s) */
  uint32_t divider;          /* interpreter waves per divider wave in the last call's programs: 0 (none), 1 or 4 */
  uint32_t streams;          /* wavefronts per tile in the last call's program (1, or 2 / 4: GWB_TILE_STREAMS*) */
} gwb_timing_t;

/* Parse + validate a `wtns.graph.001` image (deserialize_witnesscalc_graph, src/storage.rs:214-249). */
int gwb_graph_load(const void *graph_data, size_t graph_data_len, gwb_graph_t **out, gw_status_t *status);
void gwb_graph_free(gwb_graph_t *g);
int gwb_graph_info(const gwb_graph_t *g, gwb_graph_info_t *info);
/* Re-serialize the loaded graph (serialize_witnesscalc_graph, src/storage.rs:137-183): *out is malloc'ed. */
int gwb_graph_serialize(const gwb_graph_t *g, void **out, size_t *out_len, gw_status_t *status);

/* Graph producer: what the reference's build-circuit does with a Vec<graph::Node> (src/graph.rs:236-245),
 * the witness list and the InputSignalsInfo map before serialize_witnesscalc_graph (src/storage.rs:137-183, node encoding
 * :50-91).  Nodes are pushed in topological order -- every operand index is an earlier node (src/graph.rs:343-356) -- and
 * each push returns the node's index in the file, or GWB_BUILDER_BAD after an error (errors are sticky and reported by
 * gwb_builder_finish).  Operation codes are the wire codes of protos/messages.proto:5-35 (DuoOp 0..19: Mul, Div, Add, Sub,
 * Pow, Idiv, Mod, Eq, Neq, Lt, Gt, Leq, Geq, Land, Lor, Shl, Shr, Bor, Band, Bxor; UnoOp 0 Neg, 1 Id; TresOp 0 TernCond). */
typedef struct gwb_builder gwb_builder_t;
#define GWB_BUILDER_BAD 0xffffffffu
gwb_builder_t *gwb_builder_new(void);
void gwb_builder_free(gwb_builder_t *b);
uint32_t gwb_builder_input(gwb_builder_t *b, uint32_t input_index);                 /* Node::Input */
/* Node::MontConstant: value_le = little-endian bytes of the canonical value (any length; reduced mod r as the reader does) */
uint32_t gwb_builder_constant(gwb_builder_t *b, const void *value_le, size_t value_len);
uint32_t gwb_builder_uno(gwb_builder_t *b, uint32_t op, uint32_t a);                /* Node::UnoOp */
uint32_t gwb_builder_duo(gwb_builder_t *b, uint32_t op, uint32_t a, uint32_t b_idx); /* Node::Op */
uint32_t gwb_builder_tres(gwb_builder_t *b, uint32_t op, uint32_t a, uint32_t b_idx, uint32_t c); /* Node::TresOp */
int gwb_builder_witness(gwb_builder_t *b, uint32_t node);                           /* append to witness_signals */
int gwb_builder_input_signal(gwb_builder_t *b, const char *name, uint32_t offset, uint32_t len); /* InputSignalsInfo entry */
uint64_t gwb_builder_node_count(const gwb_builder_t *b);
/* serialize_witnesscalc_graph: *out is malloc'ed (`wtns.graph.001` container, loads in gw_calc_witness / gwb_graph_load
 * and in the reference's deserialize_witnesscalc_graph). */
int gwb_builder_finish(const gwb_builder_t *b, void **out, size_t *out_len, gw_status_t *status);

/* The two class graphs of BASELINE config 5 as `.bin` bytes (malloc'ed: the caller frees), generated natively.  The reference
 * front-end cannot compile such circuits (README.md:21), so these graphs are synthetic; the specification of both is the Python
 * generator library of the package (graphgen/circuits.py build_bigint_class / build_rsa_long_div_class), whose bytes these equal.
 *   bigint: `rounds` rounds of a k x k schoolbook product with n_bits-bit limbs, long division by ONE limb, limb-wise min.
 *   rsa:    `muls` chained modular multiplications on k registers of n bits (RSA-2048: n = 121, k = 17) as circom-bigint's
 *           witness hints compute them -- schoolbook product with carries, long_div by the k-register modulus (short_div estimate,
 *           long_scalar_mult, long_gt, long_sub per quotient digit) -- with the Num2Bits(n) range-check bits of q and r when
 *           range_checks is not 0. */
int gwb_graphgen_bigint_class(uint32_t k, uint32_t n_bits, uint32_t rounds, void **out, size_t *out_len, gw_status_t *status);
int gwb_graphgen_rsa_long_div_class(uint32_t n, uint32_t k, uint32_t muls, int range_checks, void **out, size_t *out_len, gw_status_t *status);
/* Operation histogram of a loaded graph: out[0..19] DuoOp wire codes (Mul .. Bxor), out[20] Neg, out[21] TernCond, out[22] Input nodes,
 * out[23] constants (n >= 24 entries). */
int gwb_graph_op_histogram(const gwb_graph_t *g, uint64_t *out, size_t n);

/* JSON -> one inputs row of n_inputs x 32 bytes, canonical little-endian, row[0] = 1
 * (deserialize_inputs + get_inputs_buffer + populate_inputs, src/lib.rs:195-247, 177-181, 154-168). */
int gwb_inputs_from_json(const gwb_graph_t *g, const char *inputs_json, void *row, gw_status_t *status);

/* Batched front-end: `text` is a JSON array of input objects or NDJSON (one object per line); fills up to max_rows
 * rows of n_inputs x 32 bytes, *n_rows = number of input sets found (also set when the buffer is too small).  The input
 * sets are parsed on CWC_PARSE_THREADS host threads (default: every core). */
int gwb_inputs_from_json_batch(const gwb_graph_t *g, const char *text, size_t text_len, void *rows, size_t max_rows,
                               size_t *n_rows, gw_status_t *status);
/* Write one `.wtns` file per input set (76-byte header + row, src/lib.rs:114-123); path_pattern takes the set index
 * through one %lu conversion, e.g. "out/witness_%05lu.wtns". */
int gwb_wtns_save_batch(const void *witness, size_t n_witness, size_t batch, const char *path_pattern, gw_status_t *status);

/* End to end and streaming: `text` (a JSON array of input objects or NDJSON) -> one `.wtns` file per input
 * set, path_pattern with one %lu conversion for first_index + the set's position.  Sub-batches move through a pipeline: the
 * next one is parsed (host threads) and evaluated while the witness rows of the previous one leave HBM in slices through
 * pinned staging and writer threads frame them as files (lib.rs:114-123).  *n_sets = input sets found; set_status (NULL or
 * max_sets words) takes the per-set status words.  A set whose status word is not zero -- the cases in which the
 * reference's evaluate() panics (src/graph.rs:634, :701, :716) -- gets NO file (one of that name from an earlier run is
 * removed): with a status buffer the call returns 0 and the words say which sets failed (stats->failed_sets counts them);
 * with set_status == NULL a failed set makes the call return 1 with a message naming the first one.  CWC_E2E_SUBBATCH
 * (default 1024), CWC_PARSE_THREADS (default all cores), CWC_WRITE_THREADS (default min(cores, 16)), CWC_E2E_SLICE_MB
 * (default 96) tune it. */
typedef struct {
  size_t n_sets, sub_batch;
  uint32_t parse_threads, write_threads;
  double parse_seconds;            /* summed over the sub-batches (overlapped with kernels and copies of the previous one) */
  double wait_for_drain_seconds;   /* the calling thread waiting for copies / file writes before it could reuse buffers */
  double total_seconds;
  uint64_t witness_bytes;
  uint64_t failed_sets;            /* input sets with a non-zero status word: no file written */
} gwb_e2e_stats_t;
int gwb_calc_witness_json_to_wtns(gwb_graph_t *g, const char *text, size_t text_len, const char *path_pattern, size_t first_index,
                                  size_t *n_sets, uint32_t *set_status, size_t max_sets, gwb_e2e_stats_t *stats, gw_status_t *status);

/* Wherever this API takes or returns a tile width it is a "program key": the width (input sets per wavefront, a power
 * of two in 1..64), optionally OR'ed with GWB_TILE_ASYNC_DIVIDER = programs for the asynchronous divider wave (one
 * extra wavefront per tile serves the field divisions while the interpreter wave goes on; widths below 64). */
#define GWB_TILE_ASYNC_DIVIDER 0x100u
/* ... or with GWB_TILE_GROUP_DIVIDER = one divider wave per FOUR interpreter waves, which packs their division requests
 * into the lanes of one inversion (throughput regime, widths up to 32). */
#define GWB_TILE_GROUP_DIVIDER 0x200u
/* ... or with GWB_TILE_TRIPLE_DIVIDER = one divider wave per THREE interpreter waves: a four-wave workgroup, one wave on
 * every SIMD of its CU, so 768 tiles run at the speed of one (the choice for 513..768 tiles). */
#define GWB_TILE_TRIPLE_DIVIDER 0x400u
/* ... and / or with GWB_TILE_STREAMS2 / GWB_TILE_STREAMS4 = two / four wavefronts per tile, each evaluating its share of
 * the graph's independent parts (components that share nothing beyond a short prologue behind the inputs) with a bundle
 * sequence of its own: the small-batch / single-call regime, where there are more SIMDs than tiles.  Combines with no
 * divider or GWB_TILE_ASYNC_DIVIDER (every stream then has a divider wave of its own).  A graph that is one piece
 * compiles to the single-stream program. */
#define GWB_TILE_STREAMS2 0x800u
#define GWB_TILE_STREAMS4 0x1000u
/* 0 = choose from the batch size (default); else a program key */
int gwb_set_tile_width(gwb_graph_t *g, uint32_t tile_width);
/* the program key the static rule names for a batch of this size (no graph needed) */
uint32_t gwb_pick_tile_width(size_t batch);
/* the program key the cost model chooses for THIS graph and batch size (what a call with tile width 0 will use); 0 on
 * failure.  Host-only work: rank 0 of a multi-GPU job asks once, exports that program and broadcasts it. */
uint32_t gwb_graph_pick_tile_width(gwb_graph_t *g, size_t batch);

/* Evaluate `batch` input sets resident in device memory (graph::evaluate per set, src/graph.rs:367-391).
 *   d_inputs  : [batch][n_inputs][32 B] canonical LE
 *   d_witness : [batch][n_witness][32 B] canonical LE (= the .wtns section-2 body of each set)
 *   d_set_status : [batch] u32, 0 = ok, bit0 = Shl overflow, bit1 = bit-op result == r (reference panics)
 *   hip_stream: hipStream_t or NULL.  Asynchronous: returns after enqueueing. */
int gwb_calc_witness_batch_device(gwb_graph_t *g, const void *d_inputs, size_t batch, void *d_witness,
                                  uint32_t *d_set_status, void *hip_stream, gw_status_t *status);
/* Ordering: the calls on one handle share its value workspace, so they execute in the order they were enqueued, whatever
 * streams they name -- a call on a different stream than the previous one first waits (hipStreamWaitEvent) for that
 * call's last kernel.  Calls on DIFFERENT handles are independent.  Reading d_witness / d_set_status from another stream
 * or from the host needs the usual synchronization with `hip_stream` (or the hand-off event below).
 *
 * Device hand-off to a prover (the consumer of `.wtns` is an MSM/NTT pipeline, reference test_circuits.sh:98): the witness
 * rows stay in HBM as [batch][n_witness][32 B] little-endian, one contiguous row of n_witness scalars per input set (what a
 * multi-scalar multiplication takes).  gwb_handoff_t adds the two things a GPU consumer needs: the scalars in Montgomery
 * form (x * 2^256 mod r, 8 x u32 / 4 x u64 little-endian limbs -- ark-ff's and most GPU provers' internal form) instead
 * of canonical integers, and an event recorded behind the last kernel of the call, for hipStreamWaitEvent on the
 * consumer's stream without a host round trip. */
#define GWB_FORM_CANONICAL 0u
#define GWB_FORM_MONTGOMERY 1u
typedef struct {
  uint32_t struct_size;      /* sizeof(gwb_handoff_t): lets the struct grow */
  uint32_t form;             /* GWB_FORM_CANONICAL (the .wtns body) or GWB_FORM_MONTGOMERY */
  void *hip_stream;          /* hipStream_t the work is enqueued on, or NULL */
  void *done_event;          /* hipEvent_t created by the caller, recorded after the last kernel; or NULL */
} gwb_handoff_t;
int gwb_calc_witness_batch_handoff(gwb_graph_t *g, const void *d_inputs, size_t batch, void *d_witness,
                                   uint32_t *d_set_status, const gwb_handoff_t *handoff, gw_status_t *status);

/* Same with host buffers (copies in/out, synchronous).  The witness rows come back in slices through pinned staging
 * while worker threads (CWC_COPY_THREADS, default min(cores, 16)) fill `witness`; a `witness` buffer from
 * gwb_host_alloc (or any pinned allocation) is the destination of the device copy itself. */
int gwb_calc_witness_batch_host(gwb_graph_t *g, const void *inputs, size_t batch, void *witness,
                                uint32_t *set_status, gw_status_t *status);
/* Pinned host memory for the rows of the host entry point (NULL on failure); release with gwb_host_free. */
void *gwb_host_alloc(size_t bytes);
void gwb_host_free(void *p);
/* Kernel times of the last batch call on this handle (synchronizes on its events). */
int gwb_last_timing(gwb_graph_t *g, gwb_timing_t *t);
/* Kernel times (ms) of the most recent launches on this handle, oldest first, at most max_launches (the handle keeps
 * the events of its last 256 launches); lets a caller time a run of asynchronous calls without synchronizing in it. */
int gwb_timing_history(gwb_graph_t *g, size_t max_launches, float *interp_ms, float *pack_ms, size_t *n_out);

/* Diagnostic build of the interpreter with in-kernel cycle stamps, shader cycles summed over the sampled waves:
 * out64[class*4 + 0] = cycles of the class's bundles, out64[class*4 + 3] = bundles; for MUL (k = 0) and LIN (k = 1)
 * bundles out64[48 + 8*k + {0: loop top + wait for staged operands, 1: LDS operand reads with the previous bundle's
 * stores issued behind them, 2: issuing the staging loads, 3: dispatch + arithmetic, 4: ring write, 5: bundles}];
 * over all interpreter waves: out64[54] = longest run time of the loop, out64[55] = 2^40 - shortest, out64[62] = sum,
 * out64[63] = waves; out64[64] / out64[67] = cycles / bundles of the fused narrow bundles (class 13), out64[68] / out64[71] of the macro
 * bundles (class 14); out64[72..79] = sections of the macro bundles: cycles up to the records, up to the operands, issuing the staging
 * loads, in the stages, number of stages, cycles re-reading operands of earlier stages, number of such stages, cycles behind the
 * last stage.  out64 must hold 96 words. */
int gwb_profile_classes(gwb_graph_t *g, const void *d_inputs, size_t batch, void *d_witness,
                        uint32_t *d_set_status, uint64_t *out64, gw_status_t *status);

/* Diagnostic: chip-wide one-lane Montgomery products per second with waves_per_simd (1..4) wavefronts on every SIMD, each
 * lane running a dependent chain of 2 * iters products (the compute ceiling bench.py reports beside the HBM model);
 * 0.0 on failure. */
double gwb_ubench_modmul(uint32_t waves_per_simd, uint32_t iters);

/* ... and with the multiplier of the interpreter's full-width bundles (one asm block, 322 issue slots, pinned accumulators:
 * waves_per_simd 1 or 2): the denominator of bench.py's `roofline.compute`. */
double gwb_ubench_modmul_block(uint32_t waves_per_simd, uint32_t iters);
/* The cost model's lone-wave cycles per bundle of a class (program_dev.h BundleClass), as loaded: built-in, or overridden by
 * CWC_MODEL_CYCLES / the calibration file (tools/gpu_calibrate.py --write; csrc/compile.cc CycleTable).  0 for an unknown class. */
double gwb_model_class_cycles(uint32_t bundle_class);
/* SHA-256 (hex) of the kernel sources the loaded library's device code was built from, as stamped by csrc/Makefile
 * ("unstamped" for a build outside it): lets a deployment or a test tell a stale kernel object from the tree's. */
const char* gwb_kernel_source_hash(void);

/* Statistics of a compiled program (program_key 0: the program the last batch call on this handle used). */
typedef struct {
  uint32_t tile_width, divider, streams, n_classes;
  uint64_t n_bundles, n_fused_nodes;
  uint64_t class_bundles[16];   /* per bundle class (program_dev.h BundleClass) */
  uint64_t class_nodes[16];
  double model_wave_cycles;     /* the cost model's lone-wave cycles for one tile */
  double lanes_active_mean;     /* of a wave's 64 lanes: mean number holding a node's work (the four lanes of a shared product all count), weighted by the bundles' modelled time */
  double values_per_bundle_mean; /* field elements (node x input set) a bundle produces, same weighting: 64 would be one per lane */
  double chain_floor_cycles;    /* the compiled graph's longest dependent chain priced at the best measured latency of each operation on a
                                 * lone wavefront, arithmetic only (no bundle front end): the floor of this execution model for one tile */
  uint64_t n_scan_steps;        /* steps of serial limb recurrences that run inside scan bundles */
  uint64_t n_conv_products;     /* limb products computed inside convolution bundles (the columns of a k x k schoolbook product in one bundle) */
} gwb_program_stats_t;
int gwb_program_stats(gwb_graph_t *g, uint32_t program_key, gwb_program_stats_t *out);

/* `.wtns` framing of one witness row (wtns_from_witness, src/lib.rs:114-123): out holds gwb_wtns_size bytes. */
size_t gwb_wtns_size(size_t n_witness);
int gwb_wtns_from_witness(const void *witness_row, size_t n_witness, void *out);

/* Compiled-program exchange between the GPUs of a node: rank 0 exports a pointer-free blob, the caller
 * moves it (RCCL broadcast over xGMI), the other ranks import it instead of re-parsing the .bin. */
int gwb_graph_export(gwb_graph_t *g, uint32_t tile_width, void **blob, size_t *blob_len, gw_status_t *status);
int gwb_graph_import(const void *blob, size_t blob_len, gwb_graph_t **out, gw_status_t *status);

/* The same exchange done by the library: rank `root` exports the program for `tile_width` (0 = what the cost model
 * chooses for a shard of batch_per_rank input sets) and broadcasts it over the caller's RCCL communicator (ncclComm_t,
 * xGMI inside a node) on hip_stream; every other rank imports it.  *out = g on the root, a new replica elsewhere.  This
 * is the only collective of the path -- input sets are independent, shards need no exchange.  RCCL is resolved in the
 * running process (dlsym), the library does not link it. */
/* A RCCL communicator for gwb_graph_broadcast made through the library (for hosts without a RCCL binding of their own):
 * one rank draws the 128-byte id, the host carries the bytes to the other ranks by whatever channel its job has (a file,
 * a socket, MPI, torch.distributed), every rank joins with the device it evaluates on current.  The handle is an
 * ncclComm_t; gwb_rccl_comm_ranks = ncclCommCount (-1 on error). */
#define GWB_RCCL_UNIQUE_ID_BYTES 128
int gwb_rccl_unique_id(void *id128, gw_status_t *status);
int gwb_rccl_comm_init(const void *id128, int n_ranks, int rank, void **comm, gw_status_t *status);
int gwb_rccl_comm_ranks(void *comm);
void gwb_rccl_comm_destroy(void *comm);
int gwb_graph_broadcast(gwb_graph_t *g, uint32_t tile_width, size_t batch_per_rank, int root, int rank, void *nccl_comm,
                        void *hip_stream, gwb_graph_t **out, gw_status_t *status);

/* exported twin of the header-inline gw_free_status, for FFI callers that cannot use the inline */
void gwb_free_status(gw_status_t *status);

#ifdef __cplusplus
}
#endif
#endif /* CWC_AMD_GRAPH_WITNESS_BATCH_H */
