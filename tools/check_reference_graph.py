"""One-command check of a REFERENCE-PRODUCED graph (what cannot be done offline in this repository: no cargo, no circom).

A maintainer with the reference toolchain runs, in the reference checkout:

    cargo run --release --bin build-circuit -- test_circuits/circuit9_authV2.circom authV2.bin -l test_deps/... 
    cargo run --release --bin calc-witness -- authV2.bin test_circuits/circuit9_authV2_inputs.json ref.wtns

and then here:

    python tools/check_reference_graph.py authV2.bin test_circuits/circuit9_authV2_inputs.json ref.wtns

Checks, in order: (1) the product's reader and the oracle's independent reader agree on the file (node count, witness list,
input map); (2) the product's writer re-serializes the loaded graph to the file's exact bytes (prost's encoding, reference
src/storage.rs:137-183); (3) the witness of the given inputs -- through gw_calc_witness on the GPU when there is one, else
through the compiled program on the CPU emulator of the tests -- equals the oracle's; (4) with a third argument, the
`.wtns` bytes equal the reference's own output byte for byte.  Exit code 0 only if every check that could run passed.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if len(sys.argv) not in (3, 4):
        sys.exit(__doc__)
    import cwc_import
    pkg = cwc_import.load()
    from oracle import model
    data = open(sys.argv[1], "rb").read()
    inputs_json = open(sys.argv[2]).read()
    ok = True
    nodes, wit, in_map = model.deserialize_witnesscalc_graph(data)
    g = pkg.Graph(data)
    same = g.n_nodes == len(nodes) and g.n_witness == len(wit) and g.n_inputs == model.get_inputs_size(nodes)
    print("1. readers agree: %s (%d nodes, %d witness elements, %d input slots, depth %d)" % (same, g.n_nodes, g.n_witness, g.n_inputs, g.depth))
    ok &= same
    again = g.serialize()
    print("2. writer reproduces the file's bytes: %s (%d bytes)" % (again == data, len(data)))
    ok &= again == data
    row = [1] + [0] * (model.get_inputs_size(nodes) - 1)
    for k, v in model.deserialize_inputs(inputs_json).items():
        off, n = in_map[k]
        assert len(v) <= n, "input %s has more values than the graph declares" % k
        row[off:off + len(v)] = v
    want = model.wtns_from_witness(model.evaluate(nodes, row, wit))
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except ImportError:
        have_gpu = False
    if have_gpu:
        got = pkg.calc_witness_wtns(inputs_json, data)
        how = "gw_calc_witness on the GPU"
    else:
        from tests import program_emulator as pe
        vals, st = pe.run(pe.Blob(g.export_blob(g.pick_tile_width(1))), row)
        got = model.wtns_from_witness(vals) if st == 0 else b""
        how = "the compiled program on the CPU emulator (no GPU here)"
    print("3. witness through %s equals the oracle's: %s" % (how, got == want))
    ok &= got == want
    if len(sys.argv) == 4:
        ref = open(sys.argv[3], "rb").read()
        print("4. `.wtns` equals the reference's own output: %s (%d bytes)" % (ref == got, len(ref)))
        ok &= ref == got
    else:
        print("4. (no reference .wtns given: skipped)")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
