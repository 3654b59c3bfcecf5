"""The C-ABI boundary from C: our conformance client (shape of the reference's examples/calc_witness.c) builds
against include/graph_witness.h; and, where /root/reference is present (this container only), the reference's
own example builds against the reference's own header and links against this library unchanged."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _build_client(pkg, tmp_path):
    exe = str(tmp_path / "capi_client")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "capi_client.c"), "-L", libdir, "-lcircom_witnesscalc_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_conformance_client_builds_and_reports_errors(pkg, tmp_path):
    exe = _build_client(pkg, tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr
    bad = tmp_path / "bad.json"
    bad.write_text('{"a": -5}')
    r = subprocess.run([exe, str(bad), os.path.join(GOLD, "circuit1.bin"), str(tmp_path / "o.wtns")], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to calculate witness" in r.stdout and "not a positive integer" in r.stdout


@pytest.mark.skipif(not os.path.exists("/root/reference/examples/calc_witness.c"), reason="reference tree not present")
def test_reference_example_links_unchanged(pkg, tmp_path):
    libdir = os.path.dirname(pkg.LIB_PATH)
    exe = str(tmp_path / "ref_example")
    subprocess.check_call(["gcc", "-w", "-o", exe, "/root/reference/examples/calc_witness.c", "-L", libdir,
                           "-lcircom_witnesscalc_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage" in r.stderr


@pytest.mark.gpu
def test_conformance_client_end_to_end(pkg, tmp_path):
    exe = _build_client(pkg, tmp_path)
    out = tmp_path / "c1.wtns"
    r = subprocess.run([exe, os.path.join(GOLD, "circuit1_inputs.json"), os.path.join(GOLD, "circuit1.bin"), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert out.read_bytes() == open(os.path.join(GOLD, "circuit1.wtns"), "rb").read()


def _build_bcast_client(pkg, tmp_path):
    exe = str(tmp_path / "bcast_client")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "bcast_client.c"), "-L", libdir, "-lcircom_witnesscalc_amd", "-ldl",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_native_broadcast_client_builds(pkg, tmp_path):
    """The RCCL-owning C host of tests/native/bcast_client.c compiles against include/graph_witness_batch.h (CPU: usage only)."""
    r = subprocess.run([_build_bcast_client(pkg, tmp_path)], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_native_broadcast_over_an_rccl_communicator(pkg, tmp_path):
    """gwb_graph_broadcast from a C host that owns the communicator (one rank: the GPU boxes have one GPU): the root
    gets its handle back and evaluates circuit1's reference inputs to the golden witness [1, 31817, 105, 303]."""
    exe = _build_bcast_client(pkg, tmp_path)
    r = subprocess.run([exe, os.path.join(GOLD, "circuit1.bin"), os.path.join(GOLD, "circuit1_inputs.json")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [ln for ln in r.stdout.split("\n") if len(ln) == 64 and all(c in "0123456789abcdef" for c in ln)]  # (RCCL prints a banner on stdout)
    assert [int(x, 16) for x in rows] == [1, 31817, 105, 303]
