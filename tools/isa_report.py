"""What the compiler made of the interpreter instances: per instance the instruction count, branches, DIVERGENT branches
(s_and_saveexec / s_or_saveexec), flag branches (s_and(n2)_b64 vcc, exec, <flag> in front of an s_cbranch_vcc*: a uniform branch
rewritten by StructurizeCFG, or an i1 phi), SGPR spill lanes (v_writelane) and the back edges of the bundle loop.

Why it matters (profiles/r05_structurizer_ab.txt): the kernels are built with -structurizecfg-skip-uniform-regions, and a
region that holds a divergent branch anywhere below it is left alone only while it has at most one conditional branch of its own.
One `a && b` with an expensive right-hand side, one `c ? f(x) : y` around an asm volatile, one `if (per_lane) ...` inside a class
body, and the uniform tests around it turn into flag registers and chains of s_cbranch_vcc*, the class paths lose their own back
edges to the loop header, and flags spill to VGPR lanes: the same program ran 10-15 % slower for code it never executed.

    python tools/isa_report.py              # the kernels as built (csrc/build/kernels.o), every interpreter instance
    python tools/isa_report.py --lines 1,0,0,1,3    # + source lines of the divergent branches of instance <T,PROF,W,PACK,MODE> (compiles with line tables)
"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(ROOT, "circom-witnesscalc_amd", "csrc")
INST = re.compile(r"interp_kernelILi(\d+)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)E")


def disassemble(obj=None, tmp=None):
    """lines of the gfx950 code object inside build/kernels.o"""
    obj = obj or os.path.join(CSRC, "build", "kernels.o")
    tmp = tmp or tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "fatbin.bin"), os.path.join(tmp, "kernels_gfx950.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout.split("\n")


def instance_stats(asm):
    """{(T, PROF, W, PACK, MODE): Counter} over the disassembly's interpreter instances"""
    stats, cur, start = collections.OrderedDict(), None, 0
    for ln in asm:
        m = re.match(r"^([0-9a-f]+) <(\w+)>:", ln)
        if m:
            k = INST.search(m.group(2))
            cur = tuple(int(x) for x in k.groups()) if k else None
            if cur is not None:
                stats[cur] = collections.Counter()
                start = int(m.group(1), 16)
            continue
        if cur is None or not ln.startswith("\t"):
            continue
        c, s = stats[cur], ln.split("//")[0]
        c["instructions"] += 1
        a = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
        if a:
            c["bytes"] = int(a.group(1), 16) - start
        if "saveexec" in s:
            c["divergent_branches"] += 1
        if re.search(r"s_andn?2?_b64 vcc, exec, s\[", s):
            c["flag_branches"] += 1
        if "v_writelane_b32" in s:
            c["sgpr_spill_writes"] += 1
        if re.search(r"\bs_c?branch", s):
            c["branches"] += 1
        if "s_setpc_b64" in s:
            c["long_branches"] += 1
    return stats


def divergent_sites(instance):
    """source lines of the divergent branches of one instance: a compile of kernels.hip with line tables (about two minutes)"""
    kflags = re.search(r"^KFLAGS \?= (.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M).group(1).split()  # (the build's code generation flags)
    out = os.path.join(tempfile.mkdtemp(), "kernels_g.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950"] + kflags + ['-DCWC_KSRC_HASH="x"', "--cuda-device-only", "-gline-tables-only", "-S",
                           os.path.join(CSRC, "kernels.hip"), "-o", out], stderr=subprocess.DEVNULL)
    want = "interp_kernelILi%dELb%dELi%dELi%dELi%dE" % instance
    files, cur, loc, sites = {}, None, None, collections.Counter()
    for ln in open(out):
        m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        m = re.match(r"^(_ZN3cwc\w+):", ln)
        if m:
            cur = m.group(1)
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
        if cur is None or want not in cur:
            continue
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            loc = "%s:%s" % (files.get(int(m.group(1)), m.group(1)), m.group(2))
        elif "saveexec" in ln.split(";")[0]:
            sites[loc] += 1
    return sites


if __name__ == "__main__":
    st = instance_stats(disassemble())
    print("%-22s %8s %8s %9s %10s %6s %7s %6s" % ("<T,PROF,W,PACK,MODE>", "bytes", "instr", "branches", "divergent", "flag", "spills", "long"))
    for k, c in st.items():
        print("%-22s %8d %8d %9d %10d %6d %7d %6d" % ("<%d,%d,%d,%d,%d>" % k, c["bytes"], c["instructions"], c["branches"], c["divergent_branches"], c["flag_branches"], c["sgpr_spill_writes"], c["long_branches"]))
    if "--lines" in sys.argv:
        inst = tuple(int(x) for x in sys.argv[sys.argv.index("--lines") + 1].split(","))
        for loc, n in sorted(divergent_sites(inst).items(), key=lambda kv: str(kv[0])):
            print("  divergent branch at %-28s x%d" % (loc, n))
