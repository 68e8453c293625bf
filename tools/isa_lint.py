"""ISA lint of the gfx950 code objects the build produced:  python3 tools/isa_lint.py fvgp_amd/csrc/*.o

Run by the Makefile after every link (a build whose code objects break a rule FAILS) and by the CPU suite
(tests/test_host_logic.py).  It reads what hipcc really emitted -- the device code object inside each .o is unbundled and
disassembled with the toolchain's own llvm-objdump -- so the rules hold for whatever compiler built the shipped library.

Rules (the LAPACK dpotrf these kernels stand in for: fvgp/gp_lin_alg.py:245):
  R1  no 12/16-byte buffer store carries an SGPR offset.  The ">64-bit store data" hazard (a VALU write of the store's data registers
      right behind the store) is documented as absent for such stores, the compiler therefore inserts no wait state -- and on gfx950
      the store was seen going out with the NEW register contents (profiles/r05_store_hazard_chain_verify.txt).
  R2  the instruction right behind ANY 12/16-byte vector store (buffer / global / flat / scratch) does not VALU-write one of the
      store's data registers: the wait state the hazard needs is there (s_nop or an unrelated instruction), whoever scheduled it.
  R3  in the resident panel kernel (chain_kernel) and the one-launch vector sweeps EVERY vector memory load carries sc1: bytes
      another workgroup of the same launch has produced are never served from the compute unit's vector L1 or a stale L2 line
      (hand-off form).  The kernels read nothing but such bytes and their own inputs, so the rule is "all".
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
ALL_SC1_KERNELS = ("chain_kernel",)          # substrings of the (mangled) kernel names R3 applies to

WIDE_STORE = re.compile(r"^(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$")
VECTOR_LOAD = re.compile(r"^(buffer_load_\w+|global_load_\w+|flat_load_\w+)\s")


def disassemble(obj, workdir):
    """text of llvm-objdump -d of the gfx950 code object embedded in a host object / shared library (None: no device code)"""
    base = os.path.join(workdir, os.path.basename(obj))
    fat = base + ".fatbin"
    res = subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    co = base + ".co"
    res = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", f"--targets={TARGET}", f"--input={fat}", f"--output={co}", "--unbundle"],
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"{obj}: cannot unbundle the gfx950 code object: {res.stderr.strip()}")
    res = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"{obj}: llvm-objdump failed: {res.stderr.strip()}")
    return res.stdout


def _regs(tok):
    """set of VGPR numbers named by an operand token like v12 or v[4:7]; empty for anything else"""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def lint_text(text, name):
    """violations [(rule, kernel, instruction)] in one disassembly"""
    bad = []
    kernel = None
    pending = None                       # (data registers, store text) of the wide store on the previous line
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            kernel, pending = m.group(1), None
            continue
        ins = line.split("//")[0].strip()
        if not ins or kernel is None:
            continue
        if pending is not None:
            data, store = pending
            pending = None
            mn = ins.split()[0]
            if mn.startswith("v_") and not mn.startswith("v_cmp") and len(ins.split()) > 1:
                dst = _regs(ins.split(None, 1)[1].split(",")[0])
                if dst & data:
                    bad.append(("R2", kernel, f"{store}  ->  {ins}"))
        m = WIDE_STORE.match(ins)
        if m:
            ops = [o.strip() for o in m.group(2).split(",")]
            if m.group(1).startswith("buffer_"):
                # buffer_store_dwordx4 vdata, vaddr, srsrc, soffset [offen ...]
                soff = ops[3].split()[0] if len(ops) > 3 else "0"
                if re.fullmatch(r"s\d+", soff):
                    bad.append(("R1", kernel, ins))
                data = _regs(ops[0])
            else:
                # global_store_dwordx4 vaddr, vdata, saddr / flat_store_dwordx4 vaddr, vdata / scratch_store_dwordx4 vaddr, vdata, ...
                data = _regs(ops[1]) if len(ops) > 1 else set()
            pending = (data, ins)
        if any(k in kernel for k in ALL_SC1_KERNELS) and VECTOR_LOAD.match(ins) and not re.search(r"\bsc1\b", ins):
            bad.append(("R3", kernel, ins))
    return [(r, name + ": " + k, i) for r, k, i in bad]


def lint_objects(paths):
    bad, seen = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for p in paths:
            text = disassemble(p, tmp)
            if text is None:
                continue
            seen += 1
            bad += lint_text(text, os.path.basename(p))
    return bad, seen


def main(argv):
    paths = [p for p in argv if os.path.exists(p)]
    if not paths:
        print("isa_lint: no objects given", file=sys.stderr)
        return 2
    bad, seen = lint_objects(paths)
    for rule, kernel, ins in bad[:40]:
        print(f"isa_lint {rule}: {kernel}\n    {ins}", file=sys.stderr)
    if bad:
        print(f"isa_lint: {len(bad)} violation(s) in {seen} code object(s) -- see tools/isa_lint.py for the rules", file=sys.stderr)
        return 1
    print(f"isa_lint: {seen} gfx950 code object(s) clean (R1 wide stores without SGPR offset, R2 wait state behind wide stores, R3 sc1 loads in the resident kernels)")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
