"""ISA hazard scan of libvgx.so's gfx950 code objects: the DPP read hazards the compiler cannot see inside inline-asm chains.

gfx9 / CDNA need software wait states in two cases that involve a DPP instruction (LLVM GCNHazardRecognizer::checkDPPHazards,
MI300 ISA guide "manually inserted wait states"):
  * a VALU instruction writes a VGPR and a DPP instruction reads that VGPR as its permuted source within 2 wait states;
  * a VALU instruction writes EXEC (v_cmpx*) and a DPP instruction follows within 5 wait states.
hipcc inserts the s_nops for code it schedules itself, but an `asm volatile` statement is opaque to its hazard recogniser: whatever
stands right before the statement (a spill reload through v_accvgpr_read, a v_mov of a phi copy) can write the chain's source
operand.  The kernels' chains carry their own leading s_nop; this scan checks the SHIPPED code: every DPP instruction of every
kernel, walking backwards over all paths that reach it (fall-through and branch edges), counting an s_nop N as N + 1 wait states.

python tools/isa_hazard_scan.py [path/to/libvgx.so]   -> exit code 1 and a list of violations if there is any."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _find_objdump():
    """llvm-objdump of the ROCm installation that compiled the library: next to $HIPCC, under $ROCM_PATH, /opt/rocm, or on PATH."""
    import shutil
    cands = []
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cands.append(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"))
    for c in cands:
        if os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return shutil.which("llvm-objdump")


OBJDUMP = _find_objdump()
TOOLS_MISSING = 3      # exit code: objcopy / llvm-objdump not found (the scan did not run; not a hazard)
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
DPP_VGPR_WAIT, DPP_EXEC_WAIT = 2, 5


def code_objects(lib):
    """The gfx950 ELF images of every translation unit bundled into the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, p)
            ident = blob[p + 24:p + 24 + idlen].decode()
            p += 24 + idlen
            if "amdgcn" in ident and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


REG = re.compile(r"^(v|s|a)(\d+)$|^(v|s|a)\[(\d+):(\d+)\]$")


def regs(op):
    """('v', {indices}) for a VGPR / SGPR / AccVGPR operand, else None."""
    op = op.strip()
    for neg in ("-", "|", "neg(", "abs(", "sext("):
        op = op.replace(neg, "")
    op = op.rstrip(")")
    m = REG.match(op)
    if not m:
        return None
    if m.group(1):
        return m.group(1), {int(m.group(2))}
    return m.group(3), set(range(int(m.group(4)), int(m.group(5)) + 1))


DPP_MARK = ("quad_perm:", "row_shl:", "row_shr:", "row_ror:", "wave_shl", "wave_shr", "wave_rol", "wave_ror", "row_mirror", "row_half_mirror",
            "row_bcast:", "row_newbcast:", "row_share:", "row_xmask:")


class Ins:
    __slots__ = ("addr", "text", "mnem", "ops", "valu", "dpp_src", "vdef", "exec_def", "wait", "target", "uncond")

    def __init__(self, addr, text):
        self.addr, self.text = addr, text
        parts = text.split(None, 1)
        self.mnem = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        # operands end where the modifiers begin (first token without a comma before it that is not an operand)
        self.ops = [o.strip() for o in rest.split(",")]
        self.valu = self.mnem.startswith("v_")
        self.dpp_src = None
        if self.valu and (self.mnem.endswith("_dpp") or any(k in text for k in DPP_MARK)):
            src = self.ops[1].split()[0] if len(self.ops) > 1 else ""
            r = regs(src)
            self.dpp_src = r[1] if r and r[0] == "v" else set()
        self.vdef = set()
        self.exec_def = False
        if self.valu:
            d = regs(self.ops[0].split()[0]) if self.ops and self.ops[0] else None
            if d and d[0] == "v":
                self.vdef = d[1]
            if self.mnem.startswith("v_cmpx"):
                self.exec_def = True
            if self.mnem.startswith("v_swap") and len(self.ops) > 1:
                d2 = regs(self.ops[1].split()[0])
                if d2 and d2[0] == "v":
                    self.vdef |= d2[1]
        self.wait = 1
        if self.mnem == "s_nop":
            self.wait = int(self.ops[0], 0) + 1
        self.target = None
        self.uncond = self.mnem in ("s_branch", "s_endpgm", "s_setpc_b64")


def parse(disasm):
    """{kernel name: [Ins]} from llvm-objdump -d output."""
    funcs, cur = {}, None
    label = re.compile(r"^([0-9a-f]+) <([^>]+)>:")
    ins = re.compile(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):")
    for line in disasm.splitlines():
        m = label.match(line)
        if m:
            name = m.group(2)
            if not name.startswith("L") and not name.startswith(".L") and "$" not in name:
                cur = funcs.setdefault(name, [])
            continue
        m = ins.match(line)
        if m and cur is not None:
            cur.append(Ins(int(m.group(2), 16), m.group(1)))
    # branch targets: "s_cbranch_scc1 65" is a signed dword offset relative to the next instruction
    for body in funcs.values():
        addr_ix = {i.addr: k for k, i in enumerate(body)}
        for k, i in enumerate(body):
            if i.mnem.startswith("s_cbranch") or i.mnem == "s_branch":
                try:
                    off = int(i.ops[0].split()[0], 0)
                except ValueError:
                    continue
                if off >= 0x8000:
                    off -= 0x10000
                nxt = body[k + 1].addr if k + 1 < len(body) else i.addr + 4
                i.target = addr_ix.get(nxt + 4 * off)
    return funcs


def scan(body):
    preds = {}
    for k, i in enumerate(body):
        if i.target is not None:
            preds.setdefault(i.target, []).append(k)
    bad = []

    def walk(k, budget_v, budget_e, src, seen, origin):
        """instructions that can execute right before body[k], within the remaining wait-state budgets"""
        cands = []
        if k > 0 and not body[k - 1].uncond:
            cands.append(k - 1)
        cands += preds.get(k, [])
        for j in cands:
            if (j, budget_v, budget_e) in seen:
                continue
            seen.add((j, budget_v, budget_e))
            p = body[j]
            if p.valu and budget_v > 0 and (p.vdef & src):
                bad.append((origin, j, "VGPR written %d wait state(s) before its DPP read" % (DPP_VGPR_WAIT - budget_v)))
            if p.exec_def and budget_e > 0:
                bad.append((origin, j, "EXEC written by a VALU instruction %d wait state(s) before a DPP instruction" % (DPP_EXEC_WAIT - budget_e)))
            bv, be = budget_v - p.wait, budget_e - p.wait
            if bv > 0 or be > 0:
                walk(j, max(bv, 0), max(be, 0), src, seen, origin)

    n_dpp = 0
    for k, i in enumerate(body):
        if i.dpp_src is None:
            continue
        n_dpp += 1
        walk(k, DPP_VGPR_WAIT, DPP_EXEC_WAIT, i.dpp_src, set(), k)
    return n_dpp, bad


def main(lib):
    import shutil
    if not OBJDUMP or not shutil.which("objcopy"):
        print("isa_hazard_scan: SKIPPED — %s not found (looked next to $HIPCC, under $ROCM_PATH, /opt/rocm and on PATH); the DPP hazards of "
              "%s were NOT checked" % ("llvm-objdump" if not OBJDUMP else "objcopy (binutils)", os.path.relpath(lib, ROOT)), file=sys.stderr)
        return TOOLS_MISSING
    total, viol = 0, []
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        for name, body in parse(dis).items():
            n, bad = scan(body)
            total += n
            for origin, j, why in bad:
                viol.append("%s: %s\n    writer   %x: %s\n    dpp read %x: %s" % (name, why, body[j].addr, body[j].text, body[origin].addr, body[origin].text))
    print("isa_hazard_scan: %d DPP instructions checked in %s, %d violation(s)" % (total, os.path.relpath(lib, ROOT), len(viol)))
    for v in viol:
        print(v)
    return 1 if viol else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "vgsim_amd", "libvgx.so")))
