"""Static scan of hipcc's gfx950 assembly for the software-visible wait-state rules that LLVM's hazard recognizer enforces for
compiler-generated code but NOT for the text of an inline-asm block (r06).

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only x.hip -o x.s
    python -m tomosar2height_amd.csrc.isa_pass scan x.s [...]                 # report
    python -m tomosar2height_amd.csrc.isa_pass pad N x.s x_padded.s [regex]   # `s_nop N` after every VALU write of an SGPR / VCC
                                                                              # (in functions matching regex), for the A/B builds

Rules (gfx940 family = gfx942 / gfx950; LLVM GCNHazardRecognizer.cpp, `hasVDecCoExecHazard()` == GFX940Insts, and the CDNA3 ISA
guide's "manually inserted wait states" table).  A wait state = one issued instruction of the wave; `s_nop N` = N + 1.

    A  VALU writes an SGPR / VCC  ->  VALU reads it (operand, or implicitly: v_cndmask_e32, v_addc, v_subb, v_div_fmas)   2
    B  VALU writes an SGPR / VCC  ->  v_readlane / v_writelane with it as the lane select                                  4
    C  VALU writes EXEC           ->  v_readlane / v_readfirstlane / v_writelane                                           4
    D  VALU writes a VGPR         ->  v_readlane / v_readfirstlane reads it                                                1
    E  VALU writes an SGPR        ->  VMEM (buffer_/global_/flat_/scratch_) reads it                                       5
    F  SALU writes M0             ->  LDS-DMA / GDS / s_sendmsg / s_movrel                                                 1
    G  transcendental VALU writes a VGPR -> non-transcendental VALU reads it                                               1

The scan is linear over each function's text (labels do not reset it: a fall-through is a possible path; a taken branch only
adds wait states).  It reports every place where fewer wait states separate the pair than the rule asks, and whether either
instruction sits inside an inline-asm block (between `;;#ASMSTART` and `;;#ASMEND`)."""
import re
import sys

SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b|\b(vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0)\b")
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
VMEM = ("buffer_", "global_", "flat_", "scratch_")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.update(f"s{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1))
        elif m.group(3):
            out.add(f"s{m.group(3)}")
        else:
            g = m.group(4)
            out.update({"vcc": ("vcc_lo", "vcc_hi"), "exec": ("exec_lo", "exec_hi")}.get(g, (g,)))
    return out


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def split_ops(ops):
    return [o.strip() for o in ops.split(",")] if ops else []


class Ins:
    __slots__ = ("line", "op", "ops", "asm", "text")

    def __init__(self, line, text, asm):
        self.line, self.text, self.asm = line, text, asm
        parts = text.split(None, 1)
        self.op = parts[0]
        self.ops = split_ops(parts[1] if len(parts) > 1 else "")

    def is_valu(self):
        return self.op.startswith("v_") and not self.op.startswith("v_mfma") and not self.op.startswith("v_smfma")

    def sgpr_defs(self):
        """SGPRs / VCC / EXEC this VALU instruction writes."""
        op = self.op
        if not self.is_valu():
            return set()
        if op.startswith("v_cmpx"):
            return {"exec_lo", "exec_hi"}
        if op.startswith("v_cmp"):
            return sregs(self.ops[0]) if op.endswith("_e64") or (self.ops and re.match(r"s\[|vcc", self.ops[0])) else {"vcc_lo", "vcc_hi"}
        if op.startswith(("v_readlane", "v_readfirstlane")):
            return sregs(self.ops[0])
        if re.match(r"v_(add|sub|subrev)_co_|v_(addc|subb|subbrev)_co_|v_div_scale|v_mad_[ui]64_[ui]32", op):
            return sregs(self.ops[1]) if len(self.ops) > 1 else set()
        return set()

    def sgpr_uses(self):
        """SGPRs / VCC a VALU instruction reads (sources only; implicit VCC of the e32 carry / select forms)."""
        if not (self.is_valu() or self.op.startswith(("v_mfma", "v_smfma"))):
            return set()
        op = self.op
        ndef = 1
        if re.match(r"v_(add|sub|subrev)_co_|v_(addc|subb|subbrev)_co_|v_div_scale|v_mad_[ui]64_[ui]32", op):
            ndef = 2
        if op.startswith("v_cmpx") or (op.startswith("v_cmp") and op.endswith("_e32")):
            ndef = 0 if not (self.ops and re.match(r"vcc", self.ops[0])) else 1
        use = set()
        for o in self.ops[ndef:]:
            use |= sregs(o)
        if op.endswith("_e32") and re.match(r"v_cndmask|v_addc|v_subb|v_subbrev", op):
            use |= {"vcc_lo", "vcc_hi"}
        if op.startswith("v_div_fmas"):
            use |= {"vcc_lo", "vcc_hi"}
        return use

    def vgpr_def(self):
        if not self.is_valu() or self.op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
            return set()
        return vregs(self.ops[0]) if self.ops else set()

    def vgpr_uses(self):
        if not self.op.startswith("v_"):
            return set()
        start = 0 if self.op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and False else 1
        use = set()
        for o in self.ops[start:]:
            use |= vregs(o)
        if self.op.startswith("v_writelane") or "fmac" in self.op or "_mac_" in self.op:
            use |= vregs(self.ops[0])
        return use

    def states(self):
        if self.op == "s_nop":
            return int(self.ops[0], 0) + 1
        return 1


def parse(path):
    funcs, cur, asm = {}, None, False
    for n, raw in enumerate(open(path), 1):
        s = raw.split(";", 1)
        code = s[0].strip()
        if "#ASMSTART" in raw:
            asm = True
            continue
        if "#ASMEND" in raw:
            asm = False
            continue
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", raw.strip())
        if m and not raw.startswith("\t") and not m.group(1).startswith(".L"):
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None or not code or code.startswith(".") or code.endswith(":"):
            continue
        if re.match(r"^(v_|s_|ds_|buffer_|global_|flat_|scratch_)", code):
            cur.append(Ins(n, code, asm))
    return funcs


def scan(funcs):
    found = []
    for name, ins in funcs.items():
        for i, a in enumerate(ins):
            sdefs = a.sgpr_defs()
            vdef = a.vgpr_def()
            trans = a.op.startswith(TRANS)
            m0 = a.op.startswith("s_") and a.ops and a.ops[0] == "m0"
            if not (sdefs or vdef or m0):
                continue
            waited = 0
            for b in ins[i + 1:i + 8]:
                if waited >= 5:
                    break
                rule = None
                if sdefs:
                    exec_def = "exec_lo" in sdefs
                    lane = b.op.startswith(("v_readlane", "v_writelane"))
                    if lane and len(b.ops) >= 3 and (sregs(b.ops[2]) & sdefs) and waited < 4:
                        rule = ("B", 4)
                    elif exec_def and b.op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) and waited < 4:
                        rule = ("C", 4)
                    elif (b.sgpr_uses() & sdefs) and waited < 2:
                        rule = ("A", 2)
                    elif b.op.startswith(VMEM) and (sregs(" ".join(b.ops)) & sdefs) and waited < 5:
                        rule = ("E", 5)
                if rule is None and vdef and waited < 1:
                    if b.op.startswith(("v_readlane", "v_readfirstlane")) and (vregs(" ".join(b.ops[1:2])) & vdef):
                        rule = ("D", 1)
                    elif trans and b.is_valu() and not b.op.startswith(TRANS) and (b.vgpr_uses() & vdef):
                        rule = ("G", 1)
                if rule is None and m0 and waited < 1 and (("lds" in b.text and b.op.startswith(VMEM)) or
                                                           b.op.startswith(("s_sendmsg", "s_movrel", "ds_gws", "ds_add_tid"))):
                    rule = ("F", 1)
                if rule:
                    found.append((name, rule[0], rule[1], waited, a, b))
                # a redefinition of everything `a` wrote ends the window for the SGPR rules
                waited += b.states()
    return found


def main(paths):
    total = 0
    for p in paths:
        funcs = parse(p)
        hits = scan(funcs)
        n_ins = sum(len(v) for v in funcs.values())
        print(f"{p}: {len(funcs)} functions, {n_ins} instructions, {len(hits)} finding(s)")
        for name, rule, need, have, a, b in hits:
            where = "inline asm" if (a.asm or b.asm) else "compiler"
            print(f"  rule {rule} (needs {need}, has {have}) [{where}] {name[:60]}\n      {a.line}: {a.text}\n      {b.line}: {b.text}")
        total += len(hits)
    return total


def pad(src, dst, nops, only=None, kinds=("sgpr",)):
    """Copy the assembly `src` to `dst` with `s_nop nops` inserted after every VALU instruction that writes an SGPR, VCC or
    EXEC (compares, v_readlane / v_readfirstlane, carry-outs), inline-asm text included -- every consumer of such a result then
    sits at least nops + 1 wait states behind its producer.  `only`: regex on the function label.  Returns the count."""
    rx = re.compile(only) if only else None
    cur, n, out = None, 0, []
    for raw in open(src):
        out.append(raw)
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", raw.strip())
        if m and not raw.startswith("\t") and not m.group(1).startswith(".L"):
            cur = m.group(1)
            continue
        code = raw.split(";", 1)[0].strip()
        if cur is None or not code.startswith("v_") or (rx and not rx.search(cur)):
            continue
        if Ins(0, code, False).sgpr_defs():
            out.append(f"\ts_nop {nops}\n")
            n += 1
    open(dst, "w").writelines(out)
    return n


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pad":
        print(pad(sys.argv[3], sys.argv[4], int(sys.argv[2]), sys.argv[5] if len(sys.argv) > 5 else None), "s_nop inserted")
        sys.exit(0)
    sys.exit(1 if main(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "scan" else sys.argv[1:]) else 0)


def lifetimes(funcs, min_len=12):
    """For every VALU write of an SGPR pair / VCC: the number of instructions until its LAST read by a vector instruction before
    the register is written again (linear scan; a read after a backward branch is not followed).  Long-lived compare results are
    the values the r05 / r06 co-residency fault was seen to eat: returns {function: [(length, def, use)]}
    for lengths >= min_len."""
    out = {}
    for name, ins in funcs.items():
        rows = []
        for i, a in enumerate(ins):
            d = a.sgpr_defs() - {"exec_lo", "exec_hi"}
            if not d or not a.op.startswith("v_cmp"):
                continue
            last = None
            for j in range(i + 1, min(i + 4000, len(ins))):
                b = ins[j]
                if b.sgpr_uses() & d:
                    last = j
                wr = set()
                if b.op.startswith("s_") and b.ops:
                    wr = sregs(b.ops[0])
                wr |= b.sgpr_defs()
                if wr & d and j != i:
                    if d <= wr:
                        break
                    d = d - wr
            if last is not None and last - i >= min_len:
                rows.append((last - i, a, ins[last]))
        if rows:
            out[name] = sorted(rows, key=lambda r: -r[0])
    return out


def report_lifetimes(paths, min_len=12):
    for p in paths:
        for name, rows in lifetimes(parse(p), min_len).items():
            print(f"{p}: {name[:90]}: {len(rows)} compare result(s) read by a vector instruction >= {min_len} instructions after the compare; "
                  f"longest {rows[0][0]}")
            for ln, a, b in rows[:3]:
                print(f"      {ln:5d}  {a.line}: {a.text}   ...   {b.line}: {b.text}")
