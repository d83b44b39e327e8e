#!/usr/bin/env python3
"""Static check of the gfx9 hazard "VALU writes a VGPR -> a DPP instruction reads that VGPR: 2 wait states" in a gfx950 assembly
listing (hipcc -save-temps / -S).  The compiler's hazard recogniser does not look inside inline asm, and the row-cooperative
kernel (csrc/mtg_coop.hip) issues its v_fmac_f64_dpp through inline asm -- this script is the proof that its s_nop placement
holds in the code the compiler actually emitted.  usage: check_dpp_hazards.py file.s [kernel-name-substring]
Exit status 1 if a violation is found.  Also prints instruction-class counts per kernel."""
import collections
import re
import sys


def regs(tok):
    """VGPR numbers named by an operand token like v12, v[12:13]."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.startswith(".") or line.endswith(":"):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = [t.strip() for t in re.split(r",\s*", parts[1])] if len(parts) > 1 else []
    ops = [o.split()[0] if o else o for o in ops]     # drop modifiers glued to the last operand
    return op, ops


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    kernels, cur, name = {}, None, None
    for raw in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):", raw)
        if m and not raw.startswith(".L"):
            name = m.group(1)
            cur = kernels.setdefault(name, [])
            continue
        if cur is None:
            continue
        if re.match(r"^\.L\w+:", raw):
            cur.append(("label", []))
            continue
        if raw.startswith("\t"):
            p = parse(raw)
            if p:
                cur.append(p)
    bad = 0
    for name, ins in kernels.items():
        if want not in name or not any("dpp" in op for op, _ in ins):
            continue
        counts = collections.Counter()
        n_dpp = 0
        for idx, (op, ops) in enumerate(ins):
            if op == "label":
                continue
            cls = ("dpp" if "dpp" in op else "f64" if "f64" in op else "salu" if op.startswith("s_") else
                   "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
                   "accvgpr" if "accvgpr" in op else "valu")
            counts[cls] += 1
            if "dpp" not in op:
                continue
            n_dpp += 1
            src = regs(ops[1]) if len(ops) > 1 else set()
            waits, k = 0, idx - 1
            while k >= 0 and waits < 2:
                pop, pops = ins[k]
                if pop == "label":
                    break          # (a branch target: the fall-through / jump paths were checked up to here)
                if pop == "s_nop":
                    waits += int(pops[0]) + 1 if pops else 1
                else:
                    if pop.startswith("v_") and pops and regs(pops[0]) & src:
                        print(f"HAZARD in {name}: '{pop} {', '.join(pops)}' writes the DPP source of '{op} {', '.join(ops)}' {waits} wait state(s) earlier")
                        bad += 1
                    waits += 1
                k -= 1
        print(f"{name[:90]}: {sum(counts.values())} instructions, {n_dpp} DPP, classes {dict(counts)}")
    print("DPP hazard check:", "FAILED (%d)" % bad if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
