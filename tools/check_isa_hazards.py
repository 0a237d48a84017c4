#!/usr/bin/env python3
"""check_isa_hazards.py — static check of the compiled kernels (fs_kernels.s) for the one hazard hand-issued loads have:

The traversal requests its records with inline-asm `global_load_dwordx4` and waits for them later with an inline-asm
`s_waitcnt vmcnt(0)` (fs_kernels.hip: trav_issue / trav_wait).  The compiler does not know that the asm's output
registers are still in flight between the two: if register allocation puts a copy, a spill or a temporary on them in
between, the kernel reads stale data (round 3: a build whose closest-hit loop copied the requested triangle registers
right behind the request returned wrong hits and faulted).  This script proves, per kernel, on the control-flow graph
of the final ISA, that no instruction reads or writes a VGPR that an inline-asm load may still be writing.

    python tools/check_isa_hazards.py audio-pathtracer_amd/csrc/build/*gfx950.s     (exit code 1 on a hazard; run by the Makefile)

Rules: a register range becomes IN FLIGHT at an asm `global_load_*` (between ;;#ASMSTART / ;;#ASMEND); every
`s_waitcnt vmcnt(0)` (ours or the compiler's) lands everything; a later asm load to the same registers is allowed
(loads return in order); any other instruction that names an in-flight register is a hazard.  The in-flight set is
propagated over the CFG (union over predecessors, to a fixed point).  Numeric local labels of inline-asm statements (`2:`,
`s_cbranch_execz 2f`) are basic-block boundaries and edges like the compiler's own: a branch around a wait keeps its registers in flight.
Counted waits (round 4, the cooperative traversal): vector memory operations complete in issue order, so
`s_waitcnt vmcnt(N)` lands everything but the N youngest.  The checker keeps the ORDER of the vector memory instructions
of the current basic block (asm or not: loads, stores, atomics all count); a counted wait that has at least N of them
behind it in its own block keeps only the destinations of those N youngest in flight — everything older, including
whatever was in flight at the block's entry, has landed.  With fewer than N in the block nothing is concluded.
"""
from __future__ import annotations

import re
import sys
from collections import defaultdict

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
LOCAL = re.compile(r"^\s*(\d+):\s*$")
LOCAL_BRANCH = re.compile(r"^s_(cbranch_\w+|branch)\s+(\d+)([fb])\b")
FUNC = re.compile(r"^(_Z\w+):")
BRANCH = re.compile(r"^\s+s_(cbranch_\w+|branch)\s+(\.LBB\d+_\d+)")
VMCNT = re.compile(r"vmcnt\((\d+)\)")
VMEM = re.compile(r"^(global|buffer|scratch)_(load|store|atomic)")


def regs_of(text: str) -> set[int]:
    out: set[int] = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_function(name: str, lines: list[str]) -> list[str]:
    # basic blocks: (label or index) -> list of (lineno, text, in_asm)
    blocks: list[dict] = [{"label": None, "ins": [], "succ": [], "line": 0}]
    in_asm = False
    for no, raw in lines:
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw.lstrip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.lstrip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = LABEL.match(raw)
        if m:
            blocks.append({"label": m.group(1), "ins": [], "succ": [], "line": no})
            continue
        m = LOCAL.match(raw)
        if m:   # a numeric local label of an inline-asm statement (`2:`; branches name it `2f` / `2b`)
            blocks.append({"label": None, "local": m.group(1), "ins": [], "succ": [], "line": no})
            continue
        if not line.strip() or line.strip().startswith(".") or line.strip().startswith(";"):
            continue
        blocks[-1]["ins"].append((no, line.strip(), in_asm))
        if re.match(r"^\s*s_(cbranch|branch|endpgm|setpc)", line):
            blocks.append({"label": None, "ins": [], "succ": [], "line": no})   # fallthrough block after a branch
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    for i, b in enumerate(blocks):
        fall = True
        if b["ins"]:
            no, last, _ = b["ins"][-1]
            m = re.match(r"^s_(cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", last)
            if m:
                if m.group(2) in index:
                    b["succ"].append(index[m.group(2)])
                if m.group(1) == "branch":
                    fall = False
            ml = LOCAL_BRANCH.match(last)
            if ml:   # the nearest local label of that number, forwards or backwards — an edge like any other (a branch AROUND a
                     # wait leaves more in flight than the fall-through path: it must not be dropped)
                cands = [j for j, c in enumerate(blocks) if c.get("local") == ml.group(2) and ((j > i) if ml.group(3) == "f" else (j <= i))]
                if not cands:
                    raise SystemExit(f"{name}: line {no}: local label of `{last}` not found")
                b["succ"].append(min(cands) if ml.group(3) == "f" else max(cands))
                if ml.group(1) == "branch":
                    fall = False
            if last.startswith("s_endpgm") or last.startswith("s_setpc"):
                fall = False
        if fall and i + 1 < len(blocks):
            b["succ"].append(i + 1)
    preds = defaultdict(list)
    for i, b in enumerate(blocks):
        for s in b["succ"]:
            preds[s].append(i)

    def transfer(bi: int, inflight: frozenset, report: list | None):
        cur = set(inflight)
        recent: list[set[int]] = []     # vector memory instructions of this block so far, in issue order: the asm loads' destinations
        ordered = True                  # False behind a flat_* operation (those complete out of order)
        for no, ins, asm in blocks[bi]["ins"]:
            if ins.startswith("s_waitcnt") and "vmcnt(0)" in ins:
                cur.clear()
                recent.clear()
                ordered = True
                continue
            mw = VMCNT.search(ins) if ins.startswith("s_waitcnt") else None
            if mw:
                n = int(mw.group(1))
                if ordered and 0 < n <= len(recent):
                    cur = set().union(*recent[-n:])
                    recent = recent[-n:]
                continue
            touched = regs_of(ins)
            if asm and ins.startswith("global_load"):
                dst = regs_of(ins.split(",")[0])
                srcs = touched - dst
                bad = srcs & cur
                if bad and report is not None:
                    report.append(f"{name}: line {no}: address of `{ins}` uses in-flight v{sorted(bad)}")
                cur |= dst
                recent.append(set(dst))
                continue
            if VMEM.match(ins):
                recent.append(set())
            elif ins.startswith("flat_"):
                ordered = False
            bad = touched & cur
            if bad and report is not None:
                report.append(f"{name}: line {no}: `{ins}` touches in-flight v{sorted(bad)}")
        return frozenset(cur)

    state_in = [frozenset()] * len(blocks)
    state_out = [frozenset()] * len(blocks)
    work = list(range(len(blocks)))
    while work:
        bi = work.pop(0)
        new_in = frozenset().union(*[state_out[p] for p in preds[bi]]) if preds[bi] else frozenset()
        new_out = transfer(bi, new_in, None)
        if new_in != state_in[bi] or new_out != state_out[bi]:
            state_in[bi], state_out[bi] = new_in, new_out
            for s in blocks[bi]["succ"]:
                if s not in work:
                    work.append(s)
    report: list[str] = []
    for bi in range(len(blocks)):
        transfer(bi, state_in[bi], report)
    return report


def main() -> int:
    paths = sys.argv[1:] or ["audio-pathtracer_amd/csrc/fs_kernels.s"]
    problems: list[str] = []
    checked = 0
    for path in paths:
        funcs: dict[str, list] = {}
        cur = None
        with open(path) as f:
            for no, raw in enumerate(f, 1):
                m = FUNC.match(raw)
                if m:
                    cur = m.group(1)
                    funcs[cur] = []
                    continue
                if cur is None:
                    continue
                if raw.startswith(".Lfunc_end"):
                    cur = None
                    continue
                funcs[cur].append((no, raw.rstrip("\n")))
        for name, lines in funcs.items():
            if not any("global_load" in t for _, t in lines):
                continue
            checked += 1
            problems += [f"{path}: {p}" for p in check_function(name, lines)]
    for p in problems[:40]:
        print("HAZARD", p)
    print(f"check_isa_hazards: {checked} kernels with loads checked in {len(paths)} file(s), {len(problems)} hazards")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
