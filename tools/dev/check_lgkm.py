#!/usr/bin/env python3
"""Static check of the LDS-read waits in a compiled k loop (no GPU needed).

The pipelined k loops with hand-counted waits (k_gemm_bf16x.hip ASMW, k_gemm3x.hip HOIST = 3) issue their fragment reads as inline asm
and place `s_waitcnt lgkmcnt(n)` themselves; a count that is one too large reads a register before its data has landed -- silently, and
only sometimes.  This tool takes the COMPILED instruction stream of a kernel (hipcc --save-temps), walks the function linearly up to the
end of its innermost loop and then around the loop twice more, and models the wave's LGKM queue exactly as the hardware defines it for
LDS operations (they complete in issue order; `lgkmcnt(n)` returns when at most n are outstanding):

  * every ds_read* pushes its destination registers; other DS / scalar-memory operations push an anonymous entry;
  * `s_waitcnt ... lgkmcnt(n)` retires all but the newest n entries (`s_waitcnt` forms without an lgkmcnt field retire nothing);
  * any instruction that names a register with an unretired read pending -- as a source OR as a destination -- is an error.

    python tools/dev/check_lgkm.py stable_diffusion_burn_amd/csrc/k_gemm_bf16x.hip --kernel ILi8ELi5ELi2ELi4ELi2E

Exit status 1 on the first hazard found (printed with the offending instruction and the read it races with).  It also prints, per
wait inside the loop, how many matrix instructions lie between the newest read the wait covers and the wait itself (the latency the
schedule hides).  tests/test_lgkm_waits_cpu.py runs it on every hand-counted kernel.
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import tempfile

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text: str) -> set[int]:
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_asm(src: str) -> str:
    d = tempfile.mkdtemp(prefix="lgkm_")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-c", os.path.abspath(src), "-o", "x.o", "--save-temps"]
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    return open(os.path.join(d, asm)).read()


def function_lines(asm: str, name: str) -> list[str]:
    i = asm.index(name + ":")
    j = asm.index(".Lfunc_end", i)
    out = []
    for l in asm[i:j].split("\n")[1:]:
        l = l.split(";")[0].strip() if not l.strip().startswith(".LBB") else l.strip()
        if l:
            out.append(l)
    return out


def innermost_loop(lines: list[str]) -> tuple[int, int]:
    """(index of the loop header label, index of its backward branch): the LAST backward branch whose target is the closest label above"""
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if l.startswith(".LBB")}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\S+)", l) or re.match(r"s_branch\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = i - labels[m.group(1)]
            n_mfma = sum("v_mfma" in x for x in lines[labels[m.group(1)]:i])
            if n_mfma and (best is None or n_mfma > best[2]):
                best = (labels[m.group(1)], i, n_mfma)
    if best is None:
        raise SystemExit("no loop with matrix instructions found")
    return best[0], best[1]


class Queue:
    def __init__(self):
        self.q = []          # [regs, text, index]
        self.mfma_at = []    # running count of matrix instructions, for the "latency hidden" report
        self.n_mfma = 0

    def push(self, regs, text):
        self.q.append((regs, text, self.n_mfma))

    def wait(self, n):
        covered = None
        while len(self.q) > n:
            covered = self.q.pop(0)
        return covered

    def pending(self, regs):
        for r, text, _ in self.q:
            if r & regs:
                return text
        return None


def check(lines: list[str], verbose: bool) -> int:
    h, b = innermost_loop(lines)
    stream = [(i, l) for i, l in enumerate(lines[:b + 1])] + 2 * [(i, l) for i, l in enumerate(lines[h:b + 1], start=h)]
    q = Queue()
    report = []
    passes = 0
    for pos, (i, l) in enumerate(stream):
        if l.startswith(".LBB"):
            continue
        op = l.split()[0]
        in_loop_steady = pos > b       # second and third time around
        if op.startswith("ds_read") or (op.startswith("ds_") and "rtn" in op):
            dst = regs_of(l.split(",")[0])
            src = regs_of(",".join(l.split(",")[1:]))
            bad = q.pending(dst | src)
            if bad:
                print(f"HAZARD line {i}: `{l}` touches a register of the pending `{bad}`")
                return 1
            q.push(dst, l)
            continue
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            bad = q.pending(regs_of(l))
            if bad:
                print(f"HAZARD line {i}: `{l}` touches a register of the pending `{bad}`")
                return 1
            q.push(set(), l)
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                covered = q.wait(int(m.group(1)))
                if in_loop_steady and covered is not None:
                    report.append((i, l, q.n_mfma - covered[2], len(q.q)))
            elif not re.search(r"vmcnt|expcnt", l):      # bare numeric form: treat as a full wait
                q.wait(0)
            continue
        if op == "s_barrier":
            continue
        if "v_mfma" in op:
            q.n_mfma += 1
        bad = q.pending(regs_of(l))
        if bad:
            print(f"HAZARD line {i}: `{l}` uses a register of the pending `{bad}`")
            return 1
    n_wait = len(report) // 2 if report else 0
    if verbose:
        print(f"loop lines {h}..{b}: {sum('v_mfma' in l for l in lines[h:b + 1])} matrix instructions, "
              f"{sum(l.split()[0].startswith('ds_read') for l in lines[h:b + 1] if not l.startswith('.'))} LDS reads, {n_wait} waits that retire a read")
        for i, l, dist, left in report[:n_wait]:
            print(f"  line {i}: {l:32s} newest read it covers was issued {dist:3d} matrix instructions earlier; {left} reads stay in flight")
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--kernel", required=True, help="substring of the mangled kernel name (all matches are checked)")
    ap.add_argument("-q", "--quiet", action="store_true")
    args = ap.parse_args()
    asm = compile_asm(args.src)
    names = [m for m in re.findall(r"^(\w+):\s*; @", asm, re.M) if args.kernel in m]
    if not names:
        sys.exit(f"no kernel matches {args.kernel}")
    rc = 0
    for name in names:
        if not args.quiet:
            print("==", name)
        rc |= check(function_lines(asm, name), not args.quiet)
    print("lgkm waits OK" if rc == 0 else "lgkm waits: HAZARD")
    sys.exit(rc)


if __name__ == "__main__":
    main()
