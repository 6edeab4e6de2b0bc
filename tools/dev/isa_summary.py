#!/usr/bin/env python3
"""Static look at a cross-compiled kernel file (no GPU needed): registers / scratch / LDS per kernel from
-Rpass-analysis=kernel-resource-usage, and the instruction stream of one kernel's k loop in compressed form.

    python tools/dev/isa_summary.py stable_diffusion_burn_amd/csrc/k_gemm3x.hip [--kernel SUBSTR] [--mix]

--mix: the instruction mix of the k loop (the innermost loop with matrix instructions) of every kernel that matches --kernel:
matrix / other VALU / SALU / LDS / VMEM / waits per trip -- the numbers DESIGN.md and profiles/README.md quote.
"""
import argparse, re, subprocess, tempfile, os, sys


def function_lines(asm, name) :
    i = asm.index(name + ":")
    j = asm.index(".Lfunc_end", i)
    out = []
    for l in asm[i:j].split("\n")[1:]:
        l = l.split(";")[0].strip() if not l.strip().startswith(".LBB") else l.strip()
        if l:
            out.append(l)
    return out


def innermost_loop(lines) :
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


ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("--kernel", default=None, help="substring of the mangled name: dump that kernel's stream")
ap.add_argument("--full", action="store_true", help="dump every instruction, not only the compressed stream")
ap.add_argument("--mix", action="store_true", help="instruction mix of the k loop instead of the stream")
args = ap.parse_args()
d = tempfile.mkdtemp(prefix="isa_")
src = os.path.abspath(args.src)
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-c", src, "-o", "x.o",
       "--save-temps", "-Rpass-analysis=kernel-resource-usage"]
r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr)
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = b.split()[0]
    g = lambda k: re.search(k + r": (\S+)", b).group(1)
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{name[:110]:110s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} SGPR {g('SGPRs'):>3s} scratch {scr} occ {occ}")
if args.kernel and args.mix:
    import collections
    asm_text = open(os.path.join(d, [f for f in os.listdir(d) if f.endswith("gfx950.s")][0])).read()
    for name in [m for m in re.findall(r"^(\w+):\s*; @", asm_text, re.M) if args.kernel in m]:
        lines = function_lines(asm_text, name)
        try:
            h, b = innermost_loop(lines)
        except SystemExit:
            continue
        c = collections.Counter()
        for l in lines[h:b + 1]:
            if l.startswith("."):
                continue
            op = l.split()[0]
            c["matrix" if "v_mfma" in op else "valu" if op.startswith("v_") else "wait" if op.startswith(("s_waitcnt", "s_barrier", "s_nop")) else
              "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"] += 1
        print(f"{name[:100]:100s} k loop: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())) + f"; total {sum(c.values())}")
    sys.exit(0)
if args.kernel:
    asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(d, asm)).read()
    names = [m for m in re.findall(r"^(\w+):\s*; @", s, re.M) if args.kernel in m]
    for name in names:
        i = s.index(name + ":")
        j = s.index(".Lfunc_end", i)
        print("\n==== " + name)
        row = []
        for l in s[i:j].split("\n"):
            l = l.strip()
            if not l or l.startswith(";"):
                continue
            if l.startswith("."):
                if l.startswith(".LBB"):
                    print(" | ".join(row)); row = []; print(l)
                continue
            m = l.split()[0]
            keep = args.full or m.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "global_load_lds", "ds_read", "ds_write", "s_setprio", "s_nop", "global_load", "global_store", "buffer_"))
            row.append(l.split(";")[0].strip() if keep else m)
            if len(row) == 8:
                print(" | ".join(row)); row = []
        print(" | ".join(row))
