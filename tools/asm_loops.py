#!/usr/bin/env python3
"""Dev tool: innermost loops of one kernel in a gfx950 .s file with an instruction census per loop body.
    python tools/asm_loops.py lib/asm/kr_mso_f64.s <kernel-substring> [min_instr]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(key and l.rstrip()[-1]))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
labels = {}
for i in range(start, end):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m: labels[m.group(1)] = i
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"):
        if "f64" in op: return "valu_f64"
        if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")): return "trans"
        return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"
loops = []
for i in range(start, end):
    m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", lines[i])
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i, m.group(2)))
for a, b, lab in sorted(loops):
    c = collections.Counter()
    for l in lines[a:b + 1]:
        s = l.strip()
        if not s or s.startswith((";", ".")) or s.endswith(":"): continue
        c[cls(s.split()[0])] += 1
    n = sum(c.values())
    if n < minlen: continue
    inner = [x for x in loops if x[0] > a and x[1] < b]
    print(f"{lab}: lines {a+1}-{b+1} instr {n} inner_loops {len(inner)}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
