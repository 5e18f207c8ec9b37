#!/usr/bin/env python3
"""Scan AMDGPU assembly (hipcc -S --cuda-device-only) for a miscompile seen with hipcc 7.2 on register-starved
kernels: an ordinary VGPR spill / reload placed INSIDE a whole-wave-mode bracket

    s_or_saveexec_b64 s[a:b], -1   ...   s_mov_b64 exec, s[a:b]

that the compiler opened only to copy its SGPR-spill VGPRs.  Inside the bracket every lane is active, so a spill store
writes the register's content of lanes that were inactive when the value was defined (garbage) over their copies in
the slot; a later reload under a wider exec mask then sees that garbage.  (Found with rocgdb in msw_sim_kernel<double,
true, 2, NN> - a zero offset register of the last wavefront's record store - see DESIGN.md section 9 and LABBOOK.md section 4.)
    python tools/wwm_spill_scan.py file.s [...]      exit code 1 if any hit
"""
import re, sys
hits = 0
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    # registers a function writes with v_writelane from an SGPR hold spilled SGPRs: saving THOSE in whole-wave mode is right
    wwm, func = {}, None
    for line in lines:
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m: func = m.group(1); wwm[func] = set()
        m = re.match(r"v_writelane_b32 (v\d+), s\d+, \d+$", s)
        if m and func: wwm[func].add(m.group(1))
    func, open_reg, n_open = None, None, 0
    for ln, line in enumerate(lines, 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m: func = m.group(1); open_reg = None
        m = re.match(r"s_or_saveexec_b64 (s\[\d+:\d+\]), -1", s)
        if m: open_reg, n_open = m.group(1), ln; continue
        if open_reg and re.match(r"s_mov_b64 exec, " + re.escape(open_reg), s): open_reg = None; continue
        if open_reg and (s.startswith(".LBB") or s.startswith("s_cbranch") or s.startswith("s_branch")): open_reg = None; continue
        if open_reg and re.match(r"scratch_(store|load)_dword", s) and "Folded" in s:
            regs = re.findall(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+))", s.split(";")[0])
            names = set()
            for kind, lo, hi, one in regs:
                if one: names.add(kind + one)
                else: names.update(kind + str(i) for i in range(int(lo), int(hi) + 1))
            if names and names <= wwm.get(func, set()): continue   # an SGPR-spill VGPR: whole-wave save is what it needs
            hits += 1
            print(f"{path}:{ln}: {func[:60] if func else '?'}: {s}   (bracket opened at line {n_open})")
print(f"{hits} spill(s) inside whole-wave-mode brackets")
sys.exit(1 if hits else 0)
