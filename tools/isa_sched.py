#!/usr/bin/env python
"""Compact view of a kernel's instruction schedule: M=mfma v=VALU e=transcendental r=ds_read G=global_load_lds B=barrier n=s_nop w[..]=waitcnt .=other
Usage: python tools/isa_sched.py file.hip mangled_substring [first_line last_line]"""
import re, subprocess, sys, os
src, key = sys.argv[1], sys.argv[2]
out = "/tmp/probe/_isa.s"
os.makedirs("/tmp/probe", exist_ok=True)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", root + "/include", "-I", os.path.dirname(src),
                "-S", src, "-o", out, "--cuda-device-only"], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
m = re.search(r"^(\S*%s\S*):" % re.escape(key), s, re.M)
start = m.start()
end = s.index(".Lfunc_end", start)
body = s[start:end].split("\n")
comp = []
for l in body:
    t = l.strip()
    if not t or t.startswith((";", "//")):
        continue
    if t.startswith(".LBB"):
        comp.append("\n" + t.split()[0] + " ")
        continue
    if t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"): comp.append("M")
    elif op.startswith("ds_read"): comp.append("r")
    elif op.startswith("ds_write"): comp.append("W")
    elif op.startswith("global_load_lds"): comp.append("G")
    elif op.startswith(("global_load", "buffer_load")): comp.append("L")
    elif op.startswith(("global_store", "buffer_store")): comp.append("S")
    elif op.startswith("s_waitcnt"): comp.append("w[" + t.split(None, 1)[1].replace(" ", "") + "]")
    elif op.startswith("s_barrier"): comp.append("B")
    elif op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): comp.append("e")
    elif op.startswith("v_"): comp.append("v")
    elif op.startswith("s_nop"): comp.append("n")
    elif op.startswith(("s_cbranch", "s_branch")): comp.append("<" + t.split()[1] + ">")
    else: comp.append(".")
txt = "".join(comp)
print(txt[:int(sys.argv[3])] if len(sys.argv) > 3 else txt)
m2 = re.search(r"\.vgpr_count:\s+(\d+)", s[end:])
print("vgpr_count(first after):", m2.group(1) if m2 else "?")
