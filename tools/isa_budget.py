#!/usr/bin/env python
"""Per-instruction budget of a kernel's hottest loop, from the ISA hipcc emits (no GPU needed).
    python tools/isa_budget.py kb-ner_amd/csrc/attention.hip attn_bwd_dq2_kernelILb0ELb1 [more mangled substrings ...]
For each kernel: the basic block (between two labels) with the most v_mfma instructions = one trip of the key-chunk loop; opcode
histogram of that block, grouped into issue classes with the port cycles the round-3 probe measured (tools/micro/valu_probe.hip,
profiles/round3_valu_probe.txt): plain VALU 4 cycles per wave64 instruction, v_exp / v_rcp 8, v_cvt_pk_bf16_f32 4, MFMA 16x16x32
4 of issue + 16.4 of matrix pipe; LDS / scalar / waits listed, not priced (other ports)."""
import collections, os, re, subprocess, sys

src, keys = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/probe/_budget.s"
os.makedirs("/tmp/probe", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", root + "/include", "-I", os.path.dirname(src),
                "-S", src, "-o", out, "--cuda-device-only"], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "transcendental"
    if op.startswith("v_cvt_pk_bf16"): return "pack (v_cvt_pk_bf16_f32)"
    if op.startswith(("v_pk_mul", "v_pk_fma", "v_pk_add")): return "packed fp32"
    if op.startswith(("v_mul_f32", "v_fma_f32", "v_add_f32", "v_sub_f32", "v_fmac")): return "fp32 multiply / add"
    if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_lshl", "v_and", "v_or", "v_xor", "v_add3", "v_lshl_add", "v_mad_u")): return "integer / address"
    if op.startswith(("v_cndmask", "v_cmp")): return "compare / select"
    if op.startswith(("v_mov", "v_accvgpr", "v_perm", "v_bfi", "v_readfirstlane", "v_readlane")): return "move / permute"
    if op.startswith("v_"): return "other VALU (" + op + ")"
    if op.startswith("ds_read"): return "LDS read"
    if op.startswith("ds_"): return "LDS other"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "scalar"
    return "memory / other"


PORT = {"transcendental": 8}
for key in keys:
    m = re.search(r"^(\S*%s\S*):" % re.escape(key), s, re.M)
    if not m:
        print("no kernel matching", key)
        continue
    body = s[m.start():s.index(".Lfunc_end", m.start())].split("\n")
    blocks, cur = [], []
    for l in body:
        t = l.strip()
        if not t or t.startswith((";", "//")):
            continue
        if t.startswith(".LBB") or t.endswith(":"):
            blocks.append(cur); cur = []
            continue
        if t.startswith("."):
            continue
        cur.append(t.split()[0])
    blocks.append(cur)
    hot = max(blocks, key=lambda b: sum(o.startswith("v_mfma") for o in b))
    h = collections.Counter(classify(o) for o in hot)
    ops = collections.Counter(o for o in hot if o.startswith("v_") and not o.startswith("v_mfma"))
    nm = h["mfma"]
    valu = sum(v for k, v in h.items() if k not in ("mfma", "LDS read", "LDS other", "s_waitcnt", "s_nop", "scalar", "memory / other"))
    port = sum(v * PORT.get(k, 4) for k, v in h.items() if k not in ("mfma", "LDS read", "LDS other", "s_waitcnt", "s_nop", "scalar", "memory / other"))
    print("== %s: hottest block %d instructions, %d MFMA" % (m.group(1)[:70], len(hot), nm))
    for k, v in sorted(h.items(), key=lambda kv: -kv[1]):
        print("   %-34s %4d" % (k, v))
    print("   VALU-class instructions per MFMA: %.2f   (VALU port cycles %d + MFMA issue %d; matrix pipe %.0f) per wave and trip" % (valu / max(nm, 1), port, 4 * nm, 16.4 * nm))
    print("   VALU opcodes:", ", ".join("%s x%d" % kv for kv in ops.most_common(14)))
