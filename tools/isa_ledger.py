#!/usr/bin/env python3
"""Instruction ledger of one kernel from hipcc's -save-temps assembly: per basic block the number of VALU, transcendental,
packed / fp64, SALU, LDS and vector-memory instructions, with the loop nesting the compiler records.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -c meso_amd/csrc/pair_ring.hip -save-temps -o /tmp/x.o
    tools/isa_ledger.py pair_ring-hip-amdgcn-amd-amdhsa-gfx950.s _ZN4meso15k_pair_dpd_ringILb1ELi0ELb1ELb1ELi1ELb1EEEvNS_8PairArgsE
"""
import re, sys
path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
def klass(op):
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_") and ("_f64" in op or op.startswith("v_pk_")): return "valu64pk"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    return "other"
blocks, cur = [], {"name": "entry", "note": "", "n": {}}
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if m:
        blocks.append(cur)
        cur = {"name": m.group(1), "note": (m.group(2) or "").strip("; "), "n": {}, "marks": set()}
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    k = klass(op)
    cur["n"][k] = cur["n"].get(k, 0) + 1
    for key in ("v_sin_f32", "buffer_load_dwordx3", "buffer_load_dwordx4", "ds_add_u64", "ds_write_b128", "v_mbcnt_lo", "global_store", "s_barrier", "v_xad_u32"):
        if op.startswith(key): cur.setdefault("marks", set()).add(key)
blocks.append(cur)
cols = ["valu", "valu64pk", "trans", "salu", "branch", "lds", "vmem", "wait"]
tot = {c: 0 for c in cols}
print("%-12s " % "block" + " ".join("%8s" % c for c in cols) + "  marks / compiler note")
for b in blocks:
    if not b["n"]: continue
    print("%-12s " % b["name"] + " ".join("%8d" % b["n"].get(c, 0) for c in cols) + "  " + ",".join(sorted(b.get("marks", []))) + "  " + b["note"][:60])
    for c in cols: tot[c] += b["n"].get(c, 0)
print("%-12s " % "TOTAL" + " ".join("%8d" % tot[c] for c in cols))
for l in lines[end:end + 120]:
    if any(k in l for k in ("vgpr_count", "sgpr_count", "vgpr_spill", "sgpr_spill", "; Occupancy", "; LDSByteSize", "; ScratchSize", "; NumVgprs", "; NumSgprs")): print(l.strip())
