#!/usr/bin/env python3
"""Static instruction mix of one kernel from the gfx950 disassembly, priced with the measured issue costs of DESIGN.md 4.0
(profiles/r01_valu_issue_cost_*.txt): a VALU instruction costs 1 (v_add/sub/shift-right/logic/mov/f32 mul-add) or ~1.7 units
(everything else), 1 unit = one `v_add_u32` issue slot of a SIMD.  Straight-line kernels only (loops are counted once; the
transform kernels have none on their hot path).

usage: tools/isa_mix.py <code object or .s> <kernel name substring> [...]
  (code objects: llvm-objdump --offloading libsvt_hip_dsp.so, then pass the *.gfx950 file)"""
import collections
import re
import subprocess
import sys

CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_lshrrev_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
         "v_mov_b32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_max_i16", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}


def disassemble(path):
    if path.endswith(".s"):
        return open(path).read()
    return subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", path]).decode()


def kernels(text):
    cur, out = None, collections.OrderedDict()
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1); out[cur] = []
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if m:
            out[cur].append((m.group(1), m.group(2)))
    return out


def price(instrs):
    cnt = collections.Counter()
    units = 0.0
    for op, args in instrs:
        base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
        if base.startswith("v_") and not base.startswith("v_mfma"):
            sgpr_src1 = False
            if base in CHEAP and (op.endswith("_e64") or op.endswith("sdwa") or op.endswith("dpp")):
                sgpr_src1 = True                       # VOP3 / SDWA / DPP encodings of a cheap op issue at the slow rate
            cheap = base in CHEAP and not sgpr_src1
            cnt["valu_cheap" if cheap else "valu_slow"] += 1
            units += 1.0 if cheap else 1.7
            cnt["op:" + base] += 1
        elif base.startswith("ds_"):
            cnt["lds"] += 1
        elif base.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cnt["vmem"] += 1
        elif base.startswith("s_"):
            cnt["salu" if not base.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_cbranch", "s_branch")) else "sctl"] += 1
    return cnt, units


def main():
    text = disassemble(sys.argv[1])
    ks = kernels(text)
    for pat in sys.argv[2:]:
        for name, ins in ks.items():
            if pat in name and ins:
                cnt, units = price(ins)
                valu = cnt["valu_cheap"] + cnt["valu_slow"]
                print(f"== {name[:150]}")
                print(f"   VALU {valu} (cheap {cnt['valu_cheap']}, slow {cnt['valu_slow']}) = {units:.0f} issue units; LDS {cnt['lds']}, VMEM {cnt['vmem']}, SALU {cnt['salu']}")
                top = sorted(((v, k[3:]) for k, v in cnt.items() if k.startswith("op:")), reverse=True)[:14]
                print("   top: " + ", ".join(f"{k} {v}" for v, k in top))


if __name__ == "__main__":
    main()
