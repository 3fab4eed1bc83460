#!/usr/bin/env python3
"""Weighted VALU issue-cost estimate of a range of gfx950 assembly (hipcc -S output), using the per-instruction
issue costs measured by tools/valu_rate_bench.hip on MI355X (wave64, 4 waves/SIMD):
  fast  2.3 cycles: v_mov_b32, v_add/sub/mul_f32, v_fma/fmac_f32, v_add/sub_u32, v_and/or/xor/not_b32, v_lshrrev/ashrrev
                    -- only in their plain form: an SGPR operand, DPP or SDWA makes them slow
  trans 8.2 cycles: v_exp/log/rcp/rsq/sqrt_f32 (and v_swap_b32)
  slow  4.4 cycles: everything else (v_cndmask, v_cmp, min/max/med3, all 64-bit and packed ops, conversions, v_lshlrev, ...)
usage: isa_cost.py file.s [first_line last_line]   (no range: every loop-free listing of the file)"""
import re, sys
FAST = {"v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_fmaak_f32", "v_fmamk_f32"}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_swap_b32", "v_rcp_iflag_f32", "v_sin_f32", "v_cos_f32"}
def classify(line):
    t = line.split(";")[0].split()
    if not t or not t[0].startswith("v_"): return None
    op = re.sub(r"_(e32|e64)$", "", t[0])
    if op in TRANS: return "trans"
    ops = " ".join(t[1:])
    if op in FAST and not re.search(r"\bs\d+\b|\bs\[|vcc|exec|quad_perm|row_|sdwa|_sel:", ops): return "fast"
    if op.endswith("_dpp") : return "slow"
    return "slow"
def main():
    lines = open(sys.argv[1]).read().splitlines()
    a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, len(lines))
    cnt = {"fast": 0, "slow": 0, "trans": 0}; other = {}; slow_ops = {}
    for l in lines[a - 1:b]:
        c = classify(l)
        if c: 
            cnt[c] += 1
            if c == "slow":
                k = l.split()[0]; 
                if re.search(r"\bs\d+\b|\bs\[", " ".join(l.split(";")[0].split()[1:])) and re.sub(r"_(e32|e64)$", "", k) in FAST: k += " (sgpr operand)"
                slow_ops[k] = slow_ops.get(k, 0) + 1
        else:
            t = l.split(";")[0].split()
            if t and re.match(r"(s_|ds_|global_|buffer_|flat_)", t[0]): 
                k = t[0].split("_")[0]; other[k] = other.get(k, 0) + 1
    cyc = 2.3 * cnt["fast"] + 4.4 * cnt["slow"] + 8.2 * cnt["trans"]
    print(f"lines {a}-{b}: VALU fast {cnt['fast']}  slow {cnt['slow']}  trans {cnt['trans']}  -> ~{cyc:.0f} SIMD cycles;  other: {other}")
    for k, v in sorted(slow_ops.items(), key=lambda kv: -kv[1])[:40]: print(f"   {v:4d} {k}")
main()
