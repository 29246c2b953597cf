#!/usr/bin/env python
"""Generates diffassemble_amd/csrc/da_attn_dual_asm.inc: the two straight-line regions of k_attn_dual's fast key block
(da_attn_dual.hip) as inline-asm text with FIXED physical registers.

Why generated asm: the regions must interleave, instruction by instruction, the S^T MFMA chain of one query slab (with its
K-fragment LDS reads three ahead) with the exponentials / packs / sums of the other slab -- hipcc's scheduler issues each
chain as one burst, and sched_group_barrier pipelines of this length fall apart after two groups (measured, round 3).
Inline asm cannot name sub-registers of a tuple operand, so every value the regions touch is pinned to a physical register
through "{v[a:b]}" constraints (which the register allocator honours without copies) and the text names registers directly.

  python tools/gen_attn_dual_asm.py            (re-run after changing the register map or the slot schedule)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- register map (VGPR numbers); v0 .. v75 and v92 .. v99 stay with the compiler
VADDR, KADDR, ACC = 76, 77, 79            # (v78 is free: the (1, 1) bf16 constant of the row sums is an SGPR operand)
ET = 80                     # exp temporaries: pair A = 80, 81; pair B = 82, 83
VF = 84                     # V fragments: v0 = 84..87 (vlo0, vhi0), v1 = 88..91 (vlo1, vhi1)
PFB = 248                   # packed P of slab 1: the SAME registers as slab 0's (R2 reads slab 0's P at its two PV MFMAs, slots 0 - 1, and writes
                            # slab 1's packs from slot 2 on; MFMA operands are read at issue) -- frees v92..v99 for the compiler
Q0, Q1 = 100, 136           # Q fragments, 9 x 4 registers per slab (C = 144); C = 32 uses the first 2 x 4 of each
SA, SB = 172, 188           # score tuples
O0, O1 = 204, 220           # output accumulators
KT = 236                    # K fragment ring: 236..239, 240..243, 244..247
PFA = 248                   # packed P of slab 0: 248..255


def tup(base, n):
    return f"v[{base}:{base + n - 1}]"


def region(nch, s_out, q_base, s_in, pf, pre_pv=None, read_v=False, rsv=64):
    """One region: chain over `nch` K-dim chunks into s_out (B operands at q_base), exponentials of s_in -> packed into pf,
    row-sum partial into ACC.  pre_pv = (O, pf_prev): two PV MFMAs first (V fragments VF, B operand pf_prev)."""
    L = []
    nslots = nch + (2 if pre_pv else 0)
    # LDS reads up front: V fragments of this block (region 1 only), then the first three K fragments
    if read_v:
        for mm in range(2):
            L.append(f"ds_read_b64_tr_b16 {tup(VF + 4 * mm, 2)}, v{VADDR} offset:{(8 * mm) * rsv}")
            L.append(f"ds_read_b64_tr_b16 {tup(VF + 4 * mm + 2, 2)}, v{VADDR} offset:{(8 * mm + 4) * rsv}")
    npre = min(3, nch)
    for ch in range(npre):
        L.append(f"ds_read_b128 {tup(KT + 4 * ch, 4)}, v{KADDR} offset:{32 * ch}")
    # VALU stream: pair j -> exp, exp | (one slot later) pack, dot
    first_exp_slot = 1 if pre_pv else 0          # the scores were written by the previous region's last MFMA: keep >= 12 wait states
    pairs = list(range(8))

    def valu(slot):
        out = []
        j = slot - first_exp_slot
        if 0 <= j < 8:
            a = ET + 2 * (j % 2)
            if j == 0 and pre_pv:
                out.append("s_nop 3")         # s_in was written by the previous region's last MFMA: >= 12 wait states before a VALU read
            out.append(f"v_exp_f32 v{a}, v{s_in + 2 * j}")
            out.append(f"v_exp_f32 v{a + 1}, v{s_in + 2 * j + 1}")
        jj = j - 1
        if 0 <= jj < 8:
            a = ET + 2 * (jj % 2)
            out.append(f"v_cvt_pk_bf16_f32 v{pf + jj}, v{a}, v{a + 1}")
            out.append(f"v_dot2_f32_bf16 v{ACC}, v{pf + jj}, %[ones], " + ("0" if jj == 0 else f"v{ACC}"))     # (1, 1) bf16 from an SGPR
        return out

    slot = 0
    if pre_pv:
        o, pfp = pre_pv
        # V fragments (oldest LDS reads of the block) must have landed: at most the three K reads just issued stay outstanding
        L.append(f"s_waitcnt lgkmcnt({npre})")
        for mm in range(2):
            L.append(f"v_mfma_f32_32x32x16_bf16 {tup(o, 16)}, {tup(VF + 4 * mm, 4)}, {tup(pfp + 4 * mm, 4)}, {tup(o, 16)}")
            L += valu(slot)
            slot += 1
    for ch in range(nch):
        outstanding = min(npre - 1, nch - 1 - ch)              # younger K reads that may stay in flight
        L.append(f"s_waitcnt lgkmcnt({outstanding})")
        k = KT + 4 * (ch % 3)
        L.append(f"v_mfma_f32_32x32x16_bf16 {tup(s_out, 16)}, {tup(k, 4)}, {tup(q_base + 4 * ch, 4)}, " + ("0" if ch == 0 else tup(s_out, 16)))
        if ch + 3 < nch:
            L.append(f"ds_read_b128 {tup(k, 4)}, v{KADDR} offset:{32 * (ch + 3)}")
        L += valu(slot)
        slot += 1
    while slot - first_exp_slot - 1 < 8:                         # packs / sums still owed
        L += valu(slot)
        slot += 1
    # the row sum leaves in ACC: a DOT result needs 3 wait states before a DIFFERENT VALU instruction may read it (gfx940+ "dot
    # write -> different VALU read"; found the hard way: the compiler's v_add right behind the region saw the sum without its last pair)
    L.append("s_nop 2")
    return L


def pad_nops(lines):
    """debug (GEN_NOPS=1): a long s_nop behind every instruction, to tell hazards from logic errors"""
    if not os.environ.get("GEN_NOPS"):
        return lines
    out = []
    for ln in lines:
        out.append(ln)
        if not ln.startswith("s_"):
            out.append("s_nop 15")
    return out


def ablate(lines):
    """timing experiments (results wrong): GEN_ABLATE = comma list of nowait / novalu / nomfma / noread"""
    ab = set(filter(None, os.environ.get("GEN_ABLATE", "").split(",")))
    out = []
    for ln in lines:
        if "nowait" in ab and ln.startswith("s_waitcnt lgkmcnt") and ln != "s_waitcnt lgkmcnt(0)":
            continue
        if "novalu" in ab and ln.split()[0] in ("v_exp_f32", "v_cvt_pk_bf16_f32", "v_dot2_f32_bf16"):
            continue
        if "nomfma" in ab and ln.startswith("v_mfma"):
            continue
        if "noread" in ab and ln.startswith("ds_read_b128"):
            continue
        out.append(ln)
    return out


def stmt(name, lines, outs, ins, clob):
    """A complete `asm volatile` statement as a macro; operands are the kernel's variables pinned to their registers."""
    lines = ablate(lines)
    lines = pad_nops(lines)
    body = " \\\n".join('        "' + ln + '\\n"' for ln in lines)
    def op(c, v):
        if c.startswith("["):                                   # "[name]constraint"
            name, cons = c[1:].split("]")
            return f'[{name}] "{cons}"({v})'
        return f'"{c}"({v})'
    o = ", ".join(op(c, v) for c, v in outs)
    i = ", ".join(op(c, v) for c, v in ins)
    c = ", ".join(f'"{x}"' for x in clob)
    return f"#define {name}() asm volatile( \\\n{body} \\\n        : {o} \\\n        : {i} \\\n        : {c})\n"


def R(base, n=1):
    return f"{{v{base}}}" if n == 1 else f"{{v[{base}:{base + n - 1}]}}"


def main():
    out = ["// GENERATED by tools/gen_attn_dual_asm.py -- do not edit.  Register map and slot schedule: see the generator.\n",
           "// Variables the statements name (k_attn_dual): sA, sB, Oa, Ob (f32x16); qf[2][NCH], pfA0, pfA1, pfB0, pfB1, vf0, vf1 (u32x4);\n",
           "// accv (float); kaddr, vaddr, ones (unsigned).\n"]
    clob_common = [f"v{ET + k}" for k in range(4)] + [f"v{KT + k}" for k in range(12)]
    for c, nch in ((144, 9), (32, 2)):
        r1 = region(nch, SB, Q1, SA, PFA, read_v=True)
        r2 = region(nch, SA, Q0, SB, PFB, pre_pv=(O0, PFA))
        pv1 = ["s_nop 1"] + [f"v_mfma_f32_32x32x16_bf16 {tup(O1, 16)}, {tup(VF + 4 * mm, 4)}, {tup(PFB + 4 * mm, 4)}, {tup(O1, 16)}" for mm in range(2)]
        q1 = [(R(Q1 + 4 * ch, 4), f"qf[1][{ch}]") for ch in range(nch)]
        q0 = [(R(Q0 + 4 * ch, 4), f"qf[0][{ch}]") for ch in range(nch)]
        out.append(stmt(f"DA_DUAL_R1_C{c}", r1,
                        [("=" + R(SB, 16), "sB"), ("=" + R(PFA, 4), "pfA0"), ("=" + R(PFA + 4, 4), "pfA1"), ("=" + R(ACC), "accv"),
                         ("=" + R(VF, 4), "vf0"), ("=" + R(VF + 4, 4), "vf1")],
                        [(R(SA, 16), "sA"), (R(KADDR), "kaddr"), (R(VADDR), "vaddr"), ("[ones]s", "ones")] + q1, clob_common))
        out.append(stmt(f"DA_DUAL_R2_C{c}", r2,
                        [("=" + R(SA, 16), "sA"), ("=" + R(PFB, 4), "pfB0"), ("=" + R(PFB + 4, 4), "pfB1"), ("=" + R(ACC), "accv"), ("+" + R(O0, 16), "Oa")],
                        [(R(SB, 16), "sB"), (R(KADDR), "kaddr"), ("[ones]s", "ones"), (R(PFA, 4), "pfA0"), (R(PFA + 4, 4), "pfA1"),
                         (R(VF, 4), "vf0"), (R(VF + 4, 4), "vf1")] + q0, clob_common))
        out.append(stmt(f"DA_DUAL_PV1_C{c}", pv1, [("+" + R(O1, 16), "Ob")],
                        [(R(PFB, 4), "pfB0"), (R(PFB + 4, 4), "pfB1"), (R(VF, 4), "vf0"), (R(VF + 4, 4), "vf1")], []))
    path = os.path.join(ROOT, "diffassemble_amd", "csrc", "da_attn_dual_asm.inc")
    open(path, "w").write("".join(out))
    print("wrote", path, sum(len(x.splitlines()) for x in out), "lines")


if __name__ == "__main__":
    main()
