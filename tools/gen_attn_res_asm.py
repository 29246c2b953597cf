#!/usr/bin/env python
"""Generates diffassemble_amd/csrc/da_attn_res_asm.inc: the software-pipelined steady-state key blocks of k_attn_res<.., 256> (da_attn_opt.hip)
as ONE inline-asm statement per PAIR of 32-key blocks, on fixed physical registers ("the hand-scheduled block loop, by
generator").  Inside a wave the score product of block b + 1 (two MFMAs) is issued in front of the sixteen exponentials of block b, so the matrix
pipe works while the vector port does -- hipcc schedules the same source as "chain, wait, exponentials" (one wave's latency chain per block is what
bounds the kernel: profiles/r05/NOTES.md) and spills when asked to keep two score tiles alive.

Step(sCur, sNext, kfUse, kfLoad):   [sCur = scores of block b, kfUse = K fragments of block b + 1 (requested a step ago)]
    wait kfUse | sNext = kfUse . Q (2 MFMAs) | request V(b) (4 transposing reads) and kfLoad = K(b + 2) | 16 exp / 8 pack of sCur -> P | wait V |
    O += V . P (2 MFMAs) | 8 dots (row sum)
Pair = Step(sA, sB, kfA, kfB) ; Step(sB, sA, kfB, kfA) for blocks (b, b + 1), b even: V(b + 1) and K(b + 3) are V(b) / K(b + 2) + 2048 bytes.

    python tools/gen_attn_res_asm.py
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- register map (VGPR numbers < 128: sixteen waves per workgroup); everything else stays with the compiler
QF = 32                     # Q fragments: 32..35, 36..39
VF = 40                     # V fragments: v0 = 40..43 (vlo0, vhi0), v1 = 44..47 (vlo1, vhi1)
KA, KB = 48, 56             # K fragment sets: 48..51 | 52..55 and 56..59 | 60..63
SA, SB = 64, 80             # score tuples
O = 96                      # output accumulator 96..111
PF = 112                    # packed P: 112..119
ET = 120                    # exp temporaries 120..123
ACC = 124                   # row sum
KAD0, KAD1, VAD = 125, 126, 127   # LDS addresses: K fragments (chunk 0 / 1) of block b + 2, V fragments of block b


def tup(base, n):
    return f"v[{base}:{base + n - 1}]"


# pairs of scores in the order the C++ loop sums their packs: P registers 0, 4, 1, 5, 2, 6, 3, 7
ORDER = [0, 4, 1, 5, 2, 6, 3, 7]


def step(s_cur, s_next, kf_use, kf_load, blk_off, first):
    L = []
    # kfUse was requested a step ago (two reads, older than anything this step issues)
    L.append("s_waitcnt lgkmcnt(0)")
    L.append(f"v_mfma_f32_32x32x16_bf16 {tup(s_next, 16)}, {tup(kf_use, 4)}, {tup(QF, 4)}, 0")
    L.append(f"v_mfma_f32_32x32x16_bf16 {tup(s_next, 16)}, {tup(kf_use + 4, 4)}, {tup(QF + 4, 4)}, {tup(s_next, 16)}")
    for mm in range(2):
        L.append(f"ds_read_b64_tr_b16 {tup(VF + 4 * mm, 2)}, v{VAD} offset:{blk_off + (8 * mm) * 64}")
        L.append(f"ds_read_b64_tr_b16 {tup(VF + 4 * mm + 2, 2)}, v{VAD} offset:{blk_off + (8 * mm + 4) * 64}")
    L.append(f"ds_read_b128 {tup(kf_load, 4)}, v{KAD0} offset:{blk_off}")
    L.append(f"ds_read_b128 {tup(kf_load + 4, 4)}, v{KAD1} offset:{blk_off}")
    # VALU stream: pair j -> exp, exp; one slot later its pack
    for j in range(9):
        if j < 8:
            pi = ORDER[j]
            src = s_cur + (2 * pi if pi < 4 else 8 + 2 * (pi - 4))
            a = ET + 2 * (j % 2)
            L.append(f"v_exp_f32 v{a}, v{src}")
            L.append(f"v_exp_f32 v{a + 1}, v{src + 1}")
        if j >= 1:
            pj = ORDER[j - 1]
            a = ET + 2 * ((j - 1) % 2)
            L.append(f"v_cvt_pk_bf16_f32 v{PF + pj}, v{a}, v{a + 1}")
    # V fragments there (the two K reads behind them may stay in flight)
    L.append("s_waitcnt lgkmcnt(2)")
    L.append("s_nop 1")
    L.append(f"v_mfma_f32_32x32x16_bf16 {tup(O, 16)}, {tup(VF, 4)}, {tup(PF, 4)}, {tup(O, 16)}")
    L.append(f"v_mfma_f32_32x32x16_bf16 {tup(O, 16)}, {tup(VF + 4, 4)}, {tup(PF + 4, 4)}, {tup(O, 16)}")
    for pj in ORDER:
        L.append(f"v_dot2_f32_bf16 v{ACC}, v{PF + pj}, %[ones], v{ACC}")
    return L


def R(base, n=1):
    return f"{{v{base}}}" if n == 1 else f"{{v[{base}:{base + n - 1}]}}"


def ablate(lines, kind):
    """timing experiments (results wrong): nowait / novalu / nomfma / noread / halfexp"""
    out = []
    nexp = 0
    for ln in lines:
        opc = ln.split()[0]
        if kind == "nowait" and opc == "s_waitcnt":
            continue
        if kind == "novalu" and opc in ("v_exp_f32", "v_cvt_pk_bf16_f32", "v_dot2_f32_bf16"):
            continue
        if kind == "nomfma" and opc.startswith("v_mfma"):
            continue
        if kind == "noread" and opc.startswith("ds_read"):
            continue
        if kind == "halfexp" and opc == "v_exp_f32":
            nexp += 1
            if nexp % 2 == 0:
                continue
        out.append(ln)
    return out


def main():
    base = step(SA, SB, KA, KB, 0, True) + step(SB, SA, KB, KA, 2048, False)
    base.append("s_nop 2")          # a dot result needs 3 wait states before a different VALU instruction reads it
    outs = [("+" + R(SA, 16), "sA"), ("+" + R(KA, 4), "kfA0"), ("+" + R(KA + 4, 4), "kfA1"), ("+" + R(O, 16), "O"), ("+" + R(ACC), "ls")]
    ins = [(R(QF, 4), "qf[0]"), (R(QF + 4, 4), "qf[1]"), (R(KAD0), "kad0"), (R(KAD1), "kad1"), (R(VAD), "vad"), ("[ones]s", "ones")]
    clob = [f"v{r}" for r in list(range(VF, VF + 8)) + list(range(KB, KB + 8)) + list(range(SB, SB + 16)) + list(range(PF, PF + 8)) + list(range(ET, ET + 4))] + ["memory"]
    def op(c, v):
        if c.startswith("["):
            name, cons = c[1:].split("]")
            return f'[{name}] "{cons}"({v})'
        return f'"{c}"({v})'
    text = ["// GENERATED by tools/gen_attn_res_asm.py -- do not edit.  Register map, slot schedule and the hazards padded by hand: see the generator.\n",
            "// Variables the statement names (k_attn_res): sA, O (f32x16); kfA0, kfA1, qf[2] (u32x4); ls (float); kad0, kad1, vad, ones (unsigned).\n",
            "// On entry sA = scores of block b (even), kfA = K fragments of block b + 1; on exit sA = scores of block b + 2, kfA = K fragments of block\n",
            "// b + 3 STILL IN FLIGHT (the next statement, or an explicit s_waitcnt lgkmcnt(0), waits for them).\n",
            "// DA_RES_PAIR_<ablation>: timing experiments with WRONG results (DA_ATTN_RES_PIPE = 2 .. 6, experiments build).\n"]
    for name, kind in (("DA_RES_PAIR", None), ("DA_RES_PAIR_NOWAIT", "nowait"), ("DA_RES_PAIR_NOVALU", "novalu"), ("DA_RES_PAIR_NOMFMA", "nomfma"),
                       ("DA_RES_PAIR_NOREAD", "noread"), ("DA_RES_PAIR_HALFEXP", "halfexp")):
        lines = ablate(base, kind) if kind else base
        body = " \\\n".join('        "' + ln + '\\n"' for ln in lines)
        text.append(f"#define {name}() asm volatile( \\\n" + body + " \\\n        : " + ", ".join(op(c, v) for c, v in outs) + " \\\n        : " +
                    ", ".join(op(c, v) for c, v in ins) + " \\\n        : " + ", ".join(f'"{x}"' for x in clob) + ")\n")
    path = os.path.join(ROOT, "diffassemble_amd", "csrc", "da_attn_res_asm.inc")
    open(path, "w").write("".join(text))
    print("wrote", path, len(base), "instructions")


if __name__ == "__main__":
    main()
