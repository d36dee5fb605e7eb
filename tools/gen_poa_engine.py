#!/usr/bin/env python3
"""Generator of elector_amd/csrc/poa_engine_gen.h: the loops of k_poa's two dynamic programs as inline-asm statements,
one set per geometry class (G lanes per window pair, R rows per lane).

Why generated assembly: with the recurrence as C++ around per-row asm blocks the compiler moved the column arrays
between registers from step to step (12-36 v_mov per step, whichever way the operands were declared) and paid for every
address, border and role swap in 4-cycle instructions.  Here the arrays live in FIXED registers (the statements' operands
are register tuples pinned by constraint, "+{v[a:b]}"), the two steps of a pair swap the arrays' roles by name, and the
step's bookkeeping is written out: a step of alignment #2's plain-chain form is 12 + 13 (R - 1) row instructions plus 13.

Three statements per class:
  Dp2Engine<G, R>::run        alignment #2, steps G + 1 .. up to the first step in which a window's last row can meet a
                              final node; four forms of the step (plain chain / predecessor two back / second
                              predecessor / far virtual start)
  Dp2Engine<G, R>::run_first  the same for steps 1 .. G: the rows run under the mask of the lanes that have reached their
                              first column
  Dp1Engine<G, R>::run        alignment #1 (linear x linear), steps G .. up to the step in which a window's score appears
The C++ around them (poa_pack.hip) runs the other steps -- the closing ones with their tests, and any step whose node
records ask for a form the statements have not got (second predecessor AND far virtual start together; the far-edge
instance): a statement leaves at such a step with its state in the agreed registers and is entered again behind it.

Hazards kept by hand (the compiler does not look into the statements): two wait states between a VALU write of a register
and a DPP instruction that reads it (source or old value); s_waitcnt before LDS results are used and before leaving.

Usage: python3 tools/gen_poa_engine.py  (rewrites the header; --check: exit 1 if the header is not up to date)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "elector_amd", "csrc", "poa_engine_gen.h")

NT = 12          # vector temporaries
NST1 = 8         # pinned state a[0..7]
NST2 = 2         # pinned state b[0..1]
NCN = 8          # pinned per-lane constants: xea xeb cs0 cab orda ordb g kext


def stride_of(R):
    """registers between the starts of two arrays: tuples of this chip start at even registers"""
    return R + (R & 1)


def base_of(R):
    """first pinned register (even): the block ends at the top of the instances' budget (168 registers: k_poa's launch
    bounds, three wavefronts per SIMD)"""
    top = 168
    return (top - (5 * stride_of(R) + NST1 + NST2 + NCN + NT)) & ~1


class Regs:
    def __init__(self, R):
        B = base_of(R)
        Q = stride_of(R)
        self.B = B
        self.R = R
        self.YL = ["v%d" % (B + k) for k in range(R)]
        self.S = [["v%d" % (B + Q + k) for k in range(R)], ["v%d" % (B + 3 * Q + k) for k in range(R)]]
        self.E = [["v%d" % (B + 2 * Q + k) for k in range(R)], ["v%d" % (B + 4 * Q + k) for k in range(R)]]
        s = B + 5 * Q
        self.A = ["v%d" % (s + i) for i in range(8)]            # state a[0..7]
        self.X = [(self.A[0], self.A[1]), (self.A[2], self.A[3])]
        self.OA, self.OB = self.A[4], self.A[5]
        self.U = [self.A[6], self.A[7]]
        self.EE = ["v%d" % (s + 8), "v%d" % (s + 9)]            # state b[0..1]
        c = s + 10
        self.XEA, self.XEB, self.CS0, self.CAB, self.ORDA, self.ORDB, self.GV, self.KEXT = ["v%d" % (c + i) for i in range(8)]
        self.T = ["v%d" % (c + NCN + i) for i in range(NT)]
        self.ranges = dict(YL=(B, B + R - 1), SA=(B + Q, B + Q + R - 1), EA=(B + 2 * Q, B + 2 * Q + R - 1), SB=(B + 3 * Q, B + 3 * Q + R - 1),
                           EB=(B + 4 * Q, B + 4 * Q + R - 1), ST1=(s, s + 7), ST2=(s + 8, s + 9), CN=(c, c + NCN - 1))
        self.temps = (c + NCN, c + NCN + NT - 1)


def shift_dpp(G):
    if G == 8:
        return "row_shr:2 row_mask:0xf bank_mask:0xf"
    if G == 16:
        return "row_shr:1 row_mask:0xf bank_mask:0xf"
    return "wave_shr:1 row_mask:0xf bank_mask:0xf"


def sub32(o, d, a, b):
    """d = a - b per 16-bit half as ONE 32-bit v_sub_u32 (the 2-cycle class, profiles/r05_valu_rate2.txt; a packed subtraction
    is 4 cycles): exact when no half borrows.  Every score in the loops is kept one BELOW its value (poa_pack.hip: the
    borders start at -1), so every half is negative -- as unsigned numbers they are ordered like the scores -- and the
    subtractions below have a >= b per half by construction (a maximum less one of its arguments, a score less a small
    positive constant).  All operands VGPRs: with a scalar operand the instruction falls back to 4 cycles."""
    o.append("v_sub_u32 %s, %s, %s" % (d, a, b))


def core(o, xl, yl, ix, iy, dm, sn, en, mv, sh, t0, mx, kext):
    """the affine-gap recurrence of one row (two cells): 12 instructions for the step's first row, 13 otherwise.
    ix may be mx itself (a form that has built the x-gap offer there); en may be ix (alignment #1 keeps E in place)."""
    o.append("v_xor_b32 %s, %s, %s" % (t0, xl, yl))
    o.append("v_pk_max_i16 %s, %s, %s" % (mx, ix, iy))
    o.append("v_pk_min_u16 %s, %s, %%[one]" % (t0, t0))
    o.append("v_pk_mad_i16 %s, %s, %%[ksub], %s" % (t0, t0, dm))
    o.append("v_pk_max_i16 %s, %s, %s" % (sn, t0, mx))
    sub32(o, t0, sn, mx)
    sub32(o, mx, mx, iy)
    sub32(o, en, sn, kext)
    o.append("v_pk_min_u16 %s, %s, %%[one]" % (t0, t0))
    o.append("v_pk_min_u16 %s, %s, %%[one]" % (mx, mx))
    o.append("v_pk_mad_i16 %s, %s, %%[kdelta], %s" % (en, t0, en))
    if sh == 0:
        o.append("v_lshl_or_b32 %s, %s, 1, %s" % (mv, t0, mx))
    else:
        o.append("v_lshl_or_b32 %s, %s, 1, %s" % (mx, t0, mx))
        o.append("v_lshl_or_b32 %s, %s, %d, %s" % (mv, mx, sh, mv))


def dpp_shift(o, G, dst, src, tmp):
    """dst holds the border on entry (lanes g == 0 keep it), the lane above's src on exit.  The caller has put at least
    two instructions between the last VALU write of dst / src and this."""
    if G == 32:
        o.append("v_mov_b32 %s, %s" % (tmp, dst))
        o.append("s_nop 1")
    o.append("v_mov_b32_dpp %s, %s %s" % (dst, src, shift_dpp(G)))
    if G == 32:
        o.append("v_cndmask_b32 %s, %s, %s, %%[g0]" % (dst, dst, tmp))


def masks(o, x, bit, dst, tmp):
    """per-half mask from bit `bit` of the two windows' records"""
    o.append("v_bfe_i32 %s, %s, %d, 1" % (dst, x[0], bit))
    o.append("v_bfe_i32 %s, %s, %d, 1" % (tmp, x[1], bit))
    o.append("v_bfi_b32 %s, %%[k16], %s, %s" % (dst, dst, tmp))


def dp2_step(o, rg, G, role, masked):
    """one anti-diagonal of alignment #2.  role 0: reads (S1, E1), writes (S2, E2), records in X0, next records to X1,
    BR1 in U0 (becomes up1), up2 in U1 (receives the next BR1), BE1 in E0, BE2 in E1 (receives upE: lane g = 0 keeps the
    new BE1).  role 1: everything swapped.  masked: steps 1 .. G -- the rows run under %[sm], the lanes with g < step; the
    record offsets advance for the lanes with g <= step (the next step's lanes, %[sn])."""
    R = rg.R
    cS, cE, nS, nE = rg.S[role], rg.E[role], rg.S[1 - role], rg.E[1 - role]
    Xc, Xn = rg.X[role], rg.X[1 - role]
    Uc, Uo = rg.U[role], rg.U[1 - role]
    Ec, Eo = rg.EE[role], rg.EE[1 - role]
    T = rg.T
    t0, XL, mx, MV, M1, M2, DA, DB, e1, SEC, d1, TX = T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7], T[8], T[9], T[10], T[11]
    L = lambda s: "%s%d_%%=" % (s, role)
    rows_on = (lambda: o.append("s_mov_b64 exec, %[sm]")) if masked else (lambda: None)
    o.append("%s:" % L("step"))
    o.append("s_waitcnt lgkmcnt(0)")
    if masked:
        o.append("s_add_i32 %%[st], %%[t], %d" % (1 + role))
        o.append("v_cmp_gt_u32 vcc, %%[st], %s" % rg.GV)
        o.append("s_mov_b64 %[sn], vcc")
        o.append("s_mov_b64 exec, vcc")
    o.append("v_add_u32 %s, 4, %s" % (rg.OA, rg.OA))
    o.append("v_add_u32 %s, 4, %s" % (rg.OB, rg.OB))
    if masked:
        o.append("s_mov_b64 exec, -1")
    o.append("v_or_b32 %s, %s, %s" % (t0, Xc[0], Xc[1]))
    o.append("v_min_u32 %s, %s, %s" % (rg.OA, rg.OA, rg.XEA))
    o.append("v_min_u32 %s, %s, %s" % (rg.OB, rg.OB, rg.XEB))
    o.append("v_and_b32 %s, 15, %s" % (mx, t0))
    o.append("ds_read_b32 %s, %s" % (Xn[0], rg.OA))
    o.append("ds_read_b32 %s, %s" % (Xn[1], rg.OB))
    o.append("v_cmp_ne_u32 vcc, 0, %s" % mx)
    o.append("v_perm_b32 %s, %s, %s, %%[psel]" % (XL, Xc[1], Xc[0]))
    o.append("s_cbranch_vccnz %s" % L("notnear"))
    # ---- NEAR
    dpp_shift(o, G, Uc, nS[R - 1], TX)                       # up1 over BR1
    sub32(o, Eo, Ec, rg.KEXT)                                # BEj = BE1 - ext (BRj = BE1)
    o.append("v_mov_b32 %s, %s" % (Uo, Ec))                  # the next step's BR1
    o.append("s_nop 0")
    dpp_shift(o, G, Eo, cE[R - 1], TX)                       # upE; lane g = 0 keeps BEj
    rows_on()
    for k in range(R):
        core(o, XL, rg.YL[k], cE[k], Eo if k == 0 else nE[k - 1], Uc if k == 0 else cS[k - 1], nS[k], nE[k], MV, 2 * k, t0, mx, rg.KEXT)
    o.append("s_branch %s" % L("store"))
    # ---- PLAIN
    o.append("%s:" % L("notnear"))
    o.append("v_and_b32 %s, 12, %s" % (mx, t0))
    o.append("v_cmp_ne_u32 vcc, 0, %s" % mx)
    o.append("s_cbranch_vccnz %s" % L("notplain"))
    masks(o, Xc, 0, M1, M2)
    dpp_shift(o, G, Uc, nS[R - 1], TX)
    o.append("v_bfi_b32 %s, %s, %s, %s" % (DA, M1, Uo, Uc))              # diagonal of the first row: up2 or up1
    o.append("v_bfi_b32 %s, %s, %s, %s" % (Uo, M1, Eo, Ec))              # BRj: what row -1 of the predecessor column offers
    sub32(o, Eo, Uo, rg.KEXT)
    o.append("s_nop 1")
    dpp_shift(o, G, Eo, cE[R - 1], TX)
    rows_on()
    for k in range(R):
        dm, dn = (DA, DB) if k % 2 == 0 else (DB, DA)
        o.append("v_bfi_b32 %s, %s, %s, %s" % (dn, M1, nS[k], cS[k]))     # the predecessor's cell: the next row's diagonal
        o.append("v_bfi_b32 %s, %s, %s, %s" % (mx, M1, nE[k], cE[k]))     # ... and its x-gap offer
        core(o, XL, rg.YL[k], mx, Eo if k == 0 else nE[k - 1], dm, nS[k], nE[k], MV, 2 * k, t0, mx, rg.KEXT)
    o.append("s_branch %s" % L("store"))
    # ---- TWO
    o.append("%s:" % L("notplain"))
    o.append("v_and_b32 %s, 8, %s" % (mx, t0))
    o.append("v_cmp_ne_u32 vcc, 0, %s" % mx)
    o.append("s_cbranch_vccnz %s" % L("virt"))
    masks(o, Xc, 0, M1, e1)
    masks(o, Xc, 1, M2, e1)
    dpp_shift(o, G, Uc, nS[R - 1], TX)
    dt, dm = DA, DB
    o.append("v_bfi_b32 %s, %s, %s, %s" % (dt, M1, Uo, Uc))
    o.append("v_bfi_b32 %s, %s, %s, %s" % (dm, M2, Uo, Uc))
    o.append("v_pk_max_i16 %s, %s, %s" % (dm, dt, dm))
    o.append("v_bfi_b32 %s, %s, %s, %s" % (e1, M1, Eo, Ec))
    o.append("v_bfi_b32 %s, %s, %s, %s" % (Uo, M2, Eo, Ec))
    o.append("v_pk_max_i16 %s, %s, %s" % (Uo, e1, Uo))                    # BRj
    sub32(o, Eo, Uo, rg.KEXT)
    o.append("v_mov_b32 %s, 0" % SEC)
    o.append("v_mov_b32 %s, 0" % MV)
    dpp_shift(o, G, Eo, cE[R - 1], TX)
    rows_on()
    for k in range(R):
        iy = Eo if k == 0 else nE[k - 1]
        o.append("v_xor_b32 %s, %s, %s" % (t0, XL, rg.YL[k]))
        o.append("v_pk_min_u16 %s, %s, %%[one]" % (t0, t0))
        sub32(o, d1, dm, dt)
        o.append("v_pk_mad_i16 %s, %s, %%[ksub], %s" % (t0, t0, dm))
        o.append("v_bfi_b32 %s, %s, %s, %s" % (dt, M1, nS[k], cS[k]))
        o.append("v_bfi_b32 %s, %s, %s, %s" % (dm, M2, nS[k], cS[k]))
        o.append("v_pk_max_i16 %s, %s, %s" % (dm, dt, dm))
        o.append("v_bfi_b32 %s, %s, %s, %s" % (e1, M1, nE[k], cE[k]))
        o.append("v_bfi_b32 %s, %s, %s, %s" % (mx, M2, nE[k], cE[k]))
        o.append("v_pk_max_i16 %s, %s, %s" % (mx, e1, mx))
        sub32(o, e1, mx, e1)
        o.append("v_pk_max_i16 %s, %s, %s" % (mx, mx, iy))
        o.append("v_pk_max_i16 %s, %s, %s" % (nS[k], t0, mx))
        sub32(o, t0, nS[k], mx)
        sub32(o, mx, mx, iy)
        sub32(o, nE[k], nS[k], rg.KEXT)
        o.append("v_pk_min_u16 %s, %s, %%[one]" % (t0, t0))
        o.append("v_pk_min_u16 %s, %s, %%[one]" % (mx, mx))
        o.append("v_pk_mad_i16 %s, %s, %%[kdelta], %s" % (nE[k], t0, nE[k]))
        o.append("v_lshl_or_b32 %s, %s, 1, %s" % (mx, t0, mx))
        o.append("v_lshl_or_b32 %s, %s, %d, %s" % (MV, mx, 2 * k, MV))
        o.append("v_pk_sub_i16 %s, 0, %s" % (t0, t0))
        o.append("v_bfi_b32 %s, %s, %s, %s" % (e1, t0, d1, e1))
        o.append("v_pk_min_u16 %s, %s, %%[one]" % (e1, e1))
        o.append("v_lshl_or_b32 %s, %s, %d, %s" % (SEC, e1, k, SEC))
    # which predecessor the cells of a two-predecessor node took: one byte per lane in the node's row of ordinal bytes
    lg = {8: 3, 16: 4, 32: 5, 64: 6}[G]
    for h, (ordo, wr) in enumerate(((rg.ORDA, "ds_write_b8"), (rg.ORDB, "ds_write_b8_d16_hi"))):
        o.append("v_and_b32 %s, 4, %s" % (t0, Xc[h]))
        o.append("v_cmp_ne_u32 vcc, 0, %s" % t0)
        o.append("s_and_saveexec_b64 %[sx], vcc")
        o.append("v_lshrrev_b32 %s, 24, %s" % (t0, Xc[h]))
        o.append("v_lshl_add_u32 %s, %s, %d, %s" % (t0, t0, lg, ordo))
        o.append("%s %s, %s" % (wr, t0, SEC))
        o.append("s_mov_b64 exec, %[sx]")
    o.append("s_branch %s" % L("store"))
    # ---- VIRT1
    o.append("%s:" % L("virt"))
    o.append("v_and_b32 %s, 4, %s" % (mx, t0))
    o.append("v_cmp_ne_u32 vcc, 0, %s" % mx)
    o.append("s_cbranch_vccnz %s" % L("event"))
    V1 = M2
    masks(o, Xc, 0, M1, e1)
    masks(o, Xc, 3, V1, e1)
    dpp_shift(o, G, Uc, nS[R - 1], TX)
    o.append("v_bfi_b32 %s, %s, %s, %s" % (DA, M1, Uo, Uc))
    o.append("v_bfi_b32 %s, %s, %s, %s" % (DA, V1, rg.CAB, DA))
    o.append("v_bfi_b32 %s, %s, %s, %s" % (Uo, M1, Eo, Ec))
    o.append("v_bfi_b32 %s, %s, %%[kopen], %s" % (Uo, V1, Uo))
    sub32(o, Eo, Uo, rg.KEXT)
    o.append("v_mov_b32 %s, %s" % (SEC, rg.CS0))                          # column 0 at the lane's first row
    o.append("s_nop 0")
    dpp_shift(o, G, Eo, cE[R - 1], TX)
    rows_on()
    for k in range(R):
        dm, dn = (DA, DB) if k % 2 == 0 else (DB, DA)
        vc, ve = (SEC, d1) if k % 2 == 0 else (d1, SEC)
        sub32(o, ve, vc, rg.KEXT)                                         # column 0 one row down = what this row's offers a gap
        o.append("v_bfi_b32 %s, %s, %s, %s" % (dn, M1, nS[k], cS[k]))
        o.append("v_bfi_b32 %s, %s, %s, %s" % (mx, M1, nE[k], cE[k]))
        o.append("v_bfi_b32 %s, %s, %s, %s" % (dn, V1, vc, dn))
        o.append("v_bfi_b32 %s, %s, %s, %s" % (mx, V1, ve, mx))
        core(o, XL, rg.YL[k], mx, Eo if k == 0 else nE[k - 1], dm, nS[k], nE[k], MV, 2 * k, t0, mx, rg.KEXT)
    # ---- the step's moves (every lane stores: a word of a lane that has no cell yet or none any more is never read), loop
    o.append("%s:" % L("store"))
    if masked:
        o.append("s_mov_b64 exec, -1")
        o.append("s_mov_b64 %[sm], %[sn]")
    o.append("global_store_dword %%[loff], %s, %%[mvp]%s" % (MV, " offset:256" if role else ""))
    if role == 1:
        o.append("v_add_u32 %[loff], 0x200, %[loff]")
        o.append("s_add_i32 %[t], %[t], 2")
        o.append("s_cmp_lt_i32 %[t], %[tend]")
        o.append("s_cbranch_scc1 step0_%=")
        o.append("s_branch done_%=")


def dp2_engine(G, R, masked):
    rg = Regs(R)
    o = []
    if masked:
        o.append("v_cmp_gt_u32 vcc, %%[t], %s" % rg.GV)                   # the lanes of the first step
        o.append("s_mov_b64 %[sm], vcc")
    dp2_step(o, rg, G, 0, masked)
    dp2_step(o, rg, G, 1, masked)
    # a step the statement has no form for: leave with the state as the C++ step expects it at that step
    o.append("event0_%=:")
    o.append("s_branch done_%=")
    o.append("event1_%=:")                                               # behind the first step of a pair: roles back in place
    X, U, E, T = rg.X, rg.U, rg.EE, rg.T
    o.append("s_waitcnt lgkmcnt(0)")
    o.append("v_mov_b32 %s, %s" % (X[0][0], X[1][0]))
    o.append("v_mov_b32 %s, %s" % (X[0][1], X[1][1]))
    o.append("v_mov_b32 %s, %s" % (T[0], U[0]))
    o.append("v_mov_b32 %s, %s" % (U[0], U[1]))
    o.append("v_mov_b32 %s, %s" % (U[1], T[0]))
    o.append("v_mov_b32 %s, %s" % (T[0], E[0]))
    o.append("v_mov_b32 %s, %s" % (E[0], E[1]))
    o.append("v_mov_b32 %s, %s" % (E[1], T[0]))
    o.append("s_add_i32 %[t], %[t], 1")
    o.append("done_%=:")
    o.append("s_waitcnt lgkmcnt(0)")
    return rg, o


def dp1_step(o, rg, G, role):
    """one anti-diagonal of alignment #1 (linear x linear): the lane's column in S[role], written to S[1 - role]; E in
    place.  a[0], a[1]: the two windows' reference letters of this step (fetched by the step before), a[4], a[5]: their LDS
    addresses; a[6]: the row above on the diagonal; b[role]: row -1 at this column in the group's first lane."""
    R = rg.R
    cS, nS, E = rg.S[role], rg.S[1 - role], rg.E[0]
    XA, XB = rg.A[0], rg.A[1]
    P = rg.A[6]
    KX = rg.A[7]                                             # the extension penalty, per half, in a VGPR (sub32)
    Qc, Qn = rg.EE[role], rg.EE[1 - role]
    T = rg.T
    t0, XL, mx, MV, TX = T[0], T[1], T[2], T[3], T[11]
    o.append("step%d_%%=:" % role)
    o.append("s_waitcnt lgkmcnt(0)")
    o.append("v_lshl_or_b32 %s, %s, 16, %s" % (XL, XB, XA))
    o.append("v_add_u32 %s, 1, %s" % (rg.OA, rg.OA))
    o.append("v_add_u32 %s, 1, %s" % (rg.OB, rg.OB))
    dpp_shift(o, G, Qc, cS[R - 1], TX)                       # the row above at this column (next step's diagonal); lane g = 0: row -1
    o.append("ds_read_u8 %s, %s" % (XA, rg.OA))
    o.append("ds_read_u8 %s, %s" % (XB, rg.OB))
    sub32(o, Qn, Qc, KX)                                     # what it offers a y-gap; lane g = 0: row -1 one column on
    o.append("s_nop 1")
    dpp_shift(o, G, Qn, E[R - 1], TX)
    for k in range(R):
        core(o, XL, rg.YL[k], E[k], Qn if k == 0 else E[k - 1], P if k == 0 else cS[k - 1], nS[k], E[k], MV, 2 * k, t0, mx, KX)
    o.append("v_mov_b32 %s, %s" % (P, Qc))
    o.append("global_store_dword %%[loff], %s, %%[mvp]%s" % (MV, " offset:256" if role else ""))
    if role == 1:
        o.append("v_add_u32 %[loff], 0x200, %[loff]")
        o.append("s_add_i32 %[t], %[t], 2")
        o.append("s_cmp_lt_i32 %[t], %[tend]")
        o.append("s_cbranch_scc1 step0_%=")


def dp1_engine(G, R):
    rg = Regs(R)
    o = []
    dp1_step(o, rg, G, 0)
    dp1_step(o, rg, G, 1)
    o.append("s_waitcnt lgkmcnt(0)")
    return rg, o


def quote(o):
    return "\n".join('      "%s\\n\\t"' % ln for ln in o)


def emit(G, R):
    rg, o2 = dp2_engine(G, R, False)
    _, o2f = dp2_engine(G, R, True)
    _, o1 = dp1_engine(G, R)
    r = rg.ranges
    pin = lambda n: "{v[%d:%d]}" % r[n]
    clob = ", ".join('"v%d"' % i for i in range(rg.temps[0], rg.temps[1] + 1))
    g0 = ', [g0] "s"(g0mask)' if G == 32 else ""
    d = dict(G=G, R=R, body2=quote(o2), body2f=quote(o2f), body1=quote(o1), YL=pin("YL"), SA=pin("SA"), EA=pin("EA"), SB=pin("SB"),
             EB=pin("EB"), ST1=pin("ST1"), ST2=pin("ST2"), CN=pin("CN"), clob=clob, g0=g0)
    return """template <>
struct Dp2Engine<%(G)d, %(R)d> {
  static constexpr bool kHave = true;
  typedef uint32_t VR __attribute__((ext_vector_type(%(R)d)));
  // Steps t .. tend - 1 of alignment #2 (t odd, tend - t even) or up to a step it has no form for: t is that step on return.
  static __device__ __forceinline__ void run(VR &YL, VR &S1, VR &E1, VR &S2, VR &E2, EngState &st, EngLane &cn, const EngConsts &c, int &t, int tend,
                                             uint32_t &loff, const uint32_t *mv, unsigned long long g0mask)
  {
    (void)g0mask;
    unsigned long long sx;
    asm volatile(
%(body2)s
      : "+%(YL)s"(YL), "+%(SA)s"(S1), "+%(EA)s"(E1), "+%(SB)s"(S2), "+%(EB)s"(E2), "+%(ST1)s"(st.a), "+%(ST2)s"(st.b), "+%(CN)s"(cn.v),
        [t] "+s"(t), [loff] "+v"(loff), [sx] "=&s"(sx)
      : [one] "s"(c.one), [ksub] "s"(c.ksub), [kext] "s"(c.kext), [kdelta] "s"(c.kdelta), [kopen] "s"(c.kopen), [k16] "s"(c.k16),
        [psel] "s"(c.psel), [tend] "s"(tend), [mvp] "s"(mv)%(g0)s
      : "vcc", "scc", "memory", %(clob)s);
  }
  // The same for the steps in which some lane has not reached its first column (t odd, <= G; tend = G + 1): the rows run
  // under the mask of the lanes with g < t.
  static __device__ __forceinline__ void run_first(VR &YL, VR &S1, VR &E1, VR &S2, VR &E2, EngState &st, EngLane &cn, const EngConsts &c, int &t,
                                                   int tend, uint32_t &loff, const uint32_t *mv, unsigned long long g0mask)
  {
    (void)g0mask;
    unsigned long long sx, sm, sn;
    int stmp;
    asm volatile(
%(body2f)s
      : "+%(YL)s"(YL), "+%(SA)s"(S1), "+%(EA)s"(E1), "+%(SB)s"(S2), "+%(EB)s"(E2), "+%(ST1)s"(st.a), "+%(ST2)s"(st.b), "+%(CN)s"(cn.v),
        [t] "+s"(t), [loff] "+v"(loff), [sx] "=&s"(sx), [sm] "=&s"(sm), [sn] "=&s"(sn), [st] "=&s"(stmp)
      : [one] "s"(c.one), [ksub] "s"(c.ksub), [kext] "s"(c.kext), [kdelta] "s"(c.kdelta), [kopen] "s"(c.kopen), [k16] "s"(c.k16),
        [psel] "s"(c.psel), [tend] "s"(tend), [mvp] "s"(mv)%(g0)s
      : "vcc", "scc", "memory", %(clob)s);
  }
};
template <>
struct Dp1Engine<%(G)d, %(R)d> {
  static constexpr bool kHave = true;
  typedef uint32_t VR __attribute__((ext_vector_type(%(R)d)));
  // Steps t .. tend - 1 of alignment #1 (tend - t even and positive; every lane past its first column).
  static __device__ __forceinline__ void run(VR &YL, VR &S, VR &E, VR &S2, EngState &st, const EngConsts &c, int &t, int tend, uint32_t &loff,
                                             const uint32_t *mv, unsigned long long g0mask)
  {
    (void)g0mask;
    asm volatile(
%(body1)s
      : "+%(YL)s"(YL), "+%(SA)s"(S), "+%(EA)s"(E), "+%(SB)s"(S2), "+%(ST1)s"(st.a), "+%(ST2)s"(st.b), [t] "+s"(t), [loff] "+v"(loff)
      : [one] "s"(c.one), [ksub] "s"(c.ksub), [kext] "s"(c.kext), [kdelta] "s"(c.kdelta), [tend] "s"(tend), [mvp] "s"(mv)%(g0)s
      : "vcc", "scc", "memory", %(clob)s);
  }
};
""" % d


CLASSES = [(8, 4), (8, 5), (8, 6), (8, 7), (8, 8), (16, 5), (16, 6), (16, 7), (16, 8), (32, 5), (32, 6), (32, 7), (32, 8),
           (64, 5), (64, 6), (64, 7), (64, 8)]


def generate():
    out = ["// GENERATED by tools/gen_poa_engine.py -- do not edit; see the generator for what this is and why.",
           "// The loops of k_poa's two dynamic programs as inline-asm statements per geometry class, the column arrays pinned to",
           "// fixed registers.  Included by poa_pack.hip inside namespace elector.",
           "#pragma once",
           "",
           "// Alignment #2 -- a: records of the step to run (a[0], a[1]), scratch pair (a[2], a[3]), running record offsets (a[4],",
           "// a[5]), row -1's score at the column before (a[6]; becomes the shifted row above), the row above two columns back (a[7]);",
           "// b: what row -1 offers a gap at the two columns before (b[0]: one back, b[1]: two back).",
           "// Alignment #1 -- a[0], a[1]: the reference letters of the step to run, a[4], a[5]: their LDS addresses, a[6]: the row above",
           "// on the diagonal, a[7]: the extension penalty per half; b[0]: row -1 at the column of the step (in the group's first lane).",
           "struct EngState {",
           "  uint32_t a __attribute__((ext_vector_type(8)));",
           "  uint32_t b __attribute__((ext_vector_type(2)));",
           "};",
           "// per-lane constants of alignment #2: LDS offsets of the two upper guard records, column 0 at the lane's first row and at",
           "// the row above it, LDS offsets of the two windows' ordinal bytes (+ g), g, the extension penalty per half",
           "struct EngLane { uint32_t v __attribute__((ext_vector_type(8))); };",
           "struct EngConsts { uint32_t one, ksub, kext, kdelta, kopen, k16, psel; };",
           "",
           "template <int G, int R>",
           "struct Dp2Engine {",
           "  static constexpr bool kHave = false;",
           "};",
           "template <int G, int R>",
           "struct Dp1Engine {",
           "  static constexpr bool kHave = false;",
           "};",
           ""]
    for G, R in CLASSES:
        out.append(emit(G, R))
    return "\n".join(out)


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, len(text.split("\n")), "lines")
