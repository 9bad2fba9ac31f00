#!/usr/bin/env python3
"""AIR specification generator for the three SIPP STARKs (G1 exp, G2 exp, Fq12 exp).

The reference delegates these AIRs to starky-bn254 @ 2d46f9e (reference src/verifier_circuit.rs:133-135;
Cargo.toml:26), whose source is NOT under /root/reference.  PARITY UNPINNED: the column layout below is this
repository's own specification, written to follow what SURVEY.md App. A.9 records about upstream
(16 x 16-bit limbs, schoolbook products with quotient + carry witnesses, u16 range checks through
permuted-column lookups, LSB-first double-and-add with 2 rows per exponent bit = 512 rows per IO,
semantics out = offset + [exp] x  /  out = offset * x^exp from src/verifier_circuit.rs:92-124).

An AIR is emitted as DATA (a flat int64 "program" + a header) consumed by two independent interpreters:
oracle/air.c (CPU restatement: witness fill, constraint evaluation for prover and verifier) and
sipp_amd/csrc/air.hpp (HIP: witness fill and quotient kernels).  Both copies of the tables are written
by this script: data/air_tables.h (one file for both).

Column model
  main columns  = [TABLE] + unchecked cells + checked cells (range-checked against the TABLE column)
  trace columns = main | perm_in[n_checked] | perm_tab[n_checked]        (Halo2-style permuted lookup)
  Z columns     = one per checked column per challenge:  Z' (pi+g)(pt+g) = Z (c+g)(table+g)
  aux columns   = public-input polynomials (NOT committed; both sides derive them from the public inputs)
  mode "u16": table = 0..65535 (needs N >= 2^16); a 16-bit limb is ONE checked cell
  mode "u8" : table = 0..255; a 16-bit limb is TWO checked cells (lo, hi), value lo + 256*hi

Program encoding (int64 words), see `Prog` below:
  VEC   := n_limbs, n_terms, (coef, base, stride, flag_per, flag_neg)*     limb_i = sum coef*F*cell[base+i*stride]
           F = 1 | per[flag_per] | 1 - per[flag_per]        (flag_per = -1: none)
  GADGET:= OP_GADGET, sign_col, carry_base, carry_limbs, carry_bits, carry_offset, group, VEC(q),
           n_prod, (coef, VEC a, VEC b)*, n_lin, (coef, VEC a)*
           with d_k = e_k - (1-2s)(q*p)_k (k = 0..31) and group g in {1, 2}: the 32/g constraints
               sum_{t<g} 2^(16 t) d_{g m + t} - c_{m-1} + 2^(16 g) c_m = 0      (c_{-1} = c_{32/g - 1} = 0)
           i.e. the limb identity is checked in base 2^(16 g); g = 2 halves the carry cells.  Plus s(s-1) = 0.
  POLY  := OP_POLY, n_mono, (coef, n_factors, (kind, index)*)*             kind: 0 local 1 next 2 aux 3 periodic
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pairing_sched as PS  # noqa: E402  (the row schedule of the final-pairing AIR)

BN_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
NL = 16          # 16-bit limbs per Fq element
NQ = 17          # quotient limbs
ROWS_PER_IO = 512
OP_GADGET, OP_POLY = 1, 2
K_LOCAL, K_NEXT, K_AUX, K_PER = 0, 1, 2, 3

P_LIMBS = [(BN_P >> (16 * i)) & 0xFFFF for i in range(NL)]

# periodic functions: index -> (m, r0): 1 on rows r = r0 (mod m), 0 on the other rows
PER_FIRST, PER_LAST, PER_ADD, PER_LIMB_END = 0, 1, 2, 3
PERIODICS = [(ROWS_PER_IO, 0), (ROWS_PER_IO, ROWS_PER_IO - 1), (2, 0), (64, 63)]
PER_MAP0 = len(PERIODICS)           # MapToG2: PER_MAP0 + t = 1 on rows = t (mod 8), t = 0 .. 7
PERIODICS += [(8, t) for t in range(8)]


class Air:
    def __init__(self, name, mode):
        self.name, self.mode = name, mode
        self.cpl = 1 if mode == "u16" else 2       # checked cells per 16-bit limb
        self.tbits = 16 if mode == "u16" else 8
        self.unchecked = 1                          # column 0 = TABLE
        self.checked = 0
        self.names = {"TABLE": 0}
        self._defer = []                            # (name, ncells) checked allocations
        self.prog = []
        self.n_ops = 0
        self.n_constraints = 0
        self.aux = []                               # (pi_word_index, part, shift_rows): part 0 = lo16, 1 = hi16, 2 = u32
        self.pi_per_io = 0
        self.max_e = 0
        self.log_rows = 9                           # log2 of the rows per IO record
        self.hardened = 0
        # VALUE-periodic columns (round 6, the pairing AIR): index N_PERIODIC + k in VEC flags and K_PER factors.  vflags: name -> 512
        # small integers (-1 / 0 / 1: usable as VEC flags); vconst: n columns whose row values come from a table of constant vectors
        self.vflag_names, self.vflags = [], []
        self.n_vconst = 0

    # ---- column allocation: all unchecked first, then all checked ----
    def alloc(self, name, n):
        base = self.unchecked
        self.names[name] = base
        self.unchecked += n
        return base

    def alloc_checked(self, name, ncells):
        self._defer.append((name, ncells))

    def finalize_columns(self):
        self.checked_base = self.unchecked
        off = self.checked_base
        for name, n in self._defer:
            self.names[name] = off
            off += n
        self.n_checked = off - self.checked_base
        self.n_main = off

    def col(self, name):
        return self.names[name]

    # ---- vectors ----
    def vec_u16(self, name, n=NL, coef=1, flag=-1, neg=0):
        """unchecked vector: one cell per limb"""
        return [(coef, self.col(name), 1, flag, neg)], n

    def vec_chk(self, name, n=NL, coef=1, flag=-1, neg=0):
        """checked limb vector: 1 or 2 cells per limb"""
        b = self.col(name)
        if self.cpl == 1:
            return [(coef, b, 1, flag, neg)], n
        return [(coef, b, 2, flag, neg), (coef * 256, b + 1, 2, flag, neg)], n

    @staticmethod
    def vsum(*vs):
        terms, n = [], vs[0][1]
        for t, m in vs:
            assert m == n
            terms += t
        return terms, n

    def _emit_vec(self, v):
        terms, n = v
        out = [n, len(terms)]
        for t in terms:
            out += list(t)
        return out

    # ---- gadget: E(2^16) == 0 mod p ----
    def gadget(self, gname, prods, lins, bound_bits):
        """prods: [(coef, vecA, vecB)], lins: [(coef, vecA)].  bound_bits: log2 bound of |e_k|."""
        # carry magnitude (any group g): |c_m| <= (|c_{m-1}| + 2^(16(g-1)) |d|) / 2^(16 g) ~ |d| / 2^16
        cbits_needed = max(bound_bits, 38) - 16 + 2      # signed, with slack
        total_bits = cbits_needed + 1
        if self.mode == "u16":
            ncl, lb = (total_bits + 15) // 16, 16
        else:
            ncl, lb = (total_bits + 7) // 8, 8
        coffset = 1 << (ncl * lb - 1)
        group = self.group
        # soundness: |D_m - c_{m-1} + 2^(16g) c_m| must stay below the Goldilocks prime for EVERY range-checked
        # assignment, so that the field identity forces the integer identity (0 is the only multiple of p there)
        worst = (1 << (ncl * lb - 1 + 16 * group)) + (1 << (bound_bits + 16 * (group - 1) + 1)) + (1 << (ncl * lb))
        assert worst < (1 << 64) - (1 << 32), (bound_bits, group, ncl, lb)
        self.gadgets.append(dict(name=gname, prods=prods, lins=lins, ncl=ncl, lb=lb, coffset=coffset, group=group))

    def emit_gadgets(self):
        for g in self.gadgets:
            nm = g["name"]
            sign = self.col(nm + "_s")
            cbase = self.col(nm + "_c")
            q = self.vec_chk(nm + "_q", NQ)
            w = [OP_GADGET, sign, cbase, g["ncl"], g["lb"], g["coffset"], g["group"]] + self._emit_vec(q)
            w += [len(g["prods"])]
            for coef, a, b in g["prods"]:
                w += [coef] + self._emit_vec(a) + self._emit_vec(b)
            w += [len(g["lins"])]
            for coef, a in g["lins"]:
                w += [coef] + self._emit_vec(a)
            self.prog += w
            self.n_ops += 1
            self.n_constraints += 2 * NL // g["group"] + 1  # 32 / group coefficient equations + sign booleanity

    def declare_gadget_cols(self, gname, bound_bits):
        cbits_needed = max(bound_bits, 38) - 16 + 2
        total_bits = cbits_needed + 1
        if self.mode == "u16":
            ncl = (total_bits + 15) // 16
        else:
            ncl = (total_bits + 7) // 8
        self.alloc(gname + "_s", 1)
        self.alloc_checked(gname + "_q", NQ * self.cpl)
        self.alloc_checked(gname + "_c", (2 * NL // self.group - 1) * ncl)

    # ---- generic polynomial constraints ----
    def poly(self, monos):
        """monos: [(coef, [(kind, index), ...])]; one constraint"""
        w = [OP_POLY, len(monos)]
        for coef, factors in monos:
            assert len(factors) <= 3
            w += [coef, len(factors)]
            for k, i in factors:
                w += [k, i]
        self.prog += w
        self.n_ops += 1
        self.n_constraints += 1

    def limb_expr(self, name, i, checked):
        """[(coef, (kind,col))] for limb i of a 16-bit-limb vector in the local row"""
        b = self.col(name)
        if not checked or self.cpl == 1:
            return [(1, b + i)]
        return [(1, b + 2 * i), (256, b + 2 * i + 1)]


def L(c):
    return (K_LOCAL, c)


def X(c):
    return (K_NEXT, c)


def PER(i):
    return (K_PER, i)


def AUX(i):
    return (K_AUX, i)


def exponent_logic(a):
    """bit / remaining-exponent-limb machinery shared by the three AIRs (2 rows per bit, LSB first)."""
    bit, e = a.col("bit"), a.col("e")
    a.poly([(1, [L(bit), L(bit)]), (-1, [L(bit)])])                                   # bit boolean
    a.poly([(2, [PER(PER_ADD), X(e)]), (1, [PER(PER_ADD), L(bit)]), (-1, [PER(PER_ADD), L(e)])])  # add rows: e0 = 2 e0' + bit
    # double rows that are not a limb end: e0' = e0          (1 - per_add - per_limb_end)
    a.poly([(1, [X(e)]), (-1, [L(e)]), (-1, [PER(PER_ADD), X(e)]), (1, [PER(PER_ADD), L(e)]),
            (-1, [PER(PER_LIMB_END), X(e)]), (1, [PER(PER_LIMB_END), L(e)])])
    a.poly([(1, [PER(PER_LIMB_END), L(e)])])                                          # limb consumed at its end
    for i in range(7):                                                                # rotate at limb ends (not block end)
        a.poly([(1, [PER(PER_LIMB_END), X(e + i)]), (-1, [PER(PER_LIMB_END), L(e + i + 1)]),
                (-1, [PER(PER_LAST), X(e + i)]), (1, [PER(PER_LAST), L(e + i + 1)])])
    for i in range(1, 8):                                                             # otherwise limbs 1..7 are copied
        a.poly([(1, [X(e + i)]), (-1, [L(e + i)]), (-1, [PER(PER_LIMB_END), X(e + i)]), (1, [PER(PER_LIMB_END), L(e + i)])])


def state_transition(a, state, result, upd_on_add, nl, use=None, late=None, copy=None):
    """state/result: column names of nl 16-bit limbs (state unchecked, result checked).
    upd_on_add: True  -> accumulator: add rows: s' = bit ? res : s ; double rows (not last): s' = s
                False -> running power: add rows: s' = s ; double rows (not last): s' = res
    hardened accumulator: `use` = the column that takes the place of bit on add rows (u = bit (1 - eq)); `late` = the column eqc
    with which a double row hands ITS result to the accumulator: s' = eqc ? res : s  (the sum R + P for R = P is the double 2 P)"""
    bit = a.col("bit") if use is None else use
    for i in range(nl):
        s = a.col(state) + i
        res = a.limb_expr(result, i, True)
        if upd_on_add:
            # per_add * (s' - s - bit*(res - s)) = 0
            m = [(1, [PER(PER_ADD), X(s)]), (-1, [PER(PER_ADD), L(s)]), (1, [PER(PER_ADD), L(bit), L(s)])]
            m += [(-c, [PER(PER_ADD), L(bit), L(cc)]) for c, cc in res]
            if copy is not None:      # ... - v (other - s): the accumulator restarts from the other state (hardened: R was the identity)
                vcol, other = copy
                m += [(1, [PER(PER_ADD), L(vcol), L(s)]), (-1, [PER(PER_ADD), L(vcol), L(a.col(other) + i)])]
            a.poly(m)
            # (1 - per_add - per_last) * (s' - s) = 0     (double rows except the block's last row)
            m = [(1, [X(s)]), (-1, [L(s)]), (-1, [PER(PER_ADD), X(s)]), (1, [PER(PER_ADD), L(s)]),
                 (-1, [PER(PER_LAST), X(s)]), (1, [PER(PER_LAST), L(s)])]
            if late is not None:
                # ... - (1 - per_add - per_last) * eqc * (res - s)
                m += [(1, [L(late), L(s)]), (-1, [PER(PER_ADD), L(late), L(s)]), (-1, [PER(PER_LAST), L(late), L(s)])]
                for c, cc in res:
                    m += [(-c, [L(late), L(cc)]), (c, [PER(PER_ADD), L(late), L(cc)]), (c, [PER(PER_LAST), L(late), L(cc)])]
            a.poly(m)
        else:
            a.poly([(1, [PER(PER_ADD), X(s)]), (-1, [PER(PER_ADD), L(s)])])
            m = [(1, [X(s)]), (-1, [PER(PER_ADD), X(s)]), (-1, [PER(PER_LAST), X(s)])]
            for c, cc in res:
                m += [(-c, [L(cc)]), (c, [PER(PER_ADD), L(cc)]), (c, [PER(PER_LAST), L(cc)])]
            a.poly(m)


def bind_pi(a, layout, tower=False):
    """layout: [(state_name, n_u32_words, row)] in PI order; row 'first' or 'last'.
    Every 16-bit limb cell of the state is bound to a value both sides derive from the public u32 words.
    aux descriptor = (word, part, shift_rows, sub):
       part 0 / 1: low / high 16 bits of public word `word`;  part 2: the whole u32 word
       part 3 (tower=True, 96-word Fq12 blocks): 16-bit limb `sub % 16` of tower component t = sub // 16 of the
               MyFq12 value stored at words [word, word + 96):  t = 2i: (c_i + 9 c_{i+6}) mod p,  t = 2i+1: c_{i+6}"""
    word = 0
    for name, nwords, row in layout:
        per = PER_FIRST if row == "first" else PER_LAST
        shift = 0 if row == "first" else ROWS_PER_IO - 1
        base = a.col(name)
        if name == "e":
            for j in range(nwords):
                ai = len(a.aux)
                a.aux.append((word + j, 2, shift, 0))
                a.poly([(1, [PER(per), L(base + j)]), (-1, [PER(per), AUX(ai)])])
        elif tower:
            assert nwords == 96
            for sub in range(12 * NL):
                ai = len(a.aux)
                a.aux.append((word, 3, shift, sub))
                a.poly([(1, [PER(per), L(base + sub)]), (-1, [PER(per), AUX(ai)])])
        else:
            for j in range(nwords):
                for part in (0, 1):
                    ai = len(a.aux)
                    a.aux.append((word + j, part, shift, 0))
                    a.poly([(1, [PER(per), L(base + 2 * j + part)]), (-1, [PER(per), AUX(ai)])])
        word += nwords
    a.pi_per_io = word


# ------------------------------------------------------------------------------------------------
def build_curve(name, mode, ext, hardened=False):
    """G1 (ext = 1, coordinates in Fq) or G2 (ext = 2, coordinates in Fq2 = Fq[u]/(u^2+1)).
    Row: one affine group operation  (x3, y3) = A (+) B  with slope lam:
       add rows  (even): A = R (accumulator), B = P (running power);  lam (xB - xA) = yB - yA
       double rows (odd): A = B = P;                                   2 lam yA = 3 xA^2
       lam^2 = xA + xB + x3 ;  lam (xA - x3) = yA + y3
    hardened = True (kinds 4 / 5, names g1h / g2h): the chord rule above says nothing when xB = xA (0 lam = 0 holds for EVERY lam when
    the accumulator meets the running power, R = P): the variant adds, per row,
       * x3 CANONICAL: T3 = p - 1 - x3 as range-checked limbs with a boolean borrow chain (so the limbs of R.x and P.x, which are
         copies of earlier x3's or public inputs, are THE limbs of their values), and
       * a flag eq on add rows ("R IS P": every limb of x and y equal), u = bit (1 - eq) in the place of the bit in the accumulator's
         transition, and  sum_j (Px_j - Rx_j) nz_j = u  with free cells nz_j: where the chord result is USED the two x's differ in a limb,
         hence mod p, and the slope is determined;
       * eqc on the double row after an add row with t1 eq = 1: that row's own result (2 P = R + P) goes to the accumulator;
       * a flag ng ("R is -P": x limbs equal, Ry + Py = p with carries cn) and a state bit inf ("the accumulator is the identity"; R's cells
         keep the last finite value): t1 = bit (1 - inf), w = t1 ng sets inf, v = bit inf copies P and clears it; the chord rule gets the
         product ng (Py - Ry) (NGV = the flag as a limb vector) so that it holds where R = -P; inf = 0 on a block's first and last row.
    Unprovable in both variants: an OUTPUT at the identity."""
    a = Air(name + ("h" if hardened else ""), mode)
    a.hardened = 1 if hardened else 0
    a.gadgets = []
    a.group = 2                        # limb identity checked in base 2^32: 15 carries per gadget instead of 31
    nc = NL * ext                      # limbs per coordinate
    a.alloc("Rx", nc); a.alloc("Ry", nc); a.alloc("Px", nc); a.alloc("Py", nc)
    a.alloc("bit", 1); a.alloc("e", 8)
    for nm in ("lam", "X3", "Y3"):
        a.alloc_checked(nm, nc * a.cpl)
    gad = []
    for eq in ("slope", "x3", "y3"):
        for c in range(ext):
            gad.append("%s%d" % (eq, c))
    bound = 42 if ext == 1 else 43
    for g in gad:
        a.declare_gadget_cols(g, bound)
    if hardened:                       # after everything else: the columns of the plain AIR keep their positions
        a.alloc("nz", nc)
        a.alloc("cb", (NL - 1) * ext)
        a.alloc("eq", 1); a.alloc("u", 1); a.alloc("eqc", 1)
        a.alloc("ng", 1); a.alloc("inf", 1); a.alloc("t1", 1); a.alloc("v", 1); a.alloc("w", 1)
        a.alloc("NGV", NL); a.alloc("cn", (NL - 1) * ext)
        a.alloc_checked("T3", nc * a.cpl)
    a.finalize_columns()

    def comp_u(nm, c, **kw):   # component c (0/1) of an unchecked Fq2/Fq value
        t, n = a.vec_u16(nm, **kw)
        return [(co, b + NL * c, st, f, ng) for (co, b, st, f, ng) in t], n

    def comp_c(nm, c, **kw):   # component c of a checked value
        t, n = a.vec_chk(nm, **kw)
        return [(co, b + NL * a.cpl * c, st, f, ng) for (co, b, st, f, ng) in t], n

    def fq2_mul_terms(va, vb):
        """va, vb: functions c -> vector for component c.  Returns {0: prods, 1: prods} of (coef, A, B)."""
        if ext == 1:
            return {0: [(1, va(0), vb(0))]}
        return {0: [(1, va(0), vb(0)), (-1, va(1), vb(1))], 1: [(1, va(0), vb(1)), (1, va(1), vb(0))]}

    lam = lambda c: comp_c("lam", c)
    # slope:  per_add * [lam*(Px - Rx) - (Py - Ry)] + (1 - per_add) * [2 lam Py - 3 Px^2] = 0
    dx = lambda c: Air.vsum(comp_u("Px", c, flag=PER_ADD), comp_u("Rx", c, coef=-1, flag=PER_ADD),
                            comp_u("Py", c, coef=2, flag=PER_ADD, neg=1))
    px_d = lambda c: comp_u("Px", c, flag=PER_ADD, neg=1)
    px = lambda c: comp_u("Px", c)
    t1 = fq2_mul_terms(lam, dx)
    t2 = fq2_mul_terms(px_d, px)
    for c in range(ext):
        prods = t1[c] + [(-3 * co, A, B) for (co, A, B) in t2[c]]
        lins = [(-1, comp_u("Py", c, flag=PER_ADD)), (1, comp_u("Ry", c, flag=PER_ADD))]
        if hardened:
            # ... + ng (Py - Ry) on add rows: where R = -P (flag ng) the chord rule would read 0 = 2 Py; NGV spells ng as a limb vector
            ngv = ([(1, a.col("NGV"), 1, PER_ADD, 0)], NL)
            prods = prods + [(1, ngv, Air.vsum(comp_u("Py", c), comp_u("Ry", c, coef=-1)))]
        a.gadget("slope%d" % c, prods, lins, bound)
    # x3:  lam^2 - xA - xB - x3 = 0 ;  xA + xB = per_add*(Rx + Px) + (1-per_add)*2Px
    t = fq2_mul_terms(lam, lam)
    for c in range(ext):
        lins = [(-1, comp_u("Rx", c, flag=PER_ADD)), (-1, comp_u("Px", c, flag=PER_ADD)),
                (-2, comp_u("Px", c, flag=PER_ADD, neg=1)), (-1, comp_c("X3", c))]
        a.gadget("x3%d" % c, t[c], lins, bound)
    # y3:  lam*(xA - x3) - yA - y3 = 0 ;  xA = per_add*Rx + (1-per_add)*Px
    xa_m = lambda c: Air.vsum(comp_u("Rx", c, flag=PER_ADD), comp_u("Px", c, flag=PER_ADD, neg=1), comp_c("X3", c, coef=-1))
    t = fq2_mul_terms(lam, xa_m)
    for c in range(ext):
        lins = [(-1, comp_u("Ry", c, flag=PER_ADD)), (-1, comp_u("Py", c, flag=PER_ADD, neg=1)), (-1, comp_c("Y3", c))]
        a.gadget("y3%d" % c, t[c], lins, bound)
    a.emit_gadgets()
    exponent_logic(a)
    if hardened:
        state_transition(a, "Rx", "X3", True, nc, use=a.col("u"), late=a.col("eqc"), copy=(a.col("v"), "Px"))
        state_transition(a, "Ry", "Y3", True, nc, use=a.col("u"), late=a.col("eqc"), copy=(a.col("v"), "Py"))
    else:
        state_transition(a, "Rx", "X3", True, nc)
        state_transition(a, "Ry", "Y3", True, nc)
    state_transition(a, "Px", "X3", False, nc)
    state_transition(a, "Py", "Y3", False, nc)
    w = 8 * ext
    # IO record order (x, offset, exp_val, output): reference src/verifier_circuit.rs:92-105
    bind_pi(a, [("Px", w, "first"), ("Py", w, "first"), ("Rx", w, "first"), ("Ry", w, "first"), ("e", 8, "first"),
                ("Rx", w, "last"), ("Ry", w, "last")])
    if hardened:
        bit = a.col("bit")
        for c in range(ext):
            for i in range(NL):
                # (p - 1)_i - x3_i - b_{i-1} - t_i + 2^16 b_i = 0,  b_{-1} = b_15 = 0: t = p - 1 - x3 >= 0
                pm1 = ((BN_P - 1) >> (16 * i)) & 0xFFFF
                m = [(pm1, [])]
                m += [(-co, [L(cc)]) for co, cc in a.limb_expr("X3", NL * c + i, True)]
                m += [(-co, [L(cc)]) for co, cc in a.limb_expr("T3", NL * c + i, True)]
                if i > 0:
                    m += [(-1, [L(a.col("cb") + (NL - 1) * c + i - 1)])]
                if i < NL - 1:
                    m += [(65536, [L(a.col("cb") + (NL - 1) * c + i)])]
                a.poly(m)
            for i in range(NL - 1):
                b = a.col("cb") + (NL - 1) * c + i
                a.poly([(1, [L(b), L(b)]), (-1, [L(b)])])
        # eq = 1: the accumulator IS the running power (all limbs of x and y equal; both canonical for honest provers, and a prover that
        # cannot set eq has to find the inequality witness below); u = bit (1 - eq): the chord result is taken
        eq, u, eqc = a.col("eq"), a.col("u"), a.col("eqc")
        a.poly([(1, [L(eq), L(eq)]), (-1, [L(eq)])])
        for j in range(nc):
            a.poly([(1, [L(eq), L(a.col("Px") + j)]), (-1, [L(eq), L(a.col("Rx") + j)])])
            a.poly([(1, [L(eq), L(a.col("Py") + j)]), (-1, [L(eq), L(a.col("Ry") + j)])])
        # ng = 1: the accumulator is MINUS the running power (x limbs equal, Ry + Py = p limb by limb with boolean carries cn): the sum is
        # the identity, which the state carries as the bit inf (R keeps its last finite value).  t1 = bit (1 - inf): a used addition
        # with a finite accumulator; u = t1 (1 - eq - ng): the chord; w = t1 ng: the accumulator becomes the identity; v = bit inf: a
        # used addition to the identity = a copy of P
        ng, inf, t1, v, w, ngv, cn = (a.col(x) for x in ("ng", "inf", "t1", "v", "w", "NGV", "cn"))
        for b in (ng, inf):
            a.poly([(1, [L(b), L(b)]), (-1, [L(b)])])
        a.poly([(1, [L(eq), L(ng)])])
        a.poly([(1, [L(ngv)]), (-1, [L(ng)])])
        for j in range(1, NL):
            a.poly([(1, [L(ngv + j)])])
        for j in range(nc):
            a.poly([(1, [L(ng), L(a.col("Px") + j)]), (-1, [L(ng), L(a.col("Rx") + j)])])
        for c in range(ext):
            for i in range(NL):
                m = [(1, [L(ng), L(a.col("Ry") + NL * c + i)]), (1, [L(ng), L(a.col("Py") + NL * c + i)]), (-P_LIMBS[i], [L(ng)])]
                if i > 0:
                    m += [(1, [L(ng), L(cn + (NL - 1) * c + i - 1)])]
                if i < NL - 1:
                    m += [(-65536, [L(ng), L(cn + (NL - 1) * c + i)])]
                a.poly(m)
            for i in range(NL - 1):
                b = cn + (NL - 1) * c + i
                a.poly([(1, [L(b), L(b)]), (-1, [L(b)])])
        a.poly([(1, [L(t1)]), (-1, [L(bit)]), (1, [L(bit), L(inf)])])
        a.poly([(1, [L(v)]), (-1, [L(bit), L(inf)])])
        a.poly([(1, [L(w)]), (-1, [L(t1), L(ng)])])
        a.poly([(1, [L(u)]), (-1, [L(t1)]), (1, [L(t1), L(eq)]), (1, [L(t1), L(ng)])])
        # the identity bit: add rows: inf' = inf - v + w; double rows: inf' = inf; finite at both ends of a block
        a.poly([(1, [PER(PER_ADD), X(inf)]), (-1, [PER(PER_ADD), L(inf)]), (1, [PER(PER_ADD), L(v)]), (-1, [PER(PER_ADD), L(w)])])
        a.poly([(1, [X(inf)]), (-1, [L(inf)]), (-1, [PER(PER_ADD), X(inf)]), (1, [PER(PER_ADD), L(inf)]),
                (-1, [PER(PER_LAST), X(inf)]), (1, [PER(PER_LAST), L(inf)])])
        a.poly([(1, [PER(PER_FIRST), L(inf)])])
        a.poly([(1, [PER(PER_LAST), L(inf)])])
        # where the chord result is taken the x's differ in a limb: sum_j (Px_j - Rx_j) nz_j = u on add rows
        m = [(-1, [PER(PER_ADD), L(u)])]
        for j in range(nc):
            m += [(1, [PER(PER_ADD), L(a.col("Px") + j), L(a.col("nz") + j)]), (-1, [PER(PER_ADD), L(a.col("Rx") + j), L(a.col("nz") + j)])]
        a.poly(m)
        # R = P with the bit set: the sum is the double the NEXT row computes: eqc (on that row) = bit eq (of this one); never on a
        # block's last row (no row left to hand the result over)
        a.poly([(1, [PER(PER_ADD), X(eqc)]), (-1, [PER(PER_ADD), L(t1), L(eq)])])
        a.poly([(1, [PER(PER_LAST), L(eqc)])])
        a.layout = [a.col("nz"), a.col("cb"), a.col("T3"), eq, u, eqc, ng, inf, t1, v, w, ngv, cn]
    a.primary = dict(kind="curve", ext=ext)
    return a


def build_fq12(mode):
    """out = offset * x^exp in Fq12.  The IO records carry MyFq12 coefficients c_0..c_11 over w with
    w^12 - 18 w^6 + 82 = 0; the TRACE works in the isomorphic tower basis
        Fq2[w] / (w^6 - xi),  xi = 9 + u,  u^2 = -1:     A_i = a_i + b_i u,  a_i = c_i + 9 c_{i+6},  b_i = c_{i+6}
    because its reduction constants are 1 and 9 instead of 18 / 82 / 242 / 1476: |e_k| < 2^44, which lets the
    gadgets pair limbs (group 2) and use 15 four-byte carries instead of 31 five-byte ones (-42 % columns).
    The basis change of the public inputs is a public computation (done natively by prover and verifier when they
    derive the public-input polynomials), so no conversion constraints exist.
    Row: one Fq12 product C = A * B;  mul rows (even): A = acc, B = pw;  square rows (odd): A = B = pw.
    Cell order of acc / pw / C: tower component t = 2 i + (0 for a_i, 1 for b_i), 16 limbs each."""
    a = Air("fq12", mode)
    a.gadgets = []
    a.group = 2
    a.alloc("acc", 12 * NL); a.alloc("pw", 12 * NL); a.alloc("bit", 1); a.alloc("e", 8)
    a.alloc_checked("C", 12 * NL * a.cpl)
    bound = 44
    for k in range(12):
        a.declare_gadget_cols("c%d" % k, bound)
    a.finalize_columns()

    def coef_u(nm, t, **kw):
        tm, n = a.vec_u16(nm, **kw)
        return [(co, b + NL * t, st, f, ng) for (co, b, st, f, ng) in tm], n

    def coef_c(nm, t, **kw):
        tm, n = a.vec_chk(nm, **kw)
        return [(co, b + NL * a.cpl * t, st, f, ng) for (co, b, st, f, ng) in tm], n

    A = lambda i, c: Air.vsum(coef_u("acc", 2 * i + c, flag=PER_ADD), coef_u("pw", 2 * i + c, flag=PER_ADD, neg=1))
    B = lambda j, c: coef_u("pw", 2 * j + c)
    for k in range(6):
        for comp in range(2):
            prods = []
            for i in range(6):
                for j in range(6):
                    if i + j == k:
                        # (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u
                        if comp == 0:
                            prods += [(1, A(i, 0), B(j, 0)), (-1, A(i, 1), B(j, 1))]
                        else:
                            prods += [(1, A(i, 0), B(j, 1)), (1, A(i, 1), B(j, 0))]
                    elif i + j == k + 6:
                        # times xi = 9 + u:  (x + y u)(9 + u) = (9x - y) + (x + 9y) u
                        if comp == 0:
                            prods += [(9, A(i, 0), B(j, 0)), (-9, A(i, 1), B(j, 1)), (-1, A(i, 0), B(j, 1)), (-1, A(i, 1), B(j, 0))]
                        else:
                            prods += [(1, A(i, 0), B(j, 0)), (-1, A(i, 1), B(j, 1)), (9, A(i, 0), B(j, 1)), (9, A(i, 1), B(j, 0))]
            a.gadget("c%d" % (2 * k + comp), prods, [(-1, coef_c("C", 2 * k + comp))], bound)
    a.emit_gadgets()
    exponent_logic(a)
    state_transition(a, "acc", "C", True, 12 * NL)
    state_transition(a, "pw", "C", False, 12 * NL)
    # IO record order (x, offset, exp_val, output): reference src/verifier_circuit.rs:111-123
    bind_pi(a, [("pw", 96, "first"), ("acc", 96, "first"), ("e", 8, "first"), ("acc", 96, "last")], tower=True)
    a.primary = dict(kind="fq12")
    return a


# ------------------------------------------------------------------------------------------------
def svdw_constants():
    """constants of the Shallue - van de Woestijne map for E'(Fp2): y^2 = x^3 + 3/(9+u), Z = 1 (RFC 9380 appendix F.1);
    the same values oracle/py/map_to_g2.py derives (tests compare)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import map_to_g2 as m
    import bn254
    return dict(C1=m.C1, C2=m.C2, C3=m.C3, C4=m.C4, BB=bn254.B2, ONE=(1, 0))


# MapToG2: eight rows per message, three Fp2 identities ("slots") per row; a value needed in a later row travels in one of six
# register column groups.  The schedule is DATA shared with the two trace generators (emitted into air_tables.h).
MAP_ROWS = 8
MAP_WIT = ["T1", "TV1", "W", "TV3", "A4", "B4", "X2", "X1", "S1", "GX1", "S2", "GX2", "D", "E", "F", "X3", "S3", "GX3", "N1", "N2", "Y"]
MAP_SLOTS = [["T1", "TV1", "W"], ["TV3", "A4", "D"], ["B4", "X2", "E"], ["X1", "F", "S2"], ["S1", "X3", "GX2"],
             ["GX1", "S3", "N2"], ["GX3", "N1", None], ["Y", None, None]]
# register k: {row t whose slot result it takes over for rows t+1 ..: witness}
MAP_REG_LOADS = [{0: "TV1", 1: "TV3", 2: "E", 3: "F", 4: "S1", 5: "S3", 6: "GX3"}, {0: "W", 1: "A4", 3: "X1"}, {1: "D", 2: "X2"},
                 {3: "S2", 5: "GX1"}, {4: "X3"}, {4: "GX2"}]


def map_reg_holds(k, t):
    """the witness register k holds AT row t of a block (None before its first load)"""
    best = None
    for lt, nm in MAP_REG_LOADS[k].items():
        if lt < t and (best is None or lt > best[0]):
            best = (lt, nm)
    return best[1] if best else None


def build_map_g2(mode):
    """u in Fp2  ->  (x, y) on E'(Fp2) by the Shallue - van de Woestijne map (RFC 9380 F.1, Z = 1): the statement behind
    `batch_map_to_g2_circuit` (reference src/bin/bls_aggregation.rs:65; the cofactor multiplication that follows the map
    there is an ordinary G2ExpStark obligation).  EIGHT ROWS PER MESSAGE: row t of a block evaluates the (up to) three Fp2
    identities MAP_SLOTS[t] in three gadget-pair slots whose operands are results of the same row, the message u, constants, or
    one of six registers (unchecked column groups) that took a slot result over at the end of an earlier row:
        next(REG_k) = RES_s on the rows MAP_REG_LOADS[k] names, = REG_k on the others (free across the block boundary).
    Row-type selectors are the periodic functions PER_MAP0 + t (1 on rows = t mod 8).  The three-way choice of x:
        e1 = 1            : x = x1, y^2 = g(x1)
        e1 = 0, e2 = 1    : N1^2 = (9+u) g(x1)  (g(x1) is NOT a square: 9+u is a non-residue),  x = x2, y^2 = g(x2)
        e1 = 0, e2 = 0    : N1^2 = (9+u) g(x1), N2^2 = (9+u) g(x2),  x = x3, y^2 = g(x3)
    with e1, e2 boolean and constant over the block.  The map's one inversion is inv0 (0 -> 0) by a flag bit z.  The sign rule sgn0(y) = sgn0(u) and the canonicity of u, x, y are PUBLIC
    checks on the public inputs (both provers and the verifier make them), not constraints.  IO record: u, x, y (16 u32 each):
    u is bound on row 0 of the block, x and y on row 7."""
    a = Air("mapg2", mode)
    a.gadgets = []
    a.group = 2
    a.log_rows = 3
    consts = svdw_constants()
    F2 = 2 * NL
    NREG, NSLOT = len(MAP_REG_LOADS), 3
    a.alloc("U", F2)
    for nm in ("ONE", "C1", "C2", "C3", "C4", "BB"):
        a.alloc(nm, F2)
    a.alloc("e1", 1); a.alloc("e2", 1)
    for nm in ("M1", "M2", "XS", "GXS"):
        a.alloc(nm, F2)
    a.alloc("REG", F2 * NREG)
    a.alloc("z", 1); a.alloc("ZV", F2)          # inv0: z = 1 where the product to invert is zero; ZV = the Fp2 value (z, 0) as limbs
    a.alloc_checked("RES", F2 * a.cpl * NSLOT)
    bound = 43
    for s in range(NSLOT):
        for c in range(2):
            a.declare_gadget_cols("slot%d_%d" % (s, c), bound)
    a.finalize_columns()

    def plain(nm):           # unchecked Fp2 value at a fixed column: (c, coef, flag) -> vector
        def f(c, coef=1, flag=-1):
            return [(coef, a.col(nm) + NL * c, 1, flag, 0)], NL
        return f

    def reg(k):
        def f(c, coef=1, flag=-1):
            return [(coef, a.col("REG") + F2 * k + NL * c, 1, flag, 0)], NL
        return f

    def res(sl):
        def f(c, coef=1, flag=-1):
            b = a.col("RES") + F2 * a.cpl * sl + NL * a.cpl * c
            if a.cpl == 1:
                return [(coef, b, 1, flag, 0)], NL
            return [(coef, b, 2, flag, 0), (coef * 256, b + 1, 2, flag, 0)], NL
        return f

    def lc(*terms):          # linear combination of Fp2 values: (coef, fn)
        def f(c, coef=1, flag=-1):
            return Air.vsum(*[fn(c, coef * k, flag) for k, fn in terms])
        return f

    def times_xi(fn):
        """(9 + u) v as two linear forms: component 0: 9 v0 - v1, component 1: v0 + 9 v1"""
        def f(c, coef=1, flag=-1):
            if c == 0:
                return Air.vsum(fn(0, 9 * coef, flag), fn(1, -coef, flag))
            return Air.vsum(fn(0, coef, flag), fn(1, 9 * coef, flag))
        return f

    def src(nm, t):
        """where the witness `nm` is read on row t: a slot of the same row or the register that holds it"""
        if nm in MAP_SLOTS[t]:
            return res(MAP_SLOTS[t].index(nm))
        for k in range(NREG):
            if map_reg_holds(k, t) == nm:
                return reg(k)
        raise AssertionError("%s is not available on row %d" % (nm, t))

    u, one = plain("U"), plain("ONE")

    def identity(nm, t):
        """(prods, lins) of the Fp2 identity that defines witness nm, operands resolved for row t"""
        g = lambda x: src(x, t)
        me = g(nm)
        if nm == "T1":  return [(1, u, u)], [(-1, me)]
        if nm == "TV1": return [(1, plain("C1"), g("T1"))], [(-1, me)]
        if nm == "W":   return [(1, lc((1, one), (-1, g("TV1"))), lc((1, one), (1, g("TV1"))))], [(-1, me)]
        if nm == "TV3": return [(1, me, g("W"))], [(-1, one), (1, plain("ZV"))]                # tv3 w = 1 - z  (inv0, see below)
        if nm == "A4":  return [(1, u, lc((1, one), (-1, g("TV1"))))], [(-1, me)]
        if nm == "D":   return [(1, lc((1, one), (1, g("TV1"))), lc((1, one), (1, g("TV1"))))], [(-1, me)]
        if nm == "B4":  return [(1, g("A4"), g("TV3"))], [(-1, me)]
        if nm == "X2":  return [(1, g("B4"), plain("C3"))], [(-1, me), (1, plain("C2"))]       # x2 = -Z/2 + tv4
        if nm == "E":   return [(1, g("D"), g("TV3"))], [(-1, me)]
        if nm == "X1":  return [], [(1, me), (1, g("X2")), (-2, plain("C2"))]                  # x1 = -Z/2 - tv4
        if nm == "F":   return [(1, g("E"), g("E"))], [(-1, me)]
        if nm in ("S1", "S2", "S3"):
            x = g("X" + nm[1])
            return [(1, x, x)], [(-1, me)]
        if nm == "X3":  return [(1, g("F"), plain("C4"))], [(1, one), (-1, me)]                # Z = 1
        if nm in ("GX1", "GX2", "GX3"):
            return [(1, g("S" + nm[2]), g("X" + nm[2]))], [(1, plain("BB")), (-1, me)]
        if nm == "N1":  return [(1, me, me)], [(-1, times_xi(plain("M1")))]
        if nm == "N2":  return [(1, me, me)], [(-1, times_xi(plain("M2")))]
        if nm == "Y":   return [(1, me, me)], [(-1, plain("GXS"))]
        raise AssertionError(nm)

    # one gadget pair per slot: the sum over the row types of (selector) x (identity of that row type)
    for sl in range(NSLOT):
        for c in range(2):
            pp, ll = [], []
            for t in range(MAP_ROWS):
                nm = MAP_SLOTS[t][sl]
                if nm is None:
                    continue
                fl = PER_MAP0 + t
                prods, lins = identity(nm, t)
                for coef, fa, fb in prods:
                    if c == 0:
                        pp += [(coef, fa(0, 1, fl), fb(0)), (-coef, fa(1, 1, fl), fb(1))]
                    else:
                        pp += [(coef, fa(0, 1, fl), fb(1)), (coef, fa(1, 1, fl), fb(0))]
                ll += [(coef, fn(c, 1, fl)) for coef, fn in lins]
            a.gadget("slot%d_%d" % (sl, c), pp, ll, bound)
    a.emit_gadgets()
    # constants
    for nm, v in consts.items():
        for c in range(2):
            for i in range(NL):
                a.poly([(1, [L(a.col(nm) + NL * c + i)]), (-((v[c] >> (16 * i)) & 0xFFFF), [])])
    e1, e2 = a.col("e1"), a.col("e2")
    a.poly([(1, [L(e1), L(e1)]), (-1, [L(e1)])])
    a.poly([(1, [L(e2), L(e2)]), (-1, [L(e2)])])
    # inv0(w) (RFC 9380: the inverse, 0 for w = 0): tv3 w = 1 - z with z boolean, and on the row that inverts z w = 0, z tv3 = 0.
    # z is a free bit on the other rows (the provers write 0); ZV spells the Fp2 element z for the gadget's linear term.
    zc, zv = a.col("z"), a.col("ZV")
    a.poly([(1, [L(zc), L(zc)]), (-1, [L(zc)])])
    a.poly([(1, [L(zv)]), (-1, [L(zc)])])
    for j in range(1, F2):
        a.poly([(1, [L(zv + j)])])
    t_inv = [t for t in range(MAP_ROWS) if "TV3" in MAP_SLOTS[t]][0]
    rw = [k for k in range(NREG) if map_reg_holds(k, t_inv) == "W"][0]
    sl_inv = MAP_SLOTS[t_inv].index("TV3")
    for j in range(F2):
        a.poly([(1, [PER(PER_MAP0 + t_inv), L(zc), L(a.col("REG") + F2 * rw + j)])])
    last = PER_MAP0 + MAP_ROWS - 1
    # u, e1, e2 are constant over a block: (1 - per_last) (next - local) = 0
    for col in [a.col("U") + j for j in range(F2)] + [e1, e2]:
        a.poly([(1, [X(col)]), (-1, [L(col)]), (-1, [PER(last), X(col)]), (1, [PER(last), L(col)])])

    def res_limb(sl, j):     # 16-bit limb j (0..31) of slot sl as [(coef, col)]
        b = a.col("RES") + F2 * a.cpl * sl
        return [(1, b + j)] if a.cpl == 1 else [(1, b + 2 * j), (256, b + 2 * j + 1)]

    for j in range(F2):
        a.poly([(co, [PER(PER_MAP0 + t_inv), L(zc), L(cc)]) for co, cc in res_limb(sl_inv, j)])
    # registers: (1 - per_last)(next - local) - sum_{t loads} per_t (res_s(t) - local) = 0
    for k in range(NREG):
        for j in range(F2):
            col = a.col("REG") + F2 * k + j
            m = [(1, [X(col)]), (-1, [L(col)]), (-1, [PER(last), X(col)]), (1, [PER(last), L(col)])]
            for t, nm in sorted(MAP_REG_LOADS[k].items()):
                sl = MAP_SLOTS[t].index(nm)
                m += [(1, [PER(PER_MAP0 + t), L(col)])]
                m += [(-co, [PER(PER_MAP0 + t), L(cc)]) for co, cc in res_limb(sl, j)]
            a.poly(m)
    # every value a later row reads sits in a register by then (checked when the identities were resolved); the selections read
    # the registers that hold x1, x2, x3, g(x1), g(x2), g(x3) on the rows that use them
    rk = {nm: [k for k in range(NREG) if map_reg_holds(k, MAP_ROWS - 1) == nm][0] for nm in ("X1", "X2", "X3", "GX1", "GX2", "GX3")}
    assert map_reg_holds(rk["GX2"], 5) == "GX2" and map_reg_holds(rk["GX1"], 6) == "GX1"       # rows of N2, N1
    R = lambda nm, j: a.col("REG") + F2 * rk[nm] + j
    for j in range(F2):
        # M1 = (1 - e1) g(x1)
        a.poly([(1, [L(a.col("M1") + j)]), (-1, [L(R("GX1", j))]), (1, [L(e1), L(R("GX1", j))])])
        # M2 = (1 - e1)(1 - e2) g(x2)
        c2_ = R("GX2", j)
        a.poly([(1, [L(a.col("M2") + j)]), (-1, [L(c2_)]), (1, [L(e1), L(c2_)]), (1, [L(e2), L(c2_)]), (-1, [L(e1), L(e2), L(c2_)])])
        # selected = e1 v1 + (1 - e1) e2 v2 + (1 - e1)(1 - e2) v3
        for dst, pre in (("XS", "X"), ("GXS", "GX")):
            v1, v2, v3 = R(pre + "1", j), R(pre + "2", j), R(pre + "3", j)
            a.poly([(1, [L(a.col(dst) + j)]), (-1, [L(e1), L(v1)]), (-1, [L(e2), L(v2)]), (1, [L(e1), L(e2), L(v2)]),
                    (-1, [L(v3)]), (1, [L(e1), L(v3)]), (1, [L(e2), L(v3)]), (-1, [L(e1), L(e2), L(v3)])])
    # public inputs: u on row 0 of the block, x and y on its last row
    word = 0
    ysl = MAP_SLOTS[MAP_ROWS - 1].index("Y")
    for nm, row in (("U", 0), ("XS", MAP_ROWS - 1), ("Y", MAP_ROWS - 1)):
        for j in range(F2):
            ai = len(a.aux)
            a.aux.append((word + j // 2, j % 2, row, 0))
            cells = res_limb(ysl, j) if nm == "Y" else [(1, a.col(nm) + j)]
            a.poly([(co, [PER(PER_MAP0 + row), L(cc)]) for co, cc in cells] + [(-1, [PER(PER_MAP0 + row), AUX(ai)])])
        word += NL
    a.pi_per_io = word
    a.primary = dict(kind="mapg2")
    a.layout = [a.col(nm) for nm in ("U", "ONE", "C1", "C2", "C3", "C4", "BB", "e1", "e2", "M1", "M2", "XS", "GXS", "REG", "RES")] + \
               [rk[nm] for nm in ("X1", "X2", "X3", "GX1", "GX2", "GX3")] + [zc, zv, t_inv]
    return a


def emit_map_schedule(f, prefix):
    wid = {nm: i for i, nm in enumerate(MAP_WIT)}
    f.write("/* MapToG2 schedule (tools/air_gen.py MAP_SLOTS / MAP_REG_LOADS): witness index (order: %s) or -1 */\n" % " ".join(MAP_WIT))
    f.write("#define %s_MAPG2_ROWS %d\n#define %s_MAPG2_NWIT %d\n" % (prefix, MAP_ROWS, prefix, len(MAP_WIT)))
    f.write("static const int32_t %s_MAPG2_SLOT_WIT[%d][3] = {%s};\n" % (prefix, MAP_ROWS, ", ".join(
        "{%s}" % ", ".join(str(wid[x]) if x else "-1" for x in row) for row in MAP_SLOTS)))
    f.write("static const int32_t %s_MAPG2_REG_WIT[%d][%d] = {%s};\n" % (prefix, MAP_ROWS, len(MAP_REG_LOADS), ", ".join(
        "{%s}" % ", ".join(str(wid[map_reg_holds(k, t)]) if map_reg_holds(k, t) else "-1" for k in range(len(MAP_REG_LOADS)))
        for t in range(MAP_ROWS))))


# ------------------------------------------------------------------------------------------------
N_PERIODIC = len(PERIODICS)


def build_pairing(mode):
    """Z = e(P, Q), the statement behind `pairing_circuit(final_A, final_B)` == final_Z of the reference's BLS example
    (src/bin/bls_aggregation.rs:76-77), arkworks' value (final exponent lambda (p^12 - 1)/r: oracle/py/bn254.py).  NARROW layout (round
    6b): ONE modular identity per row, 2^13 rows per pairing -- the operation schedule of tools/pairing_sched.py expanded by
    tools/pairing_rows.py (one row per Fq component of an Fq12 product / Frobenius map / point step; 7185 active rows).  Cells:
      REG      six Fq12 registers (12 x 16 limbs each);      A, B   the current operation's operands, multiplexed from REG by the row's
                                                                    selectors (B may be the line of the current step, or CACC)
      CACC     the components of the current Fq12 result as they are produced (committed to a register by an FCOMMIT row)
      SR       the five Fq2 results of the current point step (slope, x3, y3, -lam x_P, lam x_T - y_T), T the running twist point, QS the
               chord's other point (Q, pi(Q), -pi^2(Q) by selectors), Q1 / Q2N = pi(Q) / -pi^2(Q), P and Q constant over the block
      GC       the row's two constants (u, v): a Frobenius coefficient with its signs folded in, or 1 for the inversion check
      RES      the row's result (checked: 16 limbs), q / carries / sign of THE gadget
    The gadget's identity on a row of type (tools/pairing_rows.py):
      FMUL t   sum_{i+j=k} A_i B_j + xi sum_{i+j=k+6} A_i B_j  [component t]  - RES = 0        FINVC t: the same  - u = 0  (B = CACC)
      FFROB t  A_(k,0) u + A_(k,1) v - RES = 0       FCOPY t  B_t - RES = 0       FINVW / GW / FCOMMIT / IDLE: nothing (RES free, range-checked)
      GSL c    component c of  2 lam y_T - 3 x_T^2  (tangent)  /  lam (x_QS - x_T) - (y_QS - y_T)  (chord)
      GX3 / GY3 / GL1 / GL3 / GFQ: component c of the value's defining identity, RES the unknown
    Row types, operand selectors and loads are VALUE-PERIODIC columns (period 2^13) DERIVED from the row program: flag k on a row =
    sum of weight [row field == value] over the flag's terms (AIR_PAIRING_FLAGDEF) -- both provers and both verifiers evaluate the same
    table.  The VERIFIER checks P on E, Q on E' and [r] Q = O as public conditions (the chord rows are sound for Q of order r).
    IO record: P (16 u32), Q (32: x.c0, x.c1, y.c0, y.c1), Z (96: MyFq12 coefficients) = 144 words."""
    import pairing_rows as PR
    a = Air("pairing", mode)
    a.gadgets = []
    a.group = 2
    a.log_rows = PR.LOG_ROWS
    NR = PR.NREG
    F2, F12 = 2 * NL, 12 * NL
    # ---- derived fields of the row descriptors: one equality per flag term ----
    T = PR
    rows = []
    for r, d in enumerate(PR.ROWPROG):
        typ, t = d[PR.F_TYP], d[PR.F_T]
        ft = t if typ in (T.T_FMUL, T.T_FINVC) else -1
        fk = t // 2 if typ == T.T_FFROB else -1
        ct = t if typ == T.T_FCOPY else -1
        lc = t if typ in (T.T_FMUL, T.T_FFROB, T.T_FINVW, T.T_FCOPY) else -1
        sk = d[PR.F_SK]
        gt = -1
        if typ == T.T_GSL:
            gt = (0 if sk == T.SK_TANGENT else 2) + t
        elif typ == T.T_GX3:
            gt = (4 if sk == T.SK_TANGENT else 6) + t
        elif typ == T.T_GY3:
            gt = 8 + t
        elif typ == T.T_GL1:
            gt = 10 + t
        elif typ == T.T_GL3:
            gt = 12 + t
        elif typ == T.T_GFQ:
            gt = 14 + t
        ls = {T.T_GW: t, T.T_GX3: 2 + t, T.T_GY3: 4 + t, T.T_GL1: 6 + t, T.T_GL3: 8 + t, T.T_GFQ: t}.get(typ, -1)
        rows.append(list(d) + [ft, fk, ct, lc, gt, ls, 1 if r == 0 else 0, 1 if r == PR.ROWS - 1 else 0])
    (X_FT, X_FK, X_CT, X_LC, X_GT, X_LS, X_FIRST, X_LAST) = range(PR.N_FIELDS, PR.N_FIELDS + 8)
    a.rowprog, a.n_fields = rows, PR.N_FIELDS + 8
    a.flagdefs = []

    def vflag(name, terms):
        """terms: [(field, value, weight)]"""
        a.vflag_names.append(name)
        a.flagdefs.append(list(terms))
        return N_PERIODIC + len(a.flagdefs) - 1

    eq = lambda f, v, w=1: [(f, v, w)]
    sA = [vflag("sA%d" % k, eq(PR.F_RA, k)) for k in range(NR)]
    sB = [vflag("sB%d" % k, eq(PR.F_RB, k)) for k in range(NR)]
    sBL = vflag("sBL", eq(PR.F_BSEL, PR.B_LINE))
    sBC = vflag("sBC", eq(PR.F_BSEL, PR.B_CACC))
    fm = [vflag("fm%d" % t, eq(X_FT, t)) for t in range(12)]
    ff = [vflag("ff%d" % k, eq(X_FK, k)) for k in range(6)]
    fcp = [vflag("fcp%d" % t, eq(X_CT, t)) for t in range(12)]
    GT = lambda *codes: [(X_GT, c, 1) for c in codes]
    g_slT = [vflag("gslT%d" % c, GT(0 + c)) for c in range(2)]
    g_slC = [vflag("gslC%d" % c, GT(2 + c)) for c in range(2)]
    g_x3 = [vflag("gx3_%d" % c, GT(4 + c, 6 + c)) for c in range(2)]
    g_x3T = [vflag("gx3T%d" % c, GT(4 + c)) for c in range(2)]
    g_x3C = [vflag("gx3C%d" % c, GT(6 + c)) for c in range(2)]
    g_y3 = [vflag("gy3_%d" % c, GT(8 + c)) for c in range(2)]
    g_l1 = [vflag("gl1_%d" % c, GT(10 + c)) for c in range(2)]
    g_l3 = [vflag("gl3_%d" % c, GT(12 + c)) for c in range(2)]
    g_fq = [vflag("gfq%d" % t, GT(14 + t)) for t in range(8)]
    # RES enters the identity with -1 (it is the value the other terms define) or +1 (GL1: lam x_P + RES; GFQ slot 3: conj(S1) FY + RES)
    fres = vflag("fres", [(PR.F_TYP, T.T_FMUL, -1), (PR.F_TYP, T.T_FFROB, -1), (PR.F_TYP, T.T_FCOPY, -1), (PR.F_TYP, T.T_GX3, -1),
                          (PR.F_TYP, T.T_GY3, -1), (PR.F_TYP, T.T_GL3, -1), (PR.F_TYP, T.T_GL1, 1)] +
                 [(X_GT, 14 + t, -1) for t in range(6)] + [(X_GT, 20, 1), (X_GT, 21, 1)])
    fgc = vflag("fgc", eq(PR.F_TYP, T.T_FINVC))
    lc = [vflag("lc%d" % t, eq(X_LC, t)) for t in range(12)]
    ld = [vflag("ld%d" % k, eq(PR.F_LD, k)) for k in range(NR)]
    lS = [vflag("lS%d" % q, eq(X_LS, q)) for q in range(10)]
    gCHm = [vflag("gCH%d" % m, eq(PR.F_CHM, m)) for m in range(3)]
    ldT = vflag("ldT", eq(PR.F_END, PR.END_STEP))
    ldFQ = vflag("ldFQ", eq(PR.F_END, PR.END_FQ))
    vfirst = vflag("first", eq(X_FIRST, 1))
    vlast = vflag("last", eq(X_LAST, 1))
    a.n_vconst = F2
    PG = N_PERIODIC + len(a.flagdefs)
    a.gc_pairs = PR.GC_PAIRS

    # ---- columns ----
    for nm in ("PX", "PY"):
        a.alloc(nm, NL)
    for nm in ("QX", "QY", "Q1X", "Q1Y", "Q2X", "Q2Y", "TX", "TY", "QSX", "QSY", "FXC", "FYC"):
        a.alloc(nm, F2)
    a.alloc("SR", 5 * F2)
    a.alloc("GC", F2)
    for nm in ("A", "B", "CACC"):
        a.alloc(nm, F12)
    a.alloc("REG", NR * F12)
    a.alloc_checked("RES", NL * a.cpl)
    a.declare_gadget_cols("g", 44)
    a.finalize_columns()
    col = a.col

    # ---- the gadget ----
    def U(nm, off, flag=-1, coef=1):                    # 16 unchecked limbs at column nm + off
        return [(coef, col(nm) + off, 1, flag, 0)], NL

    Av = lambda i, c, flag=-1: U("A", NL * (2 * i + c), flag)
    Bv = lambda j, c: U("B", NL * (2 * j + c))
    prods, lins = [], []
    for k in range(6):
        for comp in range(2):
            fl = fm[2 * k + comp]
            for i in range(6):
                for j in range(6):
                    if i + j == k:
                        if comp == 0:
                            prods += [(1, Av(i, 0, fl), Bv(j, 0)), (-1, Av(i, 1, fl), Bv(j, 1))]
                        else:
                            prods += [(1, Av(i, 0, fl), Bv(j, 1)), (1, Av(i, 1, fl), Bv(j, 0))]
                    elif i + j == k + 6:
                        if comp == 0:
                            prods += [(9, Av(i, 0, fl), Bv(j, 0)), (-9, Av(i, 1, fl), Bv(j, 1)), (-1, Av(i, 0, fl), Bv(j, 1)), (-1, Av(i, 1, fl), Bv(j, 0))]
                        else:
                            prods += [(1, Av(i, 0, fl), Bv(j, 0)), (-1, Av(i, 1, fl), Bv(j, 1)), (9, Av(i, 0, fl), Bv(j, 1)), (9, Av(i, 1, fl), Bv(j, 0))]
    for k in range(6):                                  # FFROB: A_(k,0) u + A_(k,1) v
        prods += [(1, Av(k, 0, ff[k]), U("GC", 0)), (1, Av(k, 1, ff[k]), U("GC", NL))]
    for t in range(12):                                 # FCOPY: B_t
        lins.append((1, U("B", NL * t, fcp[t])))
    lins.append((-1, U("GC", 0, fgc)))                  # FINVC: ... - u
    res = a.vec_chk("RES", NL, flag=fres)               # the signed flag carries the -1 / +1
    lins.append((1, res))
    S = lambda s, c, flag=-1, coef=1: U("SR", F2 * s + NL * c, flag, coef)
    Pt = lambda nm, c, flag=-1, coef=1: U(nm, NL * c, flag, coef)
    dif = lambda va, vb: Air.vsum(va, ([(-co, b, st, f, ng) for (co, b, st, f, ng) in vb[0]], NL))
    # slope checks
    prods += [(2, S(0, 0, g_slT[0]), Pt("TY", 0)), (-2, S(0, 1, g_slT[0]), Pt("TY", 1)), (-3, Pt("TX", 0, g_slT[0]), Pt("TX", 0)), (3, Pt("TX", 1, g_slT[0]), Pt("TX", 1))]
    prods += [(2, S(0, 0, g_slT[1]), Pt("TY", 1)), (2, S(0, 1, g_slT[1]), Pt("TY", 0)), (-6, Pt("TX", 0, g_slT[1]), Pt("TX", 1))]
    dX = lambda c: dif(Pt("QSX", c), Pt("TX", c))
    prods += [(1, S(0, 0, g_slC[0]), dX(0)), (-1, S(0, 1, g_slC[0]), dX(1))]
    lins += [(-1, Pt("QSY", 0, g_slC[0])), (1, Pt("TY", 0, g_slC[0]))]
    prods += [(1, S(0, 0, g_slC[1]), dX(1)), (1, S(0, 1, g_slC[1]), dX(0))]
    lins += [(-1, Pt("QSY", 1, g_slC[1])), (1, Pt("TY", 1, g_slC[1]))]
    # x3 = lam^2 - x_T - x_B
    prods += [(1, S(0, 0, g_x3[0]), S(0, 0)), (-1, S(0, 1, g_x3[0]), S(0, 1)), (2, S(0, 0, g_x3[1]), S(0, 1))]
    for c in range(2):
        lins += [(-1, Pt("TX", c, g_x3[c])), (-1, Pt("TX", c, g_x3T[c])), (-1, Pt("QSX", c, g_x3C[c]))]
    # y3 = lam (x_T - x3) - y_T
    dT = lambda c: dif(Pt("TX", c), S(1, c))
    prods += [(1, S(0, 0, g_y3[0]), dT(0)), (-1, S(0, 1, g_y3[0]), dT(1)), (1, S(0, 0, g_y3[1]), dT(1)), (1, S(0, 1, g_y3[1]), dT(0))]
    for c in range(2):
        lins.append((-1, Pt("TY", c, g_y3[c])))
    # L1N: lam x_P + RES = 0;  L3 = lam x_T - y_T
    for c in range(2):
        prods.append((1, S(0, c, g_l1[c]), U("PX", 0)))
    prods += [(1, S(0, 0, g_l3[0]), Pt("TX", 0)), (-1, S(0, 1, g_l3[0]), Pt("TX", 1)), (1, S(0, 0, g_l3[1]), Pt("TX", 1)), (1, S(0, 1, g_l3[1]), Pt("TX", 0))]
    for c in range(2):
        lins.append((-1, Pt("TY", c, g_l3[c])))
    # GFQ: conj(x) F: c = 0: x0 f0 + x1 f1;  c = 1: x0 f1 - x1 f0
    for slot, (src, cst) in enumerate(((lambda c, fl: Pt("QX", c, fl), "FXC"), (lambda c, fl: Pt("QY", c, fl), "FYC"),
                                       (lambda c, fl: S(0, c, fl), "FXC"), (lambda c, fl: S(1, c, fl), "FYC"))):
        f0, f1 = g_fq[2 * slot], g_fq[2 * slot + 1]
        prods += [(1, src(0, f0), Pt(cst, 0)), (1, src(1, f0), Pt(cst, 1)), (1, src(0, f1), Pt(cst, 1)), (-1, src(1, f1), Pt(cst, 0))]
    a.gadget("g", prods, lins, 44)
    a.emit_gadgets()

    # ---- polynomial constraints ----
    def climb(j):                    # 16-bit limb j of RES as [(coef, col)]
        b = col("RES")
        return [(1, b + j)] if a.cpl == 1 else [(1, b + 2 * j), (256, b + 2 * j + 1)]

    def keep_or_load(c, loads):
        """(1 - last)(next - local) - sum_f per[f] (value - local) = 0; no load flag is set on a block's last row"""
        m = [(1, [X(c)]), (-1, [L(c)]), (-1, [PER(vlast), X(c)]), (1, [PER(vlast), L(c)])]
        for fl, val in loads:
            m += [(1, [PER(fl), L(c)])] + [(-co, [PER(fl), L(cc)]) for co, cc in val]
        a.poly(m)

    for j in range(F12):
        t, l = divmod(j, NL)
        a.poly([(1, [L(col("A") + j)])] + [(-1, [PER(sA[k]), L(col("REG") + F12 * k + j)]) for k in range(NR)])
        m = [(1, [L(col("B") + j)])] + [(-1, [PER(sB[k]), L(col("REG") + F12 * k + j)]) for k in range(NR)]
        m += [(-1, [PER(sBC), L(col("CACC") + j)])]
        if t == 0:                   # the line y_P - lam x_P w + (lam x_T - y_T) w^3: tower components 0 (y_P), 2 / 3 (SR3), 6 / 7 (SR4)
            m += [(-1, [PER(sBL), L(col("PY") + l)])]
        elif t in (2, 3):
            m += [(-1, [PER(sBL), L(col("SR") + F2 * 3 + NL * (t - 2) + l)])]
        elif t in (6, 7):
            m += [(-1, [PER(sBL), L(col("SR") + F2 * 4 + NL * (t - 6) + l)])]
        a.poly(m)
        keep_or_load(col("CACC") + j, [(lc[t], climb(l))])
    for j in range(F2):
        a.poly([(1, [L(col("GC") + j)]), (-1, [PER(PG + j)])])
    for k in range(NR):
        for j in range(F12):
            keep_or_load(col("REG") + F12 * k + j, [(ld[k], [(1, col("CACC") + j)])])
    for q in range(10):
        for l in range(NL):
            keep_or_load(col("SR") + NL * q + l, [(lS[q], climb(l))])
    for nm, n in (("PX", NL), ("PY", NL), ("QX", F2), ("QY", F2)):
        for j in range(n):
            keep_or_load(col(nm) + j, [])
    for j in range(F2):
        keep_or_load(col("Q1X") + j, [(ldFQ, [(1, col("SR") + F2 * 0 + j)])])
        keep_or_load(col("Q1Y") + j, [(ldFQ, [(1, col("SR") + F2 * 1 + j)])])
        keep_or_load(col("Q2X") + j, [(ldFQ, [(1, col("SR") + F2 * 2 + j)])])
        # (the last GFQ row produces component 1 of -pi^2(Q).y itself: that half comes straight from RES, the SR cell takes it at the same time)
        keep_or_load(col("Q2Y") + j, [(ldFQ, [(1, col("SR") + F2 * 3 + j)] if j < NL else climb(j - NL))])
        keep_or_load(col("TX") + j, [(ldT, [(1, col("SR") + F2 * 1 + j)]), (ldFQ, [(1, col("QX") + j)])])
        keep_or_load(col("TY") + j, [(ldT, [(1, col("SR") + F2 * 2 + j)]), (ldFQ, [(1, col("QY") + j)])])
        for dst, srcs in (("QSX", ("QX", "Q1X", "Q2X")), ("QSY", ("QY", "Q1Y", "Q2Y"))):
            a.poly([(1, [L(col(dst) + j)])] + [(-1, [PER(gCHm[m_]), L(col(srcs[m_]) + j)]) for m_ in range(3)])
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import bn254
    for nm, v in (("FXC", bn254.FROB_X), ("FYC", bn254.FROB_Y)):
        for c in range(2):
            for i in range(NL):
                a.poly([(1, [L(col(nm) + NL * c + i)]), (-((v[c] >> (16 * i)) & 0xFFFF), [])])
    # public inputs: P, Q on the block's first row, Z (tower limbs of the MyFq12 words) in the result register on its last row
    word = 0
    for nm, nwords in (("PX", 8), ("PY", 8), ("QX", 16), ("QY", 16)):
        for j in range(nwords):
            for part in (0, 1):
                ai = len(a.aux)
                a.aux.append((word + j, part, 0, 0))
                a.poly([(1, [PER(vfirst), L(col(nm) + 2 * j + part)]), (-1, [PER(vfirst), AUX(ai)])])
        word += nwords
    resreg = col("REG") + F12 * PR.RESULT_REG
    for sub in range(F12):
        ai = len(a.aux)
        a.aux.append((word, 3, PR.ROWS - 1, sub))
        a.poly([(1, [PER(vlast), L(resreg + sub)]), (-1, [PER(vlast), AUX(ai)])])
    word += 96
    a.pi_per_io = word
    a.primary = dict(kind="pairing")
    a.layout = [col(nm) for nm in ("PX", "PY", "QX", "QY", "Q1X", "Q1Y", "Q2X", "Q2Y", "TX", "TY", "QSX", "QSY", "FXC", "FYC", "SR", "GC", "A", "B",
                                   "CACC", "REG", "RES")]
    return a


def emit_pairing_schedule(f, prefix, a):
    """the row program, the row constants and the flag definitions of the pairing AIR: for the trace generators AND the constraint side"""
    import pairing_rows as PR
    f.write("/* final-pairing AIR (tools/pairing_rows.py): one row descriptor per trace row of a block; fields: %s,\n"
            "   then derived: fmul component, frobenius coefficient, copy component, CACC load, g2 identity code, SR load, first row, last row */\n"
            % "typ t ra rb bsel gc ld sk chm end")
    f.write("#define %s_PAIRING_LOG_ROWS %d\n#define %s_PAIRING_ROWS %d\n#define %s_PAIRING_NFIELDS %d\n#define %s_PAIRING_NREG %d\n"
            "#define %s_PAIRING_RESULT_REG %d\n#define %s_PAIRING_ACTIVE_ROWS %d\n" % (
                prefix, PR.LOG_ROWS, prefix, PR.ROWS, prefix, a.n_fields, prefix, PR.NREG, prefix, PR.RESULT_REG, prefix, PR.N_ACTIVE))
    f.write("static const int8_t %s_PAIRING_ROWPROG[%d][%d] = {\n" % (prefix, PR.ROWS, a.n_fields))
    for i in range(0, PR.ROWS, 4):
        f.write("    " + ", ".join("{%s}" % ", ".join(map(str, r)) for r in a.rowprog[i:i + 4]) + ",\n")
    f.write("};\n")
    f.write("/* row constants (u, v) as 16-bit limbs: FFROB rows RES = A_(k,0) u + A_(k,1) v; FINVC component 0: u = 1 */\n")
    f.write("#define %s_PAIRING_NGC %d\n" % (prefix, len(a.gc_pairs)))
    f.write("static const int64_t %s_PAIRING_GC[%d][%d] = {\n" % (prefix, len(a.gc_pairs), 2 * NL))
    for u, v in a.gc_pairs:
        f.write("    {" + ", ".join(str((x >> (16 * l)) & 0xFFFF) for x in (u, v) for l in range(NL)) + "},\n")
    f.write("};\n")
    f.write("/* selector columns: %s\n   flag k on a row = sum over its terms FLAGDEF[FLAGOFF[k] .. FLAGOFF[k+1]) of weight [row field == value] */\n"
            % " ".join(a.vflag_names))
    flat, off = [], [0]
    for terms in a.flagdefs:
        flat += terms
        off.append(len(flat))
    f.write("#define %s_PAIRING_NVFLAG %d\n" % (prefix, len(a.flagdefs)))
    f.write("static const int16_t %s_PAIRING_FLAGDEF[%d][3] = {%s};\n" % (prefix, len(flat), ", ".join("{%d, %d, %d}" % t for t in flat)))
    f.write("static const int16_t %s_PAIRING_FLAGOFF[%d] = {%s};\n" % (prefix, len(off), ", ".join(map(str, off))))
    # the operation schedule the row program was expanded from: the GPU's witness kernel walks operations, not rows (pairing.hip)
    S = PS.SCHEDULE
    f.write("/* the operation schedule (tools/pairing_sched.py): {fq12 op (0 idle 1 mul 2 line 3 inv 4 frob), register of A, register of B, constant vector,\n"
            "   register loaded, g2 op (0 idle 1 tangent 2 / 3 / 4 chord with Q / pi(Q) / -pi^2(Q) 5 frobenius of Q)} and its constant vectors */\n")
    f.write("#define %s_PAIRING_OPS %d\n" % (prefix, PS.ROWS))
    f.write("static const int8_t %s_PAIRING_SCHED[%d][6] = {\n" % (prefix, PS.ROWS))
    for i in range(0, PS.ROWS, 8):
        f.write("    " + ", ".join("{%d, %d, %d, %d, %d, %d}" % (r["fop"], r["ra"], r["rb"], r["gc"], r["rd"], r["gop"]) for r in S[i:i + 8]) + ",\n")
    f.write("};\n")
    vecs = [[(g[i][c] >> (16 * l)) & 0xFFFF for i in range(6) for c in range(2) for l in range(NL)] for g in PS.G_CONSTS] + [[0] * (12 * NL)]
    f.write("#define %s_PAIRING_NGCONST %d\n" % (prefix, len(vecs)))
    f.write("static const int64_t %s_PAIRING_GCONST[%d][%d] = {\n" % (prefix, len(vecs), 12 * NL))
    for v in vecs:
        f.write("    {" + ", ".join(map(str, v)) + "},\n")
    f.write("};\n")
    f.write("static const int8_t %s_PAIRING_GCONJ[%d] = {%s};\n" % (prefix, len(vecs), ", ".join(map(str, PS.G_CONJ_COEF + [0]))))


# ------------------------------------------------------------------------------------------------
def emit(a, f, prefix):
    tag = "%s_%s_%s" % (prefix, a.name, a.mode)
    f.write("static const int64_t %s_PROG[] = {\n" % tag)
    for i in range(0, len(a.prog), 16):
        f.write("    " + ", ".join("%dLL" % v for v in a.prog[i:i + 16]) + ",\n")
    f.write("};\n")
    f.write("static const int32_t %s_AUX[] = {\n" % tag)
    for i in range(0, len(a.aux), 8):
        f.write("    " + ", ".join("%d, %d, %d, %d" % t for t in a.aux[i:i + 8]) + ",\n")
    f.write("};\n")


def header_entry(a, prefix):
    tag = "%s_%s_%s" % (prefix, a.name, a.mode)
    g0 = a.gadgets[0]
    vp = "%d, %s_PAIRING_ROWPROG[0], %d, %s_PAIRING_FLAGDEF[0], %s_PAIRING_FLAGOFF, %d, %s_PAIRING_GC[0], 5" % (
        len(a.flagdefs), prefix, a.n_fields, prefix, prefix, a.n_vconst, prefix) if getattr(a, "flagdefs", None) else "0, 0, 0, 0, 0, 0, 0, 0"
    return ("    {\"%s\", %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %s_PROG, %d, %s_AUX, %d, %d, %s},\n" % (
        a.name + "_" + a.mode, {"g1": 0, "g2": 1, "fq12": 2, "mapg2": 3, "g1h": 0, "g2h": 1, "pairing": 6}[a.name], a.tbits, a.cpl, a.n_main,
        a.checked_base, a.n_checked, a.n_ops, a.n_constraints, len(a.aux), a.pi_per_io, len(a.gadgets), g0["ncl"], tag, len(a.prog), tag,
        a.log_rows, a.hardened, vp))


STRUCT = """typedef struct {
    const char *name;
    int kind;            /* 0 g1, 1 g2, 2 fq12, 3 mapg2, 6 pairing */
    int table_bits;      /* 16 or 8 */
    int cells_per_limb;  /* checked cells per 16-bit limb: 1 (u16 table) or 2 (u8 table) */
    int n_main;          /* TABLE + unchecked + checked cells */
    int checked_base;    /* first checked column; checked columns are [checked_base, n_main) */
    int n_checked;
    int n_ops;
    int n_constraints;   /* constraints produced by the program (gadgets + polys) */
    int n_aux;           /* public-input polynomials */
    int pi_per_io;       /* u32 words per IO record */
    int n_gadgets;
    int carry_limbs;
    const int64_t *prog;
    int prog_len;
    const int32_t *aux;  /* (pi word, part, row shift, sub) per aux column -- see bind_pi in tools/air_gen.py */
    int log_rows;        /* log2 of the trace rows per IO record: 9 for the exponentiation AIRs, 3 for mapg2 */
    int hardened;        /* 1: the curve AIR with canonical x3 and the x-inequality witness (API kinds 4 / 5 = kind + 4) */
    /* VALUE-periodic columns (period 2^log_rows, arbitrary values; index AIR_N_PERIODIC + k in VEC flags and periodic factors), DERIVED
     * from a row program: k < n_vflag: selector = sum over the terms flagdef[flagoff[k] .. flagoff[k + 1]) = {field, value, weight} of
     * weight [rowprog[row][field] == value];  n_vflag <= k < n_vflag + n_vconst: vconst[rowprog[row][vconst_field]][k - n_vflag] */
    int n_vflag;
    const int8_t *rowprog;      /* [2^log_rows][n_fields] */
    int n_fields;
    const int16_t *flagdef;     /* [][3] */
    const int16_t *flagoff;     /* [n_vflag + 1] */
    int n_vconst;
    const int64_t *vconst;      /* [][n_vconst] */
    int vconst_field;
} air_spec_t;
/* value of value-periodic column k (0-based) on row `row` of a block */
static inline int64_t air_vper_value(const air_spec_t *a, int k, int row) {
    const int8_t *d = a->rowprog + (size_t)row * (size_t)a->n_fields;
    if (k < a->n_vflag) {
        int64_t v = 0;
        for (int i = a->flagoff[k]; i < a->flagoff[k + 1]; i++)
            if (d[a->flagdef[3 * i]] == a->flagdef[3 * i + 1]) v += a->flagdef[3 * i + 2];
        return v;
    }
    return a->vconst[(size_t)d[a->vconst_field] * (size_t)a->n_vconst + (size_t)(k - a->n_vflag)];
}
"""


def column_map(a):
    rows = sorted(a.names.items(), key=lambda kv: kv[1])
    return ", ".join("%s@%d" % kv for kv in rows)


def main():
    airs = []
    for mode in ("u16", "u8"):
        airs += [build_curve("g1", mode, 1), build_curve("g2", mode, 2), build_fq12(mode), build_map_g2(mode),
                 build_curve("g1", mode, 1, hardened=True), build_curve("g2", mode, 2, hardened=True), build_pairing(mode)]
    # ONE file, included by the product (sipp_amd/csrc, -I data) and by the checker (oracle/, -I data) alike
    for path, prefix, guard in ((os.path.join(ROOT, "data", "air_tables.h"), "AIR", "SIPP_AIR_TABLES_H"),):
        with open(path, "w") as f:
            f.write("/* GENERATED by tools/air_gen.py -- the AIR specification as data; do not edit. */\n")
            f.write("#ifndef %s\n#define %s\n#include <stdint.h>\n" % (guard, guard))
            f.write(STRUCT)
            f.write("#define %s_N_PERIODIC %d\n" % (prefix, len(PERIODICS)))
            f.write("static const int32_t %s_PERIODIC[%d][2] = {%s};\n" % (
                prefix, len(PERIODICS), ", ".join("{%d, %d}" % p for p in PERIODICS)))
            f.write("static const uint32_t %s_BN_P_LIMBS[16] = {%s};\n" % (prefix, ", ".join(map(str, P_LIMBS))))
            emit_map_schedule(f, prefix)
            emit_pairing_schedule(f, prefix, [x for x in airs if x.name == "pairing"][0])
            for a in airs:
                f.write("/* %s: %s */\n" % (a.name + "_" + a.mode, column_map(a)))
                emit(a, f, prefix)
                if a.hardened:
                    f.write("/* hardened %s: columns of nz (x-inequality witness), cb (borrow bits), T3 (p - 1 - x3), eq, u, eqc, ng, inf, t1, v, w, NGV, cn */\n" % a.name)
                    f.write("static const int32_t %s_HARD_LAYOUT_%s_%s[13] = {%s};\n" % (prefix, a.name.upper(), a.mode.upper(), ", ".join(map(str, a.layout))))
                if a.name == "pairing":
                    f.write("/* columns of PX PY QX QY Q1X Q1Y Q2X Q2Y TX TY QSX QSY FXC FYC SR GC A B CACC REG RES */\n")
                    f.write("static const int32_t %s_PAIRING_LAYOUT_%s[%d] = {%s};\n" % (prefix, a.mode.upper(), len(a.layout), ", ".join(map(str, a.layout))))
                if a.name == "mapg2":
                    f.write("/* columns of U ONE C1 C2 C3 C4 BB e1 e2 M1 M2 XS GXS REG RES, the registers of x1 x2 x3 g(x1) g(x2) g(x3), columns of z ZV, the row type that inverts */\n")
                    f.write("static const int32_t %s_MAPG2_LAYOUT_%s[%d] = {%s};\n" % (prefix, a.mode.upper(), len(a.layout), ", ".join(map(str, a.layout))))
            f.write("static const air_spec_t %s_AIRS[%d] = {\n" % (prefix, len(airs)))
            for a in airs:
                f.write(header_entry(a, prefix))
            f.write("};\n#endif\n")
    for a in airs:
        W = a.n_main + 2 * a.n_checked
        print("%-10s n_main %5d checked %5d  W %6d  Z %6d  aux %4d  constraints %6d  prog %7d words  carry limbs %d" % (
            a.name + "_" + a.mode, a.n_main, a.n_checked, W, 2 * a.n_checked, len(a.aux), a.n_constraints, len(a.prog),
            a.gadgets[0]["ncl"]))


if __name__ == "__main__":
    main()
