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
    """Z = e(P, Q): ONE optimal ate pairing per 512-row block, the statement behind `pairing_circuit(final_A, final_B)` ==
    final_Z of the reference's BLS example (src/bin/bls_aggregation.rs:76-77), with arkworks' value (final exponent
    lambda (p^12 - 1)/r: oracle/py/bn254.py).  The row schedule is tools/pairing_sched.py (458 active rows: Miller loop in affine
    coordinates, easy part, ark-ec's hard-part chain); a row holds
      * the Fq12 unit: operands A, B (unchecked cells, multiplexed from the NREG registers by the schedule's selector columns; B
        may be the line of the same row or, on the inversion row, the result C itself), constants G (the row's Frobenius /
        conjugation constants, cells tied to periodic columns), result C (checked): twelve gadgets
            sum_{i+j=k} A_i B_j + xi sum_{i+j=k+6} A_i B_j + [conj^c(A_k) G_k] - sC C_k - sBC G_k = 0        (tower basis, build_fq12)
        so that MUL / LINE rows state C = A B, the INV row A C = 1 (G = 1 there), FROB rows C_k = conj^c(A_k) G_k (B = 0 there);
      * the G2 unit: T (the running twist point), QS (the chord's other point: Q, pi(Q) or -pi^2(Q) by the selectors), five Fq2
        results S0..S4 = (lam, x3, y3, -lam x_P, lam x_T - y_T) on step rows, the Frobenius images of Q on row 0: ten gadgets.
    Selector columns are VALUE-periodic (period 512, arbitrary values -- N_PERIODIC + k in flags and K_PER factors): both sides
    interpolate them from the schedule.  Soundness of the chord rows (x_T != x_QS) rests on Q having order r: the VERIFIER checks
    P on E, Q on E' and [r] Q = O as public conditions (oracle/stark.c), like the curve checks of the other kinds.
    IO record: P (16 u32), Q (32: x.c0, x.c1, y.c0, y.c1), Z (96: MyFq12 coefficients) = 144 words."""
    a = Air("pairing", mode)
    a.gadgets = []
    a.group = 2
    NR = PS.NREG
    F2, F12 = 2 * NL, 12 * NL
    sched = PS.SCHEDULE
    rows = PS.ROWS
    assert rows == ROWS_PER_IO

    # ---- value-periodic selector columns ----
    def vflag(name, fn):
        a.vflag_names.append(name)
        a.vflags.append([int(fn(r)) for r in sched])
        return N_PERIODIC + len(a.vflags) - 1

    is_f = lambda *ops: (lambda r: r["fop"] in ops)
    is_g = lambda *ops: (lambda r: r["gop"] in ops)
    sA = [vflag("sA%d" % k, lambda r, k=k: r["fop"] != PS.F_IDLE and r["ra"] == k) for k in range(NR)]
    sB = [vflag("sB%d" % k, lambda r, k=k: r["fop"] == PS.F_MUL and r["rb"] == k) for k in range(NR)]
    sBL = vflag("sBL", is_f(PS.F_LINE))
    sBC = vflag("sBC", is_f(PS.F_INV))
    sC = vflag("sC", is_f(PS.F_MUL, PS.F_LINE, PS.F_FROB))
    sFR = vflag("sFR", is_f(PS.F_FROB))
    sFS = vflag("sFS", lambda r: 0 if r["fop"] != PS.F_FROB else (-1 if PS.G_CONJ_COEF[r["gc"]] else 1))
    ld = [vflag("ld%d" % k, lambda r, k=k: r["fop"] != PS.F_IDLE and r["rd"] == k) for k in range(NR)]
    ldG = vflag("ldG", lambda r: r["fop"] == PS.F_IDLE and r["rd"] >= 0)
    rd0 = [r["rd"] for r in sched if r["fop"] == PS.F_IDLE and r["rd"] >= 0]
    assert rd0 == [sched[0]["rd"]] and sched[0]["gop"] == PS.G_FQ       # the one load from the constants: row 0
    gTG = vflag("gTG", is_g(PS.G_TG))
    gCHm = [vflag("gCH%d" % m, is_g(PS.G_CH0 + m)) for m in range(3)]
    gCH = vflag("gCH", is_g(PS.G_CH0, PS.G_CH1, PS.G_CH2))
    gST = vflag("gST", is_g(PS.G_TG, PS.G_CH0, PS.G_CH1, PS.G_CH2))
    gFQ = vflag("gFQ", is_g(PS.G_FQ))
    gLT = vflag("gLT", is_g(PS.G_TG, PS.G_CH0, PS.G_CH1))
    a.n_vconst = F12
    PG = N_PERIODIC + len(a.vflags)             # PG + j: the value of constant cell G_j on the row

    # ---- columns ----
    for nm in ("PX", "PY"):
        a.alloc(nm, NL)
    for nm in ("QX", "QY", "Q1X", "Q1Y", "Q2X", "Q2Y", "TX", "TY", "QSX", "QSY", "FXC", "FYC"):
        a.alloc(nm, F2)
    for nm in ("A", "B", "G"):
        a.alloc(nm, F12)
    a.alloc("REG", NR * F12)
    a.alloc_checked("C", F12 * a.cpl)
    for sl in range(5):
        a.alloc_checked("S%d" % sl, F2 * a.cpl)
    for k in range(12):
        a.declare_gadget_cols("c%d" % k, 44)
    for sl in range(5):
        for c in range(2):
            a.declare_gadget_cols("s%d_%d" % (sl, c), 43)
    a.finalize_columns()

    # ---- the Fq12 unit ----
    def coef_u(nm, t, **kw):
        tm, n = a.vec_u16(nm, **kw)
        return [(co, b + NL * t, st, f, ng) for (co, b, st, f, ng) in tm], n

    def coef_c(nm, t, **kw):
        tm, n = a.vec_chk(nm, **kw)
        return [(co, b + NL * a.cpl * t, st, f, ng) for (co, b, st, f, ng) in tm], n

    A = lambda i, c: coef_u("A", 2 * i + c)
    B = lambda j, c: coef_u("B", 2 * j + c)
    for k in range(6):
        for comp in range(2):
            prods = []
            for i in range(6):
                for j in range(6):
                    if i + j == k:
                        if comp == 0:
                            prods += [(1, A(i, 0), B(j, 0)), (-1, A(i, 1), B(j, 1))]
                        else:
                            prods += [(1, A(i, 0), B(j, 1)), (1, A(i, 1), B(j, 0))]
                    elif i + j == k + 6:
                        if comp == 0:
                            prods += [(9, A(i, 0), B(j, 0)), (-9, A(i, 1), B(j, 1)), (-1, A(i, 0), B(j, 1)), (-1, A(i, 1), B(j, 0))]
                        else:
                            prods += [(1, A(i, 0), B(j, 0)), (-1, A(i, 1), B(j, 1)), (9, A(i, 0), B(j, 1)), (9, A(i, 1), B(j, 0))]
            # FROB rows: (a0 + s a1 u)(g0 + g1 u), s = sFS = +-1
            if comp == 0:
                prods += [(1, A(k, 0), coef_u("G", 2 * k, flag=sFR)), (-1, A(k, 1), coef_u("G", 2 * k + 1, flag=sFS))]
            else:
                prods += [(1, A(k, 0), coef_u("G", 2 * k + 1, flag=sFR)), (1, A(k, 1), coef_u("G", 2 * k, flag=sFS))]
            t = 2 * k + comp
            a.gadget("c%d" % t, prods, [(-1, coef_c("C", t, flag=sC)), (-1, coef_u("G", t, flag=sBC))], 44)

    # ---- the G2 unit ----
    def u2(nm):                      # unchecked Fq2 value: (component, coef, flag) -> vector
        return lambda c, coef=1, flag=-1: ([(coef, a.col(nm) + NL * c, 1, flag, 0)], NL)

    def fq_as_f2(nm):                # an Fq value x as the Fq2 element (x, 0): component 1 has no terms
        return lambda c, coef=1, flag=-1: ([(coef, a.col(nm), 1, flag, 0)] if c == 0 else [], NL)

    def c2(nm):                      # checked Fq2 value
        def f(c, coef=1, flag=-1):
            b = a.col(nm) + NL * a.cpl * c
            if a.cpl == 1:
                return [(coef, b, 1, flag, 0)], NL
            return [(coef, b, 2, flag, 0), (coef * 256, b + 1, 2, flag, 0)], NL
        return f

    def lc(*terms):
        return lambda c, coef=1, flag=-1: Air.vsum(*[fn(c, coef * k, flag) for k, fn in terms])

    def f2prod(coef, fa, fb, flag, conj=False):
        """{component: [(coef, vecA, vecB)]} of coef * fa * fb (fa conjugated first if conj); the flag sits on fa"""
        s = -1 if conj else 1
        out = {0: [(coef, fa(0, 1, flag), fb(0)), (-coef * s, fa(1, 1, flag), fb(1))],
               1: [(coef, fa(0, 1, flag), fb(1)), (coef * s, fa(1, 1, flag), fb(0))]}
        return {c: [(co, va, vb) for co, va, vb in v if va[0] and vb[0]] for c, v in out.items()}

    LAM, X3, Y3, L1N, L3 = (c2("S%d" % sl) for sl in range(5))
    TX, TY, QSX, QSY, QX, QY, FXC, FYC = (u2(nm) for nm in ("TX", "TY", "QSX", "QSY", "QX", "QY", "FXC", "FYC"))
    PXf = fq_as_f2("PX")
    slots = [
        # S0 = lam:  TG: 2 lam y_T - 3 x_T^2;  CH: lam (x_QS - x_T) - (y_QS - y_T);  FQ: conj(x_Q) FX - S0
        ([f2prod(2, LAM, TY, gTG), f2prod(-3, TX, TX, gTG), f2prod(1, LAM, lc((1, QSX), (-1, TX)), gCH), f2prod(1, QX, FXC, gFQ, conj=True)],
         [(-1, QSY, gCH), (1, TY, gCH), (-1, LAM, gFQ)]),
        # S1 = x3:   lam^2 - x_T - x_B - x3 (x_B = x_T on tangent rows, x_QS on chord rows);  FQ: conj(y_Q) FY - S1
        ([f2prod(1, LAM, LAM, gST), f2prod(1, QY, FYC, gFQ, conj=True)],
         [(-1, TX, gST), (-1, TX, gTG), (-1, QSX, gCH), (-1, X3, gST), (-1, X3, gFQ)]),
        # S2 = y3:   lam (x_T - x3) - y_T - y3;  FQ: conj(S0) FX - S2  (x of pi^2(Q))
        ([f2prod(1, LAM, lc((1, TX), (-1, X3)), gST), f2prod(1, LAM, FXC, gFQ, conj=True)],
         [(-1, TY, gST), (-1, Y3, gST), (-1, Y3, gFQ)]),
        # S3 = -lam x_P:  lam x_P + S3;  FQ: conj(S1) FY + S3  (y of -pi^2(Q))
        ([f2prod(1, LAM, PXf, gST), f2prod(1, X3, FYC, gFQ, conj=True)],
         [(1, L1N, gST), (1, L1N, gFQ)]),
        # S4 = lam x_T - y_T
        ([f2prod(1, LAM, TX, gST)], [(-1, TY, gST), (-1, L3, gST)]),
    ]
    for sl, (prods, lins) in enumerate(slots):
        for c in range(2):
            pp = [p for d in prods for p in d[c]]
            ll = [(coef, fn(c, 1, fl)) for coef, fn, fl in lins]
            a.gadget("s%d_%d" % (sl, c), pp, ll, 43)
    a.emit_gadgets()

    # ---- polynomial constraints ----
    col = a.col
    last = PER_LAST

    def climb(name, j):              # 16-bit limb j of a checked vector as [(coef, col)]
        b = col(name)
        return [(1, b + j)] if a.cpl == 1 else [(1, b + 2 * j), (256, b + 2 * j + 1)]

    def keep_or_load(c, loads):
        """(1 - per_last)(next - local) - sum_f per[f] (value - local) = 0 for loads = [(flag, [(coef, col)])]; the flags are 0 on
        a block's last row (nothing is loaded there), where the next block starts free"""
        m = [(1, [X(c)]), (-1, [L(c)]), (-1, [PER(last), X(c)]), (1, [PER(last), L(c)])]
        for fl, val in loads:
            m += [(1, [PER(fl), L(c)])] + [(-co, [PER(fl), L(cc)]) for co, cc in val]
        a.poly(m)

    # operands
    for j in range(F12):
        a.poly([(1, [L(col("A") + j)])] + [(-1, [PER(sA[k]), L(col("REG") + F12 * k + j)]) for k in range(NR)])
        t, l = divmod(j, NL)
        m = [(1, [L(col("B") + j)])] + [(-1, [PER(sB[k]), L(col("REG") + F12 * k + j)]) for k in range(NR)]
        m += [(-co, [PER(sBC), L(cc)]) for co, cc in climb("C", j)]
        # the line y_P - lam x_P w + (lam x_T - y_T) w^3: tower components 0 (y_P), 2 / 3 (S3), 6 / 7 (S4)
        if t == 0:
            m += [(-1, [PER(sBL), L(col("PY") + l)])]
        elif t in (2, 3):
            m += [(-co, [PER(sBL), L(cc)]) for co, cc in climb("S3", NL * (t - 2) + l)]
        elif t in (6, 7):
            m += [(-co, [PER(sBL), L(cc)]) for co, cc in climb("S4", NL * (t - 6) + l)]
        a.poly(m)
        a.poly([(1, [L(col("G") + j)]), (-1, [PER(PG + j)])])
    # registers
    for k in range(NR):
        for j in range(F12):
            loads = [(ld[k], climb("C", j))]
            if k == rd0[0]:
                loads.append((ldG, [(1, col("G") + j)]))
            keep_or_load(col("REG") + F12 * k + j, loads)
    # P and Q are constant over a block; T, pi(Q), -pi^2(Q) are registers of the G2 unit
    for nm, n in (("PX", NL), ("PY", NL), ("QX", F2), ("QY", F2)):
        for j in range(n):
            keep_or_load(col(nm) + j, [])
    for j in range(F2):
        keep_or_load(col("Q1X") + j, [(gFQ, climb("S0", j))])
        keep_or_load(col("Q1Y") + j, [(gFQ, climb("S1", j))])
        keep_or_load(col("Q2X") + j, [(gFQ, climb("S2", j))])
        keep_or_load(col("Q2Y") + j, [(gFQ, climb("S3", j))])
        keep_or_load(col("TX") + j, [(gLT, climb("S1", j)), (gFQ, [(1, col("QX") + j)])])
        keep_or_load(col("TY") + j, [(gLT, climb("S2", j)), (gFQ, [(1, col("QY") + j)])])
        for d, srcs in (("QSX", ("QX", "Q1X", "Q2X")), ("QSY", ("QY", "Q1Y", "Q2Y"))):
            a.poly([(1, [L(col(d) + j)])] + [(-1, [PER(gCHm[m_]), L(col(srcs[m_]) + j)]) for m_ in range(3)])
    # the twist's Frobenius constants
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
                a.poly([(1, [PER(PER_FIRST), L(col(nm) + 2 * j + part)]), (-1, [PER(PER_FIRST), AUX(ai)])])
        word += nwords
    res = col("REG") + F12 * PS.RESULT_REG
    for sub in range(F12):
        ai = len(a.aux)
        a.aux.append((word, 3, ROWS_PER_IO - 1, sub))
        a.poly([(1, [PER(PER_LAST), L(res + sub)]), (-1, [PER(PER_LAST), AUX(ai)])])
    word += 96
    a.pi_per_io = word
    a.primary = dict(kind="pairing")
    a.layout = [col(nm) for nm in ("PX", "PY", "QX", "QY", "Q1X", "Q1Y", "Q2X", "Q2Y", "TX", "TY", "QSX", "QSY", "FXC", "FYC", "A", "B", "G",
                                   "REG", "C", "S0")]
    return a


def emit_pairing_schedule(f, prefix, a):
    """the schedule and the constant vectors for the two trace generators; the selector columns for everybody"""
    S = PS.SCHEDULE
    f.write("/* final-pairing AIR (tools/pairing_sched.py): per row {fq12 op (0 idle 1 mul 2 line 3 inv 4 frob), register of A, register of B,\n"
            "   constant vector, register loaded at the end of the row, g2 op (0 idle 1 tangent 2 / 3 / 4 chord with Q / pi(Q) / -pi^2(Q) 5 frobenius of Q)} */\n")
    f.write("#define %s_PAIRING_ROWS %d\n#define %s_PAIRING_NREG %d\n#define %s_PAIRING_RESULT_REG %d\n#define %s_PAIRING_ACTIVE_ROWS %d\n" % (
        prefix, PS.ROWS, prefix, PS.NREG, prefix, PS.RESULT_REG, prefix, PS.N_ACTIVE))
    f.write("static const int8_t %s_PAIRING_SCHED[%d][6] = {\n" % (prefix, PS.ROWS))
    for i in range(0, PS.ROWS, 8):
        f.write("    " + ", ".join("{%d, %d, %d, %d, %d, %d}" % (r["fop"], r["ra"], r["rb"], r["gc"], r["rd"], r["gop"]) for r in S[i:i + 8]) + ",\n")
    f.write("};\n")
    # constant vectors as 16-bit limbs in cell order (tower component t = 2 i + c, 16 limbs each); the last one is zero
    vecs = [[(g[i][c] >> (16 * l)) & 0xFFFF for i in range(6) for c in range(2) for l in range(NL)] for g in PS.G_CONSTS] + [[0] * (12 * NL)]
    f.write("/* constant vectors: 1, conjugation, Frobenius p / p^2 / p^3, zero; whether the operand's coefficients are conjugated first */\n")
    f.write("#define %s_PAIRING_NGCONST %d\n" % (prefix, len(vecs)))
    f.write("static const int64_t %s_PAIRING_GCONST[%d][%d] = {\n" % (prefix, len(vecs), 12 * NL))
    for v in vecs:
        f.write("    {" + ", ".join(map(str, v)) + "},\n")
    f.write("};\n")
    f.write("static const int8_t %s_PAIRING_GCONJ[%d] = {%s};\n" % (prefix, len(vecs), ", ".join(map(str, PS.G_CONJ_COEF + [0]))))
    f.write("static const int8_t %s_PAIRING_GIDX[%d] = {%s};\n" % (prefix, PS.ROWS, ", ".join(
        str(r["gc"] if r["gc"] >= 0 else len(vecs) - 1) for r in S)))
    f.write("/* selector columns (period %d): %s */\n" % (PS.ROWS, " ".join(a.vflag_names)))
    f.write("#define %s_PAIRING_NVFLAG %d\n" % (prefix, len(a.vflags)))
    f.write("static const int8_t %s_PAIRING_VFLAG[%d][%d] = {\n" % (prefix, len(a.vflags), PS.ROWS))
    for v in a.vflags:
        f.write("    {" + ", ".join(map(str, v)) + "},\n")
    f.write("};\n")


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
    vp = "%d, %s_PAIRING_VFLAG[0], %d, %s_PAIRING_GCONST[0], %s_PAIRING_GIDX" % (len(a.vflags), prefix, a.n_vconst, prefix, prefix) \
        if a.vflags else "0, 0, 0, 0, 0"
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
    /* VALUE-periodic columns (period 2^log_rows, arbitrary values; index AIR_N_PERIODIC + k in VEC flags and periodic factors):
     * k < n_vflag: selector vflag[k][row] in {-1, 0, 1};  n_vflag <= k < n_vflag + n_vconst: vconst[vconst_idx[row]][k - n_vflag] */
    int n_vflag;
    const int8_t *vflag;
    int n_vconst;
    const int64_t *vconst;
    const int8_t *vconst_idx;
} air_spec_t;
/* value of value-periodic column k (0-based) on row `row` of a block */
static inline int64_t air_vper_value(const air_spec_t *a, int k, int row) {
    if (k < a->n_vflag) return a->vflag[((size_t)k << a->log_rows) + (size_t)row];
    return a->vconst[(size_t)a->vconst_idx[row] * (size_t)a->n_vconst + (size_t)(k - a->n_vflag)];
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
                    f.write("/* columns of PX PY QX QY Q1X Q1Y Q2X Q2Y TX TY QSX QSY FXC FYC A B G REG C S0 */\n")
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
