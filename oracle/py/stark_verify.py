"""oracle/py/stark_verify.py -- a second, independent reading of the STARK VERIFIER (the C one is oracle/stark.c).

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.  Written from the specification, not from the C code: the flat proof layout of
INTEGRATION.md section 2, the Fiat-Shamir order stated there, the AIR encoding documented at the top of tools/air_gen.py
(the AIR tables are PARSED out of data/air_tables.h: the specification as data), and the generic layer of
oracle/py/plonky2_generic.py (Poseidon, Merkle, challenger, FRI query check: the Python reading of plonky2).  Everything is
evaluated over the quadratic extension at zeta with Python integers.

    verify(proof_words, cfg=DEFAULT) -> None on success, else a string naming the first failed check

What it re-derives on its own: the public checks on the records (canonical limbs, points on their curve, the sign rule of
MapToG2 records), the statement words and the Merkle root of the records, every challenge, the public-input polynomials at
zeta, the periodic selectors, ALL constraints of the AIR program (gadgets in base 2^(16 g), polynomial constraints), the range
table / permuted lookup / permutation-Z constraints, the quotient identity, and the FRI opening proof (proof of work, Merkle
paths of the three oracles and of every fold layer, fold consistency, final polynomial).  Slow (pure-Python Poseidon, ~600
permutations per second): meant for small proofs and few queries.
"""
import os
import re

from . import bn254 as bn
from . import plonky2_generic as g

P = g.P
Ext = g.Ext
ROOT = g.ROOT
MAGIC = 0x5349505053544b31

DEFAULT = dict(rate_bits=1, cap_height=4, pow_bits=16, arity_bits=4, final_poly_bits=5, num_queries=84, num_challenges=2, pow_rule=0, fs_rule=0,
               lookup_rule=0)

_TABLES = None


def air_tables():
    """the AIR specification as tools/air_gen.py emitted it: periodic functions, limbs of the BN254 prime, and per AIR the header
    fields with its program and public-input descriptors"""
    global _TABLES
    if _TABLES is not None:
        return _TABLES
    txt = open(os.path.join(ROOT, "data", "air_tables.h")).read()

    def ints(body):
        return [int(x) for x in re.findall(r"-?\d+", body.replace("LL", ""))]
    per = ints(re.search(r"AIR_PERIODIC\[\d+\]\[2\] = \{(.*?)\};", txt, re.S).group(1))
    periodic = [(per[2 * i], per[2 * i + 1]) for i in range(len(per) // 2)]
    p_limbs = ints(re.search(r"AIR_BN_P_LIMBS\[16\] = \{(.*?)\};", txt, re.S).group(1))
    arrays = {m.group(1): ints(m.group(2)) for m in re.finditer(r"static const int(?:64|32)_t (AIR_\w+_(?:PROG|AUX))\[\] = \{(.*?)\};", txt, re.S)}
    airs = []
    body = re.search(r"AIR_AIRS\[\d+\] = \{(.*?)\n\};", txt, re.S).group(1)
    for m in re.finditer(r'\{"(\w+)", ([^}]*)\}', body):
        f = [x.strip() for x in m.group(2).split(",")]
        names = ["kind", "table_bits", "cpl", "n_main", "checked_base", "n_checked", "n_ops", "n_constraints", "n_aux", "pi_per_io",
                 "n_gadgets", "carry_limbs"]
        a = {k: int(v) for k, v in zip(names, f[:12])}
        a["name"], a["prog"], a["aux"] = m.group(1), arrays[f[12]], arrays[f[14]]
        assert len(a["prog"]) == int(f[13])
        a["log_rows"], a["hardened"] = int(f[15]), int(f[16])
        a["n_vflag"], a["n_fields"], a["n_vconst"], a["vconst_field"] = int(f[17]), int(f[19]), int(f[22]), int(f[24])
        airs.append(a)
    # the pairing AIR's value-periodic columns (period 2^13), DERIVED from its row program: flag k on a row = sum of weight over the terms
    # (field, value, weight) of the flag whose field of the row's descriptor equals value; constants = a table row named by a field
    def table(name):
        m = re.search(r"%s\[(\d+)\](?:\[(\d+)\])? = \{(.*?)\};" % name, txt, re.S)
        v = ints(m.group(3))
        if m.group(2) is None:
            return v
        w = int(m.group(2))
        return [v[i * w:(i + 1) * w] for i in range(int(m.group(1)))]
    vper = dict(rowprog=table("AIR_PAIRING_ROWPROG"), flagdef=table("AIR_PAIRING_FLAGDEF"), flagoff=table("AIR_PAIRING_FLAGOFF"),
                gc=table("AIR_PAIRING_GC"))
    _TABLES = (periodic, p_limbs, airs, vper)
    return _TABLES


def vper_values(a, k):
    """the 2^log_rows values of value-periodic column k of AIR a (air_vper_value of data/air_tables.h)"""
    vper = air_tables()[3]
    rows = vper["rowprog"]
    if k < a["n_vflag"]:
        terms = vper["flagdef"][vper["flagoff"][k]:vper["flagoff"][k + 1]]
        return [sum(w for f, v, w in terms if d[f] == v) for d in rows]
    return [vper["gc"][d[a["vconst_field"]]][k - a["n_vflag"]] for d in rows]


def vper_at(a, log_n, zeta):
    """value-periodic column k is P_k(x^(N / R)), P_k interpolating its R = 2^log_rows values over the order-R subgroup"""
    y = zeta ** (1 << (log_n - a["log_rows"]))
    out = []
    cache = {}
    for k in range(a["n_vflag"] + a["n_vconst"]):
        vals = tuple(v % P for v in vper_values(a, k))
        if vals not in cache:
            cache[vals] = g.eval_poly([Ext(c) for c in g.ifft(list(vals))], y)
        out.append(cache[vals])
    return out


def air_of(kind, log_n):
    periodic, p_limbs, airs, _ = air_tables()
    base, hard = (kind - 4, 1) if kind in (4, 5) else (kind, 0)
    for a in airs:
        if a["kind"] == base and a["hardened"] == hard and (a["table_bits"] == 16) == (log_n >= 16):
            return a
    return None


# ---------------------------------------------------------------------------------------------------- public side of the records
def _fq(words):
    return sum(int(w) << (32 * i) for i, w in enumerate(words))


def records_ok(kind, pis, num_io, ppi):
    """None, or why the public inputs are refused: every field element canonical, the points of a record on their curve, and for
    MapToG2 the sign rule sgn0(y) = sgn0(u)"""
    base = kind - 4 if kind in (4, 5) else kind
    for io in range(num_io):
        rec = pis[io * ppi:(io + 1) * ppi]
        if base == 6:      # (P, Q, Z): canonical, P on E, Q on E' and of order r (the pairing AIR's chord rows are sound for such Q only)
            vals = [_fq(rec[8 * k: 8 * k + 8]) for k in range(18)]
            if any(v >= bn.P for v in vals):
                return "non-canonical record"
            q = ((vals[2], vals[3]), (vals[4], vals[5]))
            if not bn.g1_on_curve((vals[0], vals[1])) or not bn.g2_on_curve(q):
                return "record off the curve"
            if bn.g2_mul(q, bn.R - 1) != bn.g2_neg(q):
                return "Q outside the r-torsion"
            continue
        if base == 3:
            vals = [_fq(rec[8 * k: 8 * k + 8]) for k in range(6)]
            if any(v >= bn.P for v in vals):
                return "non-canonical record"
            u, q = (vals[0], vals[1]), ((vals[2], vals[3]), (vals[4], vals[5]))
            if not bn.g2_on_curve(q):
                return "record off the curve"
            s = lambda a: (a[0] & 1) | (int(a[0] == 0) & (a[1] & 1))
            if s(u) != s(q[1]):
                return "sign rule"
            continue
        fe = (2, 4, 12)[base]
        for k in range(3 * fe + 1):
            if k != 2 * fe and _fq(rec[8 * k: 8 * k + 8]) >= bn.P:      # the exponent may be any 256-bit value
                return "non-canonical record"
        if base == 0:
            pts = [(_fq(rec[0:8]), _fq(rec[8:16])), (_fq(rec[16:24]), _fq(rec[24:32]))]
            if not all(bn.g1_on_curve(p) for p in pts):
                return "record off the curve"
        elif base == 1:
            f2 = lambda o: ((_fq(rec[o:o + 8]), _fq(rec[o + 8:o + 16])), (_fq(rec[o + 16:o + 24]), _fq(rec[o + 24:o + 32])))
            if not all(bn.g2_on_curve(f2(o)) for o in (0, 32)):
                return "record off the curve"
    return None


def pi_root(pis, num_io, ppi):
    d = [g.hash_no_pad(pis[io * ppi:(io + 1) * ppi]) for io in range(num_io)]
    while len(d) > 1:
        d = [g.two_to_one(d[2 * i], d[2 * i + 1]) for i in range(len(d) // 2)]
    return d[0]


# ---------------------------------------------------------------------------------------------------- the AIR at a point
def periodic_at(log_n, m, r0, x):
    """S_{m,r0}(x) = (K / N) ((x g^-r0)^N - 1) / ((x g^-r0)^K - 1), K = N / m: 1 on the rows = r0 mod m, 0 on the others"""
    n = 1 << log_n
    k = n // m
    y = x * g.inv(pow(g.primitive_root_of_unity(log_n), r0, P))
    return (y ** n - 1) * (y ** k - 1).inverse() * (k * g.inv(n) % P)


def aux_value(a, rec, ai):
    word, part, _, sub = a["aux"][4 * ai: 4 * ai + 4]
    if part == 3:      # limb sub % 16 of tower component t = sub // 16 of the MyFq12 value at words [word, word + 96)
        c = [_fq(rec[word + 8 * k: word + 8 * k + 8]) for k in range(12)]
        t = sub // 16
        v = c[t // 2 + 6] if t & 1 else (c[t // 2] + 9 * c[t // 2 + 6]) % bn.P
        return (v >> (16 * (sub % 16))) & 0xFFFF
    w = int(rec[word])
    return w & 0xFFFF if part == 0 else w >> 16 if part == 1 else w


def aux_at(a, pis, num_io, log_n, zeta):
    """the public-input polynomials at zeta: interpolation of value(io) over the order-num_io subgroup, x -> x g^-shift"""
    ppi = a["pi_per_io"]
    out = []
    gN = g.primitive_root_of_unity(log_n)
    cache = {}
    for ai in range(a["n_aux"]):
        shift = a["aux"][4 * ai + 2]
        vals = [aux_value(a, pis[io * ppi:(io + 1) * ppi], ai) for io in range(num_io)]
        coeffs = g.ifft(vals)
        if shift not in cache:
            cache[shift] = zeta * g.inv(pow(gN, shift, P))
        out.append(g.eval_poly([Ext(c) for c in coeffs], cache[shift]))
    return out


def eval_constraints(a, log_n, local, nxt, aux, per, z_local, z_next, lag_first, lag_last, z_last, alphas, betas, gammas):
    """all constraints folded with alpha (acc = acc alpha + c) for each of the two challenges, in the specification's order:
    the program (gadgets, polynomial constraints), the range table, the permuted lookups, the permutation products"""
    p_limbs = air_tables()[1]
    acc = [Ext(0), Ext(0)]

    def emit(v):
        for i in range(2):
            acc[i] = acc[i] * alphas[i] + v

    prog = a["prog"]
    pos = 0

    def vec():
        nonlocal pos
        n, nt = prog[pos], prog[pos + 1]
        out = [Ext(0)] * n
        for t in range(nt):
            coef, base, stride, flag, neg = prog[pos + 2 + 5 * t: pos + 7 + 5 * t]
            f = Ext(coef)
            if flag >= 0:
                f = f * ((Ext(1) - per[flag]) if neg else per[flag])
            out = [o + f * local[base + i * stride] for i, o in enumerate(out)]
        pos += 2 + 5 * nt
        return out

    while pos < len(prog):
        op = prog[pos]
        if op == 1:      # gadget: E(2^16) = 0 mod p with quotient q, sign s and carries, checked in base 2^(16 grp)
            sign_col, cbase, ncl, lb, coffset, grp = prog[pos + 1: pos + 7]
            pos += 7
            q = vec()
            e = [Ext(0)] * 33
            np_ = prog[pos]; pos += 1
            for _ in range(np_):
                coef = prog[pos]; pos += 1
                va, vb = vec(), vec()
                for i, x in enumerate(va):
                    cx = x * coef
                    for j, y in enumerate(vb):
                        e[i + j] = e[i + j] + cx * y
            nl = prog[pos]; pos += 1
            for _ in range(nl):
                coef = prog[pos]; pos += 1
                for i, x in enumerate(vec()):
                    e[i] = e[i] + x * coef
            s = local[sign_col]
            sgn = Ext(1) - s - s
            cprev = Ext(0)
            nm = 32 // grp
            for m in range(nm):
                v = Ext(0)
                for t in range(grp - 1, -1, -1):
                    k = grp * m + t
                    qp = Ext(0)
                    for i, qi in enumerate(q):
                        if 0 <= k - i < 16:
                            qp = qp + qi * p_limbs[k - i]
                    v = v * 65536 + (e[k] - sgn * qp)
                ck = Ext(0)
                if m < nm - 1:
                    for l in range(ncl):
                        ck = ck + local[cbase + m * ncl + l] * (1 << (lb * l))
                    ck = ck - coffset
                emit(v - cprev + ck * (1 << (16 * grp)))
                cprev = ck
            emit(s * (s - 1))
        else:            # polynomial constraint: sum of monomials of cells (local / next row), public-input polynomials, periodic functions
            nmono = prog[pos + 1]
            pos += 2
            total = Ext(0)
            for _ in range(nmono):
                t = Ext(prog[pos])
                nf = prog[pos + 1]
                pos += 2
                for _ in range(nf):
                    kind, idx = prog[pos], prog[pos + 1]
                    pos += 2
                    t = t * (local[idx] if kind == 0 else nxt[idx] if kind == 1 else aux[idx] if kind == 2 else per[idx])
                total = total + t
            emit(total)
    nm_, nc, cb = a["n_main"], a["n_checked"], a["checked_base"]
    tl, tn = local[0], nxt[0]
    d = tn - tl
    emit(lag_first * tl)                                           # the table column starts at 0,
    emit(z_last * (d * (d - 1)))                                   # steps by 0 or 1,
    emit(lag_last * (tl - ((1 << a["table_bits"]) - 1)))           # and ends at T - 1
    for j in range(nc):
        pin, ptab, npin, nptab = local[nm_ + j], local[nm_ + nc + j], nxt[nm_ + j], nxt[nm_ + nc + j]
        emit(lag_first * (pin - ptab))
        emit(z_last * ((npin - pin) * (npin - nptab)))
    for i in range(2):
        for j in range(nc):
            z, zn = z_local[i * nc + j], z_next[i * nc + j]
            lhs = (local[cb + j] + gammas[i]) * (tl + betas[i])
            rhs = (local[nm_ + j] + gammas[i]) * (local[nm_ + nc + j] + betas[i])
            emit(lag_first * (z - 1))
            emit(zn * rhs - z * lhs)
    return acc


# ---------------------------------------------------------------------------------------------------- the verifier
def verify(proof, cfg=None):
    cfg = dict(DEFAULT, **(cfg or {}))
    pf = [int(x) for x in proof]
    if len(pf) < 16 or pf[0] != MAGIC:
        return "magic"
    if any(v >= P for v in pf[16:]) or any(v >> 32 for v in pf[1:16]):
        return "non-canonical word"
    kind, log_n, num_io, W, Pz, Q, cap_h, n_rounds, final_len, nq, ppi, total, rate_bits, arity_bits, zero = pf[1:16]
    log_rows = 3 if kind == 3 else 13 if kind == 6 else 9
    if not (0 <= kind <= 6) or not (10 <= log_n <= 26) or num_io != 1 << (log_n - log_rows):
        return "header"
    a = air_of(kind, log_n)
    if a is None or W != a["n_main"] + 2 * a["n_checked"] or Pz != 2 * a["n_checked"] or Q != 4 or cap_h != cfg["cap_height"] or \
            nq != cfg["num_queries"] or ppi != a["pi_per_io"] or total != len(pf) or rate_bits != cfg["rate_bits"] or \
            arity_bits != cfg["arity_bits"] or zero != (cfg["fs_rule"] | (cfg["lookup_rule"] << 1)):
        return "header / configuration"
    n = 1 << log_n
    arities = g.reduction_arity_bits(cfg["arity_bits"], cfg["final_poly_bits"], log_n, rate_bits, cap_h)
    if n_rounds != len(arities) or final_len != n >> sum(arities):
        return "FRI shape"
    n_pi = num_io * ppi
    pis = pf[len(pf) - n_pi:]
    if any(v >> 32 for v in pis):
        return "public input word"
    why = records_ok(kind, pis, num_io, ppi)
    if why:
        return why
    pos = [16]

    def take(k):
        v = pf[pos[0]:pos[0] + k]
        pos[0] += k
        return v
    ch = g.Challenger()
    if not cfg["fs_rule"]:        # this repository's format; fs_rule = 1: starky's recalled order, nothing before the trace cap
        ch.observe_many([kind, log_n, num_io, W, Pz, Q, rate_bits, cap_h, cfg["pow_bits"], arity_bits, cfg["final_poly_bits"], nq,
                         cfg["num_challenges"], cfg["pow_rule"], ppi, cfg["lookup_rule"]])
        ch.observe_many(pi_root(pis, num_io, ppi))
    ncap = 1 << cap_h
    caps3 = []
    trace_cap = [take(4) for _ in range(ncap)]
    ch.observe_cap(trace_cap)
    betas, gammas = [0, 0], [0, 0]
    for i in range(2):
        betas[i] = ch.get()
        gammas[i] = ch.get()
        if cfg["lookup_rule"]:    # both factors of a lookup under gamma
            betas[i] = gammas[i]
    z_cap = [take(4) for _ in range(ncap)]
    ch.observe_cap(z_cap)
    alphas = [ch.get(), ch.get()]
    q_cap = [take(4) for _ in range(ncap)]
    ch.observe_cap(q_cap)
    caps3 = [trace_cap, z_cap, q_cap]
    zeta = ch.get_ext()
    n_open = 2 * W + 2 * Pz + Q
    op = [Ext(*take(2)) for _ in range(n_open)]
    local, nxt = op[:W], op[W:2 * W]
    z_local, z_next = op[2 * W:2 * W + Pz], op[2 * W + Pz:2 * W + 2 * Pz]
    quot = op[2 * W + 2 * Pz:]
    for v in local + z_local + quot + nxt + z_next:      # the zeta batch, then the g zeta batch
        ch.observe_ext(v)
    # ---- the constraints at zeta
    periodic = air_tables()[0]
    per = [periodic_at(log_n, m, r0, zeta) for m, r0 in periodic] + vper_at(a, log_n, zeta)
    gN = g.primitive_root_of_unity(log_n)
    gi = g.inv(gN)
    zn = zeta ** n
    zh = zn - 1
    if zh == Ext(0):
        return "zeta in the subgroup"
    ninv = g.inv(n)
    lag_first = zh * ninv * (zeta - 1).inverse()
    lag_last = zh * (ninv * gi % P) * (zeta - gi).inverse()
    z_last = zeta - gi
    aux = aux_at(a, pis, num_io, log_n, zeta)
    acc = eval_constraints(a, log_n, local, nxt, aux, per, z_local, z_next, lag_first, lag_last, z_last, alphas, betas, gammas)
    for i in range(2):
        if acc[i] != zh * (quot[2 * i] + zn * quot[2 * i + 1]):
            return "quotient identity %d" % i
    # ---- the opening proof
    alpha = ch.get_ext()
    fcaps, fbetas = [], []
    log_m = log_n + rate_bits
    for r, ab in enumerate(arities):
        leaves_log = log_m - sum(arities[:r + 1])
        cap_n = 1 << min(cap_h, leaves_log)
        fcaps.append([take(4) for _ in range(cap_n)])
        ch.observe_cap(fcaps[-1])
        fbetas.append(ch.get_ext())
    final_poly = [Ext(*take(2)) for _ in range(final_len)]
    for c in final_poly:
        ch.observe_ext(c)
    w = take(1)[0]
    if not g.pow_ok(g.pow_response(ch, cfg["pow_rule"], w), cfg["pow_bits"]):
        return "proof of work"
    gzeta = zeta * gN
    ncols3 = [W, Pz, Q]
    for _ in range(nq):
        x = ch.get() % (1 << log_m)
        rows = []
        for o in range(3):
            row, sib = take(ncols3[o]), [take(4) for _ in range(log_m - cap_h)]
            if not g.verify_merkle_proof_to_cap(row, x, caps3[o], sib):
                return "Merkle path of oracle %d" % o
            rows.append(row)
        steps, xi = [], x
        for r, ab in enumerate(arities):
            ev = take(2 << ab)
            xi >>= ab
            leaves_log = log_m - sum(arities[:r + 1])
            sib = [take(4) for _ in range(max(0, leaves_log - cap_h))]
            if not g.verify_merkle_proof_to_cap(ev, xi, fcaps[r], sib):
                return "Merkle path of fold layer %d" % r
            steps.append([Ext(ev[2 * k], ev[2 * k + 1]) for k in range(1 << ab)])
        fb = [(zeta, rows[0] + rows[1] + rows[2], local + z_local + quot), (gzeta, rows[0] + rows[1], nxt + z_next)]
        why = g.fri_verify_query(x, log_n, rate_bits, arities, alpha, fb, rows, steps, fbetas, final_poly)
        if why:
            return "FRI query: " + why
    if pos[0] + n_pi != len(pf):
        return "length"
    return None
