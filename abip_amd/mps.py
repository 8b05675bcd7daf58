"""MPS -> ABIP standard form: the data format on the input side of the hot path.

``mpsread`` returns what Matlab's ``mpsread`` returns to the reference's ``scripts/bench-lp/preprocess.m:16``
(``f, Aineq, bineq, Aeq, beq, lb, ub`` and, when present, ``objcon``); ``preprocess`` is that script's conversion
(:22-82) to ``min c'x  s.t.  Ax = b, x >= 0``: inequality rows get slacks, finite upper bounds become rows
``x_j + w_j = ub_j - lb_j``, variables are shifted by their lower bound (``-inf`` becomes ``-1e6 - 1e8`` exactly as upstream:
:33 forms ``0 * -inf = NaN``, :34 sets the NaN to ``-1e6``, :35 adds ``-1e8``) and ``data.objcon = f'lb (+ objcon)``.  One documented
deviation: upstream multiplies ``f`` with the ORIGINAL ``prob.lb`` (:76), which is ``-inf``/NaN as soon as one variable is free; here the
shifted finite bound is used for those entries so that ``objcon`` stays a number.

Fixed and free MPS are both accepted (fields are split on white space; names must not contain blanks).  Sections: NAME,
ROWS (N/E/L/G), COLUMNS (MARKER lines ignored), RHS (a value on the objective row is minus the objective constant), RANGES,
BOUNDS (UP, LO, FX, FR, MI, PL, BV, LI, UI), ENDATA.  An ``OBJSENSE`` / ``OBJSENSE MAX`` section asking for maximisation is refused
(the reference pipeline minimises ``f'x`` whatever the file says; silently minimising a MAX model would return the wrong answer).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = ["mpsread", "preprocess", "load_standard_form"]


def mpsread(path: str) -> dict:
    rows, sense, obj = {}, [], None
    cols, entries = {}, []           # (row index, col index, value)
    rhs, ranges, bounds = {}, {}, []
    section = None
    with open(path) as fh:
        for raw in fh:
            if not raw.strip() or raw.lstrip().startswith("*"):
                continue
            if not raw[0].isspace():
                head = raw.split()
                section = head[0].upper()
                if section == "ENDATA":
                    break
                if section in ("OBJSENSE", "OBJSENSEMAX") and (section == "OBJSENSEMAX" or (len(head) > 1 and head[1].upper().startswith("MAX"))):
                    raise ValueError(f"{path}: OBJSENSE MAX is not supported (negate the objective and minimise)")
                continue
            t = raw.split()
            if section == "OBJSENSE":
                if t[0].upper().startswith("MAX"):
                    raise ValueError(f"{path}: OBJSENSE MAX is not supported (negate the objective and minimise)")
                continue
            if section == "ROWS":
                kind, name = t[0].upper(), t[1]
                if kind == "N":
                    if obj is None:
                        obj = name
                    rows[name] = -1 if name == obj else -2      # extra free rows are dropped
                else:
                    rows[name] = len(sense); sense.append(kind)
            elif section == "COLUMNS":
                if len(t) >= 3 and t[1].upper() == "'MARKER'":
                    continue
                cj = cols.setdefault(t[0], len(cols))
                for rn, val in zip(t[1::2], t[2::2]):
                    entries.append((rows[rn], cj, float(val)))
            elif section == "RHS":
                pairs = t[1:] if len(t) % 2 == 1 else t
                for rn, val in zip(pairs[0::2], pairs[1::2]):
                    rhs[rn] = float(val)
            elif section == "RANGES":
                pairs = t[1:] if len(t) % 2 == 1 else t
                for rn, val in zip(pairs[0::2], pairs[1::2]):
                    ranges[rn] = float(val)
            elif section == "BOUNDS":
                kind = t[0].upper()
                if kind in ("FR", "MI", "PL", "BV"):
                    # "FR BND X" / "FR X" / "FR BND X 0": the column is the last field that names a column, not blindly the last field
                    cn = t[2] if len(t) >= 3 and t[2] in cols else (t[1] if t[1] in cols else t[-1])
                    bounds.append((kind, cn, 0.0))
                else:
                    bounds.append((kind, t[-2], float(t[-1])))
    n, mrows = len(cols), len(sense)
    f = np.zeros(n)
    ri, ci, vi = [], [], []
    for r, cidx, v in entries:
        if r == -1:
            f[cidx] += v
        elif r >= 0:
            ri.append(r); ci.append(cidx); vi.append(v)
    Afull = sp.csr_matrix((vi, (ri, ci)), shape=(mrows, n))
    bfull = np.zeros(mrows)
    objcon = 0.0
    for rn, v in rhs.items():
        if rows.get(rn, -2) == -1:
            objcon = -v
        elif rows.get(rn, -2) >= 0:
            bfull[rows[rn]] = v
    lo_r, hi_r = np.full(mrows, -np.inf), np.full(mrows, np.inf)
    for i, k in enumerate(sense):
        if k == "E":
            lo_r[i] = hi_r[i] = bfull[i]
        elif k == "L":
            hi_r[i] = bfull[i]
        else:
            lo_r[i] = bfull[i]
    name_of = {v: k for k, v in rows.items() if v >= 0}
    for rn, R in ranges.items():
        i = rows[rn]
        k = sense[i]
        if k == "L":
            lo_r[i] = hi_r[i] - abs(R)
        elif k == "G":
            hi_r[i] = lo_r[i] + abs(R)
        elif R >= 0:
            hi_r[i] = lo_r[i] + R
        else:
            lo_r[i] = hi_r[i] + R
    eq = np.flatnonzero(lo_r == hi_r)
    ineq_rows, ineq_rhs = [], []
    for i in range(mrows):
        if lo_r[i] == hi_r[i]:
            continue
        if np.isfinite(hi_r[i]):
            ineq_rows.append(Afull[i]); ineq_rhs.append(hi_r[i])
        if np.isfinite(lo_r[i]):
            ineq_rows.append(-Afull[i]); ineq_rhs.append(-lo_r[i])
    lb, ub = np.zeros(n), np.full(n, np.inf)
    for kind, cn, v in bounds:
        j = cols[cn]
        if kind == "UP":
            ub[j] = v
            if v < 0 and lb[j] == 0:
                lb[j] = -np.inf
        elif kind == "LO":
            lb[j] = v
        elif kind == "FX":
            lb[j] = ub[j] = v
        elif kind == "FR":
            lb[j], ub[j] = -np.inf, np.inf
        elif kind == "MI":
            lb[j] = -np.inf
        elif kind == "PL":
            ub[j] = np.inf
        elif kind == "BV":
            lb[j], ub[j] = 0.0, 1.0
        elif kind == "LI":
            lb[j] = v
        elif kind == "UI":
            ub[j] = v
    return dict(f=f, Aeq=sp.csr_matrix(Afull[eq]), beq=lo_r[eq],
                Aineq=sp.vstack(ineq_rows, format="csr") if ineq_rows else sp.csr_matrix((0, n)), bineq=np.array(ineq_rhs, dtype=float),
                lb=lb, ub=ub, objcon=objcon, colnames=sorted(cols, key=cols.get), rownames=[name_of[i] for i in range(mrows)])


def preprocess(prob: dict) -> dict:
    """scripts/bench-lp/preprocess.m:22-82."""
    Aeq, Aineq = sp.csr_matrix(prob["Aeq"]), sp.csr_matrix(prob["Aineq"])
    beq, bineq = np.asarray(prob["beq"], float), np.asarray(prob["bineq"], float)
    m2, m1, n = Aineq.shape[0], Aeq.shape[0], Aeq.shape[1]
    plb, pub = np.asarray(prob["lb"], float), np.asarray(prob["ub"], float)
    # :33 (prob.lb > -inf) .* prob.lb is NaN (0 * -inf) where the bound is -inf; :34 turns the NaN into -1e6; :35 adds -1e8 there
    lb = np.where(plb == -np.inf, -1e6 + -1e8, plb)
    idxub = pub < np.inf
    m3 = int(idxub.sum())
    Dm = sp.identity(n, format="csr")[np.flatnonzero(idxub)]
    brhs = pub[idxub] - lb[idxub]
    A = sp.bmat([[Aeq, sp.csr_matrix((m1, m2)), sp.csr_matrix((m1, m3))],
                 [Aineq, sp.identity(m2, format="csr"), sp.csr_matrix((m2, m3))],
                 [Dm, sp.csr_matrix((m3, m2)), sp.identity(m3, format="csr")]], format="csc")
    b = np.concatenate([beq - Aeq @ lb, bineq - Aineq @ lb, brhs])
    c = np.concatenate([np.asarray(prob["f"], float), np.zeros(m2 + m3)])
    A.eliminate_zeros(); A.sort_indices()
    data = dict(A=A, b=b, c=c, m=A.shape[0], n=A.shape[1], lb=np.zeros(n + m2 + m3), lb_shift=lb, n_orig=n,
                objcon=float(np.asarray(prob["f"], float) @ plb if np.all(np.isfinite(plb)) else np.asarray(prob["f"], float) @ lb) + float(prob.get("objcon", 0.0)),
                sparsity=1 - A.nnz / (A.shape[0] * A.shape[1]), presolve=0)
    return data


def load_standard_form(path: str):
    """(A_csc, b, c, data) ready for ``abip(data, {'l': n}, params)``; the original variables are ``x[:n_orig] + lb_shift``."""
    data = preprocess(mpsread(path))
    return data["A"], data["b"], data["c"], data


def mpswrite(path: str, prob: dict, name: str = "ABIPLP") -> None:
    """Write ``prob`` (the fields ``mpsread`` returns) as a fixed-name free-format MPS file: equality rows E, inequality rows L,
    bounds LO/UP/FX/FR/MI, objective constant as the negated RHS of the objective row.  Inverse of ``mpsread`` up to row order."""
    f = np.asarray(prob["f"], float)
    Aeq, Ain = sp.csc_matrix(prob["Aeq"]), sp.csc_matrix(prob["Aineq"])
    beq, bin_ = np.asarray(prob["beq"], float), np.asarray(prob["bineq"], float)
    lb, ub = np.asarray(prob["lb"], float), np.asarray(prob["ub"], float)
    n = f.size
    with open(path, "w") as fh:
        fh.write(f"NAME {name}\nROWS\n N COST\n")
        for i in range(Aeq.shape[0]):
            fh.write(f" E E{i}\n")
        for i in range(Ain.shape[0]):
            fh.write(f" L L{i}\n")
        fh.write("COLUMNS\n")
        for j in range(n):
            if f[j] != 0.0:
                fh.write(f" X{j} COST {float(f[j])!r}\n")
            for M, tag in ((Aeq, "E"), (Ain, "L")):
                for q in range(M.indptr[j], M.indptr[j + 1]):
                    fh.write(f" X{j} {tag}{M.indices[q]} {float(M.data[q])!r}\n")
            if f[j] == 0.0 and Aeq.indptr[j] == Aeq.indptr[j + 1] and Ain.indptr[j] == Ain.indptr[j + 1]:
                fh.write(f" X{j} COST 0.0\n")
        fh.write("RHS\n")
        if float(prob.get("objcon", 0.0)) != 0.0:
            fh.write(f" RHS COST {-float(prob['objcon'])!r}\n")
        for i, v in enumerate(beq):
            if v != 0.0:
                fh.write(f" RHS E{i} {float(v)!r}\n")
        for i, v in enumerate(bin_):
            if v != 0.0:
                fh.write(f" RHS L{i} {float(v)!r}\n")
        fh.write("BOUNDS\n")
        for j in range(n):
            lo, hi = lb[j], ub[j]
            if lo == hi:
                fh.write(f" FX BND X{j} {float(lo)!r}\n")
            elif lo == -np.inf and hi == np.inf:
                fh.write(f" FR BND X{j}\n")
            else:
                if lo == -np.inf:
                    fh.write(f" MI BND X{j}\n")
                elif lo != 0.0:
                    fh.write(f" LO BND X{j} {float(lo)!r}\n")
                if hi != np.inf:
                    fh.write(f" UP BND X{j} {float(hi)!r}\n")
        fh.write("ENDATA\n")
