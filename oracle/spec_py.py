"""docs/SPEC.md §1-§4 once more, literally, in pure Python (dicts, sets, string slices) — TEST INFRASTRUCTURE ONLY.

A second, independent restatement of the same normative text as oracle/nb_oracle.cpp, written from docs/SPEC.md and not from the
C++ source, for small cases: tests/test_oracle_kat.py runs both on random libraries and reads and requires identical results,
so that a misreading of the spec would have to be made twice, in two different formulations, to go unnoticed. PARITY UNPINNED
against the Rust nimble-aligner all the same (docs/SPEC.md header): this checks the oracle against the spec, not the spec
against the reference. Only tests/ may import it.
"""
TOO_SHORT, NO_HIT, SCORE_THRESHOLD, SCORE_PERCENT, EMPTY_EQCLASS, MULTIPLE_MATCH, INVALID_PAIR, STRAND, SCORE_FILTER = 1, 2, 3, 4, 5, 6, 7, 8, 9
CALLED = 0
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _norm(s):
    """§1: A C G T case-insensitive, everything else is N."""
    return "".join(c if c in "ACGT" else "N" for c in s.upper())


def revcomp(s):
    return "".join(_COMP.get(c, "N") for c in reversed(_norm(s)))


class SpecIndex:
    """§2: stranded coloured de Bruijn graph with edge-centric extensions and the local unitig-interior rule."""

    def __init__(self, sequences, k):
        self.k = k
        self.colour, self.rext, self.lext = {}, {}, {}
        for i, raw in enumerate(sequences):
            s = _norm(raw)
            for p in range(len(s) - k + 1):
                x = s[p:p + k]
                if "N" in x:
                    continue
                self.colour.setdefault(x, set()).add(i)
                self.rext.setdefault(x, set())
                self.lext.setdefault(x, set())
                if p + k < len(s) and s[p + k] != "N":          # the (K+1)-mer x.b occurs
                    self.rext[x].add(s[p + k])
                if p >= 1 and s[p - 1] != "N":                  # the (K+1)-mer a.x occurs
                    self.lext[x].add(s[p - 1])
        self.colour = {x: frozenset(c) for x, c in self.colour.items()}

    def r_internal(self, x):
        if len(self.rext[x]) != 1:
            return False
        y = x[1:] + next(iter(self.rext[x]))
        return len(self.lext[y]) == 1 and self.colour[y] == self.colour[x] and y != x

    def l_internal(self, y):
        if len(self.lext[y]) != 1:
            return False
        return self.r_internal(next(iter(self.lext[y])) + y[:-1])

    def map(self, raw, budget):
        """§3. Returns ("none", reason) or ("ok", eq frozenset, cov, mm)."""
        s, K = _norm(raw), self.k
        L = len(s)
        if L < K:
            return ("none", TOO_SHORT)
        last = L - K

        def seed(b):
            for p in range(b, last + 1, 3):
                x = s[p:p + K]
                if "N" not in x and x in self.colour:
                    return p
            return None

        p = seed(0)
        if p is None:
            return ("none", NO_HIT)
        x = s[p:p + K]
        cov, mm, visited = 0, 0, []
        if p >= L // 5:                                         # left extension from the first hit
            y, q = x, p - 1
            while q >= 0:
                c = s[q]
                if c != "N" and c in self.lext[y]:
                    y = c + y[:-1]
                elif self.l_internal(y) and mm < budget:
                    y = next(iter(self.lext[y])) + y[:-1]
                    mm += 1
                else:
                    break
                cov += 1
                visited.append(self.colour[y])
                q -= 1
        while True:                                             # forward walk
            cov += K
            visited.append(self.colour[x])
            b = p + K
            while b < L:
                c = s[b]
                if c != "N" and c in self.rext[x]:
                    x = x[1:] + c
                elif self.r_internal(x) and mm < budget:
                    x = x[1:] + next(iter(self.rext[x]))
                    mm += 1
                else:
                    break
                cov += 1
                visited.append(self.colour[x])
                b += 1
            if b > last:
                break
            p = seed(b)
            if p is None:
                break
            x = s[p:p + K]
        eq = visited[0]
        for v in visited[1:]:
            eq = eq & v
        return ("ok", eq, cov, mm)


def _validate(m, L, cfg):
    """§4 checks 1-4 -> (valid, reason, cov, eq)."""
    if m[0] == "none":
        return (False, m[1], 0, frozenset())
    _, eq, cov, _ = m
    if cov < cfg["score_threshold"]:
        return (False, SCORE_THRESHOLD, cov, frozenset())
    if float(cov) / float(L) < cfg["score_percent"]:
        return (False, SCORE_PERCENT, cov, frozenset())
    if not eq:
        return (False, EMPTY_EQCLASS, cov, frozenset())
    if cfg["discard_multiple_matches"] and len(eq) != 1:
        return (False, MULTIPLE_MATCH, cov, frozenset())
    return (True, CALLED, cov, eq)


def _orientation(a1, a2, cfg):
    """§4 per orientation -> (valid, reason, score, mates, set)."""
    if a2 is None:
        return (a1[0], a1[1], a1[2] if a1[0] else 0, 1 if a1[0] else 0, a1[3])
    if a1[0] and a2[0]:
        both, either = a1[3] & a2[3], a1[3] | a2[3]
        if cfg["intersect_level"] == 0:
            chosen = either
        elif cfg["intersect_level"] == 1:
            chosen = both if both else either
        else:
            if not both:
                return (False, EMPTY_EQCLASS, 0, 0, frozenset())
            chosen = both
        return (True, CALLED, a1[2] + a2[2], 2, chosen)
    if a1[0] or a2[0]:
        if cfg["require_valid_pair"]:
            return (False, INVALID_PAIR, 0, 0, frozenset())
        a = a1 if a1[0] else a2
        return (True, CALLED, a[2], 1, a[3])
    return (False, max(a1[1], a2[1]), 0, 0, frozenset())


def align_template(index, cfg, r1, r2=None):
    """§4 -> dict(reason, score, orientation, mates, set (sorted list), cov (r1 fwd, r1 rev, r2 fwd, r2 rev))."""
    M = cfg["num_mismatches"]
    m1f, m1r = index.map(r1, M), index.map(revcomp(r1), M)
    cov = [m1f[2] if m1f[0] == "ok" else 0, m1r[2] if m1r[0] == "ok" else 0, 0, 0]
    a1f, a1r = _validate(m1f, len(r1), cfg), _validate(m1r, len(r1), cfg)
    a2f = a2r = None
    if r2 is not None:
        m2f, m2r = index.map(r2, M), index.map(revcomp(r2), M)
        cov[2], cov[3] = (m2f[2] if m2f[0] == "ok" else 0), (m2r[2] if m2r[0] == "ok" else 0)
        a2f, a2r = _validate(m2f, len(r2), cfg), _validate(m2r, len(r2), cfg)
    F = list(_orientation(a1f, a2r, cfg))        # F = {map(r1), map(rc(r2))}
    R = list(_orientation(a1r, a2f, cfg))        # R = {map(rc(r1)), map(r2)}
    strand = cfg["strand_filter"]                # 0 unstranded / none, 1 fiveprime (F only), 2 threeprime (R only)
    if F[0] and strand == 2:
        F[0], F[1] = False, STRAND
    if R[0] and strand == 1:
        R[0], R[1] = False, STRAND
    out = {"reason": CALLED, "score": 0, "orientation": 0, "mates": 0, "set": [], "cov": cov}
    if F[0] and R[0]:
        pick, orient = (R, 1) if R[2] > F[2] else (F, 0)
    elif F[0]:
        pick, orient = F, 0
    elif R[0]:
        pick, orient = R, 1
    else:
        out["reason"] = max(F[1], R[1])
        return out
    out.update(score=pick[2], orientation=orient, mates=pick[3])
    if pick[2] < cfg["score_filter"]:
        out["reason"] = SCORE_FILTER
        return out
    out["set"] = sorted(pick[4])
    return out
