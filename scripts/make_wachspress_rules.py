#!/usr/bin/env python3
"""Writes the table of integration rules on the unit triangle that the Wachspress basis integrals may use:
the Dunavant rules (orders 1-10 and 12) and the Fekete rules (orders 1-6, 8, 9) with points and weights to exactly
the digits the reference carries, so that the basis integrals round as the reference's do.

The numbers are data: they are read from the literal lists of get_integration_factors_dunavant / _fekete
(/root/reference/src/shared/mpas_seaice_velocity_solver_wachspress.F:1401-1941) and written, as numbers only, into
one C header that the library and the oracle each keep a copy of:

    python scripts/make_wachspress_rules.py /root/reference/src/shared/mpas_seaice_velocity_solver_wachspress.F

tests/test_wachspress_rules_cpu.py checks the table against mathematics, not against the reference: weights sum
to one and every rule integrates the monomials up to its degree."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NUM = re.compile(r"(-?\d+\.\d*(?:[eE][-+]?\d+)?)_RKIND")


def parse(path):
    text = open(path).read().splitlines()
    rules = []
    kind = None
    orders = None
    cur = None
    vec = None
    for line in text:
        s = line.strip()
        m = re.match(r"subroutine get_integration_factors_(\w+)\(", s)
        if m:
            kind = m.group(1) if m.group(1) in ("dunavant", "fekete") else None
            continue
        if kind is None:
            continue
        if s.startswith("end subroutine"):
            kind = None
            continue
        m = re.match(r"case\s*\(\s*(\d+)(?::(\d+))?\s*\)", s)
        if m:
            lo = int(m.group(1))
            hi = int(m.group(2)) if m.group(2) else lo
            orders = list(range(lo, hi + 1))
            cur = {"kind": kind, "orders": orders, "u": [], "v": [], "weights": []}
            rules.append(cur)
            vec = None
            continue
        if cur is None:
            continue
        m = re.match(r"(u|v|weights)\s*=\s*\(/", s)
        if m:
            vec = m.group(1)
        if vec:
            cur[vec] += NUM.findall(s)
            if "/)" in s:
                vec = None
    for r in rules:
        assert len(r["u"]) == len(r["v"]) == len(r["weights"]) > 0, r
    return rules


def main():
    rules = parse(sys.argv[1])
    out = []
    out.append("/* Integration rules on the unit triangle for the Wachspress basis integrals: D. A. Dunavant, Int. J. Num.")
    out.append(" * Meth. Engng 21 (1985) 1129-1148, orders 1-10 and 12, and the Fekete points of M. A. Taylor, B. A. Wingate,")
    out.append(" * R. E. Vincent, SIAM J. Numer. Anal. 38 (2000) 1707-1720, orders 1-6, 8, 9; points and weights to the digits")
    out.append(" * the reference carries (mpas_seaice_velocity_solver_wachspress.F:1401-1941), normalisation factor 2 for all")
    out.append(" * of them (:1426, :1751).  Numbers only; written by scripts/make_wachspress_rules.py.")
    out.append(" * kind: 1 dunavant, 3 fekete.  A rule's points are entries [first, first + n) of the three arrays. */")
    out.append("#ifndef SIH_WACHSPRESS_RULES_H")
    out.append("#define SIH_WACHSPRESS_RULES_H")
    out.append("typedef struct { int kind, order, n, first; } sih_rule_entry;")
    entries, U, V, W = [], [], [], []
    for r in rules:
        first = len(U)
        U += r["u"]
        V += r["v"]
        W += r["weights"]
        for o in r["orders"]:
            entries.append((1 if r["kind"] == "dunavant" else 3, o, len(r["u"]), first))
    out.append("static const sih_rule_entry sih_rule_table[] = {")
    out.append("    " + ", ".join("{%d, %d, %d, %d}" % e for e in entries))
    out.append("};")
    out.append("#define SIH_RULE_COUNT %d" % len(entries))

    def arr(name, vals):
        out.append("static const double %s[%d] = {" % (name, len(vals)))
        for i in range(0, len(vals), 6):
            out.append("    " + ", ".join(vals[i:i + 6]) + ",")
        out.append("};")
    arr("sih_rule_u", U)
    arr("sih_rule_v", V)
    arr("sih_rule_w", W)
    out.append("#endif")
    text = "\n".join(out) + "\n"
    for dest in ("mpas_seaice_amd/csrc/wachspress_rules.h", "oracle/wachspress_rules.h"):
        with open(os.path.join(ROOT, dest), "w") as fh:
            fh.write(text)
    print(len(entries), "rules,", len(U), "points")


if __name__ == "__main__":
    main()
