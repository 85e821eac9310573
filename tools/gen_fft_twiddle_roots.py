#!/usr/bin/env python3
"""cos(pi / 2^i), sin(pi / 2^i) for i = 0 .. 10, correctly rounded to double: the roots complex_fft's stages start
their twiddle recurrences from (the reference keeps them as an 18-digit table, common/alcomplex.cpp gArgAngle).
Computed here from scratch with 60-digit decimal arithmetic (Taylor series); prints the C initialiser that
csrc/effects_api.hip carries.  Run: python tools/gen_fft_twiddle_roots.py"""
from decimal import Decimal, getcontext

getcontext().prec = 60
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494")


def cos_sin(x):
    c, s, term_c, term_s = Decimal(1), x, Decimal(1), x
    for k in range(1, 40):
        term_c = -term_c * x * x / ((2 * k - 1) * (2 * k))
        term_s = -term_s * x * x / ((2 * k) * (2 * k + 1))
        c += term_c
        s += term_s
    return c, s


rows = []
for i in range(11):
    c, s = cos_sin(PI / (1 << i))
    if i == 0:
        c, s = Decimal(-1), Decimal(0)
    if i == 1:
        c, s = Decimal(0), Decimal(1)
    rows.append((float(c), float(s)))          # float(Decimal) rounds correctly
for c, s in rows:
    print("    {%s, %s}," % (c.hex(), s.hex()))
