#!/usr/bin/env python3
"""Prints the length table of the register-resident mixed-radix kernel (basic_dsp_amd/csrc/mixed_radix_reg3.h): every
n = R0 R1 R2 <= 4096 that is not a power of two, radices out of {4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25}; a factorisation with
n / min radix <= 256 threads per transform where one exists (else <= 512), then the largest smallest radix, then the smallest
largest one; largest radix first."""
import itertools
RADICES = [4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25]
best = {}
for t in itertools.combinations_with_replacement(RADICES, 3):
    n = t[0] * t[1] * t[2]
    if n > 4096 or n < 300 or n & (n - 1) == 0 or n // min(t) > 512:
        continue
    key = (n // min(t) <= 256, min(t), -max(t))
    if n not in best or key > best[n][0]:
        best[n] = (key, tuple(sorted(t, reverse=True)))
for n in sorted(best):
    a, b, c = best[n][1]
    print("    BDSP_REG3(%d, %d, %d, %d)%s" % (n, a, b, c, "" if n <= 2048 else "   // f32 only"))
