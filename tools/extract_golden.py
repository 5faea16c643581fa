#!/usr/bin/env python3
"""Transcribe the reference's known-answer vectors into tests/golden/reference_kats.json.

Runs only in the authoring container (it reads /root/reference).  What it copies is DATA: the
numeric literal arrays that the reference's own unit/integration tests hold for the hot path
(SURVEY.md section 8c), each tagged with the test function and file:line it came from.  The
scenario each vector belongs to (inputs, operation) is re-stated by hand in
tests/test_oracle_golden.py; no reference source text is stored.

Usage: python tools/extract_golden.py   (rewrites tests/golden/reference_kats.json)
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "reference_kats.json")

# (file, [test function names]) -- every array literal inside each function body is kept, in order.
SOURCES = [
    ("tests/time_freq_test.rs", ["fft_vector64", "windowed_fft_vector64"]),
    ("vector/src/window_functions.rs", ["triangular_window32_test", "hamming_window32_test",
                                        "blackmanharris_window32_test",
                                        "rectangular_window32_test"]),
    ("vector/src/conv_types.rs", ["raised_cosine_test", "sinc_test", "sinc_freq_test",
                                  "freq_test"]),
    ("vector/src/vector_types/mod.rs", ["swap_halves_even_test", "swap_halves_odd_foward_test",
                                        "swap_halves_odd_inverse_test"]),
    ("vector/src/vector_types/general/data_reorganization.rs", [
        "swap_halves_real_even_test", "swap_halves_real_odd_test",
        "swap_halves_complex_even_test", "swap_halves_complex_odd_test",
        "zero_pad_end_test", "zero_pad_surround_test", "zero_pad_center_test",
        "zero_pad_b_center_test", "zero_pad_surround_odd_signal_test", "zero_pad_b_end_test",
        "zero_pad_b_surround_test", "zero_pad_b_surround_odd_signal_test",
        "zero_pad_surround_overlap_test", "zero_pad_center_overlap_test",
        "zero_interleave_test", "zero_interleave_even_test", "zero_interleave_b_test",
        "zero_interleave_complex_test", "zero_interleave_b_complex_test"]),
    ("vector/src/vector_types/time_freq/convolution.rs", [
        "convolve_real_time_and_time32", "convolve_complex_time_and_time32",
        "convolve_complex_vectors32", "wrapping_iterator", "wrapping_rev_iterator",
        "vector_conv_vs_freq_multiplication", "shift_left_by_1_as_conv",
        "shift_left_by_1_as_conv_shorter", "overlap_discard_test",
        "convolve_complex_freq_and_freq32", "convolve_complex_freq_and_freq_even32"]),
    ("vector/src/vector_types/time_freq/interpolation.rs", [
        "interpolatei_sinc_test", "interpolate_sinc_even_test", "interpolate_sinc_odd_test",
        "interpolatei_rc_test", "interpolatef_by_integer_sinc_even_test",
        "interpolatef_by_integer_sinc_odd_test", "interpolatef_by_fractional_sinc_test",
        "interpolate_by_fractional_sinc_test", "interpolatef_delayed_sinc_test",
        "interpolate_delayed_sinc_test", "decimatei_test", "decimate_with_interpolate_test"]),
    ("vector/src/vector_types/time_freq/correlation.rs", ["time_correlation_test",
                                                          "time_correlation_test2"]),
    ("vector/src/vector_types/time_freq/real_interpolation.rs", [
        "hermit_spline_test", "hermit_spline_test_linear_increment", "linear_test"]),
]

NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"
ARRAY = re.compile(r"\[\s*((?:" + NUM + r"\s*,\s*)*" + NUM + r")\s*,?\s*\]")
REPEAT = re.compile(r"\[\s*(" + NUM + r")\s*;\s*(\d+)\s*\]")


def function_body(lines, name):
    """Return (first_line_no, text) of `fn name(` up to the brace that closes it."""
    for i, line in enumerate(lines):
        if re.search(r"\bfn\s+" + re.escape(name) + r"\s*\(", line):
            depth, started, body = 0, False, []
            for j in range(i, len(lines)):
                body.append(lines[j])
                depth += lines[j].count("{") - lines[j].count("}")
                if "{" in lines[j]:
                    started = True
                if started and depth == 0:
                    return i + 1, j + 1, "".join(body)
    return None


def main():
    out = {}
    for rel, names in SOURCES:
        path = os.path.join(REF, rel)
        with open(path) as f:
            lines = f.readlines()
        for name in names:
            fb = function_body(lines, name)
            if fb is None:
                print("warning: %s not found in %s" % (name, rel), file=sys.stderr)
                continue
            first, last, text = fb
            arrays = []
            for m in re.finditer(r"\[[^\[\]]*\]", text):
                lit = m.group(0)
                m1 = ARRAY.fullmatch(lit)
                m2 = REPEAT.fullmatch(lit)
                if m1:
                    vals = [float(v) for v in re.findall(NUM, m1.group(1))]
                    if len(vals) >= 2:
                        arrays.append(vals)
                elif m2:
                    arrays.append([float(m2.group(1))] * int(m2.group(2)))
            out[name] = {"source": "%s:%d-%d" % (rel, first, last), "arrays": arrays}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s: %d tests, %d arrays" % (os.path.normpath(OUT), len(out),
                                             sum(len(v["arrays"]) for v in out.values())))


if __name__ == "__main__":
    main()
