#!/usr/bin/env python
"""Copies the known-answer vectors of the reference's own tests for this path into JSON fixtures:
tests/lattices/example_svp_in -> example_svp_out (tests/test_svp.cpp:54-103: a shortest vector) and
example_dsvp_in -> example_dsvp_out (:214-262: coefficients of a shortest DUAL vector).
usage: make_kat_fixtures.py /root/reference tests/golden"""
import json
import re
import sys

ref, out = sys.argv[1], sys.argv[2]


def read_matrix(path):
    txt = open(path).read()
    rows = re.findall(r"\[([^\[\]]*)\]", txt)
    return [[int(x) for x in r.split()] for r in rows if r.strip()]


for name in ("svp", "dsvp"):
    A = read_matrix("%s/tests/lattices/example_%s_in" % (ref, name))
    b = read_matrix("%s/tests/lattices/example_%s_out" % (ref, name))[0]
    json.dump({"desc": "reference KAT tests/lattices/example_%s_in/out (tests/test_svp.cpp)" % name,
               "d": len(A), "n": len(A[0]), "basis": A, "answer": b},
              open("%s/kat_%s.json" % (out, name), "w"))
    print(name, len(A), len(A[0]), len(b))
