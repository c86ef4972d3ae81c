"""Regenerates tests/golden/generator.npz from the CPU oracle's counter-based generator.
Run from the repo root: python tests/golden/make_generator_golden.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O

d, t = O.lex_term(4000, O.term_thresholds()[4000], 50_000)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "generator.npz"),
         lex_docs=d[:64], lex_tfs=t[:64], doclen=O.lex_doclen(4096)[:256], vec=O.vec_gen(O.VEC_SEED, 5, 2, 16))
print("ok", len(d))
