"""Golden key bodies in the reference's posting-list block format (single field and several fields), written once with
oracle/ref_format.py (restatement of the reference's writers) and committed with their expected decode, so that neither
side of the format code can drift unnoticed.  Run from the repo root:  python tests/golden/make_ref_format_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_format as RF  # noqa: E402

rng = np.random.default_rng(20260922)
out = {}
# single field: array / bitmap / rle containers, embedded and VINT pointers, a pivot inside the list
cases = [("array", np.sort(rng.choice(65536, 300, replace=False)), 12, 32768),
         ("bitmap", np.sort(rng.choice(65536, 4500, replace=False)), 3, 32768),
         ("rle", np.arange(1000, 3000), 5, 32768),
         ("pivot", np.sort(rng.choice(65536, 700, replace=False)), 9, 128)]
for name, docs, tf_hi, limit in cases:
    tfs = rng.integers(1, tf_hi + 1, size=len(docs))
    bid, ctp, cnt, pivot, body = RF.encode_term(docs, tfs, rng, base_bytes=bytes(range(17)), positions_limit=limit)[0]
    out[f"s_{name}_head"] = np.array([bid, ctp, cnt, pivot], np.uint64)
    out[f"s_{name}_body"] = np.frombuffer(body, np.uint8)
    out[f"s_{name}_docs"] = docs.astype(np.uint16)
    out[f"s_{name}_tfs"] = tfs.astype(np.uint16)
# three fields, longest field 1
docs = np.sort(rng.choice(65536, 600, replace=False))
d, f, t = [], [], []
for doc in docs:
    fs = np.sort(rng.choice(3, size=int(rng.integers(1, 4)), replace=False)) if rng.random() < 0.6 else [1]
    for x in fs:
        d.append(int(doc)); f.append(int(x)); t.append(int(rng.integers(1, 8)))
bid, ctp, cnt, pivot, body = RF.encode_term_fields(np.array(d), np.array(f), np.array(t), 3, 1, rng, positions_limit=256, max_gap=25)[0]
out["m_head"] = np.array([bid, ctp, cnt, pivot, 3, 1], np.uint64)
out["m_body"] = np.frombuffer(body, np.uint8)
out["m_docs"] = np.array(d, np.uint16)
out["m_fields"] = np.array(f, np.uint8)
out["m_tfs"] = np.array(t, np.uint16)
# an n-gram key (trigram type: three component tfs in front of every record, no embedded pointers), 3-byte pointers inside
docs = np.sort(rng.choice(65536, 500, replace=False))
counts = rng.integers(1, 4, size=500)
comp = rng.integers(1, 400, size=(500, 3))
comp[:20] = rng.integers(128, 20000, size=(20, 3))
bid, ctp, cnt, pivot, body = RF.encode_term(docs, counts, rng, base_bytes=bytes(range(9)), positions_limit=900, ngram_tfs=comp)[0]
out["g_head"] = np.array([bid, ctp, cnt, pivot, 3], np.uint64)
out["g_body"] = np.frombuffer(body, np.uint8)
out["g_docs"] = docs.astype(np.uint16)
out["g_tfs"] = comp.astype(np.uint16)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_format.npz"), **out)
print({k: v.shape for k, v in out.items()})
