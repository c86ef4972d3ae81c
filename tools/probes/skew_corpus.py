"""A corpus whose BM25 weights are NOT spread evenly over the doc ids (SURVEY a-12: where per-block maxima matter): docs come
in runs of 65 536 (a level) that are either "short-doc" (length ~ 12) or "long-doc" (length ~ 400) regions, 1 region in 8
short; tf is clustered the same way (higher in the short regions).  The top-k of a union then lives in the short regions
and the block maxima of the long regions lie far below the list maxima."""
import numpy as np


def build(O, n_docs, densities, seed=7, region_log2=16):
    """region_log2: docs per region = 2^region_log2 (16 = one 65 536-doc level; 20 = runs of a million docs)"""
    rng = np.random.default_rng(seed)
    region = (np.arange(n_docs) >> region_log2)
    short = (region % 8) == 3
    lens = np.where(short, rng.integers(8, 17, n_docs), rng.integers(300, 501, n_docs)).astype(np.uint32)
    L = O.lib()
    dl = np.array([L.so_int_to_byte4(int(x)) for x in np.unique(lens)], np.uint8)
    lut = dict(zip(np.unique(lens).tolist(), dl.tolist()))
    doclen = np.vectorize(lut.get, otypes=[np.uint8])(lens)
    offs, docs, tfs = [0], [], []
    for d in densities:
        m = rng.random(n_docs) < d
        ids = np.nonzero(m)[0].astype(np.uint32)
        tf = np.where(short[ids], rng.integers(2, 9, len(ids)), rng.integers(1, 3, len(ids))).astype(np.uint16)
        docs.append(ids); tfs.append(tf); offs.append(offs[-1] + len(ids))
    return doclen, np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)
