"""Stress of the one-launch path (csrc/bm25_small.hip; VERDICT r5 "next" 3): >= 10^5 launches back to back from concurrent callers --
random batch sizes (1 .. 64 queries a request, requests coalesced behind the ABI into launches of up to 64), k, result types, unions and
intersections of 1 .. 4 terms of both tiers, NOT terms, with and without tombstones, one coalescer lane and two -- and EVERY answer compared
bit for bit with the staged pipeline's answer to the same request (computed once per distinct request, through the device-pointer entry
point, which never takes the one-launch path).  What this hunts: state one launch leaves for the next (thresholds, best keys, arrival and
match counters are zeroed by whoever consumed them, never by a memset), the hand-over of the partition lists between workgroups (relaxed
agent-scope accesses ordered by s_waitcnt; tools/probes/small_litmus.hip is the same protocol bare), and two lanes' launches interleaving
on the shard's stream into one workspace.  SS_STRESS_LAUNCHES overrides the number of launches."""
import os
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_LAUNCHES = int(os.environ.get("SS_STRESS_LAUNCHES", 100_000))
N_DOCS = 200_000
DENSE = [0.2, 0.09, 0.05, 0.02, 0.011, 0.004]
RARE = [3000, 1400, 800, 300, 120, 60, 25, 9, 3, 1]


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _image(S, O, rng):
    lens = np.clip(np.round(np.exp(np.log(120) + 0.6 * rng.standard_normal(N_DOCS))), 8, 2000).astype(np.int64)
    lut = {int(x): int(O.lib().so_int_to_byte4(int(x))) for x in np.unique(lens)}
    dl = np.array([lut[int(x)] for x in lens], np.uint8)
    offs, docs, tfs = [0], [], []
    for n in [int(x * N_DOCS) for x in DENSE] + RARE:
        docs.append(np.sort(rng.choice(N_DOCS, n, replace=False)).astype(np.uint32))
        tfs.append(rng.geometric(0.5, n).clip(1, 200).astype(np.uint16))
        offs.append(offs[-1] + n)
    offs, docs, tfs = np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)
    nd = len(DENSE)
    e = int(offs[nd])
    sh = S.Shard(0)
    sh.upload_lexical(N_DOCS, dl, offs[:nd + 1], docs[:e], tfs[:e])
    assert sh.append_sparse(offs[nd:] - offs[nd], docs[e:], tfs[e:]) == nd
    return sh


def _request(S, sh, rng, tiered):
    """one caller's request: (queries, k, result type, ops mask of the device-pointer form)"""
    nd, nt_all = len(DENSE), len(DENSE) + len(RARE)
    D, R = list(range(nd)), list(range(nd, nt_all))
    nq = int(rng.choice([1, 1, 1, 2, 3, 5, 8, 13, 16, 24, 32, 48, 64]))
    k = int(rng.choice([1, 10, 32] if tiered else [1, 10, 32, 64, 100, 128]))
    rt = S.ResultType.TopkCount if rng.random() < 0.5 else S.ResultType.Topk
    lists, nots, types = [], [], []
    np_max = nn_max = 0
    for _ in range(nq):
        is_and = rng.random() < 0.4
        nt = int(rng.integers(1, 5))
        ns = int(rng.integers(0, min(nt, 2) + 1)) if tiered else 0
        n_not = int(rng.choice([0, 0, 1, 2]))
        n_not = min(n_not, nd - (nt - ns))
        sp = [int(x) for x in rng.choice(R, ns, replace=False)]
        de = [int(x) for x in rng.choice(D, nt - ns + n_not, replace=False)]
        t = sp + de[:nt - ns]
        rng.shuffle(t)
        lists.append([int(x) for x in t])
        nots.append(de[nt - ns:])  # dense NOT terms (a union's sparse NOT list is outside the one-launch shape)
        types.append(S.QueryType.Intersection if is_and else S.QueryType.Union)
        np_max, nn_max = max(np_max, nt), max(nn_max, n_not)
    q = sh.make_queries(lists, types, nots)
    ops = (1 << 28) if tiered else (3 | ((np_max + nn_max) << 8) | (np_max << 16) | (nn_max << 24))
    return q, k, rt, ops


def _staged(S, sh, q, k, rt, ops):
    from test_gpu_small_batch import _dev_search
    return _dev_search(S, sh, q, k, rt, ops)


def _differs(S, got, ref, rt):
    if not np.array_equal(got[2], ref[2]):
        return "counts"
    for i in range(len(got[2])):
        c = int(got[2][i])
        if not (np.array_equal(got[0][i][:c], ref[0][i][:c]) and np.array_equal(got[1][i][:c], ref[1][i][:c])):
            return "query %d" % i
    if rt == S.ResultType.TopkCount and not np.array_equal(got[3], ref[3]):
        return "totals"
    return None


@pytest.mark.parametrize("lanes,share", [("1", 0.4), ("2", 0.6)])
def test_one_launch_under_concurrent_callers(S, O, lanes, share):
    target = max(200, int(N_LAUNCHES * share))
    old = os.environ.get("SS_COALESCE_LANES")
    os.environ["SS_COALESCE_LANES"] = lanes  # (read when the shard is created)
    rng = np.random.default_rng(1234 + int(lanes))
    try:
        sh = _image(S, O, rng)
    finally:
        if old is None:
            os.environ.pop("SS_COALESCE_LANES", None)
        else:
            os.environ["SS_COALESCE_LANES"] = old
    try:
        gone = np.unique(rng.integers(0, N_DOCS, N_DOCS // 40)).astype(np.uint64)
        done = 0
        for phase, deleted in enumerate((False, True)):
            sh.set_deleted(gone if deleted else [])
            pool = []
            for j in range(96):
                q, k, rt, ops = _request(S, sh, rng, tiered=(j % 3 != 0))
                pool.append((q, k, rt, _staged(S, sh, q, k, rt, ops)))
            phase_target = target // 2 if phase == 0 else target - done
            start = sh.one_launch_batches()
            errors, calls = [], [0]
            stop = threading.Event()
            deadline = time.time() + 600.0

            def caller(seed):
                r = np.random.default_rng(seed)
                n = 0
                while not stop.is_set():
                    j = int(r.integers(0, len(pool)))
                    q, k, rt, ref = pool[j]
                    try:
                        got = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                    except Exception as e:  # noqa: BLE001
                        errors.append((j, repr(e)))
                        stop.set()
                        return
                    what = _differs(S, got, ref, rt)
                    if what:
                        errors.append((j, what, len(q), k, int(rt)))
                        if len(errors) > 8:
                            stop.set()
                    n += 1
                    if (n & 63) == 0 and (sh.one_launch_batches() - start >= phase_target or time.time() > deadline):
                        stop.set()
                calls[0] += n

            threads = [threading.Thread(target=caller, args=(1000 * phase + t,)) for t in range(8)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            launched = sh.one_launch_batches() - start
            done += launched
            assert not errors, ("answers differ from the staged pipeline's", lanes, deleted, errors[:8])
            assert launched >= phase_target, ("too few one-launch batches before the deadline", launched, phase_target, calls[0])
        # and the shard is in order afterwards: the per-query state is clean (one more request of every kind, alone)
        for q, k, rt, ref in pool[:6]:
            assert _differs(S, sh.search_lexical_batch(q, k, rt, reference_shortcuts=False), ref, rt) is None
        print("one-launch batches: %d (lanes %s)" % (done, lanes))
    finally:
        sh.close()


def test_pool_rows_change_hands_under_both_lanes(S, O):
    """A rationed vocabulary (rows for the longest lists + a pool built on demand, ss_bm25_set_probe_budget) under concurrent
    single-query callers, two coalescer lanes: a lane leader whose batch names a row-less list builds the row on the shard's stream while
    the OTHER lane's kernel may be in flight on its own stream -- the lane streams are drained before a row changes hands and the next lane
    launch waits for the fill (ss_api.hip ssi_bm25_ensure_probe_rows).  Every answer equals the unrationed shard's."""
    from test_gpu_pruned import _corpus
    n_docs = 150_000
    dfs = [0.2 / (1 + 0.35 * i) for i in range(56)]
    dl, offs, docs, tfs = _corpus(O, n_docs, dfs, 23)
    old = os.environ.get("SS_COALESCE_LANES")
    os.environ["SS_COALESCE_LANES"] = "2"
    try:
        full, part = S.Shard(0), S.Shard(0)
    finally:
        if old is None:
            os.environ.pop("SS_COALESCE_LANES", None)
        else:
            os.environ["SS_COALESCE_LANES"] = old
    try:
        full.upload_lexical(n_docs, dl, offs, docs, tfs)
        n_sub = (n_docs + 4095) // 4096
        part.set_probe_budget((40 + 1) * n_sub * 64 * 12)   # 40 rows: 30 fixed (the longest lists) + a pool of 10 for 26 row-less lists
        part.upload_lexical(n_docs, dl, offs, docs, tfs)
        rng = np.random.default_rng(77)
        pool = []
        for j in range(160):
            nt = int(rng.integers(1, 4))
            t = [int(rng.integers(30, 56))] + [int(x) for x in rng.choice(30, nt - 1, replace=False)]  # one row-less list per query
            qt = S.QueryType.Intersection if j % 3 == 0 else S.QueryType.Union
            rt = S.ResultType.TopkCount if j % 2 else S.ResultType.Topk
            ref = full.search_lexical_batch(full.make_queries([t], qt), 10, rt, reference_shortcuts=False)
            pool.append((part.make_queries([t], qt), rt, ref))
        start = part.one_launch_batches()
        errors, stop = [], threading.Event()

        def caller(seed):
            r = np.random.default_rng(seed)
            for _ in range(1500):
                if stop.is_set():
                    return
                j = int(r.integers(0, len(pool)))
                q, rt, ref = pool[j]
                try:
                    got = part.search_lexical_batch(q, 10, rt, reference_shortcuts=False)
                except Exception as e:  # noqa: BLE001
                    errors.append((j, repr(e)))
                    stop.set()
                    return
                what = _differs(S, got, ref, rt)
                if what:
                    errors.append((j, what, int(rt)))
                    if len(errors) > 8:
                        stop.set()

        threads = [threading.Thread(target=caller, args=(500 + t,)) for t in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, ("answers differ from the unrationed shard's", errors[:8])
        assert part.one_launch_batches() - start > 1000   # (the path under test ran: one-launch batches, not the staged pipeline)
    finally:
        full.close()
        part.close()
