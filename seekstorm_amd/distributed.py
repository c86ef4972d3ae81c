"""Multi-process (one process per GPU) leg of the search planner: index shards partition across ranks
(doc g -> shard g % S, local id g // S, index.rs:5284); every rank answers the query batch on its shard and ONE
all-gather of the per-shard top-(offset+length) lists replaces the reference's await-all-JoinHandles + Vec::append
(search.rs:1875-1917); the merge (sort desc, offset, truncate -- search.rs:2098-2119; RRF for Hybrid,
search.rs:1962-2035) then runs identically on every rank.

Backends: `nccl` (= RCCL over xGMI) with device tensors and ss_topk_merge_dev on the GPU box; `gloo` with host tensors
and ss_merge_results, used by the CPU tests (world_size 2).  Payload per rank is n_queries * k * 8 bytes: latency-,
not bandwidth-bound, so a single one-shot all-gather per list is used (no bucketing, no ring tuning).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _native as N
from .search import SearchMode, merge_results


def all_gather_topk(doc: torch.Tensor, score: torch.Tensor, count: torch.Tensor):
    """doc/score: [nq, k] of this rank's shard (local ids, sorted desc); count: [nq].  Returns [S, nq, k] x2, [S, nq]."""
    world = dist.get_world_size()

    def gather(x):
        x = x.contiguous()
        # concatenated output form: accepted by both RCCL and gloo; viewed as [S, ...] afterwards
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x)
        return out.view((world,) + tuple(x.shape))

    g_doc, g_score, g_cnt = gather(doc), gather(score), gather(count)
    return g_doc, g_score, g_cnt


def pack_topk(doc: torch.Tensor, score: torch.Tensor, count: torch.Tensor) -> torch.Tensor:
    """[nq * k doc ids | nq * k score bits | nq counts] as int32: one buffer, one all-gather"""
    return torch.cat([doc.contiguous().view(torch.int32).reshape(-1), score.contiguous().view(torch.int32).reshape(-1),
                      count.contiguous().view(torch.int32).reshape(-1)])


def all_gather_topk_packed(doc: torch.Tensor, score: torch.Tensor, count: torch.Tensor) -> torch.Tensor:
    """One collective for the whole batch (payload (2 k + 1) nq words per rank; three separate gathers cost three
    latencies).  Returns the gathered buffer [S, (2 k + 1) nq] int32 for merge_gathered_device_packed / unpack_gathered."""
    world = dist.get_world_size()
    x = pack_topk(doc, score, count)
    out = torch.empty((world * x.shape[0],), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x)
    return out.view(world, x.shape[0])


def unpack_gathered(packed: torch.Tensor, nq: int, k: int):
    """views [S, nq, k] doc (int32), score (f32), [S, nq] count of a packed gather"""
    S = packed.shape[0]
    nk = nq * k
    return (packed[:, :nk].reshape(S, nq, k), packed[:, nk:2 * nk].contiguous().view(torch.float32).reshape(S, nq, k),
            packed[:, 2 * nk:].reshape(S, nq))


def merge_gathered_device_packed(packed: torch.Tensor, nq: int, k: int, stream_ptr, device_index):
    """ss_topk_merge_dev_packed on the packed gather; returns (global ids int64 [nq,k], scores, counts)."""
    S = packed.shape[0]
    m_doc = torch.empty((nq, k), dtype=torch.int64, device=packed.device)
    m_score = torch.empty((nq, k), dtype=torch.float32, device=packed.device)
    m_cnt = torch.empty((nq,), dtype=torch.int32, device=packed.device)
    N.check(N.lib().ss_topk_merge_dev_packed(device_index, nq, S, k, packed.data_ptr(), m_doc.data_ptr(), m_score.data_ptr(),
                                             m_cnt.data_ptr(), stream_ptr), "ss_topk_merge_dev_packed")
    return m_doc, m_score, m_cnt


def merge_gathered_host(g_doc, g_score, g_cnt, offset, length, mode=SearchMode.Lexical):
    """Host merge of gathered single-mode lists -> per query (global ids, scores).  Shard of list s is rank s."""
    S, nq, k = g_doc.shape
    gd = g_doc.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    gs = g_score.cpu().numpy()
    gc = g_cnt.cpu().numpy().astype(np.int64)
    out = []
    for q in range(nq):
        ids, sc = [], []
        for s in range(S):
            n = int(gc[s, q])
            ids += [int(x) * S + s for x in gd[s, q, :n]]  # search.rs:1671
            sc += [float(x) for x in gs[s, q, :n]]
        lists = (ids, sc)
        if mode == SearchMode.Vector:
            out.append(merge_results(SearchMode.Vector, None, lists, offset, length)[:2])
        else:
            out.append(merge_results(SearchMode.Lexical, lists, None, offset, length)[:2])
    return out


def merge_gathered_hybrid_host(lex, vec, offset, length):
    """lex / vec: (g_doc, g_score, g_cnt) triples gathered separately; RRF over the cross-shard concatenations."""
    S, nq, _ = lex[0].shape
    res = []
    for q in range(nq):
        parts = []
        for g_doc, g_score, g_cnt in (lex, vec):
            gd = g_doc.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            gs = g_score.cpu().numpy()
            gc = g_cnt.cpu().numpy().astype(np.int64)
            ids, sc = [], []
            for s in range(S):
                n = int(gc[s, q])
                ids += [int(x) * S + s for x in gd[s, q, :n]]
                sc += [float(x) for x in gs[s, q, :n]]
            parts.append((ids, sc))
        res.append(merge_results(SearchMode.Hybrid, parts[0], parts[1], offset, length))
    return res


def merge_gathered_device(g_doc, g_score, g_cnt, stream_ptr, device_index):
    """ss_topk_merge_dev on the gathered device tensors; returns (global ids int64 [nq,k], scores, counts)."""
    S, nq, k = g_doc.shape
    m_doc = torch.empty((nq, k), dtype=torch.int64, device=g_doc.device)
    m_score = torch.empty((nq, k), dtype=torch.float32, device=g_doc.device)
    m_cnt = torch.empty((nq,), dtype=torch.int32, device=g_doc.device)
    N.check(N.lib().ss_topk_merge_dev(device_index, nq, S, k, g_doc.data_ptr(), g_score.data_ptr(), g_cnt.data_ptr(),
                                      m_doc.data_ptr(), m_score.data_ptr(), m_cnt.data_ptr(), stream_ptr), "ss_topk_merge_dev")
    return m_doc, m_score, m_cnt


class PeerError(RuntimeError):
    """a collective search failed on some rank: every rank raises (SS_EPEER on the ranks that were healthy)"""


def exchange_one_gather(lists, totals, local_error=0):
    """The exchange of ss_*_search_sharded (csrc/comm.hip) in torch.distributed terms, for the CPU tests and as its
    specification: ONE all-gather of [per list: nq*k doc ids | nq*k score bits | nq counts] + [nq totals (u64) | status].
    lists: [(doc [nq,k] int32, score [nq,k] f32, count [nq] int32), ...] of THIS rank; totals [nq] (uint64-valued) per
    query.  A rank whose local search failed passes local_error != 0: it still takes part, with empty lists, and every
    rank raises after the gather.  Returns ([(g_doc [S,nq,k], g_score, g_cnt), ...], summed totals [nq])."""
    world = dist.get_world_size()
    parts, shapes = [], []
    for doc, score, cnt in lists:
        shapes.append(tuple(doc.shape))
        if local_error:
            doc, score, cnt = torch.zeros_like(doc), torch.zeros_like(score), torch.zeros_like(cnt)
        parts.append(pack_topk(doc, score, cnt))
    tot = torch.zeros(len(totals) + 1, dtype=torch.int64)
    if not local_error:
        tot[:-1] = torch.as_tensor(np.asarray(totals, np.int64))
    tot[-1] = 1 if local_error else 0
    parts.append(tot.view(torch.int32).reshape(-1))
    x = torch.cat(parts)
    out = torch.empty((world * x.shape[0],), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x)
    out = out.view(world, x.shape[0])
    at, gathered = 0, []
    for nq, k in shapes:
        n = 2 * nq * k + nq
        gathered.append(unpack_gathered(out[:, at:at + n], nq, k))
        at += n
    gt = out[:, at:].contiguous().view(torch.int64)  # [S, nq + 1]
    if int(gt[:, -1].sum()) != 0:
        raise PeerError("local search failed" if local_error else "a peer's search failed")
    return gathered, gt[:, :-1].sum(dim=0).numpy().astype(np.uint64)


def search_hybrid_exchanged(lex, vec, lex_total, vec_total, offset, length, local_error=0):
    """SearchMode.Hybrid of Index.search over one shard per rank: both per-shard lists (top offset + length each) in one gather,
    RRF over the cross-shard concatenations, sort / offset / length (search.rs:1962-2035, 2098-2119); totals = sum over the
    shards of max(lexical, vector) (1919-1921).  Returns (per-query merge_results tuples, totals)."""
    both = np.maximum(np.asarray(lex_total, np.uint64), np.asarray(vec_total, np.uint64))
    (g_lex, g_vec), tot = exchange_one_gather([lex, vec], both, local_error)
    return merge_gathered_hybrid_host(g_lex, g_vec, offset, length), tot


class ShardComm:
    """One rank of the RCCL communicator behind the C ABI (ss_comm_create): the per-shard top-k exchange without
    torch.distributed on the data path.  The 128-byte unique id is handed from rank 0 to the others through
    torch.distributed's store-backed broadcast_object_list (control plane only; any channel would do)."""

    def __init__(self, rank, world, device_index):
        import ctypes as C
        self.rank, self.world, self.device = int(rank), int(world), int(device_index)
        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            N.check(N.lib().ss_comm_unique_id(buf), "ss_comm_unique_id")
            ident[0] = bytes(buf)
        if self.world > 1:
            dist.broadcast_object_list(ident, src=0)
        self._id = (C.c_uint8 * 128).from_buffer_copy(ident[0])
        h = C.c_void_p()
        N.check(N.lib().ss_comm_create(self.device, self.rank, self.world, self._id, C.byref(h)), "ss_comm_create")
        self._h = h

    def allgather_merge(self, doc, score, count, k, stream_ptr):
        """doc / score [nq, k], count [nq] of this rank's shard (device tensors) -> merged (global ids int64 [nq, k], scores,
        counts) on every rank: ss_topk_allgather_merge (pack, ONE ncclAllGather, ss_topk_merge_dev_packed)"""
        nq = doc.shape[0]
        m_doc = torch.empty((nq, k), dtype=torch.int64, device=doc.device)
        m_score = torch.empty((nq, k), dtype=torch.float32, device=doc.device)
        m_cnt = torch.empty((nq,), dtype=torch.int32, device=doc.device)
        N.check(N.lib().ss_topk_allgather_merge(self._h, nq, int(k), doc.data_ptr(), score.data_ptr(), count.data_ptr(),
                                                m_doc.data_ptr(), m_score.data_ptr(), m_cnt.data_ptr(), stream_ptr),
                "ss_topk_allgather_merge")
        return m_doc, m_score, m_cnt

    def search_lexical_sharded(self, shard, queries, k, result_type=N.RT_TOPKCOUNT):
        """this rank's part of Index.search over shards on different GPUs (ss_bm25_search_sharded): search `shard`, exchange,
        merge -> (global ids uint64 [nq, k], scores, counts, totals summed over the shards), the same on every rank"""
        import numpy as np
        nq = len(queries)
        kk = max(int(k), 1)
        doc = np.full((nq, kk), np.iinfo(np.uint64).max, np.uint64)
        score = np.zeros((nq, kk), np.float32)
        cnt = np.zeros(nq, np.uint32)
        tot = np.zeros(nq, np.uint64)
        N.check(N.lib().ss_bm25_search_sharded(shard._h, self._h, nq, queries.ctypes.data, int(k), int(result_type), doc.ctypes.data,
                                               score.ctypes.data, cnt.ctypes.data, tot.ctypes.data), "ss_bm25_search_sharded")
        return doc, score, cnt, tot

    def search_vector_sharded(self, shard, queries, k, threshold=N.FLT_MIN_NEG):
        """the vector shard task of the same search (ss_vec_search_sharded): AnnMode.All f32 scan of `shard`, exchange, merge"""
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        doc = np.full((nq, int(k)), np.iinfo(np.uint64).max, np.uint64)
        score = np.zeros((nq, int(k)), np.float32)
        cnt = np.zeros(nq, np.uint32)
        tot = np.zeros(nq, np.uint64)
        N.check(N.lib().ss_vec_search_sharded(shard._h, self._h, nq, q.ctypes.data, int(k), float(threshold), doc.ctypes.data,
                                              score.ctypes.data, cnt.ctypes.data, tot.ctypes.data), "ss_vec_search_sharded")
        return doc, score, cnt, tot

    def search_hybrid_sharded(self, shard, queries, query_vectors, offset, length, result_type=N.RT_TOPKCOUNT, threshold=N.FLT_MIN_NEG):
        """SearchMode.Hybrid over shards on different GPUs (ss_hybrid_search_sharded): both shard tasks at k = offset + length, ONE
        all-gather, RRF over the cross-shard concatenations -> (global ids uint64 [nq, length], fused scores, sources, counts,
        totals = sum over the shards of max(lexical, vector))"""
        q = np.ascontiguousarray(query_vectors, np.float32)
        nq = q.shape[0]
        assert len(queries) == nq
        doc = np.full((nq, int(length)), np.iinfo(np.uint64).max, np.uint64)
        score = np.zeros((nq, int(length)), np.float32)
        src = np.zeros((nq, int(length)), np.uint8)
        cnt = np.zeros(nq, np.uint32)
        tot = np.zeros(nq, np.uint64)
        N.check(N.lib().ss_hybrid_search_sharded(shard._h, self._h, nq, queries.ctypes.data, int(result_type), q.ctypes.data, float(threshold),
                                                 int(offset + length), int(offset), int(length), doc.ctypes.data, score.ctypes.data,
                                                 src.ctypes.data, cnt.ctypes.data, tot.ctypes.data), "ss_hybrid_search_sharded")
        return doc, score, src, cnt, tot

    def info(self):
        """(rank, ranks the communicator really spans -- ncclCommCount --, device) as the library sees them (ss_comm_info)"""
        import ctypes as C
        r, n, d = C.c_int(), C.c_int(), C.c_int()
        N.check(N.lib().ss_comm_info(self._h, C.byref(r), C.byref(n), C.byref(d)), "ss_comm_info")
        return int(r.value), int(n.value), int(d.value)

    def profile(self, on=True):
        N.check(N.lib().ss_comm_profile(self._h, 1 if on else 0), "ss_comm_profile")

    def profile_read(self, reset=True):
        """(collectives, mean microseconds per all-gather) of the sharded searches since the last reset"""
        import ctypes as C
        n, us = C.c_uint64(), C.c_double()
        N.check(N.lib().ss_comm_profile_read(self._h, C.byref(n), C.byref(us), 1 if reset else 0), "ss_comm_profile_read")
        return int(n.value), (us.value / n.value if n.value else 0.0)

    def close(self):
        if self._h:
            N.lib().ss_comm_destroy(self._h)
            self._h = None
