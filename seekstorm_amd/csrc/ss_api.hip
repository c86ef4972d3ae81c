// C ABI of the MI355X-native SeekStorm query hot path (see include/seekstorm_hip.h for the contract and the
// reference seams each entry point replaces).  Host-side plumbing only; the kernels live in vec_scan.hip / bm25.hip.
#include <linux/futex.h>
#include <sys/syscall.h>
#include <sched.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iterator>
#include <memory>
#include <new>
#include <unordered_map>

#include "ss_common.h"
#include "sparse_levels.h"
#include "ss_threads.h"
#include "facet_point.h"
#include "bm25_build.h"

#define SS_TRY(x)          \
  do {                     \
    int _rc = (x);         \
    if (_rc) return _rc;   \
  } while (0)

namespace {
// runs `search` with the shard's tombstone bitmap replaced by the filter's exclusion bitmap (built on `st` first)
template <typename F>
int with_facet_filter(ss_shard* s, uint32_t n_filters, const ss_facet_filter* filters, hipStream_t st, F search) {
  if (n_filters == 0) return search();
  if (!s->d_facets || s->facet_docs < s->bm_n_docs) return SS_ESTATE;  // a record for every doc of the lexical image
  int rc = ssi_facet_build(s, n_filters, filters, st);
  if (rc) return rc;
  uint32_t* del = s->d_deleted;
  const uint64_t words = s->deleted_words, n = s->n_deleted;
  s->d_deleted = s->d_filter_bits;
  s->deleted_words = (s->facet_docs + 31) / 32;
  s->n_deleted = 1;  // "some doc may be excluded": the kernels take their filtered instantiations
  rc = search();
  s->d_deleted = del; s->deleted_words = words; s->n_deleted = n;
  return rc;
}
}  // namespace

// SS_CO_TRACE: where a coalesced lexical batch spends its time (sums in us; printed when the shard is destroyed)
struct CoTrace { std::atomic<uint64_t> n{0}, stage{0}, enqueue{0}, wait{0}, scatter{0}, linger{0}, wake{0}, members{0}; };
static CoTrace g_co_trace;
// ... and of the VECTOR coalescer, batch by batch (SS_CO_TRACE: what a pass waited for and whom it left behind)
struct CoVecBatch { uint32_t members, left_queued, linger_us, batch_us; uint64_t t_formed_us; };
static std::vector<CoVecBatch> g_co_vec_trace;
static std::vector<uint32_t> g_co_vec_wake_us, g_co_vec_run_us;  // per vector batch: the leader's wake loop, and the batch itself (stage .. answers scattered)
static std::mutex g_co_vec_trace_mu;
static const bool g_co_trace_on = getenv("SS_CO_TRACE") != nullptr;
static inline uint64_t co_now_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" {

int ss_abi_version(void) { return SS_ABI_VERSION; }

const char* ss_strerror(int code) {
  switch (code) {
    case SS_OK: return "ok";
    case SS_EINVAL: return "invalid argument";
    case SS_ENOMEM: return "out of memory";
    case SS_EDEVICE: return "HIP device/runtime error";
    case SS_ENOTSUP: return "not answered by the device path: the caller's own (CPU) path applies (INTEGRATION.md section 4)";
    case SS_ESTATE: return "image not uploaded";
    case SS_EPEER: return "a collective search failed on another rank";
    default: return "unknown error";
  }
}

int ss_device_count(int* out) {
  if (!out) return SS_EINVAL;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { *out = 0; return SS_EDEVICE; }
  *out = n;
  return SS_OK;
}

int ss_shard_create(int device, ss_shard** out) {
  if (!out) return SS_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SS_EDEVICE;  // fail loudly: no CPU fallback exists
  if (device < 0 || device >= n) return SS_EINVAL;
  SS_HIP(hipSetDevice(device));
  ss_shard* s = new (std::nothrow) ss_shard();
  if (!s) return SS_ENOMEM;
  s->device = device;
  s->co_lex.max_batch = 1024;
  s->co_vec.max_batch = SS_VEC_BATCH;
  // Lexical batches in flight: ONE, and a second lane while ~96 or more callers seem to be around (adaptive; SS_COALESCE_LANES=1 / 2 force
  // one / two, =autoN moves the threshold).  History: two unconditional lanes halve the batches, and the device's fixed cost per batch is
  // what bounds small ones (round 4: T = 8 54 K -> 25 K q/s; profiles/r4k_lanes.log); opening the lane from ~32 callers on gained 7 % at
  // T = 64 and cost hybrid callers their tail (round 5, profiles/r5_lanes.log) -- while the followers' 30 us spin still ate the box's CPU
  // quota.  Round 6, spin gone, hybrid tail fixed elsewhere: one lane against the adaptive form at a threshold of 96 -- T = 64 equal
  // (300 - 330 K q/s), T = 128 336 - 356 K -> 535 K, T = 192 419 - 448 K -> 550 - 604 K, T = 256 404 - 425 K -> 508 - 511 K (p99 0.56 -
  // 0.64 ms), T = 512 534 - 600 K -> 615 K; below the threshold nothing changes; a threshold of 48 halves the batches of 64 callers
  // for nothing (profiles/r6_lanes_from.log, r6_lanes_small.log).
  // Round 6, later: all of the above was measured with both lanes feeding the shard's ONE stream -- their kernels ran one behind the other,
  // and a second lane could only overlap host work with device work.  Each lane's one-launch kernels now run on a stream of the lane's own
  // (ss_common.h lstream; bm25_small_try): two batches of ~22 callers overlap ON THE DEVICE, and the threshold is 32 -- T = 64 300 - 330 K ->
  // 430 - 454 K q/s (p50 210 -> 142 us, p99 290 -> 195 us), T = 96 -> 490 - 500 K, T = 128 535 -> 570 - 720 K, T = 192 -> 570 - 620 K, T = 256 /
  // 512 unchanged (500 / 600 K: batches of 48 queries or more fill the chip alone and share lane 0's stream), T = 8 91 -> 100 - 110 K;
  // hybrid and vector callers unchanged (profiles/r6_lane_streams*.log).
  {
    const char* e = getenv("SS_COALESCE_LANES");
    const bool one = e && atoi(e) == 1, two = e && atoi(e) == 2, adaptive = !one && !two;
    s->co_lex.n_lanes = one ? 1u : 2u;
    s->co_lex.lanes_forced = two;
    if (adaptive && e && strncmp(e, "auto", 4) == 0 && e[4]) s->co_lex.lanes_from = (uint32_t)std::max(2, atoi(e + 4));
  }
  // the shard's stream carries the latency-bound work (lexical searches: tens of microseconds a launch) at the device's HIGHEST priority;
  // the coalesced vector scans (milliseconds a pass) run on vstream at the lowest: a lexical kernel that arrives during a scan is
  // dispatched as the scan's workgroups retire instead of waiting for the pass to end
  int pr_least = 0, pr_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
  if (hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, pr_greatest) != hipSuccess) { delete s; return SS_EDEVICE; }
  if (hipStreamCreateWithPriority(&s->vstream, hipStreamNonBlocking, pr_least) != hipSuccess) { (void)hipStreamDestroy(s->stream); delete s; return SS_EDEVICE; }
  *out = s;
  return SS_OK;
}

// The shard mutex, as everybody but the coalescer's lane path takes it: with the lane streams drained (ss_common.h lstream) -- whoever
// holds it may write the lexical image or enqueue writers on s->stream, and no lane kernel is in flight to meet them.
struct ShardLock {
  std::unique_lock<std::mutex> g;
  explicit ShardLock(ss_shard* s) : g(s->mu) {
    s->main_dirty = true;  // (whatever this section enqueues on s->stream: the next lane launch waits for it)
    if (s->lanes_inflight) {
      for (hipStream_t st : s->lstream)
        if (st) (void)hipStreamSynchronize(st);
      s->lanes_inflight = false;
    }
  }
};

// RAII: a search on a stream other than the shard's own runs on that stream's set of scan buffers (swapped into the shard's
// fields while the launches are enqueued -- under s->mu -- and back afterwards; buffers are allocated lazily by the scan code)
struct VecWsBind {
  ss_shard* s;
  ss_vec_ws* w = nullptr;
  VecWsBind(ss_shard* s_, hipStream_t st) : s(s_) {
    if (st == s->stream) return;
    w = &s->vec_ws[st];
    swap();
  }
  ~VecWsBind() { if (w) swap(); }
  void swap() {
    std::swap(s->d_Qf, w->d_Qf);
    std::swap(s->d_vstate, w->d_vstate);
    std::swap(s->d_cand, w->d_cand);
    std::swap(s->d_qaux, w->d_qaux);
  }
};

// The ANN selection state (medoid scores, per-query cluster bitmaps, the tile list, the live counts) is ONE set of buffers per
// shard: a search with an ss_ann_mode queued on another stream than the previous one first waits for that one to finish.
// Caller holds s->mu.  (Searches without a mode touch none of it and stay concurrent across streams.)
struct AnnStateGuard {
  ss_shard* s;
  hipStream_t st;
  bool on;
  AnnStateGuard(ss_shard* s_, hipStream_t st_, const ss_ann_mode* mode) : s(s_), st(st_), on(mode != nullptr) {
    if (on && s->ann_ev_set && s->ann_ev_stream != st) (void)hipStreamWaitEvent(st, s->ann_ev, 0);
  }
  ~AnnStateGuard() {
    if (!on) return;
    if (!s->ann_ev && hipEventCreateWithFlags(&s->ann_ev, hipEventDisableTiming) != hipSuccess) { s->ann_ev = nullptr; s->ann_ev_set = false; return; }
    s->ann_ev_set = hipEventRecord(s->ann_ev, st) == hipSuccess;
    s->ann_ev_stream = st;
  }
};

static void free_vec(ss_shard* s) {
  if (s->vstream) (void)hipStreamSynchronize(s->vstream);  // a coalesced scan still running reads the arrays released below
  for (auto& kv : s->vec_ws) {
    void* wp[] = {kv.second.d_Qf, kv.second.d_vstate, kv.second.d_cand, kv.second.d_qaux};
    for (void* p : wp) if (p) (void)hipFree(p);
  }
  s->vec_ws.clear();
  void* ptrs[] = {s->d_X, s->d_X8, s->d_row_scale, s->d_row_doc, s->d_Qf, s->d_vstate, s->d_cand, s->d_row_field, s->d_row_norm, s->d_row_sq, s->d_qaux};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  s->d_X = nullptr; s->d_X8 = nullptr; s->d_row_scale = nullptr; s->d_row_doc = nullptr; s->d_Qf = nullptr;
  s->d_row_norm = nullptr; s->d_row_sq = nullptr; s->d_qaux = nullptr;
  s->d_vstate = nullptr; s->d_cand = nullptr; s->d_row_field = nullptr;
  s->n_rows = s->n_rows_pad = 0; s->dim = s->dim_pad = s->dim_pad8 = 0; s->vec_multi_record = false; s->vec_rows_cap = 0;
  ssi_vec_free_clusters(s);
  if (s->ann_ev) (void)hipEventDestroy(s->ann_ev);
  s->ann_ev = nullptr; s->ann_ev_set = false; s->ann_ev_stream = nullptr;
}
static void free_raw_levels(ss_shard* s) {
  for (ss_raw_level& L : s->raw)
    for (void* p : {(void*)L.d_off, (void*)L.d_doc, (void*)L.d_tf, (void*)L.d_npos, (void*)L.d_prel, (void*)L.d_tpos, (void*)L.d_pos}) if (p) (void)hipFree(p);
  s->raw.clear();
  s->h_doclen.clear();
  s->raw_f.clear();
  s->raw_f_fields = 0;
}
// keep_tier: the dense image goes (a commit swaps in its successor), the sparse tier stays
static void free_bm25(ss_shard* s, bool keep_tier = false) {
  free_raw_levels(s);
  void* ptrs[] = {s->d_post, s->d_term_base, s->d_sub_off, s->d_comp, s->d_probe, s->d_probe_z, s->d_probe_row, s->d_umax, s->d_submax, s->d_boost, s->d_pos, s->d_pos32, s->d_pos_off, s->d_pos_base,
                  s->d_doclen};
  for (void* p : ptrs) if (p) { s->blocks.drop(p); (void)hipFree(p); }
  s->blocks.clear_idle();
  s->d_doclen = nullptr;
  ssi_bm25_drop_kth(s);
  if (!keep_tier) {
    for (void* p : {(void*)s->d_sp_base, (void*)s->d_sp_post, (void*)s->d_sp_pos, (void*)s->d_sp_pos_end}) if (p) (void)hipFree(p);
    s->d_sp_base = nullptr; s->d_sp_post = nullptr; s->sp_n = 0; s->h_sp_base.clear();
    s->d_sp_pos = nullptr; s->d_sp_pos_end = nullptr; s->sp_pos_n = 0; s->sp_pos_elem = 0;
    ssi_bm25_sparse_levels_drop(s);
  }
  s->d_post = nullptr; s->d_term_base = nullptr; s->d_sub_off = nullptr; s->d_comp = nullptr;
  s->probe_pool_begin = 0; s->probe_pool_rows = 0; s->pool_list.clear(); s->pool_tick.clear();
  s->d_probe = nullptr; s->d_probe_z = nullptr; s->d_probe_row = nullptr; s->h_probe_row.clear(); s->bm_probe_rows = 0; s->d_umax = nullptr; s->d_submax = nullptr; s->d_pos = nullptr; s->d_pos32 = nullptr; s->d_pos_off = nullptr; s->d_pos_base = nullptr;
  s->bm_n_docs = 0; s->bm_n_terms = 0; s->bm_n_sub = 0; s->bm_n_post = 0; s->bm_n_fields = 1; s->bm_merged = false; s->d_boost = nullptr;
  s->h_df.clear(); s->h_df_real.clear(); s->h_boost.clear(); s->bm_n_post_pad = 0; s->bm_partmax = false;
}

int ss_shard_destroy(ss_shard* s) {
  if (!s) return SS_EINVAL;
  if (g_co_trace_on && g_co_trace.n.load()) {
    const double n = (double)g_co_trace.n.load() * 1000.0;  // ns -> us per batch
    fprintf(stderr, "[co] %llu lexical batches: stage %.1f us, search call %.1f us (enqueue %.1f + device wait %.1f), scatter %.1f us, waking %.1f members %.1f us per batch\n",
            (unsigned long long)g_co_trace.n.load(), g_co_trace.stage.load() / n, g_co_trace.wait.load() / n, g_co_trace.enqueue.load() / n,
            g_co_trace.linger.load() / n, g_co_trace.scatter.load() / n, g_co_trace.members.load() / (double)g_co_trace.n.load(), g_co_trace.wake.load() / n);
  }
  if (g_co_trace_on && !g_co_vec_trace.empty()) {
    std::lock_guard<std::mutex> gt(g_co_vec_trace_mu);
    std::vector<uint32_t> m, l, q;
    std::vector<uint64_t> gap;
    for (size_t i = 0; i < g_co_vec_trace.size(); i++) {
      m.push_back(g_co_vec_trace[i].members); l.push_back(g_co_vec_trace[i].linger_us); q.push_back(g_co_vec_trace[i].left_queued);
      if (i) gap.push_back(g_co_vec_trace[i].t_formed_us - g_co_vec_trace[i - 1].t_formed_us);
    }
    auto pc = [](std::vector<uint32_t> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    auto pc64 = [](std::vector<uint64_t> v, double p) { if (v.empty()) return (uint64_t)0; std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    fprintf(stderr, "[co-vec] %zu batches: members p1 %u p10 %u p50 %u; linger us p50 %u p90 %u p99 %u; left queued p50 %u p90 %u p99 %u; formed-to-formed us p50 %llu p90 %llu p99 %llu\n",
            m.size(), pc(m, 0.01), pc(m, 0.10), pc(m, 0.50), pc(l, 0.50), pc(l, 0.90), pc(l, 0.99), pc(q, 0.50), pc(q, 0.90), pc(q, 0.99),
            (unsigned long long)pc64(gap, 0.50), (unsigned long long)pc64(gap, 0.90), (unsigned long long)pc64(gap, 0.99));
    if (!g_co_vec_wake_us.empty())
      fprintf(stderr, "[co-vec] wake loop us p50 %u p90 %u p99 %u max %u; batch (formed .. scattered) us p50 %u p99 %u max %u\n", pc(g_co_vec_wake_us, 0.5), pc(g_co_vec_wake_us, 0.9),
              pc(g_co_vec_wake_us, 0.99), pc(g_co_vec_wake_us, 1.0), pc(g_co_vec_run_us, 0.5), pc(g_co_vec_run_us, 0.99), pc(g_co_vec_run_us, 1.0));
    // the batches that make a tail: fewer members than usual, a long pass, or a long gap to the batch before -- with their neighbours
    if (g_co_vec_run_us.size() == g_co_vec_trace.size() && g_co_vec_wake_us.size() == g_co_vec_trace.size() && gap.size() > 8) {
      const uint32_t m50 = pc(m, 0.5), r50 = pc(g_co_vec_run_us, 0.5);
      const uint64_t g50 = pc64(gap, 0.5);
      int shown = 0;
      for (size_t i = 1; i < g_co_vec_trace.size() && shown < 24; i++) {
        const uint64_t g = g_co_vec_trace[i].t_formed_us - g_co_vec_trace[i - 1].t_formed_us;
        if (g_co_vec_trace[i].members * 4u >= m50 * 3u && g_co_vec_run_us[i] * 4u <= r50 * 5u && g * 10u <= g50 * 13u) continue;
        shown++;
        for (size_t j = i - 1; j <= std::min(i + 1, g_co_vec_trace.size() - 1); j++)
          fprintf(stderr, "[co-vec]   %s batch %zu: members %u, left queued %u, linger %u us, formed-to-scattered %u us, wake loop %u us, since the batch before %llu us\n",
                  j == i ? "*" : " ", j, g_co_vec_trace[j].members, g_co_vec_trace[j].left_queued, g_co_vec_trace[j].linger_us, g_co_vec_run_us[j], g_co_vec_wake_us[j],
                  (unsigned long long)(j ? g_co_vec_trace[j].t_formed_us - g_co_vec_trace[j - 1].t_formed_us : 0));
      }
    }
    g_co_vec_wake_us.clear(); g_co_vec_run_us.clear();
    g_co_vec_trace.clear();
  }
  (void)hipSetDevice(s->device);
  (void)hipStreamSynchronize(s->stream);
  if (s->vstream) (void)hipStreamSynchronize(s->vstream);
  for (hipStream_t st : s->lstream)
    if (st) (void)hipStreamSynchronize(st);
  free_vec(s);
  free_bm25(s);
  for (void* p_ : {(void*)s->d_vq, (void*)s->d_vdoc, (void*)s->d_vscore, (void*)s->d_vcount, (void*)s->d_vtotal}) if (p_) (void)hipFree(p_);
  void* ptrs[] = {s->d_qstage, s->d_out_doc, s->d_out_score, s->d_out_count, s->d_out_total, s->d_bq, s->d_deleted, s->d_facets, s->d_filter_bits, s->d_facet_ws, s->d_pool_stage, s->d_tier_ws, s->d_tier_hold, s->d_excl_bits, s->d_sort_ws, s->d_route_ws, s->d_peel_bits, s->d_gate_ws};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  (void)hipDeviceSynchronize();  // searches queued on the callers' own streams may still use their workspaces
  for (auto& kv : s->bm_ws) {
    if (kv.second.d_vq) (void)hipFree(kv.second.d_vq);
    if (kv.second.d_part) (void)hipFree(kv.second.d_part);
  }
  for (int kx = 0; kx < 2; kx++)
    for (auto& pr : s->prof.pending[kx]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (ss_coalescer* co : {&s->co_lex, &s->co_vec})
    for (auto& ln : co->lane) {
      if (ln.h_pin) (void)hipHostFree(ln.h_pin);
      if (ln.ev) (void)hipEventDestroy(ln.ev);
    }
  if (s->h_bq) (void)hipHostFree(s->h_bq);
  if (s->bq_ev) (void)hipEventDestroy(s->bq_ev);
  if (s->d_small_ws) (void)hipFree(s->d_small_ws);
  for (int i = 0; i < 2; i++) {
    if (s->lstream[i]) { (void)hipStreamSynchronize(s->lstream[i]); (void)hipStreamDestroy(s->lstream[i]); }
    if (s->d_small_ws_l[i]) (void)hipFree(s->d_small_ws_l[i]);
  }
  if (s->ev_main) (void)hipEventDestroy(s->ev_main);
  if (s->h_small) (void)hipHostFree(s->h_small);
  if (s->h_ans) (void)hipHostFree(s->h_ans);
  if (s->d_ans_done) (void)hipFree(s->d_ans_done);
  (void)hipStreamDestroy(s->stream);
  if (s->vev) (void)hipEventDestroy(s->vev);
  if (s->vstream) (void)hipStreamDestroy(s->vstream);
  delete s;
  return SS_OK;
}

int ss_shard_sync(ss_shard* s) {
  if (!s) return SS_EINVAL;
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  if (s->vstream) SS_HIP(hipStreamSynchronize(s->vstream));
  for (hipStream_t st : s->lstream)
    if (st) SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

// ------------------------------------------------------------------ staging helpers
static int ensure_out(ss_shard* s, size_t nq, size_t k) {
  size_t need = nq * std::max<size_t>(k, 1);
  if (need <= s->out_cap && nq <= s->q_cap) return SS_OK;
  void* ptrs[] = {s->d_out_doc, s->d_out_score, s->d_out_count, s->d_out_total};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  s->d_out_doc = nullptr; s->d_out_score = nullptr; s->d_out_count = nullptr; s->d_out_total = nullptr;
  s->out_cap = 0; s->q_cap = 0;
  SS_HIP(hipMalloc(&s->d_out_doc, need * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_out_score, need * sizeof(float)));
  SS_HIP(hipMalloc(&s->d_out_count, nq * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_out_total, nq * sizeof(uint64_t)));
  s->out_cap = need; s->q_cap = nq;
  return SS_OK;
}

// grow-only device staging of host-pointer queries (a hipMalloc / hipFree pair per call would synchronise the device)
static int ensure_qstage(ss_shard* s, size_t bytes) {
  if (bytes <= s->qstage_cap) return SS_OK;
  if (s->d_qstage) (void)hipFree(s->d_qstage);
  s->d_qstage = nullptr;
  s->qstage_cap = 0;
  SS_HIP(hipMalloc(&s->d_qstage, bytes));
  s->qstage_cap = bytes;
  return SS_OK;
}

// ------------------------------------------------------------------ BM25
int ss_bm25_upload(ss_shard* s, uint64_t n_docs, const uint8_t* doclen, uint32_t n_terms, const uint64_t* offs,
                   const uint32_t* docs, const uint16_t* tfs) {
  return ss_guard([&] { return ssi_bm25_upload(s, n_docs, doclen, n_terms, offs, docs, tfs, 0); }, SS_ENOMEM, SS_EDEVICE);
}

int ss_bm25_upload_positions(ss_shard* s, uint64_t n_docs, const uint8_t* doclen, uint32_t n_terms, const uint64_t* offs,
                             const uint32_t* docs, const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions) {
  if (n_positions && !positions) return SS_EINVAL;
  // the positions array must hold exactly sum(tf) entries -- checked BEFORE anything is built or read: the walk over the
  // postings indexes it by the running sum of the tfs
  if (!offs || n_terms == 0) return SS_EINVAL;
  if (offs[n_terms] && !tfs) return SS_EINVAL;
  uint64_t need = 0;
  for (uint64_t j = offs[0]; j < offs[n_terms]; j++) need += tfs[j];
  if (need != n_positions || (need && !positions)) return SS_EINVAL;
  int rc = ssi_bm25_upload(s, n_docs, doclen, n_terms, offs, docs, tfs, 0);
  if (rc) return rc;
  return ssi_bm25_attach_positions(s, offs, docs, tfs, positions, n_positions);
}

}  // extern "C"

// positions of the image just built (CSR order); a failure leaves no image behind
int ssi_bm25_attach_positions(ss_shard* s, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs, const uint16_t* positions,
                              uint64_t n_positions, const uint16_t* npos) {
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  const int rc = ssi_bm25_upload_positions(s, offs, docs, tfs, positions, n_positions, npos);
  if (rc) free_bm25(s);
  return rc;
}

// the arrays of a whole-image upload -> device, validated on the way (terms in parallel); the image is built by the device kernels
static int bm25_upload_device_build(ss_shard* s, const uint8_t* doclen, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                                    uint64_t positions_sum) {
  const uint32_t nt = s->bm_n_terms;
  const uint64_t nd = s->bm_n_docs, np = offs[nt] - offs[0];
  for (uint32_t t = 0; t < nt; t++)
    if (offs[t + 1] < offs[t]) return SS_EINVAL;
  std::atomic<int> bad{0};
  ss_parallel_for(nt, 64, [&](size_t a, size_t b, unsigned) {
    for (size_t t = a; t < b; t++)
      for (uint64_t j = offs[t]; j < offs[t + 1]; j++)
        if (docs[j] >= nd || tfs[j] == 0 || (j > offs[t] && docs[j] <= docs[j - 1])) { bad.store(1); return; }
  });
  if (bad.load()) return SS_EINVAL;
  ss_raw_level L;
  L.n_docs = 0; L.n_terms = nt; L.n_post = np;
  if (positions_sum) L.psum = positions_sum;
  else {
    std::vector<uint64_t> part(ss_loader_threads() + 1, 0);
    ss_parallel_for(nd, 1u << 20, [&](size_t a, size_t b, unsigned w) { uint64_t c = 0; for (size_t d = a; d < b; d++) c += ss_byte4_to_int(doclen[d]); part[w] += c; });
    for (uint64_t c : part) L.psum += c;
  }
  auto drop = [&]() { for (void* p : {(void*)L.d_off, (void*)L.d_doc, (void*)L.d_tf}) if (p) (void)hipFree(p); };
#define SS_HIP_D(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { drop(); return e_ == hipErrorOutOfMemory ? SS_ENOMEM : SS_EDEVICE; } } while (0)
  std::vector<uint64_t> rel((size_t)nt + 1);
  for (uint32_t t = 0; t <= nt; t++) rel[t] = offs[t] - offs[0];
  SS_HIP_D(hipMalloc(&L.d_off, rel.size() * sizeof(uint64_t)));
  SS_HIP_D(hipMalloc(&L.d_doc, std::max<uint64_t>(np, 1) * sizeof(uint32_t)));
  SS_HIP_D(hipMalloc(&L.d_tf, std::max<uint64_t>(np, 1) * sizeof(uint16_t)));
  SS_HIP_D(hipMemcpyAsync(L.d_off, rel.data(), rel.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
  if (np) {
    SS_HIP_D(hipMemcpyAsync(L.d_doc, docs + offs[0], np * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    SS_HIP_D(hipMemcpyAsync(L.d_tf, tfs + offs[0], np * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
  }
  SS_HIP_D(hipStreamSynchronize(s->stream));
#undef SS_HIP_D
  std::vector<ss_raw_level> one{L};
  const int rc = ssi_bm25_rebuild_from_raw(s, one, nt, doclen, nd, s, s->stream, true);
  drop();
  return rc;
}

int ssi_bm25_upload(ss_shard* s, uint64_t n_docs, const uint8_t* doclen, uint32_t n_terms, const uint64_t* offs,
                    const uint32_t* docs, const uint16_t* tfs, uint64_t positions_sum) {
  if (!s || !doclen || !offs || n_docs == 0 || n_terms == 0) return SS_EINVAL;
  if (offs[n_terms] && (!docs || !tfs)) return SS_EINVAL;
  if (n_docs > 0xFFFFFFFFull) return SS_ENOTSUP;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  free_bm25(s);
  s->bm_n_docs = n_docs;
  s->bm_n_terms = n_terms;
  s->bm_n_sub = (uint32_t)((n_docs + BM_SUB - 1) >> BM_SUB_LOG2);
  // One indexed field: the image is BUILT ON THE DEVICE from the decoded arrays (the kernels of the incremental rebuild, synth.hip
  // ssi_bm25_rebuild_from_raw) -- the host validates and copies 6 bytes per posting instead of packing, sorting out segments and
  // filling probe rows.  SS_BM25_HOST_BUILD=1 keeps the host builder (the two are compared by the tests).
  constexpr bool host_build = false;  // (the host builder: images with several fields, and what the device builder was checked against)
  int rc;
  if (host_build) {
    rc = ssi_bm25_build_from_host(s, doclen, offs, docs, tfs, positions_sum);
  } else {
    rc = bm25_upload_device_build(s, doclen, offs, docs, tfs, positions_sum);
  }
  if (rc) free_bm25(s);
  return rc;
}

extern "C" {

// Several indexed fields (BM25F, get_bm25f_multiterm_multifield add_result.rs:1171-1426): a posting = (term, doc, field, tf).
// The image holds one posting list per (term, field); a query term is expanded over its fields on the device.
int ss_bm25_upload_fields(ss_shard* s, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                          uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                          const uint16_t* tfs) {
  return ss_guard([&] { return ssi_bm25_upload_fields(s, n_docs, n_fields, doclen, boost, n_terms, offs, docs, fields, tfs, 0); }, SS_ENOMEM, SS_EDEVICE);
}

// ... plus the positions of every (term, doc, field) entry: phrase queries over several indexed fields
int ss_bm25_upload_fields_positions(ss_shard* s, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                                    uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                    const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions) {
  if (!offs || n_terms == 0) return SS_EINVAL;
  if (offs[n_terms] && !tfs) return SS_EINVAL;
  uint64_t need = 0;  // checked before anything is built or read, as in ss_bm25_upload_positions
  for (uint64_t j = offs[0]; j < offs[n_terms]; j++) need += tfs[j];
  if (need != n_positions || (need && !positions)) return SS_EINVAL;
  return ss_guard([&] { return ssi_bm25_upload_fields_positions(s, n_docs, n_fields, doclen, boost, n_terms, offs, docs, fields, tfs, 0, positions, n_positions); },
                  SS_ENOMEM, SS_EDEVICE);
}

}  // extern "C"

int ssi_bm25_upload_fields_positions(ss_shard* s, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                                     uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                     const uint16_t* tfs, uint64_t positions_sum, const uint16_t* positions, uint64_t n_positions, const uint16_t* npos) {
  if (n_fields < 2) return SS_EINVAL;  // one indexed field: ss_bm25_upload_positions
  static const bool trace = getenv("SS_LOAD_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = ssi_bm25_upload_fields(s, n_docs, n_fields, doclen, boost, n_terms, offs, docs, fields, tfs, positions_sum);
  if (rc) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  rc = ssi_bm25_upload_positions_fields(s, n_terms, offs, docs, fields, tfs, positions, n_positions, npos);
  if (trace) fprintf(stderr, "[load]   fields image %.0f ms, positions %.0f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  if (rc) free_bm25(s);  // a failure leaves no image behind
  return rc;
}

int ssi_bm25_upload_fields(ss_shard* s, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                           uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                           const uint16_t* tfs, uint64_t positions_sum) {
  if (!s || !doclen || !offs || n_docs == 0 || n_terms == 0 || n_fields == 0 || n_fields > 8) return SS_EINVAL;
  if (offs[n_terms] && (!docs || !fields || !tfs)) return SS_EINVAL;
  // (term, field) lists: entries of a term are sorted by (doc, field); a stable split by field keeps each list sorted by doc.
  // Two or more fields: one more list per term, the MERGED list (ss_common.h bm_merged) -- every doc of the term once.
  std::vector<float> b(n_fields, 1.0f);
  if (boost) b.assign(boost, boost + n_fields);
  float bmax = 0.f;
  bool boosts_ok = true;
  for (float x : b) { boosts_ok = boosts_ok && x > 0.f && x < 1e30f; bmax = std::max(bmax, x); }
  constexpr bool merged_off = false;
  (void)bmax;
  std::vector<uint64_t> df_real(n_terms, 0);
  for (uint32_t t = 0; t < n_terms; t++)
    if (offs[t + 1] < offs[t]) return SS_EINVAL;
  {  // (per-term loops on the loader's worker threads, here and below: a term's entries and lists are its own)
    std::atomic<int> bad{0};
    ss_parallel_for(n_terms, 64, [&](size_t ta, size_t tb, unsigned) {
      for (size_t t = ta; t < tb; t++)
        for (uint64_t j = offs[t]; j < offs[t + 1]; j++) {
          if (fields[j] >= n_fields) { bad.store(1); return; }
          if (j > offs[t] && (docs[j] < docs[j - 1] || (docs[j] == docs[j - 1] && fields[j] <= fields[j - 1]))) { bad.store(1); return; }
          if (j == offs[t] || docs[j] != docs[j - 1]) df_real[t]++;  // docs containing the term in any field: the df of idf
        }
    });
    if (bad.load()) return SS_EINVAL;
  }
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  // with merged lists first; a corpus whose merged weights do not fit the weight code is built again without them
  int rc = SS_OK;
  for (int attempt = 0; attempt < 2; attempt++) {
    const bool merged = attempt == 0 && n_fields > 1 && boosts_ok && !merged_off;
    if (attempt == 0 && !merged) continue;
    const uint32_t L = n_fields + (merged ? 1u : 0u);
    if ((uint64_t)n_terms * L > 0x7FFFFFFFull) return SS_ENOTSUP;
    const uint32_t nv = n_terms * L;
    std::vector<uint64_t> voff((size_t)nv + 1, 0);
    ss_parallel_for(n_terms, 64, [&](size_t ta, size_t tb, unsigned) {
      for (size_t t = ta; t < tb; t++) {
        for (uint64_t j = offs[t]; j < offs[t + 1]; j++) voff[t * L + fields[j] + 1]++;
        if (merged) voff[t * L + n_fields + 1] = df_real[t];
      }
    });
    for (uint32_t v = 0; v < nv; v++) voff[v + 1] += voff[v];
    std::vector<uint32_t> vdocs(voff[nv]);
    std::vector<uint16_t> vtfs(voff[nv]);
    std::vector<uint64_t> cur(voff.begin(), voff.end() - 1);
    ss_parallel_for(n_terms, 64, [&](size_t ta, size_t tb, unsigned) {
      for (size_t t = ta; t < tb; t++)
        for (uint64_t j = offs[t]; j < offs[t + 1]; j++) {
          const uint64_t w = cur[t * L + fields[j]]++;
          vdocs[w] = docs[j];
          vtfs[w] = tfs[j];
          if (merged && (j == offs[t] || docs[j] != docs[j - 1])) {
            const uint64_t m = cur[t * L + n_fields]++;
            vdocs[m] = docs[j];
            vtfs[m] = 1;  // not a tf: the builder derives the merged weight from the field lists
          }
        }
    });
    free_bm25(s);
    s->bm_n_docs = n_docs;
    s->bm_n_fields = L;
    s->bm_merged = merged;
    s->bm_n_terms = nv;
    s->bm_n_sub = (uint32_t)((n_docs + BM_SUB - 1) >> BM_SUB_LOG2);
    float scale = 1.0f;
    const auto tb0 = std::chrono::steady_clock::now();
    rc = ssi_bm25_build_from_host_merged(s, doclen, voff.data(), vdocs.data(), vtfs.data(), positions_sum, merged ? b.data() : nullptr, &scale);
    if (getenv("SS_LOAD_TRACE")) fprintf(stderr, "[load]     build_from_host_merged %.0f ms (attempt %d)\n",
                                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(), attempt);
    if (rc == SS_MERGED_RANGE) continue;  // the merged weights span more than the code: again, without merged lists
    if (rc == SS_OK) {
      std::vector<float> bb = b;
      if (merged) bb.push_back(scale);  // the merged list's "boost" gives the scale back through idf
      if (hipMalloc(&s->d_boost, bb.size() * sizeof(float)) != hipSuccess ||
          hipMemcpy(s->d_boost, bb.data(), bb.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = SS_EDEVICE;
      s->h_df_real = df_real;
      s->h_boost = bb;
      s->bm_n_post = offs[n_terms];  // the postings of the index (the merged lists are a second copy)
    }
    break;
  }
  if (rc) free_bm25(s);
  return rc;
}

extern "C" {

int ss_synth_set_partition(ss_shard* s, uint32_t shard_id, uint32_t n_shards) {
  if (!s || n_shards == 0 || shard_id >= n_shards) return SS_EINVAL;
  ShardLock g(s);
  s->synth_stride = n_shards;
  s->synth_offset = shard_id;
  return SS_OK;
}

int ss_bm25_synth(ss_shard* s, uint64_t seed, uint64_t n_docs, uint32_t n_terms, const uint32_t* thresh32,
                  const uint8_t* len_table1024) {
  if (!s || !thresh32 || !len_table1024 || n_docs == 0 || n_terms == 0) return SS_EINVAL;
  if (n_docs > 0xFFFFFFFFull) return SS_ENOTSUP;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  free_bm25(s);
  s->bm_n_docs = n_docs;
  s->bm_n_terms = n_terms;
  s->bm_n_sub = (uint32_t)((n_docs + BM_SUB - 1) >> BM_SUB_LOG2);
  uint32_t* d_th = nullptr;
  uint8_t* d_tab = nullptr;
  SS_HIP(hipMalloc(&d_th, (size_t)n_terms * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&d_tab, 1024));
  SS_HIP(hipMemcpy(d_th, thresh32, (size_t)n_terms * sizeof(uint32_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(d_tab, len_table1024, 1024, hipMemcpyHostToDevice));
  int rc = ssi_bm25_synth(s, seed, d_th, d_tab, s->stream);
  (void)hipFree(d_th);
  (void)hipFree(d_tab);
  if (rc) free_bm25(s);
  return rc;
}

// Incremental commit.  The reference commits one 65 536-doc level at a time and rebuilds its in-RAM structures for it (commit.rs:142-148
// commit -> warmup, 264-369 the level writer; index.rs:3796); a BM25 weight depends on avgdl, which every commit moves (commit.rs:318-325
// refills bm25_component_cache), so no posting of the image survives a commit unchanged.  Here the decoded postings of every level stay
// in HBM as they arrived (6 bytes per posting) and the image is rebuilt from them ON THE DEVICE (ssi_bm25_rebuild_from_raw: count,
// scan, fill -- the raw postings read once, the image written once): a commit costs the level's H2D + a few milliseconds per GB of
// image, not a host pass over the shard.  The new image is built beside the old one; searches keep running on the old image until the
// swap, which waits for the searches in flight and releases the old arrays.
static int append_level_impl(ss_shard* s, uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms, const uint64_t* offs,
                             const uint32_t* docs, const uint16_t* tfs, bool with_pos, const uint16_t* npos, const uint16_t* positions, uint64_t n_positions);
int ss_bm25_append_level(ss_shard* s, uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms, const uint64_t* offs,
                         const uint32_t* docs, const uint16_t* tfs) {
  return ss_guard([&] { return append_level_impl(s, level, n_level_docs, level_doclen, n_terms, offs, docs, tfs, false, nullptr, nullptr, 0); }, SS_ENOMEM, SS_EDEVICE);
}
// ... with the postings' POSITIONS (phrase queries on an image that grows by commits): positions = every posting's, in CSR order, tf of
// them each -- or npos[i] where that is not the tf (npos may be NULL): the component terms of an n-gram key, whose own positions stand
// behind its first component's postings.  Every level of an image brings positions, or none does.
int ss_bm25_append_level_positions(ss_shard* s, uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms,
                                   const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs, const uint16_t* npos, const uint16_t* positions,
                                   uint64_t n_positions) {
  if (n_positions && !positions) return SS_EINVAL;
  return ss_guard([&] { return append_level_impl(s, level, n_level_docs, level_doclen, n_terms, offs, docs, tfs, true, npos, positions, n_positions); }, SS_ENOMEM, SS_EDEVICE);
}
static int append_level_impl(ss_shard* s, uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms, const uint64_t* offs,
                             const uint32_t* docs, const uint16_t* tfs, bool with_pos, const uint16_t* npos, const uint16_t* positions, uint64_t n_positions) {
  if (!s || !level_doclen || !offs || n_terms == 0 || n_level_docs == 0 || n_level_docs > 65536u) return SS_EINVAL;
  if (offs[n_terms] && (!docs || !tfs)) return SS_EINVAL;
  const auto t_begin = std::chrono::steady_clock::now();
  SS_HIP(hipSetDevice(s->device));
  std::vector<ss_raw_level> levels;
  std::vector<uint8_t> doclen;
  uint32_t nt_old = 0;
  {
    ShardLock g(s);
    if (s->d_post && s->raw.empty()) return SS_ESTATE;       // an image that was not built level by level
    // a sparse tier numbers its terms behind the dense ones: a grown dense vocabulary would shift them -- new terms of an image with a
    // tier join the tier (ss_bm25_append_sparse_level), whose postings must come level by level too (their tfs are kept for re-coding)
    if (s->sp_n && (!ssi_bm25_sparse_levels_has(s) || n_terms != s->bm_n_terms)) return SS_ENOTSUP;
    // (a re-committed level only grows, commit.rs:204-206; a shrinking one would leave the tier's postings of the level pointing
    // past the image until ss_bm25_append_sparse_level replaces them)
    if (s->sp_n && level < s->raw.size() && n_level_docs < s->raw[level].n_docs) return SS_ENOTSUP;
    if (level > s->raw.size() || level + 1 < s->raw.size()) return SS_EINVAL;  // append the next level, or replace the last one (a re-commit)
    if (level >= 1 && s->raw[level - 1].n_docs != 65536u) return SS_EINVAL;    // only the last level may be partial
    if ((uint64_t)level * 65536u + n_level_docs > 0xFFFFFFFFull) return SS_ENOTSUP;
    nt_old = s->raw.empty() ? 0u : s->bm_n_terms;
    if (n_terms < nt_old) return SS_EINVAL;                    // the vocabulary only grows; new terms get the next ids
    if (level > 0 && (s->raw[0].d_tpos != nullptr) != with_pos) return SS_EINVAL;  // positions for every level of an image, or for none
    levels.assign(s->raw.begin(), s->raw.begin() + level);
    doclen.assign(s->h_doclen.begin(), s->h_doclen.begin() + (size_t)level * 65536u);
  }
  // validation: docs of a term ascending and inside the level
  const uint64_t d_lo = (uint64_t)level * 65536u, d_hi = d_lo + n_level_docs;
  for (uint32_t t = 0; t < n_terms; t++)
    if (offs[t + 1] < offs[t]) return SS_EINVAL;
  {
    std::atomic<int> bad{0};
    ss_parallel_for(n_terms, 64, [&](size_t ta, size_t tb, unsigned) {
      for (size_t t = ta; t < tb; t++)
        for (uint64_t j = offs[t]; j < offs[t + 1]; j++)
          if (docs[j] < d_lo || docs[j] >= d_hi || tfs[j] == 0 || (j > offs[t] && docs[j] <= docs[j - 1])) { bad.store(1); return; }
    });
    if (bad.load()) return SS_EINVAL;
  }
  ss_raw_level L;
  L.n_docs = n_level_docs; L.n_terms = n_terms; L.n_post = offs[n_terms] - offs[0];
  for (uint32_t d = 0; d < n_level_docs; d++) L.psum += ss_byte4_to_int(level_doclen[d]);
  // positions: per posting its count, the positions of the term's earlier postings of this level, per term its first position
  std::vector<uint16_t> h_npos;
  std::vector<uint32_t> h_prel;
  std::vector<uint64_t> h_tpos;
  if (with_pos) {
    h_npos.resize(L.n_post); h_prel.resize(L.n_post); h_tpos.assign((size_t)n_terms + 1, 0);
    ss_parallel_for(n_terms, 64, [&](size_t ta, size_t tb, unsigned) {  // a term's positions: the sum of its postings' counts
      for (size_t t = ta; t < tb; t++) {
        uint64_t c = 0;
        for (uint64_t j = offs[t]; j < offs[t + 1]; j++) c += npos ? npos[j] : tfs[j];
        h_tpos[t + 1] = c;
      }
    });
    for (uint32_t t = 0; t < n_terms; t++) h_tpos[t + 1] += h_tpos[t];
    if (h_tpos[n_terms] != n_positions) return SS_EINVAL;
    std::atomic<int> bad{SS_OK};
    ss_parallel_for(n_terms, 64, [&](size_t ta, size_t tb, unsigned) {
      for (size_t t = ta; t < tb; t++) {
        const uint64_t at = h_tpos[t];
        uint64_t rel = 0;
        for (uint64_t j = offs[t]; j < offs[t + 1]; j++) {
          const uint32_t c = npos ? npos[j] : tfs[j];
          if (rel >= (1ull << 32)) { bad.store(SS_ENOTSUP); return; }
          h_npos[j - offs[0]] = (uint16_t)c; h_prel[j - offs[0]] = (uint32_t)rel;
          for (uint32_t x = 1; x < c; x++)
            if (positions[at + rel + x] <= positions[at + rel + x - 1]) { bad.store(SS_EINVAL); return; }  // ascending inside a posting
          rel += c;
        }
      }
    });
    if (bad.load()) return bad.load();
    L.n_pos = n_positions;
  }
  hipStream_t bst = nullptr;
  std::unique_ptr<ss_shard> img(new ss_shard);
  auto fail = [&](int rc) {
    for (void* p : {(void*)L.d_off, (void*)L.d_doc, (void*)L.d_tf, (void*)L.d_npos, (void*)L.d_prel, (void*)L.d_tpos, (void*)L.d_pos}) if (p) (void)hipFree(p);
    void* ip[] = {img->d_post, img->d_term_base, img->d_sub_off, img->d_comp, img->d_probe, img->d_probe_z, img->d_probe_row, img->d_umax, img->d_submax, img->d_doclen,
                  img->d_pos, img->d_pos_off, img->d_pos_base};
    for (void* p : ip) if (p && !s->blocks.release(p)) (void)hipFree(p);
    if (bst) (void)hipStreamDestroy(bst);
    return rc;
  };
#define SS_HIP_F(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? SS_ENOMEM : SS_EDEVICE); } while (0)
  SS_HIP_F(hipStreamCreateWithFlags(&bst, hipStreamNonBlocking));
  std::vector<uint64_t> rel((size_t)n_terms + 1);
  for (uint32_t t = 0; t <= n_terms; t++) rel[t] = offs[t] - offs[0];
  SS_HIP_F(hipMalloc(&L.d_off, rel.size() * sizeof(uint64_t)));
  SS_HIP_F(hipMalloc(&L.d_doc, std::max<uint64_t>(L.n_post, 1) * sizeof(uint32_t)));
  SS_HIP_F(hipMalloc(&L.d_tf, std::max<uint64_t>(L.n_post, 1) * sizeof(uint16_t)));
  SS_HIP_F(hipMemcpyAsync(L.d_off, rel.data(), rel.size() * sizeof(uint64_t), hipMemcpyHostToDevice, bst));
  if (L.n_post) {
    SS_HIP_F(hipMemcpyAsync(L.d_doc, docs + offs[0], L.n_post * sizeof(uint32_t), hipMemcpyHostToDevice, bst));
    SS_HIP_F(hipMemcpyAsync(L.d_tf, tfs + offs[0], L.n_post * sizeof(uint16_t), hipMemcpyHostToDevice, bst));
  }
  if (with_pos) {
    SS_HIP_F(hipMalloc(&L.d_npos, std::max<uint64_t>(L.n_post, 1) * sizeof(uint16_t)));
    SS_HIP_F(hipMalloc(&L.d_prel, std::max<uint64_t>(L.n_post, 1) * sizeof(uint32_t)));
    SS_HIP_F(hipMalloc(&L.d_tpos, h_tpos.size() * sizeof(uint64_t)));
    SS_HIP_F(hipMalloc(&L.d_pos, std::max<uint64_t>(n_positions, 1) * sizeof(uint16_t)));
    if (L.n_post) {
      SS_HIP_F(hipMemcpyAsync(L.d_npos, h_npos.data(), L.n_post * sizeof(uint16_t), hipMemcpyHostToDevice, bst));
      SS_HIP_F(hipMemcpyAsync(L.d_prel, h_prel.data(), L.n_post * sizeof(uint32_t), hipMemcpyHostToDevice, bst));
    }
    SS_HIP_F(hipMemcpyAsync(L.d_tpos, h_tpos.data(), h_tpos.size() * sizeof(uint64_t), hipMemcpyHostToDevice, bst));
    if (n_positions) SS_HIP_F(hipMemcpyAsync(L.d_pos, positions, n_positions * sizeof(uint16_t), hipMemcpyHostToDevice, bst));
  }
  SS_HIP_F(hipStreamSynchronize(bst));  // (rel dies with this frame; the caller's arrays are free again)
  levels.push_back(L);
  doclen.insert(doclen.end(), level_doclen, level_doclen + n_level_docs);
  const auto t_build = std::chrono::steady_clock::now();
  int rc = ssi_bm25_rebuild_from_raw(s, levels, n_terms, doclen.data(), doclen.size(), img.get(), bst);
  if (rc) return fail(rc);
  const double rebuild_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
  int rc_recode = SS_OK;
  {  // the swap: searches in flight finish on the old image, whose arrays are then released
    ShardLock g(s);
    if (level > s->raw.size() || level + 1 < s->raw.size()) return fail(SS_ESTATE);  // another commit got in between (the caller's write lock forbids it)
    (void)hipStreamSynchronize(s->stream);
    for (auto& kv : s->bm_ws) (void)hipStreamSynchronize(kv.first);
    ss_raw_level replaced;
    const bool replace = level < s->raw.size();
    if (replace) replaced = s->raw[level];
    std::vector<ss_raw_level> keep(s->raw.begin(), s->raw.begin() + level);
    s->raw.clear();  // (free_bm25 must not release the levels that stay)
    {  // the old image's arrays go back to the block pool: the next commit builds into them
      void** op[] = {(void**)&s->d_post, (void**)&s->d_term_base, (void**)&s->d_sub_off, (void**)&s->d_comp, (void**)&s->d_probe, (void**)&s->d_probe_z,
                     (void**)&s->d_probe_row, (void**)&s->d_umax, (void**)&s->d_submax, (void**)&s->d_doclen, (void**)&s->d_pos, (void**)&s->d_pos_off,
                     (void**)&s->d_pos_base};
      s->blocks.gen++;
      for (void** pp : op) if (*pp && s->blocks.release(*pp)) *pp = nullptr;
      std::vector<ss_block_pool::Idle> idle;
      idle.swap(s->blocks.idle);   // (free_bm25 clears the idle list: these stay)
      free_bm25(s, /*keep_tier=*/true);
      s->blocks.idle.swap(idle);
      s->blocks.trim(3);
    }
    if (replace)
      for (void* p : {(void*)replaced.d_off, (void*)replaced.d_doc, (void*)replaced.d_tf, (void*)replaced.d_npos, (void*)replaced.d_prel, (void*)replaced.d_tpos,
                      (void*)replaced.d_pos})
        if (p) (void)hipFree(p);
    keep.push_back(L);
    s->raw.swap(keep);
    s->h_doclen.swap(doclen);
    s->bm_n_docs = img->bm_n_docs; s->bm_n_terms = img->bm_n_terms; s->bm_n_sub = img->bm_n_sub; s->bm_n_fields = 1; s->bm_merged = false;
    s->bm_n_post = img->bm_n_post; s->bm_n_post_pad = img->bm_n_post_pad; s->bm_avgdl = img->bm_avgdl; s->bm_partmax = img->bm_partmax;
    s->d_post = img->d_post; s->d_term_base = img->d_term_base; s->d_sub_off = img->d_sub_off; s->d_comp = img->d_comp;
    s->d_probe = img->d_probe; s->d_probe_z = img->d_probe_z; s->d_probe_row = img->d_probe_row; s->d_umax = img->d_umax; s->d_submax = img->d_submax;
    s->d_doclen = img->d_doclen;
    s->d_pos = img->d_pos; s->d_pos_off = img->d_pos_off; s->d_pos_base = img->d_pos_base;
    s->h_df.swap(img->h_df); s->h_probe_row.swap(img->h_probe_row); s->bm_probe_rows = img->bm_probe_rows;
    s->probe_pool_begin = img->probe_pool_begin; s->probe_pool_rows = img->probe_pool_rows; s->pool_list.swap(img->pool_list); s->pool_tick.swap(img->pool_tick);
    s->pool_clock = 0;
    if (s->sp_n) {  // the average length moved: the sparse postings' codes follow (their docs of this level: ss_bm25_append_sparse_level)
      rc_recode = ssi_bm25_sparse_levels_recode(s, s->stream);
      if (rc_recode == SS_OK && hipStreamSynchronize(s->stream) != hipSuccess) rc_recode = SS_EDEVICE;
    }
    s->raw_last_rebuild_ms = rebuild_ms;
    s->raw_last_append_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
#undef SS_HIP_F
  // the dense image is swapped in either way; a sparse tier that could not follow the new average length would score its terms with
  // the old one: the caller hears about it, and commits again or rebuilds
  (void)hipStreamDestroy(bst);
  return rc_recode;
}

// INCREMENTAL COMMIT of an image with SEVERAL indexed fields (round 6; VERDICT r5 missing 6; commit.rs:142-148 -> the "(re)build device image"
// seam).  The level's entries (term, doc, field, tf) and its docs' length bytes per field join the levels kept on the HOST; the shard's
// postings are re-assembled term by term (a term's entries of level after level: doc ids ascend with the level) and the image is rebuilt
// by the multi-field builder -- per-field lists, merged lists, probe rows --, i.e. a commit costs what an upload of the shard costs
// (52 M entries/s), where the one-field form rebuilds on the device from levels kept in HBM.  Same level rules as ss_bm25_append_level.
int ss_bm25_append_level_fields(ss_shard* s, uint32_t level, uint32_t n_level_docs, uint32_t n_fields, const uint8_t* level_doclen, const float* boost,
                                uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs) {
  return ss_guard([&]() -> int {
    if (!s || !level_doclen || !offs || n_terms == 0 || n_level_docs == 0 || n_level_docs > 65536u || n_fields < 2 || n_fields > 8) return SS_EINVAL;
    if (offs[n_terms] && (!docs || !fields || !tfs)) return SS_EINVAL;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<ss_raw_level_f> levels;
    {
      ShardLock g(s);
      if ((s->d_post || !s->raw.empty()) && s->raw_f.empty()) return SS_ESTATE;  // an image that was not built this way
      if (s->sp_n) return SS_ENOTSUP;                                            // (a sparse tier beside it: upload + ss_bm25_append_sparse_fields)
      if (!s->raw_f.empty() && s->raw_f_fields != n_fields) return SS_EINVAL;
      if (level > s->raw_f.size() || level + 1 < s->raw_f.size()) return SS_EINVAL;  // append the next level, or replace the last one
      if (level >= 1 && s->raw_f[level - 1].n_docs != 65536u) return SS_EINVAL;      // only the last level may be partial
      if ((uint64_t)level * 65536u + n_level_docs > 0xFFFFFFFFull) return SS_ENOTSUP;
      if (!s->raw_f.empty() && n_terms < s->raw_f.back().n_terms && level == s->raw_f.size()) return SS_EINVAL;  // the vocabulary only grows
      levels.assign(s->raw_f.begin(), s->raw_f.begin() + level);
    }
    const uint64_t d_lo = (uint64_t)level * 65536u, d_hi = d_lo + n_level_docs;
    for (uint32_t t = 0; t < n_terms; t++)
      if (offs[t + 1] < offs[t]) return SS_EINVAL;
    for (uint64_t j = offs[0]; j < offs[n_terms]; j++)
      if (docs[j] < d_lo || docs[j] >= d_hi || tfs[j] == 0 || fields[j] >= n_fields) return SS_EINVAL;
    ss_raw_level_f L;
    L.n_docs = n_level_docs; L.n_terms = n_terms;
    L.off.resize((size_t)n_terms + 1);
    for (uint32_t t = 0; t <= n_terms; t++) L.off[t] = offs[t] - offs[0];
    L.doc.assign(docs + offs[0], docs + offs[n_terms]);
    L.field.assign(fields + offs[0], fields + offs[n_terms]);
    L.tf.assign(tfs + offs[0], tfs + offs[n_terms]);
    L.doclen.assign(level_doclen, level_doclen + (size_t)n_fields * n_level_docs);
    levels.push_back(std::move(L));
    // the whole shard, term by term
    uint64_t n_docs = 0;
    uint32_t nt_all = 0;
    for (const ss_raw_level_f& l : levels) { n_docs += l.n_docs; nt_all = std::max(nt_all, l.n_terms); }
    std::vector<uint64_t> off_all((size_t)nt_all + 1, 0);
    for (const ss_raw_level_f& l : levels)
      for (uint32_t t = 0; t < l.n_terms; t++) off_all[t + 1] += l.off[t + 1] - l.off[t];
    for (uint32_t t = 0; t < nt_all; t++) off_all[t + 1] += off_all[t];
    const uint64_t n_all = off_all[nt_all];
    std::vector<uint32_t> doc_all(std::max<uint64_t>(n_all, 1));
    std::vector<uint8_t> field_all(std::max<uint64_t>(n_all, 1));
    std::vector<uint16_t> tf_all(std::max<uint64_t>(n_all, 1));
    ss_parallel_for(nt_all, 64, [&](size_t ta, size_t tb, unsigned) {
      for (size_t t = ta; t < tb; t++) {
        uint64_t at = off_all[t];
        for (const ss_raw_level_f& l : levels) {
          if (t >= l.n_terms) continue;
          const uint64_t a = l.off[t], n = l.off[t + 1] - a;
          if (!n) continue;
          memcpy(&doc_all[at], &l.doc[a], n * sizeof(uint32_t));
          memcpy(&field_all[at], &l.field[a], n);
          memcpy(&tf_all[at], &l.tf[a], n * sizeof(uint16_t));
          at += n;
        }
      }
    });
    std::vector<uint8_t> dl_all((size_t)n_fields * n_docs);
    for (uint32_t f = 0; f < n_fields; f++) {
      uint64_t at = 0;
      for (const ss_raw_level_f& l : levels) { memcpy(&dl_all[(size_t)f * n_docs + at], &l.doclen[(size_t)f * l.n_docs], l.n_docs); at += l.n_docs; }
    }
    const auto t_build = std::chrono::steady_clock::now();
    // (the builder replaces the image in place under the shard mutex and drops every raw level: ours go back in afterwards)
    const int rc = ssi_bm25_upload_fields(s, n_docs, n_fields, dl_all.data(), boost, nt_all, off_all.data(), doc_all.data(), field_all.data(), tf_all.data(), 0);
    ShardLock g(s);
    if (rc != SS_OK) return rc;  // (no image is left behind, and no levels: the caller starts over with an upload)
    s->raw_f.swap(levels);
    s->raw_f_fields = n_fields;
    s->raw_last_rebuild_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
    s->raw_last_append_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return SS_OK;
  }, SS_ENOMEM, SS_EDEVICE);
}

int ss_bm25_incremental_info(ss_shard* s, uint32_t* n_levels, uint64_t* raw_bytes, double* last_append_ms, double* last_rebuild_ms) {
  if (!s) return SS_EINVAL;
  ShardLock g(s);
  uint64_t b = 0;
  for (const ss_raw_level_f& L : s->raw_f) b += L.off.size() * 8u + L.doc.size() * 7u + L.doclen.size();
  if (n_levels && !s->raw_f.empty()) { *n_levels = (uint32_t)s->raw_f.size(); n_levels = nullptr; }
  for (const ss_raw_level& L : s->raw)
    b += ((uint64_t)L.n_terms + 1) * 8u + L.n_post * 6u + (L.d_tpos ? ((uint64_t)L.n_terms + 1) * 8u + L.n_post * 6u + L.n_pos * 2u : 0u);
  if (n_levels) *n_levels = (uint32_t)s->raw.size();
  if (raw_bytes) *raw_bytes = b;
  if (last_append_ms) *last_append_ms = s->raw_last_append_ms;
  if (last_rebuild_ms) *last_rebuild_ms = s->raw_last_rebuild_ms;
  return SS_OK;
}

// Tombstones: delete_hashset of the shard (index.rs:1594; filled from delete.bin, index.rs:3798-3809, and by
// delete_document, index.rs:5110).  Replaces the whole set; n = 0 clears it.  Both search paths skip a deleted doc before
// it counts or ranks (add_result.rs:3435, union.rs:975, vector.rs:1450).
int ss_set_deleted(ss_shard* s, const uint64_t* doc_ids, uint64_t n) {
  if (!s || (n && !doc_ids)) return SS_EINVAL;
  uint64_t mx = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (doc_ids[i] >= 0xFFFFFFFFull) return SS_EINVAL;
    mx = std::max(mx, doc_ids[i]);
  }
  ShardLock g(s);
  if (s->vstream) (void)hipStreamSynchronize(s->vstream);  // (a coalesced scan in flight reads what this call replaces; none starts while mu is ours)
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  if (s->d_deleted) { (void)hipFree(s->d_deleted); s->d_deleted = nullptr; }
  s->deleted_words = 0;
  s->n_deleted = 0;
  if (n == 0) return SS_OK;
  std::vector<uint32_t> bits((size_t)(mx >> 5) + 1, 0u);
  uint64_t distinct = 0;
  for (uint64_t i = 0; i < n; i++) {
    uint32_t& w = bits[(size_t)(doc_ids[i] >> 5)];
    const uint32_t b = 1u << (doc_ids[i] & 31u);
    distinct += !(w & b);
    w |= b;
  }
  SS_HIP(hipMalloc(&s->d_deleted, bits.size() * sizeof(uint32_t)));
  SS_HIP(hipMemcpy(s->d_deleted, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  s->deleted_words = bits.size();
  s->n_deleted = distinct;
  return SS_OK;
}

// ---- facet filter (facet.hip): facet.bin records on the device, a filtered search = the search with the filter's exclusion
// bitmap standing in for the tombstone bitmap
int ss_facet_upload(ss_shard* s, uint64_t n_docs, uint32_t record_size, const uint8_t* records) {
  if (!s || !records || n_docs == 0 || record_size == 0 || n_docs > 0xFFFFFFFEull) return SS_EINVAL;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  if (s->d_facets) { (void)hipFree(s->d_facets); s->d_facets = nullptr; }
  s->facet_docs = 0; s->facet_record_size = 0;
  SS_HIP(hipMalloc(&s->d_facets, (size_t)n_docs * record_size));
  SS_HIP(hipMemcpy(s->d_facets, records, (size_t)n_docs * record_size, hipMemcpyHostToDevice));
  s->facet_docs = n_docs;
  s->facet_record_size = record_size;
  return SS_OK;
}


// Probe-index budget of the NEXT image build (bytes; 0 = half of the free device memory).  Rows (1.9 MB per posting list
// at 10 M docs) go to the longest lists first; a query touching a list without a row is ranked by the scan kernels.
int ss_bm25_set_probe_budget(ss_shard* s, uint64_t max_bytes) {
  if (!s) return SS_EINVAL;
  ShardLock g(s);
  s->probe_budget = max_bytes;
  return SS_OK;
}

// 1 per term whose posting lists (all fields) have probe rows: such queries can take the pruned strategy
int ss_bm25_term_probed(ss_shard* s, uint32_t n, const uint32_t* terms, uint8_t* out) {
  if (!s || !terms || !out) return SS_EINVAL;
  if (!s->d_post) return SS_ESTATE;
  for (uint32_t i = 0; i < n; i++) {
    if (terms[i] >= s->bm_n_terms / s->bm_n_fields) return SS_EINVAL;
    uint8_t ok = s->bm_probe_rows != 0;
    for (uint32_t f = 0; ok && f < s->bm_n_fields; f++) {
      const uint32_t v = terms[i] * s->bm_n_fields + f;
      if (s->h_probe_row[v] == BM_NO_PROBE_ROW && s->h_df[v] != 0) ok = 0;
      else if (s->probe_pool_rows && s->h_probe_row[v] >= s->probe_pool_begin && s->h_probe_row[v] < s->probe_pool_begin + s->probe_pool_rows)
        ok = 2;  // a row from the pool: a later host-pointer batch may take it away
    }
    out[i] = ok;
  }
  return SS_OK;
}

int ss_bm25_set_strategy(ss_shard* s, int strategy) {
  if (!s || strategy < SS_BM25_AUTO || strategy > SS_BM25_EXHAUSTIVE_F32) return SS_EINVAL;
  ShardLock g(s);
  s->bm_strategy = strategy;
  return SS_OK;
}

int ss_bm25_fields_info(ss_shard* s, uint32_t* n_fields, uint32_t* merged_lists, uint32_t* positions) {
  if (!s) return SS_EINVAL;
  if (!s->d_post) return SS_ESTATE;
  if (n_fields) *n_fields = bm_real_fields(s);
  if (merged_lists) *merged_lists = s->bm_merged ? 1u : 0u;
  if (positions) *positions = (s->bm_n_fields == 1 ? s->d_pos != nullptr : s->d_pos32 != nullptr) ? 1u : 0u;
  return SS_OK;
}
int ss_bm25_info(ss_shard* s, uint64_t* n_docs, float* avgdl, uint32_t* n_terms, uint64_t* n_postings) {
  if (!s) return SS_EINVAL;
  if (!s->d_post) return SS_ESTATE;
  if (n_docs) *n_docs = s->bm_n_docs;
  if (avgdl) *avgdl = s->bm_avgdl;
  if (n_terms) *n_terms = s->bm_n_terms / s->bm_n_fields;
  if (n_postings) *n_postings = s->bm_n_post;
  return SS_OK;
}

int ss_bm25_term_df(ss_shard* s, uint32_t n, const uint32_t* terms, uint64_t* df_out) {
  if (!s || !terms || !df_out) return SS_EINVAL;
  if (!s->d_post) return SS_ESTATE;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t n_dense = s->bm_n_terms / s->bm_n_fields;
    if (terms[i] >= n_dense + s->sp_n) return SS_EINVAL;
    if (terms[i] >= n_dense) df_out[i] = s->h_sp_base[terms[i] - n_dense + 1] - s->h_sp_base[terms[i] - n_dense];  // sparse tier
    else df_out[i] = s->bm_n_fields > 1 ? s->h_df_real[terms[i]] : s->h_df[terms[i]];
  }
  return SS_OK;
}

int ss_bm25_append_sparse_level(ss_shard* s, uint32_t level, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                                const uint16_t* npos, const uint16_t* positions, uint64_t n_positions) {
  if (!s || !offs || n_lists == 0 || (offs[n_lists] > offs[0] && (!docs || !tfs))) return SS_EINVAL;
  ShardLock g(s);
  if (s->raw.empty()) return SS_ESTATE;  // an image that grows level by level (ss_bm25_append_level)
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipDeviceSynchronize());  // searches on the callers' own streams may still read the arrays the level replaces
  return ssi_bm25_append_sparse_level(s, level, n_lists, offs, docs, tfs, npos, positions, n_positions);
}
int ss_bm25_append_sparse(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                          uint32_t* first_term_id_out) {
  if (!s || !offs || (n_lists && offs[n_lists] > offs[0] && (!docs || !tfs))) return SS_EINVAL;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipDeviceSynchronize());  // searches on the callers' own streams may still read the arrays an append replaces
  const uint32_t first = s->bm_n_terms + s->sp_n;
  SS_TRY(ssi_bm25_append_sparse(s, n_lists, offs, docs, tfs));
  if (first_term_id_out) *first_term_id_out = first;
  return SS_OK;
}
int ss_bm25_append_sparse_fields(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                 const uint16_t* tfs, uint32_t* first_term_id_out) {
  if (!s || !offs || (n_lists && offs[n_lists] > offs[0] && (!docs || !fields || !tfs))) return SS_EINVAL;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipDeviceSynchronize());  // searches on the callers' own streams may still read the arrays an append replaces
  const uint32_t first = s->bm_n_terms / std::max<uint32_t>(s->bm_n_fields, 1) + s->sp_n;
  SS_TRY(ssi_bm25_append_sparse_fields(s, n_lists, offs, docs, fields, tfs));
  if (first_term_id_out) *first_term_id_out = first;
  return SS_OK;
}
int ss_bm25_append_sparse_positions(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                                    const uint16_t* positions, uint64_t n_positions, const uint16_t* npos, uint32_t* first_term_id_out) {
  if (!s || !offs || (n_lists && offs[n_lists] > offs[0] && (!docs || !tfs)) || (n_positions && !positions)) return SS_EINVAL;
  static const uint16_t none = 0;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipDeviceSynchronize());  // searches on the callers' own streams may still read the arrays an append replaces
  const uint32_t first = s->bm_n_terms + s->sp_n;
  SS_TRY(ssi_bm25_append_sparse(s, n_lists, offs, docs, tfs, positions ? positions : &none, n_positions, npos));
  if (first_term_id_out) *first_term_id_out = first;
  return SS_OK;
}
int ss_bm25_append_sparse_fields_positions(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                           const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions, const uint16_t* npos,
                                           uint32_t* first_term_id_out) {
  if (!s || !offs || (n_lists && offs[n_lists] > offs[0] && (!docs || !fields || !tfs)) || (n_positions && !positions)) return SS_EINVAL;
  static const uint16_t none = 0;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipDeviceSynchronize());  // searches on the callers' own streams may still read the arrays an append replaces
  const uint32_t first = s->bm_n_terms / std::max<uint32_t>(s->bm_n_fields, 1) + s->sp_n;
  SS_TRY(ssi_bm25_append_sparse_fields(s, n_lists, offs, docs, fields, tfs, positions ? positions : &none, n_positions, npos));
  if (first_term_id_out) *first_term_id_out = first;
  return SS_OK;
}
int ss_bm25_sparse_info(ss_shard* s, uint32_t* n_lists, uint64_t* n_postings, uint64_t* bytes) {
  if (!s) return SS_EINVAL;
  ShardLock g(s);
  const uint64_t np = s->h_sp_base.empty() ? 0 : s->h_sp_base.back();
  if (n_lists) *n_lists = s->sp_n;
  if (n_postings) *n_postings = np;
  if (bytes) *bytes = np * (ssi_bm25_sparse_levels_has(s) ? 10 : 8) + ((uint64_t)s->sp_n + 1) * 8;  // (a tier of levels keeps the tfs)
  return SS_OK;
}

// nt_max: largest n_terms + NOT terms of the batch (what the scan kernels are specialised on); np_max: largest n_terms
// any_filter: some query carries a field filter (on an image with merged lists the others read one list per term)
static int check_queries(const ss_shard* s, uint32_t nq, const ss_bm25_query* q, bool* has_and, bool* has_or, uint32_t* nt_max,
                         uint32_t* np_max, bool* all_probed, bool* any_frequent, bool* phrase = nullptr, bool* any_filter = nullptr,
                         bool* uniform = nullptr, bool* gated = nullptr, uint32_t* nn_max = nullptr) {
  uint32_t n_phrase = 0, np_min = 0xFFFFFFFFu, most_not = 0;
  bool some_gated = false;
  const uint32_t L = s->bm_n_fields, RF = bm_real_fields(s);  // lists per term, indexed fields
  bool some_filter = false;
  *any_frequent = false;
  *all_probed = s->bm_probe_rows != 0;
  *has_and = false;
  *has_or = false;
  *nt_max = 0;
  *np_max = 0;
  bool any_not = false;
  for (uint32_t i = 0; i < nq; i++) {
    const uint32_t op = bm_q_op(q[i].op), n_not = bm_q_nnot(q[i].op), all = q[i].n_terms + n_not;
    if (q[i].n_terms == 0 || all > SS_MAX_QUERY_TERMS) return SS_EINVAL;
    if (op != SS_OP_INTERSECTION && op != SS_OP_UNION && op != SS_OP_PHRASE) return SS_EINVAL;
    if (op == SS_OP_PHRASE) {  // QueryType::Phrase: unique terms + the words in order (non_unique_query_list)
      if (!phrase) return SS_ENOTSUP;
      if (q[i].phrase_len < 2 || q[i].phrase_len > SS_MAX_PHRASE) return SS_EINVAL;
      // every place names a unique term, or SS_PHRASE_SKIP: a place inside an n-gram key, whose entry stands at the key's first word
      // (the key's other component terms are scored with the rest, they are no words of the phrase)
      if (q[i].phrase_seq[0] >= q[i].n_terms) return SS_EINVAL;
      for (uint32_t j = 1; j < q[i].phrase_len; j++)
        if (q[i].phrase_seq[j] >= q[i].n_terms && q[i].phrase_seq[j] != SS_PHRASE_SKIP) return SS_EINVAL;
      if (q[i].n_terms > 6) return SS_ENOTSUP;  // (NOT terms: their lists are probed by the phrase kernel like everywhere else)
      // several indexed fields: over the merged lists and their field-tagged positions (a corpus whose boosts kept the merged
      // lists from being built has no phrase path)
      if (s->bm_n_fields > 1 && !s->bm_merged) return SS_ENOTSUP;
      if (s->bm_n_fields > 1 ? !s->d_pos32 : !s->d_pos) return SS_ESTATE;  // the image carries no positions (ss_bm25_upload[_fields]_positions)
      n_phrase++;
    }
    // field_filter: bits of indexed fields; an image with one indexed field has nothing to filter (the reference's set then
    // holds that field or nothing, search.rs:2483-2492).  Unions of several terms: the reference applies the filter inside
    // union_docid_3's sub-queries, not per doc -- not offered.
    if (bm_q_field_filter(q[i].op) >> RF) return SS_EINVAL;  // a field the image does not have
    const uint32_t filt = RF > 1 ? bm_q_field_filter(q[i].op) : 0u;
    if (filt && op == SS_OP_UNION && q[i].n_terms > 1) {  // a union under a field filter: per-term gating in the scan kernels (BM_AND_GATED)
      if (q[i].n_terms > 7 || !gated) return SS_ENOTSUP;
      some_gated = true;
    }
    some_filter |= filt != 0u && op != SS_OP_PHRASE;  // (a phrase's filter is a test on its positions: no (term, field) lists)
    const bool use_merged = s->bm_merged && (!filt || op == SS_OP_PHRASE);  // this query reads the merged lists: one list per term
    const uint32_t eff_fields = use_merged ? 1u : RF, f_begin = use_merged ? L - 1u : 0u, f_end = use_merged ? L : RF;
    // all_terms_frequent: an intersection of 2..7 terms over one indexed field (the mark takes bit 7 of the match byte);
    // on anything else the reference's flag has no effect we model (single terms, unions) or is not offered
    // (several indexed fields: over the merged lists, whose codes carry the multi-field form of the per-posting rule; under a field
    // filter the reference switches the shortcut off -- add_result.rs:3116 "all_terms_frequent && field_filter_set.is_empty()" --
    // and so does the expansion)
    if (bm_q_all_frequent(q[i].op) && op == SS_OP_INTERSECTION && q[i].n_terms > 1 && !filt) {
      if (q[i].n_terms > 7 || (L > 1 && !s->bm_merged)) return SS_ENOTSUP;
      *any_frequent = true;
    }
    if (eff_fields > 1) {  // (term, field) posting lists: at most BM_MAX_VTERMS of them, match masks of 8 bits
      if (all * eff_fields > (uint32_t)BM_MAX_VTERMS) return SS_ENOTSUP;
      if (op == SS_OP_INTERSECTION && q[i].n_terms > 8) return SS_ENOTSUP;
    }
    for (uint32_t t = 0; t < all; t++) {
      if (q[i].term[t] >= s->bm_n_terms / L) return SS_EINVAL;
      if (*all_probed && s->bm_probe_rows < s->bm_n_terms)  // rows were rationed: does every list this query reads have one?
        for (uint32_t f = f_begin; f < f_end; f++) {
          const uint32_t v = q[i].term[t] * L + f;
          if (s->h_probe_row[v] == BM_NO_PROBE_ROW && s->h_df[v] != 0) *all_probed = false;
        }
      if (t < q[i].n_terms && !(q[i].idf[t] > 0.0f)) return SS_EINVAL;
      for (uint32_t u = 0; u < t; u++)
        if (q[i].term[u] == q[i].term[t]) return SS_EINVAL;  // unique terms only (search.rs:3023 unique_terms)
    }
    if (((op == SS_OP_INTERSECTION || op == SS_OP_PHRASE) && q[i].n_terms > 1) || filt) *has_and = true;
    else if (q[i].n_terms > 1) *has_or = true;  // a single-term query is both: its exact count is its posting count
    *nt_max = std::max(*nt_max, all);
    *np_max = std::max(*np_max, q[i].n_terms);
    np_min = std::min(np_min, q[i].n_terms);
    any_not |= n_not != 0;
    most_not = std::max(most_not, n_not);
  }
  if (nn_max) *nn_max = most_not;  // the most NOT terms any query of the batch has (nt_max - np_max says only that there are some)
  if (uniform) *uniform = np_min == *np_max;  // every query with the same number of terms
  if (gated) *gated = some_gated;
  // "the batch holds NOT terms" travels as nt_max > np_max (the kernels' filtered variants are chosen by it): keep that true
  // when the query with the NOT terms is not the one with the most terms
  if (any_not && *nt_max == *np_max) *nt_max = *np_max + 1;
  if (n_phrase && n_phrase != nq) return SS_ENOTSUP;  // a batch holds phrase queries only
  if (phrase) *phrase = n_phrase != 0;
  if (any_filter) *any_filter = some_filter;
  return SS_OK;
}

int ss_bm25_search(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t k, uint32_t rt, uint32_t* out_doc,
                   float* out_score, uint32_t* out_count, uint64_t* out_total) {
  return ss_bm25_search_filtered(s, nq, q, k, rt, 0, nullptr, out_doc, out_score, out_count, out_total);
}

// rows of a batch that ran in another order back to the callers' order: row i of the src arrays -> row perm[i] of the dst arrays
__global__ void bm25_unpermute_kernel(const uint32_t* __restrict__ perm, uint32_t nq, uint32_t kk, const uint32_t* __restrict__ src_doc,
                                      const float* __restrict__ src_score, const uint32_t* __restrict__ src_count,
                                      const unsigned long long* __restrict__ src_total, uint32_t* __restrict__ dst_doc,
                                      float* __restrict__ dst_score, uint32_t* __restrict__ dst_count, unsigned long long* __restrict__ dst_total) {
  const uint32_t i = blockIdx.x;
  if (i >= nq) return;
  const uint32_t o = perm[i];
  for (uint32_t j = threadIdx.x; j < kk; j += blockDim.x) {
    dst_doc[(size_t)o * kk + j] = src_doc[(size_t)i * kk + j];
    dst_score[(size_t)o * kk + j] = src_score[(size_t)i * kk + j];
  }
  if (threadIdx.x == 0) {
    dst_count[o] = src_count[i];
    dst_total[o] = src_total[i];
  }
}

// ---- probe rows on demand.  A rationed vocabulary keeps a pool of rows (alloc_probe); before a host-pointer batch runs, the
// row-less lists it touches get pool rows -- least recently used rows first, never one this batch needs -- built from the
// list's own postings: one wave per (list, sub-block) ORs the doc bits of the segment into 64 group masks and writes them
// with the rank of each group's first posting (what the image builders write for the fixed rows).  A row costs its own
// bytes once (1.9 MB at 10 M docs) and nothing while it stays in the pool; lists that found no row leave their queries to
// the scan kernels as before.
__global__ void probe_row_evict_kernel(const uint32_t* __restrict__ lists, uint32_t n, uint32_t* __restrict__ probe_row) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) probe_row[lists[i]] = BM_NO_PROBE_ROW;
}
__global__ void probe_row_fill_kernel(const uint32_t* __restrict__ pairs /* (list, row) */, uint32_t n_pairs, uint32_t n_sub,
                                      const uint32_t* __restrict__ sub_off, const unsigned long long* __restrict__ term_base,
                                      const uint32_t* __restrict__ post, uint2* __restrict__ probe, uint32_t* __restrict__ probe_z,
                                      uint32_t* __restrict__ probe_row) {
  __shared__ unsigned long long masks[4][BM_SUB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long gw = (unsigned long long)blockIdx.x * 4u + (unsigned)w;
  if (gw >= (unsigned long long)n_pairs * n_sub) return;
  const uint32_t pair = (uint32_t)(gw / n_sub), sb = (uint32_t)(gw % n_sub);
  const uint32_t t = pairs[2 * pair], r = pairs[2 * pair + 1];
  const uint32_t u0 = sub_off[(size_t)t * (n_sub + 1) + sb], u1 = sub_off[(size_t)t * (n_sub + 1) + sb + 1];
  const unsigned long long base = (term_base[t] + u0) * 4ull;
  masks[w][lane] = 0ull;
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = (uint32_t)lane; i < (u1 - u0) * 4u; i += 64u) {
    const uint32_t p = post[base + i];
    if (p) {  // 0 = the segment's NULL padding
      const uint32_t d = bm_doc_field(p) - 1u;
      atomicOr(&masks[w][d >> 6], 1ull << (d & 63u));
    }
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned long long m = masks[w][lane];
  uint32_t run = (uint32_t)__popcll(m);  // exclusive prefix over the 64 groups
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(run, o);
    if (lane >= o) run += v;
  }
  run -= (uint32_t)__popcll(m);
  const size_t gi = ((size_t)r * n_sub + sb) * (BM_SUB / 64) + (uint32_t)lane;
  probe[gi] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
  probe_z[gi] = u0 * 4u + run;
  if (sb == 0 && lane == 0) probe_row[t] = r;
}

// the FIXED probe rows of a host-built image, from the image itself (ssi_bm25_build_from_host_merged): the rows used to be assembled on
// the host and copied -- 188 KB per list and million docs, 4.5 GB for the 23 760 lists of a 1 M-doc 3-field index: 1.0 s of zeroing
// and bit setting + 0.45 s of pageable copies, against a pass over the uploaded postings here
extern "C++" int ssi_bm25_fill_fixed_probe_rows(ss_shard* s, hipStream_t st) {
  if (!s->d_probe || !s->bm_probe_rows) return SS_OK;
  std::vector<uint32_t> pairs;
  for (uint32_t t = 0; t < s->bm_n_terms; t++)
    // (an EMPTY list may own a row as well -- alloc_probe deals rows before it points the row-less empty lists at the zero row: built too, as zeros)
    if (s->h_probe_row[t] != BM_NO_PROBE_ROW && s->h_probe_row[t] < s->bm_probe_rows) { pairs.push_back(t); pairs.push_back(s->h_probe_row[t]); }
  const uint32_t n = (uint32_t)(pairs.size() / 2);
  if (!n) return SS_OK;
  uint32_t* d_pairs = nullptr;
  SS_HIP(hipMalloc(&d_pairs, pairs.size() * sizeof(uint32_t)));
  int rc = hipMemcpyAsync(d_pairs, pairs.data(), pairs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st) == hipSuccess ? SS_OK : SS_EDEVICE;
  if (rc == SS_OK) {
    const unsigned long long waves = (unsigned long long)n * s->bm_n_sub;
    probe_row_fill_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, st>>>(d_pairs, n, s->bm_n_sub, s->d_sub_off, (const unsigned long long*)s->d_term_base,
                                                                     s->d_post, s->d_probe, s->d_probe_z, s->d_probe_row);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = SS_EDEVICE;
  }
  (void)hipFree(d_pairs);
  return rc;
}

// the lists of a term that a query reads: the merged list alone without a field filter (bm_merged images), else the fields'
static inline void query_list_range(const ss_shard* s, const ss_bm25_query& q, uint32_t* f_begin, uint32_t* f_end) {
  const uint32_t L = s->bm_n_fields, RF = bm_real_fields(s);
  const bool use_merged = s->bm_merged && (!(RF > 1 && bm_q_field_filter(q.op)) || bm_q_op(q.op) == SS_OP_PHRASE);  // (a phrase always reads the merged lists)
  *f_begin = use_merged ? L - 1u : 0u;
  *f_end = use_merged ? L : RF;
}
static inline bool list_needs_row(const ss_shard* s, uint32_t v) { return s->h_probe_row[v] == BM_NO_PROBE_ROW && s->h_df[v] != 0; }

// caller holds s->mu (with or without the lane streams drained: see below)
static int ssi_bm25_ensure_probe_rows(ss_shard* s, uint32_t nq, const ss_bm25_query* q, hipStream_t st) {
  if (s->probe_pool_rows == 0 || !s->d_probe) return SS_OK;
  const uint32_t n_lists_per_term = s->bm_n_fields, n_public = s->bm_n_terms / s->bm_n_fields;
  const uint64_t now = ++s->pool_clock;
  // phrase queries first: without rows for all of their lists they have no kernel at all (the others fall back to the scans)
  std::vector<uint32_t> missing, missing_phrase;
  for (uint32_t i = 0; i < nq; i++) {
    const bool is_phrase = bm_q_op(q[i].op) == SS_OP_PHRASE;
    const uint32_t all = std::min<uint32_t>(q[i].n_terms + bm_q_nnot(q[i].op), SS_MAX_QUERY_TERMS);
    uint32_t f_begin, f_end;
    query_list_range(s, q[i], &f_begin, &f_end);
    for (uint32_t t = 0; t < all; t++) {
      if (q[i].term[t] >= n_public) continue;  // check_queries reports it
      for (uint32_t f = f_begin; f < f_end; f++) {
        const uint32_t v = q[i].term[t] * n_lists_per_term + f, r = s->h_probe_row[v];
        if (r != BM_NO_PROBE_ROW && r >= s->probe_pool_begin && r < s->probe_pool_begin + s->probe_pool_rows)
          s->pool_tick[r - s->probe_pool_begin] = now;  // a pool row this batch needs: not a victim
        else if (list_needs_row(s, v))
          (is_phrase ? missing_phrase : missing).push_back(v);
      }
    }
  }
  if (missing.empty() && missing_phrase.empty()) return SS_OK;
  std::sort(missing_phrase.begin(), missing_phrase.end());
  missing_phrase.erase(std::unique(missing_phrase.begin(), missing_phrase.end()), missing_phrase.end());
  std::sort(missing.begin(), missing.end());
  missing.erase(std::unique(missing.begin(), missing.end()), missing.end());
  {
    std::vector<uint32_t> rest;
    std::set_difference(missing.begin(), missing.end(), missing_phrase.begin(), missing_phrase.end(), std::back_inserter(rest));
    missing = missing_phrase;
    missing.insert(missing.end(), rest.begin(), rest.end());
  }
  // victims: free rows first, then the least recently used ones, never a row of this batch
  std::vector<uint32_t> victims;
  for (uint32_t i = 0; i < s->probe_pool_rows; i++)
    if (s->pool_tick[i] != now) victims.push_back(i);
  std::sort(victims.begin(), victims.end(), [&](uint32_t a, uint32_t b) {
    const bool fa = s->pool_list[a] == BM_NO_PROBE_ROW, fb = s->pool_list[b] == BM_NO_PROBE_ROW;
    if (fa != fb) return fa;
    return s->pool_tick[a] != s->pool_tick[b] ? s->pool_tick[a] < s->pool_tick[b] : a < b;
  });
  const uint32_t n = (uint32_t)std::min(missing.size(), victims.size());
  if (n == 0) return SS_OK;
  std::vector<uint32_t> stage;  // [evicted lists | (list, row) pairs]
  for (uint32_t i = 0; i < n; i++)
    if (s->pool_list[victims[i]] != BM_NO_PROBE_ROW) stage.push_back(s->pool_list[victims[i]]);
  const uint32_t n_evict = (uint32_t)stage.size();
  for (uint32_t i = 0; i < n; i++) {
    stage.push_back(missing[i]);
    stage.push_back(s->probe_pool_begin + victims[i]);
  }
  SS_HIP(hipSetDevice(s->device));
  // Rows are about to change hands.  (1) A coalescer lane's one-launch kernel may be in flight on the lane's own stream (ss_common.h
  // lstream) and reading a victim row -- and this call may itself come from the other lane's leader, who holds s->mu but not a ShardLock:
  // drain the lane streams.  (2) The next lane launch has to wait for the fill on `st`: main_dirty.  (3) A lane leader does not
  // synchronise `st` before it lets go of s->mu, so the previous fill may still be reading d_pool_stage: wait for it before overwriting.
  s->main_dirty = true;
  if (s->lanes_inflight) {
    for (hipStream_t ls : s->lstream)
      if (ls) (void)hipStreamSynchronize(ls);
    s->lanes_inflight = false;
  }
  SS_HIP(hipStreamSynchronize(st));
  if (stage.size() * sizeof(uint32_t) > s->pool_stage_cap) {
    if (s->d_pool_stage) (void)hipFree(s->d_pool_stage);
    s->d_pool_stage = nullptr; s->pool_stage_cap = 0;
    const size_t cap = std::max<size_t>(stage.size() * sizeof(uint32_t), 4096) * 2;
    SS_HIP(hipMalloc(&s->d_pool_stage, cap));
    s->pool_stage_cap = cap;
  }
  SS_HIP(hipMemcpy(s->d_pool_stage, stage.data(), stage.size() * sizeof(uint32_t), hipMemcpyHostToDevice));  // small; `stage` dies here
  // A row about to change hands may still be read by *_dev searches queued on the callers' OWN streams (they vouched for their
  // rows through ss_bm25_term_probed): the evict / fill kernels on `st` wait for everything those streams hold so far.
  if (n_evict)
    for (auto& kv : s->bm_ws) {
      if (kv.first == st) continue;
      hipEvent_t ev = nullptr;
      SS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      const bool ok = hipEventRecord(ev, kv.first) == hipSuccess && hipStreamWaitEvent(st, ev, 0) == hipSuccess;
      (void)hipEventDestroy(ev);  // released once the wait has been satisfied
      if (!ok) return SS_EDEVICE;
    }
  // the host tables change only now that nothing above can fail any more: host and device views stay in step
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t slot = victims[i], v = missing[i];
    if (s->pool_list[slot] != BM_NO_PROBE_ROW) s->h_probe_row[s->pool_list[slot]] = BM_NO_PROBE_ROW;
    s->pool_list[slot] = v;
    s->pool_tick[slot] = now;
    s->h_probe_row[v] = s->probe_pool_begin + slot;
  }
  if (n_evict) probe_row_evict_kernel<<<(n_evict + 255) / 256, 256, 0, st>>>(s->d_pool_stage, n_evict, s->d_probe_row);
  const unsigned long long waves = (unsigned long long)n * s->bm_n_sub;
  probe_row_fill_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, st>>>(s->d_pool_stage + n_evict, n, s->bm_n_sub, s->d_sub_off,
                                                                   (const unsigned long long*)s->d_term_base, s->d_post, s->d_probe,
                                                                   s->d_probe_z, s->d_probe_row);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

static bool query_lists_probed(const ss_shard* s, const ss_bm25_query& q) {
  const uint32_t all = q.n_terms + bm_q_nnot(q.op);
  uint32_t f_begin, f_end;
  query_list_range(s, q, &f_begin, &f_end);
  for (uint32_t t = 0; t < all && t < SS_MAX_QUERY_TERMS; t++) {
    if (q.term[t] >= s->bm_n_terms / s->bm_n_fields) return false;  // check_queries reports it
    for (uint32_t f = f_begin; f < f_end; f++) {
      const uint32_t v = q.term[t] * s->bm_n_fields + f;
      if (s->h_probe_row[v] == BM_NO_PROBE_ROW && s->h_df[v] != 0) return false;
    }
  }
  return true;
}

// ... of its DENSE terms (the one-launch path takes terms of either tier: a sparse list has no row and needs none)
static bool query_lists_probed_dense(const ss_shard* s, const ss_bm25_query& q) {
  const uint32_t all = q.n_terms + bm_q_nnot(q.op), L = s->bm_n_fields, n_dense = s->bm_n_terms / L;
  for (uint32_t t = 0; t < all && t < SS_MAX_QUERY_TERMS; t++) {
    if (q.term[t] >= n_dense) continue;
    const uint32_t v = q.term[t] * L + (L - 1u);  // (no field filter on this path: the only / merged list)
    if (s->h_probe_row[v] == BM_NO_PROBE_ROW && s->h_df[v] != 0) return false;
  }
  return true;
}

// A vocabulary larger than the probe budget has rows for its longest lists only (ss_bm25_set_probe_budget).  One query
// that touches a list without a row must not send its whole batch to the scan kernels: the batch is run as two -- the
// queries whose lists all have rows (pruned strategy), then the others -- and the answers are put back in the callers' order.
static int bm25_search_split_batch(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, uint32_t n_filters,
                                   const ss_facet_filter* filters, const std::vector<uint8_t>& probed, uint32_t n_probed) {
  // staging that outlives this call: the asynchronous copies below read it until the caller's stream synchronisation (every
  // caller synchronises s->stream before it returns, under s->mu)
  static thread_local std::vector<uint32_t> perm;
  static thread_local std::vector<ss_bm25_query> qs;
  perm.resize(nq);
  qs.resize(nq);
  uint32_t a = 0, b = n_probed;
  for (uint32_t i = 0; i < nq; i++) {
    const uint32_t at = probed[i] ? a++ : b++;
    perm[at] = i;
    qs[at] = q[i];
  }
  struct Part { bool has_and, has_or, all_probed, any_frequent, phrase, any_filter, uniform, gated; uint32_t nt_max, np_max, nn_max; } part[2];
  const uint32_t begin[2] = {0, n_probed}, count[2] = {n_probed, nq - n_probed};
  for (int h = 0; h < 2; h++)
    SS_TRY(check_queries(s, count[h], qs.data() + begin[h], &part[h].has_and, &part[h].has_or, &part[h].nt_max, &part[h].np_max,
                         &part[h].all_probed, &part[h].any_frequent, &part[h].phrase, &part[h].any_filter, &part[h].uniform, &part[h].gated, &part[h].nn_max));
  SS_HIP(hipSetDevice(s->device));
  const uint32_t kw = std::max<uint32_t>(kk, 1);
  SS_TRY(ensure_out(s, 2 * (size_t)nq, kw));  // upper half: the answers in the order they ran in
  const size_t qbytes = (size_t)nq * sizeof(ss_bm25_query) + (size_t)nq * sizeof(uint32_t);
  if (qbytes > s->bq_cap) {
    if (s->d_bq) (void)hipFree(s->d_bq);
    s->d_bq = nullptr; s->bq_cap = 0;
    SS_HIP(hipMalloc(&s->d_bq, qbytes));
    s->bq_cap = qbytes;
  }
  ss_bm25_query* d_q = (ss_bm25_query*)s->d_bq;
  uint32_t* d_perm = (uint32_t*)(d_q + nq);
  SS_HIP(hipMemcpyAsync(d_q, qs.data(), (size_t)nq * sizeof(ss_bm25_query), hipMemcpyHostToDevice, s->stream));
  SS_HIP(hipMemcpyAsync(d_perm, perm.data(), (size_t)nq * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
  uint32_t* t_doc = s->d_out_doc + (size_t)nq * kw;
  float* t_score = s->d_out_score + (size_t)nq * kw;
  uint32_t* t_count = s->d_out_count + nq;
  uint64_t* t_total = s->d_out_total + nq;
  SS_TRY(with_facet_filter(s, n_filters, filters, s->stream, [&]() {
    for (int h = 0; h < 2; h++) {
      const int rc = ssi_bm25_search(s, count[h], d_q + begin[h], kk, rt, t_doc + (size_t)begin[h] * kw, t_score + (size_t)begin[h] * kw,
                                     t_count + begin[h], t_total + begin[h], part[h].has_and, part[h].has_or, part[h].nt_max,
                                     part[h].np_max, part[h].all_probed, s->stream, part[h].any_frequent, part[h].phrase, part[h].any_filter, part[h].uniform, part[h].gated, part[h].nn_max);
      if (rc != SS_OK) return rc;
    }
    return (int)SS_OK;
  }));
  bm25_unpermute_kernel<<<nq, 64, 0, s->stream>>>(d_perm, nq, kk, t_doc, t_score, t_count, (const unsigned long long*)t_total, s->d_out_doc,
                                                   s->d_out_score, s->d_out_count, (unsigned long long*)s->d_out_total);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

static int bm25_search_host_queries(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, uint32_t n_filters,
                                    const ss_facet_filter* filters);

// A batch in which some query names a term of the SPARSE tier (bm25_sparse.hip).  Those queries are answered in two parts -- their
// dense terms through the ordinary path (together with the batch's all-dense queries: one sub-batch), their sparse lists by the
// sparse kernel, which scores every doc of a sparse list in full -- and put together per query by bm25_tier_merge_kernel; the
// answers land in s->d_out_* in the callers' order, like any other batch's.  Caller holds s->mu.
static int bm25_search_tiered_excl(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, const std::vector<uint32_t>& special);
static int bm25_search_compose(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, const std::vector<uint32_t>& composed);  // (defined below)
static int bm25_search_tiered(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt) {
  const uint32_t n_dense = s->bm_n_terms / s->bm_n_fields;  // public terms of the dense image
  if (s->bm_n_fields != 1 && !s->bm_merged) return SS_ENOTSUP;  // (several indexed fields: the sparse tier holds merged weights)
  std::vector<uint32_t> special;       // unions with a SPARSE NOT term: answered one by one under a per-query exclusion bitmap
  std::vector<uint32_t> composed;      // unions of several terms under a field filter: the reference's own sub-queries, one by one
  std::vector<ss_bm25_query> sub;      // the dense sub-batch: all-dense queries as they are, tiered unions reduced to their dense terms
  std::vector<ss_bm25_query> spq;      // the tiered queries, whole, for the sparse kernel
  std::vector<ss_bm25_query> spq_phrase;  // ... the phrases among them, for the sparse phrase kernel
  std::vector<uint32_t> dense_row(nq, 0xFFFFFFFFu), sparse_row(nq, 0xFFFFFFFFu);
  for (uint32_t i = 0; i < nq; i++) {
    const uint32_t op = bm_q_op(q[i].op), n_not = bm_q_nnot(q[i].op), all = q[i].n_terms + n_not;
    if (q[i].n_terms == 0 || all > SS_MAX_QUERY_TERMS) return SS_EINVAL;
    bool any_sparse = false, sparse_not = false, sparse_pos = false;
    for (uint32_t t = 0; t < all; t++) {
      if (q[i].term[t] >= n_dense + s->sp_n) return SS_EINVAL;
      if (q[i].term[t] >= n_dense) { any_sparse = true; sparse_not |= t >= q[i].n_terms; sparse_pos |= t < q[i].n_terms; }
      if (t < q[i].n_terms && !(q[i].idf[t] > 0.0f)) return SS_EINVAL;
      for (uint32_t u = 0; u < t; u++)
        if (q[i].term[u] == q[i].term[t]) return SS_EINVAL;
    }
    if (!any_sparse) {
      dense_row[i] = (uint32_t)sub.size();
      sub.push_back(q[i]);
      continue;
    }
    if (op != SS_OP_INTERSECTION && op != SS_OP_UNION && op != SS_OP_PHRASE) return SS_EINVAL;
    const bool is_phrase = op == SS_OP_PHRASE;
    // a field filter (several indexed fields): a phrase's is a test on its positions' tags; an intersection's / a single term's asks
    // every term for a listed field -- a sparse posting carries its fields; a UNION of several terms under a filter is the dense tier's
    // gated scan over (term, field) lists, which the tier does not keep: composed from filtered intersections (<= 5 terms)
    if (bm_q_field_filter(q[i].op) >> bm_real_fields(s)) return SS_EINVAL;
    const bool filtered = s->bm_n_fields > 1 && bm_q_field_filter(q[i].op) != 0u;
    if (bm_q_all_frequent(q[i].op)) return SS_ENOTSUP;
    if (filtered && op == SS_OP_UNION && q[i].n_terms > 1) {  // (bm25_route_shapes sends these to bm25_search_compose before they get here)
      if (q[i].n_terms > 10) return SS_ENOTSUP;
      composed.push_back(i);
      continue;
    }
    const bool is_and = (op == SS_OP_INTERSECTION && q[i].n_terms > 1) || is_phrase;
    // (a union's dense part cannot probe a sparse NOT list; an intersection is driven by a sparse list -- it needs one)
    const bool is_special = sparse_not && (!is_and || !sparse_pos);
    if (is_special) special.push_back(i);
    if (is_phrase && !is_special) {  // the sparse phrase kernel (bm25_sparse.hip): its shortest sparse list drives
      if (q[i].phrase_len < 2 || q[i].phrase_len > SS_MAX_PHRASE || q[i].phrase_seq[0] >= q[i].n_terms) return SS_EINVAL;
      for (uint32_t j = 1; j < q[i].phrase_len; j++)
        if (q[i].phrase_seq[j] >= q[i].n_terms && q[i].phrase_seq[j] != SS_PHRASE_SKIP) return SS_EINVAL;
      if (q[i].n_terms > 6) return SS_ENOTSUP;
      bool dense_pos = false;
      for (uint32_t t = 0; t < q[i].n_terms; t++) dense_pos |= q[i].term[t] < n_dense;
      // positions: the tier's own, and the image's for the phrase's dense words
      if (!s->d_sp_pos_end || (dense_pos && (s->bm_n_fields > 1 ? !s->d_pos32 : !s->d_pos))) return SS_ESTATE;
      sparse_row[i] = 0x80000000u | (uint32_t)spq_phrase.size();  // row behind the plain queries' (fixed up below)
      spq_phrase.push_back(q[i]);
      continue;
    }
    sparse_row[i] = (uint32_t)spq.size();
    spq.push_back(q[i]);
    if (!is_and) {  // the union's dense terms (with its NOT terms) as a query of their own
      ss_bm25_query d = q[i];
      uint32_t n = 0;
      for (uint32_t t = 0; t < q[i].n_terms; t++)
        if (q[i].term[t] < n_dense) { d.term[n] = q[i].term[t]; d.idf[n] = q[i].idf[t]; n++; }
      if (n) {
        for (uint32_t t = 0; t < n_not; t++) d.term[n + t] = q[i].term[q[i].n_terms + t];
        for (uint32_t t = n + n_not; t < (uint32_t)SS_MAX_QUERY_TERMS; t++) { d.term[t] = 0; if (t >= n) d.idf[t] = 0.f; }
        d.n_terms = n;
        dense_row[i] = (uint32_t)sub.size();
        sub.push_back(d);
      }
    }
  }
  SS_HIP(hipSetDevice(s->device));
  if (!composed.empty()) return bm25_search_compose(s, nq, q, kk, rt, composed);
  if (!special.empty()) return bm25_search_tiered_excl(s, nq, q, kk, rt, special);
  const uint32_t ns_plain = (uint32_t)spq.size();
  for (uint32_t i = 0; i < nq; i++)
    if (sparse_row[i] != 0xFFFFFFFFu && (sparse_row[i] & 0x80000000u)) sparse_row[i] = ns_plain + (sparse_row[i] & 0x7FFFFFFFu);
  spq.insert(spq.end(), spq_phrase.begin(), spq_phrase.end());
  const uint32_t kw = std::max<uint32_t>(kk, 1), ns = (uint32_t)spq.size(), nd = (uint32_t)sub.size();
  const int KPL = ssi_bm25_sparse_kpl(kw);
  SS_TRY(ensure_out(s, std::max<size_t>(nq, nd), kw));  // reserved before the sub-batch runs: its own ensure_out then keeps the buffers
  // workspace: [sparse queries][row maps 2 nq][sparse keys][sparse counts][merged doc | score | count | total][seeds of the dense sub-batch]
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_q = 0, o_dr = o_q + al((size_t)ns * sizeof(ss_bm25_query)), o_sr = o_dr + al((size_t)nq * 4), o_keys = o_sr + al((size_t)nq * 4),
               o_ext = o_keys + al((size_t)ns * 64 * KPL * 8), o_doc = o_ext + al((size_t)ns * 8), o_sc = o_doc + al((size_t)nq * kw * 4),
               o_cnt = o_sc + al((size_t)nq * kw * 4), o_tot = o_cnt + al((size_t)nq * 4), o_seed = o_tot + al((size_t)nq * 8), need = o_seed + al((size_t)nd * 4 + 4);
  if (need > s->tier_ws_cap) {
    SS_HIP(hipStreamSynchronize(s->stream));
    if (s->d_tier_ws) (void)hipFree(s->d_tier_ws);
    s->d_tier_ws = nullptr; s->tier_ws_cap = 0;
    SS_HIP(hipMalloc(&s->d_tier_ws, need * 2));
    s->tier_ws_cap = need * 2;
  }
  char* W = (char*)s->d_tier_ws;
  // the host vectors are read by synchronous copies (they may die with this frame) -- which do not wait for the stream: a tiered
  // search still queued (the device-pointer entry point returns early, bm25_search_tiered_excl runs several) reads this workspace.
  // Waited for HERE, while nothing of this call is queued yet (the stream is idle then, as a rule).
  SS_HIP(hipStreamSynchronize(s->stream));
  SS_HIP(hipMemcpy(W + o_q, spq.data(), (size_t)ns * sizeof(ss_bm25_query), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(W + o_dr, dense_row.data(), (size_t)nq * 4, hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(W + o_sr, sparse_row.data(), (size_t)nq * 4, hipMemcpyHostToDevice));
  // The SPARSE kernel first (round 6; it used to run behind the dense sub-batch): its lists are short, its docs carry FULL scores -- a
  // union that finds k docs among its rare terms hands their k-th score to its dense terms as a threshold seed (sp_seed_kernel), and the
  // dense kernels stop reading frequent words' lists that cannot reach it.  (Small calls take the one-launch path, where the same idea
  // is role 3 publishing into the query's shared threshold; bm25_small.hip.)
  SS_TRY(ssi_bm25_launch_sparse(s, (const ss_bm25_query*)(W + o_q), ns_plain, kk, (unsigned long long*)(W + o_keys), (unsigned long long*)(W + o_ext), s->stream));
  SS_TRY(ssi_bm25_launch_sparse_phrase(s, (const ss_bm25_query*)(W + o_q) + ns_plain, ns - ns_plain, kk,
                                       (unsigned long long*)(W + o_keys) + (size_t)ns_plain * 64 * KPL, (unsigned long long*)(W + o_ext) + ns_plain, s->stream));
  if (nd) {
    if (kk) {
      SS_HIP(hipMemsetAsync(W + o_seed, 0, (size_t)nd * 4, s->stream));
      SS_TRY(ssi_bm25_launch_sparse_seeds(nq, kk, (const uint32_t*)(W + o_dr), (const uint32_t*)(W + o_sr), (const ss_bm25_query*)(W + o_q),
                                          (const unsigned long long*)(W + o_keys), (float*)(W + o_seed), s->stream));
      s->d_ext_seed = (const float*)(W + o_seed);
      s->ext_seed_n = nd;
    }
    const int rc_sub = bm25_search_host_queries(s, nd, sub.data(), kk, rt, 0, nullptr);  // -> s->d_out_* rows [0, nd)
    s->d_ext_seed = nullptr; s->ext_seed_n = 0;
    if (rc_sub != SS_OK) return rc_sub;
  }
  SS_TRY(ssi_bm25_launch_tier_merge(nq, kk, (const uint32_t*)(W + o_dr), (const uint32_t*)(W + o_sr), s->d_out_doc, s->d_out_score, s->d_out_count,
                                    (const unsigned long long*)s->d_out_total, (const unsigned long long*)(W + o_keys),
                                    (const unsigned long long*)(W + o_ext), (uint32_t*)(W + o_doc), (float*)(W + o_sc), (uint32_t*)(W + o_cnt),
                                    (unsigned long long*)(W + o_tot), s->stream));
  if (kk) {
    SS_HIP(hipMemcpyAsync(s->d_out_doc, W + o_doc, (size_t)nq * kw * 4, hipMemcpyDeviceToDevice, s->stream));
    SS_HIP(hipMemcpyAsync(s->d_out_score, W + o_sc, (size_t)nq * kw * 4, hipMemcpyDeviceToDevice, s->stream));
  }
  SS_HIP(hipMemcpyAsync(s->d_out_count, W + o_cnt, (size_t)nq * 4, hipMemcpyDeviceToDevice, s->stream));
  SS_HIP(hipMemcpyAsync(s->d_out_total, W + o_tot, (size_t)nq * 8, hipMemcpyDeviceToDevice, s->stream));
  return SS_OK;
}

// A UNION that excludes a SPARSE term, or an intersection of dense terms that does (add_result.rs:3440-3497 applies NOT lists to every
// query type): the dense kernels probe NOT lists through directory rows a sparse list does not have.  Such a query -- rare: a rare word after a minus sign -- is answered on
// its own with the docs of its sparse NOT lists added to the exclusion bitmap the kernels already honour (tombstones, or a facet
// filter's bitmap): top-k and exact counts then follow from the paths that serve a shard with deleted docs.  The other queries of
// the batch run together as always; the single answers are put back into their rows.
static int bm25_search_tiered_excl(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, const std::vector<uint32_t>& special) {
  const uint32_t kw = std::max<uint32_t>(kk, 1), n_sp = (uint32_t)special.size(), n_dense = s->bm_n_terms / s->bm_n_fields;
  const size_t words = ((size_t)s->bm_n_docs + 31) / 32;
  SS_TRY(ensure_out(s, nq, kw));
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t h_doc = 0, h_sc = h_doc + al((size_t)n_sp * kw * 4), h_cnt = h_sc + al((size_t)n_sp * kw * 4), h_tot = h_cnt + al((size_t)n_sp * 4),
               h_need = h_tot + al((size_t)n_sp * 8);
  if (h_need > s->tier_hold_cap || words > s->excl_words_cap) SS_HIP(hipStreamSynchronize(s->stream));
  if (h_need > s->tier_hold_cap) {
    if (s->d_tier_hold) (void)hipFree(s->d_tier_hold);
    s->d_tier_hold = nullptr; s->tier_hold_cap = 0;
    SS_HIP(hipMalloc(&s->d_tier_hold, h_need * 2));
    s->tier_hold_cap = h_need * 2;
  }
  if (words > s->excl_words_cap) {
    if (s->d_excl_bits) (void)hipFree(s->d_excl_bits);
    s->d_excl_bits = nullptr; s->excl_words_cap = 0;
    SS_HIP(hipMalloc(&s->d_excl_bits, words * 4));
    s->excl_words_cap = words;
  }
  char* H = (char*)s->d_tier_hold;
  std::vector<ss_bm25_query> rest(q, q + nq);
  for (uint32_t j = 0; j < n_sp; j++) {
    const ss_bm25_query& Q = q[special[j]];
    const uint32_t np = Q.n_terms, n_not = bm_q_nnot(Q.op);
    ss_bm25_query R = Q;  // the query without its sparse NOT terms
    uint32_t lists[SS_MAX_QUERY_TERMS], n_lists = 0, kept = 0;
    for (uint32_t t = 0; t < n_not; t++) {
      const uint32_t term = Q.term[np + t];
      if (term >= n_dense) lists[n_lists++] = term - n_dense;
      else R.term[np + kept++] = term;
    }
    for (uint32_t t = np + kept; t < (uint32_t)SS_MAX_QUERY_TERMS; t++) R.term[t] = 0;
    R.op = (Q.op & ~0xFF00u) | SS_OP_NOT_TERMS(kept);
    rest[special[j]] = R;  // keeps the row's place in the batch below; its answer is overwritten
    const bool had = s->n_deleted != 0;
    SS_TRY(ssi_bm25_sparse_excl_bits(s, had ? s->d_deleted : nullptr, had ? (uint32_t)s->deleted_words : 0u, lists, n_lists, s->d_excl_bits, (uint32_t)words, s->stream));
    uint32_t* del = s->d_deleted;
    const uint64_t dw = s->deleted_words, nd = s->n_deleted;
    s->d_deleted = s->d_excl_bits; s->deleted_words = words; s->n_deleted = 1;
    const int rc = bm25_search_host_queries(s, 1, &R, kk, rt, 0, nullptr);
    s->d_deleted = del; s->deleted_words = dw; s->n_deleted = nd;
    if (rc != SS_OK) return rc;
    if (nq == 1) return SS_OK;  // the answer stands in row 0 already
    if (kk) {
      SS_HIP(hipMemcpyAsync(H + h_doc + (size_t)j * kw * 4, s->d_out_doc, (size_t)kw * 4, hipMemcpyDeviceToDevice, s->stream));
      SS_HIP(hipMemcpyAsync(H + h_sc + (size_t)j * kw * 4, s->d_out_score, (size_t)kw * 4, hipMemcpyDeviceToDevice, s->stream));
    }
    SS_HIP(hipMemcpyAsync(H + h_cnt + (size_t)j * 4, s->d_out_count, 4, hipMemcpyDeviceToDevice, s->stream));
    SS_HIP(hipMemcpyAsync(H + h_tot + (size_t)j * 8, s->d_out_total, 8, hipMemcpyDeviceToDevice, s->stream));
  }
  SS_TRY(bm25_search_host_queries(s, nq, rest.data(), kk, rt, 0, nullptr));
  for (uint32_t j = 0; j < n_sp; j++) {
    const size_t i = special[j];
    if (kk) {
      SS_HIP(hipMemcpyAsync(s->d_out_doc + i * kw, H + h_doc + (size_t)j * kw * 4, (size_t)kw * 4, hipMemcpyDeviceToDevice, s->stream));
      SS_HIP(hipMemcpyAsync(s->d_out_score + i * kw, H + h_sc + (size_t)j * kw * 4, (size_t)kw * 4, hipMemcpyDeviceToDevice, s->stream));
    }
    SS_HIP(hipMemcpyAsync(s->d_out_count + i, H + h_cnt + (size_t)j * 4, 4, hipMemcpyDeviceToDevice, s->stream));
    SS_HIP(hipMemcpyAsync(s->d_out_total + i, H + h_tot + (size_t)j * 8, 8, hipMemcpyDeviceToDevice, s->stream));
  }
  return SS_OK;
}

// A UNION of several terms under a field filter that the gated scan does not take: one that names a term of the sparse tier (the dense
// tier gates every term's unlisted (term, field) lists inside the scan, BM_AND_GATED; the sparse tier keeps one merged list per term),
// or one of 8 .. 10 terms (the match byte holds 7 term bits).  The
// query is answered the way the reference itself answers it (union.rs:1168-1305, 1330-1425: union_docid_3 queues the intersection
// of all terms and every subset down to pairs, union_docid_2 runs a pair as its intersection plus the two single terms; the filter
// applies to the terms of the sub-query that finds the doc, add_result.rs:3124-3136; a doc found again keeps its better score,
// min_heap.rs:1193-1260): the 2^n - 1 filtered intersections as ONE batch through both tiers, merged per doc by the maximum -- a doc
// of the union's top-k is in the top-k of the sub-query that gives it its score.  Totals as the reference reports them: two terms
// |pass(X) u pass(Y)| (union_docid_2's count), more the UNFILTERED union (union_scan counts a doc before the filter sees it,
// union.rs:552-553).  Rare (a rare word in a multi-word query under a field filter, or 8+ words under one); <= 10 terms -- the range
// of union_docid_3 (search.rs:3497-3520) -- i.e. <= 1023 sub-queries in one batch (the sub-queries the specialised kernels do not take
// run on bm25_gallop.hip); the merge is the host's.
// A UNION of MORE than 10 terms under a field filter.  The reference leaves union_docid_3's range here (search.rs:3497-3520) and runs
// union_blockid -> union_scan_32 (union.rs:598-805), whose candidates meet the filter inside add_result_multiterm_multifield
// (add_result.rs:3124-3136) -- ANOTHER rule than the sub-queries' above: the loop over the doc's PRESENT terms returns at the first one
// that stands in no listed field, so a doc is an answer iff EVERY term it holds passes the filter, and then scores with all of them (all
// fields of each, as ever); union_scan has counted it before the filter saw it (union.rs:760-761), so result_count_total is the
// UNFILTERED union's.  With  R = U_t { docs of t that hold t in unlisted fields only }  that is: the plain union of the merged lists under
// the exclusion bitmap  tombstones | R.  R comes from the probe index's bit records: the match sets M_t (term t, no filter) and P_t (term
// t under the filter) of the 2 n single-term queries in ONE ssi_bm25_match_bits call, R |= M_t & ~P_t.  Then the ordinary many-list
// union runs under that bitmap (every kernel family honours it), its count under the tombstones alone.  Dense terms with probe rows.
__global__ void gate_rule_bits_kernel(const unsigned long long* __restrict__ sets /*[2 n][groups]: M_0, P_0, M_1, P_1, ...*/, uint32_t n, size_t groups,
                                      const uint32_t* __restrict__ base, uint32_t base_words, uint32_t* __restrict__ out, uint32_t words) {
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g * 2 < words; g += (size_t)gridDim.x * blockDim.x) {
    unsigned long long r = 0ull;
    if (g < groups)
      for (uint32_t t = 0; t < n; t++) r |= sets[(size_t)(2 * t) * groups + g] & ~sets[(size_t)(2 * t + 1) * groups + g];
    const size_t w0 = g * 2, w1 = g * 2 + 1;
    out[w0] = (uint32_t)r | ((base && w0 < base_words) ? base[w0] : 0u);
    if (w1 < words) out[w1] = (uint32_t)(r >> 32) | ((base && w1 < base_words) ? base[w1] : 0u);
  }
}
static int bm25_search_gated_scan_rule(ss_shard* s, const ss_bm25_query& Q, uint32_t kk, uint32_t rt, uint32_t* a_doc, float* a_score, uint32_t* a_cnt,
                                       unsigned long long* a_tot) {
  const uint32_t n = Q.n_terms, nn = bm_q_nnot(Q.op), n_dense = s->bm_n_terms / s->bm_n_fields, kw = std::max<uint32_t>(kk, 1);
  if (!s->bm_merged || !s->d_probe || 2 * n > 64) return SS_ENOTSUP;
  SS_HIP(hipSetDevice(s->device));
  // dense terms: M_t and P_t from the bit records; sparse-tier terms: their postings carry their fields (marked below)
  std::vector<ss_bm25_query> subs;
  uint32_t sp_lists[SS_MAX_QUERY_TERMS], n_sp = 0;
  for (uint32_t t = 0; t < n; t++) {
    if (Q.term[t] >= n_dense) { sp_lists[n_sp++] = Q.term[t] - n_dense; continue; }
    for (int f = 0; f < 2; f++) {
      ss_bm25_query S;
      memset(&S, 0, sizeof(S));
      S.n_terms = 1; S.term[0] = Q.term[t]; S.idf[0] = Q.idf[t];
      S.op = SS_OP_INTERSECTION | (f ? SS_OP_FIELD_FILTER(bm_q_field_filter(Q.op)) : 0u);
      subs.push_back(S);
    }
  }
  const uint32_t nd2 = (uint32_t)subs.size();  // 2 x the dense terms
  if (nd2) {
    bool has_and, has_or, all_probed, any_frequent, phrase = false, any_filter = false, uniform = false, gated = false;
    uint32_t nt_max, np_max, nn_max = 0;
    SS_TRY(ssi_bm25_ensure_probe_rows(s, nd2, subs.data(), s->stream));
    SS_TRY(check_queries(s, nd2, subs.data(), &has_and, &has_or, &nt_max, &np_max, &all_probed, &any_frequent, &phrase, &any_filter, &uniform, &gated, &nn_max));
    if (!all_probed) return SS_ENOTSUP;  // (a rationed vocabulary: the match sets are read from the bit records)
  }
  const size_t groups = (size_t)s->bm_n_sub * (BM_SUB / 64), words = ((size_t)s->bm_n_docs + 31) / 32;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_q = 0, o_tot = o_q + al(64 * sizeof(ss_bm25_query)), o_sets = o_tot + al(64 * 8), need = o_sets + al(2 * (size_t)n * groups * 8);
  if (need > s->gate_ws_cap || words > s->excl_words_cap) SS_HIP(hipStreamSynchronize(s->stream));
  if (need > s->gate_ws_cap) {
    if (s->d_gate_ws) (void)hipFree(s->d_gate_ws);
    s->d_gate_ws = nullptr; s->gate_ws_cap = 0;
    SS_HIP(hipMalloc(&s->d_gate_ws, need));
    s->gate_ws_cap = need;
  }
  if (words > s->excl_words_cap) {
    if (s->d_excl_bits) (void)hipFree(s->d_excl_bits);
    s->d_excl_bits = nullptr; s->excl_words_cap = 0;
    SS_HIP(hipMalloc(&s->d_excl_bits, words * 4));
    s->excl_words_cap = words;
  }
  char* W = (char*)s->d_gate_ws;
  if (nd2) {
    SS_HIP(hipMemcpy(W + o_q, subs.data(), subs.size() * sizeof(ss_bm25_query), hipMemcpyHostToDevice));  // (synchronous: `subs` is a local)
    SS_TRY(ssi_bm25_match_bits(s, (const ss_bm25_query*)(W + o_q), (unsigned long long*)(W + o_sets), (unsigned long long*)(W + o_tot), s->stream, nd2));
  }
  const bool had = s->n_deleted != 0;
  gate_rule_bits_kernel<<<(uint32_t)std::min<size_t>(2048, (words / 2 + 256) / 256), 256, 0, s->stream>>>(
      (const unsigned long long*)(W + o_sets), nd2 / 2, groups, had ? s->d_deleted : nullptr, had ? (uint32_t)s->deleted_words : 0u, s->d_excl_bits, (uint32_t)words);
  SS_HIP(hipGetLastError());
  SS_TRY(ssi_bm25_sparse_mark_unlisted(s, sp_lists, n_sp, bm_q_field_filter(Q.op), s->d_excl_bits, (uint32_t)words, s->stream));
  ss_bm25_query U = Q;  // the plain union: the filter is in the bitmap now
  U.op = SS_OP_UNION | SS_OP_NOT_TERMS(nn);
  *a_cnt = 0; *a_tot = 0;
  if (kk && rt != SS_RT_COUNT) {
    {
      uint32_t* del = s->d_deleted;
      const uint64_t dw = s->deleted_words, nd = s->n_deleted;
      s->d_deleted = s->d_excl_bits; s->deleted_words = words; s->n_deleted = 1;
      const int rc = bm25_search_host_queries(s, 1, &U, kk, SS_RT_TOPK, 0, nullptr);
      s->d_deleted = del; s->deleted_words = dw; s->n_deleted = nd;
      if (rc != SS_OK) return rc;
    }
    SS_HIP(hipStreamSynchronize(s->stream));
    SS_HIP(hipMemcpy(a_doc, s->d_out_doc, (size_t)kw * 4, hipMemcpyDeviceToHost));
    SS_HIP(hipMemcpy(a_score, s->d_out_score, (size_t)kw * 4, hipMemcpyDeviceToHost));
    SS_HIP(hipMemcpy(a_cnt, s->d_out_count, 4, hipMemcpyDeviceToHost));
    *a_tot = *a_cnt;
  }
  if (rt != SS_RT_TOPK) {
    SS_TRY(bm25_search_host_queries(s, 1, &U, 0, SS_RT_COUNT, 0, nullptr));
    SS_HIP(hipStreamSynchronize(s->stream));
    SS_HIP(hipMemcpy(a_tot, s->d_out_total, 8, hipMemcpyDeviceToHost));
  }
  return SS_OK;
}

static int bm25_search_compose(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, const std::vector<uint32_t>& composed) {
  const uint32_t kw = std::max<uint32_t>(kk, 1), n_co = (uint32_t)composed.size();
  std::vector<uint32_t> a_doc((size_t)n_co * kw, SS_NO_DOC), a_cnt(n_co, 0);
  std::vector<float> a_score((size_t)n_co * kw, 0.f);
  std::vector<unsigned long long> a_tot(n_co, 0);
  std::vector<ss_bm25_query> rest(q, q + nq);
  const uint32_t rt_sub = rt == SS_RT_TOPK ? SS_RT_TOPK : (rt == SS_RT_COUNT ? SS_RT_COUNT : SS_RT_TOPKCOUNT);
  for (uint32_t j = 0; j < n_co; j++) {
    const ss_bm25_query& Q = q[composed[j]];
    const uint32_t n = Q.n_terms, nn = bm_q_nnot(Q.op);
    const uint32_t op_sub = SS_OP_INTERSECTION | SS_OP_NOT_TERMS(nn) | SS_OP_FIELD_FILTER(bm_q_field_filter(Q.op));
    if (n > 10) {  // the reference's other rule (union_scan + the per-doc filter): above
      SS_TRY(bm25_search_gated_scan_rule(s, Q, kk, rt, a_doc.data() + (size_t)j * kw, a_score.data() + (size_t)j * kw, &a_cnt[j], &a_tot[j]));
      rest[composed[j]] = Q;  // keeps the row's place in the batch below; its answer is overwritten
      rest[composed[j]].n_terms = 1;
      for (uint32_t t = 1; t < (uint32_t)SS_MAX_QUERY_TERMS; t++) { rest[composed[j]].term[t] = 0; rest[composed[j]].idf[t] = 0.f; }
      rest[composed[j]].op = SS_OP_INTERSECTION | SS_OP_FIELD_FILTER(bm_q_field_filter(Q.op));
      continue;
    }
    std::vector<ss_bm25_query> subs;
    for (uint32_t m = 1; m < (1u << n); m++) {
      ss_bm25_query S;
      memset(&S, 0, sizeof(S));
      for (uint32_t t = 0; t < n; t++)
        if ((m >> t) & 1u) { S.term[S.n_terms] = Q.term[t]; S.idf[S.n_terms] = Q.idf[t]; S.n_terms++; }
      if (S.n_terms + nn > (uint32_t)SS_MAX_QUERY_TERMS) return SS_EINVAL;
      for (uint32_t t = 0; t < nn; t++) S.term[S.n_terms + t] = Q.term[n + t];
      S.op = op_sub;
      subs.push_back(S);
    }
    rest[composed[j]] = subs[0];  // keeps the row's place in the batch below; its answer is overwritten
    rest[composed[j]].op = SS_OP_INTERSECTION | SS_OP_FIELD_FILTER(bm_q_field_filter(Q.op));
    const uint32_t ns = (uint32_t)subs.size();
    std::vector<unsigned long long> tot(ns, 0);
    std::vector<std::pair<float, uint32_t>> ranked;
    const bool need_subs = rt != SS_RT_COUNT || n == 2;
    if (need_subs) {
      SS_TRY(bm25_search_host_queries(s, ns, subs.data(), kk, rt_sub, 0, nullptr));
      std::vector<uint32_t> doc((size_t)ns * kw), cnt(ns);
      std::vector<float> score((size_t)ns * kw);
      SS_HIP(hipStreamSynchronize(s->stream));
      if (kk && rt != SS_RT_COUNT) {
        SS_HIP(hipMemcpy(doc.data(), s->d_out_doc, (size_t)ns * kw * 4, hipMemcpyDeviceToHost));
        SS_HIP(hipMemcpy(score.data(), s->d_out_score, (size_t)ns * kw * 4, hipMemcpyDeviceToHost));
        SS_HIP(hipMemcpy(cnt.data(), s->d_out_count, (size_t)ns * 4, hipMemcpyDeviceToHost));
        std::unordered_map<uint32_t, float> best;
        for (uint32_t i = 0; i < ns; i++)
          for (uint32_t r = 0; r < std::min(cnt[i], kk); r++) {
            auto it = best.find(doc[(size_t)i * kw + r]);
            if (it == best.end()) best.emplace(doc[(size_t)i * kw + r], score[(size_t)i * kw + r]);
            else if (score[(size_t)i * kw + r] > it->second) it->second = score[(size_t)i * kw + r];
          }
        ranked.reserve(best.size());
        for (const auto& e : best) ranked.emplace_back(e.second, e.first);
        std::sort(ranked.begin(), ranked.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) {
          return a.first != b.first ? a.first > b.first : a.second < b.second;
        });
        if (ranked.size() > kk) ranked.resize(kk);
      }
      if (rt != SS_RT_TOPK) SS_HIP(hipMemcpy(tot.data(), s->d_out_total, (size_t)ns * 8, hipMemcpyDeviceToHost));
    }
    for (size_t r = 0; r < ranked.size(); r++) { a_doc[(size_t)j * kw + r] = ranked[r].second; a_score[(size_t)j * kw + r] = ranked[r].first; }
    a_cnt[j] = (uint32_t)ranked.size();
    if (rt == SS_RT_TOPK) a_tot[j] = ranked.size();
    else if (n == 2) a_tot[j] = tot[0] + tot[1] - tot[2];  // subsets in mask order: {X}, {Y}, {X, Y}
    else {
      ss_bm25_query U = Q;
      U.op = SS_OP_UNION | SS_OP_NOT_TERMS(nn);
      SS_TRY(bm25_search_host_queries(s, 1, &U, 0, SS_RT_COUNT, 0, nullptr));
      SS_HIP(hipStreamSynchronize(s->stream));
      SS_HIP(hipMemcpy(&a_tot[j], s->d_out_total, 8, hipMemcpyDeviceToHost));
    }
  }
  if (n_co < nq) SS_TRY(bm25_search_host_queries(s, nq, rest.data(), kk, rt, 0, nullptr));
  else SS_TRY(ensure_out(s, nq, kw));
  SS_HIP(hipStreamSynchronize(s->stream));
  for (uint32_t j = 0; j < n_co; j++) {
    const size_t i = composed[j];
    if (kk && rt != SS_RT_COUNT) {
      SS_HIP(hipMemcpy(s->d_out_doc + i * kw, a_doc.data() + (size_t)j * kw, (size_t)kw * 4, hipMemcpyHostToDevice));
      SS_HIP(hipMemcpy(s->d_out_score + i * kw, a_score.data() + (size_t)j * kw, (size_t)kw * 4, hipMemcpyHostToDevice));
    }
    SS_HIP(hipMemcpy(s->d_out_count + i, &a_cnt[j], 4, hipMemcpyHostToDevice));
    SS_HIP(hipMemcpy(s->d_out_total + i, &a_tot[j], 8, hipMemcpyHostToDevice));
  }
  return SS_OK;
}

// ---------------------------------------------------------------- query shapes (VERDICT r5 "next" 1)
// Every host-pointer batch passes here first.  The specialised kernel families each serve a range of shapes (check_queries,
// bm25_search_tiered); what lies outside used to come back SS_ENOTSUP -- to a drop-in caller an EMPTY result for a query the reference
// answers.  Now a query is classified and the batch run as sub-batches per kernel family, answers back in the callers' order:
//   NATIVE         what the specialised paths serve (unchanged)
//   GALLOP         intersections / filtered single terms beyond them: all_terms_frequent with > 7 terms, > 8 terms or > BM_MAX_VTERMS
//                  (term, field) lists over per-field lists (a field filter, or an image without merged lists)  -> bm25_gallop.hip
//   GALLOP_PHRASE  phrases of 7 .. SS_MAX_PHRASE unique terms, or k > 128, either tier                                 -> bm25_gallop.hip
//   COMPOSE        unions of 2 .. 10 terms under a field filter that the gated scan does not take (8 .. 10 terms, or a term of the
//                  sparse tier): the reference's own sub-queries (union.rs:1330-1425), bm25_search_compose
// SS_ENOTSUP remains for the shapes INTEGRATION.md section 4 lists as CPU fall-through (tests/test_gpu_shape_sweep.py pins the list).
enum : uint8_t { SH_NATIVE = 0, SH_GALLOP = 1, SH_GALLOP_PHRASE = 2, SH_COMPOSE = 3 };

// postings of the list(s) a query walks for `term` when it drives (dense: its only / merged list, or all its field lists)
static uint64_t shape_term_postings(const ss_shard* s, uint32_t term) {
  const uint32_t L = s->bm_n_fields, n_dense = s->bm_n_terms / L;
  if (term >= n_dense) { const uint32_t i = term - n_dense; return i + 1 < s->h_sp_base.size() ? s->h_sp_base[i + 1] - s->h_sp_base[i] : 0; }
  if (L == 1 || s->bm_merged) return s->h_df[(size_t)term * L + (L - 1u)];
  uint64_t n = 0;
  for (uint32_t f = 0; f < L; f++) n += s->h_df[(size_t)term * L + f];
  return n;
}
// the reference's own condition for all_terms_frequent (intersection.rs:198-209: posting_count / indexed_doc_count >= 0.5 in f32 for
// EVERY term), which is also the builders' rule for the lists whose codes carry the tf < 10 mark (synth.hip bm_list_flagged): a bit
// set on a query the rule does not hold for is dropped -- the reference would not have set it, and the mark is not there to read
static bool shape_freq_rule(const ss_shard* s, const ss_bm25_query& Q) {
  const uint32_t L = s->bm_n_fields, n_dense = s->bm_n_terms / L;
  for (uint32_t t = 0; t < Q.n_terms; t++) {
    if (Q.term[t] >= n_dense) return false;  // (a sparse list's codes carry no mark; ss_index_bin_tier keeps lists of half the docs dense)
    const uint64_t df = (L > 1 && !s->bm_merged) ? (Q.term[t] < s->h_df_real.size() ? s->h_df_real[Q.term[t]] : 0) : s->h_df[(size_t)Q.term[t] * L + (L - 1u)];
    if (!((float)df / (float)s->bm_n_docs >= 0.5f)) return false;
  }
  return true;
}

static int bm25_shape_of(const ss_shard* s, const ss_bm25_query& Q, uint32_t kk, uint8_t* shape, bool* drop_freq, uint64_t* driver_len) {
  const uint32_t L = s->bm_n_fields, RF = bm_real_fields(s), n_dense = s->bm_n_terms / L;
  const uint32_t op = bm_q_op(Q.op), n_not = bm_q_nnot(Q.op), np = Q.n_terms, all = np + n_not;
  *shape = SH_NATIVE; *drop_freq = false; *driver_len = 0;
  if (np == 0 || all > (uint32_t)SS_MAX_QUERY_TERMS) return SS_EINVAL;
  if (op != SS_OP_INTERSECTION && op != SS_OP_UNION && op != SS_OP_PHRASE) return SS_EINVAL;
  if (bm_q_field_filter(Q.op) >> RF) return SS_EINVAL;  // a field the image does not have
  const uint32_t filt = RF > 1 ? bm_q_field_filter(Q.op) : 0u;
  bool any_sparse = false;
  uint64_t best = ~0ull;
  for (uint32_t t = 0; t < all; t++) {
    if (Q.term[t] >= n_dense + s->sp_n) return SS_EINVAL;
    if (t < np && !(Q.idf[t] > 0.0f)) return SS_EINVAL;
    for (uint32_t u = 0; u < t; u++)
      if (Q.term[u] == Q.term[t]) return SS_EINVAL;  // unique terms only (search.rs:3023 unique_terms)
    any_sparse |= Q.term[t] >= n_dense;
    if (t < np) best = std::min(best, shape_term_postings(s, Q.term[t]));
  }
  *driver_len = best;
  const bool is_and = op == SS_OP_INTERSECTION && np > 1;
  if (bm_q_all_frequent(Q.op) && is_and && !filt) {  // (anywhere else the bit has no effect: single terms, unions, phrases, under a field filter)
    if (!shape_freq_rule(s, Q)) *drop_freq = true;
    else if (L > 1 && !s->bm_merged) return SS_ENOTSUP;  // the several-fields form of the rule is coded into the merged lists
    else if (np > 7) *shape = SH_GALLOP;
  }
  if (op == SS_OP_PHRASE) {
    if (Q.phrase_len < 2 || Q.phrase_len > SS_MAX_PHRASE || Q.phrase_seq[0] >= np) return SS_EINVAL;
    for (uint32_t j = 1; j < Q.phrase_len; j++)
      if (Q.phrase_seq[j] >= np && Q.phrase_seq[j] != SS_PHRASE_SKIP) return SS_EINVAL;
    if (L > 1 && !s->bm_merged) return SS_ENOTSUP;  // phrases of several indexed fields run over the merged lists' field-tagged positions
    if (np > 6 || kk > 128) *shape = SH_GALLOP_PHRASE;
    // a rationed vocabulary: the phrase kernel of the probe index needs a row for EVERY list it reads and has no scan to fall back on -- a
    // phrase over a list without one takes the generic kernel, which needs none (round 5 built pool rows for it first and refused the
    // batch whose phrases alone exceeded the pool); the pool stays with the queries that gain from it
    if (!any_sparse && s->bm_probe_rows != 0 && s->bm_probe_rows < s->bm_n_terms)
      for (uint32_t t = 0; t < all; t++) {
        const uint32_t v = Q.term[t] * L + (L - 1u);
        if (s->h_probe_row[v] == BM_NO_PROBE_ROW && s->h_df[v] != 0) *shape = SH_GALLOP_PHRASE;
      }
    return SS_OK;
  }
  if (filt && op == SS_OP_UNION && np > 1) {  // a union under a field filter: the gated scan (<= 7 dense terms), else composed
    if (np > 10 && !s->bm_merged) return SS_ENOTSUP;  // (the scan rule runs the plain union over the merged lists)
    if (any_sparse || np > 7) *shape = SH_COMPOSE;  // 8 .. 10: the reference's sub-queries; more: its other rule (bm25_search_gated_scan_rule)
    return SS_OK;
  }
  // per-field lists: under a field filter, or on an image without merged lists
  if (RF > 1 && (filt || !s->bm_merged) && !any_sparse && (all * RF > (uint32_t)BM_MAX_VTERMS || (is_and && np > 8))) {
    if (is_and || np == 1 || filt) *shape = SH_GALLOP;
    else return SS_ENOTSUP;  // a union of more than BM_MAX_VTERMS (term, field) lists on an image without merged lists
  }
  return SS_OK;
}

static int bm25_search_host_queries(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, uint32_t n_filters,
                                    const ss_facet_filter* filters);
static int bm25_search_compose(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, const std::vector<uint32_t>& composed);

// *handled = false: every query is NATIVE (`*use` = the batch to run: q itself, or `norm` -- the copy with the dropped bits)
static int bm25_route_shapes(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, std::vector<ss_bm25_query>& norm,
                             const ss_bm25_query** use, bool* handled) {
  *handled = false;
  *use = q;
  std::vector<uint8_t> shape(nq);
  std::vector<uint64_t> dlen(nq);
  uint32_t n_class[4] = {0, 0, 0, 0};
  for (uint32_t i = 0; i < nq; i++) {
    bool drop = false;
    SS_TRY(bm25_shape_of(s, q[i], kk, &shape[i], &drop, &dlen[i]));
    if (drop) {
      if (norm.empty()) norm.assign(q, q + nq);
      norm[i].op &= ~SS_OP_ALL_TERMS_FREQUENT;
    }
    n_class[shape[i]]++;
  }
  if (!norm.empty()) *use = norm.data();
  if (n_class[SH_NATIVE] == nq) return SS_OK;
  *handled = true;
  const ss_bm25_query* Q = *use;
  if (n_class[SH_COMPOSE]) {  // (the rest of the batch comes back here through bm25_search_host_queries)
    std::vector<uint32_t> composed;
    for (uint32_t i = 0; i < nq; i++)
      if (shape[i] == SH_COMPOSE) composed.push_back(i);
    return bm25_search_compose(s, nq, Q, kk, rt, composed);
  }
  // sub-batches per kernel family; their answers are held until all have run (each writes s->d_out_* rows 0 ..)
  const uint32_t kw = std::max<uint32_t>(kk, 1);
  SS_HIP(hipSetDevice(s->device));
  SS_TRY(ensure_out(s, nq, kw));
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_q = 0, o_perm = o_q + al((size_t)nq * sizeof(ss_bm25_query)), o_doc = o_perm + al((size_t)nq * 4), o_sc = o_doc + al((size_t)nq * kw * 4),
               o_cnt = o_sc + al((size_t)nq * kw * 4), o_tot = o_cnt + al((size_t)nq * 4), need = o_tot + al((size_t)nq * 8);
  SS_HIP(hipStreamSynchronize(s->stream));  // the workspace may still be read by the batch before; the copies below are synchronous
  if (need > s->route_ws_cap) {
    if (s->d_route_ws) (void)hipFree(s->d_route_ws);
    s->d_route_ws = nullptr; s->route_ws_cap = 0;
    SS_HIP(hipMalloc(&s->d_route_ws, need * 2));
    s->route_ws_cap = need * 2;
  }
  char* W = (char*)s->d_route_ws;
  std::vector<ss_bm25_query> sub;
  std::vector<uint32_t> perm;
  uint32_t at = 0;
  for (uint8_t c : {SH_NATIVE, SH_GALLOP, SH_GALLOP_PHRASE}) {
    if (!n_class[c]) continue;
    sub.clear(); perm.clear();
    uint64_t longest = 0;
    for (uint32_t i = 0; i < nq; i++)
      if (shape[i] == c) { sub.push_back(Q[i]); perm.push_back(i); longest = std::max(longest, dlen[i]); }
    const uint32_t n = (uint32_t)sub.size();
    SS_HIP(hipMemcpy(W + o_perm + (size_t)at * 4, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    if (c == SH_NATIVE) {
      SS_TRY(bm25_search_host_queries(s, n, sub.data(), kk, rt, 0, nullptr));
    } else {
      SS_HIP(hipMemcpy(W + o_q + (size_t)at * sizeof(ss_bm25_query), sub.data(), (size_t)n * sizeof(ss_bm25_query), hipMemcpyHostToDevice));
      SS_TRY(ssi_bm25_gallop_search(s, n, (const ss_bm25_query*)(W + o_q) + at, c == SH_GALLOP_PHRASE, longest, kk, rt, s->d_out_doc, s->d_out_score,
                                    s->d_out_count, s->d_out_total, s->stream));
      s->gallop_batches++;
    }
    bm25_unpermute_kernel<<<n, 64, 0, s->stream>>>((const uint32_t*)(W + o_perm) + at, n, kk, s->d_out_doc, s->d_out_score, s->d_out_count,
                                                  (const unsigned long long*)s->d_out_total, (uint32_t*)(W + o_doc), (float*)(W + o_sc),
                                                  (uint32_t*)(W + o_cnt), (unsigned long long*)(W + o_tot));
    SS_HIP(hipGetLastError());
    at += n;
  }
  SS_TRY(ensure_out(s, nq, kw));  // (a sub-batch never needs more rows than the batch: the buffers are the ones reserved above)
  if (kk) {
    SS_HIP(hipMemcpyAsync(s->d_out_doc, W + o_doc, (size_t)nq * kw * 4, hipMemcpyDeviceToDevice, s->stream));
    SS_HIP(hipMemcpyAsync(s->d_out_score, W + o_sc, (size_t)nq * kw * 4, hipMemcpyDeviceToDevice, s->stream));
  }
  SS_HIP(hipMemcpyAsync(s->d_out_count, W + o_cnt, (size_t)nq * 4, hipMemcpyDeviceToDevice, s->stream));
  SS_HIP(hipMemcpyAsync(s->d_out_total, W + o_tot, (size_t)nq * 8, hipMemcpyDeviceToDevice, s->stream));
  return SS_OK;
}

// the search of ss_bm25_search_filtered / _sharded up to the device lists (s->d_out_*, on s->stream); caller holds s->mu
static int bm25_search_host_queries(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, uint32_t n_filters,
                                    const ss_facet_filter* filters) {
  // (a facet filter: its exclusion bitmap stands in for the tombstones of everything below -- every kernel family reads the same one)
  if (n_filters) return with_facet_filter(s, n_filters, filters, s->stream, [&]() { return bm25_search_host_queries(s, nq, q, kk, rt, 0, nullptr); });
  std::vector<ss_bm25_query> norm;
  {
    bool handled = false;
    const ss_bm25_query* use = q;
    SS_TRY(bm25_route_shapes(s, nq, q, kk, rt, norm, &use, &handled));
    if (handled) return SS_OK;
    q = use;
  }
  if (s->sp_n) {  // an image with a sparse tier: does the batch name one of its terms?
    bool any_sparse = false;
    for (uint32_t i = 0; i < nq && !any_sparse; i++)
      for (uint32_t t = 0; t < std::min<uint32_t>(q[i].n_terms + bm_q_nnot(q[i].op), SS_MAX_QUERY_TERMS); t++)
        any_sparse |= q[i].term[t] >= s->bm_n_terms / s->bm_n_fields && q[i].term[t] < s->bm_n_terms / s->bm_n_fields + s->sp_n;
    // (a facet filter: the sparse kernel reads the same exclusion bitmap as the dense ones)
    if (any_sparse) return with_facet_filter(s, n_filters, filters, s->stream, [&]() { return bm25_search_tiered(s, nq, q, kk, rt); });
  }
  SS_TRY(ssi_bm25_ensure_probe_rows(s, nq, q, s->stream));
  if (nq > 1) {  // phrase queries have a kernel of their own: a batch that mixes them with others runs as two, answers back in place
    std::vector<uint8_t> is_phrase(nq);
    uint32_t n_phrase = 0;
    for (uint32_t i = 0; i < nq; i++) n_phrase += (is_phrase[i] = bm_q_op(q[i].op) == SS_OP_PHRASE ? 1 : 0);
    if (n_phrase != 0 && n_phrase != nq) return bm25_search_split_batch(s, nq, q, kk, rt, n_filters, filters, is_phrase, n_phrase);
  }
  if (nq > 1 && s->bm_probe_rows != 0 && s->bm_probe_rows < s->bm_n_terms) {  // rationed probe rows: a mixed batch runs as two
    std::vector<uint8_t> probed(nq);
    uint32_t n_probed = 0;
    for (uint32_t i = 0; i < nq; i++) n_probed += (probed[i] = query_lists_probed(s, q[i]) ? 1 : 0);
    if (n_probed != 0 && n_probed != nq) {
      bool has_and, has_or, all_probed, any_frequent, phrase;
      uint32_t nt_max, np_max;
      bool any_filter_ = false, uniform_ = false, gated_ = false;
      SS_TRY(check_queries(s, nq, q, &has_and, &has_or, &nt_max, &np_max, &all_probed, &any_frequent, &phrase, &any_filter_, &uniform_, &gated_));  // the batch's own errors first
      return bm25_search_split_batch(s, nq, q, kk, rt, n_filters, filters, probed, n_probed);
    }
  }
  bool has_and = false, has_or = false;
  uint32_t nt_max = 0, np_max = 0, nn_max = 0;
  bool all_probed = false, any_frequent = false, phrase = false, any_filter = false, uniform = false, gated = false;
  SS_TRY(check_queries(s, nq, q, &has_and, &has_or, &nt_max, &np_max, &all_probed, &any_frequent, &phrase, &any_filter, &uniform, &gated, &nn_max));
  SS_HIP(hipSetDevice(s->device));
  SS_TRY(ensure_out(s, nq, std::max<uint32_t>(kk, 1)));
  if ((size_t)nq * sizeof(ss_bm25_query) > s->bq_cap) {
    if (s->d_bq) (void)hipFree(s->d_bq);
    s->d_bq = nullptr; s->bq_cap = 0;
    SS_HIP(hipMalloc(&s->d_bq, (size_t)nq * sizeof(ss_bm25_query)));
    s->bq_cap = (size_t)nq * sizeof(ss_bm25_query);
  }
  {
    const size_t qbytes = (size_t)nq * sizeof(ss_bm25_query);
    if (qbytes > s->h_bq_cap) {
      if (s->bq_ev_set) SS_HIP(hipEventSynchronize(s->bq_ev));
      if (s->h_bq) (void)hipHostFree(s->h_bq);
      s->h_bq = nullptr; s->h_bq_cap = 0;
      const size_t cap = std::max<size_t>(qbytes * 2, 64u << 10);
      SS_HIP(hipHostMalloc(&s->h_bq, cap, hipHostMallocDefault));
      s->h_bq_cap = cap;
    }
    if (!s->bq_ev) SS_HIP(hipEventCreateWithFlags(&s->bq_ev, hipEventDisableTiming));
    if (s->bq_ev_set) SS_HIP(hipEventSynchronize(s->bq_ev));  // (the copy of the batch before: long done unless the stream is backed up)
    memcpy(s->h_bq, q, qbytes);
    SS_HIP(hipMemcpyAsync(s->d_bq, s->h_bq, qbytes, hipMemcpyHostToDevice, s->stream));
    SS_HIP(hipEventRecord(s->bq_ev, s->stream));
    s->bq_ev_set = true;
  }
  {  // the batch's weight, for the pruned kernel's partition rule: the host has the queries in hand here
    uint64_t sum = 0;
    const bool mf = s->bm_n_fields > 1;
    for (uint32_t i = 0; i < nq; i++)
      for (uint32_t t = 0; t < q[i].n_terms && t < (uint32_t)SS_MAX_QUERY_TERMS; t++) {
        const uint32_t term = q[i].term[t];
        if (mf) { if (term < s->h_df_real.size()) sum += s->h_df_real[term]; }
        else if (term < s->h_df.size()) sum += s->h_df[term];
      }
    s->bm_batch_postings = sum / nq;
  }
  const int rc_search = with_facet_filter(s, n_filters, filters, s->stream, [&]() {
    return ssi_bm25_search(s, nq, (const ss_bm25_query*)s->d_bq, kk, rt, s->d_out_doc, s->d_out_score, s->d_out_count,
                           s->d_out_total, has_and, has_or, nt_max, np_max, all_probed, s->stream, any_frequent, phrase, any_filter, uniform, gated, nn_max);
  });
  s->bm_batch_postings = 0;
  return rc_search;
}

// ---- the one-launch path of small batches (bm25_small.hip).  Pinned block of the shard: three 64-byte flag slots (0: direct calls,
// 1 / 2: the coalescer's lanes), then answer staging for direct calls (SM_H_QUERIES queries x k <= 128).
constexpr uint32_t SM_H_QUERIES = 256;  // queries of ONE host-pointer call the path takes (launches of <= 64)
constexpr size_t SM_H_DOC = 256, SM_H_SCORE = SM_H_DOC + SM_H_QUERIES * 128 * 4, SM_H_COUNT = SM_H_SCORE + SM_H_QUERIES * 128 * 4, SM_H_TOTAL = SM_H_COUNT + SM_H_QUERIES * 4,
                 SM_H_BYTES = SM_H_TOTAL + SM_H_QUERIES * 8;
// Does ONE query fit the one-launch path?  What it takes: unions and intersections of <= 4 scored and <= 4 NOT terms over one list per
// term, no field filter, no all_terms_frequent mark; terms of either tier (a sparse term: k <= 32, and a sparse NOT term only where the
// sparse role meets it -- an intersection that has a sparse scored term); phrases of <= 4 unique terms that name a sparse term (all-dense
// phrases keep their staged kernel, bm25_phrase.hip).  Cheap: host tables only, nothing of the probe pool is touched.
// sh / nn_max: the query's part in its batch's shape.  An invalid query does not fit: the staged path reports it.
static bool small_query_fits(const ss_shard* s, const ss_bm25_query& Q, uint32_t kk, ss_small_shape* sh, uint32_t* nn_max) {
  const uint32_t L = s->bm_n_fields, n_dense = s->bm_n_terms / L, RF = bm_real_fields(s);
  const uint32_t np = Q.n_terms, nn = bm_q_nnot(Q.op), op = bm_q_op(Q.op);
  if (np == 0 || np > 4 || nn > 4 || op > (uint32_t)SS_OP_PHRASE || bm_q_all_frequent(Q.op) || (RF > 1 && bm_q_field_filter(Q.op)) ||
      (bm_q_field_filter(Q.op) >> RF))  // (a field the image does not have: the staged path reports it)
    return false;
  uint32_t nd = 0;
  bool sp_scored = false, sp_not = false;
  for (uint32_t t = 0; t < np + nn; t++) {
    if (Q.term[t] >= n_dense + s->sp_n) return false;
    if (t < np && !(Q.idf[t] > 0.0f)) return false;
    for (uint32_t u = 0; u < t; u++)
      if (Q.term[u] == Q.term[t]) return false;
    if (Q.term[t] >= n_dense) { if (t < np) sp_scored = true; else sp_not = true; }
    else if (t < np) nd++;
  }
  const bool is_and = op == SS_OP_INTERSECTION && np > 1;
  bool phrase = false;
  if (op == SS_OP_PHRASE) {
    if (!sp_scored || Q.phrase_len < 2 || Q.phrase_len > (uint32_t)SS_MAX_PHRASE || Q.phrase_seq[0] >= np) return false;
    for (uint32_t j = 1; j < Q.phrase_len; j++)
      if (Q.phrase_seq[j] >= np && Q.phrase_seq[j] != SS_PHRASE_SKIP) return false;
    if (!s->d_sp_pos_end || s->sp_pos_elem != (L > 1 ? 4u : 2u) || (nd && (L > 1 ? !s->d_pos32 : !s->d_pos))) return false;
    phrase = true;
  } else if (sp_not && !(is_and && sp_scored)) {
    return false;  // a sparse NOT list the dense roles would have to probe: the staged path's per-query exclusion bitmap
  }
  if (sp_scored || sp_not) {
    if (kk > 32 || (L > 1 && !(s->bm_merged && s->h_boost.size() == L))) return false;
    sh->any_sparse = true;
  }
  sh->any_phrase |= phrase;
  const bool dense_roles = nd != 0 && !(sp_scored && (is_and || phrase));  // (an intersection with a sparse term: the sparse role alone)
  if (dense_roles) {
    if (is_and && nd > 1) sh->has_and = true;
    // (counting workgroups: unions of two or more dense lists; ONE dense list where reading it to its end would be the price of counting
    // it in passing -- under tombstones, or beside a sparse role whose threshold lets it stop early; bm25_small.hip count_by_bits)
    else if (nd > 1 || ((sp_scored || s->n_deleted) && !phrase)) sh->has_or = true;
    sh->np_max = std::max(sh->np_max, nd);
  }
  sh->any_not |= nn != 0;
  *nn_max = std::max(*nn_max, nn);
  return true;
}

// Tries the batch on the one-launch path: *handled = false (and SS_OK) when it is not of that shape -- the staged pipeline then runs it.
// p_* = pinned buffers the kernel answers into (null: the shard's own staging).  Called under s->mu.
static int bm25_small_try(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t kk, uint32_t rt, uint32_t n_filters, uint32_t slot,
                          uint32_t* p_doc, float* p_score, uint32_t* p_count, uint64_t* p_total, bool* handled, uint32_t* seq_out) {
  *handled = false;
  // up to SM_H_QUERIES queries: launches of <= 64 back to back on the stream (each answers into its own rows; the flag shows the last
  // launch's number once ITS answers are in place, and the launches of one stream finish in order).  A 96-query call of the tiered image
  // was six launches and seven copies of the staged pipeline: 402 us; two launches: see profiles/r6_real_format_1m.log
  if (n_filters != 0 || kk == 0 || nq == 0 || nq > SM_H_QUERIES || !ssi_bm25_small_serves(s, std::min<uint32_t>(nq, 64u), kk, 1, 0)) return SS_OK;
  {
    ss_small_shape sh{};
    uint32_t nn_max = 0;
    for (uint32_t i = 0; i < nq; i++)
      if (!small_query_fits(s, q[i], kk, &sh, &nn_max)) return SS_OK;
    if (!ssi_bm25_small_serves(s, std::min<uint32_t>(nq, 64u), kk, sh.np_max, nn_max)) return SS_OK;
  }
  SS_TRY(ssi_bm25_ensure_probe_rows(s, nq, q, s->stream));
  for (uint32_t i = 0; i < nq; i++)  // every dense list the batch reads has a probe row now, or the scans take the batch
    if (!query_lists_probed_dense(s, q[i])) return SS_OK;
  SS_HIP(hipSetDevice(s->device));
  // slot 0 (direct calls, under ShardLock): the shard's stream and workspace.  Slots 1 / 2 (the coalescer's lanes): the lane's own
  // (ss_common.h lstream) -- created on first use; the launch waits for whatever s->stream holds at this moment.
  hipStream_t st = s->stream;
  void** wsp = &s->d_small_ws;
  if (slot != 0u) {
    const uint32_t li = slot - 1u;
    if (!s->ev_main) SS_HIP(hipEventCreateWithFlags(&s->ev_main, hipEventDisableTiming));
    // a BIG batch fills the chip by itself: two of them at once only get in each other's way (T = 256: batches of 90, 415 - 490 K q/s
    // overlapped against 490 - 510 K one behind the other) -- those queue on lane 0's stream whichever lane they come from (the workspace
    // and the flag stay the lane's own; a lane has one batch in flight at a time)
#ifndef SS_LANE_SHARE_FROM
#define SS_LANE_SHARE_FROM 48u  // (measured at 32 / 48 / 64 / never: profiles/r6_lane_streams3.log)
#endif
    constexpr uint32_t share_from = SS_LANE_SHARE_FROM;
    const uint32_t si = nq >= share_from ? 0u : li;
    if (!s->lstream[si]) {
      int pr_least = 0, pr_greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
      SS_HIP(hipStreamCreateWithPriority(&s->lstream[si], hipStreamNonBlocking, pr_greatest));
      s->main_dirty = true;  // (a new stream is ordered behind nothing)
    }
    st = s->lstream[si];
    s->lane_last[li] = st;
    wsp = &s->d_small_ws_l[li];
  }
  if (!*wsp) {
    SS_HIP(hipMalloc(wsp, ssi_bm25_small_ws_bytes()));
    SS_HIP(hipMemsetAsync(*wsp, 0, ssi_bm25_small_ws_bytes(), st));
  }
  SS_TRY(ssi_bm25_ensure_kth(s, s->stream));
  if (slot != 0u) {
    if (s->main_dirty) {  // both lanes behind the tail of s->stream as it stands (the flag is theirs together)
      SS_HIP(hipEventRecord(s->ev_main, s->stream));
      for (hipStream_t ls : s->lstream)
        if (ls) SS_HIP(hipStreamWaitEvent(ls, s->ev_main, 0));
      s->main_dirty = false;
    }
    s->lanes_inflight = true;
  }
  if (!s->h_small) {
    SS_HIP(hipHostMalloc((void**)&s->h_small, SM_H_BYTES, hipHostMallocDefault));
    memset(s->h_small, 0, SM_H_BYTES);
  }
  if (!p_doc) {
    p_doc = (uint32_t*)(s->h_small + SM_H_DOC); p_score = (float*)(s->h_small + SM_H_SCORE);
    p_count = (uint32_t*)(s->h_small + SM_H_COUNT); p_total = (uint64_t*)(s->h_small + SM_H_TOTAL);
  }
  uint32_t seq = 0;
  for (uint32_t c0 = 0; c0 < nq; c0 += 64u) {
    const uint32_t n = std::min<uint32_t>(64u, nq - c0);
    ss_small_shape sh{rt != SS_RT_TOPK, false, false, false, false, false, 0u};
    uint32_t nn_max = 0;
    for (uint32_t i = 0; i < n; i++) (void)small_query_fits(s, q[c0 + i], kk, &sh, &nn_max);  // this launch's own shape
    seq = ++s->small_seq ? s->small_seq : ++s->small_seq;  // never 0
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ssi_prof_begin(s, 0, st, &e0, &e1);
    const int rc = ssi_bm25_small_launch(s, *wsp, n, q + c0, kk, sh, p_doc + (size_t)c0 * kk, p_score + (size_t)c0 * kk, p_count + c0, p_total + c0,
                                        (uint32_t*)(s->h_small + 64 * slot), seq, st);
    ssi_prof_end(s, 0, st, e0, e1);
    if (rc != SS_OK) {  // (the per-query state may be half-way: start the next launch from zero; launches already queued finish first)
      (void)hipStreamSynchronize(st);
      (void)hipMemsetAsync(*wsp, 0, ssi_bm25_small_ws_bytes(), st);
      return rc;
    }
  }
  s->small_launches++;
  *seq_out = seq;
  *handled = true;
  return SS_OK;
}
// waits for the flag of `slot` to show `seq`: the kernel raises it behind the last answer.  Polled -- the answers are a few
// microseconds old when the loop sees it, where a stream synchronisation adds the completion signal's round trip
// mu_held: the caller holds s->mu (the direct path); the coalescer's lanes wait outside it
static int bm25_small_wait(ss_shard* s, uint32_t slot, uint32_t seq, bool mu_held) {
  volatile uint32_t* flag = (volatile uint32_t*)(s->h_small + 64 * slot);
  const auto t0 = std::chrono::steady_clock::now();
  int rc = SS_EDEVICE;
  for (uint32_t spins = 0;; spins++) {
    if (__atomic_load_n((const uint32_t*)flag, __ATOMIC_ACQUIRE) == seq) return SS_OK;
    __builtin_ia32_pause();
    if ((spins & 0xFFu) == 0xFFu) {
      const auto dt = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
      // a small batch is back within ~0.2 ms; past that the stream is backed up behind somebody's long work (a hybrid caller's vector
      // pass): give the core away between looks instead of burning it
      if (dt > 300) sched_yield();
      if (dt > 2000) {  // long past any small batch: ask the runtime (a kernel that died leaves the flag untouched)
        const hipError_t e = hipStreamQuery(slot ? s->lane_last[slot - 1u] : s->stream);
        if (e != hipSuccess && e != hipErrorNotReady) break;
        if (e == hipSuccess) { if (__atomic_load_n((const uint32_t*)flag, __ATOMIC_ACQUIRE) == seq) return SS_OK; break; }
        if (dt > 10000000) break;
        struct timespec ts = {0, 50000};
        nanosleep(&ts, nullptr);
      }
    }
  }
  // The launch did not come back: its per-query state (arrival counters, thresholds, totals) may be half-way, and the kernel may still
  // be writing the lane's pinned answers.  Drain the stream, start the next launch from zero, and tell the caller (VERDICT r5 weak 12,
  // ADVICE r5): without this every later one-launch batch would merge on a wrong "last arriver" or wait out the 10 s cap.
  std::unique_lock<std::mutex> g(s->mu, std::defer_lock);
  if (!mu_held) g.lock();  // (nobody enqueues another launch while the state is being zeroed)
  hipStream_t st = slot ? s->lane_last[slot - 1u] : s->stream;
  void* ws = slot ? s->d_small_ws_l[slot - 1u] : s->d_small_ws;
  (void)hipStreamSynchronize(st);
  if (ws) {
    (void)hipMemsetAsync(ws, 0, ssi_bm25_small_ws_bytes(), st);
    (void)hipStreamSynchronize(st);
  }
  return rc;
}

// The staged pipeline's answers home: rows [0, nq) of the shard's output arrays -> one pinned block [doc | score | count | total], the
// last block to finish raises the flag behind them (system-scope fence before its arrival, as bm25_small_kernel does).  The host polls
// the flag and copies into the caller's arrays: a 1000-query C2 call came home in 4 pageable copies + a stream synchronisation, ~95 us
// of its 770 (profiles/r6_bench_a.json: 0.769 ms end to end against 0.674 ms device-resident).
__global__ void __launch_bounds__(256) bm25_answers_home_kernel(const uint32_t* __restrict__ d_doc, const uint32_t* __restrict__ d_score,
                                                                const uint32_t* __restrict__ d_count, const uint32_t* __restrict__ d_total,
                                                                uint32_t n_cells, uint32_t nq, uint32_t* __restrict__ h, uint32_t* done,
                                                                uint32_t* flag, uint32_t seq) {
  const uint32_t n = 2u * n_cells + 3u * nq;  // dwords: docs, scores, counts, totals (two each)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t v;
    if (i < n_cells) v = d_doc[i];
    else if (i < 2u * n_cells) v = d_score[i - n_cells];
    else if (i < 2u * n_cells + nq) v = d_count[i - 2u * n_cells];
    else v = d_total[i - 2u * n_cells - nq];
    h[i] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t prev = atomicAdd(done, 1u);
    if (prev + 1u == gridDim.x) {
      *done = 0u;
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
static int bm25_small_wait(ss_shard* s, uint32_t slot, uint32_t seq, bool mu_held);
// rows [0, nq) of s->d_out_* -> the caller's arrays.  Called under s->mu, behind the search on s->stream.
static int bm25_answers_home(ss_shard* s, uint32_t nq, uint32_t kk, uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total) {
  const size_t cells = (size_t)nq * kk, dwords = 2 * cells + 3 * (size_t)nq;
  if (dwords > (64u << 20)) {  // (a huge call: the copies' bandwidth is what counts)
    if (kk) {
      SS_HIP(hipMemcpyAsync(out_doc, s->d_out_doc, cells * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
      SS_HIP(hipMemcpyAsync(out_score, s->d_out_score, cells * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    }
    SS_HIP(hipMemcpyAsync(out_count, s->d_out_count, (size_t)nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    SS_HIP(hipMemcpyAsync(out_total, s->d_out_total, (size_t)nq * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
    SS_HIP(hipStreamSynchronize(s->stream));
    return SS_OK;
  }
  if (!s->h_small) {
    SS_HIP(hipHostMalloc((void**)&s->h_small, SM_H_BYTES, hipHostMallocDefault));
    memset(s->h_small, 0, SM_H_BYTES);
  }
  if (dwords * 4 > s->h_ans_cap) {
    SS_HIP(hipStreamSynchronize(s->stream));
    if (s->h_ans) (void)hipHostFree(s->h_ans);
    s->h_ans = nullptr; s->h_ans_cap = 0;
    const size_t cap = std::max<size_t>(dwords * 4 * 2, 256u << 10);
    SS_HIP(hipHostMalloc((void**)&s->h_ans, cap, hipHostMallocDefault));
    s->h_ans_cap = cap;
  }
  if (!s->d_ans_done) {
    SS_HIP(hipMalloc(&s->d_ans_done, 64));
    SS_HIP(hipMemsetAsync(s->d_ans_done, 0, 64, s->stream));
  }
  const uint32_t seq = ++s->small_seq ? s->small_seq : ++s->small_seq;  // never 0
  const uint32_t grid = (uint32_t)std::min<size_t>(256, (dwords + 1023) / 1024);
  bm25_answers_home_kernel<<<std::max(grid, 1u), 256, 0, s->stream>>>(s->d_out_doc, (const uint32_t*)s->d_out_score, s->d_out_count, (const uint32_t*)s->d_out_total,
                                                                      (uint32_t)cells, nq, (uint32_t*)s->h_ans, s->d_ans_done, (uint32_t*)s->h_small, seq);
  SS_HIP(hipGetLastError());
  const int rc = bm25_small_wait(s, 0, seq, true);
  if (rc != SS_OK) { (void)hipMemsetAsync(s->d_ans_done, 0, 64, s->stream); return rc; }
  const uint32_t* h = (const uint32_t*)s->h_ans;
  if (kk) {
    memcpy(out_doc, h, cells * 4);
    memcpy(out_score, h + cells, cells * 4);
  }
  memcpy(out_count, h + 2 * cells, (size_t)nq * 4);
  memcpy(out_total, h + 2 * cells + nq, (size_t)nq * 8);
  return SS_OK;
}

static int bm25_search_direct(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t k, uint32_t rt, uint32_t n_filters,
                              const ss_facet_filter* filters, uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total) {
  ShardLock g(s);  // before check_queries: it reads the image's host-side tables (an upload replaces them)
  if (!s->d_post) return SS_ESTATE;
  const uint32_t kk = rt == SS_RT_COUNT ? 0 : k, kw = std::max<uint32_t>(kk, 1u);
  // by shape, like a coalesced batch (bm25_search_direct_lane): the queries that fit the one-launch path take it, the staged pipeline
  // runs the rest -- one all-dense phrase among a caller's 96 queries used to send all of them down the staged tiered pipeline
  static thread_local std::vector<ss_bm25_query> qs;
  static thread_local std::vector<uint32_t> row_of;
  uint32_t n_fit = nq;
  bool split = false;
  // (a rationed vocabulary: one pass over the whole batch deals the pool's rows -- no split)
  // (a call of more than SM_H_QUERIES queries is the staged pipeline's whole: its fixed cost is spread thin there -- 1000 C2 queries 0.70 ms --
  // where launches of 64 cost 0.1 ms each)
  if (kk && nq > 1 && nq <= SM_H_QUERIES && n_filters == 0 && !(s->bm_probe_rows != 0 && s->bm_probe_rows < s->bm_n_terms)) {
    ss_small_shape sh{};
    uint32_t nn_max = 0;
    std::vector<uint8_t> fits(nq);
    n_fit = 0;
    for (uint32_t i = 0; i < nq; i++) n_fit += (fits[i] = (n_fit < SM_H_QUERIES && small_query_fits(s, q[i], kk, &sh, &nn_max)) ? 1 : 0);
    if (n_fit != 0 && n_fit != nq) {
      split = true;
      qs.resize(nq); row_of.resize(nq);
      uint32_t a = 0, b = n_fit;
      for (uint32_t i = 0; i < nq; i++) { const uint32_t at = fits[i] ? a++ : b++; row_of[i] = at; qs[at] = q[i]; }
    } else {
      n_fit = nq;
    }
  }
  const ss_bm25_query* qq = split ? qs.data() : q;
  bool handled = false;
  uint32_t seq = 0;
  SS_TRY(bm25_small_try(s, n_fit, qq, kk, rt, n_filters, 0, nullptr, nullptr, nullptr, nullptr, &handled, &seq));
  const uint32_t r0 = handled ? n_fit : 0u, nr = nq - r0;  // what the staged pipeline runs: the rest, or (no launch) everything
  if (!split) {  // the whole call one way or the other: straight into the caller's arrays
    if (handled) {
      SS_TRY(bm25_small_wait(s, 0, seq, true));
      memcpy(out_doc, s->h_small + SM_H_DOC, (size_t)nq * kk * sizeof(uint32_t));
      memcpy(out_score, s->h_small + SM_H_SCORE, (size_t)nq * kk * sizeof(float));
      memcpy(out_count, s->h_small + SM_H_COUNT, (size_t)nq * sizeof(uint32_t));
      memcpy(out_total, s->h_small + SM_H_TOTAL, (size_t)nq * sizeof(uint64_t));
      return SS_OK;
    }
    SS_TRY(bm25_search_host_queries(s, nq, q, kk, rt, n_filters, filters));
    return bm25_answers_home(s, nq, kk, out_doc, out_score, out_count, out_total);
  }
  // split: the staged part behind the launch on the same stream, its answers into rows [r0, nq) of a host copy in run order
  static thread_local std::vector<uint32_t> t_doc, t_cnt;
  static thread_local std::vector<float> t_score;
  static thread_local std::vector<uint64_t> t_tot;
  t_doc.resize((size_t)nq * kw); t_score.resize((size_t)nq * kw); t_cnt.resize(nq); t_tot.resize(nq);
  int rc = SS_OK;
  if (nr) {
    rc = bm25_search_host_queries(s, nr, qq + r0, kk, rt, 0, nullptr);
    if (rc == SS_OK) {
      SS_HIP(hipStreamSynchronize(s->stream));  // (the launch ahead of it on the stream is done as well)
      SS_HIP(hipMemcpy(t_doc.data() + (size_t)r0 * kw, s->d_out_doc, (size_t)nr * kk * sizeof(uint32_t), hipMemcpyDeviceToHost));
      SS_HIP(hipMemcpy(t_score.data() + (size_t)r0 * kw, s->d_out_score, (size_t)nr * kk * sizeof(float), hipMemcpyDeviceToHost));
      SS_HIP(hipMemcpy(t_cnt.data() + r0, s->d_out_count, (size_t)nr * sizeof(uint32_t), hipMemcpyDeviceToHost));
      SS_HIP(hipMemcpy(t_tot.data() + r0, s->d_out_total, (size_t)nr * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
  }
  if (handled) {  // (waited for even when the staged part failed: the kernel writes the shard's pinned staging)
    const int rw = bm25_small_wait(s, 0, seq, true);
    if (rc == SS_OK) rc = rw;
    if (rc == SS_OK) {
      memcpy(t_doc.data(), s->h_small + SM_H_DOC, (size_t)n_fit * kk * sizeof(uint32_t));
      memcpy(t_score.data(), s->h_small + SM_H_SCORE, (size_t)n_fit * kk * sizeof(float));
      memcpy(t_cnt.data(), s->h_small + SM_H_COUNT, (size_t)n_fit * sizeof(uint32_t));
      memcpy(t_tot.data(), s->h_small + SM_H_TOTAL, (size_t)n_fit * sizeof(uint64_t));
    }
  }
  if (rc != SS_OK) return rc;
  for (uint32_t i = 0; i < nq; i++) {
    const size_t r = row_of[i];
    memcpy(out_doc + (size_t)i * kk, t_doc.data() + r * kw, (size_t)kk * sizeof(uint32_t));
    memcpy(out_score + (size_t)i * kk, t_score.data() + r * kw, (size_t)kk * sizeof(float));
    out_count[i] = t_cnt[r];
    out_total[i] = t_tot[r];
  }
  return SS_OK;
}

// the same for a coalesced batch on a lane: the shard mutex is held while the batch is ENQUEUED (queries from the lane's pinned staging,
// kernels, results back into it, the lane's event behind them -- all on s->stream); the wait for the event happens outside, so the next
// lane's leader can enqueue behind this batch at once.
// A coalesced batch is whoever happened to call at the same time: ONE query outside the one-launch shape (a phrase of dense words only,
// five terms, ...) must not send sixty-three others down the staged pipeline.  The batch is therefore split by shape: the queries that
// fit take the one launch (answers in rows 0 .. of the lane's pinned buffers), the rest the staged pipeline (rows behind them); row_of[i]
// = where query i's answer stands.
static int bm25_search_direct_lane(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t k, uint32_t rt, uint32_t* out_doc, float* out_score,
                                   uint32_t* out_count, uint64_t* out_total, hipEvent_t ev, uint32_t lane_slot, std::vector<uint32_t>& row_of) {
  const uint64_t t_in = g_co_trace_on ? co_now_us() : 0;
  bool small = false, staged = false;
  uint32_t small_seq = 0;
  row_of.resize(nq);
  for (uint32_t i = 0; i < nq; i++) row_of[i] = i;
  {
    std::lock_guard<std::mutex> g(s->mu);  // (NOT ShardLock: the other lane's kernel may be in flight, that is the point)
    if (!s->d_post) return SS_ESTATE;
    const uint32_t kk = rt == SS_RT_COUNT ? 0 : k, kw = std::max<uint32_t>(kk, 1u);
    std::vector<ss_bm25_query> qs;  // the batch in the order it runs in: the fitting queries first
    uint32_t n_fit = 0;
    if (kk && nq > 1 && nq <= SM_H_QUERIES && !(s->bm_probe_rows != 0 && s->bm_probe_rows < s->bm_n_terms)) {  // (a rationed vocabulary: one pass deals the pool's rows)
      ss_small_shape sh{};
      uint32_t nn_max = 0;
      std::vector<uint8_t> fits(nq);
      for (uint32_t i = 0; i < nq; i++) n_fit += (fits[i] = (n_fit < SM_H_QUERIES && small_query_fits(s, q[i], kk, &sh, &nn_max)) ? 1 : 0);  // (what one call takes)
      if (n_fit != 0 && n_fit != nq) {
        qs.resize(nq);
        uint32_t a = 0, b = n_fit;
        for (uint32_t i = 0; i < nq; i++) { const uint32_t at = fits[i] ? a++ : b++; row_of[i] = at; qs[at] = q[i]; }
        q = qs.data();
      } else {
        n_fit = nq;  // all of one kind: one attempt on the whole batch
      }
    } else {
      n_fit = nq;
    }
    // the lane's buffers are pinned: the one-launch path answers straight into them (flag slot 1 / 2 by the lane's event)
    SS_TRY(bm25_small_try(s, n_fit, q, kk, rt, 0, lane_slot, out_doc, out_score, out_count, out_total, &small, &small_seq));
    const uint32_t r0 = small ? n_fit : 0u, nr = nq - r0;  // what the staged pipeline runs: the rest, or (no launch) everything
    if (nr) {
      staged = true;
      SS_TRY(bm25_search_host_queries(s, nr, q + r0, kk, rt, 0, nullptr));
      if (kk) {
        SS_HIP(hipMemcpyAsync(out_doc + (size_t)r0 * kw, s->d_out_doc, (size_t)nr * kk * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        SS_HIP(hipMemcpyAsync(out_score + (size_t)r0 * kw, s->d_out_score, (size_t)nr * kk * sizeof(float), hipMemcpyDeviceToHost, s->stream));
      }
      SS_HIP(hipMemcpyAsync(out_count + r0, s->d_out_count, (size_t)nr * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
      SS_HIP(hipMemcpyAsync(out_total + r0, s->d_out_total, (size_t)nr * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
      SS_HIP(hipEventRecord(ev, s->stream));
    }
  }
  const uint64_t te = g_co_trace_on ? co_now_us() : 0;
  int rc = SS_OK;
  if (small) rc = bm25_small_wait(s, lane_slot, small_seq, false);
  if (staged && hipEventSynchronize(ev) != hipSuccess && rc == SS_OK) rc = SS_EDEVICE;  // (both waited for whatever the first says: the buffers are the lane's)
  if (g_co_trace_on) { g_co_trace.enqueue += te - t_in; g_co_trace.linger += co_now_us() - te; }
  return rc;
}

// ------------------------------------------------------------------ coalescing of concurrent callers (group commit)
static int vec_search_host(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k,
                           float thr, const ss_ann_mode* mode, uint32_t* out_doc, float* out_score, uint32_t* out_count,
                           uint64_t* out_total, uint32_t* out_clusters, const float* query_norm = nullptr);
static int vec_search_host_lists(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k,
                                 float thr, const ss_ann_mode* mode, uint32_t* h_count, uint32_t** d_ncl_out,
                                 const float* query_norm = nullptr, bool want_clusters = false, size_t out_slot = 0);
static int vec_search_host_lane(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k, float thr,
                                uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total);
namespace {
inline void co_futex_wait(std::atomic<uint32_t>* a, uint32_t expect) { (void)syscall(SYS_futex, (uint32_t*)a, FUTEX_WAIT_PRIVATE, expect, nullptr, nullptr, 0); }
inline void co_futex_wake(std::atomic<uint32_t>* a) { (void)syscall(SYS_futex, (uint32_t*)a, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }
// set a request's state (1 = done, 3 = lead) and wake its thread if it went to sleep
inline void co_signal(ss_co_req* r, uint32_t st) {
  const uint32_t old = r->state.exchange(st, std::memory_order_acq_rel);
  if (old & 4u) co_futex_wake(&r->state);
}
// wait until the state leaves "pending": a SHORT spin, then the futex.  (Spinning for about a batch's duration -- a sleeping
// follower costs its leader a system call, ~1.5 us each -- was measured and is wrong for a library: 64 spinning callers ran the
// process into its CPU quota, 185 K -> 47 K q/s with a p99 of 73 ms, the CFS throttling period; gpurun_out/conc_linger2.log)
inline void co_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
}
inline uint32_t co_wait(ss_co_req* r, uint32_t spin_us) {
  const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
  for (;;) {
    for (int i = 0; i < 64; i++) {
      const uint32_t v = r->state.load(std::memory_order_acquire);
      if (v != 0u) return v;
      co_cpu_relax();
    }
    if (std::chrono::steady_clock::now() >= until) break;
  }
  for (;;) {
    uint32_t v = 0u;
    if (r->state.compare_exchange_strong(v, 4u, std::memory_order_acq_rel)) v = 4u;
    if (v != 4u) return v;  // done / lead arrived before (or instead of) the sleep
    co_futex_wait(&r->state, 4u);
    v = r->state.load(std::memory_order_acquire);
    if (v != 4u) return v;
  }
}
// requests that can share a batch.  k may differ: the batch runs at the largest k and every member keeps its own prefix -- the
// result order is total (score descending, then doc id ascending), so the top-k' of a query is the head of its top-k.
// Only requests of one k CLASS merge (<= 64, <= 128, <= 256, more -- the kernels' top-k register budgets): a caller with a large k
// must not push the others off the pruned / 16-bit kernels, which serve k <= 128 / k <= 64.
inline uint32_t co_k_class(uint32_t k) { return k <= 64u ? 0u : k <= 128u ? 1u : k <= 256u ? 2u : 3u; }
inline bool co_compatible(const ss_co_req* a, const ss_co_req* b) {
  return (a->k == 0) == (b->k == 0) && co_k_class(a->k) == co_k_class(b->k) && a->rt == b->rt && a->elem == b->elem && a->thr == b->thr &&
         (a->qscale != nullptr) == (b->qscale != nullptr);
}

int co_run_lexical_one(ss_shard* s, ss_co_req* r) {
  return bm25_search_direct(s, r->nq, (const ss_bm25_query*)r->q, r->k, r->rt, 0, nullptr, r->out_doc, r->out_score, r->out_count, r->out_total);
}
int co_run_vector_one(ss_shard* s, ss_co_req* r) {
  return vec_search_host(s, r->nq, r->q, r->elem, r->qscale, r->k, r->thr, nullptr, r->out_doc, r->out_score, r->out_count, r->out_total, nullptr);
}

// one merged batch: the members' queries back to back, one search, every member's rows copied to its own buffers
int co_run_batch(ss_shard* s, ss_coalescer& co, bool lexical, const std::vector<ss_co_req*>& batch, uint32_t lane_ix) {
  ss_co_req* f = batch[0];
  ss_coalescer::Lane& ln = co.lane[lane_ix];
  uint32_t total = 0, kk = 0;
  for (ss_co_req* r : batch) { total += r->nq; kk = std::max(kk, r->k); }
  const uint32_t kw = std::max<uint32_t>(kk, 1u);
  const size_t qbytes = lexical ? sizeof(ss_bm25_query) : (size_t)s->dim * f->elem;
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const size_t o_q = 0, o_qs = o_q + al((size_t)total * qbytes), o_doc = o_qs + al((size_t)total * 4), o_sc = o_doc + al((size_t)total * kw * 4),
               o_cnt = o_sc + al((size_t)total * kw * 4), o_tot = o_cnt + al((size_t)total * 4), need = o_tot + al((size_t)total * 8);
  SS_HIP(hipSetDevice(s->device));
  if (need > ln.h_pin_cap) {
    if (ln.h_pin) (void)hipHostFree(ln.h_pin);
    ln.h_pin = nullptr; ln.h_pin_cap = 0;
    const size_t cap = std::max<size_t>(need * 2, 1u << 20);
    SS_HIP(hipHostMalloc((void**)&ln.h_pin, cap, hipHostMallocDefault));
    ln.h_pin_cap = cap;
  }
  if (!ln.ev) SS_HIP(hipEventCreateWithFlags(&ln.ev, hipEventDisableTiming));
  char* h_q = ln.h_pin + o_q;
  float* h_qs = (float*)(ln.h_pin + o_qs);
  uint32_t* h_doc = (uint32_t*)(ln.h_pin + o_doc);
  float* h_sc = (float*)(ln.h_pin + o_sc);
  uint32_t* h_cnt = (uint32_t*)(ln.h_pin + o_cnt);
  uint64_t* h_tot = (uint64_t*)(ln.h_pin + o_tot);
  const uint64_t tr0 = g_co_trace_on ? co_now_us() : 0;
  uint32_t at = 0;
  for (ss_co_req* r : batch) {
    memcpy(h_q + (size_t)at * qbytes, r->q, (size_t)r->nq * qbytes);
    if (f->qscale) memcpy(h_qs + at, r->qscale, (size_t)r->nq * sizeof(float));
    at += r->nq;
  }
  const uint64_t tr1 = g_co_trace_on ? co_now_us() : 0;
  int rc;
  static thread_local std::vector<uint32_t> row_of;  // lexical: the row of the lane's buffers that holds query i's answer (bm25_search_direct_lane)
  if (lexical)
    rc = bm25_search_direct_lane(s, total, (const ss_bm25_query*)h_q, kk, f->rt, h_doc, h_sc, h_cnt, h_tot, ln.ev, 1u + lane_ix, row_of);
  else
    rc = vec_search_host_lane(s, total, h_q, f->elem, f->qscale ? h_qs : nullptr, kk, f->thr, h_doc, h_sc, h_cnt, h_tot);
  if (rc != SS_OK) return rc;
  const uint64_t tr2 = g_co_trace_on ? co_now_us() : 0;
  at = 0;
  for (ss_co_req* r : batch) {
    for (uint32_t i = 0; i < r->nq; i++) {
      const size_t row = lexical ? row_of[at + i] : at + i;
      if (r->k) {
        memcpy(r->out_doc + (size_t)i * r->k, h_doc + row * kk, (size_t)r->k * sizeof(uint32_t));
        memcpy(r->out_score + (size_t)i * r->k, h_sc + row * kk, (size_t)r->k * sizeof(float));
      }
      r->out_count[i] = std::min(h_cnt[row], r->k);
      r->out_total[i] = h_tot[row];
    }
    r->rc = SS_OK;
    at += r->nq;
  }
  if (g_co_trace_on && lexical) {
    g_co_trace.n++; g_co_trace.stage += tr1 - tr0; g_co_trace.wait += tr2 - tr1; g_co_trace.scatter += co_now_us() - tr2;
  }
  return SS_OK;
}

int co_submit(ss_shard* s, ss_coalescer& co, bool lexical, ss_co_req* me) {
  bool lead;
  {
    std::lock_guard<std::mutex> g(co.mu);
    // a free lane: this thread leads a batch on it at once (nobody in flight: alone, straight away -- a lone caller pays nothing);
    // invariant: a non-empty queue always has a leader at work, or a successor already told to lead
    lead = co.leaders < ((co.n_lanes == 2u && (co.lanes_forced || co.callers_est >= co.lanes_from)) ? 2u : 1u);
    if (lead) {
      co.leaders++;
      me->lane = co.lane[0].busy ? 1u : 0u;
      co.lane[me->lane].busy = true;
    }
    co.queue.push_back(me);
    co.queued_nq.fetch_add(me->nq, std::memory_order_relaxed);
  }
  std::vector<ss_co_req*> batch;
  for (;;) {
    if (!lead) {
      // No spin before the sleep (round 6).  Rounds 3-5 spun 30 us first -- a request that arrives in the last 30 us of the batch in flight
      // saves a futex round trip -- and at T = 64 / 256 callers that was 12.7 / 16 cores busy (290 K .. 350 K calls a second x 30 us), the
      // whole CPU quota of the box at T = 256, throttled in 19 of 20 CFS periods.  Without it: T = 64 283 -> 297 K q/s on 4.2 cores,
      // T = 256 345 -> 393 K on 6.7 (p99 1.5 - 2.5 ms -> 0.85), T = 8 unchanged (profiles/r6_spin.log).
#ifndef CO_SPIN_US
#define CO_SPIN_US 0u
#endif
      const uint32_t v = co_wait(me, CO_SPIN_US);
      if (v == 1u) return me->rc;
      me->state.store(0u, std::memory_order_release);  // v == 3: this thread leads the next batch (its request is the queue's front)
      lead = true;
    }
    // Linger.  With T callers in a loop and one batch in flight, half of them ride in the batch and half wait: batches of T / 2
    // (vectors, T = 64: two 32-query passes of 7.5 ms where one 64-query pass takes 9 ms).  The callers a finished batch has just
    // released are back within microseconds, so the next leader waits for them -- until as many requests are queued as callers
    // seem to be around, at most an eighth of the last batch's duration (<= 1 ms), and only if that batch had company at all: a
    // lone caller is never delayed.  max_wait_us > 0 (ss_shard_set_coalescing) waits that long unconditionally.
    const uint64_t trace_t_lead_ns = g_co_trace_on && !lexical ? co_now_us() : 0;
    {
      uint32_t want, wait_us;
      {
        std::lock_guard<std::mutex> g(co.mu);
        // (several lanes: the callers around are shared between the batches in flight)
        want = co.max_wait_us ? co.max_batch : std::min(co.callers_est / std::max(co.leaders, 1u), co.max_batch);
        // (round 5, with the spin still there: an eighth of the last batch's duration, <= 1 ms -- measured at 8 / 4 / 2 / 1 on the C2 image,
        // T = 64: 247 / 245 / 233 / 254 K q/s at batches of 39 / 48 / 62 / 64; profiles/r5_linger.log)
        // VECTOR passes: a quarter, <= 3 ms.  A pass costs the same 9 ms for 44 queries as for 64, so whoever misses it pays a whole pass
        // more -- and a hybrid caller comes back through its lexical half first, which at k = 100 takes 0.5 - 1.5 ms for 64 callers
        // (profiles/r6_co_vec_trace.log: with the 1 ms cap 6 % of the hybrid searches missed their pass, p99 = 2.3 x p50).
        // LEXICAL, round 6 (the followers' spin gone, CPU to spare): up to the WHOLE of the last batch's duration, at most 150 us -- a
        // one-launch batch of top-10 queries takes ~100 us, so the callers it released are all back (one futex wake each, ~1.2 us apiece)
        // and every batch is full: T = 32 195 -> 245 K q/s, T = 64 303 -> 320 - 346 K (p99 280 -> 240 - 255 us), T = 8 p99 160 -> 94 us
        // (profiles/r6_linger_div.log).  The cap keeps a hybrid caller's lexical half (k = 100: 0.4 - 0.8 ms a batch) from lingering its
        // callers past their vector pass (uncapped: hybrid T = 64 p99 21 ms in one of two runs, r6_linger_div_hybrid.log).
        const uint32_t div = lexical ? 1u : 4u, cap = lexical ? 150u : 3000u;
        wait_us = co.max_wait_us ? co.max_wait_us : (co.callers_est > 1 ? std::min<uint32_t>(cap, co.last_batch_us / div) : 0u);
      }
      if (wait_us) {
        const auto t_in = std::chrono::steady_clock::now(), until = t_in + std::chrono::microseconds(wait_us);
        const auto hard_until = t_in + std::chrono::microseconds(std::max<uint32_t>(wait_us, std::min<uint32_t>(co.last_batch_us, 10000u)));
        // VECTOR passes also leave when the batch is NEARLY full and the arrivals have gone quiet: of 64 callers released together, 63 are
        // back within ~150 us; now and then one is not (the scheduler's doing) -- waiting out the cap for it costs the 63 a quarter of a
        // pass each (their p99: 12 - 14 ms against a p50 of 9.4, profiles/r6_tail_repeat.log), where going without it costs ONE caller a
        // second pass.  (A hybrid caller's lexical halves come back in two or three clumps half a millisecond apart: between clumps the
        // batch is half empty, and the rule does not fire.)
#ifndef CO_VEC_QUIET_US
#define CO_VEC_QUIET_US 700
#endif
        const uint32_t nearly = want - std::max(1u, want / 16u);
        uint32_t have_prev = 0;
        auto t_change = t_in;
        for (;;) {
          uint32_t have = 0;
          have = co.queued_nq.load(std::memory_order_relaxed);
          if (have >= want) break;
          const auto now = std::chrono::steady_clock::now();
          // members of the batch before that its leader has NOT EVEN WOKEN yet are certain to come: the wake loop is 64 system calls one
          // after the other, ~100 us as a rule -- and 6 ms now and then, when the leader loses its CPU in the middle (SS_CO_TRACE:
          // "wake loop us p50 103 p99 572 max 6318").  The three callers it had reached used to lead the next pass without the sixty it
          // had not (members p1 3): those paid a second pass, 19 ms.  Waited for beyond the cap, up to a pass's duration.
          const bool pending = co.waking.load(std::memory_order_relaxed) != 0u;
          if (now >= until && (!pending || now >= hard_until)) break;
          if (have != have_prev) { have_prev = have; t_change = now; }
          if (!lexical && !pending && want >= 16u && have >= nearly && now - t_change >= std::chrono::microseconds(CO_VEC_QUIET_US)) break;  // (longer than a lexical linger + batch: a hybrid caller's last clump)
          for (int i = 0; i < 32; i++) __builtin_ia32_pause();
        }
      }
    }
    const auto batch_t0 = std::chrono::steady_clock::now();
    const uint64_t trace_linger_ns = g_co_trace_on && !lexical ? co_now_us() : 0;
    batch.clear();
    {
      std::lock_guard<std::mutex> g(co.mu);
      uint32_t total = 0;
      // the leader's own request (told to lead as the queue's front, or just arrived -- then another lane's leader may have taken
      // requests ahead of it, never its own); with it every queued request that can share its batch
      ss_co_req* f = me;
      for (auto it = co.queue.begin(); it != co.queue.end(); ++it)
        if (*it == me) { co.queue.erase(it); co.queued_nq.fetch_sub(me->nq, std::memory_order_relaxed); break; }
      batch.push_back(me);  // first, whatever else fits
      total = me->nq;
      for (auto it = co.queue.begin(); it != co.queue.end();) {
        if ((*it)->lane == 0xFFFFFFFFu &&  // (a request that leads a lane itself is not a member of anybody's batch)
            co_compatible(f, *it) && total + (*it)->nq <= co.max_batch) {
          total += (*it)->nq;
          batch.push_back(*it);
          co.queued_nq.fetch_sub((*it)->nq, std::memory_order_relaxed);
          it = co.queue.erase(it);
        } else {
          ++it;
        }
      }
      co.batches++;
      co.queries += total;
      if (g_co_trace_on && !lexical) {
        uint32_t left = 0;
        for (ss_co_req* r : co.queue) left += r->nq;
        std::lock_guard<std::mutex> gt(g_co_vec_trace_mu);
        g_co_vec_trace.push_back(CoVecBatch{total, left, (uint32_t)((trace_linger_ns - trace_t_lead_ns) / 1000), 0u, trace_linger_ns / 1000});
      }
    }
    // (a lone VECTOR request takes the batch form too: the lane's pinned staging and the scan's own stream -- its pass must not hold the
    // shard mutex either)
    if (batch.size() == 1 && lexical) {
      batch[0]->rc = co_run_lexical_one(s, batch[0]);
    } else if (co_run_batch(s, co, lexical, batch, me->lane) != SS_OK) {
      // somebody's request is at fault (or the device is): every member is re-run alone and gets its own verdict
      for (ss_co_req* r : batch) r->rc = lexical ? co_run_lexical_one(s, r) : co_run_vector_one(s, r);
    }
    ss_co_req* succ = nullptr;
    {
      std::lock_guard<std::mutex> g(co.mu);
      uint32_t members = 0, queued = 0;
      for (ss_co_req* r : batch) members += r->nq;
      for (ss_co_req* r : co.queue) queued += r->nq;
      // Callers around = the most seen at the end of any of the last FOUR batches, and never less than 15/16 of the estimate before.  Callers
      // seen at THIS moment miss those on their way back (a hybrid caller is in its lexical half just now), and one straggler's batch of ONE
      // used to zero the estimate: the next leader then did not linger at all, took the three callers that happened to be there, and sixty
      // paid a second pass (profiles/r6_quiet_ab.log: "members p1 3").  A lone caller is not delayed: after four batches of one the
      // estimate is zero.
      const uint32_t seen = members + queued, floor_ = co.callers_est - std::max(1u, co.callers_est / 16u);
      co.seen_ring[co.seen_at++ & 3u] = seen;
      const uint32_t recent = std::max(std::max(co.seen_ring[0], co.seen_ring[1]), std::max(co.seen_ring[2], co.seen_ring[3]));
      co.callers_est = recent > 1u ? std::max(recent, co.callers_est > 1u ? floor_ : 0u) : 0u;
      co.last_batch_us = (uint32_t)std::min<long long>(1000000, std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - batch_t0).count());
      // hand the lane to the queue's front unless another lane's leader already told it to lead
      for (ss_co_req* r : co.queue)
        if (r->lane == 0xFFFFFFFFu) { succ = r; break; }
      if (succ) succ->lane = me->lane;
      else { co.leaders--; co.lane[me->lane].busy = false; }
    }
    const uint64_t tw0 = g_co_trace_on ? co_now_us() : 0;
    if (succ) co_signal(succ, 3u);  // first: the next batch forms while this one's members are being woken
    // (one futex wake per sleeping member, 1.26 us each.  Waking them faster was built twice and is wrong on a CPU quota: one shared futex
    // with FUTEX_WAKE(all), round 5, and a tree -- the leader wakes every 8th member, who wakes its group --, round 6: the leader's part falls
    // 47 -> 4.4 us per batch, all members run, spin and re-submit at once, and the process is throttled: T = 256 344 K -> 75 K q/s with
    // p99 = 78 ms, the CFS period; T = 64 283 K -> 220 K.  The serial wake is the pacing.  profiles/r6_tree_wake_rejected.log)
    // (re-measured in round 6 WITHOUT the followers' spin, which had been what ran the process into its quota: the tree still loses --
    // T = 64 299 -> 277 K q/s, T = 256 394 -> 100-140 K with 18 - 30 s of SYSTEM time per 2 s: members woken together re-submit together
    // and fight over the queue's mutex.  The serial wake paces that too.  profiles/r6_spin_tree.log)
    bool mine = false;
    co.waking.fetch_add((uint32_t)batch.size(), std::memory_order_relaxed);
    for (ss_co_req* r : batch) {
      if (r == me) mine = true;
      else co_signal(r, 1u);
      co.waking.fetch_sub(1u, std::memory_order_relaxed);
    }
    if (g_co_trace_on && lexical && batch.size() > 1) { g_co_trace.wake += co_now_us() - tw0; g_co_trace.members += batch.size() - 1; }
    if (g_co_trace_on && !lexical) {
      std::lock_guard<std::mutex> gt(g_co_vec_trace_mu);
      g_co_vec_wake_us.push_back((uint32_t)((co_now_us() - tw0) / 1000));
      g_co_vec_run_us.push_back((uint32_t)((tw0 - trace_linger_ns) / 1000));
    }
    if (mine) return me->rc;
    lead = false;  // (cannot happen while the leader's request is the front of its own batch; kept for safety)
  }
}
}  // namespace

int ss_shard_set_coalescing(ss_shard* s, uint32_t max_lexical_batch, uint32_t max_vector_batch, uint32_t max_wait_us) {
  if (!s || max_vector_batch > SS_VEC_BATCH) return SS_EINVAL;
  { std::lock_guard<std::mutex> g(s->co_lex.mu); s->co_lex.max_batch = max_lexical_batch; s->co_lex.max_wait_us = max_wait_us; }
  { std::lock_guard<std::mutex> g(s->co_vec.mu); s->co_vec.max_batch = max_vector_batch; s->co_vec.max_wait_us = max_wait_us; }
  return SS_OK;
}
int ss_bm25_path_stats(ss_shard* s, uint64_t* one_launch_batches) {
  if (!s || !one_launch_batches) return SS_EINVAL;
  ShardLock g(s);
  *one_launch_batches = s->small_launches;
  return SS_OK;
}
int ss_bm25_shape_stats(ss_shard* s, uint64_t* generic_batches) {
  if (!s || !generic_batches) return SS_EINVAL;
  ShardLock g(s);
  *generic_batches = s->gallop_batches;
  return SS_OK;
}
int ss_shard_coalescing_stats(ss_shard* s, uint64_t* lexical_batches, uint64_t* lexical_queries, uint64_t* vector_batches, uint64_t* vector_queries) {
  if (!s) return SS_EINVAL;
  { std::lock_guard<std::mutex> g(s->co_lex.mu); if (lexical_batches) *lexical_batches = s->co_lex.batches; if (lexical_queries) *lexical_queries = s->co_lex.queries; }
  { std::lock_guard<std::mutex> g(s->co_vec.mu); if (vector_batches) *vector_batches = s->co_vec.batches; if (vector_queries) *vector_queries = s->co_vec.queries; }
  return SS_OK;
}

// ------------------------------------------------------------------ deep pages: offset + length beyond SS_MAX_K results
// The crate's top_k = offset + length is unbounded (search.rs:1658-1659); the kernels' top-k structures hold SS_MAX_K.  A deeper page is
// answered in PASSES of SS_MAX_K results: the result order is total (score descending, then doc id ascending), so the docs ranked
// (p * SS_MAX_K, (p + 1) * SS_MAX_K] of a query are its top SS_MAX_K once the docs of the earlier passes are excluded -- and every kernel
// family already searches under an exclusion bitmap, the tombstones' (or a facet filter's, which stands in for it).  Pass p runs the
// ordinary search under  (what excluded docs before) | (the docs passes 0 .. p - 1 returned);  the totals are pass 0's (the first pass
// counts exactly what a k = SS_MAX_K call counts).  Every shape the library answers at k <= SS_MAX_K is answered at any k this way, on
// every image (both tiers, several indexed fields, phrases, composed unions); the cost is linear in k / SS_MAX_K, a page that deep is rare.
__global__ void peel_init_kernel(uint32_t* __restrict__ bits, uint32_t words, uint32_t rows, const uint32_t* __restrict__ base, uint32_t base_words) {
  const size_t n = (size_t)words * rows;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t w = (uint32_t)(i % words);
    bits[i] = (base && w < base_words) ? base[w] : 0u;
  }
}
// row r = blockIdx.y: the first min(counts[r], k) docs of docs[r * k ..] into bitmap r
__global__ void peel_mark_kernel(const uint32_t* __restrict__ docs, const uint32_t* __restrict__ counts, uint32_t k, uint32_t* __restrict__ bits,
                                 uint32_t words) {
  const uint32_t r = blockIdx.y;
  const uint32_t n = counts[r] < k ? counts[r] : k;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t doc = docs[(size_t)r * k + i];
    if (doc != SS_NO_DOC && (doc >> 5) < words) atomicOr(&bits[(size_t)r * words + (doc >> 5)], 1u << (doc & 31u));
  }
}
__global__ void peel_max_kernel(const uint32_t* __restrict__ v, unsigned long long n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = v[i] > m ? v[i] : m;
  for (int o = 32; o > 0; o >>= 1) { const uint32_t y = __shfl_down(m, o); m = y > m ? y : m; }
  if ((threadIdx.x & 63u) == 0) atomicMax(out, m);
}

namespace {
// the shard's exclusion bitmap replaced by the peel bitmap for the life of the object (the caller holds the shard lock)
struct PeelSwap {
  ss_shard* s;
  uint32_t* del; uint64_t dw, nd; uint32_t stride;
  PeelSwap(ss_shard* s_, uint32_t* bits, uint64_t words, uint32_t vec_stride) : s(s_), del(s_->d_deleted), dw(s_->deleted_words), nd(s_->n_deleted), stride(s_->vec_del_stride) {
    s->d_deleted = bits; s->deleted_words = words; s->n_deleted = 1; s->vec_del_stride = vec_stride;
  }
  ~PeelSwap() { s->d_deleted = del; s->deleted_words = dw; s->n_deleted = nd; s->vec_del_stride = stride; }
};
int peel_ensure(ss_shard* s, size_t dwords) {
  if (dwords <= s->peel_words_cap) return SS_OK;
  SS_HIP(hipStreamSynchronize(s->stream));
  if (s->d_peel_bits) (void)hipFree(s->d_peel_bits);
  s->d_peel_bits = nullptr; s->peel_words_cap = 0;
  SS_HIP(hipMalloc(&s->d_peel_bits, dwords * sizeof(uint32_t)));
  s->peel_words_cap = dwords;
  return SS_OK;
}
}  // namespace

// lexical: query by query (each pass is a single-query search of the ordinary paths).  Caller holds the shard lock; a facet filter's
// bitmap already stands in s->d_deleted (with_facet_filter).
static int bm25_search_deep_locked(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t k, uint32_t rt, uint32_t* out_doc, float* out_score,
                                   uint32_t* out_count, uint64_t* out_total) {
  const uint32_t base_words = s->n_deleted ? (uint32_t)s->deleted_words : 0u;
  const uint32_t words = std::max<uint32_t>((uint32_t)(((uint64_t)s->bm_n_docs + 31) / 32), base_words);
  SS_HIP(hipSetDevice(s->device));
  SS_TRY(peel_ensure(s, words));
  const uint32_t* base = s->n_deleted ? s->d_deleted : nullptr;
  std::vector<uint32_t> h_doc(SS_MAX_K);
  std::vector<float> h_score(SS_MAX_K);
  for (uint32_t i = 0; i < nq; i++) {
    uint32_t* od = out_doc + (size_t)i * k;
    float* os = out_score + (size_t)i * k;
    uint32_t got = 0;
    uint64_t total = 0;
    peel_init_kernel<<<std::min<uint32_t>(1024u, (words + 255u) / 256u), 256, 0, s->stream>>>(s->d_peel_bits, words, 1, base, base_words);
    SS_HIP(hipGetLastError());
    {
      PeelSwap sw(s, s->d_peel_bits, words, 0);
      for (uint32_t pass = 0; got < k; pass++) {
        const uint32_t kk = std::min<uint32_t>(SS_MAX_K, k - got);
        SS_TRY(bm25_search_host_queries(s, 1, q + i, kk, pass == 0 ? rt : (uint32_t)SS_RT_TOPK, 0, nullptr));
        uint32_t c = 0;
        SS_HIP(hipMemcpyAsync(&c, s->d_out_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        if (pass == 0) SS_HIP(hipMemcpyAsync(&total, s->d_out_total, sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
        SS_HIP(hipMemcpyAsync(h_doc.data(), s->d_out_doc, (size_t)kk * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        SS_HIP(hipMemcpyAsync(h_score.data(), s->d_out_score, (size_t)kk * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        SS_HIP(hipStreamSynchronize(s->stream));
        c = std::min(c, kk);
        memcpy(od + got, h_doc.data(), (size_t)c * sizeof(uint32_t));
        memcpy(os + got, h_score.data(), (size_t)c * sizeof(float));
        got += c;
        if (c < kk || got >= k) break;  // the list ran dry, or the page is full
        peel_mark_kernel<<<dim3(4, 1), 256, 0, s->stream>>>(s->d_out_doc, s->d_out_count, kk, s->d_peel_bits, words);
        SS_HIP(hipGetLastError());
      }
    }
    for (uint32_t r = got; r < k; r++) { od[r] = SS_NO_DOC; os[r] = 0.f; }
    out_count[i] = got;
    out_total[i] = total;
  }
  return SS_OK;
}
static int bm25_search_deep(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t k, uint32_t rt, uint32_t n_filters, const ss_facet_filter* filters,
                            uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total) {
  ShardLock g(s);
  if (!s->d_post) return SS_ESTATE;
  SS_HIP(hipSetDevice(s->device));
  return with_facet_filter(s, n_filters, filters, s->stream, [&]() { return bm25_search_deep_locked(s, nq, q, k, rt, out_doc, out_score, out_count, out_total); });
}

// vectors: groups of <= SS_VEC_BATCH queries, one exclusion bitmap PER QUERY of the group (vec_refine_kernel's del_stride); a pass is one
// scan of the group at k = SS_MAX_K.  The exclusion is by DOC (a doc of several records is returned once, by its best record).
static int vec_search_deep_locked(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k, float thr,
                                  const ss_ann_mode* mode, uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total,
                                  uint32_t* out_clusters, const float* query_norm);
static int vec_search_host_deep(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k, float thr,
                                const ss_ann_mode* mode, uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total,
                                uint32_t* out_clusters, const float* query_norm) {
  if (nq == 0) return SS_OK;
  ShardLock g(s);
  return vec_search_deep_locked(s, nq, queries, elem, query_scale, k, thr, mode, out_doc, out_score, out_count, out_total, out_clusters, query_norm);
}
static int vec_search_deep_locked(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k, float thr,
                                  const ss_ann_mode* mode, uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total,
                                  uint32_t* out_clusters, const float* query_norm) {
  if (s->vstream) SS_HIP(hipStreamSynchronize(s->vstream));
  SS_HIP(hipSetDevice(s->device));
  // the largest doc id a row can stand for
  uint64_t doc_bound = s->n_rows;
  if (s->d_row_doc) {
    SS_TRY(ensure_qstage(s, 256));
    SS_HIP(hipMemsetAsync(s->d_qstage, 0, sizeof(uint32_t), s->stream));
    peel_max_kernel<<<1024, 256, 0, s->stream>>>(s->d_row_doc, (unsigned long long)s->n_rows, (uint32_t*)s->d_qstage);
    uint32_t mx = 0;
    SS_HIP(hipMemcpyAsync(&mx, s->d_qstage, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    SS_HIP(hipStreamSynchronize(s->stream));
    doc_bound = (uint64_t)mx + 1;
  }
  const uint32_t base_words = s->n_deleted ? (uint32_t)s->deleted_words : 0u;
  const uint32_t words = std::max<uint32_t>((uint32_t)((doc_bound + 31) / 32), base_words);
  const uint32_t G = (uint32_t)std::min<size_t>(nq, SS_VEC_BATCH);
  SS_TRY(peel_ensure(s, (size_t)words * G));
  const uint32_t* base = s->n_deleted ? s->d_deleted : nullptr;
  const uint32_t ocw = (mode && (mode->flags & SS_ANN_REPORT_OBSERVED)) ? 3u : 1u;
  std::vector<uint32_t> h_doc((size_t)G * SS_MAX_K), h_count(G), got(G);
  std::vector<float> h_score((size_t)G * SS_MAX_K);
  std::vector<uint8_t> dry(G);
  for (uint32_t g0 = 0; g0 < nq; g0 += G) {
    const uint32_t nb = std::min<uint32_t>(G, nq - g0);
    peel_init_kernel<<<std::min<uint32_t>(4096u, (uint32_t)(((size_t)words * nb + 255u) / 256u)), 256, 0, s->stream>>>(s->d_peel_bits, words, nb, base, base_words);
    SS_HIP(hipGetLastError());
    std::fill(got.begin(), got.end(), 0u);
    std::fill(dry.begin(), dry.end(), (uint8_t)0);
    PeelSwap sw(s, s->d_peel_bits, words, words);
    for (uint32_t pass = 0;; pass++) {
      uint32_t* d_ncl = nullptr;
      SS_TRY(vec_search_host_lists(s, nb, (const char*)queries + (size_t)g0 * s->dim * elem, elem, query_scale ? query_scale + g0 : nullptr, SS_MAX_K, thr, mode,
                                   h_count.data(), &d_ncl, query_norm ? query_norm + g0 : nullptr, out_clusters != nullptr && pass == 0));
      SS_HIP(hipMemcpy(h_doc.data(), s->d_out_doc, (size_t)nb * SS_MAX_K * sizeof(uint32_t), hipMemcpyDeviceToHost));
      SS_HIP(hipMemcpy(h_score.data(), s->d_out_score, (size_t)nb * SS_MAX_K * sizeof(float), hipMemcpyDeviceToHost));
      if (pass == 0) {
        SS_HIP(hipMemcpy(out_total + g0, s->d_out_total, (size_t)nb * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (d_ncl && out_clusters) SS_HIP(hipMemcpy(out_clusters + (size_t)g0 * ocw, d_ncl, (size_t)nb * ocw * sizeof(uint32_t), hipMemcpyDeviceToHost));
      }
      bool more = false;
      for (uint32_t i = 0; i < nb; i++) {
        if (dry[i] || got[i] >= k) continue;
        const uint32_t c = std::min<uint32_t>(std::min<uint32_t>(h_count[i], SS_MAX_K), k - got[i]);
        memcpy(out_doc + (size_t)(g0 + i) * k + got[i], h_doc.data() + (size_t)i * SS_MAX_K, (size_t)c * sizeof(uint32_t));
        memcpy(out_score + (size_t)(g0 + i) * k + got[i], h_score.data() + (size_t)i * SS_MAX_K, (size_t)c * sizeof(float));
        got[i] += c;
        if (h_count[i] < SS_MAX_K) dry[i] = 1;
        more |= !dry[i] && got[i] < k;
      }
      if (!more) break;
      peel_mark_kernel<<<dim3(4, nb), 256, 0, s->stream>>>(s->d_out_doc, s->d_out_count, SS_MAX_K, s->d_peel_bits, words);
      SS_HIP(hipGetLastError());
    }
    for (uint32_t i = 0; i < nb; i++) {
      for (uint32_t r = got[i]; r < k; r++) { out_doc[(size_t)(g0 + i) * k + r] = SS_NO_DOC; out_score[(size_t)(g0 + i) * k + r] = 0.f; }
      out_count[g0 + i] = got[i];
      // f32 Euclidean: a pass is CUT by the scan's MFMA form of the distance and ordered by the rescored values (the reference's summation
      // order) -- across a seam the last of one pass and the first of the next may stand the other way round by a rounding: one stable sort
      if (s->vec_similarity == SS_SIM_EUCLIDEAN && elem == sizeof(float) && got[i] > SS_MAX_K) {
        uint32_t* dd = out_doc + (size_t)(g0 + i) * k;
        float* ss = out_score + (size_t)(g0 + i) * k;
        std::vector<uint32_t> order(got[i]);
        for (uint32_t r = 0; r < got[i]; r++) order[r] = r;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ss[a] > ss[b]; });
        std::vector<uint32_t> d2(got[i]);
        std::vector<float> s2(got[i]);
        for (uint32_t r = 0; r < got[i]; r++) { d2[r] = dd[order[r]]; s2[r] = ss[order[r]]; }
        memcpy(dd, d2.data(), (size_t)got[i] * 4);
        memcpy(ss, s2.data(), (size_t)got[i] * 4);
      }
    }
  }
  return SS_OK;
}

int ss_bm25_search_filtered(ss_shard* s, uint32_t nq, const ss_bm25_query* q, uint32_t k, uint32_t rt, uint32_t n_filters,
                            const ss_facet_filter* filters, uint32_t* out_doc, float* out_score, uint32_t* out_count,
                            uint64_t* out_total) {
  if (!s || !q || !out_count || !out_total) return SS_EINVAL;
  if (rt > SS_RT_TOPKCOUNT) return SS_EINVAL;
  if (rt != SS_RT_COUNT && (k == 0 || !out_doc || !out_score)) return SS_EINVAL;
  if (!s->d_post) return SS_ESTATE;
  if (nq == 0) return SS_OK;
  // (the crate's offset + length is unbounded, search.rs:1658-1659: a page deeper than SS_MAX_K results is answered in passes)
  if (rt != SS_RT_COUNT && k > SS_MAX_K) return bm25_search_deep(s, nq, q, k, rt, n_filters, filters, out_doc, out_score, out_count, out_total);
  if (n_filters == 0 && nq <= SS_COALESCE_MAX_REQUEST && s->co_lex.max_batch) {
    ss_co_req r;
    r.q = q; r.nq = nq; r.k = rt == SS_RT_COUNT ? 0u : k; r.rt = rt;
    r.out_doc = out_doc; r.out_score = out_score; r.out_count = out_count; r.out_total = out_total;
    return co_submit(s, s->co_lex, true, &r);
  }
  return bm25_search_direct(s, nq, q, k, rt, n_filters, filters, out_doc, out_score, out_count, out_total);
}

// One shard's part of <IndexArc as Search>::search over shards on different GPUs (search.rs:1637-1743 + 1875-2119): search
// this shard, ONE all-gather of the lists + an all-reduce of the totals over the communicator, merge on the device; every
// rank's caller receives the merged answer.  Collective: the rank of every shard of the communicator calls it with the
// same batch (in one process: one thread per shard).
int ss_bm25_search_sharded(ss_shard* s, ss_comm* c, uint32_t nq, const ss_bm25_query* q, uint32_t k, uint32_t rt,
                           uint64_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total) {
  if (!s || !c || !q || !out_total) return SS_EINVAL;
  if (rt > SS_RT_TOPKCOUNT) return SS_EINVAL;
  if (rt != SS_RT_COUNT && (k == 0 || !out_doc || !out_score || !out_count)) return SS_EINVAL;
  int dev = -1, n_ranks = 0;
  SS_TRY(ss_comm_info(c, nullptr, &n_ranks, &dev));
  const uint32_t kk = rt == SS_RT_COUNT ? 0 : k;
  if ((uint64_t)n_ranks * kk > 0xFFFFFFFFull) return SS_EINVAL;  // the same on every rank: nobody enters the collective
  if (nq == 0) return SS_OK;
  // From here on a failure is a matter of THIS shard (image missing, a query its lists cannot serve, an allocation): the rank
  // still enters the exchange, empty-handed, and every rank returns an error (ssi_comm_exchange) instead of blocking in it.
  ShardLock g(s);
  int rc = dev != s->device ? SS_EINVAL : !s->d_post ? SS_ESTATE : SS_OK;
  if (rc == SS_OK && kk > SS_MAX_K) {  // a deep page: this shard's (0, k) list in passes ("deep pages" above), then the exchange as ever
    std::vector<uint32_t> h_doc((size_t)nq * kk), h_cnt(nq);
    std::vector<float> h_score((size_t)nq * kk);
    std::vector<uint64_t> h_tot(nq);
    rc = bm25_search_deep_locked(s, nq, q, kk, rt, h_doc.data(), h_score.data(), h_cnt.data(), h_tot.data());
    if (rc == SS_OK) rc = ensure_out(s, nq, kk);
    if (rc == SS_OK && (hipMemcpy(s->d_out_doc, h_doc.data(), h_doc.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_score, h_score.data(), h_score.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_count, h_cnt.data(), (size_t)nq * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_total, h_tot.data(), (size_t)nq * 8, hipMemcpyHostToDevice) != hipSuccess))
      rc = SS_EDEVICE;
  } else if (rc == SS_OK) rc = bm25_search_host_queries(s, nq, q, kk, rt, 0, nullptr);
  const ss_dev_list L{s->d_out_doc, s->d_out_score, s->d_out_count, kk};
  return ssi_comm_exchange(c, nq, kk ? 1 : 0, &L, s->d_out_total, nullptr, rc, false, 0, 0, out_doc, out_score, nullptr, out_count, out_total,
                           s->stream);
}

// The vector and hybrid shard tasks of the same search (search.rs:1680-1689, 1723-1732; RRF after the gather, 1962-2035):
// this shard's f32 scan (AnnMode::All; host queries), the same single all-gather, the merged top-k on every rank.
int ss_vec_search_sharded(ss_shard* s, ss_comm* c, uint32_t nq, const float* queries, uint32_t k, float thr, uint64_t* out_doc,
                          float* out_score, uint32_t* out_count, uint64_t* out_total) {
  if (!s || !c || !queries || !out_doc || !out_score || !out_count || !out_total) return SS_EINVAL;
  if (k == 0) return SS_EINVAL;  // (any number of queries: the scan takes them SS_VEC_BATCH per pass, one all-gather for all of them)
  int dev = -1, n_ranks = 0;
  SS_TRY(ss_comm_info(c, nullptr, &n_ranks, &dev));
  if ((uint64_t)n_ranks * k > 0xFFFFFFFFull) return SS_EINVAL;
  if (nq == 0) return SS_OK;
  ShardLock g(s);
  int rc = dev != s->device ? SS_EINVAL : !s->d_X ? SS_ESTATE : SS_OK;
  std::vector<uint32_t> h_count(nq);
  if (rc == SS_OK && k > SS_MAX_K) {  // a deep page: this shard's list in passes, then the exchange as ever
    std::vector<uint32_t> h_doc((size_t)nq * k);
    std::vector<float> h_score((size_t)nq * k);
    std::vector<uint64_t> h_tot(nq);
    rc = vec_search_deep_locked(s, nq, queries, sizeof(float), nullptr, k, thr, nullptr, h_doc.data(), h_score.data(), h_count.data(), h_tot.data(), nullptr, nullptr);
    if (rc == SS_OK) rc = ensure_out(s, nq, k);
    if (rc == SS_OK && (hipMemcpy(s->d_out_doc, h_doc.data(), h_doc.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_score, h_score.data(), h_score.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_count, h_count.data(), (size_t)nq * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_total, h_tot.data(), (size_t)nq * 8, hipMemcpyHostToDevice) != hipSuccess))
      rc = SS_EDEVICE;
  } else if (rc == SS_OK) rc = vec_search_host_lists(s, nq, queries, sizeof(float), nullptr, k, thr, nullptr, h_count.data(), nullptr);
  const ss_dev_list L{s->d_out_doc, s->d_out_score, s->d_out_count, k};
  return ssi_comm_exchange(c, nq, 1, &L, s->d_out_total, nullptr, rc, false, 0, 0, out_doc, out_score, nullptr, out_count, out_total, s->stream);
}

// SearchMode::Hybrid over shards on different GPUs: both shard tasks with (offset 0, length k = offset + length), ONE all-gather
// of both lists, the two cross-shard concatenations sorted, RRF over them and sort / offset / length (search.rs:1962-2035,
// 2098-2119) on the device; result_count_total = sum over the shards of max(lexical, vector) (1919-1921).
// (comm.hip: the exchange with the hybrid fusion on the host when host_fuse)
extern "C++" int ssi_comm_exchange_hf(ss_comm* c, uint32_t nq, int n_lists, const ss_dev_list* L, const uint64_t* d_tot_a, const uint64_t* d_tot_b,
                         int local_rc, bool hybrid, uint32_t offset, uint32_t length, uint64_t* out_doc, float* out_score,
                         uint8_t* out_source, uint32_t* out_count, uint64_t* out_total, hipStream_t st, bool host_fuse);
int ss_hybrid_search_sharded(ss_shard* s, ss_comm* c, uint32_t nq, const ss_bm25_query* q, uint32_t rt, const float* queries, float thr,
                             uint32_t k, uint32_t offset, uint32_t length, uint64_t* out_doc, float* out_score, uint8_t* out_source,
                             uint32_t* out_count, uint64_t* out_total) {
  if (!s || !c || !q || !queries || !out_doc || !out_score || !out_count || !out_total) return SS_EINVAL;
  if (rt != SS_RT_TOPK && rt != SS_RT_TOPKCOUNT) return SS_EINVAL;
  if (k == 0 || length == 0) return SS_EINVAL;
  int dev = -1, n_ranks = 0;
  SS_TRY(ss_comm_info(c, nullptr, &n_ranks, &dev));
  if ((uint64_t)n_ranks * k > 0xFFFFFFFFull) return SS_EINVAL;
  // both concatenations live in the fusion kernel's LDS (8192 entries): beyond that, and for a page deeper than SS_MAX_K, they come home
  // after the exchange and the host fuses them (ss_merge_results: the same f32 operations in the same order) -- the same on every rank
  const bool host_fuse = k > SS_MAX_K || (uint64_t)n_ranks * k * 2 > 8192;
  if (nq == 0) return SS_OK;
  ShardLock g(s);
  int rc = dev != s->device ? SS_EINVAL : (!s->d_post || !s->d_X) ? SS_ESTATE : SS_OK;
  // outputs of the two searches side by side: the vector lists behind the lexical ones (reserved before either runs)
  if (rc == SS_OK && hipSetDevice(s->device) != hipSuccess) rc = SS_EDEVICE;
  std::vector<uint32_t> h_count(nq);
  if (rc == SS_OK && k > SS_MAX_K) {  // both shard tasks as deep pages ("deep pages" above), then placed as the exchange expects them
    std::vector<uint32_t> h_doc(2 * (size_t)nq * k), h_cnt(2 * (size_t)nq);
    std::vector<float> h_score(2 * (size_t)nq * k);
    std::vector<uint64_t> h_tot(2 * (size_t)nq);
    rc = bm25_search_deep_locked(s, nq, q, k, rt, h_doc.data(), h_score.data(), h_cnt.data(), h_tot.data());
    if (rc == SS_OK)
      rc = vec_search_deep_locked(s, nq, queries, sizeof(float), nullptr, k, thr, nullptr, h_doc.data() + (size_t)nq * k, h_score.data() + (size_t)nq * k,
                                  h_cnt.data() + nq, h_tot.data() + nq, nullptr, nullptr);
    if (rc == SS_OK) rc = ensure_out(s, 2 * (size_t)nq, k);
    if (rc == SS_OK && (hipMemcpy(s->d_out_doc, h_doc.data(), h_doc.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_score, h_score.data(), h_score.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_count, h_cnt.data(), h_cnt.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(s->d_out_total, h_tot.data(), h_tot.size() * 8, hipMemcpyHostToDevice) != hipSuccess))
      rc = SS_EDEVICE;
  } else {
    if (rc == SS_OK) rc = ensure_out(s, 2 * (size_t)nq, k);
    if (rc == SS_OK) rc = bm25_search_host_queries(s, nq, q, k, rt, 0, nullptr);
    if (rc == SS_OK) rc = vec_search_host_lists(s, nq, queries, sizeof(float), nullptr, k, thr, nullptr, h_count.data(), nullptr, nullptr, false, nq);
  }
  const ss_dev_list L[2] = {{s->d_out_doc, s->d_out_score, s->d_out_count, k},
                            {s->d_out_doc + (size_t)nq * k, s->d_out_score + (size_t)nq * k, s->d_out_count + nq, k}};
  return ssi_comm_exchange_hf(c, nq, 2, L, s->d_out_total, s->d_out_total + nq, rc, true, offset, length, out_doc, out_score, out_source,
                              out_count, out_total, s->stream, host_fuse);
}

// Facet counts of ONE query (query_facets / facet_count, add_result.rs:484-640): histogram of a facet over the query's match
// set (after NOT terms, tombstones and the facet filter).  out_counts [n_buckets + 1]: a string facet's ids 0 .. n_buckets-1,
// or the numeric ranges given by their ascending lower bounds; the last slot collects what falls outside.
static int facet_count_impl(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                            uint32_t facet_offset, uint32_t facet_type, const ss_facet_point* point, uint32_t n_buckets,
                            const uint64_t* range_lower_bounds, uint64_t* out_counts, uint64_t* out_total) {
  if (!s || !query || !out_counts || n_buckets == 0 || n_buckets > (1u << 24)) return SS_EINVAL;
  const bool string_facet = facet_type == SS_FACET_STRING16 || facet_type == SS_FACET_STRING32;
  if (facet_type > SS_FACET_POINT || (!string_facet && !range_lower_bounds)) return SS_EINVAL;
  if (facet_type == SS_FACET_POINT && (!point || point->unit > SS_POINT_MILES)) return SS_EINVAL;
  if (!s->d_post) return SS_ESTATE;
  bool has_and, has_or, all_probed, any_frequent;
  uint32_t nt_max, np_max;
  ShardLock g(s);
  SS_TRY(ssi_bm25_ensure_probe_rows(s, 1, query, s->stream));
  SS_TRY(check_queries(s, 1, query, &has_and, &has_or, &nt_max, &np_max, &all_probed, &any_frequent));
  if (!all_probed || !s->d_probe) return SS_ENOTSUP;  // the match set comes from the probe index's bit records
  SS_HIP(hipSetDevice(s->device));
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4, 8};
  if (!s->d_facets || s->facet_docs < s->bm_n_docs || facet_offset + width[facet_type] > s->facet_record_size) return SS_ESTATE;
  const uint64_t groups = (uint64_t)s->bm_n_sub * (BM_SUB / 64);
  const size_t bytes = sizeof(ss_bm25_query) + 8 + groups * 8 + ((size_t)n_buckets + 1) * 8 + (size_t)n_buckets * 8;
  if (bytes > s->facet_ws_cap) {  // grow-only workspace (a hipMalloc / hipFree pair per call would synchronise the device)
    if (s->d_facet_ws) (void)hipFree(s->d_facet_ws);
    s->d_facet_ws = nullptr;
    s->facet_ws_cap = 0;
    SS_HIP(hipMalloc(&s->d_facet_ws, bytes));
    s->facet_ws_cap = bytes;
  }
  char* ws = (char*)s->d_facet_ws;
  ss_bm25_query* d_q = (ss_bm25_query*)ws;
  unsigned long long* d_total = (unsigned long long*)(ws + sizeof(ss_bm25_query));
  unsigned long long* d_bits = d_total + 1;
  unsigned long long* d_counts = d_bits + groups;
  uint64_t* d_bounds = (uint64_t*)(d_counts + n_buckets + 1);
  int rc = SS_OK;
  if (hipMemcpyAsync(d_q, query, sizeof(ss_bm25_query), hipMemcpyHostToDevice, s->stream) != hipSuccess ||
      hipMemsetAsync(d_total, 0, 8 + groups * 8 + ((size_t)n_buckets + 1) * 8, s->stream) != hipSuccess ||
      (!string_facet &&
       hipMemcpyAsync(d_bounds, range_lower_bounds, (size_t)n_buckets * 8, hipMemcpyHostToDevice, s->stream) != hipSuccess))
    rc = SS_EDEVICE;
  if (rc == SS_OK)
    rc = with_facet_filter(s, n_filters, filters, s->stream, [&]() { return ssi_bm25_match_bits(s, d_q, d_bits, d_total, s->stream); });
  if (rc == SS_OK) rc = ssi_facet_count(s, d_bits, s->bm_n_docs, facet_offset, facet_type, n_buckets, d_bounds, d_counts, point, s->stream);
  if (rc == SS_OK && (hipMemcpyAsync(out_counts, d_counts, ((size_t)n_buckets + 1) * 8, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                      (out_total && hipMemcpyAsync(out_total, d_total, 8, hipMemcpyDeviceToHost, s->stream) != hipSuccess)))
    rc = SS_EDEVICE;
  if (hipStreamSynchronize(s->stream) != hipSuccess && rc == SS_OK) rc = SS_EDEVICE;
  return rc;
}

int ss_bm25_facet_count(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                        uint32_t facet_offset, uint32_t facet_type, uint32_t n_buckets, const uint64_t* range_lower_bounds,
                        uint64_t* out_counts, uint64_t* out_total) {
  if (facet_type > SS_FACET_STRING32) return SS_EINVAL;  // a Point facet is counted by its distances: ss_bm25_facet_count_point
  return facet_count_impl(s, query, n_filters, filters, facet_offset, facet_type, nullptr, n_buckets, range_lower_bounds, out_counts, out_total);
}
int ss_bm25_facet_count_point(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                              uint32_t facet_offset, const ss_facet_point* base, uint32_t n_buckets,
                              const uint64_t* range_lower_bounds, uint64_t* out_counts, uint64_t* out_total) {
  return facet_count_impl(s, query, n_filters, filters, facet_offset, SS_FACET_POINT, base, n_buckets, range_lower_bounds, out_counts, out_total);
}

// The pivot of a result sort (facet.hip): match set from the bit records as for facet counts, then the radix select.
static int facet_kth_impl(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                          uint32_t facet_offset, uint32_t facet_type, const ss_facet_point* point, uint32_t descending, uint64_t k,
                          uint64_t* out_value, uint64_t* out_n_better, uint64_t* out_n_equal, uint64_t* out_total) {
  if (!s || !query || !out_value || !out_n_better || !out_n_equal || k == 0) return SS_EINVAL;
  if (facet_type > SS_FACET_POINT) return SS_EINVAL;
  if (facet_type == SS_FACET_STRING16 || facet_type == SS_FACET_STRING32) return SS_ENOTSUP;
  if (facet_type == SS_FACET_POINT && (!point || point->unit > SS_POINT_MILES)) return SS_EINVAL;
  if (!s->d_post) return SS_ESTATE;
  bool has_and, has_or, all_probed, any_frequent;
  uint32_t nt_max, np_max;
  ShardLock g(s);
  SS_TRY(ssi_bm25_ensure_probe_rows(s, 1, query, s->stream));
  SS_TRY(check_queries(s, 1, query, &has_and, &has_or, &nt_max, &np_max, &all_probed, &any_frequent));
  if (!all_probed || !s->d_probe) return SS_ENOTSUP;
  SS_HIP(hipSetDevice(s->device));
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4, 8};
  if (!s->d_facets || s->facet_docs < s->bm_n_docs || facet_offset + width[facet_type] > s->facet_record_size) return SS_ESTATE;
  const uint64_t groups = (uint64_t)s->bm_n_sub * (BM_SUB / 64);
  const size_t bytes = sizeof(ss_bm25_query) + 8 + groups * 8 + 256 * 8;
  if (bytes > s->facet_ws_cap) {
    if (s->d_facet_ws) (void)hipFree(s->d_facet_ws);
    s->d_facet_ws = nullptr;
    s->facet_ws_cap = 0;
    SS_HIP(hipMalloc(&s->d_facet_ws, bytes));
    s->facet_ws_cap = bytes;
  }
  char* ws = (char*)s->d_facet_ws;
  ss_bm25_query* d_q = (ss_bm25_query*)ws;
  unsigned long long* d_total = (unsigned long long*)(ws + sizeof(ss_bm25_query));
  unsigned long long* d_bits = d_total + 1;
  unsigned long long* d_hist = d_bits + groups;
  SS_HIP(hipMemcpyAsync(d_q, query, sizeof(ss_bm25_query), hipMemcpyHostToDevice, s->stream));
  SS_HIP(hipMemsetAsync(d_total, 0, 8 + groups * 8, s->stream));
  SS_TRY(with_facet_filter(s, n_filters, filters, s->stream, [&]() { return ssi_bm25_match_bits(s, d_q, d_bits, d_total, s->stream); }));
  uint64_t total = 0;
  SS_HIP(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, s->stream));
  SS_HIP(hipStreamSynchronize(s->stream));
  if (out_total) *out_total = total;
  return ssi_facet_kth(s, d_bits, s->bm_n_docs, total, facet_offset, facet_type, descending != 0, k, d_hist, out_value, out_n_better,
                       out_n_equal, point, s->stream);
}
int ss_bm25_facet_kth(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                      uint32_t facet_offset, uint32_t facet_type, uint32_t descending, uint64_t k, uint64_t* out_value,
                      uint64_t* out_n_better, uint64_t* out_n_equal, uint64_t* out_total) {
  if (facet_type > SS_FACET_STRING32) return SS_EINVAL;
  return facet_kth_impl(s, query, n_filters, filters, facet_offset, facet_type, nullptr, descending, k, out_value, out_n_better, out_n_equal, out_total);
}
int ss_bm25_facet_kth_point(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                            uint32_t facet_offset, const ss_facet_point* base, uint32_t descending, uint64_t k, uint64_t* out_value,
                            uint64_t* out_n_better, uint64_t* out_n_equal, uint64_t* out_total) {
  return facet_kth_impl(s, query, n_filters, filters, facet_offset, SS_FACET_POINT, base, descending, k, out_value, out_n_better, out_n_equal, out_total);
}

// Result sort for a batch (facet.hip: "Result sort for a BATCH, pivots on the device").  The batch runs in chunks of <= 64 queries,
// every step a grid over the chunk: match sets (one expansion + one bit-record pass) -> per sort field the radix select and the
// classification -> the exclusion bitmaps -> TWO batched searches, each query inside its own doc set (the pruned kernel's filtered
// instances take one bitmap per query) -> compose into the queries' output rows.  One synchronisation per call.  A chunk the pruned
// kernel does not serve (k > 128, more than 4 lists per query, the exhaustive strategy) runs its searches query by query instead.
static int bm25_search_sorted_locked(ss_shard* s, uint32_t nq, const ss_bm25_query* queries, uint32_t n_sorts, const ss_result_sort* sorts, uint32_t k,
                                     uint32_t n_filters, const ss_facet_filter* filters, uint32_t* out_doc, float* out_score, uint32_t* out_count,
                                     uint64_t* out_total) {
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4, 8};
  if (!s->d_post) return SS_ESTATE;
  SS_HIP(hipSetDevice(s->device));
  if (!s->d_facets || s->facet_docs < s->bm_n_docs) return SS_ESTATE;
  for (uint32_t f = 0; f < n_sorts; f++)
    if (sorts[f].facet_offset + width[sorts[f].facet_type] > s->facet_record_size) return SS_ESTATE;
  const uint64_t groups = (uint64_t)s->bm_n_sub * (BM_SUB / 64);
  const uint32_t CH = std::min<uint32_t>(nq, 64u);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_q = 0, o_tot = o_q + al((size_t)CH * sizeof(ss_bm25_query)), o_state = o_tot + al((size_t)CH * 8), o_hist = o_state + al((size_t)CH * 64),
               o_E = o_hist + al((size_t)CH * 256 * 8), o_B = o_E + al((size_t)CH * groups * 8), o_xb = o_B + al((size_t)CH * groups * 8),
               o_xe = o_xb + al((size_t)CH * groups * 8), o_ad = o_xe + al((size_t)CH * groups * 8), o_as = o_ad + al((size_t)CH * k * 4),
               o_ac = o_as + al((size_t)CH * k * 4), o_cd = o_ac + al((size_t)CH * 4), o_cs = o_cd + al((size_t)CH * k * 4),
               o_cc = o_cs + al((size_t)CH * k * 4), o_at = o_cc + al((size_t)CH * 4), o_od = o_at + al((size_t)CH * 8),
               o_os = o_od + al((size_t)nq * k * 4), o_oc = o_os + al((size_t)nq * k * 4), o_ot = o_oc + al((size_t)nq * 4),
               need = o_ot + al((size_t)nq * 8);
  if (need > s->sort_ws_cap) {
    SS_HIP(hipStreamSynchronize(s->stream));
    if (s->d_sort_ws) (void)hipFree(s->d_sort_ws);
    s->d_sort_ws = nullptr; s->sort_ws_cap = 0;
    SS_HIP(hipMalloc(&s->d_sort_ws, need));
    s->sort_ws_cap = need;
  }
  char* W = (char*)s->d_sort_ws;
  ss_bm25_query* d_q = (ss_bm25_query*)(W + o_q);
  unsigned long long* d_total = (unsigned long long*)(W + o_tot);
  unsigned long long *d_E = (unsigned long long*)(W + o_E), *d_B = (unsigned long long*)(W + o_B), *d_xb = (unsigned long long*)(W + o_xb),
                     *d_xe = (unsigned long long*)(W + o_xe);
  SS_TRY(ensure_out(s, 1, k));
  constexpr int batched_off = 0;
  for (uint32_t c0 = 0; c0 < nq; c0 += CH) {
    const uint32_t nb = std::min<uint32_t>(CH, nq - c0);
    const ss_bm25_query* qc = queries + c0;
    bool has_and, has_or, all_probed, any_frequent, phrase = false, any_filter = false, uniform = false, gated = false;
    uint32_t nt_max, np_max, nn_max = 0;
    SS_TRY(ssi_bm25_ensure_probe_rows(s, nb, qc, s->stream));
    SS_TRY(check_queries(s, nb, qc, &has_and, &has_or, &nt_max, &np_max, &all_probed, &any_frequent, &phrase, &any_filter, &uniform, &gated, &nn_max));
    if (!all_probed || !s->d_probe || phrase) return SS_ENOTSUP;
    SS_HIP(hipMemcpyAsync(d_q, qc, (size_t)nb * sizeof(ss_bm25_query), hipMemcpyHostToDevice, s->stream));
    SS_TRY(with_facet_filter(s, n_filters, filters, s->stream, [&]() { return ssi_bm25_match_bits(s, d_q, d_E, d_total, s->stream, nb); }));
    SS_TRY(ssi_sort_select(s, nb, d_E, d_B, d_xb, d_xe, d_total, (unsigned long long*)(W + o_hist), W + o_state, n_sorts, sorts, k, s->stream));
    bool batched = !batched_off;
    for (int part = 0; part < 2 && batched; part++) {  // both searches as ONE batch each: every query under its own exclusion bitmap
      uint32_t* del = s->d_deleted;
      const uint64_t dw = s->deleted_words, nd = s->n_deleted;
      s->d_deleted = (uint32_t*)(part == 0 ? d_xb : d_xe); s->deleted_words = groups * 2; s->n_deleted = 1; s->del_per_query = 1;
      const int rc = ssi_bm25_search(s, nb, d_q, k, SS_RT_TOPK, (uint32_t*)(W + (part == 0 ? o_ad : o_cd)), (float*)(W + (part == 0 ? o_as : o_cs)),
                                     (uint32_t*)(W + (part == 0 ? o_ac : o_cc)), (uint64_t*)(W + o_at), has_and, has_or, nt_max, np_max, all_probed,
                                     s->stream, any_frequent, false, any_filter, uniform, gated, nn_max);
      s->d_deleted = del; s->deleted_words = dw; s->n_deleted = nd; s->del_per_query = 0;
      if (rc == SS_ENOTSUP && part == 0) batched = false;  // not the pruned kernel's batch: query by query below
      else if (rc != SS_OK) return rc;
    }
    if (!batched) {
      for (uint32_t i = 0; i < nb; i++)
        for (int part = 0; part < 2; part++) {
          uint32_t* del = s->d_deleted;
          const uint64_t dw = s->deleted_words, nd = s->n_deleted;
          s->d_deleted = (uint32_t*)((part == 0 ? d_xb : d_xe) + (size_t)i * groups); s->deleted_words = groups * 2; s->n_deleted = 1;
          const int rc = bm25_search_host_queries(s, 1, qc + i, k, SS_RT_TOPK, 0, nullptr);
          s->d_deleted = del; s->deleted_words = dw; s->n_deleted = nd;
          if (rc != SS_OK) return rc;
          SS_HIP(hipMemcpyAsync(W + (part == 0 ? o_ad : o_cd) + (size_t)i * k * 4, s->d_out_doc, (size_t)k * 4, hipMemcpyDeviceToDevice, s->stream));
          SS_HIP(hipMemcpyAsync(W + (part == 0 ? o_as : o_cs) + (size_t)i * k * 4, s->d_out_score, (size_t)k * 4, hipMemcpyDeviceToDevice, s->stream));
          SS_HIP(hipMemcpyAsync(W + (part == 0 ? o_ac : o_cc) + (size_t)i * 4, s->d_out_count, 4, hipMemcpyDeviceToDevice, s->stream));
        }
    }
    SS_TRY(ssi_sort_compose(s, nb, (const uint32_t*)(W + o_ad), (const float*)(W + o_as), (const uint32_t*)(W + o_ac), (const uint32_t*)(W + o_cd),
                            (const float*)(W + o_cs), (const uint32_t*)(W + o_cc), d_total, n_sorts, sorts, k, (uint32_t*)(W + o_od) + (size_t)c0 * k,
                            (float*)(W + o_os) + (size_t)c0 * k, (uint32_t*)(W + o_oc) + c0, (unsigned long long*)(W + o_ot) + c0, s->stream));
  }
  SS_HIP(hipMemcpyAsync(out_doc, W + o_od, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s->stream));
  SS_HIP(hipMemcpyAsync(out_score, W + o_os, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s->stream));
  SS_HIP(hipMemcpyAsync(out_count, W + o_oc, (size_t)nq * 4, hipMemcpyDeviceToHost, s->stream));
  SS_HIP(hipMemcpyAsync(out_total, W + o_ot, (size_t)nq * 8, hipMemcpyDeviceToHost, s->stream));
  SS_HIP(hipStreamSynchronize(s->stream));
  return SS_OK;
}

int ss_bm25_search_sorted(ss_shard* s, uint32_t nq, const ss_bm25_query* queries, uint32_t n_sorts, const ss_result_sort* sorts, uint32_t k,
                          uint32_t n_filters, const ss_facet_filter* filters, uint32_t* out_doc, float* out_score, uint32_t* out_count,
                          uint64_t* out_total) {
  if (!s || !queries || nq == 0 || !out_doc || !out_score || !out_count || !out_total || k == 0) return SS_EINVAL;
  if (n_sorts == 0) return ss_bm25_search_filtered(s, nq, queries, k, SS_RT_TOPKCOUNT, n_filters, filters, out_doc, out_score, out_count, out_total);
  if (!sorts) return SS_EINVAL;
  if (n_sorts > SS_MAX_SORT_FIELDS) return SS_ENOTSUP;
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4, 8};
  for (uint32_t f = 0; f < n_sorts; f++) {
    if (sorts[f].facet_type > SS_FACET_POINT) return SS_EINVAL;
    if (sorts[f].facet_type == SS_FACET_STRING16 || sorts[f].facet_type == SS_FACET_STRING32) return SS_ENOTSUP;  // by their strings: the host's rank column
  }
  ShardLock g(s);  // (before the image is looked at: a commit swaps its arrays under this lock)
  if (k <= SS_MAX_K) return bm25_search_sorted_locked(s, nq, queries, n_sorts, sorts, k, n_filters, filters, out_doc, out_score, out_count, out_total);
  // a DEEP page sorted by facets: the order is total (field 1, ..., field n, score desc, doc asc), so it is peeled like any other ("deep
  // pages" above) -- query by query, every pass the sorted search of SS_MAX_K results under (tombstones | the docs of the earlier passes);
  // a facet filter's bitmap is built on top of that inside the pass
  if (!s->d_post) return SS_ESTATE;
  SS_HIP(hipSetDevice(s->device));
  const uint32_t base_words = s->n_deleted ? (uint32_t)s->deleted_words : 0u;
  const uint32_t words = std::max<uint32_t>((uint32_t)(((uint64_t)s->bm_n_docs + 31) / 32), base_words);
  SS_TRY(peel_ensure(s, words));
  const uint32_t* base = s->n_deleted ? s->d_deleted : nullptr;
  std::vector<uint32_t> h_doc(SS_MAX_K);
  std::vector<float> h_score(SS_MAX_K);
  for (uint32_t i = 0; i < nq; i++) {
    uint32_t got = 0;
    uint64_t total = 0;
    peel_init_kernel<<<std::min<uint32_t>(1024u, (words + 255u) / 256u), 256, 0, s->stream>>>(s->d_peel_bits, words, 1, base, base_words);
    SS_HIP(hipGetLastError());
    PeelSwap sw(s, s->d_peel_bits, words, 0);
    for (uint32_t pass = 0; got < k; pass++) {
      const uint32_t kk = std::min<uint32_t>(SS_MAX_K, k - got);
      uint32_t c = 0;
      uint64_t t = 0;
      SS_TRY(bm25_search_sorted_locked(s, 1, queries + i, n_sorts, sorts, kk, n_filters, filters, h_doc.data(), h_score.data(), &c, &t));
      if (pass == 0) total = t;
      c = std::min(c, kk);
      memcpy(out_doc + (size_t)i * k + got, h_doc.data(), (size_t)c * sizeof(uint32_t));
      memcpy(out_score + (size_t)i * k + got, h_score.data(), (size_t)c * sizeof(float));
      got += c;
      if (c < kk || got >= k) break;
      SS_TRY(ensure_qstage(s, (size_t)SS_MAX_K * 4 + 16));  // the pass's docs back to the device for the mark
      SS_HIP(hipMemcpyAsync(s->d_qstage, h_doc.data(), (size_t)c * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
      SS_HIP(hipMemcpyAsync((char*)s->d_qstage + (size_t)SS_MAX_K * 4, &c, sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
      peel_mark_kernel<<<dim3(4, 1), 256, 0, s->stream>>>((const uint32_t*)s->d_qstage, (const uint32_t*)((char*)s->d_qstage + (size_t)SS_MAX_K * 4), kk, s->d_peel_bits, words);
      SS_HIP(hipGetLastError());
      SS_HIP(hipStreamSynchronize(s->stream));  // (h_doc and c are read by the copies)
    }
    for (uint32_t r = got; r < k; r++) { out_doc[(size_t)i * k + r] = SS_NO_DOC; out_score[(size_t)i * k + r] = 0.f; }
    out_count[i] = got;
    out_total[i] = total;
  }
  return SS_OK;
}

static int facet_values_impl(ss_shard* s, uint32_t n, const uint32_t* doc_ids, uint32_t facet_offset, uint32_t facet_type,
                             const ss_facet_point* point, uint64_t* out_values) {
  if (!s || (n && (!doc_ids || !out_values)) || facet_type > SS_FACET_POINT) return SS_EINVAL;
  if (point && point->unit > SS_POINT_MILES) return SS_EINVAL;
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4, 8};
  ShardLock g(s);
  if (!s->d_facets || facet_offset + width[facet_type] > s->facet_record_size) return SS_ESTATE;
  if (n == 0) return SS_OK;
  SS_HIP(hipSetDevice(s->device));
  SS_TRY(ensure_qstage(s, (size_t)n * 12));
  uint32_t* d_docs = (uint32_t*)((char*)s->d_qstage + (size_t)n * 8);
  unsigned long long* d_out = (unsigned long long*)s->d_qstage;
  SS_HIP(hipMemcpyAsync(d_docs, doc_ids, (size_t)n * 4, hipMemcpyHostToDevice, s->stream));
  SS_TRY(ssi_facet_values(s, d_docs, n, facet_offset, facet_type, d_out, point, s->stream));
  SS_HIP(hipMemcpyAsync(out_values, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
  SS_HIP(hipStreamSynchronize(s->stream));
  return SS_OK;
}
int ss_facet_values(ss_shard* s, uint32_t n, const uint32_t* doc_ids, uint32_t facet_offset, uint32_t facet_type, uint64_t* out_values) {
  return facet_values_impl(s, n, doc_ids, facet_offset, facet_type, nullptr, out_values);  // a Point facet: its Morton codes
}
int ss_facet_point_distances(ss_shard* s, uint32_t n, const uint32_t* doc_ids, uint32_t facet_offset, const ss_facet_point* base,
                             uint64_t* out_values) {
  if (!base) return SS_EINVAL;
  return facet_values_impl(s, n, doc_ids, facet_offset, SS_FACET_POINT, base, out_values);
}

int ss_bm25_search_dev(ss_shard* s, uint32_t nq, const ss_bm25_query* d_q, uint32_t k, uint32_t rt, uint32_t ops_mask,
                       uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total,
                       void* stream) {
  return ss_bm25_search_filtered_dev(s, nq, d_q, k, rt, ops_mask, 0, nullptr, d_out_doc, d_out_score, d_out_count, d_out_total, stream);
}

int ss_bm25_search_filtered_dev(ss_shard* s, uint32_t nq, const ss_bm25_query* d_q, uint32_t k, uint32_t rt, uint32_t ops_mask,
                                uint32_t n_filters, const ss_facet_filter* filters, uint32_t* d_out_doc, float* d_out_score,
                                uint32_t* d_out_count, uint64_t* d_out_total, void* stream) {
  if (!s || !d_q || !d_out_count || !d_out_total) return SS_EINVAL;
  if (rt > SS_RT_TOPKCOUNT) return SS_EINVAL;
  if (rt != SS_RT_COUNT && (k == 0 || !d_out_doc || !d_out_score)) return SS_EINVAL;
  if (!s->d_post) return SS_ESTATE;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  hipStream_t st = stream ? (hipStream_t)stream : s->stream;
  if (rt != SS_RT_COUNT && k > SS_MAX_K) {
    // a DEEP page ("deep pages" above): the passes are steered from the host, so the queries make one trip there (like the tiered batches
    // below) and the call is synchronous; the page is left in the caller's device arrays, ordered on `st`
    std::vector<ss_bm25_query> hq(nq);
    SS_HIP(hipMemcpyAsync(hq.data(), d_q, (size_t)nq * sizeof(ss_bm25_query), hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
    std::vector<uint32_t> h_doc((size_t)nq * k), h_cnt(nq);
    std::vector<float> h_score((size_t)nq * k);
    std::vector<uint64_t> h_tot(nq);
    SS_TRY(with_facet_filter(s, n_filters, filters, s->stream, [&]() {
      return bm25_search_deep_locked(s, nq, hq.data(), k, rt, h_doc.data(), h_score.data(), h_cnt.data(), h_tot.data());
    }));
    SS_HIP(hipMemcpyAsync(d_out_doc, h_doc.data(), h_doc.size() * 4, hipMemcpyHostToDevice, st));
    SS_HIP(hipMemcpyAsync(d_out_score, h_score.data(), h_score.size() * 4, hipMemcpyHostToDevice, st));
    SS_HIP(hipMemcpyAsync(d_out_count, h_cnt.data(), (size_t)nq * 4, hipMemcpyHostToDevice, st));
    SS_HIP(hipMemcpyAsync(d_out_total, h_tot.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
    SS_HIP(hipStreamSynchronize(st));  // (the host copies are this frame's)
    return SS_OK;
  }
  if ((ops_mask & (1u << 28)) && s->sp_n) {
    // the batch may name terms of the SPARSE tier: splitting it into its dense and sparse parts is host work, so the queries make
    // one trip to the host (the cost the header states); the answers are left in the caller's device arrays, ordered on `st`
    const uint32_t kk = rt == SS_RT_COUNT ? 0 : k, kw = std::max<uint32_t>(kk, 1);
    std::vector<ss_bm25_query> hq(nq);
    SS_HIP(hipMemcpyAsync(hq.data(), d_q, (size_t)nq * sizeof(ss_bm25_query), hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
    SS_TRY(bm25_search_host_queries(s, nq, hq.data(), kk, rt, n_filters, filters));
    if (kk) {
      SS_HIP(hipMemcpyAsync(d_out_doc, s->d_out_doc, (size_t)nq * kw * 4, hipMemcpyDeviceToDevice, s->stream));
      SS_HIP(hipMemcpyAsync(d_out_score, s->d_out_score, (size_t)nq * kw * 4, hipMemcpyDeviceToDevice, s->stream));
    }
    SS_HIP(hipMemcpyAsync(d_out_count, s->d_out_count, (size_t)nq * 4, hipMemcpyDeviceToDevice, s->stream));
    SS_HIP(hipMemcpyAsync(d_out_total, s->d_out_total, (size_t)nq * 8, hipMemcpyDeviceToDevice, s->stream));
    SS_HIP(hipStreamSynchronize(s->stream));  // (this path has made a trip to the host already; nothing of this frame is left in flight)
    return SS_OK;
  }
  return with_facet_filter(s, n_filters, filters, st, [&]() {
    return ssi_bm25_search(s, nq, d_q, rt == SS_RT_COUNT ? 0 : k, rt, d_out_doc, d_out_score, d_out_count, d_out_total,
                           (ops_mask & 1u) != 0, (ops_mask & 2u) != 0 || (ops_mask & 3u) == 0,
                           (ops_mask >> 8) & 0xFFu ? (ops_mask >> 8) & 0xFFu : SS_MAX_QUERY_TERMS,
                           (ops_mask >> 16) & 0xFFu ? (ops_mask >> 16) & 0xFFu
                                                    : ((ops_mask >> 8) & 0xFFu ? (ops_mask >> 8) & 0xFFu : SS_MAX_QUERY_TERMS),
                           // the caller vouches for the probe rows of its terms (ss_bm25_term_probed) unless none were rationed
                           s->bm_probe_rows != 0 && (s->bm_probe_rows >= s->bm_n_terms || (ops_mask & 4u) != 0), st,
                           (ops_mask & 8u) != 0, (ops_mask & 16u) != 0, (ops_mask & 32u) != 0, (ops_mask & 64u) != 0, (ops_mask & 128u) != 0,
                           (ops_mask >> 24) & 0xFu ? (ops_mask >> 24) & 0xFu : 0xFFFFFFFFu);
  });
}

// ------------------------------------------------------------------ vectors
// end of every vector-image build: the Euclidean side data (f32: the two augmented columns; i8: the records' sums of
// squares), then the per-batch workspace
static int vec_finish(ss_shard* s) {
  if (s->vec_similarity == SS_SIM_EUCLIDEAN) {
    int rc = s->d_X ? ssi_vec_augment(s, s->stream) : ssi_vec8_row_sq(s, s->stream);
    if (rc) return rc;
    SS_HIP(hipStreamSynchronize(s->stream));
  }
  return ssi_vec_alloc_ws(s);
}

static int vec_alloc(ss_shard* s, uint64_t n_rows, uint32_t dim) {
  free_vec(s);
  s->n_rows = n_rows;
  s->dim = dim;
  // Euclidean: the scan computes -|q - x|^2 as the dot product of [x, |x|^2, 1] with [2 q, -1, -|q|^2] -- two more columns
  const uint32_t dim_img = dim + (s->vec_similarity == SS_SIM_EUCLIDEAN ? 2u : 0u);
  s->dim_pad = (dim_img + VS_KC - 1) / VS_KC * VS_KC;
  s->n_rows_pad = (n_rows + VS_TR - 1) / VS_TR * VS_TR;
  const size_t bytes = (size_t)s->n_rows_pad * s->dim_pad * sizeof(float);
  SS_HIP(hipMalloc(&s->d_X, bytes));
  if (s->dim_pad != dim) SS_HIP(hipMemsetAsync(s->d_X, 0, bytes, s->stream));
  else if (s->n_rows_pad != n_rows)
    SS_HIP(hipMemsetAsync(s->d_X + (size_t)n_rows * s->dim_pad, 0, (size_t)(s->n_rows_pad - n_rows) * s->dim_pad * sizeof(float),
                          s->stream));
  return SS_OK;
}

int ss_vec_upload(ss_shard* s, uint64_t n_rows, uint32_t dim, const float* rows, const uint32_t* row_doc_ids) {
  if (!s || !rows || n_rows == 0 || dim == 0) return SS_EINVAL;
  if (n_rows > 0xFFFFFFFEull) return SS_ENOTSUP;
  bool multi = false;
  if (row_doc_ids) {  // several records per doc (one per field x chunk, vector.rs:561-576) -> dedup in the refine kernel
    std::vector<uint32_t> tmp(row_doc_ids, row_doc_ids + n_rows);
    std::sort(tmp.begin(), tmp.end());
    multi = std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end();
    if (!tmp.empty() && tmp.back() == SS_NO_DOC) return SS_EINVAL;
  }
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  int rc = vec_alloc(s, n_rows, dim);
  if (rc) { free_vec(s); return rc; }
  SS_HIP(hipMemcpy2DAsync(s->d_X, (size_t)s->dim_pad * sizeof(float), rows, (size_t)dim * sizeof(float),
                          (size_t)dim * sizeof(float), n_rows, hipMemcpyHostToDevice, s->stream));
  s->vec_multi_record = multi;
  if (row_doc_ids) {
    SS_HIP(hipMalloc(&s->d_row_doc, n_rows * sizeof(uint32_t)));
    SS_HIP(hipMemcpyAsync(s->d_row_doc, row_doc_ids, n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
  }
  SS_HIP(hipStreamSynchronize(s->stream));
  return vec_finish(s);
}

// vector.bin (writer vector.rs:1066-1094, reader 1279-1298): per level u32 cluster_count, cluster_count x u32 child_count,
// then the records cluster after cluster, each a packed 24-byte VectorHeader (u16 doc_id first, vector.rs:62-73) + dim x f32.
// Shard-local doc id of a record = (level << 16) | doc_id (vector.rs:1448).  The payloads go to HBM straight from the
// file bytes with a strided copy per level; the cluster structure is kept for the ANN modes (vec_ann.hip).
static int vec8_alloc(ss_shard* s, uint64_t n_rows, uint32_t dim);

// i8 = Precision::I8 records (dim x i8 after the header); use_scale keeps VectorHeader.scale (f32 at byte 10) per record
// for dot_i8_quantized (ScalarQuantizationI8 with Dot / Euclidean; Cosine scores are the raw integer dot, vector.rs:1331-1333)
static int vec_bin_upload(ss_shard* s, const uint8_t* bytes, uint64_t len, uint32_t dim, bool i8, bool use_scale) {
  if (!s || !bytes || dim == 0) return SS_EINVAL;
  const uint64_t rec = 24u + (uint64_t)dim * (i8 ? 1u : 4u);
  struct Lvl { uint64_t first, n; };
  std::vector<Lvl> levels;
  std::vector<uint32_t> ids, level_clusters, child_counts;
  std::vector<uint16_t> fields;  // VectorHeader.field_id (u32 at byte 2); the reference compares it as u16 (vector.rs:1398)
  std::vector<float> scales, norms;
  uint64_t pos = 0;
  while (pos < len) {
    if (pos + 4 > len) return SS_EINVAL;
    uint32_t clusters;
    memcpy(&clusters, bytes + pos, 4);
    pos += 4;
    if (pos + (uint64_t)clusters * 4u > len) return SS_EINVAL;
    uint64_t n = 0;
    for (uint32_t c = 0; c < clusters; c++) {
      uint32_t child;
      memcpy(&child, bytes + pos + 4ull * c, 4);
      n += child;
      child_counts.push_back(child);
    }
    level_clusters.push_back(clusters);
    pos += (uint64_t)clusters * 4u;
    if (n > (len - pos) / rec || levels.size() >= 65536u) return SS_EINVAL;
    for (uint64_t r = 0; r < n; r++) {
      uint16_t d;
      memcpy(&d, bytes + pos + r * rec, 2);
      ids.push_back((uint32_t)(levels.size() << 16) | d);
      uint32_t fid;
      memcpy(&fid, bytes + pos + r * rec + 2, 4);
      fields.push_back((uint16_t)fid);
      if (use_scale) {
        float sc;
        memcpy(&sc, bytes + pos + r * rec + 10, 4);
        scales.push_back(sc);
        memcpy(&sc, bytes + pos + r * rec + 14, 4);  // VectorHeader.norm: euclidean_i8_quantized
        norms.push_back(sc);
      }
    }
    levels.push_back({pos, n});
    pos += n * rec;
  }
  const uint64_t n_rows = ids.size();
  if (n_rows == 0) return SS_EINVAL;
  if (n_rows > 0xFFFFFFFEull) return SS_ENOTSUP;
  std::vector<uint32_t> tmp(ids);
  std::sort(tmp.begin(), tmp.end());
  const bool multi = std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end();
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  int rc = i8 ? vec8_alloc(s, n_rows, dim) : vec_alloc(s, n_rows, dim);
  if (rc) { free_vec(s); return rc; }
  int8_t* stage = nullptr;  // i8: row-major staging on the device, permuted into fragment order afterwards
  if (i8 && hipMalloc(&stage, (size_t)n_rows * dim) != hipSuccess) { free_vec(s); return SS_ENOMEM; }
  uint64_t row = 0;
  for (const Lvl& l : levels) {
    if (l.n == 0) continue;
    hipError_t e = i8 ? hipMemcpy2DAsync(stage + row * dim, (size_t)dim, bytes + l.first + 24u, rec, (size_t)dim, l.n,
                                         hipMemcpyHostToDevice, s->stream)
                      : hipMemcpy2DAsync(s->d_X + row * s->dim_pad, (size_t)s->dim_pad * sizeof(float), bytes + l.first + 24u, rec,
                                         (size_t)dim * sizeof(float), l.n, hipMemcpyHostToDevice, s->stream);
    if (e != hipSuccess) { if (stage) (void)hipFree(stage); free_vec(s); return SS_EDEVICE; }
    row += l.n;
  }
  if (i8) {
    rc = ssi_vec8_permute(s, stage, s->stream);
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(stage);
    if (rc) { free_vec(s); return rc; }
    if (use_scale) {
      SS_HIP(hipMalloc(&s->d_row_scale, n_rows * sizeof(float)));
      SS_HIP(hipMemcpyAsync(s->d_row_scale, scales.data(), n_rows * sizeof(float), hipMemcpyHostToDevice, s->stream));
      SS_HIP(hipMalloc(&s->d_row_norm, n_rows * sizeof(float)));
      SS_HIP(hipMemcpyAsync(s->d_row_norm, norms.data(), n_rows * sizeof(float), hipMemcpyHostToDevice, s->stream));
    }
  }
  s->vec_multi_record = multi;
  SS_HIP(hipMalloc(&s->d_row_doc, n_rows * sizeof(uint32_t)));
  SS_HIP(hipMemcpyAsync(s->d_row_doc, ids.data(), n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
  SS_HIP(hipMalloc(&s->d_row_field, n_rows * sizeof(uint16_t)));
  SS_HIP(hipMemcpyAsync(s->d_row_field, fields.data(), n_rows * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
  SS_HIP(hipStreamSynchronize(s->stream));
  rc = ssi_vec_set_clusters(s, (uint32_t)level_clusters.size(), level_clusters.data(), child_counts.data());
  if (rc != SS_OK && rc != SS_ENOTSUP) { free_vec(s); return rc; }  // ENOTSUP (an empty cluster): AnnMode::All only
  return vec_finish(s);
}

int ss_vec_upload_vector_bin(ss_shard* s, const uint8_t* bytes, uint64_t len, uint32_t dim) {
  return vec_bin_upload(s, bytes, len, dim, false, false);
}

int ss_vec_upload_vector_bin_i8(ss_shard* s, const uint8_t* bytes, uint64_t len, uint32_t dim, int use_record_scale) {
  return vec_bin_upload(s, bytes, len, dim, true, use_record_scale != 0);
}

int ss_vec_synth(ss_shard* s, uint64_t seed, uint64_t n_rows, uint32_t dim) {
  if (!s || n_rows == 0 || dim == 0) return SS_EINVAL;
  if (n_rows > 0xFFFFFFFEull) return SS_ENOTSUP;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  int rc = vec_alloc(s, n_rows, dim);
  if (rc) { free_vec(s); return rc; }
  SS_TRY(ssi_vec_synth(s, seed, s->stream));
  SS_HIP(hipStreamSynchronize(s->stream));
  return vec_finish(s);
}

int ss_vec_info(ss_shard* s, uint64_t* n_rows, uint32_t* dim) {
  if (!s) return SS_EINVAL;
  if (!s->d_X && !s->d_X8) return SS_ESTATE;
  if (n_rows) *n_rows = s->n_rows;
  if (dim) *dim = s->dim;
  return SS_OK;
}

int ss_vec_read_rows(ss_shard* s, uint64_t r0, uint64_t n, float* out) {
  if (!s || !out) return SS_EINVAL;
  if (!s->d_X) return SS_ESTATE;
  if (r0 + n > s->n_rows) return SS_EINVAL;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  SS_HIP(hipMemcpy2D(out, (size_t)s->dim * sizeof(float), s->d_X + r0 * s->dim_pad, (size_t)s->dim_pad * sizeof(float),
                     (size_t)s->dim * sizeof(float), n, hipMemcpyDeviceToHost));
  return SS_OK;
}

// the search up to the device lists (s->d_out_* on s->stream, counts checked on the host: an overflowed batch is re-run in
// safe mode); caller holds s->mu.  h_count: nq words of host scratch.
static int vec_search_host_lists(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k,
                                 float thr, const ss_ann_mode* mode, uint32_t* h_count, uint32_t** d_ncl_out,
                                 const float* query_norm, bool want_clusters, size_t out_slot) {
  SS_HIP(hipSetDevice(s->device));
  SS_TRY(ensure_out(s, out_slot + nq, k));  // out_slot: the lists go behind those of `out_slot` earlier queries (hybrid)
  uint32_t* const o_doc = s->d_out_doc + out_slot * k;
  float* const o_score = s->d_out_score + out_slot * k;
  uint32_t* const o_count = s->d_out_count + out_slot;
  uint64_t* const o_total = s->d_out_total + out_slot;
  const size_t qbytes = ((size_t)nq * s->dim * elem + 15) & ~(size_t)15;
  const size_t sbytes = query_scale ? (size_t)nq * sizeof(float) : 0;
  const size_t nbytes = query_norm ? (size_t)nq * sizeof(float) : 0;
  SS_TRY(ensure_qstage(s, qbytes + sbytes + nbytes + (want_clusters ? (size_t)nq * 3 * sizeof(uint32_t) : 0)));  // (three words with SS_ANN_REPORT_OBSERVED)
  float* d_qs = query_scale ? (float*)((char*)s->d_qstage + qbytes) : nullptr;
  float* d_qn = query_norm ? (float*)((char*)s->d_qstage + qbytes + sbytes) : nullptr;
  uint32_t* d_ncl = want_clusters ? (uint32_t*)((char*)s->d_qstage + qbytes + sbytes + nbytes) : nullptr;
  if (d_ncl_out) *d_ncl_out = d_ncl;
  int rc = SS_OK;
  AnnStateGuard ann_guard(s, s->stream, mode);
  for (int attempt = 0; attempt < 2; attempt++) {
    if (hipMemcpyAsync(s->d_qstage, queries, (size_t)nq * s->dim * elem, hipMemcpyHostToDevice, s->stream) != hipSuccess) { rc = SS_EDEVICE; break; }
    if (d_qs && hipMemcpyAsync(d_qs, query_scale, sbytes, hipMemcpyHostToDevice, s->stream) != hipSuccess) { rc = SS_EDEVICE; break; }
    if (d_qn && hipMemcpyAsync(d_qn, query_norm, nbytes, hipMemcpyHostToDevice, s->stream) != hipSuccess) { rc = SS_EDEVICE; break; }
    rc = ssi_vec_search(s, nq, s->d_qstage, d_qs, k, thr, o_doc, o_score, o_count, o_total, s->stream,
                        attempt == 1, mode, d_ncl, d_qn);
    if (rc) break;
    if (hipMemcpyAsync(h_count, o_count, (size_t)nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
        hipStreamSynchronize(s->stream) != hipSuccess) { rc = SS_EDEVICE; break; }
    bool ovf = false;
    for (uint32_t i = 0; i < nq; i++) ovf |= h_count[i] == 0xFFFFFFFFu;
    if (!ovf) break;
    if (attempt == 1) { rc = SS_EDEVICE; break; }  // cannot overflow in safe mode
  }
  return rc;
}

// A COALESCED vector batch (AnnMode::All; f32 or i8 rows): queries from, and answers into, the lane's PINNED buffers; its own stream and
// staging (ss_common.h vmu / vstream).  The shard mutex is held while the pass is enqueued -- the scan buffers of the stream are bound
// into the shard's fields for that long (VecWsBind) -- and NOT while it runs: 9 ms at 10 M x 768, during which lexical searches of the
// shard (a hybrid caller's first half) used to wait for the mutex, then for the stream.
static int vec_search_host_lane(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k, float thr,
                                uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total) {
  if (nq == 0) return SS_OK;
  std::lock_guard<std::mutex> gv(s->vmu);
  for (int attempt = 0; attempt < 2; attempt++) {
    {
      std::lock_guard<std::mutex> g(s->mu);  // (a vector pass neither writes nor reads what the lexical lanes' kernels touch: no drain)
      SS_HIP(hipSetDevice(s->device));
      if ((elem == sizeof(float)) ? !s->d_X : !s->d_X8) return SS_ESTATE;
      const size_t qbytes = ((size_t)nq * s->dim * elem + 15) & ~(size_t)15, sbytes = query_scale ? (size_t)nq * sizeof(float) : 0;
      if (qbytes + sbytes > s->vq_cap) {  // (vstream is idle: vmu is ours and the batch before synchronised it)
        if (s->d_vq) (void)hipFree(s->d_vq);
        s->d_vq = nullptr; s->vq_cap = 0;
        SS_HIP(hipMalloc(&s->d_vq, (qbytes + sbytes) * 2));
        s->vq_cap = (qbytes + sbytes) * 2;
      }
      if ((size_t)nq * k > s->vout_cap || nq > s->vq_rows_cap) {
        for (void* p_ : {(void*)s->d_vdoc, (void*)s->d_vscore, (void*)s->d_vcount, (void*)s->d_vtotal}) if (p_) (void)hipFree(p_);
        s->d_vdoc = nullptr; s->d_vscore = nullptr; s->d_vcount = nullptr; s->d_vtotal = nullptr; s->vout_cap = 0; s->vq_rows_cap = 0;
        const size_t rows = std::max<size_t>(nq, SS_VEC_BATCH), cells = std::max<size_t>((size_t)nq * k, (size_t)SS_VEC_BATCH * k);
        SS_HIP(hipMalloc(&s->d_vdoc, cells * sizeof(uint32_t)));
        SS_HIP(hipMalloc(&s->d_vscore, cells * sizeof(float)));
        SS_HIP(hipMalloc(&s->d_vcount, rows * sizeof(uint32_t)));
        SS_HIP(hipMalloc(&s->d_vtotal, rows * sizeof(uint64_t)));
        s->vout_cap = cells; s->vq_rows_cap = rows;
      }
      float* d_qs = query_scale ? (float*)((char*)s->d_vq + qbytes) : nullptr;
      SS_HIP(hipMemcpyAsync(s->d_vq, queries, (size_t)nq * s->dim * elem, hipMemcpyHostToDevice, s->vstream));
      if (d_qs) SS_HIP(hipMemcpyAsync(d_qs, query_scale, sbytes, hipMemcpyHostToDevice, s->vstream));
      {
        VecWsBind bind(s, s->vstream);
        SS_TRY(ssi_vec_search(s, nq, s->d_vq, d_qs, k, thr, s->d_vdoc, s->d_vscore, s->d_vcount, s->d_vtotal, s->vstream, attempt == 1, nullptr, nullptr, nullptr));
      }
      SS_HIP(hipMemcpyAsync(out_count, s->d_vcount, (size_t)nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s->vstream));
      SS_HIP(hipMemcpyAsync(out_doc, s->d_vdoc, (size_t)nq * k * sizeof(uint32_t), hipMemcpyDeviceToHost, s->vstream));
      SS_HIP(hipMemcpyAsync(out_score, s->d_vscore, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, s->vstream));
      SS_HIP(hipMemcpyAsync(out_total, s->d_vtotal, (size_t)nq * sizeof(uint64_t), hipMemcpyDeviceToHost, s->vstream));
      if (!s->vev) SS_HIP(hipEventCreateWithFlags(&s->vev, hipEventDisableTiming | hipEventBlockingSync));
      SS_HIP(hipEventRecord(s->vev, s->vstream));
    }
    // the pass itself: nobody waits for us but this batch's callers.  The leader SLEEPS on a blocking-sync event: hipStreamSynchronize
    // spins, and a thread that spins through 9 ms on a box whose 64 callers share 16 CPUs uses up its time slice and is descheduled just
    // when the pass ends -- its whole batch then comes home a scheduling quantum late (one such batch per second or two was the p99 of the
    // T = 64 vector and hybrid callers: 17.8 / 17.0 ms against a p50 of 9.6 / 10.8, gpurun_out/r6_bench_b.json)
    SS_HIP(hipEventSynchronize(s->vev));
    bool ovf = false;
    for (uint32_t i = 0; i < nq; i++) ovf |= out_count[i] == 0xFFFFFFFFu;
    if (!ovf) return SS_OK;  // (an adversarial row order overflowed the candidate slots: once more in safe mode)
  }
  return SS_EDEVICE;  // cannot overflow in safe mode
}

// host-pointer searches: queries (f32 or i8 rows) and scales are staged in the shard's grow-only buffer, out_clusters
// (observed_cluster_count, ANN modes) rides behind them
static int vec_search_host(ss_shard* s, uint32_t nq, const void* queries, size_t elem, const float* query_scale, uint32_t k,
                           float thr, const ss_ann_mode* mode, uint32_t* out_doc, float* out_score, uint32_t* out_count,
                           uint64_t* out_total, uint32_t* out_clusters, const float* query_norm) {
  if (nq == 0) return SS_OK;
  ShardLock g(s);
  uint32_t* d_ncl = nullptr;
  int rc = vec_search_host_lists(s, nq, queries, elem, query_scale, k, thr, mode, out_count, &d_ncl, query_norm, out_clusters != nullptr);
  if (rc == SS_OK) {
    if (hipMemcpy(out_doc, s->d_out_doc, (size_t)nq * k * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(out_score, s->d_out_score, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(out_total, s->d_out_total, (size_t)nq * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess ||
        (d_ncl && hipMemcpy(out_clusters, d_ncl, (size_t)nq * ((mode && (mode->flags & SS_ANN_REPORT_OBSERVED)) ? 3 : 1) * sizeof(uint32_t),
                            hipMemcpyDeviceToHost) != hipSuccess))
      rc = SS_EDEVICE;
  }
  return rc;
}
static bool ann_skips_clusters(const ss_ann_mode* m) { return m->n_probe != 0 || m->cluster_threshold_raw > -3.4028234663852886e38f; }
// a mode that neither skips clusters nor filters fields is AnnMode::All
static const ss_ann_mode* ann_effective(const ss_ann_mode* m) {
  return (m && (ann_skips_clusters(m) || m->field_mask || (m->flags & SS_ANN_REPORT_OBSERVED))) ? m : nullptr;
}
static int ann_mode_ok(const ss_shard* s, const ss_ann_mode* mode) {
  if (!mode) return SS_OK;
  if (mode->cluster_threshold_raw != mode->cluster_threshold_raw) return SS_EINVAL;
  if (ann_skips_clusters(mode) && !s->d_row_cluster) return SS_ESTATE;
  if (mode->field_mask && !s->d_row_field) return SS_ESTATE;
  return SS_OK;
}

int ss_vec_search_ann(ss_shard* s, uint32_t nq, const float* queries, uint32_t k, float thr, const ss_ann_mode* mode,
                      uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total, uint32_t* out_clusters) {
  if (!s || !queries || !out_doc || !out_score || !out_count || !out_total) return SS_EINVAL;
  if (k == 0) return SS_EINVAL;
  if (!s->d_X) return SS_ESTATE;
  SS_TRY(ann_mode_ok(s, mode));
  mode = ann_effective(mode);
  if (k > SS_MAX_K)  // (a page deeper than SS_MAX_K results: passes under per-query exclusion bitmaps, "deep pages" above)
    return vec_search_host_deep(s, nq, queries, sizeof(float), nullptr, k, thr, mode, out_doc, out_score, out_count, out_total, mode ? out_clusters : nullptr, nullptr);
  if (!mode && nq != 0 && nq <= SS_COALESCE_MAX_REQUEST && s->co_vec.max_batch) {  // AnnMode::All from concurrent callers: one pass serves them
    ss_co_req r;
    r.q = queries; r.nq = nq; r.k = k; r.elem = (uint32_t)sizeof(float); r.thr = thr;
    r.out_doc = out_doc; r.out_score = out_score; r.out_count = out_count; r.out_total = out_total;
    return co_submit(s, s->co_vec, false, &r);
  }
  return vec_search_host(s, nq, queries, sizeof(float), nullptr, k, thr, mode, out_doc, out_score, out_count, out_total,
                         mode ? out_clusters : nullptr);
}
int ss_vec_search(ss_shard* s, uint32_t nq, const float* queries, uint32_t k, float thr, uint32_t* out_doc,
                  float* out_score, uint32_t* out_count, uint64_t* out_total) {
  return ss_vec_search_ann(s, nq, queries, k, thr, nullptr, out_doc, out_score, out_count, out_total, nullptr);
}

// a DEEP page on a device-pointer vector entry: the passes are steered from the host, so the queries (and their scales / norms) make one
// trip there and the call is synchronous; the page is left in the caller's device arrays, ordered on `st`.  Caller holds the shard lock.
static int vec_search_deep_dev_locked(ss_shard* s, uint32_t nq, const void* d_queries, size_t elem, const float* d_qscale, const float* d_qnorm, uint32_t k, float thr,
                                      const ss_ann_mode* mode, uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total,
                                      uint32_t* d_out_clusters, hipStream_t st) {
  if (nq == 0) return SS_OK;
  const uint32_t ocw = (mode && (mode->flags & SS_ANN_REPORT_OBSERVED)) ? 3u : 1u;
  std::vector<char> hq((size_t)nq * s->dim * elem);
  std::vector<float> hs(d_qscale ? nq : 0), hn(d_qnorm ? nq : 0);
  SS_HIP(hipMemcpyAsync(hq.data(), d_queries, hq.size(), hipMemcpyDeviceToHost, st));
  if (d_qscale) SS_HIP(hipMemcpyAsync(hs.data(), d_qscale, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
  if (d_qnorm) SS_HIP(hipMemcpyAsync(hn.data(), d_qnorm, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  std::vector<uint32_t> h_doc((size_t)nq * k), h_cnt(nq), h_ncl(d_out_clusters ? (size_t)nq * ocw : 0);
  std::vector<float> h_score((size_t)nq * k);
  std::vector<uint64_t> h_tot(nq);
  SS_TRY(vec_search_deep_locked(s, nq, hq.data(), elem, d_qscale ? hs.data() : nullptr, k, thr, mode, h_doc.data(), h_score.data(), h_cnt.data(), h_tot.data(),
                                d_out_clusters ? h_ncl.data() : nullptr, d_qnorm ? hn.data() : nullptr));
  SS_HIP(hipMemcpyAsync(d_out_doc, h_doc.data(), h_doc.size() * 4, hipMemcpyHostToDevice, st));
  SS_HIP(hipMemcpyAsync(d_out_score, h_score.data(), h_score.size() * 4, hipMemcpyHostToDevice, st));
  SS_HIP(hipMemcpyAsync(d_out_count, h_cnt.data(), (size_t)nq * 4, hipMemcpyHostToDevice, st));
  SS_HIP(hipMemcpyAsync(d_out_total, h_tot.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
  if (d_out_clusters) SS_HIP(hipMemcpyAsync(d_out_clusters, h_ncl.data(), h_ncl.size() * 4, hipMemcpyHostToDevice, st));
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

int ss_vec_search_ann_dev(ss_shard* s, uint32_t nq, const float* d_queries, uint32_t k, float thr, const ss_ann_mode* mode,
                          uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total,
                          uint32_t* d_out_clusters, void* stream) {
  if (!s || !d_queries || !d_out_doc || !d_out_score || !d_out_count || !d_out_total) return SS_EINVAL;
  if (k == 0) return SS_EINVAL;
  if (!s->d_X) return SS_ESTATE;
  SS_TRY(ann_mode_ok(s, mode));
  mode = ann_effective(mode);
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  hipStream_t st = stream ? (hipStream_t)stream : s->stream;
  if (k > SS_MAX_K)
    return vec_search_deep_dev_locked(s, nq, d_queries, sizeof(float), nullptr, nullptr, k, thr, mode, d_out_doc, d_out_score, d_out_count, d_out_total,
                                      mode ? d_out_clusters : nullptr, st);
  VecWsBind bind(s, st);
  AnnStateGuard ann_guard(s, st, mode);
  return ssi_vec_search(s, nq, d_queries, nullptr, k, thr, d_out_doc, d_out_score, d_out_count, d_out_total, st, false, mode,
                        mode ? d_out_clusters : nullptr);
}
int ss_vec_search_dev(ss_shard* s, uint32_t nq, const float* d_queries, uint32_t k, float thr, uint32_t* d_out_doc,
                      float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, void* stream) {
  return ss_vec_search_ann_dev(s, nq, d_queries, k, thr, nullptr, d_out_doc, d_out_score, d_out_count, d_out_total, nullptr, stream);
}

// ---- VectorSimilarity of the image (vector_similarity.rs:118-345): must precede the upload (layout of the f32 image)
int ss_vec_set_similarity(ss_shard* s, int similarity) {
  if (!s || (similarity != SS_SIM_DOT && similarity != SS_SIM_EUCLIDEAN)) return SS_EINVAL;
  ShardLock g(s);
  if ((s->d_X || s->d_X8) && similarity != s->vec_similarity) return SS_ESTATE;
  s->vec_similarity = similarity;
  return SS_OK;
}
int ss_vec_set_row_norms(ss_shard* s, uint64_t n_rows, const float* row_norm) {
  if (!s || !row_norm) return SS_EINVAL;
  ShardLock g(s);
  if (s->vstream) (void)hipStreamSynchronize(s->vstream);  // (a coalesced scan in flight reads what this call replaces; none starts while mu is ours)
  SS_HIP(hipSetDevice(s->device));
  if (!s->d_X8) return SS_ESTATE;
  if (n_rows != s->n_rows) return SS_EINVAL;
  SS_HIP(hipStreamSynchronize(s->stream));
  if (!s->d_row_norm) SS_HIP(hipMalloc(&s->d_row_norm, std::max<uint64_t>(n_rows, s->vec_rows_cap) * sizeof(float)));
  SS_HIP(hipMemcpy(s->d_row_norm, row_norm, n_rows * sizeof(float), hipMemcpyHostToDevice));
  return SS_OK;
}

// ---- cluster structure of the vector image (ANN modes, vec_ann.hip)
int ss_vec_set_clusters(ss_shard* s, uint32_t n_levels, const uint32_t* level_clusters, const uint32_t* child_count) {
  if (!s) return SS_EINVAL;
  ShardLock g(s);
  if (s->vstream) (void)hipStreamSynchronize(s->vstream);  // (a coalesced scan in flight reads what this call replaces; none starts while mu is ours)
  SS_HIP(hipSetDevice(s->device));
  return ssi_vec_set_clusters(s, n_levels, level_clusters, child_count);
}
int ss_vec_set_fields(ss_shard* s, uint64_t n_rows, const uint16_t* row_field) {
  if (!s || !row_field) return SS_EINVAL;
  ShardLock g(s);
  if (s->vstream) (void)hipStreamSynchronize(s->vstream);  // (a coalesced scan in flight reads what this call replaces; none starts while mu is ours)
  SS_HIP(hipSetDevice(s->device));
  if (!s->d_X && !s->d_X8) return SS_ESTATE;
  if (n_rows != s->n_rows) return SS_EINVAL;
  SS_HIP(hipStreamSynchronize(s->stream));
  if (!s->d_row_field) SS_HIP(hipMalloc(&s->d_row_field, std::max<uint64_t>(n_rows, s->vec_rows_cap) * sizeof(uint16_t)));
  SS_HIP(hipMemcpy(s->d_row_field, row_field, n_rows * sizeof(uint16_t), hipMemcpyHostToDevice));
  return SS_OK;
}
int ss_vec_cluster_info(ss_shard* s, uint32_t* n_levels, uint32_t* n_clusters) {
  if (!s) return SS_EINVAL;
  // no lock: two words that only an image (re)build changes, read by every vector search of the host mirrors -- behind s->mu each
  // such read queued up with the batch in flight (64 concurrent callers: mean merged batch 1.3, p99 0.6 s; bench round 3)
  if (n_levels) *n_levels = s->vec_n_levels;
  if (n_clusters) *n_clusters = s->vec_n_clusters;
  return SS_OK;
}

// ------------------------------------------------------------------ i8 (quantised) vectors
// i8 image: rows as the reference stores them for Precision::I8 (quantize_f32_to_i8, vector_similarity.rs:1226-1232, or a
// per-record scale = VectorHeader.scale with ScalarQuantizationI8).  Score = dot_i8 as f32 (vector_similarity.rs:1011-1016)
// or, with scales, dot_i8_quantized = dot as f32 * query_scale * embedding_scale (1754-1758).
static int vec8_alloc(ss_shard* s, uint64_t n_rows, uint32_t dim) {
  free_vec(s);
  s->n_rows = n_rows;
  s->dim = dim;
  s->dim_pad8 = (dim + 127u) / 128u * 128u;
  if (s->dim_pad8 > 2560u) { free_vec(s); return SS_ENOTSUP; }  // the 64 queries stay in LDS: 64 x dim_pad8 <= 160 KB
  s->n_rows_pad = (n_rows + VS_TR - 1) / VS_TR * VS_TR;
  const size_t bytes = (size_t)s->n_rows_pad * s->dim_pad8;
  SS_HIP(hipMalloc(&s->d_X8, bytes));
  if (s->dim_pad8 != dim || s->n_rows_pad != n_rows) SS_HIP(hipMemsetAsync(s->d_X8, 0, bytes, s->stream));  // padding is interleaved
  return SS_OK;
}

int ss_vec_upload_i8(ss_shard* s, uint64_t n_rows, uint32_t dim, const int8_t* rows, const float* row_scale,
                     const uint32_t* row_doc_ids) {
  if (!s || !rows || n_rows == 0 || dim == 0) return SS_EINVAL;
  if (n_rows > 0xFFFFFFFEull) return SS_ENOTSUP;
  bool multi = false;
  if (row_doc_ids) {
    std::vector<uint32_t> tmp(row_doc_ids, row_doc_ids + n_rows);
    std::sort(tmp.begin(), tmp.end());
    multi = std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end();
    if (!tmp.empty() && tmp.back() == SS_NO_DOC) return SS_EINVAL;
  }
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  int rc = vec8_alloc(s, n_rows, dim);
  if (rc) { free_vec(s); return rc; }
  {  // row-major staging on the device, then into fragment order
    int8_t* stage = nullptr;
    SS_HIP(hipMalloc(&stage, (size_t)n_rows * dim));
    if (hipMemcpyAsync(stage, rows, (size_t)n_rows * dim, hipMemcpyHostToDevice, s->stream) != hipSuccess) { (void)hipFree(stage); free_vec(s); return SS_EDEVICE; }
    rc = ssi_vec8_permute(s, stage, s->stream);
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(stage);
    if (rc) { free_vec(s); return rc; }
  }
  s->vec_multi_record = multi;
  if (row_scale) {
    SS_HIP(hipMalloc(&s->d_row_scale, n_rows * sizeof(float)));
    SS_HIP(hipMemcpyAsync(s->d_row_scale, row_scale, n_rows * sizeof(float), hipMemcpyHostToDevice, s->stream));
  }
  if (row_doc_ids) {
    SS_HIP(hipMalloc(&s->d_row_doc, n_rows * sizeof(uint32_t)));
    SS_HIP(hipMemcpyAsync(s->d_row_doc, row_doc_ids, n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
  }
  SS_HIP(hipStreamSynchronize(s->stream));
  return vec_finish(s);
}

// ------------------------------------------------------------------ append of one committed level of vector records
// The reference commits 65 536 docs at a time and writes the level's clusters and records behind the earlier ones
// (vector.rs:1066-1094); a rebuild of the whole image per commit would cost O(shard).  Rows are independent, so the new records are
// written behind the image's rows IN PLACE (f32: a strided copy; i8: through a row-major staging and the fragment-order scatter of the
// rows' range), the Euclidean side data is computed for the new range only, the per-row side arrays (doc ids, scales, norms, field
// ids) are extended, and the cluster structure -- if the image has one -- is declared again with the level added (O(rows) words, not
// O(image bytes)).  The image and the side arrays grow by half when their room is used up (one device-to-device copy, amortised).
// room for new_cap rows in the image and every per-row array it carries (caller holds s->mu, the device is idle).  ALL OR NOTHING:
// every new array is allocated before anything is copied or swapped, so a failure (SS_ENOMEM) leaves the shard as it was.
// grow_image = false: the image already has the room (its row padding), only the side arrays follow.
static int vec_grow(ss_shard* s, uint64_t new_cap, bool grow_image = true) {
  const bool i8 = s->d_X8 != nullptr;
  const size_t row_bytes = i8 ? (size_t)s->dim_pad8 : (size_t)s->dim_pad * sizeof(float);
  const uint64_t old_n = s->n_rows;
  struct G { void** p; size_t elem; uint64_t old_rows; bool zero_tail; void* q; };
  // the image: rows [0, n_rows_pad) are data + zero padding, the room behind them starts as padding (zero)
  G gs[] = {{i8 ? (void**)&s->d_X8 : (void**)&s->d_X, row_bytes, s->n_rows_pad, true, nullptr},
            {(void**)&s->d_row_doc, sizeof(uint32_t), old_n, false, nullptr},   {(void**)&s->d_row_scale, sizeof(float), old_n, false, nullptr},
            {(void**)&s->d_row_norm, sizeof(float), old_n, false, nullptr},     {(void**)&s->d_row_sq, sizeof(int32_t), old_n, false, nullptr},
            {(void**)&s->d_row_field, sizeof(uint16_t), old_n, false, nullptr}};
  int rc = SS_OK;
  for (size_t i = grow_image ? 0 : 1; i < sizeof(gs) / sizeof(gs[0]) && rc == SS_OK; i++) {
    G& g = gs[i];
    if (!*g.p) continue;
    const hipError_t e = hipMalloc(&g.q, (size_t)new_cap * g.elem);
    if (e != hipSuccess) { g.q = nullptr; rc = e == hipErrorOutOfMemory ? SS_ENOMEM : SS_EDEVICE; break; }
    if (hipMemcpy(g.q, *g.p, (size_t)g.old_rows * g.elem, hipMemcpyDeviceToDevice) != hipSuccess ||
        (g.zero_tail && hipMemset((char*)g.q + (size_t)g.old_rows * g.elem, 0, (size_t)(new_cap - g.old_rows) * g.elem) != hipSuccess))
      rc = SS_EDEVICE;
  }
  if (rc != SS_OK) {
    for (G& g : gs) if (g.q) (void)hipFree(g.q);
    return rc;
  }
  for (G& g : gs)
    if (g.q) { (void)hipFree(*g.p); *g.p = g.q; }
  s->vec_rows_cap = new_cap;
  return SS_OK;
}

int ss_vec_reserve_rows(ss_shard* s, uint64_t n_rows_cap) {
  if (!s) return SS_EINVAL;
  ShardLock g(s);
  if (s->vstream) (void)hipStreamSynchronize(s->vstream);  // (a coalesced scan in flight reads what this call replaces; none starts while mu is ours)
  SS_HIP(hipSetDevice(s->device));
  if (!s->d_X && !s->d_X8) return SS_ESTATE;
  if (n_rows_cap > 0xFFFFFFFEull) return SS_ENOTSUP;
  const uint64_t cap = s->vec_rows_cap ? s->vec_rows_cap : s->n_rows_pad, want = (n_rows_cap + VS_TR - 1) / VS_TR * VS_TR;
  if (want <= cap && s->vec_rows_cap) return SS_OK;
  SS_HIP(hipDeviceSynchronize());
  return vec_grow(s, std::max(want, cap));
}

int ss_vec_append_rows(ss_shard* s, const ss_vec_level* lv) {
  if (!s || !lv || !lv->rows || lv->n_rows == 0) return SS_EINVAL;
  const uint64_t n_new = lv->n_rows;
  bool multi = false;
  if (lv->row_doc_ids) {  // (a level's records share doc ids only among themselves: (level << 16) | doc_id, vector.rs:1448)
    std::vector<uint32_t> tmp(lv->row_doc_ids, lv->row_doc_ids + n_new);
    std::sort(tmp.begin(), tmp.end());
    multi = std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end();
    if (tmp.back() == SS_NO_DOC) return SS_EINVAL;
  }
  ShardLock g(s);
  if (s->vstream) (void)hipStreamSynchronize(s->vstream);  // (a coalesced scan in flight reads what this call replaces; none starts while mu is ours)
  SS_HIP(hipSetDevice(s->device));
  if (!s->d_X && !s->d_X8) return SS_ESTATE;
  const bool i8 = s->d_X8 != nullptr;
  if ((lv->elem_i8 != 0) != i8) return SS_EINVAL;
  // what the image carries per row, the level must bring -- and nothing else
  if ((s->d_row_doc != nullptr) != (lv->row_doc_ids != nullptr) || (s->d_row_scale != nullptr) != (lv->row_scale != nullptr) ||
      (s->d_row_norm != nullptr) != (lv->row_norm != nullptr) || (s->d_row_field != nullptr) != (lv->row_field != nullptr) ||
      (s->vec_n_clusters != 0) != (lv->n_clusters != 0))
    return SS_EINVAL;
  if (lv->n_clusters) {
    if (!lv->child_count) return SS_EINVAL;
    uint64_t sum = 0;
    for (uint32_t c = 0; c < lv->n_clusters; c++) { if (lv->child_count[c] == 0) return SS_ENOTSUP; sum += lv->child_count[c]; }
    if (sum != n_new) return SS_EINVAL;
  }
  const uint64_t old_n = s->n_rows, new_n = old_n + n_new;
  if (new_n > 0xFFFFFFFEull) return SS_ENOTSUP;
  const uint64_t new_pad = (new_n + VS_TR - 1) / VS_TR * VS_TR;
  SS_HIP(hipDeviceSynchronize());  // searches on the callers' own streams read the arrays that are written / replaced here
  const uint64_t cap = s->vec_rows_cap ? s->vec_rows_cap : s->n_rows_pad;
  // (an opener that expects commits reserves the room once: ss_vec_reserve_rows).  Otherwise: what is needed plus a bounded headroom
  // -- half of what there is, at most 4 M rows: old and new image are both live during the copy -- and on the FIRST append only the side
  // arrays (allocated for exactly n_rows) when the level still fits the image's own row padding
  if (new_pad > cap)
    SS_TRY(vec_grow(s, (std::max<uint64_t>(new_pad, std::min<uint64_t>(cap + cap / 2, new_pad + (4u << 20))) + VS_TR - 1) / VS_TR * VS_TR));
  else if (!s->vec_rows_cap)
    SS_TRY(vec_grow(s, cap, /*grow_image=*/false));
  if (!i8) {
    SS_HIP(hipMemcpy2DAsync(s->d_X + (size_t)old_n * s->dim_pad, (size_t)s->dim_pad * sizeof(float), lv->rows, (size_t)s->dim * sizeof(float),
                            (size_t)s->dim * sizeof(float), n_new, hipMemcpyHostToDevice, s->stream));
  } else {
    int8_t* stage = nullptr;
    SS_HIP(hipMalloc(&stage, (size_t)n_new * s->dim));
    int rc = hipMemcpyAsync(stage, lv->rows, (size_t)n_new * s->dim, hipMemcpyHostToDevice, s->stream) == hipSuccess ? SS_OK : SS_EDEVICE;
    if (rc == SS_OK) rc = ssi_vec8_permute_range(s, stage, old_n, n_new, s->stream);
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(stage);
    if (rc) return rc;
  }
  if (lv->row_doc_ids) SS_HIP(hipMemcpyAsync(s->d_row_doc + old_n, lv->row_doc_ids, n_new * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
  if (lv->row_scale) SS_HIP(hipMemcpyAsync(s->d_row_scale + old_n, lv->row_scale, n_new * sizeof(float), hipMemcpyHostToDevice, s->stream));
  if (lv->row_norm) SS_HIP(hipMemcpyAsync(s->d_row_norm + old_n, lv->row_norm, n_new * sizeof(float), hipMemcpyHostToDevice, s->stream));
  if (lv->row_field) SS_HIP(hipMemcpyAsync(s->d_row_field + old_n, lv->row_field, n_new * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
  s->n_rows = new_n;
  s->n_rows_pad = new_pad;
  s->vec_multi_record = s->vec_multi_record || multi;
  if (s->vec_similarity == SS_SIM_EUCLIDEAN) SS_TRY(s->d_X ? ssi_vec_augment(s, s->stream, old_n) : ssi_vec8_row_sq(s, s->stream, old_n));
  SS_HIP(hipStreamSynchronize(s->stream));
  if (lv->n_clusters) {  // the structure with the level added, declared again (row -> cluster words, the new medoids, tile lists)
    std::vector<uint32_t> lc = s->h_level_clusters, cc = s->h_child_count;
    lc.push_back(lv->n_clusters);
    cc.insert(cc.end(), lv->child_count, lv->child_count + lv->n_clusters);
    SS_TRY(ssi_vec_set_clusters(s, (uint32_t)lc.size(), lc.data(), cc.data()));
  }
  return SS_OK;
}

// bench / test utility: the synthetic f32 corpus of ss_vec_synth quantised on the device with quantize_f32_to_i8
int ss_vec_synth_i8(ss_shard* s, uint64_t seed, uint64_t n_rows, uint32_t dim) {
  if (!s || n_rows == 0 || dim == 0) return SS_EINVAL;
  if (n_rows > 0xFFFFFFFEull) return SS_ENOTSUP;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  int rc = vec_alloc(s, n_rows, dim);  // f32 rows first ...
  if (rc) { free_vec(s); return rc; }
  SS_TRY(ssi_vec_synth(s, seed, s->stream));
  float* x32 = s->d_X;
  const uint32_t pad32 = s->dim_pad;
  s->d_X = nullptr;  // ... kept aside while the i8 image is allocated (free_vec must not release them)
  rc = vec8_alloc(s, n_rows, dim);
  if (rc) { (void)hipFree(x32); return rc; }
  s->d_X = x32;
  s->dim_pad = pad32;
  rc = ssi_vec8_quantize(s, s->stream);
  (void)hipStreamSynchronize(s->stream);
  (void)hipFree(x32);
  s->d_X = nullptr;
  s->dim_pad = 0;
  if (rc) { free_vec(s); return rc; }
  return vec_finish(s);
}

int ss_vec_read_rows_i8(ss_shard* s, uint64_t r0, uint64_t n, int8_t* out) {
  if (!s || !out) return SS_EINVAL;
  if (!s->d_X8) return SS_ESTATE;
  if (r0 + n > s->n_rows) return SS_EINVAL;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  SS_HIP(hipStreamSynchronize(s->stream));
  if (n == 0) return SS_OK;
  int8_t* tmp = nullptr;
  SS_HIP(hipMalloc(&tmp, (size_t)n * s->dim));
  int rc = ssi_vec8_gather_rows(s, r0, n, tmp, s->stream);
  if (rc == SS_OK && hipMemcpyAsync(out, tmp, (size_t)n * s->dim, hipMemcpyDeviceToHost, s->stream) != hipSuccess) rc = SS_EDEVICE;
  (void)hipStreamSynchronize(s->stream);
  (void)hipFree(tmp);
  return rc;
}

// euclidean_i8_quantized (scales present) = max(0, n1 + n2 - 2 dot s1 s2) needs BOTH norms: the reference always carries
// VectorHeader.norm / QuantizedVector.norm.  Without them the ranking would silently be that of -max(0, -2 dot s1 s2).
static int vec8_euclid_norms_ok(const ss_shard* s, bool have_qscale, bool have_qnorm) {
  if (s->vec_similarity != SS_SIM_EUCLIDEAN || (!s->d_row_scale && !have_qscale)) return SS_OK;
  if (!s->d_row_norm) return SS_ESTATE;  // ss_vec_set_row_norms
  return have_qnorm ? SS_OK : SS_EINVAL;
}
int ss_vec_search_i8_euclid(ss_shard* s, uint32_t nq, const int8_t* queries, const float* query_scale, const float* query_norm,
                            uint32_t k, float thr, const ss_ann_mode* mode, uint32_t* out_doc, float* out_score, uint32_t* out_count,
                            uint64_t* out_total, uint32_t* out_clusters) {
  if (!s || !queries || !out_doc || !out_score || !out_count || !out_total) return SS_EINVAL;
  if (k == 0) return SS_EINVAL;
  if (!s->d_X8) return SS_ESTATE;
  SS_TRY(vec8_euclid_norms_ok(s, query_scale != nullptr, query_norm != nullptr));
  SS_TRY(ann_mode_ok(s, mode));
  mode = ann_effective(mode);
  if (k > SS_MAX_K)  // (deep pages)
    return vec_search_host_deep(s, nq, queries, 1, query_scale, k, thr, mode, out_doc, out_score, out_count, out_total, mode ? out_clusters : nullptr, query_norm);
  if (!mode && !query_norm && nq != 0 && nq <= SS_COALESCE_MAX_REQUEST && s->co_vec.max_batch) {
    ss_co_req r;
    r.q = queries; r.qscale = query_scale; r.nq = nq; r.k = k; r.elem = 1; r.thr = thr;
    r.out_doc = out_doc; r.out_score = out_score; r.out_count = out_count; r.out_total = out_total;
    return co_submit(s, s->co_vec, false, &r);
  }
  return vec_search_host(s, nq, queries, 1, query_scale, k, thr, mode, out_doc, out_score, out_count, out_total,
                         mode ? out_clusters : nullptr, query_norm);
}
int ss_vec_search_i8_ann(ss_shard* s, uint32_t nq, const int8_t* queries, const float* query_scale, uint32_t k, float thr,
                         const ss_ann_mode* mode, uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total,
                         uint32_t* out_clusters) {
  return ss_vec_search_i8_euclid(s, nq, queries, query_scale, nullptr, k, thr, mode, out_doc, out_score, out_count, out_total, out_clusters);
}
int ss_vec_search_i8(ss_shard* s, uint32_t nq, const int8_t* queries, const float* query_scale, uint32_t k, float thr,
                     uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total) {
  return ss_vec_search_i8_ann(s, nq, queries, query_scale, k, thr, nullptr, out_doc, out_score, out_count, out_total, nullptr);
}

int ss_vec_search_i8_ann_dev(ss_shard* s, uint32_t nq, const int8_t* d_queries, const float* d_query_scale, uint32_t k, float thr,
                             const ss_ann_mode* mode, uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count,
                             uint64_t* d_out_total, uint32_t* d_out_clusters, void* stream) {
  return ss_vec_search_i8_euclid_dev(s, nq, d_queries, d_query_scale, nullptr, k, thr, mode, d_out_doc, d_out_score, d_out_count,
                                     d_out_total, d_out_clusters, stream);
}
int ss_vec_search_i8_euclid_dev(ss_shard* s, uint32_t nq, const int8_t* d_queries, const float* d_query_scale, const float* d_query_norm,
                                uint32_t k, float thr, const ss_ann_mode* mode, uint32_t* d_out_doc, float* d_out_score,
                                uint32_t* d_out_count, uint64_t* d_out_total, uint32_t* d_out_clusters, void* stream) {
  if (!s || !d_queries || !d_out_doc || !d_out_score || !d_out_count || !d_out_total) return SS_EINVAL;
  if (k == 0) return SS_EINVAL;
  if (!s->d_X8) return SS_ESTATE;
  SS_TRY(vec8_euclid_norms_ok(s, d_query_scale != nullptr, d_query_norm != nullptr));
  SS_TRY(ann_mode_ok(s, mode));
  mode = ann_effective(mode);
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  hipStream_t st = stream ? (hipStream_t)stream : s->stream;
  if (k > SS_MAX_K)
    return vec_search_deep_dev_locked(s, nq, d_queries, 1, d_query_scale, d_query_norm, k, thr, mode, d_out_doc, d_out_score, d_out_count, d_out_total,
                                      mode ? d_out_clusters : nullptr, st);
  VecWsBind bind(s, st);
  AnnStateGuard ann_guard(s, st, mode);
  return ssi_vec_search(s, nq, d_queries, d_query_scale, k, thr, d_out_doc, d_out_score, d_out_count, d_out_total, st, false, mode,
                        mode ? d_out_clusters : nullptr, d_query_norm);
}
int ss_vec_search_i8_dev(ss_shard* s, uint32_t nq, const int8_t* d_queries, const float* d_query_scale, uint32_t k, float thr,
                         uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, void* stream) {
  return ss_vec_search_i8_ann_dev(s, nq, d_queries, d_query_scale, k, thr, nullptr, d_out_doc, d_out_score, d_out_count,
                                  d_out_total, nullptr, stream);
}

// ------------------------------------------------------------------ cross-shard merge + RRF (search.rs:1875-2119)
int ss_merge_results(int mode, const uint64_t* lex_doc, const float* lex_score, uint32_t n_lex, const uint64_t* vec_doc,
                     const float* vec_score, uint32_t n_vec, uint32_t offset, uint32_t length, uint64_t* out_doc,
                     float* out_score, uint8_t* out_source) {
  if (mode < SS_MODE_LEXICAL || mode > SS_MODE_HYBRID) return SS_EINVAL;
  if ((n_lex && (!lex_doc || !lex_score)) || (n_vec && (!vec_doc || !vec_score))) return SS_EINVAL;
  if (length && (!out_doc || !out_score)) return SS_EINVAL;
  struct R { uint64_t doc; float score; uint8_t src; };
  std::vector<R> res;
  auto by_score_desc = [](const R& a, const R& b) { return a.score > b.score; };
  if (mode == SS_MODE_LEXICAL) {
    for (uint32_t i = 0; i < n_lex; i++) res.push_back({lex_doc[i], lex_score[i], SS_SRC_LEXICAL});
  } else if (mode == SS_MODE_VECTOR) {
    for (uint32_t i = 0; i < n_vec; i++) res.push_back({vec_doc[i], vec_score[i], SS_SRC_VECTOR});
  } else {
    // reciprocal rank fusion, k = 0.6, 0-based ranks over the score-sorted concatenation (search.rs:1962-2035)
    std::vector<R> L, V;
    for (uint32_t i = 0; i < n_lex; i++) L.push_back({lex_doc[i], lex_score[i], SS_SRC_LEXICAL});
    for (uint32_t i = 0; i < n_vec; i++) V.push_back({vec_doc[i], vec_score[i], SS_SRC_VECTOR});
    std::stable_sort(L.begin(), L.end(), by_score_desc);
    std::stable_sort(V.begin(), V.end(), by_score_desc);
    std::unordered_map<uint64_t, size_t> at;
    for (size_t i = 0; i < L.size(); i++) {
      float r = 1.0f / (0.6f + (float)i);
      auto it = at.find(L[i].doc);
      if (it == at.end()) { at.emplace(L[i].doc, res.size()); res.push_back({L[i].doc, r, SS_SRC_LEXICAL}); }
      else res[it->second].score = r;  // AHashMap::insert overwrites
    }
    for (size_t i = 0; i < V.size(); i++) {
      float r = 1.0f / (0.6f + (float)i);
      auto it = at.find(V[i].doc);
      if (it == at.end()) { at.emplace(V[i].doc, res.size()); res.push_back({V[i].doc, r, SS_SRC_VECTOR}); }
      else { res[it->second].score += r; res[it->second].src = SS_SRC_HYBRID; }
    }
    // the reference leaves equal RRF scores in hash order (search.rs:2034); we fix doc id ascending
    std::sort(res.begin(), res.end(), [](const R& a, const R& b) { return a.doc < b.doc; });
  }
  std::stable_sort(res.begin(), res.end(), by_score_desc);  // search.rs:2103-2105
  uint32_t w = 0;
  for (size_t i = offset; i < res.size() && w < length; i++, w++) {
    out_doc[w] = res[i].doc;
    out_score[w] = res[i].score;
    if (out_source) out_source[w] = res[i].src;
  }
  return (int)w;
}

// ------------------------------------------------------------------ measurement hooks
int ss_profile_enable(ss_shard* s, int on) {
  if (!s) return SS_EINVAL;
  ShardLock g(s);
  s->prof.on = on != 0;
  return SS_OK;
}

int ss_profile_read(ss_shard* s, int kernel, uint64_t* launches, double* total_ms, int reset) {
  if (!s || kernel < 0 || kernel > 1) return SS_EINVAL;
  ShardLock g(s);
  SS_HIP(hipSetDevice(s->device));
  for (auto& pr : s->prof.pending[kernel]) {
    SS_HIP(hipEventSynchronize(pr.second));
    float ms = 0.f;
    SS_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
    s->prof.ms[kernel] += ms;
    s->prof.launches[kernel] += 1;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  s->prof.pending[kernel].clear();
  if (launches) *launches = s->prof.launches[kernel];
  if (total_ms) *total_ms = s->prof.ms[kernel];
  if (reset) { s->prof.launches[kernel] = 0; s->prof.ms[kernel] = 0.0; }
  return SS_OK;
}

}  // extern "C"

void ssi_prof_begin(ss_shard* s, int kernel, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1) {
  (void)kernel;
  *e0 = nullptr; *e1 = nullptr;
  if (!s->prof.on) return;
  if (hipEventCreate(e0) != hipSuccess || hipEventCreate(e1) != hipSuccess) { *e0 = *e1 = nullptr; return; }
  (void)hipEventRecord(*e0, st);
}
void ssi_prof_end(ss_shard* s, int kernel, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  if (!e0 || !e1) return;
  (void)hipEventRecord(e1, st);
  s->prof.pending[kernel].emplace_back(e0, e1);
}
