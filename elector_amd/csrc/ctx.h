// elector_amd/csrc/ctx.h -- the context object behind the C ABI (one per GPU):
// stream, constant tables, grow-on-demand device workspace, error text, timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "elector_poa.h"
#include "poa_device.h"

namespace elector {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes)
  {
    if (bytes <= cap) return 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 4096;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return ELECTOR_E_NOMEM; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct HostPinned {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes)
  {
    if (bytes <= cap) return 0;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return ELECTOR_E_NOMEM; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct TimedSpan { hipEvent_t a, b; int kind; };

// one merge + statistics job of the device pipeline (elector_msa_stats_enqueue / _collect)
// the rows' copy on the DMA engine through the HSA runtime (rows_dma.cpp)
bool rows_dma_ready(int hip_device, uint64_t *sig);
int rows_dma_start(int hip_device, void *dst, const void *src, size_t n, uint64_t *sig);
int rows_dma_wait(uint64_t sig);
void rows_dma_release(uint64_t *sig);

struct StatsSlot {
  DevBuf rows, rowoff, cols, first, clips, cnt, mask, woff;
  DevBuf dense, outoff;         // elector_msa_stats_enqueue_rows: the merged rows packed (host destinations), their offsets
  DevBuf wcnt, wpiece;          // window-parallel merge: surviving columns per window and their scan, piece of a window
  HostPinned h;                 // [overflow flag, pad to 16][counters][cols][inputs]
  hipEvent_t done = nullptr;
  hipEvent_t rows_done = nullptr;   // the packed rows have arrived at rows_host (recorded on the context's copy stream) ...
  uint64_t rows_sig = 0;            // ... or, the copy started through HSA, its completion signal (rows_by_dma)
  bool rows_by_dma = false;
  uint8_t *rows_host = nullptr;     // this job's rows go to this page-locked host address: from the stream's host function
                                    // behind the packing kernel (rows_go, stats.hip), else when the job is collected
  bool rows_inflight = false;
  const void *rows_src = nullptr;   // where the job's packed rows lie on the device: `dense`, or `rows` itself when the merge wrote them without gaps
  size_t rows_src_cap = 0;
  int device = -1;
  int64_t n_pieces = 0, n_reads = 0, total = 0, last_piece = 0, max_windows = 0;
  bool has_clips = false;
  void release()
  {
    rows.release(); rowoff.release(); cols.release(); first.release(); clips.release(); cnt.release();
    mask.release(); woff.release(); h.release(); dense.release(); outoff.release(); wcnt.release(); wpiece.release();
    if (done) { (void)hipEventDestroy(done); done = nullptr; }
    if (rows_done) { (void)hipEventDestroy(rows_done); rows_done = nullptr; }
    rows_dma_release(&rows_sig);
  }
};

}  // namespace elector

struct elector_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  // fused-kernel classes run concurrently on auxiliary streams (fork/join on `stream`)
  static constexpr int kAux = 4;
  hipStream_t aux[kAux] = {};
  hipEvent_t aux_done[kAux] = {};
  hipEvent_t fork = nullptr;
  bool aux_ready = false;
  std::vector<hipEvent_t> hb_events;   // k_poa -> hand-back launches (one per bin of a batch)
  int chains = 0;               // launch chains for the fused classes (elector_ctx_option "chains"; 0 = default)
  int priority = 0;             // elector_ctx_option "priority": -1 / 0 / +1 = the context's streams at the device's highest / default / lowest priority
  // a stream of this context (every one goes through here: the priority)
  int cu_lo = 0, cu_hi = 0;     // elector_ctx_option "cus": the context's streams on the compute units [cu_lo, cu_hi) of the queue mask
  int make_stream(hipStream_t *s) const
  {
    if (cu_hi > cu_lo) {
      // (bit i of the mask: the runtime deals the bits out to the XCDs in turn, so a range of 8 k bits is k units on each)
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = cu_lo; i < cu_hi && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
      return hipExtStreamCreateWithCUMask(s, 8, mask) == hipSuccess ? 0 : 1;
    }
    if (priority == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess ? 0 : 1;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return 1;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority < 0 ? greatest : least) == hipSuccess ? 0 : 1;
  }
  elector_params params;
  elector::KParams kp;
  bool gen = false;
  std::string err;
  std::mutex mu;
  // constant tables
  elector::DevBuf d_tab, d_linx, d_liny;
  // per-batch workspace
  elector::DevBuf d_off, d_perm, d_mv1, d_mv2, d_sym, d_xinfo, d_ring1, d_map16, d_carry, d_moves,
      d_n1, d_cls, d_score1, d_score2, d_bx2, d_list, d_done, d_rowinit, d_fmv, d_tstate, d_tlist, d_gring,
      d_hand, d_mvpool, d_mvbusy, d_pdesc, d_psym,
      d_far,                             // k_poa's far-edge lists (one region per lane-group size) and their four counters
      d_bin16, d_wkey, d_acc, d_ginfo;   // device-side classification (poa_classify.hip): launch bin and size key per window, per-bin totals
  // device splitter (split_dev.hip)
  elector::DevBuf d_sp_reads, d_sp_off, d_sp_hdr, d_sp_keys, d_sp_vals, d_sp_ca, d_sp_cb, d_sp_wl, d_sp_win, d_sp_first,
      d_sp_cnt, d_sp_wfirst, d_sp_wlen, d_sp_woff, d_sp_scan, d_sp_bases, d_sp_anc;
  // host API staging
  elector::DevBuf d_bases, d_cols, d_ncol, d_status, d_scores, d_rowoff, d_rows;
  // pinned upload staging of the per-batch metadata, double-buffered: a batch's host-to-device copies
  // are queued behind the previous batch's kernels, so the host may already prepare the next batch
  // while they are pending; a buffer is rewritten only after its copies have run (h_meta_done)
  elector::HostPinned h_meta_buf[2];
  elector::HostPinned h_rows;          // merged rows on their way to msa.fa (elector_msa_records_write)
  std::vector<uint8_t> h_text;         // ... and the formatted records
  hipEvent_t h_meta_done[2] = {nullptr, nullptr};
  elector::HostPinned h_acc, h_gen;    // per-bin totals / the generic list's windows on their way back (read after a wait)
  elector::HostPinned h_off;           // a caller's host offsets on their way to the device
  hipEvent_t h_off_done = nullptr;
  int h_meta_cur = 0;
  // statistics workspace
  elector::DevBuf d_st_rows, d_st_rowoff, d_st_cols, d_st_first, d_st_clips, d_st_cnt, d_st_mask, d_st_scr, d_st_dense, d_st_outoff;
  static constexpr int kStatsSlots = 2;
  elector::StatsSlot st_slot[kStatsSlots];
  int st_head = 0, st_tail = 0, st_inflight = 0, st_last = -1;
  uint64_t fetch_sig = 0;              // completion signal of elector_msa_rows_fetch's copy on the DMA engine (rows_dma.cpp)
  hipStream_t copy_stream = nullptr;   // the rows' way to the host: the copy engine works beside the kernels of the next batch
  // timing
  bool timing = false;
  std::vector<elector::TimedSpan> spans;
  static constexpr int kTimedKinds = 7;   // alignment #1 stage, alignment #2 stage, other POA kernels, merge + statistics, k_poa, k_bundle, k_poa far instance
  double ms_acc[kTimedKinds] = {};
  int64_t launches_acc[kTimedKinds] = {};
  int64_t last_n = 0;          // windows of the last POA batch (their offsets stay in d_off)
  int64_t last_total = 0;      // bases of that batch
  // a12: heaviest-bundle consensus (bundle.hip) works on the graph data of the last batch
  bool keep_graph = false;     // POA kernels also leave the x -> y map of alignment #2 in d_map16
  bool graph_valid = false;    // the last batch ran with keep_graph
  // elector_poa_bundles_enqueue notes the search and the context queues it at the next call that waits for the context
  // anyway (statistics collect, sync, the fetch of the bundles, the next batch): the search's two host waits -- scratch
  // size, class counts -- then last what its own first stages last instead of the alignment in front of them, and the
  // thread that feeds several contexts is not held up (option "bundles_now": 1 = queue it inside the call as before)
  bool bundles_pending = false, bundles_now = false;
  int64_t bundles_n = 0;
  float bundles_fraction = 0.0f;
  const int32_t *last_ncol = nullptr;   // device arrays of the last batch (caller's or the staging ones)
  int32_t *last_status = nullptr;
  elector::DevBuf d_bnode, d_bscore, d_bpath, d_bcons, d_binfo, d_bplan, d_bin, d_bcls;
  int64_t last_n_generic = 0;           // windows of the last batch on the generic list (d_perm); the others are in d_list
};

// queues a noted bundle search (bundle.hip); the caller holds c->mu
int elector_bundles_flush(elector_ctx *c);

inline int elector_fail(elector_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
  if (c) {
    c->err = what;
    if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
  }
  return code;
}

// HIP-event brackets around kernel launches, summed per kind by elector_ctx_timing_read
inline void timed_begin(elector_ctx *c, int kind, hipStream_t st)
{
  if (!c->timing) return;
  elector::TimedSpan s;
  s.kind = kind;
  if (hipEventCreate(&s.a) != hipSuccess) return;
  if (hipEventCreate(&s.b) != hipSuccess) { (void)hipEventDestroy(s.a); return; }
  (void)hipEventRecord(s.a, st);
  c->spans.push_back(s);
}
inline void timed_end(elector_ctx *c, hipStream_t st)
{
  // ELECTOR_DEBUG_PROGRESS=<file>: wait for every bracketed launch group and log it (finds a hanging kernel)
  static const char *progress = std::getenv("ELECTOR_DEBUG_PROGRESS");
  if (progress) {
    static int seq = 0;
    FILE *f = std::fopen(progress, "a");
    if (f) { std::fprintf(f, "group %d launched, waiting\n", seq); std::fclose(f); }
    const hipError_t e = hipStreamSynchronize(st);
    f = std::fopen(progress, "a");
    if (f) { std::fprintf(f, "group %d done: %s\n", seq++, hipGetErrorString(e)); std::fclose(f); }
  }
  if (!c->timing || c->spans.empty()) return;
  (void)hipEventRecord(c->spans.back().b, st);
}

#define HIPCHK(ctx, call)                                                        \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) return elector_fail((ctx), ELECTOR_E_HIP, #call, e_);          \
  } while (0)

