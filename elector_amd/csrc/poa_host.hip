// elector_amd/csrc/poa_host.hip -- host side of the C ABI (include/elector_poa.h):
// scoring-parameter parsing (a1), context/workspace management and the batch
// orchestration that replaces one `poa` process (src/poa-graph/main.c:241-287).
// There is no CPU compute path in this file: without a gfx950 device every
// compute entry fails with ELECTOR_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <time.h>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "elector_poa.h"
#include "poa_device.h"
#include "poa_classes.h"

namespace elector {
void launch_symbolize(const uint8_t *in, uint8_t *out, int64_t nbytes, const DevTables *tab, hipStream_t st);
void launch_dp1(const BatchArgs &a, bool gen, hipStream_t st, int nw);
void launch_fuse1(const BatchArgs &a, hipStream_t st);
void launch_dp1_tile(const BatchArgs &a, const TileArgs &ta, bool gen, int ntiles, int nw, hipStream_t st);
void launch_dp2_tile(const BatchArgs &a, const TileArgs &ta, bool gen, int ntiles, int nw, hipStream_t st);
void launch_dp2(const BatchArgs &a, bool gen, int cls, hipStream_t st, int32_t *gring, int64_t gring_block, int blocks, int nw);
void launch_dp2_list(const BatchArgs &a, bool gen, hipStream_t st, int32_t *gring, int64_t gring_block, int blocks, int nw);
void launch_fuse2(const BatchArgs &a, hipStream_t st);
void launch_left_b(const BatchArgs &a, uint32_t *list, int32_t *count, const uint8_t *done_b, int64_t *mv2,
                   unsigned long long *bump, unsigned long long bump_base, unsigned long long bump_cap, int round,
                   int last_round, hipStream_t st);
void launch_rows(const uint8_t *cols, const int64_t *off, const int32_t *ncol, const int64_t *row_off,
                 uint8_t *rows, int64_t n, hipStream_t st);
void launch_trivial(const BatchArgs &a, uint8_t *done_a, uint8_t *triv, uint8_t *pkey, bool flags_only, hipStream_t st);
int partition_buckets();
void launch_partition(const uint32_t *in, uint32_t *out, const int64_t *bins, int nbins, const void *chunks, int nchunks,
                      const int32_t *bin_chunks, const uint8_t *pkey, int32_t *chunk_cnt, int32_t *count, int32_t *count_a1,
                      hipStream_t st);
struct FusedArgs {
  BatchArgs b;
  const uint32_t *list;
  int64_t nlist;
  int slot_bytes;
  uint8_t *done_a;
  uint8_t *done_b;
  int32_t *rowinit;
  int debug;
  int keep_map;
  uint8_t *mv_pool;
  int mv_tw, mv_ns;
  const int32_t *nlist_dev;
  const uint8_t *triv;
  int64_t grid_blocks;
};
void launch_gather(const GatherArgs &a, hipStream_t st);
int launch_poa(const PackArgs &a, int G, int R, hipStream_t st);
int launch_poa_far(const PackArgs &a, int G, hipStream_t st);
bool poa_debug_built();
void launch_poa_pool_init(int32_t *q, int nq, int slots, hipStream_t st);
int launch_fused_a(const FusedArgs &a, int G, int R, hipStream_t st);
int launch_fused_b(const FusedArgs &a, int G, int R, int D, hipStream_t st);
}  // namespace elector

using namespace elector;

#include "ctx.h"

static int fail(elector_ctx *c, int code, const char *what, hipError_t e = hipSuccess) { return elector_fail(c, code, what, e); }

// ------------------------------------------------------------------ probes ---

extern "C" const char *elector_version(void) { return "elector_amd 0.1 (gfx950)"; }

extern "C" const char *elector_strerror(int code)
{
  switch (code) {
    case ELECTOR_OK: return "ok";
    case ELECTOR_E_INVAL: return "invalid argument";
    case ELECTOR_E_NODEVICE: return "no usable gfx950 device (this library has no CPU fallback)";
    case ELECTOR_E_NOMEM: return "out of memory";
    case ELECTOR_E_HIP: return "HIP runtime error";
    case ELECTOR_E_PARAMS: return "scoring parameters outside device limits";
    case ELECTOR_E_IO: return "matrix file unreadable or malformed";
    case ELECTOR_E_WINDOW: return "one or more windows failed (see status[])";
    case ELECTOR_E_LIMIT: return "input beyond an on-chip limit of the device entry (use the host entry)";
    default: return "unknown error";
  }
}

static bool device_is_gfx950(int dev)
{
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

extern "C" int elector_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; ++d) ok += device_is_gfx950(d) ? 1 : 0;
  return ok;
}

// --------------------------------------------------------------------- a1 ---

static void build_gap_arrays(elector_params *p, const int set[2][3], int T, int D)
{
  // seq_util.c:168-196
  const int M = T + D;
  p->max_gap_length = M;
  std::memset(p->gap_penalty_x, 0, sizeof p->gap_penalty_x);
  std::memset(p->gap_penalty_y, 0, sizeof p->gap_penalty_y);
  if (M + 1 >= ELECTOR_MAX_GAPTAB || M < 0) return;
  p->gap_penalty_x[0] = set[0][0];
  p->gap_penalty_y[0] = set[1][0];
  for (int i = 1; i < T; ++i) { p->gap_penalty_x[i] = set[0][1]; p->gap_penalty_y[i] = set[1][1]; }
  for (int i = 0; i < D; ++i) {
    const double dx = (set[0][1] - set[0][2]) / (double)(D + 1);
    const double dy = (set[1][1] - set[1][2]) / (double)(D + 1);
    p->gap_penalty_x[i + T] = (int)(set[0][1] - (i + 1) * dx);
    p->gap_penalty_y[i + T] = (int)(set[1][1] - (i + 1) * dy);
  }
  p->gap_penalty_x[M] = set[0][2];
  p->gap_penalty_y[M] = set[1][2];
  p->gap_penalty_x[M + 1] = 0;
  p->gap_penalty_y[M + 1] = 0;
}

extern "C" void elector_params_default(elector_params *p)
{
  // the values ELECTOR ships (src/poa-graph/blosum80.mat:8-42): 31 symbols,
  // identity 0 / mismatch -10, GAP-PENALTIES=10 5 5, truncation 10, decay 5
  static const char alphabet[] = "ARNDCQEGHILKMFPSTWYVBZX?agtcu]n";
  std::memset(p, 0, sizeof *p);
  p->nsymbol = (int)std::strlen(alphabet);
  std::memcpy(p->symbol, alphabet, (size_t)p->nsymbol);
  for (int i = 0; i < p->nsymbol; ++i)
    for (int j = 0; j < p->nsymbol; ++j) p->score[i][j] = (i == j) ? 0 : -10;
  const int set[2][3] = {{10, 5, 5}, {10, 5, 5}};
  build_gap_arrays(p, set, 10, 5);
}

extern "C" int elector_params_read(const char *path, elector_params *p)
{
  // replaces read_score_matrix (seq_util.c:82-217): '#'/blank lines skipped,
  // GAP-* directives, then the symbol line, then one score row per symbol.
  if (!path || !p) return ELECTOR_E_INVAL;
  std::FILE *f = std::fopen(path, "r");
  if (!f) return ELECTOR_E_IO;
  std::memset(p, 0, sizeof *p);
  int set[2][3] = {{12, 2, 0}, {12, 2, 0}};     // :89-91
  int T = 16, D = 0;                            // poa.h:13,19
  char line[1024];
  int nsym = 0;
  bool have_symbols = false;
  int rc = ELECTOR_OK;
  while (std::fgets(line, 1023, f)) {
    int i, j, k;
    if (line[0] == '#' || line[0] == '\n') continue;
    if (std::sscanf(line, "GAP-TRUNCATION-LENGTH=%d", &i) == 1) { T = i; continue; }
    if (std::sscanf(line, "GAP-DECAY-LENGTH=%d", &i) == 1) { D = i; continue; }
    if (std::sscanf(line, "GAP-PENALTIES=%d %d %d", &i, &j, &k) == 3) {
      set[0][0] = set[1][0] = i; set[0][1] = set[1][1] = j; set[0][2] = set[1][2] = k;
      continue;
    }
    if (std::sscanf(line, "GAP-PENALTIES-X=%d %d %d", &i, &j, &k) == 3) {
      set[1][0] = i; set[1][1] = j; set[1][2] = k;             // lands in the y arrays (:119-123)
      continue;
    }
    if (!have_symbols) {
      for (const char *c = line; *c; ++c)
        if (!std::isspace((unsigned char)*c)) {
          if (nsym >= ELECTOR_MAX_SYMBOL) { rc = ELECTOR_E_PARAMS; break; }
          p->symbol[nsym++] = *c;
        }
      have_symbols = true;
      if (rc) break;
      continue;
    }
    int row = -1;
    for (int s = nsym - 1; s >= 0; --s)
      if (p->symbol[s] == line[0]) { row = s; break; }           // backward scan, first hit (default.h:24)
    if (row < 0) { rc = ELECTOR_E_IO; break; }
    int pos = 1;
    for (int c = 0; c < nsym; ++c) {
      int v, used;
      if (std::sscanf(line + pos, "%d%n", &v, &used) != 1) { rc = ELECTOR_E_IO; break; }
      p->score[row][c] = v;
      pos += used;
    }
    if (rc) break;
  }
  std::fclose(f);
  if (rc) return rc;
  if (nsym <= 0) return ELECTOR_E_IO;
  p->nsymbol = nsym;
  if (T < 0 || D < 0 || T + D + 2 > ELECTOR_MAX_GAPTAB) return ELECTOR_E_PARAMS;
  build_gap_arrays(p, set, T, D);
  return ELECTOR_OK;
}

// ---------------------------------------------------------------- context ---

static int validate_params(const elector_params *p)
{
  if (p->nsymbol < 1 || p->nsymbol > ELECTOR_MAX_SYMBOL) return ELECTOR_E_PARAMS;
  if (p->max_gap_length < 0 || p->max_gap_length + 2 > ELECTOR_MAX_GAPTAB) return ELECTOR_E_PARAMS;
  // packed DP cell = score << 6 | tag in 32 bits; the per-window check in run_device_batch keeps
  // |score| * (cells on a path) below 2^24
  for (int i = 0; i < p->nsymbol; ++i)
    for (int j = 0; j < p->nsymbol; ++j)
      if (std::abs(p->score[i][j]) > 500) return ELECTOR_E_PARAMS;
  for (int g = 0; g <= p->max_gap_length + 1; ++g)
    if (std::abs(p->gap_penalty_x[g]) > 500 || std::abs(p->gap_penalty_y[g]) > 500) return ELECTOR_E_PARAMS;
  return ELECTOR_OK;
}

// Can the DP kernels use compile-time-simple scoring?  (uniform substitution
// scores, one extension penalty per direction) -- true for the shipped matrix.
static bool params_are_simple(const elector_params *p, KParams *kp)
{
  const int M = p->max_gap_length;
  kp->M = M;
  if (M < 1) return false;
  const int match = p->score[0][0], mismatch = p->nsymbol > 1 ? p->score[0][1] : 0;
  for (int i = 0; i < p->nsymbol; ++i)
    for (int j = 0; j < p->nsymbol; ++j)
      if (p->score[i][j] != (i == j ? match : mismatch)) return false;
  for (int g = 1; g <= M; ++g)
    if (p->gap_penalty_x[g] != p->gap_penalty_x[1] || p->gap_penalty_y[g] != p->gap_penalty_y[1]) return false;
  kp->open_x = p->gap_penalty_x[0]; kp->ext_x = p->gap_penalty_x[1];
  kp->open_y = p->gap_penalty_y[0]; kp->ext_y = p->gap_penalty_y[1];
  kp->match = match; kp->mismatch = mismatch;
  return true;
}

extern "C" int elector_ctx_create(int device, const elector_params *p, elector_ctx **out)
{
  if (!p || !out) return ELECTOR_E_INVAL;
  *out = nullptr;
  int rc = validate_params(p);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ELECTOR_E_NODEVICE;
  if (device < 0 || device >= ndev || !device_is_gfx950(device)) return ELECTOR_E_NODEVICE;
  elector_ctx *c = new (std::nothrow) elector_ctx();
  if (!c) return ELECTOR_E_NOMEM;
  c->device = device;
  c->params = *p;
  c->gen = !params_are_simple(p, &c->kp);
  if (std::getenv("ELECTOR_FORCE_GENERAL")) c->gen = true;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return ELECTOR_E_HIP;
  }
  // constant tables
  DevTables tab;
  std::memset(&tab, 0, sizeof tab);
  const int M = p->max_gap_length;
  for (int g = 0; g <= M; ++g) { tab.gpx[g] = p->gap_penalty_x[g]; tab.gpy[g] = p->gap_penalty_y[g]; }
  tab.gpx[M + 1] = tab.gpx[0]; tab.gpy[M + 1] = tab.gpy[0];   // global mode (align_lpo_po2.c:244-246)
  for (int i = 0; i < p->nsymbol; ++i)
    for (int j = 0; j < p->nsymbol; ++j) tab.sub[i * 32 + j] = p->score[i][j];
  for (int b = 0; b < 256; ++b) {
    // a2: tolower, then anything outside the alphabet becomes symbol[0]
    const char ch = (char)std::tolower(b);
    int idx = -1;
    for (int s = 0; s < p->nsymbol; ++s) if (p->symbol[s] == ch) { idx = s; break; }
    if (idx < 0) for (int s = 0; s < p->nsymbol; ++s) if (p->symbol[s] == p->symbol[0]) { idx = s; break; }
    tab.lut[b] = (uint8_t)idx;
  }
  for (int s = 0; s < p->nsymbol; ++s) tab.chr[s] = (uint8_t)p->symbol[s];
  // border cells: k gap steps from the origin along x (row -1) and y (column -1)
  std::vector<int32_t> linx(ELECTOR_MAX_SEQ + 2), liny(ELECTOR_MAX_SEQ + 2);
  {
    int sx = 0, gx = 0, sy = 0, gy = 0;
    for (int k = 0; k < ELECTOR_MAX_SEQ + 2; ++k) {
      linx[k] = (sx << kTagBits) | gx;
      liny[k] = (sy << kTagBits) | gy;
      sx -= tab.gpx[gx]; gx = std::min(gx + 1, M);
      sy -= tab.gpy[gy]; gy = std::min(gy + 1, M);
    }
  }
  rc = c->d_tab.ensure(sizeof tab) | c->d_linx.ensure(linx.size() * 4) | c->d_liny.ensure(liny.size() * 4);
  if (!rc) {
    if (hipMemcpy(c->d_tab.p, &tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_linx.p, linx.data(), linx.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_liny.p, liny.data(), liny.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      rc = ELECTOR_E_HIP;
  }
  if (rc) { elector_ctx_destroy(c); return rc; }
  *out = c;
  return ELECTOR_OK;
}

extern "C" void elector_ctx_destroy(elector_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto &s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  for (auto &e : c->hb_events) (void)hipEventDestroy(e);
  DevBuf *bufs[] = {&c->d_tab, &c->d_linx, &c->d_liny, &c->d_off, &c->d_perm, &c->d_mv1, &c->d_mv2, &c->d_sym,
                    &c->d_xinfo, &c->d_ring1, &c->d_map16, &c->d_carry, &c->d_moves, &c->d_n1, &c->d_cls,
                    &c->d_score1, &c->d_score2, &c->d_bx2, &c->d_bases, &c->d_cols, &c->d_ncol, &c->d_status,
                    &c->d_scores, &c->d_rowoff, &c->d_rows, &c->d_st_rows, &c->d_st_rowoff, &c->d_st_cols,
                    &c->d_st_first, &c->d_st_clips, &c->d_st_cnt, &c->d_st_mask, &c->d_st_scr, &c->d_st_dense, &c->d_st_outoff,
                    &c->d_list, &c->d_done, &c->d_rowinit,
                    &c->d_bnode, &c->d_bscore, &c->d_bpath, &c->d_bcons, &c->d_binfo, &c->d_bplan, &c->d_bin, &c->d_bcls, &c->d_fmv, &c->d_tstate, &c->d_tlist, &c->d_gring, &c->d_hand, &c->d_mvpool, &c->d_mvbusy, &c->d_pdesc, &c->d_psym,
                    &c->d_bin16, &c->d_wkey, &c->d_acc, &c->d_ginfo, &c->d_far,
                    &c->d_sp_reads, &c->d_sp_off, &c->d_sp_hdr, &c->d_sp_keys, &c->d_sp_vals, &c->d_sp_ca, &c->d_sp_cb, &c->d_sp_wl,
                    &c->d_sp_win, &c->d_sp_first, &c->d_sp_cnt, &c->d_sp_wfirst, &c->d_sp_wlen, &c->d_sp_woff, &c->d_sp_scan, &c->d_sp_bases, &c->d_sp_anc};
  for (DevBuf *b : bufs) b->release();
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  for (auto &s : c->st_slot) s.release();
  c->h_acc.release(); c->h_gen.release(); c->h_off.release();
  if (c->h_off_done) (void)hipEventDestroy(c->h_off_done);
  for (int k = 0; k < 2; ++k) {
    c->h_meta_buf[k].release();
    c->h_rows.release();
    if (c->h_meta_done[k]) (void)hipEventDestroy(c->h_meta_done[k]);
  }
  if (c->aux_ready) {
    for (int k = 0; k < elector_ctx::kAux; ++k) { (void)hipStreamDestroy(c->aux[k]); (void)hipEventDestroy(c->aux_done[k]); }
    (void)hipEventDestroy(c->fork);
  }
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  elector::rows_dma_release(&c->fetch_sig);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char *elector_ctx_last_error(const elector_ctx *c) { return c ? c->err.c_str() : ""; }

extern "C" int elector_ctx_sync(elector_ctx *c)
{
  if (!c) return ELECTOR_E_INVAL;
  {
    std::lock_guard<std::mutex> lock(c->mu);
    const int rc = elector_bundles_flush(c);
    if (rc) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ELECTOR_OK;
}

// ----------------------------------------------------------------- timing ---

extern "C" int elector_ctx_timing_enable(elector_ctx *c, int on)
{
  if (!c) return ELECTOR_E_INVAL;
  c->timing = on != 0;
  return ELECTOR_OK;
}

static void spans_collect(elector_ctx *c)
{
  for (auto &s : c->spans) {
    float ms = 0;
    if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      c->ms_acc[s.kind] += ms;
      c->launches_acc[s.kind] += 1;
    }
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  c->spans.clear();
}

extern "C" int elector_ctx_timing_read(elector_ctx *c, int kernel, double *ms, int64_t *launches)
{
  if (!c || kernel < 0 || kernel >= elector_ctx::kTimedKinds) return ELECTOR_E_INVAL;
  (void)hipSetDevice(c->device);
  spans_collect(c);
  if (ms) *ms = c->ms_acc[kernel];
  if (launches) *launches = c->launches_acc[kernel];
  return ELECTOR_OK;
}

extern "C" int elector_ctx_timing_reset(elector_ctx *c)
{
  if (!c) return ELECTOR_E_INVAL;
  (void)hipSetDevice(c->device);
  spans_collect(c);
  for (int k = 0; k < elector_ctx::kTimedKinds; ++k) { c->ms_acc[k] = 0; c->launches_acc[k] = 0; }
  return ELECTOR_OK;
}

extern "C" int elector_ctx_option(elector_ctx *c, const char *name, int64_t value)
{
  if (!c || !name) return ELECTOR_E_INVAL;
  std::lock_guard<std::mutex> lock(c->mu);
  if (!std::strcmp(name, "chains")) {                 // concurrent launch chains of the fused classes; 0 = default
    if (value < 0 || value > elector_ctx::kAux) return fail(c, ELECTOR_E_INVAL, "chains must be 0..4");
    c->chains = (int)value;
    return ELECTOR_OK;
  }
  if (!std::strcmp(name, "priority")) {               // the context's streams at the device's highest (-1) / lowest (+1) priority
    if (value < -1 || value > 1) return fail(c, ELECTOR_E_INVAL, "priority must be -1, 0 or 1");
    if (c->aux_ready || c->copy_stream) return fail(c, ELECTOR_E_INVAL, "priority must be set before the context's first call");
    c->priority = (int)value;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipStream_t ns = nullptr;
    if (c->make_stream(&ns)) return fail(c, ELECTOR_E_HIP, "stream");
    (void)hipStreamDestroy(c->stream);
    c->stream = ns;
    return ELECTOR_OK;
  }
  if (!std::strcmp(name, "bundles_now")) {            // 1: elector_poa_bundles_enqueue queues the search inside the call (ctx.h)
    c->bundles_now = value != 0;
    return ELECTOR_OK;
  }
  if (!std::strcmp(name, "cus")) {                    // the context's streams on compute units lo .. hi - 1 of the queue mask: value = lo * 1000 + hi
    const int lo = (int)(value / 1000), hi = (int)(value % 1000);
    if (value < 0 || lo >= hi || hi > 256) return fail(c, ELECTOR_E_INVAL, "cus: lo * 1000 + hi with 0 <= lo < hi <= 256");
    if (c->aux_ready || c->copy_stream) return fail(c, ELECTOR_E_INVAL, "cus must be set before the context's first call");
    c->cu_lo = lo; c->cu_hi = hi;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipStream_t ns = nullptr;
    if (c->make_stream(&ns)) return fail(c, ELECTOR_E_HIP, "stream with a compute-unit mask");
    (void)hipStreamDestroy(c->stream);
    c->stream = ns;
    return ELECTOR_OK;
  }
  return fail(c, ELECTOR_E_INVAL, "unknown option");
}

extern "C" int elector_ctx_last_po_sizes(elector_ctx *c, int64_t n, int32_t *po_nodes)
{
  if (!c || !po_nodes || n < 0 || n > c->last_n) return ELECTOR_E_INVAL;
  std::lock_guard<std::mutex> lock(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(po_nodes, c->d_n1.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return ELECTOR_OK;
}

// ------------------------------------------------------------------ batch ---

// Moves scratch of the generic kernels is the only part of the workspace that
// grows faster than the input; generic-path windows are processed in chunks whose
// scratch stays below this, and the device-side bump allocator (windows the fused
// kernels hand back) gets a fixed budget on top.
static const int64_t kMovesBudgetDwords = (int64_t)3 << 28;   // 3 GiB
static const int64_t kBumpBudgetDwords = (int64_t)1 << 30;    // 4 GiB
static const int64_t kWindowMovesMaxDwords = (int64_t)8 << 30;   // moves of ONE window (4 bits per DP cell): 32 GiB
static const int kLeftRoundsMax = 6;                          // passes over the handed-back windows (see run_device_batch)
static const int64_t kDeepRingBytes = (int64_t)4 << 30;       // HBM shadow rings of the deep-graph alignment #2 (k_dp2 DEEP)

static const int kPartChunk = 2048;     // list entries per block of the trivial-window partition
// a bin with fewer windows than this joins the next larger populated slot tier of its class
// (16,384 since the end of round 4, 4,096 before: a class of a few thousand windows of the 32- and 64-lane groups is a few
// hundred wavefronts that last as long as their longest window -- a launch that is all ramp and tail.  Same-box A/B,
// two runs each: un-overlapped k_poa 5.66 -> 5.47 ms on the E. coli batch, 9.93 -> 8.87 on the yeast -split batch
// (140 -> 130 and 180 -> 170 launches per ten steps), the pipelined rate +0.9 / +0.4 %; 65,536: 5.15 / 8.83 ms but
// -0.5 % pipelined on E. coli)
static const int64_t kMinBinWindows = std::getenv("ELECTOR_MIN_BIN") ? std::max<long long>(1, std::atoll(std::getenv("ELECTOR_MIN_BIN"))) : 16384;

static int ensure_streams(elector_ctx *c)
{
  if (c->aux_ready) return 0;
  for (int k = 0; k < elector_ctx::kAux; ++k) {
    if (c->make_stream(&c->aux[k])) return ELECTOR_E_HIP;
    if (hipEventCreateWithFlags(&c->aux_done[k], hipEventDisableTiming) != hipSuccess) return ELECTOR_E_HIP;
  }
  if (hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess) return ELECTOR_E_HIP;
  c->aux_ready = true;
  return 0;
}

// pinned staging of a caller's host offsets on their way to the device: several threads copy, one DMA follows
static int upload_offsets(elector_ctx *c, int64_t n, const int64_t *off, hipStream_t st)
{
  const size_t bytes = (size_t)(3 * n + 1) * 8;
  int rc = c->h_off.ensure(bytes) | c->d_off.ensure(bytes);
  if (rc) return ELECTOR_E_NOMEM;
  if (c->h_off_done) { if (hipEventSynchronize(c->h_off_done) != hipSuccess) return ELECTOR_E_HIP; }
  else if (hipEventCreateWithFlags(&c->h_off_done, hipEventDisableTiming) != hipSuccess) return ELECTOR_E_HIP;
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(8, n / 65536));
  auto work = [&](int t) {
    const size_t b0 = bytes * (size_t)t / (size_t)T & ~(size_t)7, b1 = t + 1 == T ? bytes : (bytes * (size_t)(t + 1) / (size_t)T & ~(size_t)7);
    std::memcpy(c->h_off.as<uint8_t>() + b0, reinterpret_cast<const uint8_t *>(off) + b0, b1 - b0);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto &x : th) x.join();
  if (hipMemcpyAsync(c->d_off.p, c->h_off.p, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return ELECTOR_E_HIP;
  if (hipEventRecord(c->h_off_done, st) != hipSuccess) return ELECTOR_E_HIP;
  return 0;
}

// d_off: the 3n + 1 window offsets in DEVICE memory (c->d_off itself, or a caller's array that is copied there: the
// merge / statistics stage and the bundle search read the last batch's offsets from the context)
static int run_device_batch(elector_ctx *c, int64_t n, const uint8_t *d_bases, const int64_t *d_off, int64_t total,
                            uint8_t *d_cols, int32_t *d_ncol, int32_t *d_status, int32_t *d_scores)
{
  // (from here to the end of a successful call the context describes no batch: a call that fails half way -- malformed
  // offsets are only found by the classification kernel, after d_off has been overwritten -- must not leave the previous
  // batch's sizes over the new batch's arrays for elector_msa_stats_enqueue / elector_poa_bundles to trust)
  {   // a bundle search noted for the previous batch reads that batch's graph: it is queued before this batch overwrites it
    const int rcb = elector_bundles_flush(c);
    if (rcb) return rcb;
  }
  c->last_n = 0; c->last_total = 0; c->graph_valid = false;
  if (n == 0) return ELECTOR_OK;
  if (total < 0) return fail(c, ELECTOR_E_INVAL, "negative total");
  const bool use_fused = !c->gen && !std::getenv("ELECTOR_NO_FUSED");
  // alignment #1 without a dynamic program for windows whose corrected sequence equals the reference:
  // valid when the diagonal is strictly best (see k_trivial)
  const bool use_trivial = use_fused && !std::getenv("ELECTOR_NO_TRIVIAL") && c->kp.match >= 0 &&
                           c->kp.mismatch <= c->kp.match && c->kp.open_x > 0 && c->kp.ext_x > 0 && c->kp.open_y > 0 &&
                           c->kp.ext_y > 0;
  // the one-kernel, two-windows-per-lane-group path (poa_pack.hip): symmetric gap penalties and a match score of 0
  // (the shipped parameters).  With elector_ctx_keep_graph it leaves the graph the bundle search reads in HBM as the
  // two-kernel path does (PackArgs::keep_graph; round 4 sent such batches through the two-kernel path: 2.4-3 x the step)
  const bool use_pack = use_trivial && !std::getenv("ELECTOR_NO_PACK") && c->kp.open_x == c->kp.open_y &&
                        c->kp.ext_x == c->kp.ext_y && c->kp.match == 0;

  const bool host_prof = std::getenv("ELECTOR_DEBUG_HOST") != nullptr;
  auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
  const double tp0 = now_ms();
  hipStream_t st = c->stream;

  // ---- per-window bookkeeping on the device (poa_classify.hip): status, launch class, size key; the host reads
  // back the per-class totals only ----
  const size_t part_chunks_max = (size_t)n / kPartChunk + kBins + 1;
  const size_t acc_glob = ((size_t)kAccRows * kBins + 1) & ~(size_t)1;         // the four 64-bit totals behind the rows, 8-byte aligned
  const size_t acc_ints = acc_glob + 8;
  int rc = c->d_off.ensure((size_t)(3 * n + 1) * 8) | c->d_bin16.ensure((size_t)n * 2 + 64) | c->d_wkey.ensure((size_t)n + 64) |
           c->d_acc.ensure(acc_ints * 4 + (size_t)kSortDestMax * kKeys * 4 + (size_t)(kBins + 2) * 2 + (size_t)kSortDestMax * 8 + 256) |
           c->d_perm.ensure((size_t)n * 4 + 64) | c->d_mv1.ensure((size_t)n * 8) | c->d_mv2.ensure((size_t)n * 8) |
           c->d_list.ensure((size_t)3 * n * 4 + (size_t)kBins * 32 + part_chunks_max * (16 + 4 * (size_t)partition_buckets()) + 64) |
           c->h_acc.ensure(acc_ints * 4 + 64);
  if (rc) return fail(c, ELECTOR_E_NOMEM, "classification workspace");
  if (d_off != c->d_off.as<int64_t>())
    HIPCHK(c, hipMemcpyAsync(c->d_off.p, d_off, (size_t)(3 * n + 1) * 8, hipMemcpyDeviceToDevice, st));
  int pen_abs_max = 1;                       // largest |score| or gap penalty of the parameter set
  for (int i = 0; i < c->params.nsymbol; ++i)
    for (int j = 0; j < c->params.nsymbol; ++j) pen_abs_max = std::max(pen_abs_max, std::abs(c->params.score[i][j]));
  for (int g = 0; g <= c->params.max_gap_length + 1; ++g)
    pen_abs_max = std::max(pen_abs_max, std::max(std::abs(c->params.gap_penalty_x[g]), std::abs(c->params.gap_penalty_y[g])));
  int cls_max_slot[kNC];
  for (int ci = 0; ci < kNC; ++ci) cls_max_slot[ci] = class_max_slot(ci);
  // testing knob: every window into one geometry class (multi-strip paths of the small classes)
  const int force_cls = std::getenv("ELECTOR_FORCE_CLASS") ? std::atoi(std::getenv("ELECTOR_FORCE_CLASS")) : -1;
  int32_t *d_acc = c->d_acc.as<int32_t>();
  uint32_t *d_hist = reinterpret_cast<uint32_t *>(d_acc + acc_ints);
  int64_t *d_dest_first = reinterpret_cast<int64_t *>(d_hist + (size_t)kSortDestMax * kKeys);
  int16_t *d_dest_of = reinterpret_cast<int16_t *>(d_dest_first + kSortDestMax);
  {
    HIPCHK(c, hipMemsetAsync(d_acc, 0, acc_ints * 4 + (size_t)kSortDestMax * kKeys * 4, st));
    ClassifyArgs ca;
    ca.n = n; ca.total = total; ca.off = c->d_off.as<int64_t>(); ca.kp = c->kp; ca.pen_abs_max = pen_abs_max;
    ca.use_fused = use_fused ? 1 : 0; ca.force_cls = force_cls;
    ca.coarse = 0;
    ca.window_moves_max = kWindowMovesMaxDwords;
    ca.status = d_status; ca.bin = c->d_bin16.as<int16_t>(); ca.wkey = c->d_wkey.as<uint8_t>();
    ca.acc = d_acc; ca.glob = reinterpret_cast<unsigned long long *>(d_acc + acc_glob);
    timed_begin(c, 2, st);
    launch_classify(ca, st);
    timed_end(c, st);
    // the totals come back as a kernel's stores into page-locked memory, not through the copy engine (which may be
    // busy for milliseconds with an earlier batch's merged rows)
    if (launch_words_to_host(c->h_acc.p, d_acc, acc_ints * 4, st))
      HIPCHK(c, hipMemcpyAsync(c->h_acc.p, d_acc, acc_ints * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  const double tp1 = now_ms();
  const int32_t *h_acc = c->h_acc.as<int32_t>();
  const unsigned long long *h_glob = reinterpret_cast<const unsigned long long *>(h_acc + acc_glob);
  if (h_glob[3]) return fail(c, ELECTOR_E_INVAL, "offsets must start at 0, be non-decreasing and end at the total");
  const int64_t n_generic = (int64_t)h_glob[0];
  const int64_t left_worst = (int64_t)h_glob[1];     // moves (dwords) of alignment #2 if every fused-routed window were handed back
  const int64_t max_po_bound = (int64_t)h_glob[2];   // largest Lr + Lc of the batch (bounds |PO| and with it a strip's steps)
  std::vector<int64_t> bin_cnt((size_t)kBins, 0), bin_need_a((size_t)7 * kBins, 0);   // need_a, then maxima of Lr, Lc, Lu, Lr + Lc, k_poa's slot need (any window / a trivial one)
  int64_t *bin_max_lr = bin_need_a.data() + kBins, *bin_max_lc = bin_max_lr + kBins, *bin_max_lu = bin_max_lc + kBins,
          *bin_max_po = bin_max_lu + kBins, *bin_need_pack = bin_max_po + kBins, *bin_need_triv = bin_need_pack + kBins;
  std::vector<int16_t> bin_final((size_t)kBins);
  // Launches are expensive in tails and ramps (measured: one launch per geometry class is 7 % faster
  // than one per 4096-window slot tier, and a handful of tiny extra launches costs 8 %), so:
  //  * a geometry class with few windows joins the next class of its group size (more rows per lane
  //    never hurts correctness or the slot need);
  //  * a class becomes ONE bin with its largest needed slot (a larger slot always fits the smaller
  //    needs) unless that is more than 1.5x what 98 % of its windows need -- then those 98 % form the
  //    main bin and the outliers keep their own tiers, sparse ones joining the next larger populated one.
  // (min_bin grows until the batch has no more lists than the device sort takes.)
  auto decide_bins = [&](int64_t min_bin) {
    for (int b = 0; b < kBins; ++b) {
      bin_cnt[(size_t)b] = h_acc[b];
      for (int q = 0; q < 7; ++q) bin_need_a[(size_t)(q * kBins + b)] = h_acc[(size_t)(q + 1) * kBins + b];
      bin_final[(size_t)b] = (int16_t)b;
    }
    auto merge_into = [&](int b, int into) {
      bin_cnt[(size_t)into] += bin_cnt[(size_t)b];
      for (int q = 0; q < 7; ++q)
        bin_need_a[(size_t)(q * kBins + into)] = std::max(bin_need_a[(size_t)(q * kBins + into)], bin_need_a[(size_t)(q * kBins + b)]);
      bin_cnt[(size_t)b] = 0;
      for (int x = 0; x < kBins; ++x) if (bin_final[(size_t)x] == b) bin_final[(size_t)x] = (int16_t)into;
    };
    for (int ci = 0; ci + 1 < kNC; ++ci) {
      if (cls_G(ci + 1) != cls_G(ci)) continue;
      int64_t tot = 0;
      for (int t = 0; t < kNT; ++t) tot += bin_cnt[(size_t)(ci * kNT + t)];
      if (!tot || tot >= min_bin / 2) continue;
      for (int t = 0; t < kNT; ++t)
        if (bin_cnt[(size_t)(ci * kNT + t)] && tier_bytes(t) <= cls_max_slot[ci + 1]) merge_into(ci * kNT + t, (ci + 1) * kNT + t);
    }
    for (int ci = 0; ci < kNC; ++ci) {
      int64_t tot = 0, acc = 0;
      int t_max = -1;
      for (int t = 0; t < kNT; ++t) { tot += bin_cnt[(size_t)(ci * kNT + t)]; if (bin_cnt[(size_t)(ci * kNT + t)]) t_max = t; }
      if (!tot) continue;
      int t_main = t_max;
      for (int t = 0; t < kNT; ++t) {
        acc += bin_cnt[(size_t)(ci * kNT + t)];
        if (acc * 50 >= tot * 49) { t_main = t; break; }
      }
      if (2 * tier_bytes(t_max) <= 3 * tier_bytes(t_main)) t_main = t_max;
      int into = -1;                                     // nearest larger tier that stays a launch
      for (int t = kNT - 1; t > t_main; --t) {
        const int b = ci * kNT + t;
        if (!bin_cnt[(size_t)b]) continue;
        if (into >= 0 && bin_cnt[(size_t)b] < min_bin) merge_into(b, into);
        else into = b;
      }
      for (int t = 0; t < t_main; ++t)
        if (bin_cnt[(size_t)(ci * kNT + t)]) merge_into(ci * kNT + t, ci * kNT + t_main);
    }
    int lists = 0;
    for (int b = 0; b < kBins; ++b) lists += bin_cnt[(size_t)b] != 0;
    return lists;
  };
  {
    int64_t min_bin = kMinBinWindows;
    while (decide_bins(min_bin) + 1 > kSortDestMax) min_bin = min_bin < ((int64_t)1 << 40) ? min_bin * 8 : min_bin;   // ends: at most two bins per class are left
  }
  std::vector<int64_t> bin_first((size_t)kBins + 1);
  bin_first[0] = 0;
  for (int b = 0; b < kBins; ++b) bin_first[(size_t)b + 1] = bin_first[(size_t)b] + bin_cnt[(size_t)b];

  // ---- the small tables of this batch in pinned memory (double-buffered: the copies are asynchronous) ----
  c->h_meta_cur ^= 1;
  elector::HostPinned &h_meta = c->h_meta_buf[c->h_meta_cur];
  hipEvent_t &h_done = c->h_meta_done[c->h_meta_cur];
  if (!h_done) HIPCHK(c, hipEventCreateWithFlags(&h_done, hipEventDisableTiming));
  else HIPCHK(c, hipEventSynchronize(h_done));                     // the copies of the batch before last have run
  const size_t meta_bytes = (size_t)kBins * 24 + part_chunks_max * 16 + (size_t)(kBins + 2) * 2 + (size_t)kSortDestMax * 8 + 256;
  rc = h_meta.ensure(meta_bytes);
  if (rc) return fail(c, rc, "pinned metadata");
  int64_t *h_bins = h_meta.as<int64_t>();                      // (first, count) of every occupied fused bin
  int64_t *h_chunks = h_bins + 2 * kBins;                      // partition chunks: first, (len | bin << 32)
  int64_t *h_dest_first = h_chunks + 2 * part_chunks_max;
  int32_t *h_bin_chunks = reinterpret_cast<int32_t *>(h_dest_first + kSortDestMax);   // first chunk, #chunks per bin
  int16_t *h_dest_of = reinterpret_cast<int16_t *>(h_bin_chunks + 2 * kBins);

  // ---- the lists: counting sort of the windows by (list, size key) on the device ----
  uint32_t *d_generic = c->d_perm.as<uint32_t>();
  uint32_t *d_lists = c->d_list.as<uint32_t>();
  {
    int ndest = 0;
    std::vector<int> dest_of_bin((size_t)kBins, -1);
    for (int b = 0; b < kBins; ++b)
      if (bin_cnt[(size_t)b]) { dest_of_bin[(size_t)b] = ndest; h_dest_first[ndest++] = bin_first[(size_t)b]; }
    h_dest_first[ndest] = 0;                                            // the generic list, an array of its own
    for (int b = 0; b < kBins; ++b) h_dest_of[b] = (int16_t)std::max(0, dest_of_bin[(size_t)bin_final[(size_t)b]]);
    h_dest_of[kBins] = (int16_t)ndest;
    HIPCHK(c, hipMemcpyAsync(d_dest_of, h_dest_of, (size_t)(kBins + 1) * 2, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_dest_first, h_dest_first, (size_t)(ndest + 1) * 8, hipMemcpyHostToDevice, st));
    SortArgs sa;
    sa.n = n; sa.bin = c->d_bin16.as<int16_t>(); sa.wkey = c->d_wkey.as<uint8_t>(); sa.dest_of = d_dest_of; sa.ndest = ndest + 1;
    sa.hist = d_hist; sa.dest_first = d_dest_first; sa.lists = d_lists; sa.generic = d_generic;
    timed_begin(c, 2, st);
    if (launch_sort(sa, st)) return fail(c, ELECTOR_E_HIP, "list sort");
    timed_end(c, st);
  }
  if (std::getenv("ELECTOR_DEBUG_BINS")) {
    std::fprintf(stderr, "[elector] n=%lld generic=%lld classes:", (long long)n, (long long)n_generic);
    for (int b = 0; b < kBins; ++b)
      if (bin_cnt[(size_t)b])
        std::fprintf(stderr, " G%dxR%d/%d:%lld", cls_G(b / kNT), cls_R(b / kNT), tier_bytes(b % kNT), (long long)bin_cnt[(size_t)b]);
    std::fprintf(stderr, "\n");
  }
  const double tp2 = now_ms();
  // moves scratch of the generic-path windows, in chunks.  The generic list is short on the shipped parameters (windows
  // no fused class takes); the host reads its windows' lengths back, lays their moves out and sends the offsets
  // down again.  Every other window's entry stays -1.
  struct Chunk { int64_t k0, k1, dwords; };
  std::vector<Chunk> chunks;
  HIPCHK(c, hipMemsetAsync(c->d_mv1.p, 0xFF, (size_t)n * 8, st));
  HIPCHK(c, hipMemsetAsync(c->d_mv2.p, 0xFF, (size_t)n * 8, st));
  const int32_t *g_info = nullptr;                     // per generic-list entry: Lr, Lc, Lu, status
  std::vector<uint32_t> g_list;                        // ... and its window
  if (n_generic) {
    if (c->h_gen.ensure((size_t)n_generic * (16 + 16) + 64) || c->d_ginfo.ensure((size_t)n_generic * (16 + 16) + 64))
      return fail(c, ELECTOR_E_NOMEM, "generic-list staging");
    int32_t *d_info = c->d_ginfo.as<int32_t>();
    int64_t *d_gmv = reinterpret_cast<int64_t *>(d_info + 4 * n_generic);
    launch_generic_info(d_generic, n_generic, c->d_off.as<int64_t>(), d_status, d_info, st);
    HIPCHK(c, hipMemcpyAsync(c->h_gen.p, d_info, (size_t)n_generic * 16, hipMemcpyDeviceToHost, st));
    g_list.resize((size_t)n_generic);
    HIPCHK(c, hipMemcpyAsync(g_list.data(), d_generic, (size_t)n_generic * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    g_info = c->h_gen.as<int32_t>();
    int64_t *h_gmv = reinterpret_cast<int64_t *>(c->h_gen.as<int32_t>() + 4 * n_generic);
    int64_t k0 = 0, acc = 0;
    for (int64_t k = 0; k < n_generic; ++k) {
      int64_t d1 = 0, d2 = 0;
      if (!g_info[4 * k + 3]) {
        const int64_t lr = g_info[4 * k], lc = g_info[4 * k + 1], lu = g_info[4 * k + 2];
        d1 = (int64_t)n_strips((int)lc) * mv_tw((int)lr) * 64;
        d2 = (int64_t)n_strips((int)lu) * mv_tw((int)(lr + lc)) * 64;   // |PO| <= Lr + Lc
      }
      if (acc + d1 + d2 > kMovesBudgetDwords && k > k0) { chunks.push_back({k0, k, acc}); k0 = k; acc = 0; }
      h_gmv[2 * k] = acc; acc += d1;
      h_gmv[2 * k + 1] = acc; acc += d2;
    }
    chunks.push_back({k0, n_generic, acc});
    HIPCHK(c, hipMemcpyAsync(d_gmv, h_gmv, (size_t)n_generic * 16, hipMemcpyHostToDevice, st));
    launch_generic_moves(d_generic, n_generic, d_gmv, c->d_mv1.as<int64_t>(), c->d_mv2.as<int64_t>(), st);
  } else chunks.push_back({0, 0, 0});
  int64_t max_dwords = 0;
  for (auto &ch : chunks) max_dwords = std::max(max_dwords, ch.dwords);
  // Windows the fused kernels hand back get their moves from a device-side bump allocator.  Its budget covers
  // the worst case (every window handed back) up to 4 GiB; beyond that the leftovers are worked off in several
  // rounds (collect what fits -> alignment #2 -> fusion #2), so no window within the documented limits is ever
  // refused for lack of scratch.  Rounds that find nothing left cost four empty launches.
  const int64_t bump_dwords = use_fused ? std::min<int64_t>(kBumpBudgetDwords, std::max<int64_t>((int64_t)1 << 22, left_worst)) : 0;
  const int left_rounds = use_fused ? (int)std::min<int64_t>(kLeftRoundsMax, (left_worst + bump_dwords - 1) / std::max<int64_t>(1, bump_dwords) +
                                                                            (left_worst > bump_dwords ? 1 : 0)) : 0;
  // HBM shadow rings of the deep-graph alignment #2: one region of (|PO| + 66) x 64 cells per block
  const int64_t gring_block = (max_po_bound + 66) * 64;
  const int deep_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(4096, kDeepRingBytes / (gring_block * 4)));

  // moves scratch of the fused kernels: launches on one stream run one after the other and the moves
  // of a launch die with it, so every stream owns one region as large as its largest launch needs
  // ([block][strips][steps][64 lanes] words; strips and steps from the bin's maxima)
  int64_t fmv_stream[4] = {0, 0, 0, 0};
  // blocks of the k_fused_a / k_fused_b launches of a bin: one per 64 / G windows -- behind k_poa they only see
  // the windows it handed back (a device-built list), with a grid of a sixteenth and a loop inside
  auto old_grid = [&](int b) {
    const int G = cls_G(b / kNT);
    const int64_t full = (bin_cnt[(size_t)b] + 64 / G - 1) / (64 / G);
    return use_pack ? std::min<int64_t>(full, std::max<int64_t>(256, full / 16)) : full;
  };
  auto fmv_geom = [&](int b, bool second, int *tw, int *ns) {
    const int G = cls_G(b / kNT), R = cls_R(b / kNT);
    *tw = (int)(second ? bin_max_po[b] : bin_max_lr[b]) + G;
    *ns = (int)(((second ? bin_max_lu[b] : bin_max_lc[b]) + G * R - 1) / (G * R));
    const int64_t blocks = old_grid(b);
    return blocks * *ns * *tw * 64 * fused_mv_bytes(R);
  };
  // k_poa: LDS slot, moves geometry and scratch-slot pool of a bin
  struct PackGeom { int slot, tw, slots; int64_t pool_words; };
  auto pack_geom = [&](int b) {
    const int G = cls_G(b / kNT), nw = 2 * (64 / G);
    PackGeom pg;
    const int max_slot = ((160 * 1024 - 256 - 64) / nw) & ~15;
    pg.slot = (int)std::min<int64_t>(max_slot, (bin_need_pack[b] + 15) & ~(int64_t)15);
    // (Slots sized for the LDS banks -- 16 mod 32 bytes for the 8-lane classes, 32 mod 64 for the 16-lane ones -- took a
    // third off the bank-conflict cycles and made k_poa 0.5-2 % slower: the wavefronts never wait for the LDS pipeline and
    // the bytes a slot grows by cost occupancy.  A list in two launches, its shortcut-graph windows with a third less LDS,
    // lost 1-3 % to the second launch's ramp and tail.  Both measured in round 4; neither is kept.)
    pg.tw = (int)bin_max_po[b] + 8 + G + 4;
    const int lds_block = 64 + nw * pg.slot;
    const int waves_cu = std::max(1, std::min(32, (160 * 1024) / lds_block));
    pg.slots = 32 * waves_cu + 8;                                  // per XCD: 32 CUs, every wave they can hold, and a margin
    pg.pool_words = (int64_t)8 * pg.slots * pg.tw * 64;
    return pg;
  };
  int64_t pool_stream[4] = {0, 0, 0, 0};
  int pool_slots[4] = {0, 0, 0, 0}, pool_tw[4] = {0, 0, 0, 0};     // per launch chain: slots per XCD, longest moves region
  // Launch chains.  Two kernels of a context side by side use the chip best: a single chain pays every kernel's
  // tail, four and more compete for LDS and L2 (bench batch, one context: 8.1 ms per step with two chains,
  // 9.9 ms with four or more).  With several contexts in flight and sixteen hardware queues (GPU_MAX_HW_QUEUES,
  // set by the Python package) three chains are within +-3 % of two, box by box: two it is.  Within a chain big
  // classes first.  ELECTOR_CHAINS=1|3|4 are the alternatives for experiments (4 = one stream per group size).
  const int n_chains = c->chains > 0 ? c->chains
                       : std::getenv("ELECTOR_CHAINS") ? std::max(1, std::min(4, std::atoi(std::getenv("ELECTOR_CHAINS")))) : 2;
  std::vector<int> bin_stream((size_t)kBins, 0), bin_order;
  int n_used = n_chains;                               // auxiliary streams that carry launch chains
  {
    auto group_of = [&](int b) { const int G = cls_G(b / kNT); return G == 64 ? 0 : G == 32 ? 1 : G == 16 ? 2 : 3; };
    int64_t gwork[4] = {0, 0, 0, 0};
    for (int b = 0; b < kBins; ++b)
      if (bin_cnt[(size_t)b]) gwork[group_of(b)] += bin_cnt[(size_t)b] * (bin_max_lr[b] + 8) * (bin_max_lu[b] + 8);
    int chain_of_group[4] = {0, 1, 2, 3};
    if (n_chains == 1) chain_of_group[1] = chain_of_group[2] = chain_of_group[3] = 0;
    else if (n_chains < 4) {
      int64_t best = -1;
      for (int m = 0; m < 8; ++m) {                      // group 0 on chain 0; the others either way
        int64_t load[2] = {gwork[0], 0};
        for (int gi = 1; gi < 4; ++gi) load[(m >> (gi - 1)) & 1] += gwork[gi];
        const int64_t mx = std::max(load[0], load[1]);
        if (best < 0 || mx < best) { best = mx; for (int gi = 1; gi < 4; ++gi) chain_of_group[gi] = (m >> (gi - 1)) & 1; }
      }
      chain_of_group[0] = 0;
    }
    // two (or three) chains: the bins, in class order, are dealt in turn -- the chains then hold equal shares
    // of every group's work whatever the window distribution (as fast as the best hand-picked split of
    // the groups; a split by the work estimate above was 4 % slower), and the kernels that run side
    // by side are of neighbouring classes.
    const bool deal = true;
    int turn = 0;
    for (int b = kBins - 1; b >= 0; --b)
      if (bin_cnt[(size_t)b]) {
        bin_stream[(size_t)b] = (deal && (n_chains == 2 || n_chains == 3)) ? (turn++ % n_chains) : chain_of_group[group_of(b)];
        bin_order.push_back(b);
      }
    if (std::getenv("ELECTOR_DEBUG_BINS"))
      std::fprintf(stderr, "[elector] chains: G64->%d G32->%d G16->%d G8->%d (work %lld %lld %lld %lld)\n", chain_of_group[0],
                   chain_of_group[1], chain_of_group[2], chain_of_group[3], (long long)gwork[0], (long long)gwork[1],
                   (long long)gwork[2], (long long)gwork[3]);
  }
  auto stream_of = [&](int b) { return bin_stream[(size_t)b]; };
  // Behind k_poa the two-kernel path sees what was handed back, about one window in a hundred: one launch pair per
  // geometry class was thirty-odd launches of a few dozen wavefronts each, every one of them all latency.  The
  // classes of one lane-group size share a hand-back list instead (their regions of d_hand are neighbours) and ONE
  // launch pair with 8 rows per lane, which holds every window of the group's classes.
  struct HandGroup { int G = 0, first_bin = -1, ci8 = 0, slot_a = 0, slot_b = 0, tw_a = 0, ns_a = 0, tw_b = 0, ns_b = 0;
                     int64_t cnt = 0, blocks = 0; std::vector<int> bins;
                     // the group's far-edge launch (k_poa<G, 8, true>): list capacity, LDS slot, moves steps, symbol stride,
                     // where its descriptors / symbols start behind the bins', the moves pool it borrows from
                     int64_t far_cap = 0, far_desc_first = 0, far_psym_first = 0; int far_slot = 0, far_tw = 0, far_stride = 0, far_pool = 0; };
  HandGroup hgrp[4];
  const bool merge_hand = use_pack;
  // graphs with ONE far edge (a corrected piece that aligns at both ends of its window, an indel of two or more letters)
  // stay in k_poa: a launch of their own per lane-group size behind the group's bins.  ELECTOR_NO_FAR=1: the two-kernel
  // path and the generic kernels take them, as up to round 3 (A/B)
  const bool use_far = merge_hand && !std::getenv("ELECTOR_NO_FAR") && n_chains <= 3;
  if (merge_hand) {
    for (int b = 0; b < kBins; ++b) {
      if (!bin_cnt[(size_t)b]) continue;
      const int ci = b / kNT, G = cls_G(ci);
      HandGroup &hg = hgrp[G == 8 ? 0 : G == 16 ? 1 : G == 32 ? 2 : 3];
      if (hg.first_bin < 0) { hg.first_bin = b; hg.G = G; }
      hg.bins.push_back(b);
      hg.cnt += bin_cnt[(size_t)b];
      hg.slot_a = std::max<int>(hg.slot_a, (int)((bin_need_a[(size_t)b] + 127) & ~(int64_t)127));
      hg.slot_b = std::max(hg.slot_b, tier_bytes(b % kNT));
      hg.tw_a = std::max<int>(hg.tw_a, (int)bin_max_lr[b] + G);
      hg.tw_b = std::max<int>(hg.tw_b, (int)bin_max_po[b] + G);
      hg.ns_a = std::max<int>(hg.ns_a, (int)((bin_max_lc[b] + G * 8 - 1) / (G * 8)));
      hg.ns_b = std::max<int>(hg.ns_b, (int)((bin_max_lu[b] + G * 8 - 1) / (G * 8)));
    }
    for (HandGroup &hg : hgrp) {
      if (hg.first_bin < 0) continue;
      for (int ci = 0; ci < kNC; ++ci) if (cls_G(ci) == hg.G && cls_R(ci) == 8) hg.ci8 = ci;
      hg.slot_a = std::min(hg.slot_a, cls_max_slot[hg.ci8]);
      hg.slot_b = std::min(hg.slot_b, cls_max_slot[hg.ci8]);
      const int64_t full = (hg.cnt + 64 / hg.G - 1) / (64 / hg.G);
      hg.blocks = std::min<int64_t>(full, std::max<int64_t>(256, full / 16));
      const int64_t bytes = hg.blocks * std::max<int64_t>((int64_t)hg.ns_a * hg.tw_a, (int64_t)hg.ns_b * hg.tw_b) * 64 * fused_mv_bytes(8);
      fmv_stream[3] = std::max(fmv_stream[3], bytes);
    }
  }
  if (use_fused)
    for (int b = 0; b < kBins; ++b) {
      if (!bin_cnt[(size_t)b]) continue;
      int tw, ns;
      int64_t &r = fmv_stream[use_pack ? 3 : stream_of(b)];      // behind k_poa the two-kernel path runs on the last stream
      r = std::max(r, std::max(fmv_geom(b, false, &tw, &ns), fmv_geom(b, true, &tw, &ns)));
      if (use_pack) {
        const PackGeom pg = pack_geom(b);
        const int sk = stream_of(b);
        pool_slots[sk] = std::max(pool_slots[sk], pg.slots);
        pool_tw[sk] = std::max(pool_tw[sk], pg.tw);
      }
    }
  if (use_far)
    for (HandGroup &hg : hgrp) {
      if (hg.first_bin < 0) continue;
      const int nw = 2 * (64 / hg.G), max_slot = ((160 * 1024 - 256 - 64) / nw) & ~15;
      for (int b : hg.bins) {
        const PackGeom pg = pack_geom(b);
        hg.far_slot = std::max(hg.far_slot, pg.slot);
        hg.far_tw = std::max(hg.far_tw, pg.tw);
        hg.far_stride = std::max<int>(hg.far_stride, (int)(((bin_max_lr[b] + bin_max_lc[b] + bin_max_lu[b] + 3) / 4 + 3) & ~(int64_t)3));
      }
      hg.far_slot = std::min(hg.far_slot, max_slot);
      hg.far_cap = std::min<int64_t>(hg.cnt, std::max<int64_t>(2048, hg.cnt / 8));
      // with several chains the launch runs on the hand-back stream (a pool of its own), otherwise in line on the chain's
      hg.far_pool = n_chains > 1 ? 3 : stream_of(hg.first_bin);
      const int waves_cu = std::max(1, std::min(8, (160 * 1024) / (64 + nw * hg.far_slot)));   // (two waves per SIMD: launch bounds)
      pool_slots[hg.far_pool] = std::max(pool_slots[hg.far_pool], 32 * waves_cu + 8);
      pool_tw[hg.far_pool] = std::max(pool_tw[hg.far_pool], hg.far_tw);
    }
  if (use_pack)
    for (int k = 0; k < 4; ++k) pool_stream[k] = (int64_t)8 * pool_slots[k] * pool_tw[k] * 64;
  for (int k = 0; k < 4; ++k) fmv_stream[k] = (fmv_stream[k] + 255) & ~(int64_t)255;

  // ---- workspace ----
  const size_t nodes = (size_t)total + (size_t)n + 8;
  // (the offsets, both lists and the moves offsets were sized in front of the classification)
  rc = c->d_sym.ensure((size_t)total + 64) | c->d_xinfo.ensure(nodes * 8) |
       c->d_ring1.ensure(nodes * 2) | c->d_map16.ensure(nodes * 4) | c->d_carry.ensure(nodes * 4) |
       c->d_moves.ensure((size_t)(max_dwords + bump_dwords) * 4 + 1024) | c->d_n1.ensure((size_t)n * 4) |
       c->d_cls.ensure((size_t)n) | c->d_score1.ensure((size_t)n * 4) | c->d_score2.ensure((size_t)n * 4) |
       c->d_bx2.ensure((size_t)n * 4) |
       c->d_done.ensure((size_t)5 * n + 64) |
       c->d_rowinit.ensure(1024 + 256 * (size_t)kBins) |
       c->d_fmv.ensure((size_t)(fmv_stream[0] + fmv_stream[1] + fmv_stream[2] + fmv_stream[3]) + 256) |
       c->d_gring.ensure((size_t)deep_blocks * (size_t)gring_block * 4 + 256);
  // k_poa's inputs in list order (k_gather): a 32-byte descriptor per fused-routed window and its symbols at the
  // bin's stride (dwords; the sum of the bin's three length maxima bounds every window's total)
  std::vector<int64_t> psym_first((size_t)kBins + 1, 0);
  std::vector<int> pstride((size_t)kBins, 0);
  if (use_pack)
    for (int b = 0; b < kBins; ++b) {
      if (bin_cnt[(size_t)b]) pstride[(size_t)b] = (int)(((bin_max_lr[b] + bin_max_lc[b] + bin_max_lu[b] + 3) / 4 + 3) & ~(int64_t)3);
      psym_first[(size_t)b + 1] = psym_first[(size_t)b] + bin_cnt[(size_t)b] * pstride[(size_t)b];
    }
  int64_t far_desc_total = 0, far_psym_total = 0;
  if (use_far)
    for (HandGroup &hg : hgrp) {
      if (hg.first_bin < 0) continue;
      hg.far_desc_first = (n - n_generic) + far_desc_total;
      hg.far_psym_first = psym_first[(size_t)kBins] + far_psym_total;
      far_desc_total += hg.far_cap;
      far_psym_total += hg.far_cap * hg.far_stride;
    }
  if (!rc && use_pack)
    rc = c->d_pdesc.ensure((size_t)(n - n_generic + far_desc_total) * 32 + 64) |
         c->d_psym.ensure((size_t)(psym_first[(size_t)kBins] + far_psym_total) * 4 + 256) |
         c->d_far.ensure((size_t)n * 4 + 64) |
         c->d_hand.ensure((size_t)n * 4 + (size_t)kBins * 4 + 64) |
         c->d_mvpool.ensure((size_t)(pool_stream[0] + pool_stream[1] + pool_stream[2] + pool_stream[3]) * 4 + 256) |
         c->d_mvbusy.ensure((size_t)4 * 8 * kPoolStride * 4);
  if (rc) return fail(c, ELECTOR_E_NOMEM, "device workspace");
  if (use_fused && (rc = ensure_streams(c))) return fail(c, rc, "auxiliary streams");

  const double tp3 = now_ms();
  uint32_t *d_leftb = d_lists + n;                 // device-built list for alignment #2 leftovers
  uint32_t *d_lists2 = d_lists + 2 * n;            // the fused bins' lists, windows that need alignment #1 first
  int64_t *d_bins = reinterpret_cast<int64_t *>(d_lists2 + n + (n & 1));
  int64_t *d_chunks = d_bins + 2 * kBins;
  int32_t *d_bin_need = reinterpret_cast<int32_t *>(d_chunks + 2 * part_chunks_max);
  int32_t *d_bin_chunks = d_bin_need + kBins, *d_chunk_need = d_bin_chunks + 2 * kBins;
  static_assert(kSortDestMax <= kBins / 2, "d_bin_need holds two counts per list");
  int32_t *d_bin_a1 = d_bin_need + kBins / 2;     // per list: the entries in front that run alignment #1 (no shortcut graph)
  uint8_t *d_done_a = c->d_done.as<uint8_t>(), *d_done_b = d_done_a + n, *d_triv = d_done_b + n, *d_pkey = d_triv + n,
          *d_tiled = d_pkey + n;
  int nbins_used = 0, nchunks = 0;
  std::vector<int> bin_slot((size_t)kBins, -1);
  if (use_trivial)
    for (int b = 0; b < kBins; ++b)
      if (bin_cnt[(size_t)b]) {
        h_bins[2 * nbins_used] = bin_first[(size_t)b];
        h_bins[2 * nbins_used + 1] = bin_cnt[(size_t)b];
        h_bin_chunks[2 * nbins_used] = nchunks;
        for (int64_t i = 0; i < bin_cnt[(size_t)b]; i += kPartChunk) {
          h_chunks[2 * nchunks] = bin_first[(size_t)b] + i;
          const int32_t len = (int32_t)std::min<int64_t>(kPartChunk, bin_cnt[(size_t)b] - i);
          h_chunks[2 * nchunks + 1] = (int64_t)(uint32_t)len | ((int64_t)nbins_used << 32);
          ++nchunks;
        }
        h_bin_chunks[2 * nbins_used + 1] = nchunks - h_bin_chunks[2 * nbins_used];
        bin_slot[(size_t)b] = nbins_used++;
      }
  if (nbins_used) {
    HIPCHK(c, hipMemcpyAsync(d_bins, h_bins, (size_t)nbins_used * 16, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_chunks, h_chunks, (size_t)nchunks * 16, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_bin_chunks, h_bin_chunks, (size_t)nbins_used * 8, hipMemcpyHostToDevice, st));
  }
  // device counters: [0] = leftover count (int32), [2..3] = bump allocator (u64)
  int32_t *d_counters = reinterpret_cast<int32_t *>(c->d_rowinit.p);
  HIPCHK(c, hipEventRecord(h_done, st));
  HIPCHK(c, hipMemsetAsync(c->d_done.p, 0, (size_t)5 * n, st));
  HIPCHK(c, hipMemsetAsync(d_counters, 0, 64, st));
  if (use_pack) {
    HIPCHK(c, hipMemsetAsync(c->d_hand.as<uint32_t>() + n, 0, (size_t)kBins * 4, st));
    HIPCHK(c, hipMemsetAsync(c->d_far.as<uint32_t>() + n, 0, 16, st));
    for (int k = 0; k < 4; ++k)
      if (pool_slots[k] > 0) launch_poa_pool_init(c->d_mvbusy.as<int32_t>() + (size_t)k * 8 * kPoolStride, 8, pool_slots[k], st);
  }
  if (std::getenv("ELECTOR_DEBUG_FUSED")) HIPCHK(c, hipMemsetAsync(c->d_rowinit.as<uint8_t>() + 1024, 0, 256 * (size_t)kBins, st));

  const double tp4 = now_ms();
  if (host_prof)
    std::fprintf(stderr, "[elector] host: classification (device, with the wait) %.2f ms, bins + list sort %.2f ms, generic list + workspace %.2f ms, tables %.2f ms\n",
                 tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3);
  timed_begin(c, 2, st);
  launch_symbolize(d_bases, c->d_sym.as<uint8_t>(), total, c->d_tab.as<DevTables>(), st);
  timed_end(c, st);

  BatchArgs a;
  std::memset(&a, 0, sizeof a);
  a.off = c->d_off.as<int64_t>();
  a.bases = d_bases;
  a.sym = c->d_sym.as<uint8_t>();
  a.xinfo = c->d_xinfo.as<int2>();
  a.ring1 = c->d_ring1.as<uint16_t>();
  a.map16 = c->d_map16.as<uint32_t>();
  a.carry = c->d_carry.as<int32_t>();
  a.moves = c->d_moves.as<uint32_t>();
  a.mv1 = c->d_mv1.as<int64_t>();
  a.mv2 = c->d_mv2.as<int64_t>();
  a.n1 = c->d_n1.as<int32_t>();
  a.cls = c->d_cls.as<uint8_t>();
  a.score1 = c->d_score1.as<int32_t>();
  a.score2 = c->d_score2.as<int32_t>();
  a.bx2 = c->d_bx2.as<int32_t>();
  a.cols = d_cols;
  a.ncol = d_ncol;
  a.status = d_status;
  a.linx = c->d_linx.as<int32_t>();
  a.liny = c->d_liny.as<int32_t>();
  a.tab = c->d_tab.as<DevTables>();
  a.kp = c->kp;
  a.skip_a = d_done_a;
  a.skip_b = d_done_b;
  a.tiled = d_tiled;

  // Long windows of the generic path (the reference's whole-read fallback): both dynamic programs as
  // tiles of 63 rows x kTileCols columns, one launch per tile anti-diagonal (k_dp1_tile / k_dp2_tile),
  // so that one window keeps up to min(strips, column blocks) wavefronts busy instead of one.
  const int64_t tile_cells = std::getenv("ELECTOR_TILE_CELLS") ? std::atoll(std::getenv("ELECTOR_TILE_CELLS")) : ((int64_t)1 << 24);
  struct LongWin { uint32_t w; int lr, lc, lu; int64_t k; };
  auto long_windows = [&](int64_t k0, int64_t k1) {
    std::vector<LongWin> v;
    for (int64_t k = k0; k < k1; ++k) {
      if (g_info[4 * k + 3]) continue;
      const int64_t lr = g_info[4 * k], lc = g_info[4 * k + 1], lu = g_info[4 * k + 2];
      if (std::max(lr * lc, (lr + lc) * lu) >= tile_cells) v.push_back({g_list[(size_t)k], (int)lr, (int)lc, (int)lu, k});
    }
    return v;
  };
  // phase 1: zero the windows' moves, mark them, alignment #1 by tiles; phase 2 (after fusion #1): alignment #2
  auto run_tiles = [&](const std::vector<LongWin> &lw, int phase) -> int {
    if (lw.empty()) return 0;
    const int nw = (int)lw.size();
    std::vector<int64_t> st_off((size_t)nw);
    std::vector<uint32_t> wl((size_t)nw);
    int64_t ints = 0;
    int max_d = 0, max_tiles = 1;
    for (int k = 0; k < nw; ++k) {
      const LongWin &x = lw[(size_t)k];
      const int ns = n_strips(phase == 1 ? x.lc : x.lu);
      const int ncb = ((phase == 1 ? x.lr : x.lr + x.lc) + kTileCols - 1) / kTileCols;
      wl[(size_t)k] = x.w;
      st_off[(size_t)k] = ints;
      ints += (int64_t)ns * 2 * (phase == 1 ? kTileState1 : kTileState2);
      max_d = std::max(max_d, ns + ncb - 2);
      max_tiles = std::max(max_tiles, std::min(ns, ncb));
    }
    if (c->d_tstate.ensure((size_t)ints * 4 + 64) || c->d_tlist.ensure((size_t)nw * 12 + 64)) return ELECTOR_E_NOMEM;
    HIPCHK(c, hipStreamSynchronize(st));                 // the small tables below are reused from batch to batch
    uint32_t *d_wl = c->d_tlist.as<uint32_t>();
    int64_t *d_so = reinterpret_cast<int64_t *>(c->d_tlist.as<uint8_t>() + (((size_t)nw * 4 + 7) & ~(size_t)7));
    HIPCHK(c, hipMemcpy(d_wl, wl.data(), (size_t)nw * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_so, st_off.data(), (size_t)nw * 8, hipMemcpyHostToDevice));
    if (phase == 1)
      for (const LongWin &x : lw) {
        const int64_t d1 = (int64_t)n_strips(x.lc) * mv_tw(x.lr) * 64, d2 = (int64_t)n_strips(x.lu) * mv_tw(x.lr + x.lc) * 64;
        HIPCHK(c, hipMemsetAsync(c->d_moves.as<uint32_t>() + reinterpret_cast<const int64_t *>(g_info + 4 * n_generic)[2 * x.k], 0, (size_t)(d1 + d2) * 4, st));
        HIPCHK(c, hipMemsetAsync(d_tiled + x.w, 1, 1, st));
      }
    TileArgs ta;
    ta.wlist = d_wl; ta.st_off = d_so; ta.tstate = c->d_tstate.as<int32_t>(); ta.tiled = d_tiled;
    for (int d = 0; d <= max_d; ++d) {
      ta.d = d;
      if (phase == 1) launch_dp1_tile(a, ta, c->gen, max_tiles, nw, st);
      else launch_dp2_tile(a, ta, c->gen, max_tiles, nw, st);
    }
    return 0;
  };

  if (use_trivial) {
    a.n = n;
    timed_begin(c, 2, st);
    launch_trivial(a, d_done_a, d_triv, d_pkey, use_pack, st);
    launch_partition(d_lists, d_lists2, d_bins, nbins_used, d_chunks, nchunks, d_bin_chunks, d_pkey, d_chunk_need, d_bin_need, d_bin_a1, st);
    timed_end(c, st);
  }
  const uint32_t *d_fused_lists = use_trivial ? d_lists2 : d_lists;

  // ---- fused classes: launch chains on the auxiliary streams (A then B of each bin) ----
  // (k_poa's inputs of every list in ONE launch in front of the chains instead of a launch per list on the list's chain was
  // +1 % on the E. coli batch and -2 % on the yeast -split batch -- no k_poa starts before the whole pass is through --
  // and is not kept: DESIGN.md section 4.2.)
  if (use_fused && n > n_generic) {
    HIPCHK(c, hipEventRecord(c->fork, st));
    const int used = n_used;                           // launch chains = auxiliary streams in use
    for (int k = 0; k < used; ++k) HIPCHK(c, hipStreamWaitEvent(c->aux[k], c->fork, 0));
    if (use_pack && used <= 3) HIPCHK(c, hipStreamWaitEvent(c->aux[3], c->fork, 0));
    int hb_used = 0;
    for (int b : bin_order) {                       // within a chain: most work first
      if (!bin_cnt[(size_t)b]) continue;
      const int bG = cls_G(b / kNT), bR = cls_R(b / kNT), bslot = tier_bytes(b % kNT);
      const int sk = stream_of(b);
      hipStream_t sx = c->aux[sk];
      const int fdebug = std::getenv("ELECTOR_DEBUG_FUSED") ? std::atoi(std::getenv("ELECTOR_DEBUG_FUSED")) : 0;
      if (use_pack && (fdebug & (4 | 8 | 32 | 64 | 128 | 256)) && !poa_debug_built()) {
        static std::atomic<bool> told{false};
        if (!told.exchange(true))
          std::fprintf(stderr, "[elector] ELECTOR_DEBUG_FUSED=%d: k_poa was built without its debug branches; rebuild with "
                               "ELECTOR_HIPCC_FLAGS=-DELECTOR_POA_DEBUG=1 python -m elector_amd.build --force\n", fdebug);
      }
      // the hand-back list of this bin -- or, merged, of the bin's lane-group size (the region of the group's first bin,
      // which its neighbours' regions follow)
      HandGroup *hg = merge_hand ? &hgrp[bG == 8 ? 0 : bG == 16 ? 1 : bG == 32 ? 2 : 3] : nullptr;
      const int hb = hg ? hg->first_bin : b;
      uint32_t *d_hand_list = use_pack ? c->d_hand.as<uint32_t>() + bin_first[(size_t)hb] : nullptr;
      int32_t *d_hand_cnt = use_pack ? reinterpret_cast<int32_t *>(c->d_hand.as<uint32_t>() + n) + bin_slot[(size_t)hb] : nullptr;
      if (use_pack) {
        // the whole window in one kernel; what it cannot take lands on the bin's hand-back list
        const PackGeom pg = pack_geom(b);
        GatherArgs ga;
        ga.list = d_fused_lists + bin_first[(size_t)b];
        ga.nlist = bin_cnt[(size_t)b];
        ga.off = a.off; ga.sym = a.sym; ga.status = d_status;
        ga.done_a = d_done_a; ga.done_b = d_done_b; ga.triv = d_triv;
        ga.pdesc = c->d_pdesc.as<uint4>() + 2 * bin_first[(size_t)b];
        ga.psym = c->d_psym.as<uint32_t>() + psym_first[(size_t)b];
        ga.pstride = pstride[(size_t)b];
        ga.nlist_dev = nullptr;
        timed_begin(c, 2, sx);
        launch_gather(ga, sx);
        timed_end(c, sx);
        PackArgs pa;
        pa.b = a;
        pa.list = ga.list;
        pa.nlist = bin_cnt[(size_t)b];
        pa.pdesc = ga.pdesc;
        pa.psym = ga.psym;
        pa.pstride = ga.pstride;
        pa.slot_bytes = pg.slot;
        pa.done_a = d_done_a;
        pa.done_b = d_done_b;
        pa.triv = d_triv;
        pa.mv_pool = c->d_mvpool.as<uint32_t>();
        for (int k = 0; k < sk; ++k) pa.mv_pool += pool_stream[k];
        pa.mv_tw = pg.tw;
        pa.mv_q = c->d_mvbusy.as<int32_t>() + (size_t)sk * 8 * kPoolStride;
        pa.mv_slots = pool_slots[sk];
        pa.hand = d_hand_list;
        pa.hand_count = d_hand_cnt;
        const int gi = bG == 8 ? 0 : bG == 16 ? 1 : bG == 32 ? 2 : 3;
        pa.far = use_far && hg ? c->d_far.as<uint32_t>() + bin_first[(size_t)hg->first_bin] : nullptr;
        pa.far_count = use_far && hg ? reinterpret_cast<int32_t *>(c->d_far.as<uint32_t>() + n) + gi : nullptr;
        pa.far_cap = use_far && hg ? (int)hg->far_cap : 0;
        pa.nlist_dev = nullptr;
        pa.debug = fdebug;
        pa.keep_graph = c->keep_graph ? 1 : 0;
        pa.stamps = reinterpret_cast<unsigned long long *>(c->d_rowinit.as<uint8_t>() + 1024 + 256 * (size_t)b) + 16;
        timed_begin(c, 4, sx);
        if (launch_poa(pa, bG, bR, sx)) return fail(c, ELECTOR_E_HIP, "k_poa attribute");
        timed_end(c, sx);
      }
      // behind k_poa the two-kernel path only sees the windows handed back: its small launches go to a stream
      // of their own (after this bin's k_poa), so that the chain can start its next k_poa at once
      // (with a single chain -- bench.py --serial, every kernel alone on the chip -- they stay in line)
      const bool hb_own = use_pack && n_chains > 1;
      const int so = hb_own ? 3 : sk;
      if (hg) {
        // merged: remember where this bin's k_poa ends; the group's launch pair follows its last bin
        if (hb_own) {
          if (c->hb_events.size() <= (size_t)hb_used) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->hb_events.push_back(e);
          }
          HIPCHK(c, hipEventRecord(c->hb_events[(size_t)hb_used], sx));
          HIPCHK(c, hipStreamWaitEvent(c->aux[3], c->hb_events[(size_t)hb_used], 0));
          ++hb_used;
        }
        if (b != hg->bins.front()) continue;          // bin_order runs from the last bin down: the group's first bin comes last
        hipStream_t sg = hb_own ? c->aux[3] : sx;
        if (use_far && hg->far_cap > 0) {
          // the group's far-edge windows, gathered and aligned by k_poa<G, 8, true>; what it cannot take either joins
          // the group's hand-back list
          const int gi = hg->G == 8 ? 0 : hg->G == 16 ? 1 : hg->G == 32 ? 2 : 3;
          const int32_t *d_far_cnt = reinterpret_cast<const int32_t *>(c->d_far.as<uint32_t>() + n) + gi;
          GatherArgs ga;
          ga.list = c->d_far.as<uint32_t>() + bin_first[(size_t)hg->first_bin];
          ga.nlist = hg->far_cap;
          ga.nlist_dev = d_far_cnt;
          ga.off = a.off; ga.sym = a.sym; ga.status = d_status;
          ga.done_a = d_done_a; ga.done_b = d_done_b; ga.triv = d_triv;
          ga.pdesc = c->d_pdesc.as<uint4>() + 2 * hg->far_desc_first;
          ga.psym = c->d_psym.as<uint32_t>() + hg->far_psym_first;
          ga.pstride = hg->far_stride;
          timed_begin(c, 2, sg);
          launch_gather(ga, sg);
          timed_end(c, sg);
          PackArgs pa;
          pa.b = a;
          pa.list = ga.list; pa.nlist = hg->far_cap; pa.nlist_dev = d_far_cnt;
          pa.pdesc = ga.pdesc; pa.psym = ga.psym; pa.pstride = ga.pstride;
          pa.slot_bytes = hg->far_slot;
          pa.done_a = d_done_a; pa.done_b = d_done_b; pa.triv = d_triv;
          pa.mv_pool = c->d_mvpool.as<uint32_t>();
          for (int k = 0; k < hg->far_pool; ++k) pa.mv_pool += pool_stream[k];
          pa.mv_tw = hg->far_tw;
          pa.mv_q = c->d_mvbusy.as<int32_t>() + (size_t)hg->far_pool * 8 * kPoolStride;
          pa.mv_slots = pool_slots[hg->far_pool];
          pa.hand = d_hand_list;
          pa.hand_count = d_hand_cnt;
          pa.far = nullptr; pa.far_count = nullptr; pa.far_cap = 0;
          pa.debug = fdebug & ~(32 | 64 | 128 | 256);
          pa.keep_graph = c->keep_graph ? 1 : 0;
          pa.stamps = reinterpret_cast<unsigned long long *>(c->d_rowinit.as<uint8_t>() + 1024 + 256 * (size_t)hb) + 16;
          timed_begin(c, 6, sg);
          if (launch_poa_far(pa, hg->G, sg)) return fail(c, ELECTOR_E_HIP, "k_poa attribute");
          timed_end(c, sg);
        }
        FusedArgs fa;
        fa.b = a;
        fa.grid_blocks = hg->blocks;
        fa.list = d_hand_list;
        fa.nlist_dev = d_hand_cnt;
        fa.nlist = hg->cnt;
        fa.done_a = d_done_a;
        fa.done_b = d_done_b;
        fa.rowinit = reinterpret_cast<int32_t *>(c->d_rowinit.as<uint8_t>() + 1024 + 256 * (size_t)hb);
        fa.debug = fdebug;
        fa.keep_map = c->keep_graph ? 1 : 0;
        fa.triv = nullptr;
        fa.mv_pool = c->d_fmv.as<uint8_t>() + fmv_stream[0] + fmv_stream[1] + fmv_stream[2];   // region 3: the merged launches' own
        fa.slot_bytes = hg->slot_a; fa.mv_tw = hg->tw_a; fa.mv_ns = hg->ns_a;
        timed_begin(c, 0, sg);
        if (launch_fused_a(fa, hg->G, 8, sg)) return fail(c, ELECTOR_E_HIP, "fused kernel attribute");
        timed_end(c, sg);
        fa.slot_bytes = hg->slot_b; fa.mv_tw = hg->tw_b; fa.mv_ns = hg->ns_b;
        timed_begin(c, 1, sg);
        if (launch_fused_b(fa, hg->G, 8, 8, sg)) return fail(c, ELECTOR_E_HIP, "fused kernel attribute");
        timed_end(c, sg);
        continue;
      }
      if (hb_own) {
        if (c->hb_events.size() <= (size_t)hb_used) {
          hipEvent_t e;
          HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
          c->hb_events.push_back(e);
        }
        HIPCHK(c, hipEventRecord(c->hb_events[(size_t)hb_used], sx));
        HIPCHK(c, hipStreamWaitEvent(c->aux[3], c->hb_events[(size_t)hb_used], 0));
        ++hb_used;
        sx = c->aux[3];
      }
      FusedArgs fa;
      fa.b = a;
      fa.grid_blocks = use_pack ? old_grid(b) : 0;
      fa.list = use_pack ? d_hand_list : d_fused_lists + bin_first[(size_t)b];
      fa.nlist_dev = use_pack ? d_hand_cnt : use_trivial ? d_bin_need + bin_slot[(size_t)b] : nullptr;
      fa.nlist = bin_cnt[(size_t)b];
      fa.slot_bytes = bslot;
      fa.done_a = d_done_a;
      fa.done_b = d_done_b;
      fa.rowinit = reinterpret_cast<int32_t *>(c->d_rowinit.as<uint8_t>() + 1024 + 256 * (size_t)b);   // phase stamps (debug)
      fa.debug = fdebug;
      fa.keep_map = c->keep_graph ? 1 : 0;
      fa.triv = use_trivial && !use_pack ? d_triv : nullptr;
      fa.slot_bytes = (int)((bin_need_a[(size_t)b] + 127) & ~(int64_t)127);   // alignment #1: the bin's own maximum
      fa.mv_pool = c->d_fmv.as<uint8_t>();
      for (int k = 0; k < so; ++k) fa.mv_pool += fmv_stream[k];
      (void)fmv_geom(b, false, &fa.mv_tw, &fa.mv_ns);
      timed_begin(c, 0, sx);
      if (launch_fused_a(fa, bG, bR, sx)) return fail(c, ELECTOR_E_HIP, "fused kernel attribute");
      timed_end(c, sx);
      fa.slot_bytes = bslot;
      fa.nlist_dev = use_pack ? d_hand_cnt : nullptr;
      (void)fmv_geom(b, true, &fa.mv_tw, &fa.mv_ns);
      timed_begin(c, 1, sx);
      // Ring depth 8.  (Measured: a 4-deep ring for the 99.8 % of windows that need no more, followed by
      // a second launch for the rest, raises the occupancy but not the throughput, and the second launch
      // costs more than it saves.)
      if (launch_fused_b(fa, bG, bR, 8, sx)) return fail(c, ELECTOR_E_HIP, "fused kernel attribute");
      timed_end(c, sx);
    }
    const int join_n = use_pack ? (int)elector_ctx::kAux : std::min(used, (int)elector_ctx::kAux);
    for (int k = 0; k < join_n; ++k) {
      if (k >= used && k != 3) continue;
      HIPCHK(c, hipEventRecord(c->aux_done[k], c->aux[k]));
    }
    // generic alignment #1 for the host-routed windows runs on the main stream meanwhile
    for (auto &ch : chunks) {
      if (ch.k1 <= ch.k0) continue;
      a.n = ch.k1 - ch.k0;
      a.perm = d_generic + ch.k0;
      a.count_ptr = nullptr;
      timed_begin(c, 2, st);
      const std::vector<LongWin> lw = long_windows(ch.k0, ch.k1);
      if ((rc = run_tiles(lw, 1))) return fail(c, rc, "tiled alignment #1");
      launch_dp1(a, c->gen, st, 16);
      launch_fuse1(a, st);
      if ((rc = run_tiles(lw, 2))) return fail(c, rc, "tiled alignment #2");
      {   // finish the host-routed windows here, beside the fused launch chains: a few long windows on one
          // wavefront each are a long critical path that must not wait for the chains to end
        a.mark_b = d_done_b;
        launch_dp2_list(a, c->gen, st, c->d_gring.as<int32_t>(), gring_block, deep_blocks, 16);
        launch_fuse2(a, st);
        a.mark_b = nullptr;
      }
      timed_end(c, st);
    }
    for (int k = 0; k < join_n; ++k) {
      if (k >= used && k != 3) continue;
      HIPCHK(c, hipStreamWaitEvent(st, c->aux_done[k], 0));
    }
    // everything alignment #2 still owes: the windows the fused kernels handed back
    a.n = n;
    a.perm = nullptr;
    a.count_ptr = nullptr;
    timed_begin(c, 2, st);
    for (int round = 0; round < std::max(1, left_rounds); ++round) {
      if (round > 0) HIPCHK(c, hipMemsetAsync(d_counters, 0, 16, st));            // list length, bump pointer
      a.n = n;
      a.perm = nullptr;
      a.count_ptr = nullptr;
      launch_left_b(a, d_leftb, d_counters, d_done_b, c->d_mv2.as<int64_t>(),
                    reinterpret_cast<unsigned long long *>(d_counters + 2), (unsigned long long)max_dwords,
                    (unsigned long long)bump_dwords, round, round + 1 >= std::max(1, left_rounds) ? 1 : 0, st);
      if (std::getenv("ELECTOR_DEBUG_BINS")) {
        (void)hipStreamSynchronize(st);
        int32_t hc[4] = {0, 0, 0, 0};
        (void)hipMemcpy(hc, d_counters, sizeof hc, hipMemcpyDeviceToHost);
        std::fprintf(stderr, "[elector] round %d: %d windows left for the generic alignment #2\n", round, hc[0]);
        if (hc[0] > 0) {
          std::vector<uint32_t> lw((size_t)std::min(hc[0], 12));
          (void)hipMemcpy(lw.data(), d_leftb, lw.size() * 4, hipMemcpyDeviceToHost);
          std::vector<uint8_t> cls((size_t)n);
          (void)hipMemcpy(cls.data(), c->d_cls.p, (size_t)n, hipMemcpyDeviceToHost);
          for (uint32_t w : lw) {
            int64_t o[4] = {0, 0, 0, 0};
            (void)hipMemcpy(o, c->d_off.as<int64_t>() + 3 * (size_t)w, sizeof o, hipMemcpyDeviceToHost);
            std::fprintf(stderr, "[elector]   window %u: Lr %lld Lc %lld Lu %lld class byte %d\n", w, (long long)(o[1] - o[0]),
                         (long long)(o[2] - o[1]), (long long)(o[3] - o[2]), (int)cls[w]);
          }
        }
      }
      a.perm = d_leftb;
      a.count_ptr = d_counters;
      a.mark_b = d_done_b;
      launch_dp2_list(a, c->gen, st, c->d_gring.as<int32_t>(), gring_block, deep_blocks, 8);
      launch_fuse2(a, st);
      a.mark_b = nullptr;
    }
    timed_end(c, st);
  } else {
    // ---- generic path only (general scoring parameters, or fused kernels disabled) ----
    for (auto &ch : chunks) {
      if (ch.k1 <= ch.k0) continue;
      a.n = ch.k1 - ch.k0;
      a.perm = d_generic + ch.k0;
      a.count_ptr = nullptr;
      timed_begin(c, 0, st);
      const std::vector<LongWin> lw = long_windows(ch.k0, ch.k1);
      if ((rc = run_tiles(lw, 1))) return fail(c, rc, "tiled alignment #1");
      launch_dp1(a, c->gen, st, 1);
      timed_end(c, st);
      timed_begin(c, 2, st);
      launch_fuse1(a, st);
      timed_end(c, st);
      if ((rc = run_tiles(lw, 2))) return fail(c, rc, "tiled alignment #2");
      timed_begin(c, 1, st);
      for (int cls = 0; cls < 2; ++cls) launch_dp2(a, c->gen, cls, st, c->d_gring.as<int32_t>(), gring_block, deep_blocks, 1);
      timed_end(c, st);
      timed_begin(c, 2, st);
      launch_fuse2(a, st);
      timed_end(c, st);
    }
  }
  HIPCHK(c, hipGetLastError());

  if (d_scores) {
    // interleave (score1, score2) per window
    HIPCHK(c, hipMemcpy2DAsync(d_scores, 8, c->d_score1.p, 4, 4, (size_t)n, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipMemcpy2DAsync(d_scores + 1, 8, c->d_score2.p, 4, 4, (size_t)n, hipMemcpyDeviceToDevice, st));
  }
  c->last_n = n;
  c->last_n_generic = n_generic;
  c->last_total = total;
  c->graph_valid = c->keep_graph;
  c->last_ncol = d_ncol;
  c->last_status = d_status;
  if (use_pack && std::getenv("ELECTOR_DEBUG_BINS")) {
    (void)hipStreamSynchronize(st);
    std::vector<int32_t> hc((size_t)kBins, 0);
    (void)hipMemcpy(hc.data(), c->d_hand.as<uint32_t>() + n, (size_t)nbins_used * 4, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[elector] k_poa handed back:");
    for (int b = 0; b < kBins; ++b)
      if (bin_cnt[(size_t)b]) std::fprintf(stderr, " G%dxR%d:%d/%lld", cls_G(b / kNT), cls_R(b / kNT), hc[(size_t)bin_slot[(size_t)b]], (long long)bin_cnt[(size_t)b]);
    std::fprintf(stderr, "\n");
    std::vector<int32_t> ha((size_t)kBins, 0);
    (void)hipMemcpy(ha.data(), d_bin_a1, (size_t)nbins_used * 4, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[elector] k_poa lists (windows that run alignment #1 / all, LDS slot):");
    for (int b = 0; b < kBins; ++b)
      if (bin_cnt[(size_t)b]) {
        const auto pg = pack_geom(b);
        std::fprintf(stderr, " G%dxR%d:%d/%lld,%d", cls_G(b / kNT), cls_R(b / kNT), ha[(size_t)bin_slot[(size_t)b]], (long long)bin_cnt[(size_t)b],
                     pg.slot);
      }
    std::fprintf(stderr, "\n");
  }
  if (std::getenv("ELECTOR_DEBUG_FUSED") && (std::atoi(std::getenv("ELECTOR_DEBUG_FUSED")) & 8)) {
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> hs((size_t)32 * kBins);
    (void)hipMemcpy(hs.data(), c->d_rowinit.as<uint8_t>() + 1024, hs.size() * 8, hipMemcpyDeviceToHost);
    int32_t fc[4] = {0, 0, 0, 0};
    if (use_pack) (void)hipMemcpy(fc, c->d_far.as<uint32_t>() + n, sizeof fc, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[elector] far lists (G8 G16 G32 G64): %d %d %d %d\n", fc[0], fc[1], fc[2], fc[3]);
    for (int b = 0; b < kBins; ++b) {
      const unsigned long long *p = hs.data() + 32 * (size_t)b + 16;
      if (p[1] | p[2] | p[3] | p[4] | p[5] | p[6])
        std::fprintf(stderr, "[elector] bin G%dxR%d leaves k_poa: records beyond the slot %llu, broken path %llu, one far edge %llu, several far edges %llu, "
                             "ordinal rows / steps %llu, refused at the door %llu (of %lld)\n", cls_G(b / kNT), cls_R(b / kNT), p[1], p[2], p[3], p[4], p[5], p[6],
                     (long long)bin_cnt[(size_t)b]);
      if (p[10] | p[11] | p[12] | p[13] | p[14] | p[15])
        std::fprintf(stderr, "[elector]   ... windows with 2 / 3 / 4 / 5 / 6 / 7+ far edges: %llu %llu %llu %llu %llu %llu\n", p[10], p[11], p[12], p[13], p[14], p[15]);
    }
  }
  if (std::getenv("ELECTOR_DEBUG_FUSED") && (std::atoi(std::getenv("ELECTOR_DEBUG_FUSED")) & 4)) {
    (void)hipStreamSynchronize(st);
    int32_t hc[4];
    (void)hipMemcpy(hc, c->d_rowinit.p, sizeof hc, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[elector] windows handed back to the generic alignment #2: %d\n", hc[0]);
    std::vector<unsigned long long> hs((size_t)32 * kBins);
    (void)hipMemcpy(hs.data(), c->d_rowinit.as<uint8_t>() + 1024, hs.size() * 8, hipMemcpyDeviceToHost);
    {
      unsigned long long hist[16] = {0};
      for (int b = 0; b < kBins; ++b) for (int k = 0; k < 16; ++k) hist[k] += hs[32 * (size_t)b + 16 + k];
      unsigned long long one = 0, two = 0;
      for (int b = 0; b < kBins; ++b) { one += hs[32 * (size_t)b + 14]; two += hs[32 * (size_t)b + 15]; }
      std::fprintf(stderr, "[elector] k_fused_b steps on the one-predecessor path %llu, on the two-predecessor path %llu\n", one, two);
      std::fprintf(stderr, "[elector] ring depth needed (max predecessor distance + 2), windows:");
      for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %d:%llu", k, hist[k]);
      std::fprintf(stderr, "\n");
    }
    for (int b = 0; b < kBins; ++b) {
      const unsigned long long *p = hs.data() + 32 * (size_t)b;
      if (p[16 + 15])
        std::fprintf(stderr, "[elector] bin G%dxR%d  k_poa waves %llu: stage %.0f dpA %.0f tbA %.0f fus1 %.0f ord %.0f dpB %.0f tbB %.0f cols+out %.0f | of stage: descriptors %.0f symbols %.0f slot %.0f rest %.0f; traceback #2 rounds %.1f; alignment #2 steps %.1f, two-predecessor %.1f, virtual %.1f  (cycles per wave)\n",
                     cls_G(b / kNT), cls_R(b / kNT), p[31], (double)(p[16] + p[24] + p[25] + p[26]) / p[31], (double)p[17] / p[31], (double)p[18] / p[31],
                     (double)p[19] / p[31], (double)p[20] / p[31], (double)p[21] / p[31], (double)p[22] / p[31], (double)p[23] / p[31],
                     (double)p[24] / p[31], (double)p[25] / p[31], (double)p[26] / p[31], (double)p[16] / p[31], (double)p[27] / p[31],
                     (double)p[28] / p[31], (double)p[29] / p[31], (double)p[30] / p[31]);
      if (!p[4]) continue;
      std::fprintf(stderr, "[elector] bin G%dxR%d/%d  A waves %llu: stage %.0f dp %.0f traceback %.0f fusion %.0f out %.0f | B waves %llu: stage %.0f dp %.0f traceback %.0f fusion %.0f out %.0f  (cycles per wave)\n",
                   cls_G(b / kNT), cls_R(b / kNT), tier_bytes(b % kNT), p[4], (double)p[0] / p[4], (double)p[1] / p[4], (double)p[5] / p[4], (double)p[2] / p[4], (double)p[3] / p[4],
                   p[12], p[12] ? (double)p[8] / p[12] : 0, p[12] ? (double)p[9] / p[12] : 0, p[12] ? (double)p[13] / p[12] : 0,
                   p[12] ? (double)p[10] / p[12] : 0, p[12] ? (double)p[11] / p[12] : 0);
    }
  }
  return ELECTOR_OK;
}

extern "C" int elector_poa_batch_device(elector_ctx *c, int64_t n, const uint8_t *d_bases, const int64_t *off,
                                         uint8_t *d_cols, int32_t *d_ncol, int32_t *d_status, int32_t *d_scores)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n < 0 || n > 0x7fffffff || !off || (n > 0 && (!d_bases || !d_cols || !d_ncol || !d_status)))
    return fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  {   // a bundle search noted for the previous batch goes in front of everything this call queues (offsets, buffers)
    const int rcb = elector_bundles_flush(c);
    if (rcb) return rcb;
  }
  if (n == 0) return run_device_batch(c, 0, d_bases, nullptr, 0, d_cols, d_ncol, d_status, d_scores);
  if (off[0] != 0 || off[3 * n] < 0) return fail(c, ELECTOR_E_INVAL, "off[0] must be 0");
  const int rc = upload_offsets(c, n, off, c->stream);
  if (rc) return fail(c, rc, "offsets upload");
  return run_device_batch(c, n, d_bases, c->d_off.as<int64_t>(), off[3 * n], d_cols, d_ncol, d_status, d_scores);
}

extern "C" int elector_poa_batch_device_offsets(elector_ctx *c, int64_t n, const uint8_t *d_bases, const int64_t *d_off,
                                                 int64_t total, uint8_t *d_cols, int32_t *d_ncol, int32_t *d_status,
                                                 int32_t *d_scores)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n < 0 || n > 0x7fffffff || (n > 0 && (!d_off || !d_bases || !d_cols || !d_ncol || !d_status)))
    return fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  {   // a bundle search noted for the previous batch goes in front of everything this call queues (offsets, buffers)
    const int rcb = elector_bundles_flush(c);
    if (rcb) return rcb;
  }
  return run_device_batch(c, n, d_bases, d_off, total, d_cols, d_ncol, d_status, d_scores);
}

extern "C" int elector_poa_batch(elector_ctx *c, int64_t n, const uint8_t *bases, const int64_t *off, uint8_t *rows,
                                 int64_t rows_cap, int64_t *row_off, int32_t *ncol, int32_t *status, int32_t *scores)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n < 0 || n > 0x7fffffff || !off || !row_off || (n > 0 && (!bases || !rows || !ncol || !status)))
    return fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  {   // a bundle search noted for the previous batch goes in front of everything this call queues (offsets, buffers)
    const int rcb = elector_bundles_flush(c);
    if (rcb) return rcb;
  }
  row_off[0] = 0;
  if (n == 0) return ELECTOR_OK;
  const int64_t total = off[3 * n];
  if (off[0] != 0 || total < 0) return fail(c, ELECTOR_E_INVAL, "off[0] must be 0");
  int rc = c->d_bases.ensure((size_t)total + 64) | c->d_cols.ensure((size_t)3 * total + 64) |
           c->d_ncol.ensure((size_t)n * 4) | c->d_status.ensure((size_t)n * 4) | c->d_scores.ensure((size_t)n * 8) |
           c->d_rowoff.ensure((size_t)(n + 1) * 8);
  if (rc) return fail(c, ELECTOR_E_NOMEM, "device staging");
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(c->d_bases.p, bases, (size_t)total, hipMemcpyHostToDevice, st));
  rc = upload_offsets(c, n, off, st);
  if (rc) return fail(c, rc, "offsets upload");
  rc = run_device_batch(c, n, c->d_bases.as<uint8_t>(), c->d_off.as<int64_t>(), total, c->d_cols.as<uint8_t>(), c->d_ncol.as<int32_t>(),
                        c->d_status.as<int32_t>(), scores ? c->d_scores.as<int32_t>() : nullptr);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(ncol, c->d_ncol.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(status, c->d_status.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (scores) HIPCHK(c, hipMemcpyAsync(scores, c->d_scores.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  bool any_bad = false;
  for (int64_t w = 0; w < n; ++w) {
    if (status[w]) { any_bad = true; ncol[w] = 0; }
    row_off[w + 1] = row_off[w] + 3 * (int64_t)ncol[w];
  }
  if (row_off[n] > rows_cap) return fail(c, ELECTOR_E_INVAL, "rows buffer too small");
  rc = c->d_rows.ensure((size_t)row_off[n] + 64);
  if (rc) return fail(c, ELECTOR_E_NOMEM, "device rows");
  HIPCHK(c, hipMemcpyAsync(c->d_rowoff.p, row_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
  launch_rows(c->d_cols.as<uint8_t>(), c->d_off.as<int64_t>(), c->d_ncol.as<int32_t>(), c->d_rowoff.as<int64_t>(),
              c->d_rows.as<uint8_t>(), n, st);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(rows, c->d_rows.p, (size_t)row_off[n], hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return any_bad ? ELECTOR_E_WINDOW : ELECTOR_OK;
}
