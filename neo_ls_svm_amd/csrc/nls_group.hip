// Several GPUs behind one handle and one host call (nls_ctx_create(devs, ndev), "multi-GPU is internal to the ctx"; 8(e)),
// and the gamma x sigma leave-one-out grid of BASELINE config 5 behind the C ABI (8(b): sigmas[Sg] / sigma_idx / loo_errors[Sg G]; 8(d)).
//
// A group is nothing but N ordinary contexts - one per device - joined by an RCCL communicator, plus a fan-out: every group call starts one
// short-lived host thread per device (rank 0 runs on the calling thread) and each thread makes the SAME per-rank call a process-per-GPU launch
// makes (nls_primal_fit on its row block; nls_primal_predict on its query rows).  There is no second implementation of the sharded fit.
#include <cmath>
#include <functional>
#include <limits>
#include <mutex>

#include "nls_host.h"

struct nls_group_factor {
  std::vector<nls_factor*> f;  // one inverse factor per member context
  int D = 0;
};

struct nls_group {
  std::vector<nls_ctx*> ctx;
  std::vector<int> devices;
  std::string err;
  std::vector<nls_group_factor*> factors;
  // 1 + rank of the first member that failed outside a status vote during the current call: the other members stop waiting for it (comm_wait)
  std::atomic<int> abort{0};
  bool rejoin = false;  // a member's communicator was aborted: the next collective call joins the members to a new one first
  // host buffers of ranks > 0 in the sigma-sharded grid (their incumbents' full results), kept between calls: fresh pages cost first-touch faults
  std::vector<std::vector<double>> grid_rows, grid_L, grid_small;
};

static std::string g_group_create_error;
static std::mutex g_group_create_mutex;

static int gfail(nls_group* g, int code, const char* fmt, ...) {
  char buf[1200];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (g)
    g->err = buf;
  else
    g_group_create_error = buf;
  return code;
}

// fn(rank) on every member: rank 0 on the calling thread, the others on threads of their own.  Returns the failure of the rank that failed
// FIRST of its own accord (not one that merely stopped because a peer had failed) and records its message ("rank r: ...").
// One member failing must not leave the others waiting in a collective and this call in join(): failures that the
// ranks' status votes carry end the call on every member at the same point (comm_vote); a member that fails any other way - an exception on
// its thread included - raises the group's abort flag, which the others' bounded waits poll (comm_wait): they abort their communicators and
// return NLS_ERR_COMM.  The group then joins a fresh communicator at its next collective call.
static int fan_out(nls_group* g, const std::function<int(int)>& fn) {
  const int nd = (int)g->ctx.size();
  std::vector<int> rc((size_t)nd, NLS_OK);
  g->abort.store(0, std::memory_order_release);
  auto run = [&](int r) {
    nls_ctx* c = g->ctx[(size_t)r];
    int code = NLS_OK;
    try {
      code = fn(r);
    } catch (const std::exception& e) {
      code = fail(c, NLS_ERR_HIP, "exception on the rank's host thread: %s", e.what());
    } catch (...) {
      code = fail(c, NLS_ERR_HIP, "unknown exception on the rank's host thread");
    }
    if (code != NLS_OK && !c->voted_out) {
      int none = 0;
      g->abort.compare_exchange_strong(none, r + 1, std::memory_order_acq_rel);
    }
    rc[(size_t)r] = code;
  };
  std::vector<std::thread> th;
  th.reserve((size_t)nd);
  for (int r = 1; r < nd; ++r) {
    try {
      th.emplace_back(run, r);
    } catch (const std::exception& e) {  // no thread for this rank: it fails, the others are told
      rc[(size_t)r] = fail(g->ctx[(size_t)r], NLS_ERR_HIP, "cannot start the rank's host thread: %s", e.what());
      int none = 0;
      g->abort.compare_exchange_strong(none, r + 1, std::memory_order_acq_rel);
    }
  }
  run(0);
  for (auto& t : th) t.join();
  for (nls_ctx* c : g->ctx)
    if (c->comm_broken) g->rejoin = true;
  int pick = -1;
  const int flagged = g->abort.load(std::memory_order_acquire) - 1;
  if (flagged >= 0 && rc[(size_t)flagged] != NLS_OK) pick = flagged;
  for (int r = 0; r < nd && pick < 0; ++r)  // a voted failure: the member that returned its own code
    if (rc[(size_t)r] != NLS_OK && rc[(size_t)r] != NLS_ERR_COMM && !g->ctx[(size_t)r]->vote_victim) pick = r;
  for (int r = 0; r < nd && pick < 0; ++r)
    if (rc[(size_t)r] != NLS_OK) pick = r;
  if (pick < 0) return NLS_OK;
  const char* m = nls_last_error(g->ctx[(size_t)pick]);
  return gfail(g, rc[(size_t)pick], "rank %d of %d (device %d): %s", pick, nd, g->devices[(size_t)pick], m && m[0] ? m : "failed");
}

// Joins the member contexts to a fresh communicator (group creation; again after a failure that cost a member its communicator).
static int group_join(nls_group* g) {
  const int nd = (int)g->ctx.size();
  for (nls_ctx* c : g->ctx) (void)nls_comm_destroy(c);  // (members that still hold the old one: every collective was waited for, nothing is pending)
  unsigned char id[NLS_COMM_ID_BYTES];
  int rc = nls_comm_get_unique_id(id);
  if (rc != NLS_OK) return gfail(g, rc, "communicator id: %s", nls_last_error(nullptr));
  rc = fan_out(g, [&](int r) { return nls_comm_init_rank(g->ctx[(size_t)r], id, r, nd); });
  if (rc == NLS_OK) g->rejoin = false;
  return rc;
}

extern "C" const char* nls_group_last_error(const nls_group* g) {
  if (g) return g->err.c_str();
  static thread_local std::string copy;  // (nls_group_create may be writing the shared message on another thread)
  std::lock_guard<std::mutex> lock(g_group_create_mutex);
  copy = g_group_create_error;
  return copy.c_str();
}
extern "C" int nls_group_size(const nls_group* g) { return g ? (int)g->ctx.size() : 0; }
extern "C" nls_ctx* nls_group_ctx(nls_group* g, int rank) { return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[(size_t)rank] : nullptr; }

extern "C" void nls_group_destroy(nls_group* g) {
  if (!g) return;
  for (nls_group_factor* gf : g->factors) {
    for (size_t r = 0; r < gf->f.size(); ++r)
      if (gf->f[r]) (void)nls_factor_destroy(g->ctx[r], gf->f[r]);
    delete gf;
  }
  // communicators first, all ranks at the same time (ncclCommDestroy is a collective in spirit: a rank must not tear down under a peer)
  if (g->ctx.size() > 1) (void)fan_out(g, [&](int r) { return nls_comm_destroy(g->ctx[(size_t)r]); });
  for (nls_ctx* c : g->ctx) nls_ctx_destroy(c);
  delete g;
}

extern "C" int nls_group_create(const int* devices, int ndev, nls_group** out) {
  std::lock_guard<std::mutex> lock(g_group_create_mutex);
  if (!out) return gfail(nullptr, NLS_ERR_ARG, "group output pointer is NULL");
  if (!devices || ndev < 1 || ndev > 64) return gfail(nullptr, NLS_ERR_ARG, "devices NULL or ndev outside [1, 64] (ndev = %d)", ndev);
  nls_group* g = new nls_group();
  for (int r = 0; r < ndev; ++r) {
    nls_ctx* c = nullptr;
    const int rc = nls_ctx_create(devices[r], &c);
    if (rc != NLS_OK) {
      gfail(nullptr, rc, "context of rank %d (device %d): %s", r, devices[r], nls_last_error(nullptr));
      nls_group_destroy(g);
      return rc;
    }
    g->ctx.push_back(c);
    g->devices.push_back(devices[r]);
  }
  if (ndev > 1) {
    for (nls_ctx* c : g->ctx) c->abort_flag = &g->abort;
    const int rc = group_join(g);
    if (rc != NLS_OK) {
      g_group_create_error = g->err;
      nls_group_destroy(g);
      return rc;
    }
  }
  g->grid_rows.resize((size_t)ndev);
  g->grid_L.resize((size_t)ndev);
  g->grid_small.resize((size_t)ndev);
  *out = g;
  return NLS_OK;
}

// Bulk inputs of a group call are host pointers, or device pointers that every member context can dereference (all members on that device).
static int check_bulk_pointer(nls_group* g, const void* p, const char* name) {
  if (!p) return NLS_OK;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return NLS_OK;  // plain host memory
  }
  if (attr.type != hipMemoryTypeDevice) return NLS_OK;
  for (int dev : g->devices)
    if (dev != attr.device)
      return gfail(g, NLS_ERR_ARG, "%s is resident on device %d but the group spans other devices: pass host pointers (each rank uploads its own row block)",
                   name, attr.device);
  return NLS_OK;
}

static inline int64_t block_lo(int64_t n, int r, int nd) { return n * r / nd; }

extern "C" int nls_group_primal_fit(nls_group* g, const nls_primal_fit_args* a) {
  if (!g) return NLS_ERR_ARG;
  if (!a) return gfail(g, NLS_ERR_ARG, "args is NULL");
  const int nd = (int)g->ctx.size();
  if (nd == 1) {
    const int rc = nls_primal_fit(g->ctx[0], a);
    if (rc != NLS_OK) g->err = nls_last_error(g->ctx[0]);
    return rc;
  }
  if (!a->X || !a->y || !a->s || !a->gammas) return gfail(g, NLS_ERR_ARG, "X, y, s and gammas must not be NULL");
  if (a->n < nd) return gfail(g, NLS_ERR_ARG, "a group of %d devices needs at least %d rows (n = %ld)", nd, nd, (long)a->n);
  if (a->d < 1 || a->D < 1 || a->G < 1) return gfail(g, NLS_ERR_ARG, "d, D and G must be >= 1");
  NLSCHK(check_bulk_pointer(g, a->X, "X"));
  NLSCHK(check_bulk_pointer(g, a->y, "y"));
  NLSCHK(check_bulk_pointer(g, a->s, "s"));
  if (g->rejoin) NLSCHK(group_join(g));  // an earlier failure cost a member its communicator
  return fan_out(g, [&](int r) {
    const int64_t lo = block_lo(a->n, r, nd), hi = block_lo(a->n, r + 1, nd);
    nls_primal_fit_args b = *a;
    b.X = a->X + lo * a->d;
    b.y = a->y + lo;
    b.s = a->s + lo;
    b.n = hi - lo;
    auto rows = [&](double* p) { return p ? p + lo : nullptr; };
    b.loo_residuals = rows(a->loo_residuals);
    b.loo_leverage = rows(a->loo_leverage);
    b.loo_std = rows(a->loo_std);
    b.residuals = rows(a->residuals);
    if (r != 0) {  // replicated outputs: identical on every rank (all-reduced / broadcast inside the fit) - rank 0 writes them
      b.beta = b.L = b.lam = b.loo_errors = b.objective = b.loo_score = b.timings = nullptr;
      b.gamma_index = b.finished = nullptr;
    }
    return nls_primal_fit(g->ctx[(size_t)r], &b);
  });
}

extern "C" int nls_group_factor_create(nls_group* g, const double* L, int D, nls_group_factor** out) {
  if (!g) return NLS_ERR_ARG;
  if (!L || !out || D < 1) return gfail(g, NLS_ERR_ARG, "nls_group_factor_create: L / factor NULL or D < 1");
  NLSCHK(check_bulk_pointer(g, L, "L"));
  nls_group_factor* gf = new nls_group_factor();
  gf->D = D;
  gf->f.assign(g->ctx.size(), nullptr);
  const int rc = fan_out(g, [&](int r) { return nls_factor_create(g->ctx[(size_t)r], L, D, &gf->f[(size_t)r]); });
  if (rc != NLS_OK) {
    for (size_t r = 0; r < gf->f.size(); ++r)
      if (gf->f[r]) (void)nls_factor_destroy(g->ctx[r], gf->f[r]);
    delete gf;
    return rc;
  }
  g->factors.push_back(gf);
  *out = gf;
  return NLS_OK;
}

extern "C" int nls_group_factor_destroy(nls_group* g, nls_group_factor* gf) {
  if (!g) return NLS_ERR_ARG;
  if (!gf) return NLS_OK;
  auto it = std::find(g->factors.begin(), g->factors.end(), gf);
  if (it == g->factors.end()) return gfail(g, NLS_ERR_ARG, "nls_group_factor_destroy: not a live factor of this group");
  g->factors.erase(it);
  int first = NLS_OK;
  for (size_t r = 0; r < gf->f.size(); ++r) {
    const int rc = gf->f[r] ? nls_factor_destroy(g->ctx[r], gf->f[r]) : NLS_OK;
    if (rc != NLS_OK && first == NLS_OK) first = gfail(g, rc, "rank %zu: %s", r, nls_last_error(g->ctx[r]));
  }
  delete gf;
  return first;
}

extern "C" int nls_group_primal_predict(nls_group* g, const double* X, int64_t m, int d, const double* shift, const double* scale, const double* B,
                                        int D, const double* beta, const double* L, const nls_group_factor* gf, double* yhat, double* sigma) {
  if (!g) return NLS_ERR_ARG;
  if (!X || m < 0 || d < 1) return gfail(g, NLS_ERR_ARG, "X NULL, m < 0 or d < 1");
  if (gf && std::find(g->factors.begin(), g->factors.end(), gf) == g->factors.end())
    return gfail(g, NLS_ERR_ARG, "factor is not a live handle of this group");
  NLSCHK(check_bulk_pointer(g, X, "X"));
  NLSCHK(check_bulk_pointer(g, beta, "beta"));
  NLSCHK(check_bulk_pointer(g, L, "L"));
  NLSCHK(check_bulk_pointer(g, yhat, "yhat"));
  NLSCHK(check_bulk_pointer(g, sigma, "sigma"));
  const int nd = (int)g->ctx.size();
  return fan_out(g, [&](int r) {
    const int64_t lo = block_lo(m, r, nd), hi = block_lo(m, r + 1, nd);
    if (hi == lo) return (int)NLS_OK;
    return nls_primal_predict(g->ctx[(size_t)r], X + lo * d, hi - lo, d, shift, scale, B, D, beta, gf ? nullptr : L, gf ? gf->f[(size_t)r] : nullptr,
                              yhat ? yhat + lo : nullptr, sigma ? sigma + lo : nullptr);
  });
}

// ------------------------------------------------------------------------------------------------
// gamma x sigma grid
// ------------------------------------------------------------------------------------------------
namespace {

// numpy.argmin: first minimum; a NaN wins (the first NaN is returned).
int argmin_np(const double* v, int n) {
  int opt = 0;
  for (int i = 0; i < n; ++i) {
    if (std::isnan(v[i])) return i;
    if (v[i] < v[opt]) opt = i;
  }
  return opt;
}
// numpy.nanmin of a row (NaN when every entry is NaN)
double nanmin_np(const double* v, int n) {
  double m = std::numeric_limits<double>::quiet_NaN();
  for (int i = 0; i < n; ++i)
    if (!std::isnan(v[i]) && (std::isnan(m) || v[i] < m)) m = v[i];
  return m;
}

struct GridLocal {  // what one rank's pass over its sigmas leaves behind
  std::vector<double> table, objective, seconds, timings;  // Sg x G (rows of the other ranks' sigmas: 0), Sg, NLS_NUM_TIMINGS
  int best_k = -1, best_opt = -1, finished_count = 0;
  double best_score = 0.0;
  std::vector<double> best_lam;
};

// This rank's sigmas in visiting order: |ln sigma| ascending, ties by index.
std::vector<int> visiting_order(const double* sigmas, int Sg, int rank, int world) {
  std::vector<int> order;
  for (int k = rank; k < Sg; k += world) order.push_back(k);
  std::stable_sort(order.begin(), order.end(), [&](int i, int j) {
    const double a = std::fabs(std::log(sigmas[i])), b = std::fabs(std::log(sigmas[j]));
    return a < b || (a == b && i < j);
  });
  return order;
}

// The pass of one rank (steps 1 and 2 of the header's description).  Finished fits write into `a`'s beta / L / row outputs directly: a later
// finished sigma is strictly better than the incumbent, so what is there at the end is the incumbent's.
int grid_local_pass(nls_ctx* ctx, const nls_primal_fit_args* a, const double* sigmas, int Sg, int rank, int world, GridLocal* out) {
  const int G = a->G;
  const size_t D1 = (size_t)a->D + 1, nB = (size_t)a->d * (size_t)a->D;
  out->table.assign((size_t)Sg * G, 0.0);
  out->objective.assign((size_t)Sg * G, 0.0);
  out->seconds.assign((size_t)Sg, 0.0);
  out->timings.assign(NLS_NUM_TIMINGS, 0.0);
  out->best_lam.assign(D1, 0.0);
  std::vector<double> Bs(nB), lam(D1), tm(NLS_NUM_TIMINGS);
  // host inputs are uploaded ONCE for the whole grid (every inner fit would otherwise stage them again)
  nls_primal_fit_args b = *a;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  NLSCHK(resident(ctx, "grid.X", a->X, (size_t)a->n * a->d, &b.X));
  NLSCHK(resident(ctx, "grid.y", a->y, (size_t)a->n, &b.y));
  NLSCHK(resident(ctx, "grid.s", a->s, (size_t)a->n, &b.s));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int k : visiting_order(sigmas, Sg, rank, world)) {
    for (size_t i = 0; i < nB; ++i) Bs[i] = a->B[i] / sigmas[k];
    int32_t opt = -1, finished = 0;
    std::fill(tm.begin(), tm.end(), 0.0);
    b.B = Bs.data();
    b.flags = (a->flags & NLS_FIT_RESIDUALS_FROM_SWEEP) | (out->best_k < 0 ? 0 : NLS_FIT_FINISH_IF_BELOW);
    b.finish_below = out->best_k < 0 ? 0.0 : out->best_score;
    b.lam = lam.data();
    b.loo_errors = out->table.data() + (size_t)k * G;
    b.objective = out->objective.data() + (size_t)k * G;
    b.gamma_index = &opt;
    b.finished = &finished;
    b.timings = tm.data();
    NLSCHK(nls_primal_fit(ctx, &b));
    out->seconds[(size_t)k] = tm[NLS_T_TOTAL];
    for (int t = 0; t < NLS_NUM_TIMINGS; ++t) out->timings[(size_t)t] += tm[(size_t)t];
    const double score = out->objective[(size_t)k * G + opt];
    if (finished) ++out->finished_count;
    // (finish_below is strict: a sigma that only TIES the incumbent is not finished; the selection resolves exact ties towards the finished
    // incumbent, so the full result is never lost to a tie)
    if (finished && (out->best_k < 0 || score < out->best_score)) {
      out->best_k = k;
      out->best_opt = opt;
      out->best_score = score;
      out->best_lam = lam;
    }
  }
  return NLS_OK;
}

// Steps 3 / 4 on tables that hold every OWNED sigma (owned[k]); incumbent >= 0: the unmerged single-rank tie rule.
void grid_select(const std::vector<double>& objective, const std::vector<char>& owned, int Sg, int G, int incumbent, int* k_opt, int* g_opt) {
  std::vector<double> col_min((size_t)Sg);
  for (int k = 0; k < Sg; ++k) col_min[(size_t)k] = owned[(size_t)k] ? nanmin_np(objective.data() + (size_t)k * G, G) : std::numeric_limits<double>::infinity();
  int ko = argmin_np(col_min.data(), Sg);
  if (incumbent >= 0 && col_min[(size_t)incumbent] == col_min[(size_t)ko]) ko = incumbent;
  *k_opt = ko;
  *g_opt = argmin_np(objective.data() + (size_t)ko * G, G);
}

int grid_check_args(nls_ctx* ctx, nls_group* g, const nls_primal_fit_args* a, const nls_sigma_grid* gr) {
  auto bad = [&](const char* msg) { return g ? gfail(g, NLS_ERR_ARG, "%s", msg) : fail(ctx, NLS_ERR_ARG, "%s", msg); };
  if (!a || !gr) return bad("args / grid is NULL");
  if (!a->X || !a->y || !a->s || !a->gammas || !a->B) return bad("X, y, s, B and gammas must not be NULL");
  if (a->n < 1 || a->G < 1 || a->d < 1 || a->D < 1) return bad("n, d, D and G must be >= 1");
  if (a->gamma_index_in != -1 || (a->flags & ~NLS_FIT_RESIDUALS_FROM_SWEEP) != 0)
    return bad("the grid selects gamma itself: gamma_index_in must be -1 and flags 0 (or NLS_FIT_RESIDUALS_FROM_SWEEP)");
  if (!gr->sigmas || gr->Sg < 1) return bad("sigmas NULL or Sg < 1");
  if (!gr->sigma_index || !gr->gamma_index || !gr->best_valid) return bad("sigma_index, gamma_index and best_valid must not be NULL");
  for (int k = 0; k < gr->Sg; ++k)
    if (!(gr->sigmas[k] > 0.0) || !std::isfinite(gr->sigmas[k])) return bad("sigmas must be positive and finite");
  return NLS_OK;
}

}  // namespace

extern "C" int nls_grid_visiting_order(const double* sigmas, int Sg, int rank, int world, int32_t* order) {
  if (!sigmas || !order || Sg < 1 || world < 1 || rank < 0 || rank >= world) return -1;
  const std::vector<int> o = visiting_order(sigmas, Sg, rank, world);
  for (size_t i = 0; i < o.size(); ++i) order[i] = o[i];
  return (int)o.size();
}

extern "C" int nls_grid_select(const double* objective, const unsigned char* owned, int Sg, int G, int incumbent, int32_t* sigma_index, int32_t* gamma_index) {
  if (!objective || !owned || !sigma_index || !gamma_index || Sg < 1 || G < 1 || incumbent >= Sg) return NLS_ERR_ARG;
  std::vector<double> obj(objective, objective + (size_t)Sg * G);
  std::vector<char> own(owned, owned + Sg);
  int k = 0, g = 0;
  grid_select(obj, own, Sg, G, incumbent, &k, &g);
  *sigma_index = k;
  *gamma_index = g;
  return NLS_OK;
}

extern "C" int nls_primal_fit_grid(nls_ctx* ctx, const nls_primal_fit_args* a, const nls_sigma_grid* gr) {
  if (!ctx) return NLS_ERR_ARG;
  NLSCHK(grid_check_args(ctx, nullptr, a, gr));
  const int Sg = gr->Sg, G = a->G;
  const int world = gr->world > 1 ? gr->world : 1, rank = gr->world > 1 ? gr->rank : 0;
  if (rank < 0 || rank >= world) return fail(ctx, NLS_ERR_ARG, "grid rank %d outside world %d", rank, world);
  if (world > 1 && multi_rank(ctx))
    return fail(ctx, NLS_ERR_ARG, "the grid shards sigmas, not rows: the fitting context must not be in a communicator (put the communicator "
                                  "of the merge on a second context, grid->merge)");
  if (gr->merge == ctx) return fail(ctx, NLS_ERR_ARG, "grid->merge must be a context other than the fitting context");
  GridLocal loc;
  const bool merged = world > 1 && gr->merge != nullptr;
  {
    // A rank whose own sigmas failed must not leave its peers in the merge's all-reduces: its status goes into a vote on the merge
    // communicator first (comm_vote, nls_host.h), and every rank of the grid leaves together.  (Without a merge context no rank waits for another.)
    const int rc_local = grid_local_pass(ctx, a, gr->sigmas, Sg, rank, world, &loc);
    if (merged) {
      gr->merge->voted_out = gr->merge->vote_victim = false;
      const int rc = comm_vote(gr->merge, rc_local, "before the merge of the sigma grid's tables");
      if (rc != NLS_OK) return rc_local != NLS_OK ? rc_local : fail(ctx, rc, "%s", nls_last_error(gr->merge));
    } else {
      NLSCHK(rc_local);
    }
  }
  std::vector<char> owned((size_t)Sg, 0);
  if (merged) {  // step 3: every sigma is owned by one rank, the others hold zeros -> a sum all-reduce in pieces of NLS_COMM_UTIL_MAX
    auto sum_all = [&](std::vector<double>& v) -> int {
      for (size_t off = 0; off < v.size(); off += NLS_COMM_UTIL_MAX) {
        const int rc = nls_comm_allreduce(gr->merge, v.data() + off, std::min<size_t>(NLS_COMM_UTIL_MAX, v.size() - off), 0);
        if (rc != NLS_OK) return fail(ctx, rc, "merge of the grid tables: %s", nls_last_error(gr->merge));
      }
      return NLS_OK;
    };
    NLSCHK(sum_all(loc.table));
    NLSCHK(sum_all(loc.objective));
    NLSCHK(sum_all(loc.seconds));
    std::fill(owned.begin(), owned.end(), 1);
  } else {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (int k = 0; k < Sg; ++k) {
      owned[(size_t)k] = (k % world) == rank;
      if (!owned[(size_t)k]) {
        std::fill(loc.table.begin() + (size_t)k * G, loc.table.begin() + (size_t)(k + 1) * G, nan);
        std::fill(loc.objective.begin() + (size_t)k * G, loc.objective.begin() + (size_t)(k + 1) * G, nan);
      }
    }
  }
  int k_opt = 0, g_opt = 0;
  grid_select(loc.objective, owned, Sg, G, merged ? -1 : loc.best_k, &k_opt, &g_opt);
  if (gr->loo_errors) std::memcpy(gr->loo_errors, loc.table.data(), sizeof(double) * (size_t)Sg * G);
  if (gr->objective) std::memcpy(gr->objective, loc.objective.data(), sizeof(double) * (size_t)Sg * G);
  if (gr->seconds) std::memcpy(gr->seconds, loc.seconds.data(), sizeof(double) * (size_t)Sg);
  if (gr->timings) std::memcpy(gr->timings, loc.timings.data(), sizeof(double) * NLS_NUM_TIMINGS);
  if (gr->finished_count) *gr->finished_count = loc.finished_count;
  *gr->sigma_index = k_opt;
  *gr->gamma_index = g_opt;
  *gr->best_valid = (loc.best_k == k_opt) ? 1 : 0;
  if ((k_opt % world) == rank || merged) {  // the winning sigma's curve
    if (a->loo_errors) std::memcpy(a->loo_errors, loc.table.data() + (size_t)k_opt * G, sizeof(double) * G);
    if (a->objective) std::memcpy(a->objective, loc.objective.data() + (size_t)k_opt * G, sizeof(double) * G);
  }
  if (a->gamma_index) *a->gamma_index = g_opt;
  if (a->finished) *a->finished = *gr->best_valid;
  if (a->lam && loc.best_k == k_opt) std::memcpy(a->lam, loc.best_lam.data(), sizeof(double) * ((size_t)a->D + 1));
  if (a->timings) std::memcpy(a->timings, loc.timings.data(), sizeof(double) * NLS_NUM_TIMINGS);
  return NLS_OK;
}

extern "C" int nls_group_primal_fit_grid(nls_group* g, const nls_primal_fit_args* a, const nls_sigma_grid* gr) {
  if (!g) return NLS_ERR_ARG;
  NLSCHK(grid_check_args(nullptr, g, a, gr));
  if (gr->world > 1 || gr->rank != 0 || gr->merge) return gfail(g, NLS_ERR_ARG, "a group deals the sigmas over its own devices: grid->rank / world / merge must be 0 / 1 / NULL");
  const int nd = (int)g->ctx.size();
  if (nd == 1) {
    const int rc = nls_primal_fit_grid(g->ctx[0], a, gr);
    if (rc != NLS_OK) g->err = nls_last_error(g->ctx[0]);
    return rc;
  }
  NLSCHK(check_bulk_pointer(g, a->X, "X"));
  NLSCHK(check_bulk_pointer(g, a->y, "y"));
  NLSCHK(check_bulk_pointer(g, a->s, "s"));
  const int Sg = gr->Sg, G = a->G;
  const size_t n = (size_t)a->n, D1 = (size_t)a->D + 1;
  std::vector<GridLocal> loc((size_t)nd);
  // ranks > 0 keep their incumbent's full result in the group's own host buffers (rank 0 writes into the caller's)
  struct Priv {
    double *beta = nullptr, *L = nullptr, *rows = nullptr, score = 0.0;
  };
  std::vector<Priv> priv((size_t)nd);
  // (sized from the outputs actually requested; a buffer more than twice what this call needs - n or D dropped - is given back)
  const bool any_rows = a->loo_residuals || a->loo_leverage || a->loo_std || a->residuals;
  auto fit = [](std::vector<double>& v, size_t need) {
    if (v.size() < need || v.capacity() > 2 * need + 1024) {
      std::vector<double>().swap(v);
      v.resize(need);
    }
  };
  for (int r = 1; r < nd; ++r) {
    auto& rows = g->grid_rows[(size_t)r];
    auto& small = g->grid_small[(size_t)r];
    auto& Lb = g->grid_L[(size_t)r];
    fit(rows, any_rows ? 4 * n : 0);
    fit(small, 2 * D1);
    fit(Lb, a->L ? 2 * D1 * D1 : 0);
    priv[(size_t)r].rows = any_rows ? rows.data() : nullptr;
    priv[(size_t)r].beta = small.data();
    priv[(size_t)r].L = a->L ? Lb.data() : nullptr;
  }
  int rc = fan_out(g, [&](int r) {
    nls_ctx* c = g->ctx[(size_t)r];
    nls_primal_fit_args b = *a;
    if (r != 0) {
      const Priv& p = priv[(size_t)r];
      b.beta = a->beta ? p.beta : nullptr;
      b.L = p.L;
      b.loo_residuals = a->loo_residuals ? p.rows : nullptr;
      b.loo_leverage = a->loo_leverage ? p.rows + n : nullptr;
      b.loo_std = a->loo_std ? p.rows + 2 * n : nullptr;
      b.residuals = a->residuals ? p.rows + 3 * n : nullptr;
    }
    b.loo_score = a->loo_score ? (r == 0 ? a->loo_score : &priv[(size_t)r].score) : nullptr;
    c->solo = true;  // the member contexts share a communicator; here each fits all rows on its own
    const int rc1 = grid_local_pass(c, &b, gr->sigmas, Sg, r, nd, &loc[(size_t)r]);
    c->solo = false;
    return rc1;
  });
  if (rc != NLS_OK) return rc;
  // step 3 on the host: the tables of the ranks are disjoint by rows
  std::vector<double> table((size_t)Sg * G), objective((size_t)Sg * G), seconds((size_t)Sg), timings(NLS_NUM_TIMINGS, 0.0);
  int finished_count = 0;
  for (int k = 0; k < Sg; ++k) {
    const GridLocal& l = loc[(size_t)(k % nd)];
    std::memcpy(table.data() + (size_t)k * G, l.table.data() + (size_t)k * G, sizeof(double) * G);
    std::memcpy(objective.data() + (size_t)k * G, l.objective.data() + (size_t)k * G, sizeof(double) * G);
    seconds[(size_t)k] = l.seconds[(size_t)k];
  }
  for (int r = 0; r < nd; ++r) {
    finished_count += loc[(size_t)r].finished_count;
    for (int t = 0; t < NLS_NUM_TIMINGS; ++t) timings[(size_t)t] += loc[(size_t)r].timings[(size_t)t];
  }
  std::vector<char> owned((size_t)Sg, 1);
  int k_opt = 0, g_opt = 0;
  grid_select(objective, owned, Sg, G, -1, &k_opt, &g_opt);
  const int owner = k_opt % nd;
  GridLocal& lw = loc[(size_t)owner];
  if (lw.best_k != k_opt) {
    // An exact tie between two sigmas of the owner left its (equally good) incumbent finished instead of the smaller index the merged rule
    // selects: fit the selected sigma once more, unconditionally, into the owner's buffers.
    nls_ctx* c = g->ctx[(size_t)owner];
    std::vector<double> Bs((size_t)a->d * (size_t)a->D), lam(D1), cur(G), obj(G);
    for (size_t i = 0; i < Bs.size(); ++i) Bs[i] = a->B[i] / gr->sigmas[k_opt];
    nls_primal_fit_args b = *a;
    b.B = Bs.data();
    b.gamma_index_in = g_opt;
    if (owner != 0) {
      const Priv& p = priv[(size_t)owner];
      b.beta = a->beta ? p.beta : nullptr;
      b.L = p.L;
      b.loo_residuals = a->loo_residuals ? p.rows : nullptr;
      b.loo_leverage = a->loo_leverage ? p.rows + n : nullptr;
      b.loo_std = a->loo_std ? p.rows + 2 * n : nullptr;
      b.residuals = a->residuals ? p.rows + 3 * n : nullptr;
      b.loo_score = a->loo_score ? &priv[(size_t)owner].score : nullptr;
    }
    b.lam = lam.data();
    b.loo_errors = cur.data();
    b.objective = obj.data();
    b.gamma_index = nullptr;
    b.finished = nullptr;
    b.timings = nullptr;
    c->solo = true;
    rc = nls_primal_fit(c, &b);
    c->solo = false;
    if (rc != NLS_OK) return gfail(g, rc, "rank %d (re-fit of the tied winner): %s", owner, nls_last_error(c));
    lw.best_k = k_opt;
    lw.best_lam = lam;
  }
  if (owner != 0) {  // the winner's full result lives in the owner's private buffers: hand it to the caller
    const Priv& p = priv[(size_t)owner];
    if (a->beta) std::memcpy(a->beta, p.beta, sizeof(double) * 2 * D1);
    if (a->L) {  // the defined (upper, row-major) triangle only
      const double2* src = reinterpret_cast<const double2*>(p.L);
      double2* dst = reinterpret_cast<double2*>(a->L);
      for (size_t i = 0; i < D1; ++i) std::memcpy(dst + i * D1 + i, src + i * D1 + i, sizeof(double2) * (D1 - i));
    }
    if (a->loo_residuals) std::memcpy(a->loo_residuals, p.rows, sizeof(double) * n);
    if (a->loo_leverage) std::memcpy(a->loo_leverage, p.rows + n, sizeof(double) * n);
    if (a->loo_std) std::memcpy(a->loo_std, p.rows + 2 * n, sizeof(double) * n);
    if (a->residuals) std::memcpy(a->residuals, p.rows + 3 * n, sizeof(double) * n);
    if (a->loo_score) *a->loo_score = p.score;
  }
  if (gr->loo_errors) std::memcpy(gr->loo_errors, table.data(), sizeof(double) * table.size());
  if (gr->objective) std::memcpy(gr->objective, objective.data(), sizeof(double) * objective.size());
  if (gr->seconds) std::memcpy(gr->seconds, seconds.data(), sizeof(double) * seconds.size());
  if (gr->timings) std::memcpy(gr->timings, timings.data(), sizeof(double) * NLS_NUM_TIMINGS);
  if (gr->finished_count) *gr->finished_count = finished_count;
  *gr->sigma_index = k_opt;
  *gr->gamma_index = g_opt;
  *gr->best_valid = 1;
  if (a->loo_errors) std::memcpy(a->loo_errors, table.data() + (size_t)k_opt * G, sizeof(double) * G);
  if (a->objective) std::memcpy(a->objective, objective.data() + (size_t)k_opt * G, sizeof(double) * G);
  if (a->gamma_index) *a->gamma_index = g_opt;
  if (a->finished) *a->finished = 1;
  if (a->lam) std::memcpy(a->lam, lw.best_lam.data(), sizeof(double) * D1);
  if (a->timings) std::memcpy(a->timings, timings.data(), sizeof(double) * NLS_NUM_TIMINGS);
  return NLS_OK;
}
