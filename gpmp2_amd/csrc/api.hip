// api.hip -- the C ABI of include/gpmp2mi.h: handle management, host-side constant
// precomputation (GP matrices, whitening weights), marshalling and the optimizer driver loop.
// There is deliberately no CPU compute path in this file: every entry point launches kernels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <fstream>
#include <utility>
#include <vector>

#include "common.h"
#include "launch.h"
#include "plan.h"

namespace g2 {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

// RAII device buffer used by the host-pointer convenience entry points
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    n = count;
    if (count == 0) return GPMP2MI_OK;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) {
      set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
      p = nullptr;
      return (e == hipErrorNoDevice) ? GPMP2MI_ERR_NO_DEVICE : GPMP2MI_ERR_ALLOC;
    }
    return GPMP2MI_OK;
  }
  int upload(const T* h, size_t count) {
    int rc = alloc(count);
    if (rc) return rc;
    if (count) G2_HIP(hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice));
    return GPMP2MI_OK;
  }
  int download(T* h) const {
    if (n && h) G2_HIP(hipMemcpy(h, p, n * sizeof(T), hipMemcpyDeviceToHost));
    return GPMP2MI_OK;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};
static int ensure_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no usable HIP device (this library has no CPU fallback)");
    return GPMP2MI_ERR_NO_DEVICE;
  }
  return GPMP2MI_OK;
}

// 2x2 scalar GP matrices (gpmp2/gp/GPutils.h:25-59 with Qc factored out)
static void mm2(const double A[4], const double B[4], double C[4]) {
  const double c0 = A[0] * B[0] + A[1] * B[2], c1 = A[0] * B[1] + A[1] * B[3];
  const double c2 = A[2] * B[0] + A[3] * B[2], c3 = A[2] * B[1] + A[3] * B[3];
  C[0] = c0; C[1] = c1; C[2] = c2; C[3] = c3;
}
static void gp_winv(double dt, double W[4]) {
  W[0] = 12.0 * std::pow(dt, -3.0);
  W[1] = W[2] = (-6.0) * std::pow(dt, -2.0);
  W[3] = 4.0 * std::pow(dt, -1.0);
}
static GpCoef gp_coef(double dt, double tau) {
  const double A[4] = {1.0 / 3 * std::pow(tau, 3.0), 1.0 / 2 * std::pow(tau, 2.0),
                       1.0 / 2 * std::pow(tau, 2.0), tau};
  const double CtT[4] = {1.0, 0.0, dt - tau, 1.0};  // Phi(dt - tau)^T
  double W[4], T[4], Psi[4], PC[4];
  gp_winv(dt, W);
  mm2(A, CtT, T);
  mm2(T, W, Psi);
  const double Cdt[4] = {1.0, dt, 0.0, 1.0};
  mm2(Psi, Cdt, PC);
  GpCoef c;
  c.l11 = 1.0 - PC[0];
  c.l12 = tau - PC[1];
  c.l21 = 0.0 - PC[2];
  c.l22 = 1.0 - PC[3];
  c.p11 = Psi[0];
  c.p12 = Psi[1];
  c.p21 = Psi[2];
  c.p22 = Psi[3];
  return c;
}

static bool invert_small(int n, const double* A, double* Ainv) {
  std::vector<double> M(A, A + n * n);
  for (int i = 0; i < n * n; i++) Ainv[i] = 0.0;
  for (int i = 0; i < n; i++) Ainv[i * n + i] = 1.0;
  for (int c = 0; c < n; c++) {
    int p = c;
    for (int i = c + 1; i < n; i++)
      if (std::fabs(M[i * n + c]) > std::fabs(M[p * n + c])) p = i;
    if (M[p * n + c] == 0.0) return false;
    if (p != c)
      for (int j = 0; j < n; j++) {
        std::swap(M[p * n + j], M[c * n + j]);
        std::swap(Ainv[p * n + j], Ainv[c * n + j]);
      }
    const double inv = 1.0 / M[c * n + c];
    for (int j = 0; j < n; j++) {
      M[c * n + j] *= inv;
      Ainv[c * n + j] *= inv;
    }
    for (int i = 0; i < n; i++) {
      if (i == c) continue;
      const double f = M[i * n + c];
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) {
        M[i * n + j] -= f * M[c * n + j];
        Ainv[i * n + j] -= f * Ainv[c * n + j];
      }
    }
  }
  return true;
}
}  // namespace g2

using namespace g2;

// ============================================================================================ handles
struct gpmp2mi_robot {
  RobotDev h;
  RobotDev* d = nullptr;
  ~gpmp2mi_robot() {
    if (d) (void)hipFree(d);
  }
};
struct gpmp2mi_sdf {
  SdfDev h;
  double* plain = nullptr;
  double* cells = nullptr;
  ~gpmp2mi_sdf() {   // also runs when a create function fails half-way (unique_ptr)
    if (plain) (void)hipFree(plain);
    if (cells) (void)hipFree(cells);
  }
};

struct KernelTimer {
  // One HIP event per kernel boundary on the launch stream: mark(name) is recorded right before
  // kernel `name`, close() after the last kernel of a pass; a kernel's time is the distance to the
  // next mark.
  bool enabled = false;
  struct Rec {
    const char* name;  // nullptr = closing mark
    hipEvent_t ev;
  };
  std::vector<Rec> recs;
  std::vector<std::string> names;
  std::vector<double> ms;
  std::vector<int> launches;
  std::vector<const char*> cnames;
  std::vector<hipEvent_t> pool;
  size_t pool_used = 0;
  hipEvent_t get() {
    if (pool_used == pool.size()) {
      hipEvent_t e;
      // timing only: no system-scope fence when the event fires (the default flushes caches between the
      // two kernels it separates)
      (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
      pool.push_back(e);
    }
    return pool[pool_used++];
  }
  void begin(const char* name, hipStream_t st) {
    if (!enabled) return;
    Rec r{name, get()};
    (void)hipEventRecord(r.ev, st);
    recs.push_back(r);
  }
  void end(hipStream_t) {}
  void close(hipStream_t st) { begin(nullptr, st); }
  void reset() {
    recs.clear();
    pool_used = 0;
    names.clear();
    ms.clear();
    launches.clear();
  }
  void collect() {
    names.clear();
    ms.clear();
    launches.clear();
    for (size_t i = 0; i + 1 < recs.size(); i++) {
      if (!recs[i].name) continue;
      float t = 0;
      if (hipEventElapsedTime(&t, recs[i].ev, recs[i + 1].ev) != hipSuccess) continue;
      size_t k = 0;
      for (; k < names.size(); k++)
        if (names[k] == recs[i].name) break;
      if (k == names.size()) {
        names.push_back(recs[i].name);
        ms.push_back(0.0);
        launches.push_back(0);
      }
      ms[k] += t;
      launches[k] += 1;
    }
    cnames.clear();
    for (auto& s : names) cnames.push_back(s.c_str());
  }
  ~KernelTimer() {
    for (auto e : pool) (void)hipEventDestroy(e);
  }
};

// a pinned, device-mapped int array from the pool of flags_acquire / flags_release
struct FlagBuf {
  int* host = nullptr;
  int* dev = nullptr;
  int cap = 0;
  int device = -1;   // pooled buffers are reused on the device they were mapped / allocated for only
};

// Host-mapped pass-flag arrays are recycled across plans: pinning and unpinning host memory costs more than the
// whole solve of a small plan (one-shot gpmp2mi_batch_optimize calls create and destroy a plan each time).
static std::mutex g_flag_mu;
static std::vector<FlagBuf> g_flag_pool;
static int flags_acquire(int need, FlagBuf* out) {
  {
    std::lock_guard<std::mutex> lk(g_flag_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (size_t k = 0; k < g_flag_pool.size(); k++)
      if (g_flag_pool[k].cap >= need && g_flag_pool[k].device == cur) {
        *out = g_flag_pool[k];
        g_flag_pool.erase(g_flag_pool.begin() + k);
        return GPMP2MI_OK;
      }
  }
  FlagBuf f;
  (void)hipGetDevice(&f.device);
  f.cap = std::max(need, 1024);
  G2_HIP(hipHostMalloc((void**)&f.host, (size_t)f.cap * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
  G2_HIP(hipHostGetDevicePointer((void**)&f.dev, f.host, 0));
  *out = f;
  return GPMP2MI_OK;
}
static void flags_release(const FlagBuf& f) {
  if (!f.host) return;
  std::lock_guard<std::mutex> lk(g_flag_mu);
  if (g_flag_pool.size() < 16) g_flag_pool.push_back(f);
  else (void)hipHostFree(f.host);
}

// Plan buffers come out of a few zero-filled arena chunks instead of one hipMalloc + hipMemset + hipFree each (a plan
// has about 65 of them: 0.6 ms of a one-shot gpmp2mi_batch_optimize call was allocation and release).
// standard-size chunks are recycled as well (zero-filled again on reuse); larger ones go back to the driver
constexpr size_t ARENA_CHUNK = (size_t)8 << 20;
static std::vector<std::pair<void*, int>> g_chunk_pool;   // (chunk, device); guarded by g_flag_mu
static void* chunk_acquire() {
  int cur = 0;
  (void)hipGetDevice(&cur);
  std::lock_guard<std::mutex> lk(g_flag_mu);
  for (size_t k = 0; k < g_chunk_pool.size(); k++)
    if (g_chunk_pool[k].second == cur) {
      void* q = g_chunk_pool[k].first;
      g_chunk_pool.erase(g_chunk_pool.begin() + k);
      return q;
    }
  return nullptr;
}
static void chunk_release(void* q, int device) {
  {
    std::lock_guard<std::mutex> lk(g_flag_mu);
    if (g_chunk_pool.size() < 8) {
      g_chunk_pool.push_back({q, device});
      return;
    }
  }
  (void)hipFree(q);
}

// live-resource counters for the lifetime tests (gpmp2mi_debug_resource_counts)
static std::atomic<long> g_live_chunks{0}, g_live_flagbufs{0}, g_leaked_plans{0};

struct gpmp2mi_plan {
  const gpmp2mi_robot* robot = nullptr;
  const gpmp2mi_sdf* sdf = nullptr;
  PlanParams hp;
  PlanBuffers pb;
  std::vector<void*> allocs;   // arena chunks (plan_alloc)
  std::vector<size_t> alloc_bytes;
  char* arena_cur = nullptr;   // bump pointer into the newest chunk
  size_t arena_left = 0;
  int alloc_calls = 0;         // plan_alloc calls so far (GPMP2MI_FAIL_ALLOC_AT injects a failure at the k-th)
  int device = -1;             // the device the plan was created on: its chunks / flags go back to that device's pools
  FlagBuf flagbuf;
  int* h_flags = nullptr;    // pinned + device-mapped [n_active_len]: per-pass active count, -1 = not yet known
  KernelTimer timer;
  bool wide_dense = false;   // GPMP2MI_WIDE_DENSE=1: 8..11-dof plans through the dense block solver (A/B, fallback)
  bool generic_gn = false;   // GPMP2MI_GENERIC_GN=1: run GaussNewton through the LM/Dogleg machinery
  int n_active_len = 0;
  std::vector<int> h_xp_n;   // host mirror of the extra-prior counts
  PlanExtras ex;             // extra factors carried as data (host copy of the specs + device workspace)
  bool has_extras = false;
  bool problem_set = false;
  bool optimized = false;
  // Streams that may still carry work of this plan (asynchronous copies / kernels enqueued without a closing
  // synchronisation).  gpmp2mi_plan_destroy waits for exactly these, never for the whole device.
  std::vector<hipStream_t> dirty_streams;
  bool null_stream_dirty = false;
  // A pass that did not finish within GPMP2MI_WAIT_TIMEOUT_MS: the stream may hold a hung kernel of this plan.  The
  // plan refuses further work, and its memory is neither waited for nor recycled (a hung kernel would hang the wait,
  // a late one would write into recycled memory): it is deliberately leaked.
  bool poisoned = false;
  size_t tsz() const { return (size_t)hp.B * (hp.N + 1) * hp.n; }
  void mark_dirty(hipStream_t st) {
    if (!st) { null_stream_dirty = true; return; }
    if (std::find(dirty_streams.begin(), dirty_streams.end(), st) == dirty_streams.end()) dirty_streams.push_back(st);
  }
  void mark_clean(hipStream_t st) {
    if (!st) { null_stream_dirty = false; return; }
    dirty_streams.erase(std::remove(dirty_streams.begin(), dirty_streams.end(), st), dirty_streams.end());
  }
  // wait for whatever this plan still has in flight (a no-op after the usual optimize -> get_result sequence)
  void drain() {
    if (poisoned) return;
    for (hipStream_t st : dirty_streams) (void)hipStreamSynchronize(st);
    dirty_streams.clear();
    if (null_stream_dirty) (void)hipStreamSynchronize(nullptr);
    null_stream_dirty = false;
  }
  // Returns every arena chunk and the flag buffer (also on a create that failed half-way: the unique_ptr in
  // gpmp2mi_plan_create runs this).  The caller has drained the plan's streams.
  ~gpmp2mi_plan() {
    if (poisoned) {
      g_leaked_plans.fetch_add(1);
      return;
    }
    drain();
    for (size_t k = 0; k < allocs.size(); k++) {
      g_live_chunks.fetch_sub(1);
      if (alloc_bytes[k] == ARENA_CHUNK) chunk_release(allocs[k], device);
      else (void)hipFree(allocs[k]);
    }
    if (flagbuf.host) {
      g_live_flagbufs.fetch_sub(1);
      flags_release(flagbuf);
    }
  }
};

static int plan_run(gpmp2mi_plan* p, hipStream_t st, const double* start);

// linearize `traj` into record buffer `bufsel` of every (active) trajectory: the fused obstacle / GP-prior kernel,
// then -- only for plans that carry extra factors -- the workspace / self-collision factor kernels on the support
// states and their accumulation into the unary records
// dst / pass: fused finish (launch_linearize); the extra-factor kernels then run on the NEW states in dst
static int plan_linearize(gpmp2mi_plan* p, const double* traj, int bufsel, const int* active, hipStream_t st,
                          double* dst = nullptr, int pass = 0, bool trial = false) {
  const PlanParams& P = p->hp;
  G2_TRY(launch_linearize(p->robot->h, p->robot->d, p->sdf->h, P, p->pb, traj, bufsel, active, st, dst, pass, trial));
  if (!p->has_extras) return GPMP2MI_OK;
  if (dst) traj = dst;
  const PlanExtras& ex = p->ex;
  const RobotDev& h = p->robot->h;
  const int M = P.B * (P.N + 1), D = P.D, L = h.nr_links, S = h.nr_spheres;
  if (ex.n_ws > 0) {
    G2_TRY(launch_fk(h, p->robot->d, M, traj, ex.poses, ex.Jp, st, 2 * D));
    for (int f = 0; f < ex.n_ws; f++)
      G2_TRY(launch_workspace_prior(ex.ws_mode[f], ex.ws_link[f], L, D, M, ex.des + 16 * f, ex.poses, ex.Jp,
                                    ex.ws_err + (size_t)f * M * 6, ex.ws_H + (size_t)f * M * 6 * D, st));
  }
  if (ex.n_sc > 0) {
    G2_TRY(launch_sphere_centers(h, p->robot->d, M, traj, ex.cen, ex.Jc, st, 2 * D));
    G2_TRY(launch_self_collision(ex.n_sc, S, D, M, ex.sc_data, ex.radius, ex.cen, ex.Jc, ex.sc_err, ex.sc_H, st));
  }
  return launch_extra_accumulate(P, p->pb, ex, L, S, bufsel, active, st);
}

// GPMP2MI_FAIL_ALLOC_AT=k (tests): the k-th plan_alloc call of every plan creation fails as if the device were out
// of memory, so that the half-built plan's release path can be exercised
static int fail_alloc_at() {
  const char* e = getenv("GPMP2MI_FAIL_ALLOC_AT");
  return e ? atoi(e) : 0;
}
template <class T>
static int plan_alloc(gpmp2mi_plan* p, T** ptr, size_t count) {
  constexpr size_t ALIGN = 256, CHUNK = ARENA_CHUNK;
  const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + ALIGN - 1) / ALIGN * ALIGN;
  if (++p->alloc_calls == fail_alloc_at()) {
    set_error("hipMalloc: injected failure (GPMP2MI_FAIL_ALLOC_AT)");
    return GPMP2MI_ERR_ALLOC;
  }
  if (bytes > p->arena_left) {
    const size_t chunk = std::max(bytes, CHUNK);
    void* q = (chunk == CHUNK) ? chunk_acquire() : nullptr;
    if (!q) {
      hipError_t e = hipMalloc(&q, chunk);
      if (e != hipSuccess) {
        set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
        return GPMP2MI_ERR_ALLOC;
      }
    }
    // owned by the plan from here on: a failing memset below is released with everything else by ~gpmp2mi_plan
    p->allocs.push_back(q);
    p->alloc_bytes.push_back(chunk);
    g_live_chunks.fetch_add(1);
    p->arena_cur = (char*)q;
    p->arena_left = chunk;
    p->null_stream_dirty = true;   // the zero fill runs on the null stream; plan_create's closing copy waits for it
    G2_HIP(hipMemsetAsync(q, 0, chunk, nullptr));
  }
  *ptr = (T*)p->arena_cur;
  p->arena_cur += bytes;
  p->arena_left -= bytes;
  return GPMP2MI_OK;
}

// test hook kernel (gpmp2mi_debug_stall_begin): spins on a host-mapped word, bounded by the device's real-time clock
__global__ void k_debug_stall(const int* flag, long long max_ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) break;
    __builtin_amdgcn_s_sleep(64);
  }
}

extern "C" {

const char* gpmp2mi_last_error(void) { return g_last_error.c_str(); }
int gpmp2mi_version(void) { return GPMP2MI_VERSION; }
int gpmp2mi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// -------------------------------------------------------------------------------------------- robot
int gpmp2mi_robot_create(const gpmp2mi_robot_desc* d, gpmp2mi_robot** out) {
  G2_CHECK(d && out, GPMP2MI_ERR_INVALID, "null argument");
  *out = nullptr;
  G2_CHECK(d->kind >= 0 && d->kind <= GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_2ARMS, GPMP2MI_ERR_INVALID, "unknown robot kind");
  G2_CHECK(d->arm_dof >= 0 && d->arm_dof <= MAXJ, GPMP2MI_ERR_UNSUPPORTED, "more than 14 arm joints");
  G2_CHECK(d->nr_spheres >= 0 && d->nr_spheres <= MAXS, GPMP2MI_ERR_UNSUPPORTED, "too many body spheres");
  const bool mobile = d->kind >= GPMP2MI_ROBOT_POSE2_MOBILE_BASE;
  const bool lift = d->kind == GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_ARM || d->kind == GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_2ARMS;
  const bool two = d->kind == GPMP2MI_ROBOT_POSE2_MOBILE_2ARMS || d->kind == GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_2ARMS;
  const int base = mobile ? 3 : 0;
  const int dof = (d->kind == GPMP2MI_ROBOT_POINT) ? 2 : base + (lift ? 1 : 0) + d->arm_dof;
  G2_CHECK(d->dof == dof, GPMP2MI_ERR_INVALID, "dof does not match robot kind / arm_dof");
  G2_CHECK(dof <= MAXD, GPMP2MI_ERR_UNSUPPORTED, "total dof > 18");
  if (d->kind == GPMP2MI_ROBOT_ARM || d->kind >= GPMP2MI_ROBOT_POSE2_MOBILE_ARM)
    G2_CHECK(d->arm_dof > 0 && d->a && d->alpha && d->d, GPMP2MI_ERR_INVALID, "missing DH parameters");
  if (two) G2_CHECK(d->arm2_dof > 0 && d->arm2_dof < d->arm_dof, GPMP2MI_ERR_INVALID, "arm2_dof must split arm_dof into two arms");
  G2_TRY(ensure_device());
  auto r = std::make_unique<gpmp2mi_robot>();
  RobotDev& h = r->h;
  std::memset(&h, 0, sizeof(h));
  h.kind = d->kind;
  h.dof = dof;
  h.arm_dof = d->arm_dof;
  h.arm2_dof = two ? d->arm2_dof : 0;
  h.reverse_linact = lift ? (d->reverse_linact != 0) : 0;
  h.base_dof = base;
  h.nr_links = (d->kind == GPMP2MI_ROBOT_ARM) ? d->arm_dof : mobile ? 1 + (lift ? 1 : 0) + d->arm_dof : 1;
  h.nr_spheres = d->nr_spheres;
  for (int j = 0; j < d->arm_dof; j++) {
    h.a[j] = d->a[j];
    h.d[j] = d->d[j];
    h.ca[j] = std::cos(d->alpha[j]);
    h.sa[j] = std::sin(d->alpha[j]);
    h.bias[j] = d->theta_bias ? d->theta_bias[j] : 0.0;
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) {
      h.base[i * 4 + j] = d->base_pose[i * 4 + j];
      h.base2[i * 4 + j] = (lift || two) ? d->base_pose2[i * 4 + j] : (i == j ? 1.0 : 0.0);
      h.base3[i * 4 + j] = (lift && two) ? d->base_pose3[i * 4 + j] : (i == j ? 1.0 : 0.0);
    }
  // sort spheres by link (stable) so the kinematic chain visits them in order
  std::vector<int> order(d->nr_spheres);
  std::iota(order.begin(), order.end(), 0);
  for (int s = 0; s < d->nr_spheres; s++)
    G2_CHECK(d->sphere_link[s] >= 0 && d->sphere_link[s] < h.nr_links, GPMP2MI_ERR_INVALID,
             "sphere link id out of range");
  std::stable_sort(order.begin(), order.end(),
                   [&](int a, int b) { return d->sphere_link[a] < d->sphere_link[b]; });
  for (int s = 0; s < d->nr_spheres; s++) {
    const int o = order[s];
    h.sph_link[s] = d->sphere_link[o];
    h.sph_orig[s] = o;
    h.sph_r[s] = d->sphere_radius[o];
    for (int i = 0; i < 3; i++) h.sph_c[3 * s + i] = d->sphere_center[3 * o + i];
  }
  int s = 0;
  for (int l = 0; l <= h.nr_links; l++) {
    while (s < d->nr_spheres && h.sph_link[s] < l) s++;
    h.link_first[l] = s;
  }
  h.link_first[h.nr_links] = d->nr_spheres;
  G2_HIP(hipMalloc((void**)&r->d, sizeof(RobotDev)));
  G2_HIP(hipMemcpy(r->d, &h, sizeof(RobotDev), hipMemcpyHostToDevice));
  *out = r.release();
  return GPMP2MI_OK;
}
void gpmp2mi_robot_destroy(gpmp2mi_robot* r) { delete r; }
int gpmp2mi_robot_dof(const gpmp2mi_robot* r) { return r ? r->h.dof : -1; }
int gpmp2mi_robot_nr_links(const gpmp2mi_robot* r) { return r ? r->h.nr_links : -1; }
int gpmp2mi_robot_nr_spheres(const gpmp2mi_robot* r) { return r ? r->h.nr_spheres : -1; }

// -------------------------------------------------------------------------------------------- sdf
// geometry + device storage of a field handle; the caller fills s->plain ([nz][ny][nx]) and packs
static int sdf_alloc(int dim, const double origin[3], double cell, int nx, int ny, int nz,
                     std::unique_ptr<gpmp2mi_sdf>& s) {
  G2_CHECK(dim == 2 || dim == 3, GPMP2MI_ERR_INVALID, "dim must be 2 or 3");
  G2_CHECK(nx > 0 && ny > 0 && nz > 0 && cell > 0, GPMP2MI_ERR_INVALID, "bad field size");
  G2_TRY(ensure_device());
  s = std::make_unique<gpmp2mi_sdf>();
  const size_t n = (size_t)nx * ny * nz;
  SdfDev& h = s->h;
  h.dim = dim;
  h.nx = nx;
  h.ny = ny;
  h.nz = nz;
  h.ox = origin[0];
  h.oy = origin[1];
  h.oz = dim == 3 ? origin[2] : 0.0;
  h.cell = cell;
  h.inv_cell = 1.0 / cell;
  // upper faces exactly as SignedDistanceField.h:105-107: origin + (n - 1.0) * cell_size
  h.hix = h.ox + (nx - 1.0) * cell;
  h.hiy = h.oy + (ny - 1.0) * cell;
  h.hiz = h.oz + (nz - 1.0) * cell;
  G2_HIP(hipMalloc((void**)&s->plain, n * sizeof(double)));
  const int nc = dim == 3 ? 8 : 4;
  G2_HIP(hipMalloc((void**)&s->cells, n * nc * sizeof(double)));
  h.plain = s->plain;
  h.cells = s->cells;
  return GPMP2MI_OK;
}

// caller layout -> [nz][ny][nx]
static const double* to_zyx(const double* vox, int layout, int nx, int ny, int nz, std::vector<double>& tmp) {
  if (layout != GPMP2MI_SDF_LAYOUT_GTSAM) return vox;
  tmp.resize((size_t)nx * ny * nz);
  for (int z = 0; z < nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++) tmp[((size_t)z * ny + y) * nx + x] = vox[((size_t)z * nx + x) * ny + y];
  return tmp.data();
}

int gpmp2mi_sdf_create(int dim, const double origin[3], double cell, int nx, int ny, int nz,
                       const double* vox, int layout, gpmp2mi_sdf** out) {
  G2_CHECK(out && origin && vox, GPMP2MI_ERR_INVALID, "null argument");
  *out = nullptr;
  if (dim == 2) nz = 1;
  G2_CHECK(layout == GPMP2MI_SDF_LAYOUT_ZYX || layout == GPMP2MI_SDF_LAYOUT_GTSAM, GPMP2MI_ERR_INVALID,
           "unknown voxel layout");
  std::unique_ptr<gpmp2mi_sdf> s;
  G2_TRY(sdf_alloc(dim, origin, cell, nx, ny, nz, s));
  std::vector<double> tmp;
  const double* src = to_zyx(vox, layout, nx, ny, nz, tmp);
  G2_HIP(hipMemcpy(s->plain, src, (size_t)nx * ny * nz * sizeof(double), hipMemcpyHostToDevice));
  G2_TRY(launch_sdf_pack(s->h, s->cells, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  *out = s.release();
  return GPMP2MI_OK;
}

int gpmp2mi_sdf_field_from_occupancy(int dim, int nx, int ny, int nz, const double* occ, double cell,
                                     double* field) {
  G2_CHECK(occ && field, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK(dim == 2 || dim == 3, GPMP2MI_ERR_INVALID, "dim must be 2 or 3");
  if (dim == 2) nz = 1;
  G2_CHECK(nx > 0 && ny > 0 && nz > 0 && cell > 0, GPMP2MI_ERR_INVALID, "bad grid size");
  G2_TRY(ensure_device());
  const size_t n = (size_t)nx * ny * nz;
  DevBuf<double> d_occ, d_field;
  DevBuf<int> wa, wb;
  G2_TRY(d_occ.upload(occ, n));
  G2_TRY(d_field.alloc(n));
  G2_TRY(wa.alloc(n));
  G2_TRY(wb.alloc(n));
  G2_TRY(launch_sdf_from_occupancy(nx, ny, nz, d_occ.p, cell, wa.p, wb.p, d_field.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  return d_field.download(field);
}

int gpmp2mi_sdf_create_from_occupancy(int dim, const double origin[3], double cell, int nx, int ny, int nz,
                                      const double* occ, int layout, gpmp2mi_sdf** out) {
  G2_CHECK(out && origin && occ, GPMP2MI_ERR_INVALID, "null argument");
  *out = nullptr;
  if (dim == 2) nz = 1;
  G2_CHECK(layout == GPMP2MI_SDF_LAYOUT_ZYX || layout == GPMP2MI_SDF_LAYOUT_GTSAM, GPMP2MI_ERR_INVALID,
           "unknown voxel layout");
  std::unique_ptr<gpmp2mi_sdf> s;
  G2_TRY(sdf_alloc(dim, origin, cell, nx, ny, nz, s));
  const size_t n = (size_t)nx * ny * nz;
  std::vector<double> tmp;
  DevBuf<double> d_occ;
  DevBuf<int> wa, wb;
  G2_TRY(d_occ.upload(to_zyx(occ, layout, nx, ny, nz, tmp), n));
  G2_TRY(wa.alloc(n));
  G2_TRY(wb.alloc(n));
  G2_TRY(launch_sdf_from_occupancy(nx, ny, nz, d_occ.p, cell, wa.p, wb.p, s->plain, nullptr));
  G2_TRY(launch_sdf_pack(s->h, s->cells, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  *out = s.release();
  return GPMP2MI_OK;
}

int gpmp2mi_sdf_get_field(const gpmp2mi_sdf* s, int* dim, int* nx, int* ny, int* nz, double origin[3],
                          double* cell, double* field) {
  G2_CHECK(s, GPMP2MI_ERR_INVALID, "null argument");
  if (dim) *dim = s->h.dim;
  if (nx) *nx = s->h.nx;
  if (ny) *ny = s->h.ny;
  if (nz) *nz = s->h.nz;
  if (origin) origin[0] = s->h.ox, origin[1] = s->h.oy, origin[2] = s->h.oz;
  if (cell) *cell = s->h.cell;
  if (field)
    G2_HIP(hipMemcpy(field, s->plain, (size_t)s->h.nx * s->h.ny * s->h.nz * sizeof(double), hipMemcpyDeviceToHost));
  return GPMP2MI_OK;
}

int gpmp2mi_sdf_read_vol(const char* filename_pre, gpmp2mi_sdf** out) {
  G2_CHECK(filename_pre && out, GPMP2MI_ERR_INVALID, "null argument");
  *out = nullptr;
  const std::string pre(filename_pre);
  std::ifstream head(pre + ".vol.head");
  G2_CHECK(head.is_open(), GPMP2MI_ERR_INVALID, "cannot open " + pre + ".vol.head");
  long long cols = 0, rows = 0, nz = 0;
  double origin[3] = {0, 0, 0}, res = 0;
  head >> cols >> rows >> nz >> origin[0] >> origin[1] >> origin[2] >> res;
  G2_CHECK(!head.fail() && cols > 0 && rows > 0 && nz > 0 && res > 0, GPMP2MI_ERR_INVALID, "malformed " + pre + ".vol.head");
  std::ifstream data(pre + ".vol.data");
  G2_CHECK(data.is_open(), GPMP2MI_ERR_INVALID, "cannot open " + pre + ".vol.data");
  // x outermost, then y, then z (fileUtils.cpp:48-55)
  std::vector<double> zyx((size_t)cols * rows * nz);
  for (long long x = 0; x < cols; x++)
    for (long long y = 0; y < rows; y++)
      for (long long z = 0; z < nz; z++) {
        double v;
        data >> v;
        G2_CHECK(!data.fail(), GPMP2MI_ERR_INVALID, "short or malformed " + pre + ".vol.data");
        zyx[((size_t)z * rows + y) * cols + x] = v;
      }
  return gpmp2mi_sdf_create(3, origin, res, (int)cols, (int)rows, (int)nz, zyx.data(), GPMP2MI_SDF_LAYOUT_ZYX, out);
}
void gpmp2mi_sdf_destroy(gpmp2mi_sdf* s) { delete s; }

int gpmp2mi_sdf_query(const gpmp2mi_sdf* s, int M, const double* pts, double* dist, double* grad, int* inr) {
  G2_CHECK(s && pts && dist && M >= 0, GPMP2MI_ERR_INVALID, "null argument");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  DevBuf<double> dp, dd, dg;
  DevBuf<int> di;
  G2_TRY(dp.upload(pts, (size_t)M * s->h.dim));
  G2_TRY(dd.alloc(M));
  if (grad) G2_TRY(dg.alloc((size_t)M * s->h.dim));
  if (inr) G2_TRY(di.alloc(M));
  G2_TRY(launch_sdf_query(s->h, M, dp.p, dd.p, dg.p, di.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(dd.download(dist));
  G2_TRY(dg.download(grad));
  G2_TRY(di.download(inr));
  return GPMP2MI_OK;
}

// -------------------------------------------------------------------------------------------- settings
void gpmp2mi_settings_default(gpmp2mi_settings* s, int dof) {
  std::memset(s, 0, sizeof(*s));
  s->dof = dof;
  s->total_step = 10;
  s->total_time = 1.0;
  s->conf_prior_sigma = 0.0001;
  s->vel_prior_sigma = 0.0001;
  s->epsilon = 0.2;
  s->cost_sigma = 0.1;
  s->obs_check_inter = 5;
  s->opt_type = GPMP2MI_OPT_DOGLEG;
  s->final_iter_no_increase = 1;
  s->rel_thresh = 1e-2;
  s->max_iter = 50;
}
void gpmp2mi_graph_opts_default(gpmp2mi_graph_opts* o) {
  std::memset(o, 0, sizeof(*o));
  o->lm_lambda_initial = 100.0;
  o->lm_lambda_factor = 10.0;
  o->lm_lambda_upper = 1e5;
  o->lm_lambda_lower = 0.0;
  o->lm_min_model_fidelity = 1e-3;
  o->dogleg_delta_initial = 0.2;
  o->abs_error_tol = 1e-5;
  o->error_tol = 0.0;
}

// -------------------------------------------------------------------------------------------- factor level
int gpmp2mi_forward_kinematics(const gpmp2mi_robot* r, int M, const double* conf, double* poses, double* J) {
  G2_CHECK(r && conf && poses && M >= 0, GPMP2MI_ERR_INVALID, "null argument");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  const int D = r->h.dof, L = r->h.nr_links;
  DevBuf<double> dq, dp, dj;
  G2_TRY(dq.upload(conf, (size_t)M * D));
  G2_TRY(dp.alloc((size_t)M * L * 16));
  if (J) G2_TRY(dj.alloc((size_t)M * L * 6 * D));
  G2_TRY(launch_fk(r->h, r->d, M, dq.p, dp.p, dj.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(dp.download(poses));
  G2_TRY(dj.download(J));
  return GPMP2MI_OK;
}

int gpmp2mi_sphere_centers(const gpmp2mi_robot* r, int M, const double* conf, double* centers, double* J) {
  G2_CHECK(r && conf && centers && M >= 0, GPMP2MI_ERR_INVALID, "null argument");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  const int D = r->h.dof, S = r->h.nr_spheres;
  DevBuf<double> dq, dc, dj;
  G2_TRY(dq.upload(conf, (size_t)M * D));
  G2_TRY(dc.alloc((size_t)M * S * 3));
  if (J) G2_TRY(dj.alloc((size_t)M * S * 3 * D));
  G2_TRY(launch_sphere_centers(r->h, r->d, M, dq.p, dc.p, dj.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(dc.download(centers));
  G2_TRY(dj.download(J));
  return GPMP2MI_OK;
}

int gpmp2mi_workspace_prior_factor(const gpmp2mi_robot* r, int mode, int joint, const double des_pose[16], int M,
                                   const double* conf, double* err, double* H) {
  G2_CHECK(r && des_pose && conf && err && M >= 0, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK(mode >= GPMP2MI_WORKSPACE_POSITION && mode <= GPMP2MI_WORKSPACE_POSE, GPMP2MI_ERR_INVALID, "unknown mode");
  G2_CHECK(joint >= 0 && joint < r->h.nr_links, GPMP2MI_ERR_INVALID, "joint out of range");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  const int D = r->h.dof, L = r->h.nr_links, rows = mode == GPMP2MI_WORKSPACE_POSE ? 6 : 3;
  DevBuf<double> dq, dp, dj, dd, de, dh;
  G2_TRY(dq.upload(conf, (size_t)M * D));
  G2_TRY(dd.upload(des_pose, 16));
  G2_TRY(dp.alloc((size_t)M * L * 16));
  if (H) G2_TRY(dj.alloc((size_t)M * L * 6 * D));
  G2_TRY(de.alloc((size_t)M * rows));
  if (H) G2_TRY(dh.alloc((size_t)M * rows * D));
  G2_TRY(launch_fk(r->h, r->d, M, dq.p, dp.p, dj.p, nullptr));
  G2_TRY(launch_workspace_prior(mode, joint, L, D, M, dd.p, dp.p, dj.p, de.p, dh.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(de.download(err));
  G2_TRY(dh.download(H));
  return GPMP2MI_OK;
}

int gpmp2mi_goal_factor_arm(const gpmp2mi_robot* r, const double dest_point[3], int M, const double* conf, double* err,
                            double* H) {
  G2_CHECK(r && dest_point, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK(r->h.kind == GPMP2MI_ROBOT_ARM, GPMP2MI_ERR_INVALID, "GoalFactorArm needs an Arm");
  const double des[16] = {1, 0, 0, dest_point[0], 0, 1, 0, dest_point[1], 0, 0, 1, dest_point[2], 0, 0, 0, 1};
  return gpmp2mi_workspace_prior_factor(r, GPMP2MI_WORKSPACE_POSITION, r->h.arm_dof - 1, des, M, conf, err, H);
}

int gpmp2mi_self_collision_factor(const gpmp2mi_robot* r, int n_pairs, const double* data, int M, const double* conf,
                                  double* err, double* H) {
  G2_CHECK(r && data && conf && err && M >= 0 && n_pairs >= 0, GPMP2MI_ERR_INVALID, "null argument");
  const int D = r->h.dof, S = r->h.nr_spheres;
  for (int i = 0; i < n_pairs; i++) {
    const double a = data[i * 4], b = data[i * 4 + 1];
    G2_CHECK(a >= 0 && a < S && b >= 0 && b < S, GPMP2MI_ERR_INVALID, "sphere id out of range");
  }
  if (M == 0 || n_pairs == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  std::vector<double> radius(S);
  for (int s = 0; s < S; s++) radius[r->h.sph_orig[s]] = r->h.sph_r[s];
  DevBuf<double> dq, dc, dj, dd, dr, de, dh;
  G2_TRY(dq.upload(conf, (size_t)M * D));
  G2_TRY(dd.upload(data, (size_t)n_pairs * 4));
  G2_TRY(dr.upload(radius.data(), S));
  G2_TRY(dc.alloc((size_t)M * S * 3));
  if (H) G2_TRY(dj.alloc((size_t)M * S * 3 * D));
  G2_TRY(de.alloc((size_t)M * n_pairs));
  if (H) G2_TRY(dh.alloc((size_t)M * n_pairs * D));
  G2_TRY(launch_sphere_centers(r->h, r->d, M, dq.p, dc.p, dj.p, nullptr));
  G2_TRY(launch_self_collision(n_pairs, S, D, M, dd.p, dr.p, dc.p, dj.p, de.p, dh.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(de.download(err));
  G2_TRY(dh.download(H));
  return GPMP2MI_OK;
}

int gpmp2mi_obstacle_factor(const gpmp2mi_robot* r, const gpmp2mi_sdf* s, double eps, int M,
                            const double* conf, double* err, double* H1) {
  G2_CHECK(r && s && conf && err && M >= 0, GPMP2MI_ERR_INVALID, "null argument");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  const int D = r->h.dof, S = r->h.nr_spheres;
  DevBuf<double> dq, de, dh;
  G2_TRY(dq.upload(conf, (size_t)M * D));
  G2_TRY(de.alloc((size_t)M * S));
  if (H1) G2_TRY(dh.alloc((size_t)M * S * D));
  G2_TRY(launch_obstacle(r->h, r->d, s->h, eps, M, dq.p, de.p, dh.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(de.download(err));
  G2_TRY(dh.download(H1));
  return GPMP2MI_OK;
}

int gpmp2mi_obstacle_gp_factor(const gpmp2mi_robot* r, const gpmp2mi_sdf* s, double eps, const double* Qc,
                               double delta_t, double tau, int M, const double* c1, const double* v1,
                               const double* c2, const double* v2, double* err, double* H1, double* H2,
                               double* H3, double* H4) {
  (void)Qc;  // Lambda / Psi do not depend on Qc (pinned by tests/test_oracle_known_answers.py)
  G2_CHECK(r && s && c1 && v1 && c2 && v2 && err && M >= 0, GPMP2MI_ERR_INVALID, "null argument");
  const bool jac = H1 || H2 || H3 || H4;
  G2_CHECK(!jac || (H1 && H2 && H3 && H4), GPMP2MI_ERR_INVALID, "pass all four Jacobians or none");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  const int D = r->h.dof, S = r->h.nr_spheres;
  DevBuf<double> a, b, c, d, de, h1, h2, h3, h4;
  G2_TRY(a.upload(c1, (size_t)M * D));
  G2_TRY(b.upload(v1, (size_t)M * D));
  G2_TRY(c.upload(c2, (size_t)M * D));
  G2_TRY(d.upload(v2, (size_t)M * D));
  G2_TRY(de.alloc((size_t)M * S));
  if (jac) {
    G2_TRY(h1.alloc((size_t)M * S * D));
    G2_TRY(h2.alloc((size_t)M * S * D));
    G2_TRY(h3.alloc((size_t)M * S * D));
    G2_TRY(h4.alloc((size_t)M * S * D));
  }
  const GpCoef gc = gp_coef(delta_t, tau);
  G2_TRY(launch_obstacle_gp(r->h, r->d, s->h, eps, gc, M, a.p, b.p, c.p, d.p, de.p, h1.p, h2.p, h3.p, h4.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(de.download(err));
  G2_TRY(h1.download(H1));
  G2_TRY(h2.download(H2));
  G2_TRY(h3.download(H3));
  G2_TRY(h4.download(H4));
  return GPMP2MI_OK;
}

int gpmp2mi_gp_prior_factor(int D, int lie, double dt, int M, const double* c1, const double* v1,
                            const double* c2, const double* v2, double* err, double* H1, double* H2,
                            double* H3, double* H4) {
  G2_CHECK(c1 && v1 && c2 && v2 && err && M >= 0 && D > 0, GPMP2MI_ERR_INVALID, "null argument");
  const bool jac = H1 || H2 || H3 || H4;
  G2_CHECK(!jac || (H1 && H2 && H3 && H4), GPMP2MI_ERR_INVALID, "pass all four Jacobians or none");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  DevBuf<double> a, b, c, d, de, h1, h2, h3, h4;
  G2_TRY(a.upload(c1, (size_t)M * D));
  G2_TRY(b.upload(v1, (size_t)M * D));
  G2_TRY(c.upload(c2, (size_t)M * D));
  G2_TRY(d.upload(v2, (size_t)M * D));
  G2_TRY(de.alloc((size_t)M * 2 * D));
  if (jac) {
    G2_TRY(h1.alloc((size_t)M * 2 * D * D));
    G2_TRY(h2.alloc((size_t)M * 2 * D * D));
    G2_TRY(h3.alloc((size_t)M * 2 * D * D));
    G2_TRY(h4.alloc((size_t)M * 2 * D * D));
  }
  if (lie) G2_TRY(launch_gp_prior_lie(D, dt, M, a.p, b.p, c.p, d.p, de.p, h1.p, h2.p, h3.p, h4.p, nullptr));
  else G2_TRY(launch_gp_prior_linear(D, dt, M, a.p, b.p, c.p, d.p, de.p, h1.p, h2.p, h3.p, h4.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(de.download(err));
  G2_TRY(h1.download(H1));
  G2_TRY(h2.download(H2));
  G2_TRY(h3.download(H3));
  G2_TRY(h4.download(H4));
  return GPMP2MI_OK;
}

int gpmp2mi_gp_interpolate(int D, int lie, const double* Qc, double dt, double tau, int M, const double* c1,
                           const double* v1, const double* c2, const double* v2, double* conf, double* vel) {
  (void)Qc;
  G2_CHECK(c1 && v1 && c2 && v2 && M >= 0 && D > 0, GPMP2MI_ERR_INVALID, "null argument");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  DevBuf<double> a, b, c, d, oc, ov;
  G2_TRY(a.upload(c1, (size_t)M * D));
  G2_TRY(b.upload(v1, (size_t)M * D));
  G2_TRY(c.upload(c2, (size_t)M * D));
  G2_TRY(d.upload(v2, (size_t)M * D));
  if (conf) G2_TRY(oc.alloc((size_t)M * D));
  if (vel) G2_TRY(ov.alloc((size_t)M * D));
  if (lie) G2_TRY(launch_gp_interp_lie(D, gp_coef(dt, tau), M, a.p, b.p, c.p, d.p, oc.p, ov.p, nullptr));
  else G2_TRY(launch_gp_interp_linear(D, gp_coef(dt, tau), M, a.p, b.p, c.p, d.p, oc.p, ov.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(oc.download(conf));
  G2_TRY(ov.download(vel));
  return GPMP2MI_OK;
}

int gpmp2mi_interpolate_traj_dev(int D, int lie, double dt, int inter, int B, int N, int start, int end,
                                 const double* traj, double* out, void* stream) {
  G2_CHECK(traj && out && B >= 0 && D > 0 && inter >= 0 && dt > 0, GPMP2MI_ERR_INVALID, "bad argument");
  G2_CHECK(start >= 0 && start < end && end <= N, GPMP2MI_ERR_INVALID, "need 0 <= start_index < end_index <= total_step");
  if (B == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  const long long Mo = (long long)(end - start) * (inter + 1) + 1;
  G2_CHECK(Mo * B < (1ll << 31), GPMP2MI_ERR_INVALID, "too many output states for one launch");
  return launch_interpolate_traj(D, lie != 0, dt, inter, B, N, start, (int)Mo, traj, out, (hipStream_t)stream);
}

int gpmp2mi_interpolate_traj(int D, int lie, const double* Qc, double dt, int inter, int B, int N, int start,
                             int end, const double* traj, double* out) {
  (void)Qc;
  G2_CHECK(traj && out && B >= 0 && D > 0 && inter >= 0, GPMP2MI_ERR_INVALID, "bad argument");
  G2_CHECK(start >= 0 && start < end && end <= N, GPMP2MI_ERR_INVALID, "need 0 <= start_index < end_index <= total_step");
  if (B == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  const size_t Mo = (size_t)(end - start) * (inter + 1) + 1;
  DevBuf<double> a, o;
  G2_TRY(a.upload(traj, (size_t)B * (N + 1) * 2 * D));
  G2_TRY(o.alloc((size_t)B * Mo * 2 * D));
  G2_TRY(gpmp2mi_interpolate_traj_dev(D, lie, dt, inter, B, N, start, end, a.p, o.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(o.download(out));
  return GPMP2MI_OK;
}

int gpmp2mi_vehicle_dynamics_factor(int D, int lie, int M, const double* conf, const double* vel, double* err, double* Hp,
                                    double* Hv) {
  G2_CHECK(conf && vel && err && M >= 0 && D >= 3, GPMP2MI_ERR_INVALID, "null argument or dof < 3");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  DevBuf<double> dc, dv, de, dp, dh;
  G2_TRY(dc.upload(conf, (size_t)M * D));
  G2_TRY(dv.upload(vel, (size_t)M * D));
  G2_TRY(de.alloc(M));
  if (Hp) G2_TRY(dp.alloc((size_t)M * D));
  if (Hv) G2_TRY(dh.alloc((size_t)M * D));
  G2_TRY(launch_vehicle_dynamics(D, lie, M, dc.p, dv.p, de.p, dp.p, dh.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(de.download(err));
  G2_TRY(dp.download(Hp));
  G2_TRY(dh.download(Hv));
  return GPMP2MI_OK;
}

int gpmp2mi_joint_limit_factor(int D, const double* down, const double* up, const double* th, int M,
                               const double* x, double* err, double* Hd) {
  G2_CHECK(down && up && th && x && err && M >= 0 && D > 0, GPMP2MI_ERR_INVALID, "null argument");
  if (M == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  DevBuf<double> a, b, c, dx, de, dh;
  G2_TRY(a.upload(down, D));
  G2_TRY(b.upload(up, D));
  G2_TRY(c.upload(th, D));
  G2_TRY(dx.upload(x, (size_t)M * D));
  G2_TRY(de.alloc((size_t)M * D));
  if (Hd) G2_TRY(dh.alloc((size_t)M * D));
  G2_TRY(launch_joint_limit(D, a.p, b.p, c.p, M, dx.p, de.p, dh.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(de.download(err));
  G2_TRY(dh.download(Hd));
  return GPMP2MI_OK;
}

int gpmp2mi_block_tridiag_solve(int B, int nblk, int n, const double* Hd, const double* Ho, const double* b,
                                double* x, int* ok) {
  G2_CHECK(Hd && b && x && B >= 0 && nblk > 0 && n > 0, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK(nblk == 1 || Ho, GPMP2MI_ERR_INVALID, "null argument");
  if (B == 0) return GPMP2MI_OK;
  G2_TRY(ensure_device());
  DevBuf<double> dd, dob, db, dx, ds;
  DevBuf<int> dk;
  G2_TRY(dd.upload(Hd, (size_t)B * nblk * n * n));
  G2_TRY(dob.upload(Ho, (size_t)B * (nblk - 1) * n * n));
  G2_TRY(db.upload(b, (size_t)B * nblk * n));
  G2_TRY(dx.alloc((size_t)B * nblk * n));
  G2_TRY(ds.alloc((size_t)B * nblk * 512));
  G2_TRY(dk.alloc(B));
  G2_TRY(launch_block_tridiag_solve(B, nblk, n, dd.p, dob.p, db.p, dx.p, dk.p, ds.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(dx.download(x));
  G2_TRY(dk.download(ok));
  return GPMP2MI_OK;
}

// -------------------------------------------------------------------------------------------- plan
int gpmp2mi_plan_create(const gpmp2mi_robot* robot, const gpmp2mi_sdf* sdf, const gpmp2mi_settings* s,
                        const gpmp2mi_graph_opts* o_in, int B, gpmp2mi_plan** out) {
  G2_CHECK(robot && sdf && s && out, GPMP2MI_ERR_INVALID, "null argument");
  *out = nullptr;
  gpmp2mi_graph_opts o;
  if (o_in) o = *o_in;
  else gpmp2mi_graph_opts_default(&o);
  const int D = robot->h.dof;
  G2_CHECK(B > 0, GPMP2MI_ERR_INVALID, "batch size must be positive");
  G2_CHECK(s->dof == D, GPMP2MI_ERR_INVALID, "[TrajOptimizerSetting] dof does not match the robot");
  G2_CHECK(s->total_step >= 1 && s->total_time > 0, GPMP2MI_ERR_INVALID, "bad total_step / total_time");
  G2_CHECK(s->obs_check_inter >= 0 && s->obs_check_inter <= MAXI, GPMP2MI_ERR_UNSUPPORTED, "obs_check_inter > 16");
  G2_CHECK(D <= MAXD, GPMP2MI_ERR_UNSUPPORTED, "plans are instantiated for dof <= 18");
  // the dense path (dof > 11) exists for the reference's PR2-class models only: normal-equation export and the dense
  // block solve are instantiated for dof 17 and 18 (plan_kernels.hip G2_EXP_CASE), so 12..16 would be created and then
  // fail inside optimize
  G2_CHECK(D <= 11 || D == 17 || D == 18, GPMP2MI_ERR_UNSUPPORTED,
           "plans are instantiated for dof <= 11 and for dof 17 / 18 (SE(2) base [+ lift] + two 7-joint arms)");
  const bool wide = 2 * D > 15;  // blocks wider than one 16x16 tile: 2x2-tile cyclic reduction (dof <= 11)
  const bool dense_only = D > 11; // 12 <= dof <= 18 (PR2): dense normal equations + cyclic reduction over dense blocks
  {
    // the assembler stages an interval with at most NLD2 16-B loads per lane (assembler.h: 6, 9 on the wide path)
    const int nd = D * (D + 1) / 2 + D + 1 + ((robot->h.base_dof == 3 && s->obs_check_inter > 0) ? 36 : 0);
    const int gpr = 2 * D + 1 + (robot->h.base_dof == 3 ? 18 : 0);
    const int nds = (nd + 1) & ~1, gps = (gpr + 1) & ~1;
    G2_CHECK((s->obs_check_inter + 1) * nds + gps + 24 * s->obs_check_inter <= 2 * 64 * (dense_only ? 14 : wide ? 9 : 6), GPMP2MI_ERR_UNSUPPORTED,
             "obs_check_inter too large for the staged assembly");
  }
  G2_CHECK(s->opt_type >= GPMP2MI_OPT_GAUSS_NEWTON && s->opt_type <= GPMP2MI_OPT_DOGLEG, GPMP2MI_ERR_INVALID,
           "unknown opt_type");
  G2_CHECK(s->cost_sigma > 0 && s->conf_prior_sigma > 0 && s->vel_prior_sigma > 0, GPMP2MI_ERR_INVALID,
           "sigmas must be positive");
  if (s->flag_vel_limit && s->vel_limits)
    for (int k = 0; k < D; k++)
      G2_CHECK(s->vel_limits[k] > 0, GPMP2MI_ERR_INVALID, "[VelocityLimitFactorVector] velocity limit <= 0");
  G2_TRY(ensure_device());

  auto p = std::make_unique<gpmp2mi_plan>();   // ~gpmp2mi_plan returns whatever has been allocated if anything below fails
  p->robot = robot;
  p->sdf = sdf;
  G2_HIP(hipGetDevice(&p->device));
  PlanParams& P = p->hp;
  std::memset(&P, 0, sizeof(P));
  P.B = B;
  P.N = s->total_step;
  P.I = s->obs_check_inter;
  P.P = 1 + P.N * (P.I + 1);
  P.Ppad = (P.P + 63) / 64 * 64;
  P.D = D;
  P.n = 2 * D;
  P.NG = D * (D + 1) / 2;
  P.REC = P.NG + D + 1 + ((robot->h.base_dof == 3 && P.I > 0) ? 36 : 0);
  P.Npad = (P.N + 1 + 63) / 64 * 64;
  P.lie = robot->h.base_dof == 3 ? 1 : 0;
  P.wide = wide ? 1 : 0;
  const char* wd_env = getenv("GPMP2MI_WIDE_DENSE");
  const bool dense_path = dense_only || (wide && wd_env && wd_env[0] == '1');   // dense block solver: no split tail
  P.split_back = (!dense_path && P.N >= 16) ? 1 : 0;   // the finish kernels take groups of 8 blocks (levels 4, 2, 1)
  if (const char* e = getenv("GPMP2MI_SPLIT_BACK")) if (e[0] == '0') P.split_back = 0;   // A/B: whole back-substitution in the step kernel
  P.spart_groups = (P.N + 8) / 8;
  // wide blocks: the first forward levels (2, 4) run chip-wide when they are not among the last two of the tree
  P.wide_h0 = 2;
  if (wide) {
    const char* e = getenv("GPMP2MI_WIDE_H0");
    const int want = e ? atoi(e) : 8;   // (16 measured: see DESIGN)
    while (P.wide_h0 < want && 4 * P.wide_h0 <= P.N) P.wide_h0 *= 2;
  }
  P.GPREC = P.n + 1 + (P.lie ? 18 : 0);
  P.RECS = (P.REC + 1) & ~1;
  P.GPS = (P.GPREC + 1) & ~1;
  P.obs_skip_first = o.obs_skip_first_state;
  P.flag_pos_limit = s->flag_pos_limit;
  P.flag_vel_limit = s->flag_vel_limit;
  P.opt_type = s->opt_type;
  P.max_iter = s->max_iter;
  P.no_increase = s->final_iter_no_increase;
  P.fixed_iters = o.fixed_iterations;
  P.end_conf_prior_off = o.end_conf_prior_off ? 1 : 0;
  {
    // sphere-split linearization for fixed-base arms: four wavefronts per 64 points sharing one walk of the chain
    // (k_linearize_arm), or two wavefronts that each walk it (k_linearize NSPLIT = 2).  Alone, the four-wavefront form wins
    // up to 256 trajectories (scripts/probes/split_sweep.sh, round 3: 14.4 / 16.8 / 22.5 / 35.5 / 59.3 us against 17.6 / 18.7 /
    // 24.2 / 35.6 / 58.1 us at 32 / 64 / 128 / 256 / 512); with the fused finish, which only it has, it wins the Gauss-Newton
    // pass at every size (195.3 / 226.9 / 224.8 k against 193.4 / 225.1 / 218.6 k traj/s at 256 / 512 / 1 024), and the LM pass
    // too, if barely (97.7 / 102.0 k against 96.6 / 101.3 k at 512 / 1 024).  So: Gauss-Newton and LM plans always, Dogleg plans
    // (no fusion there) up to 256 trajectories.  GPMP2MI_LIN_SPLIT=1 / 2 / 4 forces a form.
    const char* e = getenv("GPMP2MI_LIN_SPLIT");
    const bool four = B <= 256 || s->opt_type != GPMP2MI_OPT_DOGLEG;
    P.lin_split = (robot->h.kind == GPMP2MI_ROBOT_ARM && robot->h.nr_spheres >= 2) ? (four ? 4 : 2) : 1;
    if (e && (e[0] == '1' || e[0] == '2' || e[0] == '4')) P.lin_split = e[0] - '0';
    // (the four-wavefront form keeps the <= 24 states of a chunk in LDS: two or more sub-steps per interval)
    if (P.lin_split == 4 && (robot->h.kind != GPMP2MI_ROBOT_ARM || s->obs_check_inter < 2)) P.lin_split = 2;
    // fused finish of the Gauss-Newton fast path (k_linearize_arm); GPMP2MI_FUSED_FINISH=0: k_finish_step as before
    const char* ff = getenv("GPMP2MI_FUSED_FINISH");
    P.fuse_finish = (P.lin_split == 4 && P.split_back && !wide && !(ff && ff[0] == '0')) ? 1 : 0;
    if (P.fuse_finish) P.spart_groups = P.Ppad / 64;   // the trial-step shares then come per chunk of k_linearize_arm
  }
  P.eps = s->epsilon;
  P.obs_w = 1.0 / (s->cost_sigma * s->cost_sigma);
  // planner/BatchTrajOptimizer-inl.h:30-31
  P.delta_t = s->total_time / static_cast<double>(s->total_step);
  const double inter_dt = P.delta_t / static_cast<double>(s->obs_check_inter + 1);
  P.conf_prior_w = 1.0 / (s->conf_prior_sigma * s->conf_prior_sigma);
  P.vel_prior_w = 1.0 / (s->vel_prior_sigma * s->vel_prior_sigma);
  P.vdyn_w = o.vehicle_dynamics_sigma > 0 ? 1.0 / (o.vehicle_dynamics_sigma * o.vehicle_dynamics_sigma) : 0.0;
  P.rel_thresh = s->rel_thresh;
  P.abs_tol = o.abs_error_tol;
  P.err_tol = o.error_tol;
  P.lm_lambda0 = o.lm_lambda_initial;
  P.lm_factor = o.lm_lambda_factor;
  P.lm_upper = o.lm_lambda_upper;
  P.lm_lower = o.lm_lambda_lower;
  P.lm_min_fidelity = o.lm_min_model_fidelity;
  P.dl_delta0 = o.dogleg_delta_initial;
  for (int j = 0; j < P.I; j++) {
    P.coef[j] = gp_coef(P.delta_t, inter_dt * static_cast<double>(j + 1));
    const double lam[2] = {P.coef[j].l11, P.coef[j].l12}, psi[2] = {P.coef[j].p11, P.coef[j].p12};
    double* q = P.coefq[j];
    for (int ar = 0; ar < 2; ar++)
      for (int ac = 0; ac < 2; ac++) {
        q[0 + ar * 2 + ac] = psi[ar] * psi[ac];
        q[4 + ar * 2 + ac] = lam[ar] * lam[ac];
        q[8 + ar * 2 + ac] = lam[ar] * psi[ac];
        q[12 + ar * 2 + ac] = psi[ar] * lam[ac];
      }
    q[16] = lam[0]; q[17] = lam[1]; q[18] = psi[0]; q[19] = psi[1];
  }
  gp_winv(P.delta_t, P.Winv);
  std::vector<double> Qc(D * D, 0.0), Qi(D * D, 0.0);
  for (int i = 0; i < D; i++) Qc[i * D + i] = 1.0;
  if (s->Qc) std::copy(s->Qc, s->Qc + D * D, Qc.begin());
  G2_CHECK(invert_small(D, Qc.data(), Qi.data()), GPMP2MI_ERR_INVALID, "Qc is singular");
  std::copy(Qi.begin(), Qi.end(), P.Qc_inv);
  for (int k = 0; k < D; k++) {
    P.pos_lo[k] = s->joint_pos_limits_down ? s->joint_pos_limits_down[k] : -1e6;
    P.pos_hi[k] = s->joint_pos_limits_up ? s->joint_pos_limits_up[k] : 1e6;
    P.pos_th[k] = s->pos_limit_thresh ? s->pos_limit_thresh[k] : 1e-3;
    const double ps = s->pos_limit_sigmas ? s->pos_limit_sigmas[k] : 1e-3;
    P.pos_w[k] = 1.0 / (ps * ps);
    P.vel_lim[k] = s->vel_limits ? s->vel_limits[k] : 1e6;
    P.vel_th[k] = s->vel_limit_thresh ? s->vel_limit_thresh[k] : 1e-3;
    const double vs = s->vel_limit_sigmas ? s->vel_limit_sigmas[k] : 1e-3;
    P.vel_w[k] = 1.0 / (vs * vs);
  }
  // GP prior Hessian blocks: W = B(dt) (x) Qc^-1, Phi = [[I, dt I],[0, I]]
  {
    const int n = P.n;
    const double dt = P.delta_t;
    auto W = [&](int r, int c) { return P.Winv[(r / D) * 2 + (c / D)] * Qi[(r % D) * D + (c % D)]; };
    // (Phi^T W)[r][c] = W[r][c] for x rows; for v rows: dt * W[x row][c] + W[v row][c]
    auto PtW = [&](int r, int c) { return r < D ? W(r, c) : dt * W(r - D, c) + W(r, c); };
    for (int r = 0; r < n; r++)
      for (int c = 0; c < n; c++) {
        P.KB[r * n + c] = W(r, c);
        P.KO[r * n + c] = -PtW(r, c);
        // (Phi^T W Phi)[r][c] = PtW[r][c] for x cols; v cols: dt * PtW[r][x col] + PtW[r][c]
        P.KA[r * n + c] = c < D ? PtW(r, c) : dt * PtW(r, c - D) + PtW(r, c);
      }
  }

  // ---- extra factors as data: checked and copied to host vectors BEFORE the first allocation (an invalid
  // description must not cost a round trip through the allocator)
  PlanExtras& ex = p->ex;
  std::vector<double> des, scd, radius;
  {
    std::memset(&ex, 0, sizeof(ex));
    G2_CHECK(o.n_workspace >= 0 && o.n_workspace <= GPMP2MI_MAX_WORKSPACE_FACTORS, GPMP2MI_ERR_INVALID, "too many workspace factors");
    G2_CHECK(o.n_self_collision >= 0 && o.n_self_collision <= GPMP2MI_MAX_SELF_COLLISION_PAIRS, GPMP2MI_ERR_INVALID,
             "too many self-collision pairs");
    const RobotDev& h = robot->h;
    const size_t M = (size_t)B * (P.N + 1);
    ex.n_ws = o.n_workspace;
    ex.n_sc = o.n_self_collision;
    des.assign(16 * std::max(ex.n_ws, 1), 0.0);
    scd.assign(4 * std::max(ex.n_sc, 1), 0.0);
    radius.assign(std::max(h.nr_spheres, 1), 0.0);
    for (int f = 0; f < ex.n_ws; f++) {
      const gpmp2mi_workspace_factor& w = o.workspace[f];
      G2_CHECK(w.mode >= GPMP2MI_WORKSPACE_POSITION && w.mode <= GPMP2MI_WORKSPACE_POSE, GPMP2MI_ERR_INVALID, "unknown workspace factor mode");
      G2_CHECK(w.link >= 0 && w.link < h.nr_links, GPMP2MI_ERR_INVALID, "workspace factor: link out of range");
      G2_CHECK(w.sigma > 0, GPMP2MI_ERR_INVALID, "workspace factor: sigma must be positive");
      G2_CHECK(w.first_state >= 0 && w.first_state <= w.last_state && w.last_state <= P.N, GPMP2MI_ERR_INVALID,
               "workspace factor: bad state range");
      ex.ws_mode[f] = w.mode;
      ex.ws_link[f] = w.link;
      ex.ws_first[f] = w.first_state;
      ex.ws_last[f] = w.last_state;
      ex.ws_w[f] = 1.0 / (w.sigma * w.sigma);
      std::copy(w.des_pose, w.des_pose + 16, des.begin() + 16 * f);
    }
    if (ex.n_sc > 0) {
      G2_CHECK(o.self_collision_first >= 0 && o.self_collision_first <= o.self_collision_last && o.self_collision_last <= P.N,
               GPMP2MI_ERR_INVALID, "self collision: bad state range");
      ex.sc_first = o.self_collision_first;
      ex.sc_last = o.self_collision_last;
      for (int k = 0; k < ex.n_sc; k++) {
        const double a = o.self_collision[k][0], bb = o.self_collision[k][1], sg = o.self_collision[k][3];
        G2_CHECK(a >= 0 && a < h.nr_spheres && bb >= 0 && bb < h.nr_spheres, GPMP2MI_ERR_INVALID, "self collision: sphere id out of range");
        G2_CHECK(sg > 0, GPMP2MI_ERR_INVALID, "self collision: sigma must be positive");
        for (int t = 0; t < 4; t++) scd[4 * k + t] = o.self_collision[k][t];
        ex.sc_w[k] = 1.0 / (sg * sg);
      }
      for (int sidx = 0; sidx < h.nr_spheres; sidx++) radius[h.sph_orig[sidx]] = h.sph_r[sidx];
    }
    p->has_extras = ex.n_ws > 0 || ex.n_sc > 0;
  }

  PlanBuffers& pb = p->pb;
  std::memset(&pb, 0, sizeof(pb));
  const size_t tsz = p->tsz();
  G2_TRY(plan_alloc(p.get(), &pb.params, 1));
  G2_TRY(plan_alloc(p.get(), &pb.start_conf, (size_t)B * D));
  G2_TRY(plan_alloc(p.get(), &pb.start_vel, (size_t)B * D));
  G2_TRY(plan_alloc(p.get(), &pb.end_conf, (size_t)B * D));
  G2_TRY(plan_alloc(p.get(), &pb.end_vel, (size_t)B * D));
  G2_TRY(plan_alloc(p.get(), &pb.cur, tsz));
  G2_TRY(plan_alloc(p.get(), &pb.last, tsz));
  G2_TRY(plan_alloc(p.get(), &pb.trial, tsz));
  G2_TRY(plan_alloc(p.get(), &pb.init, tsz));
  G2_TRY(plan_alloc(p.get(), &pb.result, tsz));
  G2_TRY(plan_alloc(p.get(), &pb.delta, tsz));
  const size_t tq = wide ? 4 : 1;  // wide blocks: 2 x 2 tiles, 32-wide vectors
  G2_TRY(plan_alloc(p.get(), &pb.gvec, (size_t)B * (P.N + 1) * (wide ? 32 : 16)));
  G2_TRY(plan_alloc(p.get(), &pb.htiles, P.opt_type == GPMP2MI_OPT_DOGLEG ? (size_t)B * (P.N + 1) * 512 * tq : 1));
  G2_TRY(plan_alloc(p.get(), &pb.hgpart, (size_t)B * P.Npad));
  G2_TRY(plan_alloc(p.get(), &pb.scal, (size_t)B * SC_COUNT));
  G2_TRY(plan_alloc(p.get(), &pb.which, B));
  G2_TRY(plan_alloc(p.get(), &pb.stepped, B));
  G2_TRY(plan_alloc(p.get(), &pb.spart, (size_t)B * std::max((P.N + 4) / 4, P.Ppad / 64) * 3));
  G2_TRY(plan_alloc(p.get(), &pb.xg, (size_t)B * (P.N + 1) * (wide ? 32 : 16)));
  if (dense_path) {   // dense normal equations + the factors of the dense cyclic reduction
    G2_TRY(plan_alloc(p.get(), &pb.wHd, (size_t)B * (P.N + 1) * P.n * P.n));
    G2_TRY(plan_alloc(p.get(), &pb.wHo, (size_t)B * P.N * P.n * P.n));
    G2_TRY(plan_alloc(p.get(), &pb.wg, (size_t)B * (P.N + 1) * P.n));
    G2_TRY(plan_alloc(p.get(), &pb.wWl, (size_t)B * (P.N + 1) * P.n * P.n));
    G2_TRY(plan_alloc(p.get(), &pb.wWr, (size_t)B * (P.N + 1) * P.n * P.n));
    G2_TRY(plan_alloc(p.get(), &pb.wy, (size_t)B * (P.N + 1) * P.n));
    G2_TRY(plan_alloc(p.get(), &pb.wrb, (size_t)B * (P.N + 1) * P.n));
    G2_TRY(plan_alloc(p.get(), &pb.wx, (size_t)B * (P.N + 1) * P.n));
  }
  G2_TRY(plan_alloc(p.get(), &pb.xp_n, B));
  G2_TRY(plan_alloc(p.get(), &pb.xp_state, (size_t)B * XP_MAX));
  G2_TRY(plan_alloc(p.get(), &pb.xp_has_vel, (size_t)B * XP_MAX));
  G2_TRY(plan_alloc(p.get(), &pb.xp_target, (size_t)B * XP_MAX * P.n));
  G2_TRY(plan_alloc(p.get(), &pb.xp_info, (size_t)B * XP_MAX * 2 * D * D));
  G2_TRY(plan_alloc(p.get(), &pb.goal_on, B));
  {
    std::vector<int> ones(B, 1);
    G2_HIP(hipMemcpy(pb.goal_on, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));
    p->h_xp_n.assign(B, 0);
  }
  G2_TRY(plan_alloc(p.get(), &pb.rec, (size_t)B * P.RECS * P.Ppad));
  G2_TRY(plan_alloc(p.get(), &pb.rec2, (size_t)B * P.RECS * P.Ppad));
  G2_TRY(plan_alloc(p.get(), &pb.gpu, (size_t)B * P.GPS * P.Npad));
  G2_TRY(plan_alloc(p.get(), &pb.gpu2, (size_t)B * P.GPS * P.Npad));
  G2_TRY(plan_alloc(p.get(), &pb.tiles, (size_t)B * (P.N + 1) * 256 * tq));
  G2_TRY(plan_alloc(p.get(), &pb.fac, (size_t)B * (P.N + 1) * 768 * tq));
  G2_TRY(plan_alloc(p.get(), &pb.pend, (size_t)B * ((P.N + 4) / 4) * 256));
  G2_TRY(plan_alloc(p.get(), &pb.coup, (size_t)B * ((P.N + 4) / 4) * 256));
  G2_TRY(plan_alloc(p.get(), &pb.cur_err, B));
  G2_TRY(plan_alloc(p.get(), &pb.prev_err, B));
  G2_TRY(plan_alloc(p.get(), &pb.last_err, B));
  G2_TRY(plan_alloc(p.get(), &pb.final_err, B));
  G2_TRY(plan_alloc(p.get(), &pb.lambda, B));
  G2_TRY(plan_alloc(p.get(), &pb.trace, (size_t)B * (P.max_iter + 1)));
  G2_TRY(plan_alloc(p.get(), &pb.iters, B));
  G2_TRY(plan_alloc(p.get(), &pb.status, B));
  G2_TRY(plan_alloc(p.get(), &pb.active, B));
  G2_TRY(plan_alloc(p.get(), &pb.phase, B));
  G2_TRY(plan_alloc(p.get(), &pb.notspd, B));
  G2_TRY(plan_alloc(p.get(), &pb.epart, (size_t)B * P.Npad));
  {
    const char* e = getenv("GPMP2MI_GENERIC_GN");
    p->generic_gn = e && e[0] == '1';
    p->wide_dense = dense_path;   // GPMP2MI_WIDE_DENSE=1 (read above) or dof > 11
    const int cap = std::max(P.fixed_iters, P.max_iter);   // plan_update may run any iterations <= max_iter
    // passes: GN one per iteration (+1); LM up to ~5 lambda retries per iterate; Dogleg up to ~16 halvings
    const int mult = P.opt_type == GPMP2MI_OPT_LM ? 6 : P.opt_type == GPMP2MI_OPT_DOGLEG ? 18 : 1;
    p->n_active_len = cap * mult + 3;
    P.max_pass = p->n_active_len;
  }
  {
    const RobotDev& h = robot->h;
    const size_t M = (size_t)B * (P.N + 1);
    if (ex.n_ws > 0) {
      G2_TRY(plan_alloc(p.get(), &ex.des, des.size()));
      G2_HIP(hipMemcpy(ex.des, des.data(), des.size() * sizeof(double), hipMemcpyHostToDevice));
      G2_TRY(plan_alloc(p.get(), &ex.poses, M * h.nr_links * 16));
      G2_TRY(plan_alloc(p.get(), &ex.Jp, M * h.nr_links * 6 * D));
      G2_TRY(plan_alloc(p.get(), &ex.ws_err, (size_t)ex.n_ws * M * 6));
      G2_TRY(plan_alloc(p.get(), &ex.ws_H, (size_t)ex.n_ws * M * 6 * D));
    }
    if (ex.n_sc > 0) {
      G2_TRY(plan_alloc(p.get(), &ex.sc_data, scd.size()));
      G2_HIP(hipMemcpy(ex.sc_data, scd.data(), scd.size() * sizeof(double), hipMemcpyHostToDevice));
      G2_TRY(plan_alloc(p.get(), &ex.radius, radius.size()));
      G2_HIP(hipMemcpy(ex.radius, radius.data(), radius.size() * sizeof(double), hipMemcpyHostToDevice));
      G2_TRY(plan_alloc(p.get(), &ex.cen, M * h.nr_spheres * 3));
      G2_TRY(plan_alloc(p.get(), &ex.Jc, M * h.nr_spheres * 3 * D));
      G2_TRY(plan_alloc(p.get(), &ex.sc_err, M * ex.n_sc));
      G2_TRY(plan_alloc(p.get(), &ex.sc_H, M * ex.n_sc * D));
    }
  }
  G2_TRY(plan_alloc(p.get(), &pb.n_active, p->n_active_len));
  G2_TRY(plan_alloc(p.get(), &pb.stamps, (size_t)B * 128));   // rows 0..B-1: kernel phases, rows B..2B-1: one CR task per level
  G2_HIP(hipMemcpy(pb.params, &P, sizeof(P), hipMemcpyHostToDevice));
  G2_TRY(flags_acquire(p->n_active_len, &p->flagbuf));
  g_live_flagbufs.fetch_add(1);
  p->h_flags = p->flagbuf.host;
  pb.host_flags = p->flagbuf.dev;
  G2_TRY(plan_alloc(p.get(), &pb.done, p->n_active_len));
  // the zero fills of the arena chunks ran on the null stream: done before the caller may use any other stream
  G2_HIP(hipStreamSynchronize(nullptr));
  p->null_stream_dirty = false;
  *out = p.release();
  return GPMP2MI_OK;
}

// Waits only for what THIS plan still has in flight (streams it was given since their last synchronisation; nothing
// after the usual optimize -> get_result sequence), never for the device: other host threads' plans keep running.
// A poisoned plan (timed-out pass) is not waited for at all and its memory is not recycled.
void gpmp2mi_plan_destroy(gpmp2mi_plan* p) { delete p; }

static int plan_set_problem(gpmp2mi_plan* p, const double* sc, const double* sv, const double* ec,
                            const double* ev, const double* init, hipMemcpyKind kind, hipStream_t st) {
  G2_CHECK(p && sc && sv && ec && ev && init, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK(!p->poisoned, GPMP2MI_ERR_TIMEOUT, "this plan timed out earlier: destroy it and create a new one");
  const size_t bd = (size_t)p->hp.B * p->hp.D * sizeof(double);
  G2_HIP(hipMemcpyAsync(p->pb.start_conf, sc, bd, kind, st));
  G2_HIP(hipMemcpyAsync(p->pb.start_vel, sv, bd, kind, st));
  G2_HIP(hipMemcpyAsync(p->pb.end_conf, ec, bd, kind, st));
  G2_HIP(hipMemcpyAsync(p->pb.end_vel, ev, bd, kind, st));
  G2_HIP(hipMemcpyAsync(p->pb.init, init, p->tsz() * sizeof(double), kind, st));
  p->mark_dirty(st);
  if (kind == hipMemcpyHostToDevice) {
    G2_HIP(hipStreamSynchronize(st));
    p->mark_clean(st);
  }
  p->problem_set = true;
  p->optimized = false;
  return GPMP2MI_OK;
}
int gpmp2mi_plan_set_problem(gpmp2mi_plan* p, const double* sc, const double* sv, const double* ec,
                             const double* ev, const double* init) {
  return plan_set_problem(p, sc, sv, ec, ev, init, hipMemcpyHostToDevice, nullptr);
}
int gpmp2mi_plan_set_problem_dev(gpmp2mi_plan* p, const double* sc, const double* sv, const double* ec,
                                 const double* ev, const double* init, void* stream) {
  return plan_set_problem(p, sc, sv, ec, ev, init, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

int gpmp2mi_plan_optimize(gpmp2mi_plan* p, void* stream) {
  G2_CHECK(p, GPMP2MI_ERR_INVALID, "null plan");
  G2_CHECK(p->problem_set, GPMP2MI_ERR_INVALID, "call gpmp2mi_plan_set_problem first");
  hipStream_t st = (hipStream_t)stream;
  return plan_run(p, st, p->pb.init);   // cur = init is part of the reset kernel
}

// Active-trajectory count of a finished pass.  The closing kernel of every pass publishes it to a pinned,
// device-mapped flag (publish_pass_count), so there is no copy command or event in the stream; the host spins
// on the flag, falls back to the stream state if the flag never arrives (a faulted kernel) and gives up after a
// wall-clock limit (GPMP2MI_WAIT_TIMEOUT_MS, default 5000) so that a hung kernel cannot hang the caller.
static double wait_timeout_seconds() {
  const char* e = getenv("GPMP2MI_WAIT_TIMEOUT_MS");
  const double ms = e ? atof(e) : 5000.0;
  return (ms > 0 ? ms : 5000.0) * 1e-3;
}
// `st_valid` false: no stream to query (the host-only test hook gpmp2mi_debug_wait_flag)
static int spin_wait_flag(const volatile int* flag, bool st_valid, hipStream_t st, double timeout_s, int* count) {
  const auto t0 = std::chrono::steady_clock::now();
  double next_query = 2e-3;
  for (long spin = 0;; spin++) {
    const int v = __atomic_load_n(flag, __ATOMIC_ACQUIRE);
    if (v >= 0) {
      *count = v;
      return GPMP2MI_OK;
    }
    if ((spin & 0xfff) == 0xfff) {
      const double el0 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      // A stream query is not free on the device side: with work pending the runtime answers it through a marker
      // packet (barrier + completion signal) at the tail of the queue, i.e. between this pass and the next one
      // (rocprofv3 trace: 5.6 us of idle queue per pass boundary when the query ran on every check).  The query only
      // serves to notice a faulted stream early, so it starts after 2 ms of waiting and then runs every 2 ms.
      if (st_valid && el0 >= next_query) {
        next_query = el0 + 2e-3;
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) {  // everything enqueued has run: the flag must be there now
          const int w = __atomic_load_n(flag, __ATOMIC_ACQUIRE);
          G2_CHECK(w >= 0, GPMP2MI_ERR_HIP, "pass count was never published");
          *count = w;
          return GPMP2MI_OK;
        }
        if (e != hipErrorNotReady) G2_HIP(e);
      }
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > timeout_s) {
        set_error("timed out after " + std::to_string((int)(el * 1e3)) +
                  " ms waiting for a pass to finish (kernel hung?); GPMP2MI_WAIT_TIMEOUT_MS raises the limit");
        return GPMP2MI_ERR_TIMEOUT;
      }
    }
  }
}
static int wait_pass_count(gpmp2mi_plan* p, int pass, hipStream_t st, int* count) {
  return spin_wait_flag(p->h_flags + pass, true, st, wait_timeout_seconds(), count);
}

// the optimizer driver: `cur` holds the starting values
static int plan_run_impl(gpmp2mi_plan* p, hipStream_t st, const double* start) {
  const PlanParams& P = p->hp;
  PlanBuffers& pb = p->pb;
  p->timer.reset();
  for (int k = 0; k < p->n_active_len; k++) p->h_flags[k] = -1;  // the previous run has drained (stream sync below)
  G2_TRY(launch_plan_reset(P, pb, start, st));
  const int iter_cap = (P.fixed_iters > 0 ? P.fixed_iters : P.max_iter);
  if (P.opt_type == GPMP2MI_OPT_GAUSS_NEWTON && !p->generic_gn && !P.wide) {
    // ---- Gauss-Newton fast path: 3 launches per pass, step control fused into the solve kernel.
    // Software-pipelined driver: pass k+1 is enqueued before the host looks at the active count of
    // pass k, so the GPU never waits for the host.  When pass k turns out to have finished every
    // trajectory, the already enqueued pass k+1 is a no-op (all workgroups exit on active[b] == 0).
    const int max_pass = iter_cap + 1;
    // How far the host runs ahead.  "pass" (rounds 1-2): pass k+1 is enqueued whole before the count of pass k-1 is
    // looked at, so one idle pass (four empty kernels, ~18 us) follows the last active one.  "lin": only the
    // linearization of pass k+1 is enqueued ahead; its other three kernels follow once the count of pass k is in, which
    // the host learns while the GPU still has the finish kernel of pass k and that linearization (~23 us) to run -- the
    // idle tail shrinks to one empty kernel.
    const char* ahead_env = getenv("GPMP2MI_GN_LOOKAHEAD");
    const bool ahead_lin = !(ahead_env && ahead_env[0] == 'p');
    // Fused finish (P.fuse_finish): there is no k_finish_step; the linearization of pass k applies the step of pass k - 1
    // itself, reading the states of pass k - 1 from one of the plan's two state buffers and writing those of pass k to
    // the other -- cur / last swap roles every pass, the step kernel picks them by the parity of its pass number.
    const bool fuse = P.fuse_finish != 0;
    auto states_of = [&](int pass) -> double* { return (fuse && (pass & 1)) ? pb.last : pb.cur; };
    auto enqueue_lin = [&](int pass) -> int {
      p->timer.begin("linearize", st);
      if (fuse && pass > 0) return plan_linearize(p, states_of(pass - 1), 0, pb.active, st, states_of(pass), pass);
      return plan_linearize(p, pb.cur, 0, pb.active, st);
    };
    auto enqueue_rest = [&](int pass) -> int {
      if (P.fixed_iters > 0 && pass == P.fixed_iters) {
        // closing pass of a fixed-iteration run: nothing is solved any more, only the error of the final values
        p->timer.begin("final_error", st);
        G2_TRY(launch_error_parts(P, pb, states_of(pass), 0, pb.active, st));
      } else {
        p->timer.begin("assemble", st);
        G2_TRY(launch_assemble(P, pb, states_of(pass), 0, pb.active, st));
      }
      p->timer.begin("gn_step_cr", st);
      G2_TRY(launch_gn_step_cr(P, pb, pass, st));
      if (P.split_back && !fuse) {
        p->timer.begin("finish_step", st);
        G2_TRY(launch_finish_step(P, pb, pass, st));
      }
      return GPMP2MI_OK;
    };
    if (ahead_lin) {
      G2_TRY(enqueue_lin(0));
      for (int pass = 0; pass < max_pass; pass++) {
        G2_TRY(enqueue_rest(pass));
        if (pass + 1 == max_pass) break;
        G2_TRY(enqueue_lin(pass + 1));   // ahead of the count
        p->timer.close(st);
        int cnt = 0;
        G2_TRY(wait_pass_count(p, pass, st, &cnt));
        if (cnt == 0) break;
      }
      p->timer.close(st);
    } else {
      for (int pass = 0; pass < max_pass; pass++) {
        G2_TRY(enqueue_lin(pass));
        G2_TRY(enqueue_rest(pass));
        p->timer.close(st);
        if (pass >= 1) {
          int cnt = 0;
          G2_TRY(wait_pass_count(p, pass - 1, st, &cnt));
          if (cnt == 0) break;
        }
      }
    }
  } else {
    // ---- generic trial-step path (LM, Dogleg; GN when forced): per pass
    //   assemble (+ g^T H g) -> solve + trial point -> linearize(trial) into the spare buffer -> decide
    // LM may retry an iterate with a larger lambda, Dogleg with a smaller radius, hence the cap.
    const int max_pass = p->n_active_len - 1;
    p->timer.begin("linearize", st);
    G2_TRY(plan_linearize(p, pb.cur, 0, pb.active, st));
    p->timer.begin("decide", st);
    G2_TRY(launch_decide(P, pb, 0, true, st));
    p->timer.close(st);
    for (int pass = 1; pass < max_pass; pass++) {
      if (P.wide && p->wide_dense) {
        // dof 12..18, or A-B for 8..11: dense normal equations + cyclic reduction over dense blocks
        p->timer.begin("export_dense", st);
        G2_TRY(launch_export_normal_eq(P, pb, pb.cur, 0, pb.wHd, pb.wHo, pb.wg, st, pb.active));
        p->timer.begin("solve_dense", st);
        G2_TRY(launch_solve_dense(P, pb, st));
      } else if (P.wide) {
        // blocks wider than one tile (8 <= dof <= 11): the same cyclic reduction on 2x2 tiles
        p->timer.begin("assemble_wide", st);
        G2_TRY(launch_assemble_wide(P, pb, pb.cur, 0, pb.active, st));
        if (P.opt_type == GPMP2MI_OPT_DOGLEG) {
          p->timer.begin("ghg_wide", st);
          G2_TRY(launch_ghg_wide(P, pb, st));
        }
        for (int h = 2; h < P.wide_h0; h *= 2) {
          p->timer.begin(h == 2 ? "cr_level2_wide" : "cr_level4_wide", st);
          G2_TRY(launch_cr_level_wide(P, pb, h, st));
        }
        p->timer.begin("solve_step_wide", st);
        G2_TRY(launch_solve_step_wide(P, pb, st));
        if (P.split_back && P.opt_type != GPMP2MI_OPT_DOGLEG) {   // LM / GN: levels 4, 2, 1, step and trial point chip-wide
          p->timer.begin("finish_trial_wide", st);
          G2_TRY(launch_finish_trial_wide(P, pb, st));
        }
      } else {
        p->timer.begin("assemble", st);
        G2_TRY(launch_assemble(P, pb, pb.cur, 0, pb.active, st));
        if (P.opt_type == GPMP2MI_OPT_DOGLEG) {
          p->timer.begin("ghg", st);
          G2_TRY(launch_ghg(P, pb, st));
        }
        p->timer.begin("solve_step", st);
        G2_TRY(launch_solve_step(P, pb, st));
        if (P.split_back && P.opt_type != GPMP2MI_OPT_DOGLEG && !P.fuse_finish) {   // LM / GN: levels 2, 1, step and trial point chip-wide
          p->timer.begin("finish_trial", st);
          G2_TRY(launch_finish_trial(P, pb, st));
        }
      }
      p->timer.begin("linearize", st);
      if (!P.wide && P.split_back && P.opt_type != GPMP2MI_OPT_DOGLEG && P.fuse_finish) {
        // fused finish: the linearization forms the trial point cur (+) delta itself (k_linearize_arm, `trial`)
        G2_TRY(plan_linearize(p, pb.cur, 1, pb.active, st, pb.trial, 1, true));
      } else {
        G2_TRY(plan_linearize(p, pb.trial, 1, pb.active, st));
      }
      p->timer.begin("decide", st);
      G2_TRY(launch_decide(P, pb, pass, false, st));
      p->timer.close(st);
      if (pass >= 2) {
        int cnt = 0;
        G2_TRY(wait_pass_count(p, pass - 1, st, &cnt));
        if (cnt == 0) break;
      }
    }
    // pass budget spent with trajectories still iterating (many consecutive rejected trial steps): they
    // finish with their current values and status MAX_ITER instead of returning a stale `result`
    G2_TRY(launch_finalize_unfinished(P, pb, st));
  }
  G2_HIP(hipStreamSynchronize(st));
  if (p->timer.enabled) p->timer.collect();
  p->optimized = true;
  return GPMP2MI_OK;
}
static int plan_run(gpmp2mi_plan* p, hipStream_t st, const double* start) {
  G2_CHECK(!p->poisoned, GPMP2MI_ERR_TIMEOUT,
           "this plan timed out earlier and may still have a hung kernel in its stream: destroy it and create a new one");
  p->mark_dirty(st);
  const int rc = plan_run_impl(p, st, start);
  if (rc == GPMP2MI_ERR_TIMEOUT) {
    // The stream may hold a kernel that never finishes: waiting for it here (or in gpmp2mi_plan_destroy) would hang
    // the caller after all.  The plan is poisoned instead: no further runs, no wait and no recycling at destroy.
    p->poisoned = true;
    return rc;
  }
  // any other error: the next run resets the host flags assuming the stream has drained
  if (rc != GPMP2MI_OK) (void)hipStreamSynchronize(st);
  p->mark_clean(st);   // plan_run_impl ends with a stream synchronisation as well
  return rc;
}

static int plan_get_result(gpmp2mi_plan* p, double* traj, int* iters, double* ferr, int* status,
                           double* trace, hipMemcpyKind kind, hipStream_t st) {
  G2_CHECK(p, GPMP2MI_ERR_INVALID, "null plan");
  G2_CHECK(!p->poisoned, GPMP2MI_ERR_TIMEOUT, "this plan timed out earlier: destroy it and create a new one");
  G2_CHECK(p->optimized, GPMP2MI_ERR_INVALID, "plan has not been optimized");
  const int B = p->hp.B;
  if (traj) G2_HIP(hipMemcpyAsync(traj, p->pb.result, p->tsz() * sizeof(double), kind, st));
  if (iters) G2_HIP(hipMemcpyAsync(iters, p->pb.iters, B * sizeof(int), kind, st));
  if (ferr) G2_HIP(hipMemcpyAsync(ferr, p->pb.final_err, B * sizeof(double), kind, st));
  if (status) G2_HIP(hipMemcpyAsync(status, p->pb.status, B * sizeof(int), kind, st));
  if (trace)
    G2_HIP(hipMemcpyAsync(trace, p->pb.trace, (size_t)B * (p->hp.max_iter + 1) * sizeof(double), kind, st));
  p->mark_dirty(st);
  if (kind == hipMemcpyDeviceToHost) {
    G2_HIP(hipStreamSynchronize(st));
    p->mark_clean(st);
  }
  return GPMP2MI_OK;
}
int gpmp2mi_plan_get_result(gpmp2mi_plan* p, double* traj, int* iters, double* ferr, int* status, double* trace) {
  return plan_get_result(p, traj, iters, ferr, status, trace, hipMemcpyDeviceToHost, nullptr);
}
int gpmp2mi_plan_get_result_dev(gpmp2mi_plan* p, double* traj, int* iters, double* ferr, int* status, void* stream) {
  return plan_get_result(p, traj, iters, ferr, status, nullptr, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}
const double* gpmp2mi_plan_traj_dev(const gpmp2mi_plan* p) { return p ? p->pb.result : nullptr; }

int gpmp2mi_plan_graph_error(gpmp2mi_plan* p, const double* traj, double* err) {
  G2_CHECK(p && traj && err, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK(p->problem_set, GPMP2MI_ERR_INVALID, "call gpmp2mi_plan_set_problem first");
  DevBuf<double> dt, de;
  G2_TRY(dt.upload(traj, p->tsz()));
  G2_TRY(de.alloc(p->hp.B));
  G2_TRY(plan_linearize(p, dt.p, 1, nullptr, nullptr));
  G2_TRY(launch_error_reduce(p->hp, p->pb, dt.p, 1, de.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(de.download(err));
  return GPMP2MI_OK;
}

int gpmp2mi_plan_linearize(gpmp2mi_plan* p, const double* traj, double* Hdiag, double* Hoff, double* g, double* err) {
  G2_CHECK(p && traj, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK(p->problem_set, GPMP2MI_ERR_INVALID, "call gpmp2mi_plan_set_problem first");
  const PlanParams& P = p->hp;
  const size_t nb = (size_t)P.B * (P.N + 1), n = P.n;
  DevBuf<double> dt, dd, dob, dg, de;
  G2_TRY(dt.upload(traj, p->tsz()));
  if (Hdiag) G2_TRY(dd.alloc(nb * n * n));
  if (Hoff) G2_TRY(dob.alloc((size_t)P.B * P.N * n * n));
  if (g) G2_TRY(dg.alloc(nb * n));
  if (err) G2_TRY(de.alloc(P.B));
  // evaluate into the spare record buffer (the one that does not hold the linearization at `cur`)
  const PlanBuffers& pb = p->pb;
  G2_TRY(plan_linearize(p, dt.p, 1, nullptr, nullptr));
  G2_TRY(launch_export_normal_eq(P, pb, dt.p, 1, dd.p, dob.p, dg.p, nullptr));
  if (err) G2_TRY(launch_error_reduce(P, pb, dt.p, 1, de.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(dd.download(Hdiag));
  G2_TRY(dob.download(Hoff));
  G2_TRY(dg.download(g));
  G2_TRY(de.download(err));
  return GPMP2MI_OK;
}

// -------------------------------------------------------------------------------------------- replanning
static int plan_add_prior(gpmp2mi_plan* p, int b, int state, const double* conf, const double* Wc, const double* vel,
                          const double* Wv) {
  G2_CHECK(p && conf && Wc, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK(b >= 0 && b < p->hp.B && state >= 0 && state <= p->hp.N, GPMP2MI_ERR_INVALID, "index out of range");
  G2_CHECK(p->h_xp_n[b] < XP_MAX, GPMP2MI_ERR_UNSUPPORTED, "too many state priors on this trajectory");
  const int D = p->hp.D, n = p->hp.n, e = p->h_xp_n[b];
  const size_t xe = (size_t)b * XP_MAX + e;
  std::vector<double> tg(n, 0.0), info(2 * D * D, 0.0);
  std::copy(conf, conf + D, tg.begin());
  std::copy(Wc, Wc + D * D, info.begin());
  const int has_vel = (vel && Wv) ? 1 : 0;
  if (has_vel) {
    std::copy(vel, vel + D, tg.begin() + D);
    std::copy(Wv, Wv + D * D, info.begin() + D * D);
  }
  G2_HIP(hipMemcpy(p->pb.xp_target + xe * n, tg.data(), n * sizeof(double), hipMemcpyHostToDevice));
  G2_HIP(hipMemcpy(p->pb.xp_info + xe * 2 * D * D, info.data(), info.size() * sizeof(double), hipMemcpyHostToDevice));
  G2_HIP(hipMemcpy(p->pb.xp_state + xe, &state, sizeof(int), hipMemcpyHostToDevice));
  G2_HIP(hipMemcpy(p->pb.xp_has_vel + xe, &has_vel, sizeof(int), hipMemcpyHostToDevice));
  p->h_xp_n[b] = e + 1;
  G2_HIP(hipMemcpy(p->pb.xp_n + b, &p->h_xp_n[b], sizeof(int), hipMemcpyHostToDevice));
  return GPMP2MI_OK;
}

int gpmp2mi_plan_fix_state(gpmp2mi_plan* p, int b, int state_idx, const double* conf, const double* vel) {
  G2_CHECK(p && conf && vel, GPMP2MI_ERR_INVALID, "null argument");
  const int D = p->hp.D;
  std::vector<double> Wc(D * D, 0.0), Wv(D * D, 0.0);
  for (int k = 0; k < D; k++) {
    Wc[k * D + k] = p->hp.conf_prior_w;
    Wv[k * D + k] = p->hp.vel_prior_w;
  }
  return plan_add_prior(p, b, state_idx, conf, Wc.data(), vel, Wv.data());
}

int gpmp2mi_plan_add_state_estimate(gpmp2mi_plan* p, int b, int state_idx, const double* conf, const double* conf_cov,
                                    const double* vel, const double* vel_cov) {
  G2_CHECK(p && conf && conf_cov, GPMP2MI_ERR_INVALID, "null argument");
  G2_CHECK((vel == nullptr) == (vel_cov == nullptr), GPMP2MI_ERR_INVALID, "pass vel and vel_cov together");
  const int D = p->hp.D;
  std::vector<double> Wc(D * D), Wv(D * D);
  G2_CHECK(invert_small(D, conf_cov, Wc.data()), GPMP2MI_ERR_INVALID, "pose covariance is singular");
  if (vel) G2_CHECK(invert_small(D, vel_cov, Wv.data()), GPMP2MI_ERR_INVALID, "velocity covariance is singular");
  return plan_add_prior(p, b, state_idx, conf, Wc.data(), vel, vel ? Wv.data() : nullptr);
}

int gpmp2mi_plan_change_goal(gpmp2mi_plan* p, int b, const double* goal_conf, const double* goal_vel) {
  G2_CHECK(p && goal_conf && goal_vel && b >= 0 && b < p->hp.B, GPMP2MI_ERR_INVALID, "bad argument");
  const int D = p->hp.D, one = 1;
  G2_HIP(hipMemcpy(p->pb.end_conf + (size_t)b * D, goal_conf, D * sizeof(double), hipMemcpyHostToDevice));
  G2_HIP(hipMemcpy(p->pb.end_vel + (size_t)b * D, goal_vel, D * sizeof(double), hipMemcpyHostToDevice));
  G2_HIP(hipMemcpy(p->pb.goal_on + b, &one, sizeof(int), hipMemcpyHostToDevice));
  return GPMP2MI_OK;
}

int gpmp2mi_plan_remove_goal(gpmp2mi_plan* p, int b) {
  G2_CHECK(p && b >= 0 && b < p->hp.B, GPMP2MI_ERR_INVALID, "bad argument");
  const int zero = 0;
  G2_HIP(hipMemcpy(p->pb.goal_on + b, &zero, sizeof(int), hipMemcpyHostToDevice));
  return GPMP2MI_OK;
}

int gpmp2mi_plan_clear_state_priors(gpmp2mi_plan* p, int b) {
  G2_CHECK(p && b >= 0 && b < p->hp.B, GPMP2MI_ERR_INVALID, "bad argument");
  p->h_xp_n[b] = 0;
  G2_HIP(hipMemcpy(p->pb.xp_n + b, &p->h_xp_n[b], sizeof(int), hipMemcpyHostToDevice));
  return GPMP2MI_OK;
}

int gpmp2mi_plan_update(gpmp2mi_plan* p, int iterations, void* stream) {
  G2_CHECK(p && iterations > 0, GPMP2MI_ERR_INVALID, "bad argument");
  G2_CHECK(p->problem_set, GPMP2MI_ERR_INVALID, "call gpmp2mi_plan_set_problem first");
  G2_CHECK(iterations <= p->hp.max_iter, GPMP2MI_ERR_INVALID, "iterations exceeds max_iter");
  G2_CHECK(iterations + 3 <= p->n_active_len, GPMP2MI_ERR_INVALID, "iterations exceeds the plan's pass budget");
  hipStream_t st = (hipStream_t)stream;
  // warm start: the previous estimate becomes the initial values of this run
  const double* from = p->optimized ? p->pb.result : p->pb.init;
  // temporarily switch the resident parameters to `iterations` fixed Gauss-Newton steps
  PlanParams saved = p->hp;
  p->hp.opt_type = GPMP2MI_OPT_GAUSS_NEWTON;
  p->hp.fixed_iters = iterations;
  if (const int rc0 = launch_set_mode(p->pb, p->hp.opt_type, p->hp.fixed_iters, st)) {
    p->hp = saved;
    return rc0;
  }
  const bool gg = p->generic_gn;
  p->generic_gn = false;
  const int rc = plan_run(p, st, from);
  p->generic_gn = gg;
  p->hp = saved;
  G2_TRY(launch_set_mode(p->pb, p->hp.opt_type, p->hp.fixed_iters, st));
  G2_HIP(hipStreamSynchronize(st));
  return rc;
}

int gpmp2mi_batch_optimize(const gpmp2mi_robot* robot, const gpmp2mi_sdf* sdf, const gpmp2mi_settings* s,
                           const gpmp2mi_graph_opts* o, int B, const double* sc, const double* sv,
                           const double* ec, const double* ev, const double* init, double* traj_out,
                           int* iters, double* ferr, int* status) {
  gpmp2mi_plan* p = nullptr;
  int rc = gpmp2mi_plan_create(robot, sdf, s, o, B, &p);
  if (rc) return rc;
  rc = gpmp2mi_plan_set_problem(p, sc, sv, ec, ev, init);
  if (!rc) rc = gpmp2mi_plan_optimize(p, nullptr);
  if (!rc) rc = gpmp2mi_plan_get_result(p, traj_out, iters, ferr, status, nullptr);
  gpmp2mi_plan_destroy(p);
  return rc;
}

int gpmp2mi_collision_cost(const gpmp2mi_robot* r, const gpmp2mi_sdf* s, int total_step, int B,
                           const double* traj, double* cost) {
  // internal::CollisionCost planner/BatchTrajOptimizer-inl.h:87-100: unary obstacle error with
  // epsilon = 0 summed over all states; evaluated on device, summed on the host.
  G2_CHECK(r && s && traj && cost && B >= 0 && total_step >= 0, GPMP2MI_ERR_INVALID, "null argument");
  const int D = r->h.dof, S = r->h.nr_spheres, M = B * (total_step + 1);
  std::vector<double> conf((size_t)M * D), err((size_t)M * S);
  for (int m = 0; m < M; m++)
    for (int k = 0; k < D; k++) conf[(size_t)m * D + k] = traj[(size_t)m * 2 * D + k];
  G2_TRY(gpmp2mi_obstacle_factor(r, s, 0.0, M, conf.data(), err.data(), nullptr));
  for (int b = 0; b < B; b++) {
    double c = 0.0;
    for (int i = 0; i <= total_step; i++)
      for (int k = 0; k < S; k++) c += err[((size_t)b * (total_step + 1) + i) * S + k];
    cost[b] = c;
  }
  return GPMP2MI_OK;
}

// diagnostic: raw s_memtime stamps of the last step kernel (all zero unless built with -DG2_STAMPS)
int gpmp2mi_plan_debug_stamps(gpmp2mi_plan* p, int b, unsigned long long* out64) {
  // rows B .. 2B - 1 hold the per-task stamps of the cyclic reduction (G2_TSTAMP) of trajectory b - B
  G2_CHECK(p && out64 && b >= 0 && b < 2 * p->hp.B, GPMP2MI_ERR_INVALID, "bad argument");
  G2_HIP(hipMemcpy(out64, p->pb.stamps + (size_t)b * 64, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return GPMP2MI_OK;
}

// diagnostic: the scalars of trajectory b's last trial step (PlanBuffers::scal, see plan.h SC_*) and its
// current lambda / trust radius in out[16]
int gpmp2mi_plan_debug_scalars(gpmp2mi_plan* p, int b, double* out17) {
  G2_CHECK(p && out17 && b >= 0 && b < p->hp.B, GPMP2MI_ERR_INVALID, "bad argument");
  G2_HIP(hipMemcpy(out17, p->pb.scal + (size_t)b * SC_COUNT, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost));
  G2_HIP(hipMemcpy(out17 + SC_COUNT, p->pb.lambda + b, sizeof(double), hipMemcpyDeviceToHost));
  return GPMP2MI_OK;
}

// diagnostic: out[8][64] = {bcast_row<0..3>, bcast_in_row<5>, row_sum16, sum_rows, bcast_in_row<13>}(in[64])
int gpmp2mi_debug_crosslane(const double* in64, double* out512) {
  G2_CHECK(in64 && out512, GPMP2MI_ERR_INVALID, "null argument");
  G2_TRY(ensure_device());
  DevBuf<double> di, dout;
  G2_TRY(di.upload(in64, 64));
  G2_TRY(dout.alloc(512));
  G2_TRY(launch_debug_crosslane(di.p, dout.p, nullptr));
  G2_HIP(hipStreamSynchronize(nullptr));
  G2_TRY(dout.download(out512));
  return GPMP2MI_OK;
}

// test hook (host only, no GPU needed): the bounded spin of the pass driver on a caller-owned flag
int gpmp2mi_debug_wait_flag(const int* flag, int timeout_ms, int* value) {
  G2_CHECK(flag && value && timeout_ms > 0, GPMP2MI_ERR_INVALID, "bad argument");
  return spin_wait_flag(flag, false, nullptr, timeout_ms * 1e-3, value);
}

// diagnostic: raw copy of one of the solver's hand-over buffers (0 tiles, 1 fac, 2 pend, 3 coup) to the host
int gpmp2mi_plan_debug_read(gpmp2mi_plan* p, int which, double* out, long count) {
  G2_CHECK(p && out && count >= 0, GPMP2MI_ERR_INVALID, "bad argument");
  const double* src = which == 0 ? p->pb.tiles : which == 1 ? p->pb.fac : which == 2 ? p->pb.pend : which == 3 ? p->pb.coup : nullptr;
  G2_CHECK(src, GPMP2MI_ERR_INVALID, "unknown buffer");
  G2_HIP(hipMemcpy(out, src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
  return GPMP2MI_OK;
}

// test hook: what the library currently holds (arena chunks / flag buffers owned by live plans, pooled ones, plans
// leaked because they were poisoned).  Works without a GPU (all zeros then).
int gpmp2mi_debug_resource_counts(long* live_chunks, long* pooled_chunks, long* live_flagbufs, long* pooled_flagbufs,
                                  long* leaked_plans) {
  std::lock_guard<std::mutex> lk(g_flag_mu);
  if (live_chunks) *live_chunks = g_live_chunks.load();
  if (pooled_chunks) *pooled_chunks = (long)g_chunk_pool.size();
  if (live_flagbufs) *live_flagbufs = g_live_flagbufs.load();
  if (pooled_flagbufs) *pooled_flagbufs = (long)g_flag_pool.size();
  if (leaked_plans) *leaked_plans = g_leaked_plans.load();
  return GPMP2MI_OK;
}

// test hook: a kernel that occupies `stream` until gpmp2mi_debug_stall_release (or, whatever happens, until max_ms
// of device wall clock have passed: every wave reaches that exit), so that the pass driver's timeout path can be driven
// on a real stream.  One thread; polls a host-mapped word.
struct gpmp2mi_stall_token {
  int* host = nullptr;
  int* dev = nullptr;
  hipStream_t st = nullptr;
};
// test hooks: a non-blocking HIP stream from the runtime this library is linked against (a test process must not pull
// in a second HIP runtime just to get a stream)
int gpmp2mi_debug_stream_create(void** stream) {
  G2_CHECK(stream, GPMP2MI_ERR_INVALID, "null argument");
  G2_TRY(ensure_device());
  hipStream_t st = nullptr;
  G2_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *stream = st;
  return GPMP2MI_OK;
}
int gpmp2mi_debug_stream_destroy(void* stream) {
  if (stream) G2_HIP(hipStreamDestroy((hipStream_t)stream));
  return GPMP2MI_OK;
}
int gpmp2mi_debug_stall_begin(void* stream, int max_ms, void** token) {
  G2_CHECK(token && max_ms > 0 && max_ms <= 10000, GPMP2MI_ERR_INVALID, "bad argument");
  G2_TRY(ensure_device());
  auto t = std::make_unique<gpmp2mi_stall_token>();
  G2_HIP(hipHostMalloc((void**)&t->host, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
  *t->host = 0;
  G2_HIP(hipHostGetDevicePointer((void**)&t->dev, t->host, 0));
  t->st = (hipStream_t)stream;
  k_debug_stall<<<dim3(1), dim3(1), 0, t->st>>>(t->dev, (long long)max_ms * 100000LL);   // s_memrealtime: 100 MHz
  G2_HIP(hipGetLastError());
  *token = t.release();
  return GPMP2MI_OK;
}
int gpmp2mi_debug_stall_release(void* token) {
  auto* t = static_cast<gpmp2mi_stall_token*>(token);
  G2_CHECK(t && t->host, GPMP2MI_ERR_INVALID, "null token");
  __atomic_store_n(t->host, 1, __ATOMIC_RELEASE);
  const hipError_t e = hipStreamSynchronize(t->st);
  (void)hipHostFree(t->host);
  delete t;
  G2_HIP(e);
  return GPMP2MI_OK;
}

int gpmp2mi_plan_enable_timing(gpmp2mi_plan* p, int enable) {
  G2_CHECK(p, GPMP2MI_ERR_INVALID, "null plan");
  p->timer.enabled = enable != 0;
  return GPMP2MI_OK;
}
int gpmp2mi_plan_get_timing(gpmp2mi_plan* p, int* n, const char** names, double* ms, int* launches) {
  G2_CHECK(p && n, GPMP2MI_ERR_INVALID, "null argument");
  const int cap = *n;
  const int have = (int)p->timer.names.size();
  *n = have;
  for (int i = 0; i < std::min(cap, have); i++) {
    if (names) names[i] = p->timer.cnames[i];
    if (ms) ms[i] = p->timer.ms[i];
    if (launches) launches[i] = p->timer.launches[i];
  }
  return GPMP2MI_OK;
}

}  // extern "C"
