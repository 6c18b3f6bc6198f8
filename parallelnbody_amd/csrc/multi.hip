// Single-process multi-GPU engine (see multi.h): sub-contexts driven through the public C-ABI, RCCL over xGMI between them.
#include "multi.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only: the functions are resolved with dlsym

#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

namespace nbody {

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// One handle per library name, process-wide; a library stays loaded (RCCL does not like being unloaded under live
// communicators).  NBODY_RCCL_LIB names another library with the same eight entry points: the test suite's stand-in
// (tests/cpp/fake_rccl.c), which lets one GPU play several devices — see NBODY_MULTI_SHARE_DEVICE in multi_create.
bool load_rccl(Rccl *r, std::string *err) {
  struct Loaded { Rccl api; bool ok = false; std::string why; };
  static std::map<std::string, Loaded> cache;
  const char *named = getenv("NBODY_RCCL_LIB");
  const std::string key = named && named[0] ? named : "";
  auto it = cache.find(key);
  if (it == cache.end()) {
    Loaded l;
    Rccl &g = l.api;
    if (!key.empty()) {
      g.handle = dlopen(key.c_str(), RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g.handle) break;
      }
    }
    if (!g.handle) {
      const char *de = dlerror();
      l.why = std::string("cannot load RCCL (") + (key.empty() ? "librccl.so.1" : key.c_str()) + "): " + (de ? de : "not found");
    } else {
      bool all = true;
      auto sym = [&](const char *n) { void *p = dlsym(g.handle, n); if (!p) { all = false; l.why = std::string("RCCL lacks ") + n; } return p; };
      g.CommInitAll = (decltype(g.CommInitAll))sym("ncclCommInitAll");
      g.CommDestroy = (decltype(g.CommDestroy))sym("ncclCommDestroy");
      g.GroupStart = (decltype(g.GroupStart))sym("ncclGroupStart");
      g.GroupEnd = (decltype(g.GroupEnd))sym("ncclGroupEnd");
      g.AllGather = (decltype(g.AllGather))sym("ncclAllGather");
      g.Send = (decltype(g.Send))sym("ncclSend");
      g.Recv = (decltype(g.Recv))sym("ncclRecv");
      g.GetErrorString = (decltype(g.GetErrorString))sym("ncclGetErrorString");
      l.ok = all;
    }
    it = cache.emplace(key, l).first;
  }
  if (!it->second.ok) { if (err) *err = it->second.why; return false; }
  *r = it->second.api;
  return true;
}

}  // namespace

struct Multi {
  nbody_params p;
  float theta = 0.0f;                      // > 0: Barnes-Hut frames (multi_bh_frames)
  int n_dev = 0;
  int32_t slice = 0;                       // bodies per device
  bool f64 = false;
  size_t elem = 16;
  std::vector<int> devices;
  std::vector<nbody_ctx *> part;
  std::vector<hipStream_t> stream;
  std::vector<ncclComm_t> comm;
  // the all-gather has a communicator and a stream of its own per device: it is queued behind the update by an event and
  // the next step's force pass waits for it only where it needs the other devices' positions (gather_pending)
  std::vector<ncclComm_t> comm_gather;
  std::vector<hipStream_t> gather_stream;
  std::vector<hipEvent_t> ev_updated, ev_gathered;
  bool gather_pending = false;
  bool overlap = true;                     // NBODY_MULTI_NO_OVERLAP=1: everything in one stream order (A/B measurements)
  std::vector<void *> posm;                // each device's full position array
  // symmetric algorithm: exchange buffers of every device (null when the step has no exchange)
  std::vector<void *> ex_send, ex_recv;
  size_t ex_bytes = 0;
  int ex_ranks = 0;
  Rccl rccl;
  std::string err;
};

namespace {

int fail(Multi *m, int code, const std::string &msg) { m->err = msg; return code; }

int part_fail(Multi *m, int k, int rc, const char *what) {
  return fail(m, rc, std::string(what) + " on device " + std::to_string(m->devices[(size_t)k]) + ": " + nbody_last_error(m->part[(size_t)k]));
}

int rccl_fail(Multi *m, ncclResult_t r, const char *what) {
  return fail(m, NBODY_ERR_HIP, std::string(what) + ": " + (m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "RCCL error"));
}

#define PART_TRY(m, k, expr, what) do { int rc_ = (expr); if (rc_) return part_fail((m), (k), rc_, (what)); } while (0)
#define RCCL_TRY(m, expr, what) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return rccl_fail((m), r_, (what)); } while (0)

// The owned slices -> every device's full position array: one in-place all-gather per device, grouped — on the gather
// streams, behind an event the update has recorded; the devices' own streams go on at once.
// in_order: on the devices' own streams whatever m->overlap says (a theta > 0 frame reads every position in its first kernel)
int gather_positions(Multi *m, bool in_order = false) {
  const ncclDataType_t ty = m->f64 ? ncclDouble : ncclFloat;
  const bool overlap = m->overlap && !in_order;
  for (int k = 0; k < m->n_dev; ++k) {
    (void)hipSetDevice(m->devices[(size_t)k]);
    if (!overlap) continue;
    if (hipEventRecord(m->ev_updated[(size_t)k], m->stream[(size_t)k]) != hipSuccess ||
        hipStreamWaitEvent(m->gather_stream[(size_t)k], m->ev_updated[(size_t)k], 0) != hipSuccess)
      return fail(m, NBODY_ERR_HIP, "hipEventRecord / hipStreamWaitEvent (update -> gather)");
  }
  RCCL_TRY(m, m->rccl.GroupStart(), "ncclGroupStart");
  for (int k = 0; k < m->n_dev; ++k) {
    char *base = (char *)m->posm[(size_t)k];
    (void)hipSetDevice(m->devices[(size_t)k]);              // one thread, several devices: each call on its communicator's device
    ncclResult_t r = m->rccl.AllGather(base + (size_t)k * m->slice * m->elem, base, (size_t)m->slice * 4, ty,
                                       overlap ? m->comm_gather[(size_t)k] : m->comm[(size_t)k],
                                       overlap ? m->gather_stream[(size_t)k] : m->stream[(size_t)k]);
    if (r != ncclSuccess) { (void)m->rccl.GroupEnd(); return rccl_fail(m, r, "ncclAllGather(positions)"); }
  }
  RCCL_TRY(m, m->rccl.GroupEnd(), "ncclGroupEnd");
  if (overlap) {
    for (int k = 0; k < m->n_dev; ++k) {
      (void)hipSetDevice(m->devices[(size_t)k]);
      if (hipEventRecord(m->ev_gathered[(size_t)k], m->gather_stream[(size_t)k]) != hipSuccess)
        return fail(m, NBODY_ERR_HIP, "hipEventRecord (gather)");
    }
    m->gather_pending = true;
  }
  return NBODY_OK;
}

// Order every device's own stream behind the last all-gather (no host wait).  Whatever reads the replicated positions
// outside the force pass's first go comes through here.
int wait_gather(Multi *m) {
  if (!m->gather_pending) return NBODY_OK;
  for (int k = 0; k < m->n_dev; ++k) {
    (void)hipSetDevice(m->devices[(size_t)k]);
    if (hipStreamWaitEvent(m->stream[(size_t)k], m->ev_gathered[(size_t)k], 0) != hipSuccess)
      return fail(m, NBODY_ERR_HIP, "hipStreamWaitEvent (gather -> force pass)");
  }
  m->gather_pending = false;
  return NBODY_OK;
}

// Symmetric algorithm: device k's send segment q -> device q's recv segment k (an all-to-all as grouped send/recv).
int exchange_sums(Multi *m) {
  if (m->ex_ranks == 0) return NBODY_OK;
  const ncclDataType_t ty = m->f64 ? ncclDouble : ncclFloat;
  const size_t count = m->ex_bytes / (m->f64 ? 8 : 4);
  RCCL_TRY(m, m->rccl.GroupStart(), "ncclGroupStart");
  for (int k = 0; k < m->n_dev; ++k)
    for (int q = 0; q < m->n_dev; ++q) {
      (void)hipSetDevice(m->devices[(size_t)k]);
      ncclResult_t r = m->rccl.Send((const char *)m->ex_send[(size_t)k] + (size_t)q * m->ex_bytes, count, ty, q, m->comm[(size_t)k],
                                    m->stream[(size_t)k]);
      if (r == ncclSuccess)
        r = m->rccl.Recv((char *)m->ex_recv[(size_t)k] + (size_t)q * m->ex_bytes, count, ty, q, m->comm[(size_t)k], m->stream[(size_t)k]);
      if (r != ncclSuccess) { (void)m->rccl.GroupEnd(); return rccl_fail(m, r, "ncclSend/ncclRecv(j-side sums)"); }
    }
  RCCL_TRY(m, m->rccl.GroupEnd(), "ncclGroupEnd");
  return NBODY_OK;
}

}  // namespace

int multi_create(const nbody_params *pin, const int32_t *devices, int32_t n_dev, Multi **out, std::string *err) {
  auto bad = [&](int code, const std::string &msg) { if (err) *err = msg; return code; };
  if (!pin || !devices || !out || n_dev < 1) return bad(NBODY_ERR_INVALID, "nbody_create_multi: null argument or n_dev < 1");
  if (pin->struct_size != sizeof(nbody_params)) return bad(NBODY_ERR_INVALID, "nbody_create_multi: struct_size mismatch");
  if (pin->i_begin != 0 || (pin->i_count != 0 && pin->i_count != pin->n_total))
    return bad(NBODY_ERR_INVALID, "nbody_create_multi: the multi-device context owns all bodies (i_begin = 0, i_count = 0)");
  if (pin->n_total <= 0 || pin->n_total % n_dev != 0)
    return bad(NBODY_ERR_INVALID, "nbody_create_multi: n_total must be a positive multiple of n_dev (equal slices)");
  if (pin->theta > 0.0f && pin->precision != NBODY_PREC_F32)
    return bad(NBODY_ERR_UNSUPPORTED, "nbody_create_multi: theta > 0 (Barnes-Hut) needs an fp32 context");
  // NBODY_MULTI_SHARE_DEVICE=1 (test suites only, together with NBODY_RCCL_LIB: real RCCL refuses two ranks on one device):
  // a device may be listed several times, so that one GPU runs every index of this file that a node of eight would
  const char *share = getenv("NBODY_MULTI_SHARE_DEVICE");
  if (!(share && share[0] == '1'))
    for (int a = 0; a < n_dev; ++a)
      for (int b = a + 1; b < n_dev; ++b)
        if (devices[a] == devices[b]) return bad(NBODY_ERR_INVALID, "nbody_create_multi: a device is listed twice");
  Multi *m = new (std::nothrow) Multi();
  if (!m) return bad(NBODY_ERR_NOMEM, "nbody_create_multi: out of host memory");
  std::string why;
  if (!load_rccl(&m->rccl, &why)) { delete m; return bad(NBODY_ERR_UNSUPPORTED, "nbody_create_multi: " + why); }
  m->p = *pin;
  m->theta = pin->theta;
  m->n_dev = n_dev;
  m->slice = pin->n_total / n_dev;
  m->f64 = pin->precision == NBODY_PREC_F64;
  m->elem = m->f64 ? 32 : 16;
  m->devices.assign(devices, devices + n_dev);
  auto bail = [&](int code, const std::string &msg) { multi_destroy(m); return bad(code, msg); };
  for (int k = 0; k < n_dev; ++k) {
    nbody_params q = *pin;
    q.device = devices[k];
    q.i_begin = k * m->slice;
    q.i_count = m->slice;
    nbody_ctx *c = nullptr;
    const int rc = nbody_create(&q, &c);
    if (rc) return bail(rc, std::string("nbody_create_multi: device ") + std::to_string(devices[k]) + ": " + nbody_last_error(nullptr));
    m->part.push_back(c);
    hipStream_t s = nullptr;
    if (hipSetDevice(devices[k]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
      return bail(NBODY_ERR_HIP, "nbody_create_multi: cannot create a stream on device " + std::to_string(devices[k]));
    m->stream.push_back(s);
    if (nbody_set_stream(c, (void *)s)) return bail(NBODY_ERR_HIP, "nbody_create_multi: nbody_set_stream failed");
    void *ptr = nullptr;
    if (nbody_device_ptr(c, NBODY_BUF_POSM, &ptr, nullptr)) return bail(NBODY_ERR_HIP, "nbody_create_multi: no position buffer");
    m->posm.push_back(ptr);
    void *snd = nullptr, *rcv = nullptr; size_t bytes = 0; int32_t ranks = 0;
    (void)nbody_exchange_info(c, &snd, &rcv, &bytes, &ranks);
    if (k == 0) { m->ex_ranks = ranks; m->ex_bytes = bytes; }
    else if (ranks != m->ex_ranks || bytes != m->ex_bytes)
      return bail(NBODY_ERR_STATE, "nbody_create_multi: the devices chose different force-pass geometries");
    m->ex_send.push_back(snd); m->ex_recv.push_back(rcv);
  }
  if (m->ex_ranks != 0 && m->ex_ranks != n_dev) return bail(NBODY_ERR_STATE, "nbody_create_multi: exchange geometry does not match n_dev");
  m->comm.assign((size_t)n_dev, nullptr);
  const ncclResult_t r = m->rccl.CommInitAll(m->comm.data(), n_dev, m->devices.data());
  if (r != ncclSuccess) {
    m->comm.clear();
    return bail(NBODY_ERR_HIP, std::string("nbody_create_multi: ncclCommInitAll: ") + m->rccl.GetErrorString(r));
  }
  { const char *e = getenv("NBODY_MULTI_NO_OVERLAP"); m->overlap = !(e && e[0] == '1'); }
  if (m->overlap) {
    m->comm_gather.assign((size_t)n_dev, nullptr);
    const ncclResult_t r2 = m->rccl.CommInitAll(m->comm_gather.data(), n_dev, m->devices.data());
    if (r2 != ncclSuccess) {
      m->comm_gather.clear();
      return bail(NBODY_ERR_HIP, std::string("nbody_create_multi: ncclCommInitAll (gather): ") + m->rccl.GetErrorString(r2));
    }
    for (int k = 0; k < n_dev; ++k) {
      hipStream_t gs = nullptr; hipEvent_t a = nullptr, b = nullptr;
      if (hipSetDevice(devices[k]) != hipSuccess || hipStreamCreateWithFlags(&gs, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess)
        return bail(NBODY_ERR_HIP, "nbody_create_multi: cannot create the gather stream / events on device " + std::to_string(devices[k]));
      m->gather_stream.push_back(gs); m->ev_updated.push_back(a); m->ev_gathered.push_back(b);
    }
  }
  *out = m;
  return NBODY_OK;
}

void multi_destroy(Multi *m) {
  if (!m) return;
  for (size_t k = 0; k < m->stream.size(); ++k) { (void)hipSetDevice(m->devices[k]); (void)hipStreamSynchronize(m->stream[k]); }
  for (size_t k = 0; k < m->gather_stream.size(); ++k) { (void)hipSetDevice(m->devices[k]); (void)hipStreamSynchronize(m->gather_stream[k]); }
  for (ncclComm_t c : m->comm) if (c) (void)m->rccl.CommDestroy(c);
  for (ncclComm_t c : m->comm_gather) if (c) (void)m->rccl.CommDestroy(c);
  for (size_t k = 0; k < m->gather_stream.size(); ++k) {
    (void)hipSetDevice(m->devices[k]);
    (void)hipStreamDestroy(m->gather_stream[k]); (void)hipEventDestroy(m->ev_updated[k]); (void)hipEventDestroy(m->ev_gathered[k]);
  }
  for (size_t k = 0; k < m->part.size(); ++k) {
    // the context must let go of our stream before the stream dies
    (void)nbody_set_stream(m->part[k], nullptr);
    nbody_destroy(m->part[k]);
  }
  for (size_t k = 0; k < m->stream.size(); ++k) { (void)hipSetDevice(m->devices[k]); (void)hipStreamDestroy(m->stream[k]); }
  delete m;
}

const std::string &multi_error(const Multi *m) { return m->err; }
int multi_parts(const Multi *m) { return m->n_dev; }
nbody_ctx *multi_part(const Multi *m, int k) { return m->part[(size_t)k]; }
void multi_slice(const Multi *m, int k, int32_t *i_begin, int32_t *i_count) {
  if (i_begin) *i_begin = k * m->slice;
  if (i_count) *i_count = m->slice;
}

int multi_set_particles(Multi *m, const void *aos, size_t stride, int32_t n, bool keep_history) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k)
    PART_TRY(m, k, keep_history ? nbody_push_particles(m->part[(size_t)k], aos, stride, n) : nbody_set_particles(m->part[(size_t)k], aos, stride, n),
             keep_history ? "nbody_push_particles" : "nbody_set_particles");
  return NBODY_OK;
}
int multi_set_state_soa(Multi *m, const float *posm4, const float *vel4, int32_t n) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_set_state_soa(m->part[(size_t)k], posm4, vel4, n), "nbody_set_state_soa");
  return NBODY_OK;
}
int multi_set_state_soa_f64(Multi *m, const double *posm4, const double *vel4, int32_t n) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_set_state_soa_f64(m->part[(size_t)k], posm4, vel4, n), "nbody_set_state_soa_f64");
  return NBODY_OK;
}

// One Tick body over all devices (OctreeSearch.cpp:27-31): everything is queued on the devices' streams, nothing waits
// for the host.
static int multi_bh_frames(Multi *m, float dt, int nframes, bool diagnostic, int *built_out);

int multi_forces(Multi *m, float dt) {
  if (m->theta > 0.0f) { int built = 0; return multi_bh_frames(m, dt, 1, !(dt > 0.0f), &built); }
  // the strips inside every device's own slice need no other device's positions: they run while the last step's
  // all-gather is still in flight; the rest of the pass is ordered behind it (nbody_step_begin_local / _remote)
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_step_begin_local(m->part[(size_t)k]), "force pass (own slice)");
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_step_begin_remote(m->part[(size_t)k]), "force pass");
  { const int rc = exchange_sums(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_step_end(m->part[(size_t)k], dt), "update");
  if (dt > 0.0f) return gather_positions(m);
  return NBODY_OK;
}

// ---- theta > 0 -----------------------------------------------------------------------------------------------------------------
// The reference's frame (OctreeSearch.cpp:25-31 with CreateOctree, .cpp:74-89) over several devices: the tree is ONE tree (.cpp:79-81)
// and each body's walk (.cpp:83-86) reads it and writes that body alone — so every device builds the whole tree from its copy of the
// positions (the build is the reference's arithmetic in a fixed order: the same bits everywhere), walks and integrates its own slice,
// and the in-place all-gather brings the moved bodies to everyone: every byte equals the one-device context's.  One caller thread:
// all devices' frames and the gathers between them are queued first, then every device is waited for once.  A frame the sort from
// the previous order gives up (kernels_bh_sort.hip) is given up on every device alike; what it and the frames behind it left undone is
// queued again, the first of them with the cold sorts.
static int multi_bh_frames(Multi *m, float dt, int nframes, bool diagnostic, int *built_out) {
  *built_out = 0;
  { const int rc = wait_gather(m); if (rc) return rc; }
  const bool moves = dt > 0.0f;
  int todo = nframes;
  while (todo > 0) {
    int left = todo < 64 ? todo : 64;                          // batches, as nbody_step has them: a given-up frame takes the ones queued behind it along
    todo -= left;
    while (left > 0) {
      for (int f = 0; f < left; ++f) {
        for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, part_bh_queue_frame(m->part[(size_t)k], dt, diagnostic), "Barnes-Hut frame");
        if (moves) { const int rc = gather_positions(m, true); if (rc) return rc; }
      }
      int status = 0, built = 0, refused_rc = NBODY_OK, refused_k = 0;
      for (int k = 0; k < m->n_dev; ++k) {                     // every device is collected (a refusal is cleared by that), then the verdicts compared
        int st = 0, b = 0;
        const int rc = part_bh_collect(m->part[(size_t)k], &st, &b);
        if (rc && st != 1 && st != 2 && st != 4) return part_fail(m, k, rc, "Barnes-Hut frame");
        if (rc && !refused_rc) { refused_rc = rc; refused_k = k; }
        if (k == 0) { status = st; built = b; }
        else if (st != status || b != built)
          return fail(m, NBODY_ERR_STATE, "Barnes-Hut frames: device " + std::to_string(m->devices[(size_t)k]) + " built " + std::to_string(b) +
                      " frames (status " + std::to_string(st) + "), device " + std::to_string(m->devices[0]) + " " + std::to_string(built) +
                      " (status " + std::to_string(status) + "): the devices' trees differ");
      }
      *built_out += built;
      left -= built;
      if (refused_rc) return part_fail(m, refused_k, refused_rc, "Barnes-Hut frame");   // the state is that of the frames built, on every device
      if (status != 3) break;                                  // (3: the warm sort gave a frame up; `left` frames again)
    }
  }
  return NBODY_OK;
}

int multi_bh_steps(Multi *m, float dt, int nsteps, int *built) { return multi_bh_frames(m, dt, nsteps, false, built); }

int multi_set_theta(Multi *m, float theta) {
  if (theta > 0.0f && m->f64) return fail(m, NBODY_ERR_UNSUPPORTED, "nbody_set_theta: Barnes-Hut needs an fp32 context");
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_set_theta(m->part[(size_t)k], theta), "nbody_set_theta");
  m->theta = theta;
  return NBODY_OK;
}

// every device holds the whole tree: the first one answers
int multi_bh_stats(Multi *m, int32_t *nodes, int32_t *levels, float root_com[3]) {
  PART_TRY(m, 0, nbody_bh_stats(m->part[0], nodes, levels, root_com), "nbody_bh_stats");
  return NBODY_OK;
}
int multi_bh_leaf_boxes(Multi *m, float *boxes, size_t stride) {
  PART_TRY(m, 0, nbody_bh_leaf_boxes(m->part[0], boxes, stride), "nbody_bh_leaf_boxes");
  return NBODY_OK;
}
int multi_bh_leaf_order(Multi *m, int32_t *order) {
  PART_TRY(m, 0, nbody_bh_leaf_order(m->part[0], order), "nbody_bh_leaf_order");
  return NBODY_OK;
}
int multi_bh_root(Multi *m, float root_com[3], int *has_root) {
  PART_TRY(m, 0, part_bh_root(m->part[0], root_com, has_root), "Barnes-Hut root");
  return NBODY_OK;
}

int multi_get_bounds(Multi *m, float *size) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  float best = 0.0f;
  for (int k = 0; k < m->n_dev; ++k) {
    float s = 0.0f;
    PART_TRY(m, k, nbody_get_bounds(m->part[(size_t)k], &s), "nbody_get_bounds");
    if (s > best) best = s;
  }
  *size = best;
  return NBODY_OK;
}

int multi_get_positions(Multi *m, float *xyz, size_t stride, int32_t first, int32_t count) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  PART_TRY(m, 0, nbody_get_positions(m->part[0], xyz, stride, first, count), "nbody_get_positions");   // every device holds all positions
  return NBODY_OK;
}

int multi_get_particles(Multi *m, void *aos, size_t stride) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k)
    PART_TRY(m, k, nbody_get_particles(m->part[(size_t)k], (char *)aos + (size_t)k * m->slice * stride, stride), "nbody_get_particles");
  return NBODY_OK;
}

int multi_get_state_soa(Multi *m, float *posm4, float *vel4, float *acc4) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) {
    const size_t o = (size_t)k * m->slice * 4;
    PART_TRY(m, k, nbody_get_state_soa(m->part[(size_t)k], posm4 ? posm4 + o : nullptr, vel4 ? vel4 + o : nullptr, acc4 ? acc4 + o : nullptr),
             "nbody_get_state_soa");
  }
  return NBODY_OK;
}
int multi_get_state_soa_f64(Multi *m, double *posm4, double *vel4, double *acc4) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) {
    const size_t o = (size_t)k * m->slice * 4;
    PART_TRY(m, k, nbody_get_state_soa_f64(m->part[(size_t)k], posm4 ? posm4 + o : nullptr, vel4 ? vel4 + o : nullptr, acc4 ? acc4 + o : nullptr),
             "nbody_get_state_soa_f64");
  }
  return NBODY_OK;
}

int multi_energy(Multi *m, double *ke, double *pe) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  double k_sum = 0.0, p_sum = 0.0;
  for (int k = 0; k < m->n_dev; ++k) {
    double a = 0.0, b = 0.0;
    PART_TRY(m, k, nbody_energy(m->part[(size_t)k], &a, &b), "nbody_energy");
    k_sum += a; p_sum += b;
  }
  if (ke) *ke = k_sum;
  if (pe) *pe = p_sum;
  return NBODY_OK;
}

int multi_synchronize(Multi *m) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_synchronize(m->part[(size_t)k]), "nbody_synchronize");
  return NBODY_OK;
}

int multi_kernel_time(Multi *m, int32_t which, double *total_ms, int64_t *launches) {
  double worst = 0.0; int64_t n0 = 0;
  for (int k = 0; k < m->n_dev; ++k) {
    double ms = 0.0; int64_t n = 0;
    PART_TRY(m, k, nbody_kernel_time(m->part[(size_t)k], which, &ms, &n), "nbody_kernel_time");
    if (ms > worst) worst = ms;
    if (k == 0) n0 = n;
  }
  if (total_ms) *total_ms = worst;
  if (launches) *launches = n0;
  return NBODY_OK;
}
int multi_kernel_time_reset(Multi *m) {
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_kernel_time_reset(m->part[(size_t)k]), "nbody_kernel_time_reset");
  return NBODY_OK;
}

int multi_kernel_clock(Multi *m, double *shader_mhz, int32_t *compute_units) {
  double slowest = 0.0;
  for (int k = 0; k < m->n_dev; ++k) {
    double mhz = 0.0; int32_t cus = 0;
    PART_TRY(m, k, nbody_kernel_clock(m->part[(size_t)k], &mhz, &cus), "nbody_kernel_clock");
    if (k == 0 || (mhz > 0.0 && mhz < slowest)) slowest = mhz;
    if (compute_units && k == 0) *compute_units = cus;
  }
  if (shader_mhz) *shader_mhz = slowest;
  return NBODY_OK;
}

int multi_load_checkpoint(Multi *m, const char *path, int64_t *steps_done) {
  { const int rc = wait_gather(m); if (rc) return rc; }
  for (int k = 0; k < m->n_dev; ++k) PART_TRY(m, k, nbody_load_checkpoint(m->part[(size_t)k], path, steps_done), "nbody_load_checkpoint");
  (void)nbody_get_theta(m->part[0], &m->theta);               // the file's opening angle: every device took it over
  return NBODY_OK;
}

}  // namespace nbody
