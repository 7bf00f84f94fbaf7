// Cycle plans (include/mghip.h, "Cycle plans"): the decomposed solver's per-cycle work of one rank, recorded once by
// the host side and replayed here with one call per cycle -- fused legs, halo copies, RCCL groups, the coarse gather,
// the replicated engine and the norm all-reduce, on two HIP streams.  Uses only the public entry points of the library
// plus RCCL, which is resolved at run time from the copy the process already has loaded (no link-time dependency).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mghip.h"

namespace {

thread_local std::string g_plan_error;

int plan_fail(std::string* where, int code, const std::string& msg) {
  if (where) *where = msg;
  g_plan_error = msg;
  return code;
}

// ---- RCCL, resolved by name ------------------------------------------------------------------
// The handful of declarations used here, as rccl.h (NCCL 2.x API) has them.
typedef struct ncclComm* ncclComm_t;
struct NcclId { char internal[128]; };
enum { kNcclInt8 = 0, kNcclFloat64 = 8, kNcclSum = 0 };

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
};

Rccl g_rccl;

int load_rccl(const char* path) {
  if (g_rccl.lib) return MG_OK;
  void* lib = dlopen(path && *path ? path : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return plan_fail(nullptr, MG_ERR_STATE, std::string("cannot load RCCL: ") + dlerror());
  Rccl r;
  r.lib = lib;
  bool ok = true;
  auto sym = [&](const char* name) { void* s = dlsym(lib, name); ok = ok && s; return s; };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
  r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  if (!ok) return plan_fail(nullptr, MG_ERR_STATE, "the RCCL library lacks a symbol of the NCCL 2 API");
  g_rccl = r;
  return MG_OK;
}

struct Comm {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
};

// ---- device helpers --------------------------------------------------------------------------
__global__ void add_f64_kernel(double* dst, const double* a, const double* b) { *dst = b ? *a + *b : *a; }

// Up to kBatch independent 2-D copies in one launch (blockIdx.z = copy): halo packing / unpacking and the scatter of the
// gathered coarse blocks are runs of small copies; one launch per run instead of one per copy.
constexpr int kBatch = 8;
struct CopyBatch {
  uint32_t* dst[kBatch];
  const uint32_t* src[kBatch];
  int rows[kBatch], words[kBatch];
  int64_t dpitch[kBatch], spitch[kBatch];
  int n, max_rows, max_words;
};
__global__ void __launch_bounds__(256) copy2d_batch_kernel(CopyBatch b) {
  const int k = blockIdx.z, r = blockIdx.y;
  if (r >= b.rows[k]) return;
  uint32_t* __restrict__ d = b.dst[k] + r * b.dpitch[k];
  const uint32_t* __restrict__ s = b.src[k] + r * b.spitch[k];
  for (int c = blockIdx.x * 256 + threadIdx.x; c < b.words[k]; c += gridDim.x * 256) d[c] = s[c];
}

}  // namespace

struct mg_plan {
  std::vector<mg_plan_op> ops;
  std::vector<int> batch_of;            // per op: -1, or the index into `batches` the op starts (members that follow are skipped)
  std::vector<CopyBatch> batches;
  std::vector<int> batch_len;
  Comm* comm = nullptr;
  int device = 0;
  hipEvent_t ev[8] = {};
  double* result_dev = nullptr;
  double* result_host = nullptr;       // pinned
  hipEvent_t fence[2] = {};            // compute -> communication stream and back, around RCCL calls recorded on the compute stream
  hipEvent_t done = nullptr;           // recorded behind the copy of the RESULT
  bool pending = false;                // a RESULT is in flight (mg_plan_run_async) and not collected yet (mg_plan_wait)
  bool profile = false;                // mg_plan_profile: bracket every operation with timing events
  struct Span { int phase; hipEvent_t e0, e1; };
  std::vector<Span> spans;             // recorded while `profile`, consumed by mg_plan_phase_times
  std::string err;
};

namespace {

#define PLAN_HIP(plan, call)                                                                                           \
  do {                                                                                                                 \
    hipError_t e_ = (call);                                                                                            \
    if (e_ != hipSuccess)                                                                                              \
      return plan_fail(&(plan)->err, MG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                   \
  } while (0)

int nccl_fail(mg_plan* p, int rc, const char* what) {
  return plan_fail(&p->err, MG_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}

// phase of an operation for mg_plan_phase_times (include/mghip.h)
int phase_of(int op) {
  switch (op) {
    case MG_PLAN_DOWN_LEG: case MG_PLAN_UP_LEG: case MG_PLAN_SPAN_LEG: return 0;
    case MG_PLAN_COPY2D: return 1;
    case MG_PLAN_GROUP_BEGIN: case MG_PLAN_SEND: case MG_PLAN_RECV: case MG_PLAN_GROUP_END: return 2;
    case MG_PLAN_ALLGATHER: return 3;
    case MG_PLAN_COARSE_BEGIN: case MG_PLAN_COARSE_CYCLE: case MG_PLAN_COARSE_END: return 4;
    case MG_PLAN_ADD_F64: case MG_PLAN_ALLREDUCE_F64: return 5;
    default: return 6;
  }
}

bool needs_comm(int op) {
  return op == MG_PLAN_GROUP_BEGIN || op == MG_PLAN_GROUP_END || op == MG_PLAN_SEND || op == MG_PLAN_RECV ||
         op == MG_PLAN_ALLGATHER || op == MG_PLAN_ALLREDUCE_F64;
}

int check_op(const mg_plan_op& o, int idx, std::string* err) {
  auto bad = [&](const char* why) { return plan_fail(err, MG_ERR_INVALID_VALUE, "plan op " + std::to_string(idx) + ": " + why); };
  if (o.stream != 0 && o.stream != 1) return bad("stream must be 0 or 1");
  switch (o.op) {
    case MG_PLAN_DOWN_LEG:
      if (!o.p[1] || !o.p[2] || !o.p[3] || (!o.i[12] && !o.p[0])) return bad("down leg: NULL field");
      return MG_OK;
    case MG_PLAN_UP_LEG:
      if (!o.p[0] || !o.p[1] || !o.p[2] || !o.p[3] || (o.i[15] && (!o.p[4] || !o.p[5]))) return bad("up leg: NULL field");
      return MG_OK;
    case MG_PLAN_SPAN_LEG:
      if (!o.p[0] || !o.p[1] || !o.p[3] || !o.p[4] || !o.p[5] || !o.p[6] || !o.p[7]) return bad("spanning leg: NULL field");
      return MG_OK;
    case MG_PLAN_COPY2D:
      if (!o.p[0] || !o.p[1] || o.i[0] < 0 || o.i[1] < 0 || (o.i[1] & 3) || (o.i[2] & 3) || (o.i[3] & 3) ||
          (reinterpret_cast<uintptr_t>(o.p[0]) & 3) || (reinterpret_cast<uintptr_t>(o.p[1]) & 3) ||
          (o.i[0] > 1 && (o.i[2] < o.i[1] || o.i[3] < o.i[1])))
        return bad("copy2d: NULL pointer, or sizes / pitches / pointers that are not multiples of 4 bytes");
      return MG_OK;
    case MG_PLAN_ADD_F64:
      return (o.p[0] && o.p[1]) ? MG_OK : bad("add: NULL pointer");
    case MG_PLAN_GROUP_BEGIN:
    case MG_PLAN_GROUP_END:
      return MG_OK;
    case MG_PLAN_SEND:
    case MG_PLAN_RECV:
      return (o.p[0] && o.i[0] >= 0 && o.i[1] >= 0) ? MG_OK : bad("send/recv: NULL buffer or negative peer / size");
    case MG_PLAN_ALLGATHER:
      return (o.p[0] && o.p[1] && o.i[0] >= 0) ? MG_OK : bad("allgather: NULL buffer");
    case MG_PLAN_ALLREDUCE_F64:
      return (o.p[0] && o.i[0] >= 1) ? MG_OK : bad("allreduce: NULL buffer");
    case MG_PLAN_COARSE_BEGIN:
    case MG_PLAN_COARSE_END:
      return (o.p[0] && o.p[1]) ? MG_OK : bad("coarse begin/end: NULL handle or field");
    case MG_PLAN_COARSE_CYCLE:
      return (o.p[0] && o.i[0] >= 0) ? MG_OK : bad("coarse cycle: NULL handle");
    case MG_PLAN_EVENT_RECORD:
    case MG_PLAN_STREAM_WAIT:
      return (o.i[0] >= 0 && o.i[0] < 8) ? MG_OK : bad("event id out of range");
    case MG_PLAN_RESULT:
      return o.p[0] ? MG_OK : bad("result: NULL pointer");
    default:
      return bad("unknown operation");
  }
}

// Runs of consecutive COPY2D operations on one stream whose regions do not touch (bounding byte ranges: a copy that reads what
// an earlier one of the run writes -- the column phase of a halo exchange after the row phase -- starts a new run).
void build_batches(mg_plan* p) {
  const int n = static_cast<int>(p->ops.size());
  p->batch_of.assign(n, -1);
  struct Region { uintptr_t lo, hi; int width, pitch; };
  auto region = [](const void* ptr, int rows, int width, int pitch) {
    Region r;
    r.lo = reinterpret_cast<uintptr_t>(ptr);
    r.hi = r.lo + (rows > 0 ? static_cast<uintptr_t>(rows - 1) * pitch + width : 0);
    r.width = width; r.pitch = pitch;
    return r;
  };
  // disjoint byte ranges, or -- same pitch -- disjoint column bands (the two ghost columns of one array)
  auto touch = [](const Region& a, const Region& b) {
    if (!(a.lo < b.hi && b.lo < a.hi)) return false;
    if (a.pitch == b.pitch && a.pitch > 0) {
      const uintptr_t P = static_cast<uintptr_t>(a.pitch), ca = a.lo % P, cb = b.lo % P;
      if (ca + a.width <= P && cb + b.width <= P && (ca + a.width <= cb || cb + b.width <= ca)) return false;
    }
    return true;
  };
  int k = 0;
  while (k < n) {
    if (p->ops[k].op != MG_PLAN_COPY2D) { ++k; continue; }
    CopyBatch b{};
    std::vector<Region> writes, reads;
    int m = k;
    while (m < n && b.n < kBatch && p->ops[m].op == MG_PLAN_COPY2D && p->ops[m].stream == p->ops[k].stream) {
      const mg_plan_op& o = p->ops[m];
      const Region d = region(o.p[0], o.i[0], o.i[1], o.i[2]), sr = region(o.p[1], o.i[0], o.i[1], o.i[3]);
      bool clash = false;
      for (auto& w : writes) clash = clash || touch(sr, w) || touch(d, w);
      for (auto& r : reads) clash = clash || touch(d, r);
      if (clash) break;
      if (o.i[0] > 0 && o.i[1] > 0) {
        const int j = b.n++;
        b.dst[j] = static_cast<uint32_t*>(o.p[0]); b.src[j] = static_cast<const uint32_t*>(o.p[1]);
        b.rows[j] = o.i[0]; b.words[j] = o.i[1] / 4; b.dpitch[j] = o.i[2] / 4; b.spitch[j] = o.i[3] / 4;
        b.max_rows = b.max_rows > o.i[0] ? b.max_rows : o.i[0];
        b.max_words = b.max_words > o.i[1] / 4 ? b.max_words : o.i[1] / 4;
        writes.push_back(d);
        reads.push_back(sr);
      }
      ++m;
    }
    p->batch_of[k] = static_cast<int>(p->batches.size());
    p->batches.push_back(b);
    p->batch_len.push_back(m - k);
    k = m;
  }
}

}  // namespace

extern "C" {

const char* mg_plan_error(const mg_plan* plan) { return plan ? plan->err.c_str() : g_plan_error.c_str(); }

int mg_comm_unique_id(const char* rccl_library, void* id128) {
  if (!id128) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_comm_unique_id: NULL argument");
  if (int rc = load_rccl(rccl_library)) return rc;
  NcclId id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) return plan_fail(nullptr, MG_ERR_HIP, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc));
  std::memcpy(id128, id.internal, sizeof(id.internal));
  return MG_OK;
}

int mg_comm_init(const char* rccl_library, const void* id128, int nranks, int rank, int device, void** comm) {
  if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks)
    return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_comm_init: bad argument");
  *comm = nullptr;
  if (int rc = load_rccl(rccl_library)) return rc;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return plan_fail(nullptr, MG_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  NcclId id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  Comm* c = new Comm;
  c->nranks = nranks; c->rank = rank; c->device = device;
  const int rc = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
  if (rc != 0) {
    delete c;
    return plan_fail(nullptr, MG_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc));
  }
  *comm = c;
  return MG_OK;
}

int mg_comm_destroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return MG_OK;
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  delete c;
  return MG_OK;
}

int mg_plan_create(const mg_plan_op* ops, int n_ops, void* comm, int device, mg_plan** out) {
  if (!ops || n_ops < 1 || !out) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_plan_create: bad argument");
  *out = nullptr;
  int results = 0, depth = 0;
  for (int k = 0; k < n_ops; ++k) {
    if (int rc = check_op(ops[k], k, nullptr)) return rc;
    if (needs_comm(ops[k].op) && !comm)
      return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "plan op " + std::to_string(k) + " communicates but the plan has no communicator");
    if (ops[k].op == MG_PLAN_GROUP_BEGIN) ++depth;
    if (ops[k].op == MG_PLAN_GROUP_END && --depth < 0) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "GROUP_END without GROUP_BEGIN");
    if ((ops[k].op == MG_PLAN_SEND || ops[k].op == MG_PLAN_RECV)) {
      const Comm* c = static_cast<const Comm*>(comm);
      if (ops[k].i[0] >= c->nranks) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "plan op " + std::to_string(k) + ": peer out of range");
    }
    results += (ops[k].op == MG_PLAN_RESULT);
  }
  if (depth != 0) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "GROUP_BEGIN without GROUP_END");
  if (results > 1) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "a plan has at most one RESULT");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return plan_fail(nullptr, MG_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  mg_plan* p = new mg_plan;
  p->ops.assign(ops, ops + n_ops);
  p->comm = static_cast<Comm*>(comm);
  p->device = device;
  build_batches(p);
  for (auto& ev : p->ev) {
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) { mg_plan_destroy(p); return plan_fail(nullptr, MG_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
  }
  for (auto& ev : p->fence) {
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) { mg_plan_destroy(p); return plan_fail(nullptr, MG_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
  }
  e = hipEventCreateWithFlags(&p->done, hipEventDisableTiming);
  if (e != hipSuccess) { mg_plan_destroy(p); return plan_fail(nullptr, MG_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
  e = hipHostMalloc(reinterpret_cast<void**>(&p->result_host), sizeof(double), hipHostMallocDefault);
  if (e != hipSuccess) { mg_plan_destroy(p); return plan_fail(nullptr, MG_ERR_ALLOC, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  *out = p;
  return MG_OK;
}

int mg_plan_num_ops(const mg_plan* plan, int* n) {
  if (!plan || !n) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_plan_num_ops: NULL argument");
  *n = static_cast<int>(plan->ops.size());
  return MG_OK;
}

int mg_plan_copy_launches(const mg_plan* plan, int* n_copies, int* n_launches) {
  if (!plan || !n_copies || !n_launches) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_plan_copy_launches: NULL argument");
  int copies = 0, launches = 0;
  for (const auto& o : plan->ops) copies += (o.op == MG_PLAN_COPY2D);
  for (const auto& b : plan->batches) launches += (b.n > 0);
  *n_copies = copies;
  *n_launches = launches;
  return MG_OK;
}

int mg_plan_profile(mg_plan* p, int enable) {
  if (!p) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_plan_profile: NULL plan");
  p->profile = enable != 0;
  return MG_OK;
}

int mg_plan_phase_times(mg_plan* p, double* out_ms) {
  if (!p || !out_ms) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_plan_phase_times: NULL argument");
  PLAN_HIP(p, hipSetDevice(p->device));
  int rc = MG_OK;
  for (auto& sp : p->spans) {
    if (sp.e0 && sp.e1 && rc == MG_OK) {
      float ms = 0.f;
      if (hipEventSynchronize(sp.e1) != hipSuccess || hipEventElapsedTime(&ms, sp.e0, sp.e1) != hipSuccess)
        rc = plan_fail(&p->err, MG_ERR_HIP, "mg_plan_phase_times: a timing event could not be read");
      else out_ms[sp.phase] += ms;
    }
    if (sp.e0) (void)hipEventDestroy(sp.e0);
    if (sp.e1) (void)hipEventDestroy(sp.e1);
  }
  p->spans.clear();
  return rc;
}

int mg_comm_ranks(void* comm, int* nranks, int* rank) {
  const Comm* c = static_cast<const Comm*>(comm);
  if (!c || !nranks || !rank) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_comm_ranks: NULL argument");
  *nranks = c->nranks; *rank = c->rank;
  return MG_OK;
}

int mg_plan_destroy(mg_plan* p) {
  if (!p) return MG_OK;
  for (auto& sp : p->spans) { if (sp.e0) (void)hipEventDestroy(sp.e0); if (sp.e1) (void)hipEventDestroy(sp.e1); }
  for (auto& ev : p->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : p->fence)
    if (ev) (void)hipEventDestroy(ev);
  if (p->done) (void)hipEventDestroy(p->done);
  if (p->result_host) (void)hipHostFree(p->result_host);
  delete p;
  return MG_OK;
}

int mg_plan_run_async(mg_plan* p, void* compute_stream, void* comm_stream) {
  if (!p) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_plan_run_async: NULL plan");
  PLAN_HIP(p, hipSetDevice(p->device));
  p->pending = false;
  hipStream_t st[2] = {static_cast<hipStream_t>(compute_stream), static_cast<hipStream_t>(comm_stream)};
  const double* result_dev = nullptr;
  hipStream_t result_stream = st[0];
  auto lib_fail = [&](int rc, int k, const char* what) {
    const char* m = mg_last_error(nullptr);
    return plan_fail(&p->err, rc, "plan op " + std::to_string(k) + " (" + what + "): " + (m ? m : ""));
  };
  // Every RCCL call of this process on this communicator goes to ONE stream -- the communication stream -- whatever stream it
  // was recorded on (what torch.distributed does with its own stream): a call recorded on the compute stream is fenced in
  // and out with two events, so successive collectives never reach the communicator from two streams.
  bool redirected = false;
  int group_depth = 0;
  for (size_t k = 0; k < p->ops.size(); ++k) {
    const mg_plan_op& o = p->ops[k];
    hipStream_t s = st[o.stream];
    const bool comm_op = needs_comm(o.op);
    if (comm_op && o.stream == 0 && st[0] != st[1]) {
      if (!redirected) {
        PLAN_HIP(p, hipEventRecord(p->fence[0], st[0]));
        PLAN_HIP(p, hipStreamWaitEvent(st[1], p->fence[0], 0));
        redirected = true;
      }
      s = st[1];
    }
    const int32_t* i = o.i;
    // diagnostics: a timing-event pair around the operation (a send/recv group: around the whole group) on its stream
    const bool span_op = p->profile && o.op != MG_PLAN_EVENT_RECORD && o.op != MG_PLAN_STREAM_WAIT && o.op != MG_PLAN_RESULT;
    hipEvent_t span_e0 = nullptr;
    if (span_op && !(group_depth > 0)) {
      PLAN_HIP(p, hipEventCreate(&span_e0));
      PLAN_HIP(p, hipEventRecord(span_e0, s));
      p->spans.push_back({phase_of(o.op), span_e0, nullptr});
    }
    switch (o.op) {
      case MG_PLAN_DOWN_LEG: {
        const int rc = mg_dev_down_leg_var(i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10], o.d[0], o.d[1], o.d[2],
                                           o.d[3], i[11], i[12], i[13], o.p[0], o.p[1], o.p[2], o.p[3], s, i[14],
                                           i[15] ? &i[16] : nullptr, o.p[4], o.p[5]);
        if (rc != MG_OK) return lib_fail(rc, static_cast<int>(k), "down leg");
        break;
      }
      case MG_PLAN_UP_LEG: {
        const int rc = mg_dev_up_leg_var(i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10], i[11], i[12], o.d[0], o.d[1],
                                         o.d[2], o.d[3], i[13], i[14], o.p[0], o.p[1], o.p[2], o.p[3], i[15], i[16], i[17], i[18], i[19],
                                         o.p[4], static_cast<double*>(o.p[5]), s, o.p[6], o.p[7]);
        if (rc != MG_OK) return lib_fail(rc, static_cast<int>(k), "up leg");
        break;
      }
      case MG_PLAN_SPAN_LEG: {
        const int rc = mg_dev_span_leg(i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10], i[11], i[12], o.d[0], o.d[1],
                                       o.d[2], o.d[3], i[13], i[14], i[15], o.p[0], o.p[1], o.p[2], o.p[3], o.p[4], o.p[5], i[16], i[17],
                                       i[18], i[19], o.p[6], static_cast<double*>(o.p[7]), s);
        if (rc != MG_OK) return lib_fail(rc, static_cast<int>(k), "spanning leg");
        break;
      }
      case MG_PLAN_COPY2D: {
        const int bi = p->batch_of[k];                    // every run of copies starts with its batch
        const CopyBatch& b = p->batches[bi];
        if (b.n > 0) {
          int gx = (b.max_words + 255) / 256;
          if (gx > 64) gx = 64;
          copy2d_batch_kernel<<<dim3(gx, b.max_rows, b.n), 256, 0, s>>>(b);
        }
        k += p->batch_len[bi] - 1;
        break;
      }
      case MG_PLAN_ADD_F64:
        add_f64_kernel<<<1, 1, 0, s>>>(static_cast<double*>(o.p[0]), static_cast<const double*>(o.p[1]), static_cast<const double*>(o.p[2]));
        break;
      case MG_PLAN_GROUP_BEGIN:
        if (int rc = g_rccl.GroupStart()) return nccl_fail(p, rc, "ncclGroupStart");
        break;
      case MG_PLAN_GROUP_END:
        if (int rc = g_rccl.GroupEnd()) return nccl_fail(p, rc, "ncclGroupEnd");
        break;
      case MG_PLAN_SEND:
        if (int rc = g_rccl.Send(o.p[0], static_cast<size_t>(i[1]), kNcclInt8, i[0], p->comm->comm, s)) return nccl_fail(p, rc, "ncclSend");
        break;
      case MG_PLAN_RECV:
        if (int rc = g_rccl.Recv(o.p[0], static_cast<size_t>(i[1]), kNcclInt8, i[0], p->comm->comm, s)) return nccl_fail(p, rc, "ncclRecv");
        break;
      case MG_PLAN_ALLGATHER:
        if (int rc = g_rccl.AllGather(o.p[0], o.p[1], static_cast<size_t>(i[0]), kNcclInt8, p->comm->comm, s))
          return nccl_fail(p, rc, "ncclAllGather");
        break;
      case MG_PLAN_ALLREDUCE_F64:
        if (int rc = g_rccl.AllReduce(o.p[0], o.p[0], static_cast<size_t>(i[0]), kNcclFloat64, kNcclSum, p->comm->comm, s))
          return nccl_fail(p, rc, "ncclAllReduce");
        break;
      case MG_PLAN_COARSE_BEGIN: {
        mg_handle* h = static_cast<mg_handle*>(o.p[0]);
        int rc = mg_set_stream(h, s, 0);
        if (rc == MG_OK) rc = i[2] ? mg_update_rhs_device(h, o.p[1], i[0], i[1]) : mg_set_rhs_device(h, o.p[1], i[0], i[1]);
        if (rc == MG_OK) rc = mg_zero_solution_device(h);
        if (rc != MG_OK) return plan_fail(&p->err, rc, std::string("coarse begin: ") + mg_last_error(h));
        break;
      }
      case MG_PLAN_COARSE_CYCLE: {
        mg_handle* h = static_cast<mg_handle*>(o.p[0]);
        const int rc = mg_cycle(h, i[0]);
        if (rc != MG_OK) return plan_fail(&p->err, rc, std::string("coarse cycle: ") + mg_last_error(h));
        break;
      }
      case MG_PLAN_COARSE_END: {
        mg_handle* h = static_cast<mg_handle*>(o.p[0]);
        const int rc = mg_get_solution_device(h, o.p[1], i[0], i[1]);
        if (rc != MG_OK) return plan_fail(&p->err, rc, std::string("coarse end: ") + mg_last_error(h));
        break;
      }
      case MG_PLAN_EVENT_RECORD:
        PLAN_HIP(p, hipEventRecord(p->ev[i[0]], s));
        break;
      case MG_PLAN_STREAM_WAIT:
        PLAN_HIP(p, hipStreamWaitEvent(s, p->ev[i[0]], 0));
        break;
      case MG_PLAN_RESULT:
        result_dev = static_cast<const double*>(o.p[0]);
        result_stream = s;
        break;
      default:
        return plan_fail(&p->err, MG_ERR_INVALID_VALUE, "unknown plan operation");
    }
    if (o.op == MG_PLAN_GROUP_BEGIN) ++group_depth;
    if (o.op == MG_PLAN_GROUP_END) --group_depth;
    if (span_op && group_depth == 0 && !p->spans.empty() && !p->spans.back().e1) {
      hipEvent_t e1 = nullptr;
      PLAN_HIP(p, hipEventCreate(&e1));
      PLAN_HIP(p, hipEventRecord(e1, s));
      p->spans.back().e1 = e1;
    }
    if (redirected && comm_op && group_depth == 0) {        // the call (or the whole group) is queued: hand back to the compute stream
      PLAN_HIP(p, hipEventRecord(p->fence[1], st[1]));
      PLAN_HIP(p, hipStreamWaitEvent(st[0], p->fence[1], 0));
      redirected = false;
    }
  }
  PLAN_HIP(p, hipGetLastError());
  if (result_dev) {
    PLAN_HIP(p, hipMemcpyAsync(p->result_host, result_dev, sizeof(double), hipMemcpyDeviceToHost, result_stream));
    PLAN_HIP(p, hipEventRecord(p->done, result_stream));
    p->pending = true;
  }
  return MG_OK;
}

int mg_plan_wait(mg_plan* p, double* result) {
  if (!p) return plan_fail(nullptr, MG_ERR_INVALID_VALUE, "mg_plan_wait: NULL plan");
  if (!p->pending) return plan_fail(&p->err, MG_ERR_STATE, "mg_plan_wait: no result in flight (the plan has no RESULT, or it was already collected)");
  const char* env_limit = std::getenv("MG_PLAN_TIMEOUT_S");
  const double env_s = env_limit ? std::atof(env_limit) : 0.0;
  if (p->comm || env_s > 0) {
    // a plan that talks to other ranks waits with a deadline: a peer that never posts its half of an exchange must end in
    // an error here, not in a process that hangs until somebody kills it (MG_PLAN_TIMEOUT_S, default 120 s; when the
    // variable is set the deadline also covers plans without a communicator).  The queued work stays queued: the caller
    // reports and leaves the process (MG_ERR_TIMEOUT), it does not synchronise on these streams again.
    const double limit = env_s > 0 ? env_s : 120.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; ++spins) {
      const hipError_t q = hipEventQuery(p->done);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) return plan_fail(&p->err, MG_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
      if ((spins & 63) == 63 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
        return plan_fail(&p->err, MG_ERR_TIMEOUT, "cycle plan: no result after MG_PLAN_TIMEOUT_S seconds (a peer rank is not taking part in an exchange?)");
    }
  } else {
    PLAN_HIP(p, hipEventSynchronize(p->done));
  }
  p->pending = false;
  if (result) *result = *p->result_host;
  return MG_OK;
}

int mg_plan_run(mg_plan* p, void* compute_stream, void* comm_stream, double* result) {
  const int rc = mg_plan_run_async(p, compute_stream, comm_stream);
  if (rc != MG_OK || !p->pending) return rc;
  return mg_plan_wait(p, result);
}

}  // extern "C"
