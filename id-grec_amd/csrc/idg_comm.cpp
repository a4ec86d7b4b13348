// RCCL communicator behind the C ABI (include/idgrec.h, "multi-GPU"): collectives are enqueued on the CALLER's HIP
// stream, in order with the kernels around them — no second stream, no event pair, no host-side work object per call.
// The RCCL library is not a link-time dependency: idg_comm_load() opens the one the process already uses (PyTorch
// ships its own librccl.so; loading a second copy next to it would work but doubles the bootstrap), and every entry
// point fails with a message when it has not been loaded.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

#include "idg_common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclReduceScatter) reduce_scatter = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclGetVersion) get_version = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

template <typename F>
bool bind(void* h, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(h, name));
  return out != nullptr;
}

#define IDG_RCCL(call)                                                                                    \
  do {                                                                                                    \
    ncclResult_t r_ = (call);                                                                             \
    if (r_ != ncclSuccess)                                                                                \
      return idg::fail(IDG_E_HIP, "%s failed: %s (%s:%d)", #call,                                         \
                       g_rccl.error_string ? g_rccl.error_string(r_) : "?", __FILE__, __LINE__);          \
  } while (0)

}  // namespace

struct idg_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" {

int idg_comm_load(const char* librccl_path) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return IDG_OK;
  const char* path = (librccl_path && librccl_path[0]) ? librccl_path : "librccl.so";
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return idg::fail(IDG_E_UNSUPPORTED, "idg_comm_load: dlopen(%s): %s", path, dlerror());
  Rccl r;
  r.handle = h;
  const bool ok = bind(h, "ncclGetUniqueId", r.get_unique_id) && bind(h, "ncclCommInitRank", r.comm_init_rank) &&
                  bind(h, "ncclCommDestroy", r.comm_destroy) && bind(h, "ncclAllReduce", r.all_reduce) &&
                  bind(h, "ncclAllGather", r.all_gather) && bind(h, "ncclReduceScatter", r.reduce_scatter) &&
                  bind(h, "ncclSend", r.send) && bind(h, "ncclRecv", r.recv) && bind(h, "ncclGroupStart", r.group_start) &&
                  bind(h, "ncclGroupEnd", r.group_end) && bind(h, "ncclGetErrorString", r.error_string) &&
                  bind(h, "ncclGetVersion", r.get_version);
  if (!ok) {
    dlclose(h);
    return idg::fail(IDG_E_UNSUPPORTED, "idg_comm_load: %s does not export the NCCL API", path);
  }
  g_rccl = r;
  return IDG_OK;
}

int idg_comm_rccl_version(int* version) {
  IDG_REQUIRE(version, "idg_comm_rccl_version: NULL argument");
  if (!g_rccl.handle) return idg::fail(IDG_E_UNSUPPORTED, "idg_comm_rccl_version: call idg_comm_load first");
  IDG_RCCL(g_rccl.get_version(version));
  return IDG_OK;
}

int idg_comm_unique_id(void* out_id) {
  IDG_REQUIRE(out_id, "idg_comm_unique_id: NULL argument");
  if (!g_rccl.handle) return idg::fail(IDG_E_UNSUPPORTED, "idg_comm_unique_id: call idg_comm_load first");
  static_assert(sizeof(ncclUniqueId) == IDG_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  IDG_RCCL(g_rccl.get_unique_id(&id));
  std::memcpy(out_id, &id, sizeof id);
  return IDG_OK;
}

int idg_comm_create(int rank, int world, const void* unique_id, int device, idg_comm** out) {
  IDG_REQUIRE(unique_id && out, "idg_comm_create: NULL argument");
  IDG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "idg_comm_create: rank %d outside world %d", rank, world);
  if (!g_rccl.handle) return idg::fail(IDG_E_UNSUPPORTED, "idg_comm_create: call idg_comm_load first");
  *out = nullptr;
  IDG_HIP(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  idg_comm* c = new (std::nothrow) idg_comm;
  if (!c) return idg::fail(IDG_E_NOMEM, "idg_comm_create: out of host memory");
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclResult_t r = g_rccl.comm_init_rank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return idg::fail(IDG_E_HIP, "idg_comm_create: ncclCommInitRank failed: %s", g_rccl.error_string(r));
  }
  *out = c;
  return IDG_OK;
}

int idg_comm_destroy(idg_comm* c) {
  if (!c) return IDG_OK;
  if (c->comm && g_rccl.handle) (void)g_rccl.comm_destroy(c->comm);
  delete c;
  return IDG_OK;
}

int idg_allreduce_f32(idg_comm* c, float* buf, int64_t count, int average, void* stream) {
  IDG_REQUIRE(c && c->comm, "idg_allreduce_f32: NULL communicator");
  IDG_REQUIRE(buf && count >= 0, "idg_allreduce_f32: NULL buffer / negative count");
  if (count == 0) return IDG_OK;
  IDG_RCCL(g_rccl.all_reduce(buf, buf, (size_t)count, ncclFloat32, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream));
  return IDG_OK;
}

int idg_allgather_f32(idg_comm* c, const float* in, float* out, int64_t count, void* stream) {
  IDG_REQUIRE(c && c->comm, "idg_allgather_f32: NULL communicator");
  IDG_REQUIRE(in && out && count >= 0, "idg_allgather_f32: NULL buffer / negative count");
  if (count == 0) return IDG_OK;
  IDG_RCCL(g_rccl.all_gather(in, out, (size_t)count, ncclFloat32, c->comm, (hipStream_t)stream));
  return IDG_OK;
}

int idg_reduce_scatter_f32(idg_comm* c, const float* in, float* out, int64_t count, void* stream) {
  IDG_REQUIRE(c && c->comm, "idg_reduce_scatter_f32: NULL communicator");
  IDG_REQUIRE(in && out && count >= 0, "idg_reduce_scatter_f32: NULL buffer / negative count");
  if (count == 0) return IDG_OK;
  IDG_RCCL(g_rccl.reduce_scatter(in, out, (size_t)count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
  return IDG_OK;
}

// Every rank hands block p of `send` to rank p and receives rank p's block for it into block p of `recv`: one group of
// point-to-point transfers — on a fully connected xGMI node each of the 7 peers gets its block over its own link (the
// reduce-scatter half of a DIRECT all-reduce; idg_reduce24_f32 then adds the blocks in rank order).  The own block is a
// device copy — unless IDG_ALLTOALL_OWN_THROUGH_RCCL is set (tests: the grouped ncclSend / ncclRecv path on ONE device).
int idg_alltoall_f32(idg_comm* c, const float* send, float* recv, int64_t count, int flags, void* stream) {
  IDG_REQUIRE(c && c->comm, "idg_alltoall_f32: NULL communicator");
  IDG_REQUIRE(send && recv && count >= 0, "idg_alltoall_f32: NULL buffer / negative count");
  IDG_REQUIRE(send != recv, "idg_alltoall_f32: in place is not supported");
  if (count == 0) return IDG_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool own_rccl = (flags & IDG_ALLTOALL_OWN_THROUGH_RCCL) != 0;
  if (!own_rccl)
    IDG_HIP(hipMemcpyAsync(recv + (size_t)c->rank * (size_t)count, send + (size_t)c->rank * (size_t)count,
                           (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (c->world == 1 && !own_rccl) return IDG_OK;
  IDG_RCCL(g_rccl.group_start());
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank && !own_rccl) continue;
    ncclResult_t r = g_rccl.send(send + (size_t)p * (size_t)count, (size_t)count, ncclFloat32, p, c->comm, st);
    if (r == ncclSuccess) r = g_rccl.recv(recv + (size_t)p * (size_t)count, (size_t)count, ncclFloat32, p, c->comm, st);
    if (r != ncclSuccess) {
      (void)g_rccl.group_end();
      return idg::fail(IDG_E_HIP, "idg_alltoall_f32: ncclSend / ncclRecv with rank %d failed: %s", p, g_rccl.error_string(r));
    }
  }
  IDG_RCCL(g_rccl.group_end());
  return IDG_OK;
}

}  // extern "C"
