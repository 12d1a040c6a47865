// Native RCCL communicator of the row-sharded primal fit: one process per GPU, no PyTorch.
// librccl is opened with dlopen on first use - a single-GPU process never loads it (and a process that already
// carries another copy of the ROCm communication library, e.g. an imported torch, is left alone).
#include <dlfcn.h>

#include <mutex>

#include "nls_host.h"

const RcclApi* rccl_api(std::string* why) {
  static RcclApi api;
  static bool ok = false;
  static std::string err;
  static std::once_flag once;  // (the ranks of a group - several host threads of one process - may arrive here together)
  std::call_once(once, [] {
    void* h = nullptr;
    // NLS_RCCL_LIB: an explicit path (a non-standard install; the test stand-in of tests/csrc/rccl_shim.cpp) - tried alone when set
    const char* explicit_lib = std::getenv("NLS_RCCL_LIB");
    if (explicit_lib && explicit_lib[0]) {
      h = dlopen(explicit_lib, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
      }
    }
    if (!h) {
      err = std::string("cannot load librccl: ") + dlerror();
    } else {
      ok = true;
      auto sym = [&](const char* n) -> void* {
        void* p = dlsym(h, n);
        if (!p) {
          ok = false;
          err = std::string("librccl lacks ") + n;
        }
        return p;
      };
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
      api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
      api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
      api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
      api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
      api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
      api.CommGetAsyncError = reinterpret_cast<decltype(api.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
    }
  });
  if (!ok && why) *why = err;
  return ok ? &api : nullptr;
}

extern "C" int nls_comm_get_unique_id(void* id) {
  static_assert(NLS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return fail(nullptr, NLS_ERR_ARG, "id is NULL");
  std::string why;
  const RcclApi* api = rccl_api(&why);
  if (!api) return fail(nullptr, NLS_ERR_COMM, "%s", why.c_str());
  ncclUniqueId uid;
  ncclResult_t r = api->GetUniqueId(&uid);
  if (r != ncclSuccess) return fail(nullptr, NLS_ERR_COMM, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
  std::memcpy(id, &uid, NLS_COMM_ID_BYTES);
  return NLS_OK;
}

extern "C" int nls_comm_destroy(nls_ctx* ctx) {
  if (!ctx) return NLS_ERR_ARG;
  if (ctx->comm) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);  // (nothing collective is pending between calls: every collective is waited for, comm_wait)
    const RcclApi* api = rccl_api(nullptr);
    if (api) (void)api->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->comm_broken = false;
  if (!ctx->allreduce) {
    ctx->rank = 0;
    ctx->world = 1;
  }
  return NLS_OK;
}

extern "C" int nls_comm_abort(nls_ctx* ctx) {
  if (!ctx) return NLS_ERR_ARG;
  if (!ctx->comm) return NLS_OK;
  (void)hipSetDevice(ctx->device);
  (void)comm_give_up(ctx, "nls_comm_abort was called");
  return NLS_OK;
}

extern "C" int nls_comm_set_timeout(nls_ctx* ctx, double seconds) {
  if (!ctx) return NLS_ERR_ARG;
  if (!(seconds >= 0.0) || !std::isfinite(seconds)) return fail(ctx, NLS_ERR_ARG, "nls_comm_set_timeout: seconds must be finite and >= 0 (0: NLS_COMM_TIMEOUT_S / 300 s)");
  ctx->comm_timeout_s = seconds;
  return NLS_OK;
}

extern "C" int nls_comm_set_virtual_rank(nls_ctx* ctx, int rank, int world, int capture) {
  if (!ctx) return NLS_ERR_ARG;
  if (world > 1 && (rank < 0 || rank >= world || world > NLS_COMM_UTIL_MAX)) return fail(ctx, NLS_ERR_ARG, "nls_comm_set_virtual_rank: rank %d outside world %d", rank, world);
  if ((world > 1 || capture) && (!ctx->comm || ctx->world != 1))
    return fail(ctx, NLS_ERR_ARG, "nls_comm_set_virtual_rank: the context must be the only rank of a native communicator (nls_comm_init_rank with world 1)");
  if (world > 1 && capture) return fail(ctx, NLS_ERR_ARG, "nls_comm_set_virtual_rank: capture belongs to a complete call (world <= 1)");
  ctx->virt_rank = world > 1 ? rank : 0;
  ctx->virt_world = world > 1 ? world : 0;
  ctx->virt_capture = capture != 0;
  return NLS_OK;
}

extern "C" int nls_comm_state(const nls_ctx* ctx) { return !ctx ? -1 : (ctx->comm_broken ? 2 : (ctx->comm ? 1 : 0)); }

extern "C" int nls_comm_init_rank(nls_ctx* ctx, const void* id, int rank, int world) {
  if (!ctx) return NLS_ERR_ARG;
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(ctx, NLS_ERR_ARG, "bad id / rank / world (%d of %d)", rank, world);
  std::string why;
  const RcclApi* api = rccl_api(&why);
  if (!api) return fail(ctx, NLS_ERR_COMM, "%s", why.c_str());
  NLSCHK(nls_comm_destroy(ctx));
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ncclUniqueId uid;
  std::memcpy(&uid, id, NLS_COMM_ID_BYTES);
  if (!ctx->comm_scratch) HIPCHK(ctx, hipMalloc(&ctx->comm_scratch, NLS_COMM_UTIL_MAX * sizeof(double)));  // (before joining: a rank that cannot must not be a member)
  RCCLCHK(ctx, api, api->CommInitRank(&ctx->comm, world, uid, rank));
  ctx->rank = rank;
  ctx->world = world;
  ctx->comm_broken = false;
  ctx->allreduce = nullptr;  // the communicator replaces a previously registered hook
  return NLS_OK;
}

extern "C" int nls_comm_allreduce(nls_ctx* ctx, double* host_values, size_t count, int op) {
  if (!ctx) return NLS_ERR_ARG;
  if (!host_values || count == 0 || count > NLS_COMM_UTIL_MAX || op < 0 || op > 1)
    return fail(ctx, NLS_ERR_ARG, "nls_comm_allreduce: 1..%d host doubles, op 0|1", NLS_COMM_UTIL_MAX);
  if (ctx->comm_broken) return comm_refuse_broken(ctx);
  if (!ctx->comm) {
    if (ctx->world > 1) return fail(ctx, NLS_ERR_COMM, "nls_comm_allreduce needs a native communicator (nls_comm_init_rank)");
    return NLS_OK;  // single rank: the values are the result
  }
  const RcclApi* api = rccl_api(nullptr);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(ctx->comm_scratch, host_values, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (the caller's buffer is free again; nothing collective is pending here)
  RCCL_ENQUEUE(ctx, api, api->AllReduce(ctx->comm_scratch, ctx->comm_scratch, count, ncclDouble, op == 0 ? ncclSum : ncclMax, ctx->comm, ctx->stream));
  NLSCHK(comm_wait(ctx, "ncclAllReduce (nls_comm_allreduce)"));  // bounded: a pageable download behind an unfinished collective would block in the runtime
  HIPCHK(ctx, hipMemcpyAsync(host_values, ctx->comm_scratch, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}
