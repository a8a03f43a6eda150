// K16: the exchange step of the multi-GPU calibration -- SUM all-reduce of a reconstruction unit's flattened
// dL/dW_hat buffer and of the activation deltas (reference linklink/__init__.py:6-13, linklink/dist_helper.py:33-36,
// call sites quant/reconstruction.py:72-75,193-195,298-300 and quant/quant_model.py:127-132) -- on RCCL over xGMI,
// one rank per GPU, launched on the caller's stream so that it orders with the kernels around it (the all-reduce of
// a unit's gradients sits between its backward GEMMs and the fused AdaRound-backward + Adam kernel on the same stream;
// no host synchronisation, capturable).
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy torch already mapped when one is loaded, else the ROCm
// install this library was linked against), so single-GPU users and the CPU-only symbol test never touch it.
#include "common.hpp"
#include <dlfcn.h>
#include <cstring>

namespace {

// the slice of rccl.h this file needs (ABI-stable across NCCL 2.x)
typedef struct { char internal[128]; } rccl_unique_id;
typedef void* rccl_comm;
typedef int (*fn_get_unique_id)(rccl_unique_id*);
typedef int (*fn_comm_init_rank)(rccl_comm*, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(rccl_comm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*fn_comm_count)(const rccl_comm, int*);
typedef const char* (*fn_error_string)(int);
enum { RCCL_FLOAT32 = 7, RCCL_SUM = 0 };   // ncclFloat32, ncclSum

struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_error_string error_string = nullptr;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  if (r.lib || !r.err.empty()) return r;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) {
    r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) break;
  }
  if (!r.lib) {
    r.err = std::string("RCCL not found: ") + dlerror();
    return r;
  }
  r.get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(r.lib, "ncclGetUniqueId"));
  r.comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(r.lib, "ncclCommInitRank"));
  r.comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(r.lib, "ncclCommDestroy"));
  r.all_reduce = reinterpret_cast<fn_all_reduce>(dlsym(r.lib, "ncclAllReduce"));
  r.error_string = reinterpret_cast<fn_error_string>(dlsym(r.lib, "ncclGetErrorString"));
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) {
    r.err = "RCCL: a required symbol is missing from librccl";
    r.lib = nullptr;
  }
  return r;
}

int fail(tfmq_handle h, const char* what, int rc) {
  Rccl& r = rccl();
  if (h) h->err = std::string(what) + ": " + (r.error_string ? r.error_string(rc) : "rccl error") + " (" + std::to_string(rc) + ")";
  return TFMQ_ERR_HIP;
}

}  // namespace

extern "C" int tfmq_comm_unique_id(uint8_t* id_host) {
  if (!id_host) return TFMQ_ERR_ARG;
  Rccl& r = rccl();
  if (!r.lib) return TFMQ_ERR_UNSUPPORTED;
  rccl_unique_id id;
  if (r.get_unique_id(&id) != 0) return TFMQ_ERR_HIP;
  static_assert(sizeof(id) == TFMQ_COMM_ID_BYTES, "unique id size");
  std::memcpy(id_host, id.internal, sizeof(id));
  return TFMQ_OK;
}

extern "C" int tfmq_comm_init(tfmq_handle h, const uint8_t* id_host, int rank, int world) {
  TFMQ_CHECK_ARG(h, h && id_host && world >= 1 && rank >= 0 && rank < world, "comm_init: bad rank / world / id");
  TFMQ_CHECK_ARG(h, h->comm == nullptr, "comm_init: this handle already owns a communicator (tfmq_comm_destroy first)");
  Rccl& r = rccl();
  if (!r.lib) {
    h->err = r.err;
    return TFMQ_ERR_UNSUPPORTED;
  }
  TFMQ_HIP(h, hipSetDevice(h->device));       // one rank per GPU: the communicator binds to the handle's device
  rccl_unique_id id;
  std::memcpy(id.internal, id_host, sizeof(id));
  rccl_comm c = nullptr;
  const int rc = r.comm_init_rank(&c, world, id, rank);
  if (rc != 0) return fail(h, "ncclCommInitRank", rc);
  h->comm = c;
  h->comm_rank = rank;
  h->comm_world = world;
  return TFMQ_OK;
}

extern "C" int tfmq_comm_info(tfmq_handle h, int* rank, int* world) {
  TFMQ_CHECK_ARG(h, h, "comm_info: null handle");
  if (rank) *rank = h->comm ? h->comm_rank : 0;
  if (world) *world = h->comm ? h->comm_world : 0;      // 0 = no communicator
  return TFMQ_OK;
}

extern "C" int tfmq_allreduce_sum_f32(tfmq_handle h, float* buf, size_t n, void* stream) {
  TFMQ_CHECK_ARG(h, h && (buf || n == 0), "allreduce: null buffer");
  TFMQ_CHECK_ARG(h, h->comm, "allreduce: no communicator (tfmq_comm_init)");
  if (n == 0) return TFMQ_OK;
  const int rc = rccl().all_reduce(buf, buf, n, RCCL_FLOAT32, RCCL_SUM, h->comm, as_stream(stream));
  if (rc != 0) return fail(h, "ncclAllReduce", rc);
  return TFMQ_OK;
}

extern "C" int tfmq_comm_destroy(tfmq_handle h) {
  TFMQ_CHECK_ARG(h, h, "comm_destroy: null handle");
  if (h->comm) {
    const int rc = rccl().comm_destroy(h->comm);
    h->comm = nullptr;
    h->comm_world = 0;
    if (rc != 0) return fail(h, "ncclCommDestroy", rc);
  }
  return TFMQ_OK;
}
