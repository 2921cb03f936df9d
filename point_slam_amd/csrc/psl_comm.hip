// Multi-GPU exchange inside the library: RCCL all-gather-v of per-rank record blocks (new neural
// points: position + both feature rows + add-radius; touched feature rows: row id + 64 changes) over xGMI.
//
// The reference has no distributed code; the construct is BASELINE.json's north_star (frame-parallel replicas with a
// periodic all-gather of newly added neural points).  librccl is NOT a link-time dependency: it is resolved with dlopen at
// the first psl_comm_* call (the copy torch has already loaded, when there is one), so a single-GPU process never
// touches it.  A communicator is either created here (psl_comm_unique_id on rank 0 -> hand the 128 bytes to every rank by
// any side channel -> psl_comm_init) or passed in by a host that owns one (ncclComm_t as void*).
#include <dlfcn.h>
#include <cstring>
#include <algorithm>
#include "psl_common.h"

namespace psl {

struct RcclId { char internal[128]; };           // ncclUniqueId (rccl.h:43), passed BY VALUE to ncclCommInitRank
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static Rccl g_rccl;
constexpr int kNcclInt32 = 2, kNcclFloat32 = 7;   // ncclDataType_t (rccl.h:459-466)
constexpr int kMaxCommWorld = 1024;

static int rccl_load() {
  if (g_rccl.lib) return PSL_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);     // the instance torch.distributed already uses
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) { set_error("psl_comm: librccl not found (%s)", dlerror()); return PSL_ERR_UNSUPPORTED; }
  Rccl r; r.lib = h;
  r.GetUniqueId = (int (*)(RcclId*))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
  r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
  r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
    set_error("psl_comm: librccl lacks an expected symbol"); return PSL_ERR_UNSUPPORTED;
  }
  g_rccl = r;
  return PSL_OK;
}
#define PSL_NCCL(call)                                                                         \
  do {                                                                                         \
    int e__ = (call);                                                                          \
    if (e__ != 0) {                                                                            \
      psl::set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(e__) : "?", __FILE__, __LINE__); \
      return PSL_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

__global__ void k_set_int2(int* d, int v0, int v1) { if (threadIdx.x == 0) { d[0] = v0; d[1] = v1; } }

// device ints of the counts phase: [world][2] gathered (rows, capacity) pairs + this rank's own pair; sized for the LARGEST
// world seen (a host-owned communicator passed to psl_allgather_new_points may be larger than the ctx's own)
static int comm_counts_reserve(psl_ctx* ctx, int world) {
  if (ctx->comm_counts && ctx->comm_counts_world >= world) return PSL_OK;
  if (ctx->comm_counts) (void)hipFree(ctx->comm_counts);
  ctx->comm_counts = nullptr; ctx->comm_counts_world = 0;
  PSL_HIP(hipMalloc(&ctx->comm_counts, sizeof(int) * 2 * (size_t)(world + 1)));
  ctx->comm_counts_world = world;
  return PSL_OK;
}

}  // namespace psl

using namespace psl;

extern "C" int psl_comm_unique_id(void* id_out) {
  if (!id_out) { set_error("psl_comm_unique_id: null"); return PSL_ERR_ARG; }
  int rc = rccl_load(); if (rc) return rc;
  RcclId id;
  PSL_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return PSL_OK;
}

// Everything psl_comm_init needs that can fail on ONE rank alone: librccl and the device ints of the counts phase.  A host
// calls it on every rank and lets the ranks agree on the outcome BEFORE anybody enters ncclCommInitRank (a rank that failed
// here would otherwise leave the others blocked in the communicator's rendezvous; advisor, round 4).
extern "C" int psl_comm_reserve(psl_ctx* ctx, int world) {
  if (!ctx || world < 1 || world > kMaxCommWorld) { set_error("psl_comm_reserve: bad argument"); return PSL_ERR_ARG; }
  int rc = rccl_load(); if (rc) return rc;
  PSL_HIP(hipSetDevice(ctx->device));
  return comm_counts_reserve(ctx, world);
}

extern "C" int psl_comm_init(psl_ctx* ctx, const void* id_in, int rank, int world) {
  if (!ctx || !id_in || world < 1 || rank < 0 || rank >= world) { set_error("psl_comm_init: bad argument"); return PSL_ERR_ARG; }
  if (ctx->comm) { set_error("psl_comm_init: this context already has a communicator"); return PSL_ERR_STATE; }
  int rc = psl_comm_reserve(ctx, world); if (rc) return rc;    // rank-local failures first, the collective last
  RcclId id; memcpy(&id, id_in, sizeof(id));
  void* comm = nullptr;
  PSL_NCCL(g_rccl.CommInitRank(&comm, world, id, rank));
  ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
  return PSL_OK;
}

extern "C" int psl_comm_destroy(psl_ctx* ctx) {
  if (!ctx) return PSL_ERR_ARG;
  if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
  ctx->comm = nullptr;
  if (ctx->comm_counts) (void)hipFree(ctx->comm_counts);
  if (ctx->comm_stage) (void)hipFree(ctx->comm_stage);
  ctx->comm_counts = nullptr; ctx->comm_counts_world = 0; ctx->comm_stage = nullptr; ctx->comm_stage_cap = 0;
  return PSL_OK;
}

// The decision of the counts phase, on the gathered (rows, capacity) pairs alone -- nothing rank-local enters it, so every
// rank of the communicator takes the same branch (tests/test_abi_cpu.py drives it with unequal capacities).
extern "C" int psl_allgather_decide(const int32_t* pairs, int world, int32_t* counts_out, long long* total_out, int* n_max_out) {
  if (!pairs || world < 1 || !counts_out || !total_out || !n_max_out) { set_error("psl_allgather_decide: bad argument"); return PSL_ERR_ARG; }
  long long total = 0; int n_max = 0, cap_min = pairs[1];
  for (int k = 0; k < world; ++k) {
    if (pairs[2 * k] < 0) { set_error("psl_allgather_decide: rank %d announced %d rows", k, pairs[2 * k]); return PSL_ERR_ARG; }
    counts_out[k] = pairs[2 * k];
    total += pairs[2 * k]; n_max = std::max(n_max, pairs[2 * k]); cap_min = std::min(cap_min, pairs[2 * k + 1]);
  }
  *total_out = total; *n_max_out = n_max;
  if (total > cap_min) {
    set_error("psl_allgather_new_points: %lld rows exceed the smallest receive capacity of the ranks (%d)", total, cap_min);
    return PSL_ERR_CAPACITY;
  }
  return PSL_OK;
}

extern "C" int psl_allgather_new_points(psl_ctx* ctx, void* nccl_comm, int world, const float* rec_local, int n_local,
                                        int rec_floats, float* rec_all, int capacity_rows, int32_t* counts_host, void* stream) {
  if (!ctx || n_local < 0 || rec_floats <= 0 || !counts_host || (n_local > 0 && !rec_local)) {
    set_error("psl_allgather_new_points: bad argument"); return PSL_ERR_ARG;
  }
  void* comm = nccl_comm ? nccl_comm : ctx->comm;
  if (!nccl_comm) world = ctx->comm_world;
  if (!comm || world < 1) { set_error("psl_allgather_new_points: no communicator (psl_comm_init, or pass an ncclComm_t)"); return PSL_ERR_STATE; }
  int rc = rccl_load(); if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (world > kMaxCommWorld) { set_error("psl_allgather_new_points: world %d > %d", world, kMaxCommWorld); return PSL_ERR_ARG; }
  rc = comm_counts_reserve(ctx, world); if (rc) return rc;
  // 1. counts: every rank contributes (rows, capacity of ITS receive buffer).  The capacity decision below is taken on the
  //    gathered pairs, i.e. on the same numbers on every rank: either every rank returns PSL_ERR_CAPACITY here, or every
  //    rank enters the records collective.  (Round 3 compared the total with the LOCAL capacity: with unequal buffers one
  //    rank retried the counts collective while the others had entered the records collective.)
  int* d_mine = ctx->comm_counts + 2 * world;
  hipLaunchKernelGGL(k_set_int2, dim3(1), dim3(64), 0, s, d_mine, n_local, capacity_rows);
  PSL_LAUNCH_CHECK();
  PSL_NCCL(g_rccl.AllGather(d_mine, ctx->comm_counts, 2, kNcclInt32, comm, s));
  int pairs[2 * kMaxCommWorld];
  PSL_HIP(hipMemcpyAsync(pairs, ctx->comm_counts, sizeof(int) * 2 * (size_t)world, hipMemcpyDeviceToHost, s));
  PSL_HIP(hipStreamSynchronize(s));
  long long total = 0; int n_max = 0;
  rc = psl_allgather_decide(pairs, world, counts_host, &total, &n_max);
  if (rc) return rc;
  if (n_max == 0) return 0;
  if (total > 0 && !rec_all) { set_error("psl_allgather_new_points: rec_all missing"); return PSL_ERR_ARG; }
  // 2. records: ncclAllGather wants equal send counts -> every rank sends n_max rows out of a staging buffer (the padding
  //    rows are never copied out); on point-to-point xGMI this exchange is latency-, not bandwidth-bound (<= 15 MB/rank)
  const size_t row = (size_t)rec_floats, stage_floats = (size_t)(world + 1) * n_max * row;
  if (ctx->comm_stage_cap < stage_floats) {
    if (ctx->comm_stage) (void)hipFree(ctx->comm_stage);
    ctx->comm_stage = nullptr; ctx->comm_stage_cap = 0;
    PSL_HIP(hipMalloc(&ctx->comm_stage, sizeof(float) * (stage_floats + stage_floats / 2)));
    ctx->comm_stage_cap = stage_floats + stage_floats / 2;
  }
  float* send = ctx->comm_stage;                       // [n_max][row]
  float* recv = ctx->comm_stage + (size_t)n_max * row; // [world][n_max][row]
  if (n_local > 0) PSL_HIP(hipMemcpyAsync(send, rec_local, sizeof(float) * n_local * row, hipMemcpyDeviceToDevice, s));
  PSL_NCCL(g_rccl.AllGather(send, recv, (size_t)n_max * row, kNcclFloat32, comm, s));
  size_t off = 0;
  for (int k = 0; k < world; ++k) {
    if (counts_host[k] > 0)
      PSL_HIP(hipMemcpyAsync(rec_all + off * row, recv + (size_t)k * n_max * row, sizeof(float) * counts_host[k] * row,
                             hipMemcpyDeviceToDevice, s));
    off += counts_host[k];
  }
  return (int)total;
}
