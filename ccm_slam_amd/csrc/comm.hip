// comm.hip — RCCL communicator attached to a ccm_ctx.  Used only by the landmark-sharded
// global BA (one all-reduce of the reduced camera system per LM trial).  The
// reference has no collective at all; this is new MI355X-side machinery.
#include "common.h"
#include <rccl/rccl.h>
#include <cstring>
#include <cstdlib>

// CCM_FORCE_ALLREDUCE=1 issues the collectives even on a 1-rank communicator (a legal, degenerate all-reduce):
// lets the single-GPU test box exercise the exact RCCL call sequence of the sharded global BA.
static bool force_collectives() { static int v = -1; if (v < 0) { const char* e = std::getenv("CCM_FORCE_ALLREDUCE"); v = (e && e[0] == '1') ? 1 : 0; } return v == 1; }

static_assert(sizeof(ncclUniqueId) == 128, "ccm_comm_unique_id assumes a 128-byte ncclUniqueId");

extern "C" int ccm_comm_unique_id(uint8_t id_bytes[128]) {
  if (!id_bytes) return CCM_E_ARG;
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return ccm_set_error(nullptr, CCM_E_COMM, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  std::memcpy(id_bytes, &id, 128);
  return CCM_OK;
}

extern "C" int ccm_comm_init(ccm_ctx* ctx, int nranks, int rank, const uint8_t id_bytes[128]) {
  if (!ctx || !id_bytes || nranks < 1 || rank < 0 || rank >= nranks) return ccm_set_error(ctx, CCM_E_ARG, "ccm_comm_init: bad args");
  if (ctx->comm) return ccm_set_error(ctx, CCM_E_STATE, "ccm_comm_init: communicator already attached");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, 128);
  ncclComm_t comm = nullptr;
  ncclResult_t r = ncclCommInitRank(&comm, nranks, id, rank);
  if (r != ncclSuccess) return ccm_set_error(ctx, CCM_E_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_nranks = nranks;
  return CCM_OK;
}

extern "C" int ccm_comm_destroy(ccm_ctx* ctx) {
  if (!ctx) return CCM_E_ARG;
  if (ctx->comm) { ncclCommDestroy((ncclComm_t)ctx->comm); ctx->comm = nullptr; }
  ctx->comm_rank = 0; ctx->comm_nranks = 1;
  return CCM_OK;
}

// in-place sum all-reduce of n doubles on the ctx stream
int ccm_allreduce_f64(ccm_ctx* ctx, double* d_buf, size_t n) {
  if (ctx->comm_nranks <= 1 && !(ctx->comm && force_collectives())) return CCM_OK;
  if (!ctx->comm) return ccm_set_error(ctx, CCM_E_STATE, "all-reduce requested but no communicator attached");
  ncclResult_t r = ncclAllReduce(d_buf, d_buf, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
  if (r != ncclSuccess) return ccm_set_error(ctx, CCM_E_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  return CCM_OK;
}

int ccm_allreduce_max_f64(ccm_ctx* ctx, double* d_buf, size_t n) {
  if (ctx->comm_nranks <= 1 && !(ctx->comm && force_collectives())) return CCM_OK;
  if (!ctx->comm) return ccm_set_error(ctx, CCM_E_STATE, "all-reduce requested but no communicator attached");
  ncclResult_t r = ncclAllReduce(d_buf, d_buf, n, ncclDouble, ncclMax, (ncclComm_t)ctx->comm, ctx->stream);
  if (r != ncclSuccess) return ccm_set_error(ctx, CCM_E_COMM, std::string("ncclAllReduce(max): ") + ncclGetErrorString(r));
  return CCM_OK;
}
