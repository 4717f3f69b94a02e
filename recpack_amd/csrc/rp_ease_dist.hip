// EASE fit with the n x n matrix sharded over the GPUs of one node: communicator + the distributed solve.
//
// Replaces the n^3 step of EASE._fit (recpack/algorithms/ease.py:81-82, np.linalg.inv of X^T X + l2 I) for matrices that one GPU
// should not or cannot hold; the reference has no multi-GPU path (recpack/algorithms/base.py:454-455, 686-690: a TODO).
//
// Layout.  n_pad = n rounded up to the panel width nb = 128 * panel_tiles; panel p (columns [p nb, (p + 1) nb)) lives on rank
// rp_cyc_owner(p) at local panel p / W (1-D block-cyclic with reflected odd rounds: rp_common.h).  A rank stores its nl panels side by side: S[n_pad][nl * nb] row-major.  The
// matrix is symmetric throughout, only blocks (i, j) with i >= j are kept current ("lower"); the diagonal blocks are kept in
// their lower 128-tiles.
//
// Solve: symmetric block Gauss-Jordan sweeps (Goodnight 1979), the symmetric form of the full sweeps round 2 drove from Python (removed in round 4).  For the
// pivot panel K with P = A_KK^-1 and the pivot's full column V = A[:, K] (V_i = A_iK for i > K from the owner's panel, V_i = A_Ki^T
// for i < K from row block K of everybody's panels), T = V P:
//     A_ij -= V_i T_j^T  (i, j != K),     A_iK = T_i (i > K),     A_Ki = T_i^T (i < K),     A_KK = -P,
// and after the last pivot the matrix is -(G + l2 I)^-1.  A rank updates the lower tiles of its own panels: n^2 nb / W flop per
// pivot, n^3 / W in all - the count of potrf + trtri + lauum - in ONE kind of product (K = nb) on the solve's MFMA kernel.
// What a pivot needs from the others is V: n nb floats (the row-block pieces all-gathered, the owner's part broadcast with P
// appended), n^2 * 4 bytes per rank over the whole solve; T_j = V_j P is formed locally for the own panels only.
//
// Schedule (per rank, four streams; the host never reads anything back before the end):
//     main    chain(K): pack pieces, [comm], unpack V, T_own, writes of row / column K, then the LOOK-AHEAD part of update K:
//             the diagonal block of panel K + 1 FIRST (-> pivot stream), row block K + 1 of the own panels, the rest of panel K + 1
//     pivot   the inverse of the next pivot block, started as soon as that block has its update
//     comm    all-gather of the pieces + broadcast of the owner's part of V (neither needs P), then the small broadcast of P
//     update  U(K): everything else of update K - it runs beside chain(K + 1)
// U(K) leaves row blocks and panels K, K + 1 alone, the look-ahead of K + 1 waits for U(K) (both add to the same tiles).
// V and T_own are double buffered.  A failing pivot (matrix not positive definite) leaves its index in a device flag that is
// read once, after the last pivot.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "rp_common.h"
#include "rp_comm.h"
#include "rp_gemm.h"

// ---- communicator ---------------------------------------------------------------------------------------------------------
struct rp_rccl_api {
  void* dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static rp_rccl_api g_rccl;

// librccl is opened at run time: the copy the process already holds (torch ships one) or ROCm's
static int rp_rccl_load() {
  if (g_rccl.dl) return RPHIP_OK;
  void* dl = nullptr;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)
    if ((dl = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
  if (!dl)
    for (const char* n : names)
      if ((dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!dl) RP_FAIL(RPHIP_ENODEV, "librccl could not be opened: %s", dlerror());
  rp_rccl_api a;
  a.dl = dl;
  bool ok = true;
  auto sym = [&](const char* name) {
    void* p = dlsym(dl, name);
    if (!p) ok = false;
    return p;
  };
  a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
  a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
  a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
  a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
  a.Send = (decltype(a.Send))sym("ncclSend");
  a.Recv = (decltype(a.Recv))sym("ncclRecv");
  a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
  a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  if (!ok) RP_FAIL(RPHIP_ENODEV, "librccl lacks an expected entry point");
  g_rccl = a;
  return RPHIP_OK;
}

#define RP_NCCL(expr)                                                                                            \
  do {                                                                                                           \
    ncclResult_t _r = (expr);                                                                                    \
    if (_r != ncclSuccess) RP_FAIL(RPHIP_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));               \
  } while (0)

struct rphip_comm {
  rphip_ctx* ctx = nullptr;
  int device = 0;  // (kept apart: the context may be destroyed before the communicator)
  int world = 1, rank = 0;
  bool is_rccl = false;
  bool is_loop = false;  // rphip_comm_loopback_create: one rank of a larger run measured alone
  bool loop_replicate = false;  // (rp_comm_loop_replicate)
  double loop_bw = 0.0, loop_lat = 0.0;  // bytes per second, seconds
  ncclComm_t nccl = nullptr;
  rphip_comm_ops ops{};
  int64_t bytes_recv = 0, bytes_sent = 0;
};

extern "C" int rphip_comm_rccl_unique_id(void* id_out) {
  if (!id_out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_rccl_load());
  static_assert(sizeof(ncclUniqueId) == RPHIP_COMM_ID_BYTES, "id size");
  ncclUniqueId id;
  RP_NCCL(g_rccl.GetUniqueId(&id));
  ::memcpy(id_out, &id, sizeof(id));
  return RPHIP_OK;
}

extern "C" int rphip_comm_rccl_create(rphip_ctx* ctx, int32_t world, int32_t rank, const void* id, rphip_comm** out) {
  RP_ENTER(ctx);
  if (!ctx || !id || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) RP_FAIL(RPHIP_EINVAL, "rank %d of %d", rank, world);
  RP_TRY(rp_rccl_load());
  RP_TRY(rp_set_device(ctx));
  ncclUniqueId uid;
  ::memcpy(&uid, id, sizeof(uid));
  rphip_comm* c = new rphip_comm();
  c->ctx = ctx;
  c->device = ctx->device;
  c->world = world;
  c->rank = rank;
  c->is_rccl = true;
  ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    RP_FAIL(RPHIP_EHIP, "ncclCommInitRank (rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
  }
  *out = c;
  return RPHIP_OK;
}

extern "C" int rphip_comm_from_ops(rphip_ctx* ctx, const rphip_comm_ops* ops, rphip_comm** out) {
  RP_ENTER(ctx);
  if (!ctx || !ops || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (ops->world < 1 || ops->rank < 0 || ops->rank >= ops->world) RP_FAIL(RPHIP_EINVAL, "rank %d of %d", ops->rank, ops->world);
  if (ops->world > 1 && (!ops->broadcast || !ops->all_gather || !ops->exchange || !ops->all_reduce_sum_i32))
    RP_FAIL(RPHIP_EINVAL, "a transport for more than one rank needs all four operations");
  rphip_comm* c = new rphip_comm();
  c->ctx = ctx;
  c->device = ctx->device;
  c->world = ops->world;
  c->rank = ops->rank;
  c->ops = *ops;
  *out = c;
  return RPHIP_OK;
}

// ---- loop-back transport: the time of a transfer without the transfer -----------------------------------------------------------
__global__ void k_comm_delay(long long ticks) {  // holds its stream: s_memrealtime counts at 100 MHz
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_comm_noise(float* __restrict__ p, long long n, unsigned seed) {  // small values of both signs (never all zero: the clock a
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {  // chip holds depends on the data)
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = ((float)(h & 0xffffu) - 32768.0f) * (1.0f / 33554432.0f);
  }
}
// `landed` = bytes the emulation has just written in place of the peers' incoming data: on a real node those writes happen WHILE the
// link is busy, so the time they took here (priced at 2.5 TB/s, a memset beside other traffic) is taken off the link time
static void rp_loop_delay(rphip_comm* c, int64_t bytes, hipStream_t st, int64_t landed = 0) {
  double sec = c->loop_lat + (c->loop_bw > 0.0 ? (double)bytes / c->loop_bw : 0.0);
  if (landed > 0) sec = std::max(c->loop_lat, sec - (double)landed / 2.5e12);
  hipLaunchKernelGGL(k_comm_delay, dim3(1), dim3(1), 0, st, (long long)(sec * 1e8));
}
static void rp_loop_noise(void* p, int64_t bytes, hipStream_t st, unsigned seed) {
  const long long n = bytes / 4;
  if (n > 0) hipLaunchKernelGGL(k_comm_noise, dim3((unsigned)std::min<long long>(4096, rp_cdiv(n, 256))), dim3(256), 0, st, (float*)p, n, seed);
}

extern "C" int rphip_comm_loopback_create(rphip_ctx* ctx, int32_t world, int32_t rank, double gbytes_per_s, double latency_us, rphip_comm** out) {
  RP_ENTER(ctx);
  if (!ctx || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) RP_FAIL(RPHIP_EINVAL, "rank %d of %d", rank, world);
  if (!(gbytes_per_s > 0.0) || latency_us < 0.0) RP_FAIL(RPHIP_EINVAL, "the modelled link needs a positive rate and a non-negative latency");
  rphip_comm* c = new rphip_comm();
  c->ctx = ctx;
  c->device = ctx->device;
  c->world = world;
  c->rank = rank;
  c->is_loop = true;
  c->loop_bw = gbytes_per_s * 1e9;
  c->loop_lat = latency_us * 1e-6;
  *out = c;
  return RPHIP_OK;
}

extern "C" int rphip_comm_destroy(rphip_comm* c) {
  if (!c) return RPHIP_OK;
  if (c->is_rccl && c->nccl) {
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    g_rccl.CommDestroy(c->nccl);
  }
  delete c;
  return RPHIP_OK;
}

extern "C" int rphip_comm_info(const rphip_comm* c, int32_t* world, int32_t* rank, int32_t* is_rccl) {
  if (!c) RP_FAIL(RPHIP_EINVAL, "communicator is NULL");
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (is_rccl) *is_rccl = c->is_rccl ? 1 : 0;
  return RPHIP_OK;
}

#define RP_OPS(expr, what)                                                                         \
  do {                                                                                             \
    if ((expr) != 0) RP_FAIL(RPHIP_EHIP, "the caller's transport failed in %s", what);             \
  } while (0)

int rp_comm_world(const rphip_comm* c) { return c->world; }
bool rp_comm_is_loopback(const rphip_comm* c) { return c->is_loop; }
void rp_comm_loop_replicate(rphip_comm* c, bool on) { c->loop_replicate = on; }
int rp_comm_rank(const rphip_comm* c) { return c->rank; }

int rp_comm_broadcast(rphip_comm* c, void* buf, int64_t bytes, int root, hipStream_t st) {
  if (c->world == 1 || bytes <= 0) return RPHIP_OK;
  if (c->rank == root)
    c->bytes_sent += bytes;
  else
    c->bytes_recv += bytes;
  // (round 5) A LARGE message travels as scatter + all-gather: the root sends piece p (1 / W of the message) to rank p, then every
  // rank sends its piece to the others - on a fully connected node (xGMI: every pair of GPUs has its own link) no link carries more
  // than 2 / W of the message, where a root that feeds every peer the whole message - or a ring - puts all of it on every link:
  // the pivot column of the distributed solve (0.4 GB on average at 100 000 items) was 4 of the ~8 ms of the per-pivot recurrence
  // in the 8-rank emulation.  Below 8 MiB (and for two ranks) the transport's own broadcast.
  const int W = c->world, me = c->rank;
  const bool split = W > 2 && bytes >= ((int64_t)8 << 20) && (!c->ctx || c->ctx->comm_broadcast_split);
  const int64_t chunk = split ? ((bytes + W - 1) / W + 255) / 256 * 256 : 0;
  auto piece_len = [&](int p) { return std::max<int64_t>(0, std::min<int64_t>(chunk, bytes - (int64_t)p * chunk)); };
  if (c->is_rccl) {
    if (!split) {
      RP_NCCL(g_rccl.Broadcast(buf, buf, (size_t)bytes, ncclInt8, root, c->nccl, st));
      return RPHIP_OK;
    }
    char* b = (char*)buf;
    RP_NCCL(g_rccl.GroupStart());  // scatter
    if (me == root) {
      for (int p = 0; p < W; ++p)
        if (p != me && piece_len(p) > 0) RP_NCCL(g_rccl.Send(b + (int64_t)p * chunk, (size_t)piece_len(p), ncclInt8, p, c->nccl, st));
    } else if (piece_len(me) > 0) {
      RP_NCCL(g_rccl.Recv(b + (int64_t)me * chunk, (size_t)piece_len(me), ncclInt8, root, c->nccl, st));
    }
    RP_NCCL(g_rccl.GroupEnd());
    RP_NCCL(g_rccl.GroupStart());  // every piece to everybody who lacks it (the root lacks nothing)
    for (int q = 0; q < W; ++q) {
      if (q == me) continue;
      if (q != root && piece_len(me) > 0) RP_NCCL(g_rccl.Send(b + (int64_t)me * chunk, (size_t)piece_len(me), ncclInt8, q, c->nccl, st));
      if (me != root && piece_len(q) > 0) RP_NCCL(g_rccl.Recv(b + (int64_t)q * chunk, (size_t)piece_len(q), ncclInt8, q, c->nccl, st));
    }
    RP_NCCL(g_rccl.GroupEnd());
    return RPHIP_OK;
  }
  if (c->is_loop) {
    if (c->rank != root) rp_loop_noise(buf, bytes, st, (unsigned)c->bytes_recv);
    if (split) {
      rp_loop_delay(c, chunk, st);  // scatter: one piece per link
      rp_loop_delay(c, chunk, st);  // all-gather: one piece per link and direction
    } else {
      rp_loop_delay(c, bytes, st);
    }
    RP_HIP(hipGetLastError());
    return RPHIP_OK;
  }
  RP_OPS(c->ops.broadcast(c->ops.self, buf, bytes, root, (void*)st), "broadcast");
  return RPHIP_OK;
}

int rp_comm_all_gather(rphip_comm* c, const void* send, void* recv, int64_t bytes_per_rank, hipStream_t st) {
  if (bytes_per_rank <= 0) return RPHIP_OK;
  if (c->world == 1) {
    if (send != recv) RP_HIP(hipMemcpyAsync(recv, send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice, st));
    return RPHIP_OK;
  }
  c->bytes_sent += bytes_per_rank;
  c->bytes_recv += bytes_per_rank * (c->world - 1);
  if (c->is_rccl) {
    RP_NCCL(g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, c->nccl, st));
    return RPHIP_OK;
  }
  if (c->is_loop) {  // every slot takes this rank's own part: data of the right kind and size
    // (parts of 32 MiB and more - the strips of a panel gather, whose contents nobody computes with: the other ranks' slots are WRITTEN
    // only, as the peers' incoming xGMI writes would - eight device copies per strip read 40 GB more than a real gather does, beside
    // a predict pass that is bound by HBM bandwidth)
    const bool big = bytes_per_rank >= ((int64_t)32 << 20) && !c->loop_replicate;
    for (int p = 0; p < c->world; ++p) {
      if (big && p != c->rank)
        RP_HIP(hipMemsetAsync((char*)recv + (size_t)p * bytes_per_rank, 0, (size_t)bytes_per_rank, st));
      else
        RP_HIP(hipMemcpyAsync((char*)recv + (size_t)p * bytes_per_rank, send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice, st));
    }
    rp_loop_delay(c, bytes_per_rank, st, big ? bytes_per_rank * (c->world - 1) : 0);  // (W - 1 parts over W - 1 links at once)
    RP_HIP(hipGetLastError());
    return RPHIP_OK;
  }
  RP_OPS(c->ops.all_gather(c->ops.self, send, recv, bytes_per_rank, (void*)st), "all_gather");
  return RPHIP_OK;
}

int rp_comm_exchange(rphip_comm* c, const void* send, const int64_t* soff, const int64_t* sbytes, void* recv, const int64_t* roff,
                            const int64_t* rbytes, hipStream_t st) {
  const int me = c->rank;
  for (int p = 0; p < c->world; ++p)
    if (p != me) {
      c->bytes_sent += sbytes[p];
      c->bytes_recv += rbytes[p];
    }
  if (c->world == 1 || c->is_rccl || c->is_loop) {
    if (sbytes[me] > 0)
      RP_HIP(hipMemcpyAsync((char*)recv + roff[me], (const char*)send + soff[me], (size_t)sbytes[me], hipMemcpyDeviceToDevice, st));
    if (c->world == 1) return RPHIP_OK;
    if (c->is_loop) {
      int64_t in = 0, outb = 0;
      for (int p = 0; p < c->world; ++p) {
        if (p == me) continue;
        in = std::max(in, rbytes[p]);  // (every peer over its own link)
        outb = std::max(outb, sbytes[p]);
        // what comes from rank p: this rank's own piece FOR rank p when it is large enough (data of the right kind - lists of
        // (column, score), blocks of the matrix), noise otherwise
        if (rbytes[p] > 0) {
          if (sbytes[p] >= rbytes[p])
            RP_HIP(hipMemcpyAsync((char*)recv + roff[p], (const char*)send + soff[p], (size_t)rbytes[p], hipMemcpyDeviceToDevice, st));
          else
            rp_loop_noise((char*)recv + roff[p], rbytes[p] & ~(int64_t)3, st, (unsigned)(p * 977 + 13));
        }
      }
      rp_loop_delay(c, std::max(in, outb), st);
      RP_HIP(hipGetLastError());
      return RPHIP_OK;
    }
    RP_NCCL(g_rccl.GroupStart());
    for (int p = 0; p < c->world; ++p) {
      if (p == me) continue;
      if (sbytes[p] > 0) RP_NCCL(g_rccl.Send((const char*)send + soff[p], (size_t)sbytes[p], ncclInt8, p, c->nccl, st));
      if (rbytes[p] > 0) RP_NCCL(g_rccl.Recv((char*)recv + roff[p], (size_t)rbytes[p], ncclInt8, p, c->nccl, st));
    }
    RP_NCCL(g_rccl.GroupEnd());
    return RPHIP_OK;
  }
  RP_OPS(c->ops.exchange(c->ops.self, send, soff, sbytes, recv, roff, rbytes, (void*)st), "exchange");
  return RPHIP_OK;
}

int rp_comm_all_reduce_sum_i32(rphip_comm* c, int32_t* buf, int64_t count, hipStream_t st) {
  if (count <= 0 || c->world == 1) return RPHIP_OK;
  if (c->is_rccl) {
    RP_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclInt32, ncclSum, c->nccl, st));
    return RPHIP_OK;
  }
  if (c->is_loop) {  // (the buffer keeps this rank's own part)
    rp_loop_delay(c, 2 * count * 4, st);
    RP_HIP(hipGetLastError());
    return RPHIP_OK;
  }
  RP_OPS(c->ops.all_reduce_sum_i32(c->ops.self, buf, count, (void*)st), "all_reduce_sum_i32");
  return RPHIP_OK;
}

// ---- self-test of a communicator: one operation per call, a known pattern in, every word checked on the device ------------------
__device__ __forceinline__ int32_t rp_st_pat(int who, long long i) { return (int32_t)((unsigned)(who + 1) * 1000003u + (unsigned)i * 7919u); }
__global__ void k_st_fill(int32_t* __restrict__ p, long long n, int who, int per_piece, int piece_words) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = per_piece ? rp_st_pat(who * 1024 + (int)(i / piece_words), i % piece_words) : rp_st_pat(who, i);
}
__global__ void k_st_fill_reduce(int32_t* __restrict__ p, long long n, int me) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = (me + 1) * (int)(i % 7 + 1);
}
// mode 0: p[i] == pat(who, i); 1: piece s of p == pat(s, i) (all-gather); 2: piece s == pat(s * 1024 + me, i) (exchange); 3: the sum
__global__ void k_st_check(const int32_t* __restrict__ p, long long n, int mode, int who, int world, int piece_words, unsigned long long* __restrict__ bad) {
  unsigned long long b = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    int32_t want;
    if (mode == 0)
      want = rp_st_pat(who, i);
    else if (mode == 1)
      want = rp_st_pat((int)(i / piece_words), i % piece_words);
    else if (mode == 2)
      want = rp_st_pat((int)(i / piece_words) * 1024 + who, i % piece_words);
    else
      want = (world * (world + 1) / 2) * (int)(i % 7 + 1);
    b += p[i] != want ? 1 : 0;
  }
  if (b) atomicAdd(bad, b);
}

// op: 0 broadcast (from the last rank), 1 all-gather, 2 personalised exchange, 3 int32 all-reduce, 4 the broadcast of a large message
// (scatter + all-gather); `words` int32 per rank (or piece).
// Runs the operation once on the context's stream with a known pattern, waits, and checks every received word on the device:
// *mismatches = words that differ (0 = the operation works), *ms = its wall time.  A transport that hangs hangs HERE, in a call that
// names the operation - the caller puts a watchdog around it (bench.py: 30 s) instead of finding out in the first pivot of a solve.
extern "C" int rphip_comm_selftest(rphip_ctx* ctx, rphip_comm* c, int32_t op, int64_t words, int64_t* mismatches, double* ms) {
  RP_ENTER(ctx);
  if (!ctx || !c || !mismatches) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (op < 0 || op > 4 || words < 1 || words > (1 << 26)) RP_FAIL(RPHIP_EINVAL, "self-test: operation %d, %lld words", op, (long long)words);
  const bool big_bcast = op == 4;  // the broadcast's scatter + all-gather form (messages of 8 MiB and more): at least 3M words
  if (big_bcast) {
    op = 0;
    words = std::max<int64_t>(words, (int64_t)3 << 20);
  }
  RP_TRY(rp_set_device(ctx));
  const int W = c->world, me = c->rank;
  hipStream_t st = ctx->stream;
  const long long n_send = (op == 2) ? words * W : words, n_recv = (op == 1 || op == 2) ? words * W : words;
  int32_t *send = nullptr, *recv = nullptr;
  unsigned long long* bad = nullptr;
  RP_TRY(rp_dev_alloc(ctx, (size_t)n_send * 4, (void**)&send));
  int rc = rp_dev_alloc(ctx, (size_t)n_recv * 4 + 256, (void**)&recv);
  if (rc != RPHIP_OK) {
    rp_dev_free(ctx, send);
    return rc;
  }
  bad = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(recv) + (((size_t)n_recv * 4 + 15) & ~(size_t)15));
  hipMemsetAsync(bad, 0, 8, st);
  hipMemsetAsync(recv, 0xA5, (size_t)n_recv * 4, st);
  const int fg = 1024;
  if (op == 3)
    hipLaunchKernelGGL(k_st_fill_reduce, dim3(fg), dim3(256), 0, st, recv, n_recv, me);
  else if (op == 0) {
    if (me == W - 1) hipLaunchKernelGGL(k_st_fill, dim3(fg), dim3(256), 0, st, recv, n_recv, me, 0, 1);
  } else
    hipLaunchKernelGGL(k_st_fill, dim3(fg), dim3(256), 0, st, send, n_send, me, op == 2 ? 1 : 0, (int)words);
  hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  if (op == 0)
    rc = rp_comm_broadcast(c, recv, words * 4, W - 1, st);
  else if (op == 1)
    rc = rp_comm_all_gather(c, send, recv, words * 4, st);
  else if (op == 2) {
    std::vector<int64_t> off(W), by(W);
    for (int p = 0; p < W; ++p) {
      off[p] = (int64_t)p * words * 4;
      by[p] = words * 4;
    }
    rc = rp_comm_exchange(c, send, off.data(), by.data(), recv, off.data(), by.data(), st);
  } else
    rc = rp_comm_all_reduce_sum_i32(c, recv, words, st);
  hipError_t e = hipStreamSynchronize(st);
  if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  unsigned long long h_bad = ~0ull;
  if (rc == RPHIP_OK && e == hipSuccess) {
    hipLaunchKernelGGL(k_st_check, dim3(fg), dim3(256), 0, st, recv, n_recv, op, op == 0 ? W - 1 : me, W, (int)words, bad);
    hipMemcpyAsync(&h_bad, bad, 8, hipMemcpyDeviceToHost, st);
    e = hipStreamSynchronize(st);
  }
  rp_dev_free(ctx, send);
  rp_dev_free(ctx, recv);
  if (rc != RPHIP_OK) return rc;
  if (e != hipSuccess) RP_FAIL(RPHIP_EHIP, "self-test of operation %d: %s", op, hipGetErrorString(e));
  // (a loop-back communicator moves nothing: what it "receives" is by contract not the other ranks' data)
  *mismatches = c->is_loop ? 0 : (int64_t)h_bad;
  return RPHIP_OK;
}

extern "C" int rphip_comm_all_reduce_sum_i32(rphip_ctx* ctx, rphip_comm* c, int32_t* buf, int64_t count) {
  RP_ENTER(ctx);
  if (!ctx || !c || !buf) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (count <= 0 || c->world == 1) return RPHIP_OK;
  RP_TRY(rp_set_device(ctx));
  return rp_comm_all_reduce_sum_i32(c, buf, count, ctx->stream);
}

// ---- small kernels ----------------------------------------------------------------------------------------------------------
// batch of nb x nb blocks: dst_b[c][r] = src_b[r][c]; block b = blockIdx.z at src + b * sbs, dst + b * dbs
__global__ void k_blk_transpose(const float* __restrict__ src, long long lds, long long sbs, float* __restrict__ dst, long long ldd,
                                long long dbs) {
  __shared__ float t[32][33];
  src += (long long)blockIdx.z * sbs;
  dst += (long long)blockIdx.z * dbs;
  const long long bx = (long long)blockIdx.x * 32, by = (long long)blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) t[r][threadIdx.x] = src[(by + r) * lds + bx + threadIdx.x];
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) dst[(bx + r) * ldd + by + threadIdx.x] = t[threadIdx.x][r];
}

// dst[r][c] = scale * src[r][c], rows x cols (cols a multiple of 4), 16-byte accesses
__global__ void k_copy2d(const float* __restrict__ src, long long lds, float* __restrict__ dst, long long ldd, long long rows, int cols,
                         float scale) {
  const int c4 = cols >> 2;
  const long long total = rows * c4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / c4;
    const int c = (int)(e % c4) * 4;
    float4 v = *reinterpret_cast<const float4*>(src + r * lds + c);
    v.x *= scale;
    v.y *= scale;
    v.z *= scale;
    v.w *= scale;
    *reinterpret_cast<float4*>(dst + r * ldd + c) = v;
  }
}

// pieces gathered from the ranks: slot (r, q) at src + (r * slots + q) * nb * nb is the nb x nb block of rank r's local panel q;
// blocks of panels < limit go to dst rows [panel * nb, +nb) (leading dimension ldd)
__global__ void k_unpack_pieces(const float* __restrict__ src, int slots, int W, int limit, int nb, float* __restrict__ dst, long long ldd) {
  const int z = blockIdx.y, r = z / slots, q = z % slots;
  const int panel = rp_cyc_panel(q, r, W);
  if (panel >= limit) return;
  const float* s = src + (long long)z * nb * nb;
  float* d = dst + (long long)panel * nb * ldd;
  const int c4 = nb >> 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nb * c4; e += gridDim.x * blockDim.x) {
    const int row = e / c4, c = (e % c4) * 4;
    *reinterpret_cast<float4*>(d + (long long)row * ldd + c) = *reinterpret_cast<const float4*>(s + (long long)row * nb + c);
  }
}

// upper 128-tiles of an nb x nb diagonal block from its lower ones (the block sits at P with leading dimension ld)
__global__ void __launch_bounds__(256) k_mirror_block(float* __restrict__ P, long long ld) {
  __shared__ float t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) t[r][tx] = P[(long long)(bi * 32 + r) * ld + bj * 32 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bj * 32 + r, j = bi * 32 + tx;
    if (j > i) P[(long long)i * ld + j] = t[tx][r];
  }
}

struct rp_cyc {
  int nb, W, r;
  __host__ __device__ int to_global(int l) const { return rp_cyc_panel(l / nb, r, W) * nb + l % nb; }
};

// K7 on block-cyclic panels holding sign * P: colscale_l = -w_j / S_jl (the sign cancels in P_ij / P_jj), 0 for padding columns
__global__ void k_dist_colscale(const float* __restrict__ S, long long ld, int n, int n_local, rp_cyc cy, const int32_t* __restrict__ deg,
                                float alpha, float* __restrict__ colscale) {
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < n_local; l += gridDim.x * blockDim.x) {
    const int j = cy.to_global(l);
    float sc = 0.0f;
    if (j < n) {
      const float d = S[(long long)j * ld + l];
      float w = 1.0f;
      if (alpha != 0.0f) w = deg[j] > 0 ? powf((float)deg[j], -alpha) : 0.0f;
      sc = -w / d;
    }
    colscale[l] = sc;
  }
}

__global__ void k_dist_scale(float* __restrict__ S, long long ld, long long n_rows, int n, rp_cyc cy, const float* __restrict__ colscale) {
  const long long total = n_rows * ld;
  for (long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < total; e += (long long)gridDim.x * blockDim.x * 4) {
    const long long i = e / ld;
    const int l = (int)(e % ld);
    float4 v = *reinterpret_cast<float4*>(S + e);
    float o[4] = {v.x, v.y, v.z, v.w};
    const int j0 = cy.to_global(l);  // (four consecutive local columns lie in one panel)
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (i < n && (long long)(j0 + c) != i) ? o[c] * colscale[l + c] : 0.0f;
    *reinterpret_cast<float4*>(S + e) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---- the distributed fit ------------------------------------------------------------------------------------------------------
namespace {

struct dist_res {  // everything the fit holds besides the slab; released on every way out
  rphip_ctx* ctx = nullptr;
  hipStream_t su = nullptr, sc = nullptr, sp = nullptr;
  std::vector<hipEvent_t> evs;
  std::vector<void*> bufs;
  int64_t held = 0, peak = 0;
  int alloc(size_t bytes, void** out) {
    RP_TRY(rp_dev_alloc(ctx, bytes, out));
    bufs.push_back(*out);
    held += (int64_t)bytes;
    peak = std::max(peak, held);
    return RPHIP_OK;
  }
  int event(hipEvent_t* e) {
    RP_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    evs.push_back(*e);
    return RPHIP_OK;
  }
  ~dist_res() {
    if (sp) hipStreamSynchronize(sp);
    if (su) hipStreamSynchronize(su);
    if (sc) hipStreamSynchronize(sc);
    if (ctx) hipStreamSynchronize(ctx->stream);
    for (auto e : evs) hipEventDestroy(e);
    for (auto p : bufs) rp_dev_free(ctx, p);
    if (su) hipStreamDestroy(su);
    if (sc) hipStreamDestroy(sc);
    if (sp) hipStreamDestroy(sp);
  }
};

}  // namespace

extern "C" int rphip_ease_fit_dist_stats(rphip_ctx* ctx, int64_t* out8) {
  RP_ENTER(ctx);
  if (!ctx || !out8) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  ::memcpy(out8, ctx->dist_stats, sizeof(ctx->dist_stats));
  return RPHIP_OK;
}

extern "C" int rphip_ease_fit_dist(rphip_ctx* ctx, rphip_comm* comm, const rphip_csr* X, double l2, double alpha, int32_t panel_tiles,
                                   rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !comm || !X || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (!(l2 > 0.0)) RP_FAIL(RPHIP_EINVAL, "l2 must be > 0 (got %g): G + l2*I has to be positive definite", l2);
  if (X->xt_row_end >= 0) RP_FAIL(RPHIP_ESTATE, "EASE needs a matrix with all item rows of X^T (not rphip_csr_from_device_rows)");
  RP_NEED_DEGREES(X);
  RP_TRY(rp_set_device(ctx));
  const int n = (int)X->n_items;
  if (n <= 0) RP_FAIL(RPHIP_EINVAL, "no items");
  const int W = comm->world, me = comm->rank;
  const int n128 = (int)(rp_cdiv(n, GT) * GT);
  if (panel_tiles <= 0) {
    // 16 tiles (2048 columns: the product kernel's plane path wants K >= 1024 and the pivot chain has to fit beside one update);
    // fewer when that would leave a rank with fewer than four panels
    panel_tiles = 16;
    while (panel_tiles > 1 && n128 / (GT * panel_tiles) < 4 * W) panel_tiles >>= 1;
  }
  const int nb = GT * panel_tiles;
  const int np = (int)rp_cdiv(n, nb);
  const long long n_pad_ll = (long long)np * nb;
  if (n_pad_ll > 0x7fffffffLL - nb) RP_FAIL(RPHIP_ELIMIT, "too many items");
  const int n_pad = (int)n_pad_ll;
  const int nl = rp_cyc_count(np, me, W);  // own panels: one per round (rp_cyc_panel)
  const long long ld = (long long)std::max(nl, 1) * nb;
  auto cnt_lt = [&](int K, int r) { return rp_cyc_count(K, r, W); };  // panels of rank r below K
  const int slots = (int)rp_cdiv(np, W);                                        // pieces a rank contributes at most
  const float lam = (float)l2;  // rounded to float32 as the reference does (ease.py:82)
  hipStream_t st = ctx->stream;

  dist_res R;  // (declared before the model: on a failure the streams are drained before the slab goes)
  R.ctx = ctx;
  rphip_model* m = new rphip_model();
  m->ctx = ctx;
  m->kind = RPHIP_MODEL_DENSE;
  m->n_items = n;
  m->row_end = n;
  m->ld = ld;
  m->empty_rows = -1;
  m->cyc_nb = nb;
  m->cyc_world = W;
  m->cyc_rank = me;
  auto fail = [&](int rc) {
    if (R.sp) hipStreamSynchronize(R.sp);
    if (R.su) hipStreamSynchronize(R.su);
    if (R.sc) hipStreamSynchronize(R.sc);
    hipStreamSynchronize(st);
    rphip_model_destroy(m);
    return rc;
  };
  const size_t slab_bytes = (size_t)n_pad * ld * sizeof(float);
  {
    int rc = rp_dev_alloc(ctx, slab_bytes, (void**)&m->B);
    if (rc != RPHIP_OK) return fail(rc);
  }
  float* S = m->B;
  int rc = RPHIP_OK;
#define DTRY(expr)                    \
  do {                                \
    rc = (expr);                      \
    if (rc != RPHIP_OK) return fail(rc); \
  } while (0)
#define DHIP(expr)                                                                                        \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess) {                                                                               \
      rp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);            \
      return fail(_e == hipErrorOutOfMemory ? RPHIP_ENOMEM : RPHIP_EHIP);                                 \
    }                                                                                                     \
  } while (0)

  DHIP(hipStreamCreateWithFlags(&R.su, hipStreamNonBlocking));
  DHIP(hipStreamCreateWithFlags(&R.sc, hipStreamNonBlocking));
  DHIP(hipStreamCreateWithFlags(&R.sp, hipStreamNonBlocking));
  hipStream_t su = R.su, sc = R.sc, sp = R.sp;
  // V[b]: the pivot column (n_pad rows) + P (nb rows); (W + 1) nb more rows make it the mirror's staging buffer as well
  const size_t v_rows = (size_t)n_pad + (size_t)(W + 2) * nb;
  float *V[2] = {nullptr, nullptr}, *Town[2] = {nullptr, nullptr}, *Vs = nullptr, *Vg = nullptr, *colscale = nullptr, *Ag = nullptr;
  int* info = nullptr;
  for (int b = 0; b < 2; ++b) {
    DTRY(R.alloc(v_rows * nb * sizeof(float), (void**)&V[b]));
    DTRY(R.alloc((size_t)std::max(nl, 1) * nb * nb * sizeof(float), (void**)&Town[b]));
  }
  DTRY(R.alloc((size_t)slots * nb * nb * sizeof(float), (void**)&Vs));
  if (W > 1) DTRY(R.alloc((size_t)std::max(nl, 1) * nb * nb * sizeof(float), (void**)&Ag));  // the own panels' blocks of V right of the pivot, side by side
  if (W > 1)
    DTRY(R.alloc((size_t)W * slots * nb * nb * sizeof(float), (void**)&Vg));
  else
    Vg = Vs;
  DTRY(R.alloc((size_t)ld * sizeof(float) + 256, (void**)&colscale));
  DTRY(R.alloc(256, (void**)&info));
  DHIP(hipMemsetAsync(info, 0, 4, st));
  // plane workspaces of the update stream's products are sized once (a reallocation would pull them from under a running launch)
  {
    void* w = nullptr;
    // (the sizes rp_gemm_nt will ask for: powers of two from 64 MiB, capped at the context's workspace limit - a request that
    // exceeded a smaller pre-sized buffer would free and reallocate it under the running streams)
    auto ws_size = [&](size_t need) {
      size_t c = (size_t)64 << 20;
      while (c < need) c <<= 1;
      return std::max(need, std::min(c, ctx->gemm_ws_bytes));
    };
    const size_t wa = ws_size(std::min<size_t>(ctx->gemm_ws_bytes, (size_t)n_pad * nb * 4));
    const size_t wb = ws_size(std::min<size_t>(ctx->gemm_ws_bytes, (size_t)ld * nb * 4));
    DTRY(rp_scratch(ctx, 12, wa, &w));
    DTRY(rp_scratch(ctx, 13, wb, &w));
    DTRY(rp_scratch(ctx, 8, wa, &w));  // (the main stream's too: growing them later would make the host wait for the stream)
    DTRY(rp_scratch(ctx, 9, wb, &w));
  }
  hipEvent_t e_pack, e_comm, e_chain[2], e_upd, e_la, e_diag, e_inv;
  DTRY(R.event(&e_pack));
  DTRY(R.event(&e_comm));
  DTRY(R.event(&e_chain[0]));
  DTRY(R.event(&e_chain[1]));
  DTRY(R.event(&e_upd));
  DTRY(R.event(&e_la));
  DTRY(R.event(&e_diag));
  DTRY(R.event(&e_inv));
  comm->bytes_recv = comm->bytes_sent = 0;
  double flop = 0.0;

  // ---- G + l2 I, the own panels: Gram rows of a panel (K4) into V[0], transposed into the panel (G is symmetric)
  for (int q = 0; q < nl; ++q) {
    const int p = rp_cyc_panel(q, me, W);
    // (the entry point indexes rows absolutely: row p nb of the matrix is row 0 of the staging buffer)
    DTRY(rp_ease_gram_rows(ctx, X, (int64_t)p * nb, (int64_t)(p + 1) * nb, n_pad, lam, V[0] - (long long)p * nb * n_pad, n_pad));
    DTRY(rp_transpose_f32(ctx, V[0], n_pad, nb, n_pad, S + (long long)q * nb, ld));
  }

  auto gemm = [&](const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N, float al, float be,
                  hipStream_t s, int tri = 0, int m_off = 0, int l0 = 0, int skip0 = 0, int skip1 = 0, int mskip0 = 0, int mskip1 = 0) -> int {
    if (M <= 0 || N <= 0) return RPHIP_OK;
    rp_gemm_args g{};
    g.A = A;
    g.lda = lda;
    g.B = B;
    g.ldb = ldb;
    g.C = C;
    g.ldc = ldc;
    g.M = M;
    g.N = N;
    g.k0 = 0;
    g.k1 = nb;
    g.alpha = al;
    g.beta = be;
    g.tri = tri;
    g.m_off = m_off;
    if (tri) {
      g.cyc_nb = nb;
      g.cyc_w = W;
      g.cyc_r = me;
      g.cyc_l0 = l0;
    }
    g.no_splitk = 1;
    g.n_skip0 = skip0;
    g.n_skip1 = skip1;
    g.m_skip0 = mskip0;
    g.m_skip1 = mskip1;
    if (s == su) g.ws_slot = 12;
    flop += 2.0 * M * (double)(N - (skip1 - skip0)) * nb * (tri ? 0.5 : 1.0);
    return rp_gemm_nt(ctx, g, s);
  };
  auto copy2d = [&](const float* src, long long lds_, float* dst, long long ldd, long long rows, int cols, float scale, hipStream_t s) {
    if (rows <= 0) return;
    const long long quads = rows * (cols / 4);
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)std::min<long long>(8192, rp_cdiv(quads, 256))), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols, scale);
  };
  auto timed = [&](hipStream_t s, hipEvent_t* a) {  // (event pairs on another stream than the context's)
    *a = nullptr;
    if (hipEventCreate(a) == hipSuccess) hipEventRecord(*a, s);
  };
  auto timed_end = [&](hipStream_t s, hipEvent_t a, int slot, double work) {
    hipEvent_t b = nullptr;
    if (!a || hipEventCreate(&b) != hipSuccess) return;
    hipEventRecord(b, s);
    ctx->pending.push_back({a, b, slot});
    ctx->tm.launches[slot] += 1;
    ctx->tm.work[slot] += work;
  };

  // ---- the sweeps
  // The pivot inverse (a chain of nb / 128 float64 diagonal blocks, ~8 ms at nb = 2048) is the one step that nobody can start
  // early: P_K needs the update of pivot K - 1 on its own nb x nb block only.  So that block is updated first (the first thing of
  // the look-ahead), copied out, and inverted on a stream of its own while everything else of the pivot - the rest of the
  // look-ahead, the big update, the next gather / broadcast of the pivot column (which does not need P) - goes on; P follows in a
  // small broadcast of its own.  The recurrence per pivot is then: inverse -> T of one block -> update of one block -> next inverse.
  auto start_inverse = [&](float* Pslot) -> int {  // on the pivot stream, after e_diag; the host waits inside (side-stream joins)
    RP_HIP(hipStreamWaitEvent(sp, e_diag, 0));
    const int ps = ctx->gemm_presplit;
    ctx->gemm_presplit = 0;  // (its products are small; the plane workspaces belong to the streams that are busy now)
    const int r = rp_spd_inverse(ctx, Pslot, nb, nb, info, sp);
    ctx->gemm_presplit = ps;
    if (r != RPHIP_OK) return r;
    RP_HIP(hipEventRecord(e_inv, sp));
    return RPHIP_OK;
  };
  if (np > 0 && rp_cyc_owner(0, W) == me) {  // pivot 0
    copy2d(S, ld, V[0] + (long long)n_pad * nb, nb, nb, nb, 1.0f, st);
    DHIP(hipEventRecord(e_diag, st));
    DTRY(start_inverse(V[0] + (long long)n_pad * nb));
  }
  bool upd_pending = false;
  for (int K = 0; K < np; ++K) {
    const int cur = K & 1;
    const int owner = rp_cyc_owner(K, W), lpK = K / W;
    const int cnt = cnt_lt(K, me);                     // own panels left of the pivot: local panels [0, cnt)
    const int own_K = (owner == me) ? 1 : 0;
    const long long r0 = (long long)K * nb;
    float* Vc = V[cur];
    float* Pc = Vc + (long long)n_pad * nb;            // P = A_KK^-1 sits behind the pivot column (the pivot stream is writing it)
    float* Tc = Town[cur];
    hipEvent_t t_chain;
    timed(st, &t_chain);
    const double flop0 = flop;
    // -- chain(K), main stream.  What it reads - row block K of the own panels, panel K - was finished by the look-ahead of K - 1.
    if (own_K) copy2d(S + (r0 + nb) * ld + (long long)lpK * nb, ld, Vc + (r0 + nb) * nb, nb, (long long)n_pad - r0 - nb, nb, 1.0f, st);
    if (cnt > 0)  // pieces V_i = A_Ki^T of the own panels i < K
      hipLaunchKernelGGL(k_blk_transpose, dim3(nb / 32, nb / 32, cnt), dim3(32, 8), 0, st, S + r0 * ld, ld, (long long)nb, Vs, (long long)nb,
                         (long long)nb * nb);
    DHIP(hipGetLastError());
    const int gslots = (int)rp_cdiv(K, W);  // pieces per rank in this gather (ranks with fewer send filler)
    if (W > 1) {
      DHIP(hipEventRecord(e_pack, st));
      DHIP(hipStreamWaitEvent(sc, e_pack, 0));
      if (gslots > 0) DTRY(rp_comm_all_gather(comm, Vs, Vg, (int64_t)gslots * nb * nb * 4, sc));
      // rows (K + 1) nb .. n_pad of V from the owner - while its pivot stream may still be inverting - then P
      DTRY(rp_comm_broadcast(comm, Vc + (r0 + nb) * nb, ((int64_t)n_pad - r0 - nb) * nb * 4, owner, sc));
      if (own_K) DHIP(hipStreamWaitEvent(sc, e_inv, 0));
      DTRY(rp_comm_broadcast(comm, Pc, (int64_t)nb * nb * 4, owner, sc));
      DHIP(hipEventRecord(e_comm, sc));
      DHIP(hipStreamWaitEvent(st, e_comm, 0));
    } else {
      DHIP(hipStreamWaitEvent(st, e_inv, 0));
    }
    if (gslots > 0)
      hipLaunchKernelGGL(k_unpack_pieces, dim3(64, W * gslots), dim3(256), 0, st, W > 1 ? Vg : Vs, gslots, W, K, nb, Vc, (long long)nb);
    // T_j = V_j P for the own panels: the pieces in one product, the panels right of the pivot one by one (they are every W-th
    // block of V); the owner forms T_i for every i > K straight into its panel K (the new column K)
    DTRY(gemm(Vs, nb, Pc, nb, Tc, nb, cnt * nb, nb, 1.0f, 0.0f, st));
    if (W == 1) {  // (one rank: the panels right of the pivot are consecutive rows of V)
      DTRY(gemm(Vc + (r0 + nb) * nb, nb, Pc, nb, Tc + (long long)(K + 1) * nb * nb, nb, (nl - K - 1) * nb, nb, 1.0f, 0.0f, st));
    } else {
      // (round 5) ... in ONE product when there are several: their blocks of V are copied side by side (16 MB each at nb = 2048) and
      // the product has enough tiles for the wide plane kernel; one by one they ran on the skinny kernel that cuts its planes on
      // the fly (50 ms of an emulated rank's 476 ms fit at 100 000 items and eight ranks, tools/r05_e23.sh)
      const int q0 = cnt + own_K, nq = nl - q0;
      if (nq >= 2) {
        for (int q = q0; q < nl; ++q)
          copy2d(Vc + (long long)rp_cyc_panel(q, me, W) * nb * nb, nb, Ag + (long long)(q - q0) * nb * nb, nb, nb, nb, 1.0f, st);
        DTRY(gemm(Ag, nb, Pc, nb, Tc + (long long)q0 * nb * nb, nb, nq * nb, nb, 1.0f, 0.0f, st));
      } else {
        for (int q = q0; q < nl; ++q) {
          const long long rows_at = (long long)rp_cyc_panel(q, me, W) * nb;
          DTRY(gemm(Vc + rows_at * nb, nb, Pc, nb, Tc + (long long)q * nb * nb, nb, nb, nb, 1.0f, 0.0f, st));
        }
      }
    }
    if (own_K) {
      if (W == 1)  // (one rank: T_own already holds them)
        copy2d(Tc + (long long)(K + 1) * nb * nb, nb, S + (r0 + nb) * ld + (long long)lpK * nb, ld, (long long)n_pad - r0 - nb, nb, 1.0f, st);
      else
        DTRY(gemm(Vc + (r0 + nb) * nb, nb, Pc, nb, S + (r0 + nb) * ld + (long long)lpK * nb, ld, n_pad - (int)r0 - nb, nb, 1.0f, 0.0f, st));
      copy2d(Pc, nb, S + r0 * ld + (long long)lpK * nb, ld, nb, nb, -1.0f, st);  // A_KK = -P
    }
    if (cnt > 0)  // the new row K: A_Ki = T_i^T
      hipLaunchKernelGGL(k_blk_transpose, dim3(nb / 32, nb / 32, cnt), dim3(32, 8), 0, st, Tc, (long long)nb, (long long)nb * nb, S + r0 * ld, ld,
                         (long long)nb);
    DHIP(hipGetLastError());
    DHIP(hipEventRecord(e_chain[cur], st));
    // -- look-ahead part of update K (main stream): what chain(K + 1) reads.  It adds to tiles U(K - 1) also adds to: wait for it.
    const bool own_K1 = K + 1 < np && rp_cyc_owner(K + 1, W) == me;
    const int q1 = (K + 1) / W;
    const long long r1 = r0 + nb;
    if (K + 1 < np) {
      if (upd_pending) DHIP(hipStreamWaitEvent(st, e_upd, 0));
      if (own_K1) {  // the next pivot's own block first: it goes to the pivot stream at once (below)
        DTRY(gemm(Vc + r1 * nb, nb, Tc + (long long)q1 * nb * nb, nb, S + r1 * ld + (long long)q1 * nb, ld, nb, nb, -1.0f, 1.0f, st, 1, (int)r1, q1 * nb));
        copy2d(S + r1 * ld + (long long)q1 * nb, ld, V[cur ^ 1] + (long long)n_pad * nb, nb, nb, nb, 1.0f, st);
        DHIP(hipEventRecord(e_diag, st));
      }
      DTRY(gemm(Vc + r1 * nb, nb, Tc, nb, S + r1 * ld, ld, nb, cnt * nb, -1.0f, 1.0f, st));  // row block K + 1, own panels < K
      if (own_K1)  // panel K + 1 below its diagonal block
        DTRY(gemm(Vc + (r1 + nb) * nb, nb, Tc + (long long)q1 * nb * nb, nb, S + (r1 + nb) * ld + (long long)q1 * nb, ld, n_pad - (int)r1 - nb, nb, -1.0f,
                  1.0f, st));
      DHIP(hipEventRecord(e_la, st));
    }
    timed_end(st, t_chain, RPHIP_T_SWEEP_CHAIN, flop - flop0);
    // -- U(K) on the update stream: own panels != K, K + 1, rows outside the blocks K, K + 1
    {
      const double flop1 = flop;
      DHIP(hipStreamWaitEvent(su, e_chain[cur], 0));
      hipEvent_t t_upd;
      timed(su, &t_upd);
      const int l3 = cnt + own_K + (own_K1 ? 1 : 0);  // first local panel right of K + 1
      const long long r2 = r0 + 2LL * nb;             // first row below the block K + 1
      // (round 5) ONE product for the whole update - three before, the planes of V's rows and of T cut twice: all rows of V except the
      // blocks K and K + 1 (skipped), all own panels except K and K + 1 (skipped), lower tiles (tri = 1).  The tri rule gives each
      // part what it had: rows above the pivot x panels left of it their lower tiles, rows below x panels left every tile (a column
      // left of the pivot lies below all of these rows), rows below x panels right of K + 1 the lower tiles, rows above x panels
      // right nothing.
      if ((cnt > 0 && (r0 > 0 || r2 < n_pad)) || (l3 < nl && r2 < n_pad)) {
        const double f0 = flop;
        const int m_rows = (int)std::min<long long>(n_pad, std::max<long long>(r2 < n_pad ? n_pad : r0, 0));
        DTRY(gemm(Vc, nb, Tc, nb, S, ld, m_rows, nl * nb, -1.0f, 1.0f, su, 1, 0, 0, cnt * nb, l3 * nb, (int)r0, (int)std::min<long long>(r2, m_rows)));
        flop = f0 + 2.0 * nb * (0.5 * (double)r0 * cnt * nb +
                                (r2 < n_pad ? (double)(n_pad - r2) * ((double)cnt * nb + 0.5 * (double)(nl - l3) * nb) : 0.0));
      }
      timed_end(su, t_upd, RPHIP_T_SWEEP_UPDATE, flop - flop1);
      DHIP(hipEventRecord(e_upd, su));
      upd_pending = true;
    }
    // -- the next pivot's inverse, on the pivot stream (everything else of this pivot is enqueued: the host may wait in here)
    if (own_K1) DTRY(start_inverse(V[cur ^ 1] + (long long)n_pad * nb));
  }
  DHIP(hipStreamWaitEvent(st, e_upd, 0));
  // ---- the upper triangle.  Diagonal blocks: upper tiles from the lower ones; block (i, j), i < j, is block (j, i) of panel i's
  // owner transposed: W panels at a time (every rank owns one of them) in one personalised exchange.
  hipEvent_t t_mir;
  rp_time_begin(ctx, &t_mir);
  for (int q = 0; q < nl; ++q) {
    const long long p = rp_cyc_panel(q, me, W);
    hipLaunchKernelGGL(k_mirror_block, dim3(nb / 32, nb / 32), dim3(256), 0, st, S + p * nb * ld + (long long)q * nb, ld);
  }
  DHIP(hipGetLastError());
  {
    std::vector<int64_t> soff(W), sby(W), roff(W), rby(W);
    const int64_t piece = (int64_t)nb * nb * 4;
    float* sendb = V[0];
    float* recvb = V[1];
    for (int g0 = W; g0 < np + W; g0 += W) {  // panels [g0 - W, g0) ... (the first group with a non-empty exchange starts at panel 1)
      const int base = g0 - W;
      if (base >= np) break;
      int64_t so = 0;
      for (int d = 0; d < W; ++d) {
        const int j = rp_cyc_panel(base / W, d, W);  // rank d's panel of this round
        const int c = (j < np) ? cnt_lt(j, me) : 0;
        soff[d] = so;
        sby[d] = (int64_t)c * piece;
        if (c > 0)  // my pieces of row block j, transposed
          hipLaunchKernelGGL(k_blk_transpose, dim3(nb / 32, nb / 32, c), dim3(32, 8), 0, st, S + (long long)j * nb * ld, ld, (long long)nb,
                             sendb + so / 4, (long long)nb, (long long)nb * nb);
        so += sby[d];
      }
      const int jm = rp_cyc_panel(base / W, me, W);  // my panel of this round
      const int gs = (jm < np) ? (int)rp_cdiv(jm, W) : 0;
      for (int s = 0; s < W; ++s) {
        roff[s] = (int64_t)s * gs * piece;
        rby[s] = (jm < np) ? (int64_t)cnt_lt(jm, s) * piece : 0;
      }
      DHIP(hipGetLastError());
      if (W > 1) {
        DHIP(hipEventRecord(e_pack, st));
        DHIP(hipStreamWaitEvent(sc, e_pack, 0));
      }
      DTRY(rp_comm_exchange(comm, sendb, soff.data(), sby.data(), recvb, roff.data(), rby.data(), W > 1 ? sc : st));
      if (W > 1) {
        DHIP(hipEventRecord(e_comm, sc));
        DHIP(hipStreamWaitEvent(st, e_comm, 0));
      }
      if (gs > 0)
        hipLaunchKernelGGL(k_unpack_pieces, dim3(64, W * gs), dim3(256), 0, st, recvb, gs, W, jm, nb, S + (long long)(jm / W) * nb, ld);
      DHIP(hipGetLastError());
    }
  }
  rp_time_end(ctx, t_mir, RPHIP_T_MIRROR, 4.0 * (double)n_pad * ld);
  // ---- K7 on the panels: B_ij = -P_ij / P_jj * w_j, B_jj = 0, padding zeroed (ease.py:85-90)
  hipEvent_t t_sc;
  rp_time_begin(ctx, &t_sc);
  const rp_cyc cy{nb, W, me};
  const int n_local = nl * nb;
  if (n_local > 0) {
    hipLaunchKernelGGL(k_dist_colscale, dim3((unsigned)rp_cdiv(n_local, 256)), dim3(256), 0, st, S, ld, n, n_local, cy, X->deg, (float)alpha, colscale);
    hipLaunchKernelGGL(k_dist_scale, dim3(4096), dim3(256), 0, st, S, ld, (long long)n_pad, n, cy, colscale);
  }
  DHIP(hipGetLastError());
  rp_time_end(ctx, t_sc, RPHIP_T_EASE_SCALE, 8.0 * (double)n_pad * n_local);
  // the flag of a pivot block that broke down is one rank's: summed over the ranks so that EVERY rank returns the same status
  // (the others received the broken pivot's garbage through the broadcasts and would otherwise serve it, or hang in the next
  // exchange while the failing rank raises)
  DTRY(rp_comm_all_reduce_sum_i32(comm, info, 1, st));
  int h_info = 0;
  DHIP(hipMemcpyAsync(&h_info, info, 4, hipMemcpyDeviceToHost, st));
  DHIP(hipStreamSynchronize(st));
  DHIP(hipStreamSynchronize(su));
  DHIP(hipStreamSynchronize(sc));
  DHIP(hipStreamSynchronize(sp));
  ctx->dist_stats[0] = (int64_t)slab_bytes;
  ctx->dist_stats[1] = R.peak + (int64_t)ctx->scratch[12].cap + (int64_t)ctx->scratch[13].cap + (int64_t)ctx->scratch[8].cap + (int64_t)ctx->scratch[9].cap +
                    (int64_t)ctx->scratch[10].cap + (int64_t)ctx->scratch[11].cap + (int64_t)ctx->scratch[5].cap + (int64_t)ctx->scratch[7].cap;
  ctx->dist_stats[2] = comm->bytes_recv;
  ctx->dist_stats[3] = comm->bytes_sent;
  ctx->dist_stats[4] = np;
  ctx->dist_stats[5] = nb;
  ctx->dist_stats[6] = n_pad;
  ctx->dist_stats[7] = (int64_t)flop;
  if (h_info && !comm->is_loop) {  // (a loop-back run works on noise: its result is meaningless by contract, its timing is the point)
    rp_set_error("a pivot block is not positive definite on some rank (flag sum %d): G + l2*I is not positive definite", h_info);
    return fail(RPHIP_ENUMERIC);
  }
  // items among the local columns: a prefix (the padding is the tail of the last panel)
  int64_t n_cols = 0;
  for (int q = 0; q < nl; ++q) n_cols += std::max<int64_t>(0, std::min<int64_t>(nb, (int64_t)n - (int64_t)rp_cyc_panel(q, me, W) * nb));
  m->n_cols = n_cols;
  if (n_cols == 0) {  // this rank owns only padding
    rphip_model_destroy(m);
    return RPHIP_OK;
  }
  *out = m;
  return RPHIP_OK;
#undef DTRY
#undef DHIP
}

// ---- the ranks' column panels into ONE full model on every rank --------------------------------------------------------------
// rows [r0, r0 + rows) of every rank's panels (recv[q][i][a * nb + c], leading dimension ldl) into B[r0 + i][panel(a, q) * nb + c]
__global__ void k_unpack_panels(const float* __restrict__ recv, long long slot_floats, int ldl, int rows, int nb, int W, int np, float* __restrict__ B,
                                long long ldB, long long r0) {
  const int c4 = nb >> 2;
  const int q = blockIdx.z;                 // source rank
  const int a = blockIdx.y;                 // its local panel
  const int p = rp_cyc_panel(a, q, W);      // the panel of the matrix
  if (p >= np) return;
  const float* src = recv + (long long)q * slot_floats + (long long)a * nb;
  float* dst = B + r0 * ldB + (long long)p * nb;
  const long long total = (long long)rows * c4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / c4;
    const int c = (int)(e % c4) * 4;
    *reinterpret_cast<float4*>(dst + i * ldB + c) = *reinterpret_cast<const float4*>(src + i * ldl + c);
  }
}

// ---- the int8 rows of the gathered model (phase 1 of rphip_ease_model_allgather) ------------------------------------------------------
// pm[i] = largest |B[i][c]| over this rank's item columns c < ncl of row i < n_items (0 for the padding rows)
__global__ void __launch_bounds__(256) k_panel_rowmax(const float* __restrict__ B, long long ld, int n_pad, int n_items, int ncl, float* __restrict__ pm) {
  __shared__ float red[4];
  for (int i = blockIdx.x; i < n_pad; i += gridDim.x) {
    float m = 0.0f;
    if (B && i < n_items) {
      const float* row = B + (long long)i * ld;
      for (int j = threadIdx.x * 4; j < ncl; j += 1024) {  // (ld is a multiple of 128: whole quads inside the row; columns >= ncl are padding = 0)
        const float4 v = *reinterpret_cast<const float4*>(row + j);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pm[i] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}
// out[i] = max over the ranks' parts; with_scale: scale[i] = out[i] / 127 (k_dense_to_i8's step; 0 for an all-zero or non-finite row)
__global__ void k_max_over_ranks(const float* __restrict__ g, int n, int W, float* __restrict__ out, float* __restrict__ scale, int with_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m = 0.0f;
  for (int r = 0; r < W; ++r) m = fmaxf(m, g[(long long)r * n + i]);
  out[i] = m;
  if (with_scale) scale[i] = (m > 0.0f && m < 3.0e38f) ? m / 127.0f : 0.0f;
}
// rows [0, rows) of this rank's panels (cols columns, the first ncl of them items) quantised with the rows' global maxima, exactly as
// k_dense_to_i8 does: q = rint(b * 127 / max) clamped to +-127; err[i] = the row's largest |q s - b| over this rank's columns
// (3e38 for a non-finite row).  Rows >= n_real are padding (zeros).
__global__ void __launch_bounds__(256) k_panel_quantise(const float* __restrict__ B, long long ld, int rows, int cols, int ncl, int n_real,
                                                        const float* __restrict__ rowmax, int8_t* __restrict__ Q, long long ldq, float* __restrict__ err) {
  __shared__ float red[4];
  for (int i = blockIdx.x; i < rows; i += gridDim.x) {
    const float* row = B + (long long)i * ld;
    int8_t* out = Q + (long long)i * ldq;
    const float m = rowmax[i];
    const bool ok = m > 0.0f && m < 3.0e38f && i < n_real;
    const float s = ok ? m / 127.0f : 0.0f, inv = ok ? 127.0f / m : 0.0f;
    float e = 0.0f;
    for (int j = threadIdx.x * 16; j < cols; j += 4096) {
      uint32_t w[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(row + j + 4 * g);
        const float x[4] = {v.x, v.y, v.z, v.w};
        uint32_t pk = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool item = j + 4 * g + c < ncl;
          const float qf = item ? fminf(fmaxf(rintf(x[c] * inv), -127.0f), 127.0f) : 0.0f;
          if (item) e = fmaxf(e, fabsf(qf * s - x[c]));
          pk |= ((uint32_t)(int)qf & 0xffu) << (8 * c);
        }
        w[g] = pk;
      }
      *reinterpret_cast<uint4*>(out + j) = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) e = fmaxf(e, __shfl_xor(e, d));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0 && i < n_real) err[i] = (m < 3.0e38f) ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : 3.0e38f;
  }
}
// k_unpack_panels for int8 strips (16 bytes per access)
__global__ void k_unpack_panels_i8(const int8_t* __restrict__ recv, long long slot_bytes, int ldl, int rows, int nb, int W, int np, int8_t* __restrict__ B,
                                   long long ldB, long long r0) {
  const int c16 = nb >> 4;
  const int q = blockIdx.z, a = blockIdx.y;
  const int p = rp_cyc_panel(a, q, W);
  if (p >= np) return;
  const int8_t* src = recv + (long long)q * slot_bytes + (long long)a * nb;
  int8_t* dst = B + r0 * ldB + (long long)p * nb;
  const long long total = (long long)rows * c16;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / c16;
    const int c = (int)(e % c16) * 16;
    *reinterpret_cast<uint4*>(dst + i * ldB + c) = *reinterpret_cast<const uint4*>(src + i * ldl + c);
  }
}

// While 3 n^2 floats fit a GPU (B, its fp16 copy, a staging strip) the fastest multi-GPU predict is the single-GPU one on a share
// of the USERS: every rank gathers all panels (n^2 * 4 * (W - 1) / W bytes in over W - 1 links: 50 ms at 100 000 items and 100 GB/s
// per link) and scores its own users against the whole of B - K9's two passes with their full-width rows, no per-rank candidate lists
// to rescore W times over, no exchange of lists.  The column-sharded rphip_predict_topk_dist stays for matrices that do not fit
// (5M x 200K: B alone is 160 GB).  `local` = this rank's model of rphip_ease_fit_dist (NULL on a rank that owns only padding),
// n_items and panel_cols as every rank's layout reports them.  Strips of rows travel one all-gather each.
static int rp_model_allgather_impl(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, int64_t n_items, int64_t panel_cols, bool i8_only,
                                   rphip_model** out);

extern "C" int rphip_ease_model_allgather(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, int64_t n_items, int64_t panel_cols,
                                          rphip_model** out) {
  RP_ENTER(ctx);
  return rp_model_allgather_impl(ctx, comm, local, n_items, panel_cols, false, out);
}

// (round 5) Only the int8 rows, their scales and error bounds - n^2 bytes instead of 5 n^2: the model of rphip_predict_topk_owner,
// whose exact scores are formed by the owners of the candidates' columns from the panels where the fit left them.  B of the result is
// NULL: every other entry point refuses it.
extern "C" int rphip_ease_model_allgather_i8(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, int64_t n_items, int64_t panel_cols,
                                             rphip_model** out) {
  RP_ENTER(ctx);
  return rp_model_allgather_impl(ctx, comm, local, n_items, panel_cols, true, out);
}

static int rp_model_allgather_impl(rphip_ctx* ctx, rphip_comm* comm, const rphip_model* local, int64_t n_items, int64_t panel_cols, bool i8_only,
                                   rphip_model** out) {
  if (!ctx || !comm || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  const int W = comm->world, me = comm->rank;
  const int nb = (int)panel_cols;
  if (n_items < 1 || nb < 128 || (nb % 128)) RP_FAIL(RPHIP_EINVAL, "n_items %lld, panel of %lld columns", (long long)n_items, (long long)panel_cols);
  if (local && (local->kind != RPHIP_MODEL_DENSE || local->cyc_nb != nb || local->cyc_world != W || local->cyc_rank != me || local->n_items != n_items))
    RP_FAIL(RPHIP_EINVAL, "the local model is not this rank's part of a %d-rank fit with panels of %d columns", W, nb);
  RP_TRY(rp_set_device(ctx));
  const long long n_pad = rp_cdiv(n_items, nb) * nb;
  const int np = (int)(n_pad / nb), nl_max = (int)rp_cdiv(np, W);
  const int nl = rp_cyc_count(np, me, W);
  const int ldl = nl_max * nb;
  hipStream_t st = ctx->stream;
  rphip_model* m = new rphip_model();
  m->ctx = ctx;
  m->kind = RPHIP_MODEL_DENSE;
  m->n_items = n_items;
  m->row_begin = 0;
  m->row_end = n_items;
  m->ld = n_pad;
  // the word the ranks agree on (taken before the large blocks)
  int32_t* agree = nullptr;
  {
    const int ra = rp_dev_alloc(ctx, 256, (void**)&agree);
    if (ra != RPHIP_OK) {
      delete m;
      return ra;
    }
  }
  if (i8_only && !(ctx->predict_half && ctx->predict_int8 && (n_pad % 16) == 0 && comm->world > 1)) {
    rp_dev_free(ctx, agree);
    delete m;
    RP_FAIL(RPHIP_ESTATE, "the int8 rows alone need predict_half, predict_int8, more than one rank and a padded size that is a multiple of 16");
  }
  m->rows_i8_only = i8_only;
  int rc = i8_only ? RPHIP_OK : rp_dev_alloc(ctx, (size_t)n_pad * n_pad * sizeof(float), (void**)&m->B);
  // strips of rows: W slots of strip x ldl floats, at most ~2 GiB of staging in all
  long long strip = std::max<long long>(128, ((long long)((size_t)2 << 30) / ((long long)(W + 1) * ldl * 4)) / 128 * 128);
  strip = std::min(strip, n_pad);
  float *sendb = nullptr, *recvb = nullptr;
  const long long slot = strip * ldl;
  if (rc == RPHIP_OK) rc = rp_dev_alloc(ctx, (size_t)slot * sizeof(float), (void**)&sendb);
  if (rc == RPHIP_OK) rc = rp_dev_alloc(ctx, (size_t)slot * W * sizeof(float), (void**)&recvb);
  {
    //  A rank whose allocations failed must not leave alone: the others would sit in the first all-gather for
    // ever.  The ranks sum a 0 / 1 flag and return RPHIP_ENOMEM TOGETHER; the caller can then take rphip_predict_topk_dist.
    int32_t h_bad = rc == RPHIP_OK ? 0 : 1;
    hipError_t e1 = hipMemcpyAsync(agree, &h_bad, 4, hipMemcpyHostToDevice, st);
    hipError_t e2 = hipStreamSynchronize(st);  // (h_bad is read by the copy)
    const int rca = rp_comm_all_reduce_sum_i32(comm, agree, 1, st);
    int32_t h_sum = 1;
    hipError_t e3 = hipMemcpyAsync(&h_sum, agree, 4, hipMemcpyDeviceToHost, st);
    hipError_t e4 = hipStreamSynchronize(st);
    rp_dev_free(ctx, agree);
    if (rca != RPHIP_OK || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || h_sum != 0) {
      rp_dev_free(ctx, sendb);
      rp_dev_free(ctx, recvb);
      rphip_model_destroy(m);
      if (rca != RPHIP_OK) return rca;
      RP_FAIL(RPHIP_ENOMEM, "%d of %d ranks could not allocate the gathered model (%lld x %lld floats + staging): use rphip_predict_topk_dist", (int)h_sum, W,
              n_pad, n_pad);
    }
  }
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  // ---- phase 1 (round 5): the INT8 rows K9's first pass streams - a quarter of the bytes - complete when this call returns.
  // Row maxima over the ranks' columns (one all-gather of n floats per rank), every rank quantises its own panels with the row's
  // scale exactly as k_dense_to_i8 does on one GPU (q = rint(b * 127 / max), the row's largest |q s - b| measured), the int8 strips
  // travel like the f32 ones.  The emulation of round 4 had the predict of a rank wait 0.10 s for 40 GB of f32 panels that its
  // first pass then read as int8.
  const bool q8 = ctx->predict_half && ctx->predict_int8 && (n_pad % 16) == 0 && W > 1;
  if (q8) {
    const size_t nB = (size_t)n_pad;
    float *pm = nullptr, *gm = nullptr;  // partial [2][n]: row max, then quantisation error; gathered [W][2][n]
    int r8 = rp_dev_alloc(ctx, nB * n_pad, (void**)&m->B16);
    if (r8 == RPHIP_OK) r8 = rp_dev_alloc(ctx, nB * 4, (void**)&m->b8_scale);
    if (r8 == RPHIP_OK) r8 = rp_dev_alloc(ctx, nB * 4, (void**)&m->b16_delta);
    if (r8 == RPHIP_OK) r8 = rp_dev_alloc(ctx, nB * 4, (void**)&m->b16_rowmax);
    if (r8 == RPHIP_OK) r8 = rp_dev_alloc(ctx, nB * 4, (void**)&pm);
    if (r8 == RPHIP_OK) r8 = rp_dev_alloc(ctx, nB * 4 * W, (void**)&gm);
    // (these are small beside B and its staging, which the ranks have agreed on: a failure here is reported, not negotiated)
    if (r8 != RPHIP_OK) {
      rp_dev_free(ctx, pm);
      rp_dev_free(ctx, gm);
      rp_dev_free(ctx, sendb);
      rp_dev_free(ctx, recvb);
      rphip_model_destroy(m);
      return r8;
    }
    const int ncl = local ? (int)local->n_cols : 0;  // items among this rank's local columns (a prefix)
    const int rgrid = (int)std::min<long long>(n_pad, (long long)ctx->n_cu * 16);
    hipLaunchKernelGGL(k_panel_rowmax, dim3(rgrid), dim3(256), 0, st, local ? local->B : nullptr, local ? (long long)local->ld : 0, (int)n_pad, (int)n_items, ncl, pm);
    rc = rp_comm_all_gather(comm, pm, gm, (int64_t)nB * 4, st);
    hipLaunchKernelGGL(k_max_over_ranks, dim3((unsigned)rp_cdiv(n_pad, 256)), dim3(256), 0, st, gm, (int)n_pad, W, m->b16_rowmax, m->b8_scale, 1);
    hipMemsetAsync(pm, 0, nB * 4, st);
    // int8 strips: four times the rows of an f32 strip in the same staging
    const long long strip8 = std::min<long long>(n_pad, strip * 4);
    for (long long r0 = 0; r0 < n_pad && rc == RPHIP_OK; r0 += strip8) {
      const long long rows = std::min(strip8, n_pad - r0);
      if (nl < nl_max || !local) hipMemsetAsync(sendb, 0, (size_t)strip8 * ldl, st);
      if (local && nl > 0)
        hipLaunchKernelGGL(k_panel_quantise, dim3((unsigned)std::min<long long>(rows, (long long)ctx->n_cu * 16)), dim3(256), 0, st, local->B + r0 * local->ld,
                           (long long)local->ld, (int)rows, nl * nb, ncl, (int)std::max<long long>(0, std::min<long long>(rows, n_items - r0)), m->b16_rowmax + r0,
                           reinterpret_cast<int8_t*>(sendb), (long long)ldl, pm + r0);
      rc = rp_comm_all_gather(comm, sendb, recvb, strip8 * ldl, st);
      if (rc != RPHIP_OK) break;
      hipLaunchKernelGGL(k_unpack_panels_i8, dim3((unsigned)std::min<long long>(256, rp_cdiv(rows * (nb / 16), 256)), nl_max, W), dim3(256), 0, st,
                         reinterpret_cast<const int8_t*>(recvb), strip8 * ldl, ldl, (int)rows, nb, W, np, reinterpret_cast<int8_t*>(m->B16), n_pad, r0);
    }
    if (rc == RPHIP_OK) rc = rp_comm_all_gather(comm, pm, gm, (int64_t)nB * 4, st);
    hipLaunchKernelGGL(k_max_over_ranks, dim3((unsigned)rp_cdiv(n_pad, 256)), dim3(256), 0, st, gm, (int)n_pad, W, m->b16_delta, nullptr, 0);
    if (hipGetLastError() != hipSuccess && rc == RPHIP_OK) rc = RPHIP_EHIP;
    rp_dev_free(ctx, pm);
    rp_dev_free(ctx, gm);
    m->b16_kind = 2;
    m->b16_scale = 1.0f;
  }
  // ---- phase 2: the f32 rows (the rescoring and the exact pass read them), on a stream of their own when the int8 rows exist:
  // K9's first pass for this rank's users runs beside it
  if (i8_only) {
    rp_time_end(ctx, ev, RPHIP_T_MIRROR, (double)n_pad * n_pad);
    hipStreamSynchronize(st);  // (the staging is released; the call is synchronous like the other model constructors)
    rp_dev_free(ctx, sendb);
    rp_dev_free(ctx, recvb);
    if (rc != RPHIP_OK) {
      rphip_model_destroy(m);
      return rc;
    }
    m->empty_rows = -1;
    *out = m;
    return RPHIP_OK;
  }
  hipStream_t gs = st;
  if (q8 && rc == RPHIP_OK && !comm->ops.all_gather) {  // (a callback transport completes every operation on the host: nothing to overlap)
    if (hipStreamCreateWithFlags(&m->gather_stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&m->gather_ev, hipEventDisableTiming) == hipSuccess) {
      hipEventRecord(m->gather_ev, st);  // (this rank's panels and the staging's last use)
      hipStreamWaitEvent(m->gather_stream, m->gather_ev, 0);
      gs = m->gather_stream;
    } else {
      if (m->gather_stream) hipStreamDestroy(m->gather_stream);
      m->gather_stream = nullptr;
      m->gather_ev = nullptr;
    }
  }
  for (long long r0 = 0; r0 < n_pad && rc == RPHIP_OK; r0 += strip) {
    const long long rows = std::min(strip, n_pad - r0);
    // this rank's strip with the common leading dimension (a rank with fewer panels pads with zeros)
    if (nl < nl_max || !local) hipMemsetAsync(sendb, 0, (size_t)slot * sizeof(float), gs);
    if (local && nl > 0) {
      const long long quads = rows * (long long)(nl * nb / 4);
      hipLaunchKernelGGL(k_copy2d, dim3((unsigned)std::min<long long>(8192, rp_cdiv(quads, 256))), dim3(256), 0, gs, local->B + r0 * local->ld, (long long)local->ld,
                         sendb, (long long)ldl, rows, nl * nb, 1.0f);
    }
    rc = rp_comm_all_gather(comm, sendb, recvb, slot * (long long)sizeof(float), gs);
    if (rc != RPHIP_OK) break;
    hipLaunchKernelGGL(k_unpack_panels, dim3((unsigned)std::min<long long>(256, rp_cdiv(rows * (nb / 4), 256)), nl_max, W), dim3(256), 0, gs, recvb, slot, ldl,
                       (int)rows, nb, W, np, m->B, n_pad, r0);
    if (hipGetLastError() != hipSuccess) rc = RPHIP_EHIP;
  }
  rp_time_end(ctx, ev, RPHIP_T_MIRROR, 8.0 * (double)n_pad * n_pad);
  if (gs != st && rc == RPHIP_OK) {
    // asynchronous: `local` must stay alive until the event (rp_model_gather_wait is called by everything that reads B, by
    // rphip_model_destroy of this model - and the caller keeps `local` at least as long as the gathered model)
    hipEventRecord(m->gather_ev, gs);
    m->gather_tmp[0] = sendb;
    m->gather_tmp[1] = recvb;
    m->empty_rows = -1;  // (counted when asked for: rphip_model_empty_rows)
    *out = m;
    return RPHIP_OK;
  }
  hipStreamSynchronize(gs);
  if (m->gather_ev) hipEventDestroy(m->gather_ev);
  if (m->gather_stream) hipStreamDestroy(m->gather_stream);
  m->gather_ev = nullptr;
  m->gather_stream = nullptr;
  rp_dev_free(ctx, sendb);
  rp_dev_free(ctx, recvb);
  if (rc == RPHIP_OK) rc = rp_dense_count_empty(ctx, m);
  if (rc != RPHIP_OK) {
    rphip_model_destroy(m);
    return rc;
  }
  *out = m;
  return RPHIP_OK;
}

int rp_model_gather_wait(rphip_ctx* ctx, const rphip_model* m, bool on_stream) {
  if (!m || !m->gather_ev) return RPHIP_OK;
  if (on_stream)
    RP_HIP(hipStreamWaitEvent(ctx->stream, m->gather_ev, 0));
  else
    RP_HIP(hipEventSynchronize(m->gather_ev));
  // the staging goes back to the block cache: a freed block carries the context's stream position, which now lies behind the gather
  rp_dev_free(ctx, m->gather_tmp[0]);
  rp_dev_free(ctx, m->gather_tmp[1]);
  m->gather_tmp[0] = m->gather_tmp[1] = nullptr;
  hipEventDestroy(m->gather_ev);       // (both are released by the runtime once their work is complete)
  hipStreamDestroy(m->gather_stream);
  m->gather_ev = nullptr;
  m->gather_stream = nullptr;
  return RPHIP_OK;
}

// (diagnostic, include/recpack_hip_diag.h) `dst` takes over the pending f32 gather of `src`: tools/dist_rank_emulate.py predicts on the
// REAL matrix of a single-GPU fit while the emulated rank's gather - of noise - is still in flight, and the rescoring of that predict
// has to wait for it as the rank's own would
extern "C" int rphip_model_adopt_gather(rphip_model* dst, rphip_model* src) {
  if (!dst || !src) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_ENTER(dst->ctx);
  if (dst->gather_ev) RP_FAIL(RPHIP_ESTATE, "the model already waits for a gather");
  dst->gather_ev = src->gather_ev;
  dst->gather_stream = src->gather_stream;
  dst->gather_tmp[0] = src->gather_tmp[0];
  dst->gather_tmp[1] = src->gather_tmp[1];
  src->gather_ev = nullptr;
  src->gather_stream = nullptr;
  src->gather_tmp[0] = src->gather_tmp[1] = nullptr;
  return RPHIP_OK;
}

extern "C" int rphip_model_dense_layout(const rphip_model* m, int64_t* n_local_cols, int64_t* panel, int32_t* world, int32_t* rank,
                                        int64_t* col_begin) {
  if (!m) RP_FAIL(RPHIP_EINVAL, "model is NULL");
  if (m->kind != RPHIP_MODEL_DENSE) RP_FAIL(RPHIP_ESTATE, "model is not dense");
  if (n_local_cols) *n_local_cols = m->n_cols > 0 ? m->n_cols : m->n_items;
  if (panel) *panel = m->cyc_nb;
  if (world) *world = m->cyc_nb ? m->cyc_world : 1;
  if (rank) *rank = m->cyc_nb ? m->cyc_rank : 0;
  if (col_begin) *col_begin = m->col_begin;
  return RPHIP_OK;
}
