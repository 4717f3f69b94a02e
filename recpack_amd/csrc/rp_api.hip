// Context, errors, timing and small device utilities of librecpack_hip.so.
#include "rp_common.h"

#include <string.h>

#include <algorithm>

static thread_local char g_err[1024] = "";

void rp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int rphip_version(void) { return RPHIP_VERSION; }
extern "C" const char* rphip_last_error(void) { return g_err; }

extern "C" int rphip_device_count(int* n_out) {
  if (!n_out) RP_FAIL(RPHIP_EINVAL, "n_out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *n_out = 0;
    RP_FAIL(RPHIP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *n_out = n;
  return RPHIP_OK;
}

int rp_set_device(const rphip_ctx* ctx) {
  RP_HIP(hipSetDevice(ctx->device));
  return RPHIP_OK;
}

extern "C" int rphip_ctx_create(int device, void* stream, rphip_ctx** out) {
  if (!out) RP_FAIL(RPHIP_EINVAL, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) RP_FAIL(RPHIP_ENODEV, "no HIP device visible");
  if (device < 0 || device >= n) RP_FAIL(RPHIP_EINVAL, "device %d out of range (0..%d)", device, n - 1);
  RP_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  RP_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    RP_FAIL(RPHIP_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  rphip_ctx* c = new rphip_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  c->hbm_bytes = prop.totalGlobalMem;
  // freed blocks the context keeps for the next fit: at most half of the device (torch's allocator cannot see them; see
  // rphip_ctx_trim in the header for why they are kept at all)
  c->cache_limit = std::min(c->cache_limit, (size_t)prop.totalGlobalMem / 2);
  strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      RP_FAIL(RPHIP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    c->own_stream = true;
  }
  memset(&c->tm, 0, sizeof(c->tm));
  if (hipMalloc((void**)&c->dev_counters, 64) != hipSuccess || hipMemset(c->dev_counters, 0, 64) != hipSuccess) {
    delete c;
    RP_FAIL(RPHIP_ENOMEM, "hipMalloc of the counter block failed");
  }
  *out = c;
  return RPHIP_OK;
}

extern "C" int rphip_ctx_destroy(rphip_ctx* ctx) {
  if (!ctx) return RPHIP_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->pending) {
    hipEventDestroy(p.a);
    hipEventDestroy(p.b);
  }
  rp_dev_trim(ctx);
  for (auto e : ctx->ev_pool) hipEventDestroy(e);
  for (auto& b : ctx->scratch)
    if (b.p) hipFree(b.p);
  if (ctx->dev_counters) hipFree(ctx->dev_counters);
  if (ctx->gemm_amax_ring) hipFree(ctx->gemm_amax_ring);
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return RPHIP_OK;
}

extern "C" int rphip_ctx_sync(rphip_ctx* ctx) {
  RP_ENTER(ctx);
  if (!ctx) RP_FAIL(RPHIP_EINVAL, "ctx is NULL");
  RP_TRY(rp_set_device(ctx));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

extern "C" int rphip_device_info(rphip_ctx* ctx, int* n_cu, int64_t* hbm_bytes, char* arch_buf, int arch_buf_len) {
  RP_ENTER(ctx);
  if (!ctx) RP_FAIL(RPHIP_EINVAL, "ctx is NULL");
  if (n_cu) *n_cu = ctx->n_cu;
  if (hbm_bytes) *hbm_bytes = (int64_t)ctx->hbm_bytes;
  if (arch_buf && arch_buf_len > 0) {
    strncpy(arch_buf, ctx->arch, arch_buf_len - 1);
    arch_buf[arch_buf_len - 1] = 0;
  }
  return RPHIP_OK;
}

int rp_scratch(rphip_ctx* ctx, int slot, size_t bytes, void** out) {
  rp_buf& b = ctx->scratch[slot];
  if (bytes > b.cap) {
    if (b.p) {
      RP_HIP(hipStreamSynchronize(ctx->stream));
      RP_HIP(hipFree(b.p));
      b.p = nullptr;
      b.cap = 0;
    }
    size_t cap = (bytes + (size_t)4095) & ~(size_t)4095;
    hipError_t e = hipMalloc(&b.p, cap);
    if (e != hipSuccess && !ctx->free_blocks.empty()) {  // the cached blocks of destroyed models / matrices are ours to give back
      (void)hipGetLastError();
      RP_HIP(hipStreamSynchronize(ctx->stream));
      rp_dev_trim(ctx);
      e = hipMalloc(&b.p, cap);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      RP_FAIL(RPHIP_ENOMEM, "hipMalloc of %zu bytes of scratch failed: %s", cap, hipGetErrorString(e));
    }
    b.cap = cap;
  }
  *out = b.p;
  return RPHIP_OK;
}

// the plane workspaces of the large GEMMs (slots 8 - 13: tens of GB after a solve) go back to the driver; the caller has synchronised
// the stream.  rphip_ctx_trim, and a predict whose quantised copy of the model would not fit beside them (rp_model_half)
int rp_scratch_release_planes(rphip_ctx* ctx) {
  for (int slot = 8; slot < 14; ++slot) {
    rp_buf& b = ctx->scratch[slot];
    if (b.p) RP_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  return RPHIP_OK;
}

// ---- block cache -------------------------------------------------------------------------------------
// A freed block carries an event recorded on the context's stream at the time of the free: whoever gets the block
// next waits for it first, so kernels that were still reading the block when its owner was destroyed are done
// (hipFree would have synchronised the device; the cache must not be weaker than that for host-side copies).
static void rp_block_wait(rphip_ctx::block_t& b) {
  for (hipEvent_t e : b.freed) {
    hipEventSynchronize(e);
    hipEventDestroy(e);
  }
  b.freed.clear();
}

// join free blocks that touch inside one region
static void rp_dev_coalesce(rphip_ctx* ctx) {
  auto& fb = ctx->free_blocks;
  std::sort(fb.begin(), fb.end(), [](const rphip_ctx::block_t& a, const rphip_ctx::block_t& b) { return a.p < b.p; });
  size_t o = 0;
  for (size_t i = 0; i < fb.size(); ++i) {
    if (o > 0 && fb[o - 1].region == fb[i].region && fb[o - 1].p + fb[o - 1].bytes == fb[i].p) {
      fb[o - 1].bytes += fb[i].bytes;
      fb[o - 1].freed.insert(fb[o - 1].freed.end(), fb[i].freed.begin(), fb[i].freed.end());
    } else {
      if (o != i) fb[o] = std::move(fb[i]);
      ++o;
    }
  }
  fb.resize(o);
}

// release every cached block that is a whole region (a block whose region still has a live part stays cached)
void rp_dev_trim(rphip_ctx* ctx) {
  rp_dev_coalesce(ctx);
  auto& fb = ctx->free_blocks;
  size_t o = 0;
  for (size_t i = 0; i < fb.size(); ++i) {
    if (fb[i].p == fb[i].region && fb[i].bytes == fb[i].region_bytes) {
      for (hipEvent_t e : fb[i].freed) hipEventDestroy(e);  // (hipFree waits for the device)
      hipFree(fb[i].p);
      ctx->cached_bytes -= fb[i].bytes;
    } else {
      if (o != i) fb[o] = std::move(fb[i]);
      ++o;
    }
  }
  fb.resize(o);
}

// how many of `n` requests of `bytes` the cache would serve without going to the driver (the rule of rp_dev_alloc, on a copy)
int rp_dev_cache_serves(rphip_ctx* ctx, size_t bytes, int n) {
  const size_t want = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
  rp_dev_coalesce(ctx);
  std::vector<size_t> have;
  for (auto& b : ctx->free_blocks) have.push_back(b.bytes);
  int served = 0;
  for (int r = 0; r < n; ++r) {
    size_t best = (size_t)-1;
    for (size_t i = 0; i < have.size(); ++i)
      if (have[i] >= want && want >= have[i] / 16 && (best == (size_t)-1 || have[i] < have[best])) best = i;
    if (best == (size_t)-1) break;
    have[best] -= want;
    ++served;
  }
  return served;
}

int rp_dev_alloc(rphip_ctx* ctx, size_t bytes, void** out) {
  *out = nullptr;
  const size_t want = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
  for (int pass = 0; pass < 2; ++pass) {
    size_t best = (size_t)-1;
    for (size_t i = 0; i < ctx->free_blocks.size(); ++i) {
      const size_t have = ctx->free_blocks[i].bytes;
      if (have >= want && want >= have / 16 && (best == (size_t)-1 || have < ctx->free_blocks[best].bytes)) best = i;
    }
    if (best != (size_t)-1) {
      rphip_ctx::block_t b = std::move(ctx->free_blocks[best]);
      rp_block_wait(b);
      if (b.bytes > want) {  // the rest stays cached (nothing pending on it any more)
        ctx->free_blocks[best] = rphip_ctx::block_t{b.p + want, b.bytes - want, b.region, b.region_bytes, {}};
      } else {
        ctx->free_blocks.erase(ctx->free_blocks.begin() + best);
      }
      ctx->cached_bytes -= want;
      ctx->live_blocks[b.p] = rphip_ctx::live_t{want, b.region, b.region_bytes};
      *out = b.p;
      return RPHIP_OK;
    }
    if (pass == 0 && ctx->free_blocks.size() > 1) rp_dev_coalesce(ctx);  // joined neighbours may fit
    else break;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess && !ctx->free_blocks.empty()) {  // give the cache back and try again
    (void)hipGetLastError();
    rp_dev_trim(ctx);
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    RP_FAIL(RPHIP_ENOMEM, "hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
  }
  ctx->live_blocks[p] = rphip_ctx::live_t{want, (char*)p, want};
  *out = p;
  return RPHIP_OK;
}

void rp_dev_free(rphip_ctx* ctx, void* p) {
  if (!p) return;
  auto it = ctx ? ctx->live_blocks.find(p) : decltype(ctx->live_blocks.find(p))();
  if (!ctx || it == ctx->live_blocks.end()) {
    hipFree(p);
    return;
  }
  const rphip_ctx::live_t lv = it->second;
  ctx->live_blocks.erase(it);
  const bool whole = (char*)p == lv.region && lv.bytes == lv.region_bytes;
  if (whole && ctx->cached_bytes + lv.bytes > ctx->cache_limit) {
    hipFree(p);
    return;
  }
  rphip_ctx::block_t b{(char*)p, lv.bytes, lv.region, lv.region_bytes, {}};
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
    hipEventRecord(ev, ctx->stream);
    b.freed.push_back(ev);
  } else {
    hipStreamSynchronize(ctx->stream);
  }
  ctx->free_blocks.push_back(std::move(b));
  ctx->cached_bytes += lv.bytes;
  if (!whole) {  // a part of a split region: join it with its neighbours; a region that is whole again obeys the limit
    rp_dev_coalesce(ctx);
    if (ctx->cached_bytes > ctx->cache_limit) {
      auto& fb = ctx->free_blocks;
      for (size_t i = 0; i < fb.size(); ++i)
        if (fb[i].region == lv.region && fb[i].p == fb[i].region && fb[i].bytes == fb[i].region_bytes) {
          for (hipEvent_t e : fb[i].freed) hipEventDestroy(e);
          hipFree(fb[i].p);
          ctx->cached_bytes -= fb[i].bytes;
          fb.erase(fb.begin() + i);
          break;
        }
    }
  }
}

extern "C" int rphip_ctx_trim(rphip_ctx* ctx) {
  RP_ENTER(ctx);
  if (!ctx) RP_FAIL(RPHIP_EINVAL, "ctx is NULL");
  RP_TRY(rp_set_device(ctx));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  rp_dev_trim(ctx);
  return rp_scratch_release_planes(ctx);
}

extern "C" int rphip_ctx_set_option(rphip_ctx* ctx, const char* name, int64_t value) {
  RP_ENTER(ctx);
  if (!ctx || !name) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (!strcmp(name, "predict_exact64")) {
    ctx->predict_exact64 = value ? 1 : 0;
  } else if (!strcmp(name, "predict_half")) {
    ctx->predict_half = value ? 1 : 0;
  } else if (!strcmp(name, "predict_int8")) {
    ctx->predict_int8 = value ? 1 : 0;
  } else if (!strcmp(name, "predict_user_tasks")) {
    ctx->predict_user_tasks = value ? 1 : 0;
  } else if (!strcmp(name, "predict_second_list")) {
    ctx->predict_second_list = value ? 1 : 0;
    ctx->predict_second_list_hist = value > 1 ? (int)std::min<int64_t>(value, 1 << 30) : 1024;
  } else if (!strcmp(name, "comm_broadcast_split")) {
    ctx->comm_broadcast_split = value ? 1 : 0;
  } else if (!strcmp(name, "predict_half_tile")) {
    if (value < 0 || value > 16384) RP_FAIL(RPHIP_EINVAL, "predict_half_tile must be 0 .. 16384 columns");
    ctx->predict_half_w = (int)value;
  } else if (!strcmp(name, "predict_tile32")) {
    if (value != 2048 && value != 4096) RP_FAIL(RPHIP_EINVAL, "predict_tile32 must be 2048 or 4096");
    ctx->pw_wt32 = (int)value;
  } else if (!strcmp(name, "gemm_mode")) {
    if (value != 0 && value != 2) RP_FAIL(RPHIP_EINVAL, "gemm_mode must be 0 (fp32 MFMA) or 2 (two fp16 planes)");
    ctx->gemm_mode = (int)value;
  } else if (!strcmp(name, "gemm_wide")) {
    ctx->gemm_wide = value ? 1 : 0;
  } else if (!strcmp(name, "chol_sub_panel")) {
    if (value < 0 || value > 32) RP_FAIL(RPHIP_EINVAL, "chol_sub_panel must be 0 .. 32 tiles");
    ctx->chol_sub = (int)value;
  } else if (!strcmp(name, "chol_panel")) {
    if (value != 0 && value != 8 && value != 16 && value != 32 && value != 64) RP_FAIL(RPHIP_EINVAL, "chol_panel must be 0 (automatic), 8, 16, 32 or 64 tiles");
    ctx->chol_panel = (int)value;
  } else if (!strcmp(name, "inv_panel")) {
    if (value != 8 && value != 16 && value != 32) RP_FAIL(RPHIP_EINVAL, "inv_panel must be 8, 16 or 32 tiles");
    ctx->inv_panel = (int)value;
  } else if (!strcmp(name, "inv_n22_early")) {
    if (value < 0 || value > 4096) RP_FAIL(RPHIP_EINVAL, "inv_n22_early must be 0 .. 4096 panels");
    ctx->inv_n22_early = (int)value;
  } else if (!strcmp(name, "gemm_plane_cache")) {
    if (value < 0) RP_FAIL(RPHIP_EINVAL, "gemm_plane_cache must be >= 0");
    ctx->gemm_plane_cache = value;
  } else if (!strcmp(name, "gemm_presplit")) {
    ctx->gemm_presplit = value ? 1 : 0;
  } else if (!strcmp(name, "gemm_workspace_bytes")) {
    if (value < (64 << 20)) RP_FAIL(RPHIP_EINVAL, "gemm_workspace_bytes must be at least 64 MiB");
    ctx->gemm_ws_bytes = (size_t)value;
  } else if (!strcmp(name, "gemm_supertile")) {
    ctx->gemm_supertile = value ? 1 : 0;
  } else if (!strcmp(name, "gemm_xcd_block")) {
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32) RP_FAIL(RPHIP_EINVAL, "gemm_xcd_block must be 0, 1, 2, 4, 8, 16 or 32 tile rows");
    ctx->gemm_xcd_block = (int)value;
  } else if (!strcmp(name, "gemm_flush_k")) {
    if (value < 16 || (value % 16)) RP_FAIL(RPHIP_EINVAL, "gemm_flush_k must be a positive multiple of 16");
    ctx->gemm_kc = (int)std::min<int64_t>(value, 1 << 30);
  } else if (!strcmp(name, "cache_limit_bytes")) {
    if (value < 0) RP_FAIL(RPHIP_EINVAL, "cache_limit_bytes must be >= 0");
    ctx->cache_limit = (size_t)value;
    if (ctx->cached_bytes > ctx->cache_limit) {
      RP_HIP(hipStreamSynchronize(ctx->stream));
      rp_dev_trim(ctx);
    }
  } else {
    RP_FAIL(RPHIP_EINVAL, "unknown option '%s'", name);
  }
  return RPHIP_OK;
}

// ---- timing ----------------------------------------------------------------------------------------
static hipEvent_t rp_get_event(rphip_ctx* ctx) {
  if (!ctx->ev_pool.empty()) {
    hipEvent_t e = ctx->ev_pool.back();
    ctx->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  hipEventCreate(&e);
  return e;
}

void rp_time_begin(rphip_ctx* ctx, hipEvent_t* a) {
  *a = rp_get_event(ctx);
  hipEventRecord(*a, ctx->stream);
}

void rp_time_end(rphip_ctx* ctx, hipEvent_t a, int slot, double work) {
  hipEvent_t b = rp_get_event(ctx);
  hipEventRecord(b, ctx->stream);
  ctx->pending.push_back({a, b, slot});
  ctx->tm.launches[slot] += 1;
  ctx->tm.work[slot] += work;
  if (ctx->pending.size() > 4096) rp_resolve_timings(ctx);
}

int rp_resolve_timings(rphip_ctx* ctx) {
  if (ctx->pending.empty()) return RPHIP_OK;
  RP_HIP(hipStreamSynchronize(ctx->stream));
  for (auto& p : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) ctx->tm.ms[p.slot] += ms;
    ctx->ev_pool.push_back(p.a);
    ctx->ev_pool.push_back(p.b);
  }
  ctx->pending.clear();
  return RPHIP_OK;
}

extern "C" int rphip_get_timings(rphip_ctx* ctx, rphip_timings* out) {
  RP_ENTER(ctx);
  if (!ctx || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_set_device(ctx));
  RP_TRY(rp_resolve_timings(ctx));
  *out = ctx->tm;
  return RPHIP_OK;
}

extern "C" int rphip_get_counters(rphip_ctx* ctx, int64_t* out8) {
  RP_ENTER(ctx);
  if (!ctx || !out8) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_set_device(ctx));
  RP_HIP(hipMemcpyAsync(out8, ctx->dev_counters, 64, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

extern "C" int rphip_reset_timings(rphip_ctx* ctx) {
  RP_ENTER(ctx);
  if (!ctx) RP_FAIL(RPHIP_EINVAL, "ctx is NULL");
  RP_TRY(rp_set_device(ctx));
  RP_TRY(rp_resolve_timings(ctx));
  memset(&ctx->tm, 0, sizeof(ctx->tm));
  return RPHIP_OK;
}

// ---- exclusive scan (int32) ------------------------------------------------------------------------
constexpr int SCAN_T = 1024;
constexpr int SCAN_E = 4;                 // elements per thread
constexpr int SCAN_B = SCAN_T * SCAN_E;   // elements per block

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* smem /* >= 17 ints */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = wave_incl_scan(v);
  if (lane == 63) smem[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int w = (lane < (int)(blockDim.x >> 6)) ? smem[lane] : 0;
    int wi = wave_incl_scan(w);
    if (lane < 16) smem[lane] = wi - w;
    if (lane == 15 || lane == (int)(blockDim.x >> 6) - 1) smem[16] = wi;
  }
  __syncthreads();
  int res = inc - v + smem[wave];
  *total = smem[16];
  __syncthreads();
  return res;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_block_sums(const int32_t* __restrict__ in, int32_t* __restrict__ bsum, int64_t n) {
  __shared__ int sm[32];
  int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_E;
  int s = 0;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e)
    if (base + e < n) s += in[base + e];
  int tot;
  block_excl_scan(s, &tot, sm);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_sums(int32_t* __restrict__ bsum, int64_t nb, int32_t* __restrict__ total_out) {
  __shared__ int sm[32];
  int carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += SCAN_T) {
    int64_t i = b0 + threadIdx.x;
    int v = (i < nb) ? bsum[i] : 0;
    int tot;
    int ex = block_excl_scan(v, &tot, sm);
    if (i < nb) bsum[i] = ex + carry;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_final(const int32_t* __restrict__ in, const int32_t* __restrict__ bsum,
                                                     int32_t* __restrict__ out, int64_t n) {
  __shared__ int sm[32];
  int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_E;
  int v[SCAN_E];
  int s = 0;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) {
    v[e] = (base + e < n) ? in[base + e] : 0;
    s += v[e];
  }
  int tot;
  int ex = block_excl_scan(s, &tot, sm) + bsum[blockIdx.x];
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) {
    if (base + e < n) out[base + e] = ex;
    ex += v[e];
  }
}

// n <= SCAN_B: the whole scan in one launch (K0's bucket tables: three launches of 5 us each for 196 numbers)
__global__ void __launch_bounds__(SCAN_T) k_scan_small(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n) {
  __shared__ int sm[32];
  const int base = (int)threadIdx.x * SCAN_E;
  int v[SCAN_E];
  int s = 0;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) {
    v[e] = (base + e < n) ? in[base + e] : 0;
    s += v[e];
  }
  int tot;
  int ex = block_excl_scan(s, &tot, sm);
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) {
    if (base + e < n) out[base + e] = ex;
    ex += v[e];
  }
  if (threadIdx.x == 0) out[n] = tot;
}

int rp_exclusive_scan_i32(rphip_ctx* ctx, const int32_t* in, int32_t* out, int64_t n) {
  if (n <= 0) {
    RP_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), ctx->stream));
    return RPHIP_OK;
  }
  if (n <= SCAN_B) {
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(SCAN_T), 0, ctx->stream, in, out, (int)n);
    RP_HIP(hipGetLastError());
    return RPHIP_OK;
  }
  int64_t nb = rp_cdiv(n, SCAN_B);
  int32_t* bsum = nullptr;
  RP_TRY(rp_scratch(ctx, 7, (size_t)nb * sizeof(int32_t), (void**)&bsum));
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(SCAN_T), 0, ctx->stream, in, bsum, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, ctx->stream, bsum, nb, out + n);
  hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(SCAN_T), 0, ctx->stream, in, bsum, out, n);
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}
