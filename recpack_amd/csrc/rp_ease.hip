// EASE fit (K4-K7): dense Gram, blocked Cholesky + triangular inverse + product on fp32 MFMA tiles, scaling.
//
// Replaces EASE._fit (recpack/algorithms/ease.py:63-95):
//   XTX = (X.T @ X).toarray()                         -> K4  rp_gram_dense (rp_itemknn.hip), lambda fused on the diagonal
//   P = np.linalg.inv(XTX + l2 * I)  (LU, float64)     -> K5  G + lambda I = L L^T   (left-looking blocked Cholesky)
//                                                         K6  N = L^-T (blocked triangular inverse), P = N N^T
//   B = I - P @ diag(1/diag P); B[diag] = 0; B @ diag(w) -> K7  one elementwise pass B_ij = -P_ij / P_jj * w_j
// The matrix is SPD (lambda > 0), so Cholesky replaces the reference's general LU at half the flops.
//
// Every O(n^3) step is the same kernel: C = alpha * A * B^T + beta * C on 128x128 tiles with
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-exact fp32 FMA chains, no reduced precision), both
// operands K-contiguous ("NT"), K ranges clipped per tile for the triangular factors.  The 128x128 diagonal
// blocks are factorised and inverted by one workgroup in float64 in LDS.
//
//   Cholesky, panel k : A[k:,k] -= L[k:,0:k] L[k,0:k]^T ; L_kk, W_k = chol(A_kk), L_kk^-1 ; L[k+1:,k] = A[k+1:,k] W_k^T
//   inverse, column i : N_ii = W_i^T ; T = N[0:i,0:i] L[i,0:i]^T (K from the row's first non-zero) ; N[0:i,i] = -T W_i^T
//   product           : P = N N^T, K from max(row tile, column tile)
#include <algorithm>
#include <vector>

#include "rp_common.h"
#include "rp_gemm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));


constexpr int GLD = 20;   // LDS row stride in floats (16 + 4 pad: conflict-free ds_read_b128 over 16 rows)
[[maybe_unused]] constexpr int GKC = 1024;  // default K steps between two flushes of the MFMA accumulators into the compensated master sums (option gemm_flush_k)


// Which tile a workgroup computes.  Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2.  In
// plain row-major order the 64 workgroups resident on an XCD are 64 different column tiles of one tile row: one A panel shared
// by all of them and 64 B panels that nobody else on that XCD reads - B streams from HBM once per tile row (1.4 TB/s measured
// over the solve at 50 000 items).  Here the grid is 1-D and an XCD's 64 consecutive workgroups form one SUPERTILE of sw x 64/sw
// tiles (8 x 8 when the product is that wide: 8 + 8 panels for 64 tiles); supertile s goes to XCD s % 8, so that the XCDs walk
// the matrix side by side in the same row-major (top = longest K first) order.  (Giving each XCD one contiguous range of tiles
// instead was measured in round 1 and lost 50 % on P = N N^T: all the long rows land on one XCD.)
__device__ __forceinline__ bool rp_gemm_tile(const rp_gemm_args& g, int& tm, int& tn) {
  if (g.sw == 0) {
    tm = blockIdx.y;
    tn = (g.tri == 3) ? (int)blockIdx.y : (int)blockIdx.x;
    return true;
  }
  const unsigned L = blockIdx.x, c = L >> 3;
  const unsigned sup = (c >> 6) * 8 + (L & 7), w = c & 63;
  const unsigned gx = g.N / GT, gy = g.M / GT, sh = 64u / g.sw;
  const unsigned sxn = (gx + g.sw - 1) / g.sw;
  const unsigned sy = sup / sxn, sx = sup - sy * sxn;
  tn = (int)(sx * g.sw + w % g.sw);
  tm = (int)(sy * sh + w / g.sw);
  return (unsigned)tn < gx && (unsigned)tm < gy;
}

// global column of C's tile column n0 (see rp_gemm_args::cyc_nb), and the `tri` rules
__device__ __forceinline__ int rp_gemm_ncol(const rp_gemm_args& g, int n0) {
  if (!g.cyc_nb) return g.n_off + n0;
  const int l = g.cyc_l0 + n0;
  return rp_cyc_panel(l / g.cyc_nb, g.cyc_r, g.cyc_w) * g.cyc_nb + l % g.cyc_nb;
}
__device__ __forceinline__ bool rp_gemm_skip(const rp_gemm_args& g, int m0, int n0) {
  if (n0 >= g.n_skip0 && n0 < g.n_skip1) return true;
  if (m0 >= g.m_skip0 && m0 < g.m_skip1) return true;
  if (g.tri == 1) return rp_gemm_ncol(g, n0) > g.m_off + m0;
  if (g.tri == 2) return rp_gemm_ncol(g, n0) >= g.m_off + m0;
  if (g.tri == 4) return rp_gemm_ncol(g, n0) > g.m_off + m0;  // (the wide plane product only: diagonal tiles go to g.Cd)
  return false;
}

// Accumulation.  The MFMA adds into its fp32 accumulator in K order: over a K of 10^5 that alone loses ~ sqrt(K) eps, and
// the Cholesky / inverse / product chain of EASE multiplies it by the condition of G + lambda I - measured against
// host-certified float64 columns the plain kernel was 6e-4 off at 100 000 items (tests/ease_cg.py), six times the 1e-4 bar.
// So every GKC = 512 steps of K (256: 8.2e-6 / 9.26 s at C4, 512: 1.2e-5 / 8.82 s, 1024: 2.3e-5 / 8.66 s) the accumulators are emptied into MASTER sums held as an unevaluated pair hi + lo
// (Knuth's two-sum: the rounding error of hi + acc is computed exactly and added to lo).  hi lives in registers; lo only
// needs a few bits (|lo| <= ulp(hi)), so it is kept as bf16 in LDS, two per word (32 KB per workgroup: two workgroups per
// CU still fit, and the 64 extra registers of an fp32 lo would have cost the second wave per SIMD).  The error of a sum
// is then that of its 256-step pieces, whatever K is.  The epilogue forms beta*C + alpha*hi with ONE rounding (fma) and adds
// alpha*lo, so the cancellation of C - L L^T happens before anything is rounded to fp32.
__device__ __forceinline__ uint32_t rp_pack_bf16_rn(float a, float b) {  // round to nearest even; finite inputs
  uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua = (ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16;
  ub = (ub + 0x7fffu + ((ub >> 16) & 1u)) & 0xffff0000u;
  return ua | ub;
}

__global__ void __launch_bounds__(256, 2) k_gemm_nt(rp_gemm_args g) {
  __shared__ __attribute__((aligned(16))) float As[2][GT * GLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][GT * GLD];
  __shared__ uint32_t Lo[32 * 256];  // [word][thread]: bf16 pairs of the 64 compensation terms of each thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  if (!rp_gemm_tile(g, tm, tn)) return;
  const int m0 = tm * GT, n0 = tn * GT;
  if (rp_gemm_skip(g, m0, n0)) return;
  int kb = g.k0, ke = g.k1;
  if (g.part) {
    kb = g.k0 + (int)blockIdx.z * g.k_chunk;
    ke = min(g.k1, kb + g.k_chunk);
  }
  if (g.kb_m) kb = max(kb, g.m_off + m0);
  if (g.kb_n) kb = max(kb, g.n_off + n0);
  if (g.ke_n) ke = min(ke, g.n_off + n0 + GT);
  kb &= ~(GBK - 1);

  f32x16 acc[2][2], hi[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = hi[i][j][r] = 0.0f;
  bool have_lo = false;  // Lo holds this tile's compensation terms (uniform over the workgroup)
  // acc -> (hi, lo); `first`: there is no lo yet (nothing to read)
  auto flush = [&](bool first) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int w = ((i * 2 + j) * 16 + r) >> 1;
          float l0 = 0.0f, l1 = 0.0f;
          if (!first) {
            const uint32_t pk = Lo[w * 256 + tid];
            l0 = __uint_as_float(pk << 16);
            l1 = __uint_as_float(pk & 0xffff0000u);
          }
          const float a0 = acc[i][j][r], a1 = acc[i][j][r + 1];
          const float h0 = hi[i][j][r], h1 = hi[i][j][r + 1];
          const float s0 = h0 + a0, s1 = h1 + a1;
          const float b0 = s0 - h0, b1 = s1 - h1;
          l0 += (h0 - (s0 - b0)) + (a0 - b0);
          l1 += (h1 - (s1 - b1)) + (a1 - b1);
          hi[i][j][r] = s0;
          hi[i][j][r + 1] = s1;
          acc[i][j][r] = 0.0f;
          acc[i][j][r + 1] = 0.0f;
          Lo[w * 256 + tid] = rp_pack_bf16_rn(l0, l1);
        }
  };

  if (kb < ke) {
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const float* Ap = g.A + (long long)(m0 + lr) * g.lda + lk;
    const float* Bp = g.B + (long long)(n0 + lr) * g.ldb + lk;
    float4 ra0, ra1, rb0, rb1;
    auto gload = [&](int k) {
      ra0 = *reinterpret_cast<const float4*>(Ap + k);
      ra1 = *reinterpret_cast<const float4*>(Ap + 64 * g.lda + k);
      rb0 = *reinterpret_cast<const float4*>(Bp + k);
      rb1 = *reinterpret_cast<const float4*>(Bp + 64 * g.ldb + k);
    };
    auto lstore = [&](int buf) {
      *reinterpret_cast<float4*>(&As[buf][lr * GLD + lk]) = ra0;
      *reinterpret_cast<float4*>(&As[buf][(lr + 64) * GLD + lk]) = ra1;
      *reinterpret_cast<float4*>(&Bs[buf][lr * GLD + lk]) = rb0;
      *reinterpret_cast<float4*>(&Bs[buf][(lr + 64) * GLD + lk]) = rb1;
    };
    gload(kb);
    lstore(0);
    __syncthreads();
    const int arow = (wr * 64 + (lane & 31)) * GLD + (lane >> 5) * 4;
    const int brow = (wc * 64 + (lane & 31)) * GLD + (lane >> 5) * 4;
    int cur = 0;
    int since = 0;  // K steps in acc since the last flush
    for (int k = kb; k < ke; k += GBK) {
      const bool more = (k + GBK) < ke;
      if (more) gload(k + GBK);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // lanes 0-31 take k = 8j..8j+3, lanes 32-63 k = 8j+4..8j+7: one 16-byte read feeds four MFMA k-steps
        const float4 a0 = *reinterpret_cast<const float4*>(&As[cur][arow + j * 8]);
        const float4 a1 = *reinterpret_cast<const float4*>(&As[cur][arow + 32 * GLD + j * 8]);
        const float4 b0 = *reinterpret_cast<const float4*>(&Bs[cur][brow + j * 8]);
        const float4 b1 = *reinterpret_cast<const float4*>(&Bs[cur][brow + 32 * GLD + j * 8]);
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
        const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
        }
      }
      if (more) lstore(cur ^ 1);
      since += GBK;
      if (since >= g.kc && more) {  // (the last piece is flushed by the epilogue)
        flush(!have_lo);
        have_lo = true;
        since = 0;
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  // (C's old values all at once, then arithmetic and stores: see k_gemm_hp16w's epilogue)
  const bool rmw = !g.part && g.beta != 0.0f;
  float omax = 0.0f;
  f32x16 cold[2][2];
  if (rmw) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (g.tri == 3 ? 0 : n0) + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          cold[i][j][r] = g.C[(long long)row * g.ldc + col];
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (g.tri == 3 ? 0 : n0) + wc * 64 + j * 32 + (lane & 31);  // tri == 3: C is a stack of 128-wide blocks
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        // the sum as hi + lo: the last piece still sits in acc
        float h = acc[i][j][r], l = 0.0f;
        if (have_lo) {
          const int w = ((i * 2 + j) * 16 + r) >> 1;
          const uint32_t pk = Lo[w * 256 + tid];
          l = (r & 1) ? __uint_as_float(pk & 0xffff0000u) : __uint_as_float(pk << 16);
          const float a = h, hh = hi[i][j][r];
          const float s = hh + a, b = s - hh;
          l += (hh - (s - b)) + (a - b);
          h = s;
        }
        if (g.part) {
          g.part[((long long)blockIdx.z * g.M + row) * g.N + col] = h + l;
        } else {
          const float v = rmw ? fmaf(g.alpha, h, g.beta * cold[i][j][r]) : g.alpha * h;
          const float o = fmaf(g.alpha, l, v);
          g.C[(long long)row * g.ldc + col] = o;
          omax = fmaxf(omax, fabsf(o));
        }
      }
    }
  if (g.amax_out && !g.part) {  // the largest magnitude written: the scale of the products that will read these columns
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) omax = fmaxf(omax, __shfl_xor(omax, d));
    if (lane == 0 && omax > 0.0f) {
      const unsigned b = __float_as_uint(omax);
      atomicMax(g.amax_out, b);
      atomicMax(g.amax_out + 1, b);
      if (g.amax_out2) {
        atomicMax(g.amax_out2, b);
        atomicMax(g.amax_out2 + 1, b);
      }
    }
  }
}

// ---- the same product on the bf16 matrix pipe: three-way split operands ------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 retires 16 K-steps in 32 cycles per SIMD where the fp32-input MFMA above takes 64 cycles for 2: 16x the
// rate.  An fp32 operand is cut into three bf16 planes x = x1 + x2 + x3 (8 significant bits each, by truncation: x1 = the top 16
// bits of x, x2 = the top 16 bits of x - x1, x3 likewise - every subtraction is exact), and a product keeps the six plane
// products of weight >= 2^-16: a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1); the dropped ones are below 2^-24 of |a b|,
// the size of fp32's own rounding.  Six MFMAs per K16 step and output tile instead of eight fp32 ones per K16: 2.7x fewer
// matrix-pipe cycles.  Accumulation is fp32 in the MFMA with the same compensated master sums (hi + lo) as above.
// The planes are cut when a tile is staged (global fp32 -> registers -> three bf16 images in LDS, 5.5 VALU per element in the
// shadow of the other workgroup's MFMAs): memory holds fp32 only, 200 000 items still fit one GPU.
// LDS: a plane row is 16 bf16 + 8 of padding (48 B: the 16-B fragment reads of 16 consecutive rows then cover all 64 banks
// once), two buffers per operand and one barrier per step; the compensation terms lo sit in registers here (packed bf16
// pairs, 32 registers), which is what lets both buffers fit twice per CU (74 KB per workgroup).
// Measured (MI355X, 8192^3, fp32-equivalent 2 n^3 / t): 165 TF/s against 113 for the fp32 MFMA kernel, the same error to the
// digit (2.5e-8 .. 2.5e-7 of sum |a||b|).  That is 40 % of the 417 TF/s the six-product scheme could reach: a step writes 37 KB
// of planes to LDS per workgroup (ds_write_b128 sustains ~79 B/clk/CU) and reads 48 KB - the LDS pipe is as busy as the
// matrix pipe.  Variants measured and dropped: one buffer + two barriers with lo in LDS (162), the next tile's conversion
// hoisted in front of the MFMAs (145: 24 more live registers, no interleaving by the compiler), K32 steps in one buffer (168
// on the square product, but 7.6 s instead of 6.5 s for the C4 fit: the clipped K ranges of the triangular factors).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- two fp16 planes: three plane products ---------------------------------------------------------------------------------------
// fp16 carries 11 significant bits where bf16 carries 8: with round-to-nearest, x = h1 + h2 + e with h1 = fp16(x), h2 = fp16(x - h1)
// (the subtraction is exact) and |e| <= 2^-12 |h2| <= 2^-24 |x| - TWO planes hold an fp32 operand to fp32's own rounding, and a
// product keeps a1 b1 + (a1 b2 + a2 b1); the dropped a2 b2 is below 2^-24 |a b|.  Three MFMAs (v_mfma_f32_32x32x16_f16, the bf16
// rate) per K16 step and output tile instead of six, 4 bytes of LDS per element instead of 6, and a cheaper cut (two conversions
// and one subtraction per element).  What fp16 lacks is RANGE (2^-24 .. 65504), so each operand is multiplied by a power of two
// that puts its largest magnitude - over exactly the rows and K range this launch reads: k_gemm_amax, one pass over the operand
// beforehand - into [2^14, 2^15): entries down to 2^-18 of the largest keep both planes normal (fp32's precision), smaller ones
// are held to an ABSOLUTE 2^-40 of the largest (the second plane goes subnormal): a fixed-point tail that sum |a||b| over a row
// of comparable entries does not see - the factors and inverses of the EASE solve - but a row 2^-20 the size of the largest one
// would (a per-row scale would lift that; the solve has no such rows).  The scales leave the accumulators at
// the flush (acc * 1 / (sA sB), exact), the compensated master sums and the epilogue are those of the kernels above.
// (A register-side cut of bf16 planes - fp32 tile in LDS, each wave converting its own fragments, 2.3x less LDS traffic for 2x
// the conversions - was measured at 157 TF/s against 165 for k_gemm_nt_b3, and dropped.)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float rp_h2_scale(unsigned amax_bits) {  // power of two s with s * amax in [2^14, 2^15)
  const int E = (int)((amax_bits >> 23) & 0xffu);
  if (E == 0 || E == 255) return 1.0f;  // all zero (or subnormal), or not finite: nothing to protect
  const int se = min(max(268 - E, 67), 187);  // s = 2^(141 - E), kept within 2^+-60
  return __uint_as_float((unsigned)se << 23);
}

struct rp_amax_op {
  const float* P;
  long long ld;
  int tiles;    // 128-row tiles
  int kb_flag;  // the tile's K range begins at off + r0 (see rp_gemm_args::kb_m / kb_n)
  int ke_flag;  // ... ends at off + r0 + 128 (ke_n)
  int off;
  int skip0, skip1;  // row tiles with r0 in [skip0, skip1) are not measured (rp_gemm_args::n_skip0 / n_skip1: nothing reads them)
};

// out[0] = max |A|, out[1] = max |B| (fp32 bit patterns, zeroed by the caller) over the K ranges the product will read.
// One workgroup per (row tile, 256-column chunk): a wave reads one row's chunk (1 KB) per instruction.
__global__ void __launch_bounds__(256) k_gemm_amax(rp_amax_op a, rp_amax_op b, int k0, int k1, int dup, unsigned* out) {
  __shared__ float red[4];
  int t = blockIdx.x;
  const bool is_b = t >= a.tiles;
  const rp_amax_op o = is_b ? b : a;
  if (is_b) t -= a.tiles;
  const int r0 = t * GT;
  if (r0 >= o.skip0 && r0 < o.skip1) return;  // (uniform over the workgroup)
  int kb = k0, ke = k1;
  if (o.kb_flag) kb = max(kb, o.off + r0);
  if (o.ke_flag) ke = min(ke, o.off + r0 + GT);
  kb &= ~(GBK - 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = k0 + (int)blockIdx.y * 256 + lane * 4;
  float m = 0.0f;
  if (c >= kb && c < ke) {
    const float* p = o.P + (long long)r0 * o.ld + c;
#pragma unroll 8
    for (int r = wave; r < GT; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + (long long)r * o.ld);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  }
  // NaN never wins an fmaxf: a non-finite operand shows up as inf or not at all, and the product is NaN / inf either way
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.0f) {
      atomicMax(out + (is_b ? 1 : 0), __float_as_uint(m));
      if (dup) atomicMax(out + 1, __float_as_uint(m));
    }
  }
}

constexpr int HLD = 24;  // plane row stride in fp16 elements (48 B)

__global__ void __launch_bounds__(256, 2) k_gemm_nt_h2(rp_gemm_args g) {
  __shared__ __attribute__((aligned(16))) unsigned short As[2][2][GT * HLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][2][GT * HLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  if (!rp_gemm_tile(g, tm, tn)) return;
  const int m0 = tm * GT, n0 = tn * GT;
  if (rp_gemm_skip(g, m0, n0)) return;
  int kb = g.k0, ke = g.k1;
  if (g.part) {
    kb = g.k0 + (int)blockIdx.z * g.k_chunk;
    ke = min(g.k1, kb + g.k_chunk);
  }
  if (g.kb_m) kb = max(kb, g.m_off + m0);
  if (g.kb_n) kb = max(kb, g.n_off + n0);
  if (g.ke_n) ke = min(ke, g.n_off + n0 + GT);
  kb &= ~(GBK - 1);
  const float sa = rp_h2_scale(g.amax[0]), sb = rp_h2_scale(g.amax[1]);
  const float unscale = 1.0f / (sa * sb);  // a power of two

  f32x16 acc[2][2], hi[2][2];
  uint32_t lo[2][2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = hi[i][j][r] = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) lo[i][j][r] = 0u;
    }
  auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t pk = lo[i][j][r >> 1];
          float l0 = __uint_as_float(pk << 16), l1 = __uint_as_float(pk & 0xffff0000u);
          const float a0 = acc[i][j][r] * unscale, a1 = acc[i][j][r + 1] * unscale;
          const float h0 = hi[i][j][r], h1 = hi[i][j][r + 1];
          const float s0 = h0 + a0, s1 = h1 + a1;
          const float b0 = s0 - h0, b1 = s1 - h1;
          l0 += (h0 - (s0 - b0)) + (a0 - b0);
          l1 += (h1 - (s1 - b1)) + (a1 - b1);
          hi[i][j][r] = s0;
          hi[i][j][r + 1] = s1;
          acc[i][j][r] = 0.0f;
          acc[i][j][r + 1] = 0.0f;
          lo[i][j][r >> 1] = rp_pack_bf16_rn(l0, l1);
        }
  };

  if (kb < ke) {
    const int lr = tid >> 1, lk = (tid & 1) * 8;  // one row, eight consecutive K per thread and operand
    const float* Ap = g.A + (long long)(m0 + lr) * g.lda + lk;
    const float* Bp = g.B + (long long)(n0 + lr) * g.ldb + lk;
    float4 ra0, ra1, rb0, rb1;
    auto gload = [&](int k) {
      ra0 = *reinterpret_cast<const float4*>(Ap + k);
      ra1 = *reinterpret_cast<const float4*>(Ap + k + 4);
      rb0 = *reinterpret_cast<const float4*>(Bp + k);
      rb1 = *reinterpret_cast<const float4*>(Bp + k + 4);
    };
    // eight floats -> two planes of eight fp16, both rounded to nearest
    auto split_store = [&](const float4& u0, const float4& u1, float s, unsigned short (*dst)[GT * HLD]) {
      const float x[8] = {u0.x * s, u0.y * s, u0.z * s, u0.w * s, u1.x * s, u1.y * s, u1.z * s, u1.w * s};
      f16x8 h1, h2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const _Float16 t = (_Float16)x[e];
        h1[e] = t;
        h2[e] = (_Float16)(x[e] - (float)t);
      }
      *reinterpret_cast<f16x8*>(&dst[0][lr * HLD + lk]) = h1;
      *reinterpret_cast<f16x8*>(&dst[1][lr * HLD + lk]) = h2;
    };
    gload(kb);
    split_store(ra0, ra1, sa, As[0]);
    split_store(rb0, rb1, sb, Bs[0]);
    __syncthreads();
    // fragment of a 32-row block: lane (r = lane & 31, h = lane >> 5) holds row r, k = 8h .. 8h+7 (the bf16 map)
    const int arow = (wr * 64 + (lane & 31)) * HLD + (lane >> 5) * 8;
    const int brow = (wc * 64 + (lane & 31)) * HLD + (lane >> 5) * 8;
    int since = 0, cur = 0;
    for (int k = kb; k < ke; k += GBK) {
      const bool more = (k + GBK) < ke;
      if (more) gload(k + GBK);
      f16x8 a[2][2], b[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[p][i] = *reinterpret_cast<const f16x8*>(&As[cur][p][arow + i * 32 * HLD]);
          b[p][i] = *reinterpret_cast<const f16x8*>(&Bs[cur][p][brow + i * 32 * HLD]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];  // small products first
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
      if (more) {  // the next tile goes to the other buffer: one barrier per step
        split_store(ra0, ra1, sa, As[cur ^ 1]);
        split_store(rb0, rb1, sb, Bs[cur ^ 1]);
      }
      since += GBK;
      if (since >= g.kc && more) {
        flush();
        since = 0;
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  flush();  // the last piece
  // (C's old values all at once, into the dead accumulators, then arithmetic and stores: see k_gemm_hp16w's epilogue)
  const bool rmw = !g.part && g.beta != 0.0f;
  if (rmw) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (g.tri == 3 ? 0 : n0) + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          acc[i][j][r] = g.C[(long long)row * g.ldc + col];
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (g.tri == 3 ? 0 : n0) + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const uint32_t pk = lo[i][j][r >> 1];
        const float h = hi[i][j][r];
        const float l = (r & 1) ? __uint_as_float(pk & 0xffff0000u) : __uint_as_float(pk << 16);
        if (g.part) {
          g.part[((long long)blockIdx.z * g.M + row) * g.N + col] = h + l;
        } else {
          const float v = rmw ? fmaf(g.alpha, h, g.beta * acc[i][j][r]) : g.alpha * h;
          g.C[(long long)row * g.ldc + col] = fmaf(g.alpha, l, v);
        }
      }
    }
}

// ---- the large products: operands cut into planes ONCE, then a product kernel that only moves and multiplies -------------------
// In k_gemm_nt_h2 every workgroup cuts the tiles it stages, so an operand element is converted N / 128 times and the conversion
// (~100 VALU instructions per K16 step and wave beside 12 MFMAs) keeps the matrix pipe at 31 % busy.  For a product with many
// tiles the operands are instead cut once per launch by k_split_planes into a workspace, K chunk by K chunk, laid out in the
// order the product kernel consumes it: 1 KB pieces [plane][32-row block][K16 step][lane], lane l = row (l & 31), k = 8 (l >> 5)
// .. +7 - the operand map of v_mfma_f32_32x32x16_f16.  k_gemm_hp then stages a tile with LDS-DMA loads (global_load_lds_dwordx4:
// a wave moves one contiguous 1 KB piece straight into LDS, no registers, no ds_write, no arithmetic), reads fragments at
// lane * 16 (conflict-free by construction) and runs 24 MFMAs per K32 step and wave; two buffers, one barrier per step.
// A product whose K exceeds the workspace chunk is a sequence of launches, the later ones with beta = 1.
struct rp_split_args {
  const float* P;       // operand, row-major fp32
  long long ld;
  int rows;             // a multiple of 128
  int kc0, kc1;         // K chunk (multiples of 32)
  int kb_flag, ke_flag, off;  // the K range of a 128-row tile, as in rp_amax_op
  int k0, k1;           // the whole product's K range
  const unsigned* amax; // the operand's largest magnitude (one word)
  unsigned char* W;     // planes: W + p * plane_stride
  long long plane_stride;
  long long KSw;        // K16 steps per 32-row block in W (0: those of this chunk) and the K its first step stands for: a chunk
  int wk0;              // cut into a wider workspace (rp_plane_cache)
};

__global__ void __launch_bounds__(256) k_split_planes(rp_split_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rb = blockIdx.x;                      // 32-row block
  const int KS = (a.kc1 - a.kc0) >> 4;
  const int r0t = (rb * 32) & ~(GT - 1);          // its tile's first row
  int kb = a.k0, ke = a.k1;
  if (a.kb_flag) kb = max(kb, a.off + r0t);
  if (a.ke_flag) ke = min(ke, a.off + r0t + GT);
  const float sc = rp_h2_scale(*a.amax);
  const float* src = a.P + (long long)(rb * 32 + (lane & 31)) * a.ld + (lane >> 5) * 8;
  unsigned char* dst = a.W + (a.KSw ? (long long)rb * a.KSw + ((a.kc0 - a.wk0) >> 4) : (long long)rb * KS) * 1024 + lane * 16;
#pragma unroll 2
  for (int ks = blockIdx.y * 16 + wave; ks < min(KS, (int)blockIdx.y * 16 + 16); ks += 4) {
    const int k = a.kc0 + ks * 16;
    if (k + 16 <= kb || k >= ke) continue;  // never read by the product
    const float4 u0 = *reinterpret_cast<const float4*>(src + k);
    const float4 u1 = *reinterpret_cast<const float4*>(src + k + 4);
    const float x[8] = {u0.x * sc, u0.y * sc, u0.z * sc, u0.w * sc, u1.x * sc, u1.y * sc, u1.z * sc, u1.w * sc};
    f16x8 h1, h2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 t = (_Float16)x[e];
      h1[e] = t;
      h2[e] = (_Float16)(x[e] - (float)t);
    }
    *reinterpret_cast<f16x8*>(dst + (long long)ks * 1024) = h1;
    *reinterpret_cast<f16x8*>(dst + a.plane_stride + (long long)ks * 1024) = h2;
  }
}

struct rp_hp_args {
  rp_gemm_args g;       // k0 / k1 = this launch's K chunk; A / B unused
  const unsigned char* WA;
  const unsigned char* WB;
  long long psa, psb;   // plane strides
  int KS;               // K16 steps per 32-row block in the workspace
  int kc0;              // first K of the workspace
  int first;            // the product's first chunk: tiles with an empty K range still get beta * C
  int KSb, kc0b;        // k_gemm_hp16w: B's workspace geometry (A's planes may come from a cache with its own)
  const unsigned* sc_tab;  // k_gemm_hp16w: A's scale changes with the K chunk (k >> sc_shift): rp_plane_cache::amax
  int sc_shift, sc_mask;
  int b_tab;            // ... and B's is the same chunk's (B read from the same cache)
  int xb;               // k_gemm_hp16w: tile rows of an XCD's block of 32 tiles (0: plain 2-D grid)
};

// The same product on v_mfma_f32_16x16x32_f16 (same workspace, same LDS image: the 16-row operand of a K32 step is read from two
// K16 pieces).  The solve runs with the chip holding its clock down (1.49 GHz at 51 % matrix-pipe occupancy in the 32x32x16
// kernel); MI355X_MICROARCH.md reports a higher clock for this shape at equal cycles per flop.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) k_gemm_hp16(rp_hp_args q) {
  __shared__ __attribute__((aligned(1024))) unsigned char sm[2][2][2][4][2][1024];
  const rp_gemm_args& g = q.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  if (!rp_gemm_tile(g, tm, tn)) return;
  const int m0 = tm * GT, n0 = tn * GT;
  if (rp_gemm_skip(g, m0, n0)) return;
  int kb = g.k0, ke = g.k1;
  if (g.kb_m) kb = max(kb, g.m_off + m0);
  if (g.kb_n) kb = max(kb, g.n_off + n0);
  if (g.ke_n) ke = min(ke, g.n_off + n0 + GT);
  kb &= ~31;
  if (kb >= ke && !q.first) return;
  const float unscale = 1.0f / (rp_h2_scale(g.amax[0]) * rp_h2_scale(g.amax[1]));

  f32x4 acc[4][4], hi[4][4];
  uint32_t lo[4][4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = hi[i][j][r] = 0.0f;
      lo[i][j][0] = lo[i][j][1] = 0u;
    }
  auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const uint32_t pk = lo[i][j][r >> 1];
          float l0 = __uint_as_float(pk << 16), l1 = __uint_as_float(pk & 0xffff0000u);
          const float a0 = acc[i][j][r] * unscale, a1 = acc[i][j][r + 1] * unscale;
          const float h0 = hi[i][j][r], h1 = hi[i][j][r + 1];
          const float s0 = h0 + a0, s1 = h1 + a1;
          const float b0 = s0 - h0, b1 = s1 - h1;
          l0 += (h0 - (s0 - b0)) + (a0 - b0);
          l1 += (h1 - (s1 - b1)) + (a1 - b1);
          hi[i][j][r] = s0;
          hi[i][j][r + 1] = s1;
          acc[i][j][r] = 0.0f;
          acc[i][j][r + 1] = 0.0f;
          lo[i][j][r >> 1] = rp_pack_bf16_rn(l0, l1);
        }
  };

  if (kb < ke) {
    const unsigned char* asrc = q.WA + ((long long)(m0 / 32 + wave) * q.KS) * 1024 + lane * 16;
    const unsigned char* bsrc = q.WB + ((long long)(n0 / 32 + wave) * q.KS) * 1024 + lane * 16;
    auto issue = [&](int buf, int k) __attribute__((always_inline)) {
      const long long ko = (long long)((k - q.kc0) >> 4) * 1024;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + p * q.psa + ko + s2 * 1024),
                                           (__attribute__((address_space(3))) void*)&sm[buf][0][p][wave][s2][0], 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + p * q.psb + ko + s2 * 1024),
                                           (__attribute__((address_space(3))) void*)&sm[buf][1][p][wave][s2][0], 16, 0, 0);
        }
    };
    issue(0, kb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // 16x16x32 operand: lane (r = lane & 15, kg = lane >> 4) holds row r, k = 8 kg .. 8 kg + 7 of the K32 step: K16 piece kg >> 1,
    // half kg & 1; the 16-row block t of a 32-row piece sits at rows 16 (t & 1) + r of it
    const int foff = (lane >> 5) * 1024 + ((((lane >> 4) & 1) * 32) + (lane & 15)) * 16;
    int since = 0, cur = 0;
    for (int k = kb; k < ke; k += 32) {
      const bool more = (k + 32) < ke;
      if (more) issue(cur ^ 1, k + 32);
      f16x8 a[2][4], b[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a[p][t] = *reinterpret_cast<const f16x8*>(&sm[cur][0][p][wr * 2 + (t >> 1)][0][foff + (t & 1) * 256]);
          b[p][t] = *reinterpret_cast<const f16x8*>(&sm[cur][1][p][wc * 2 + (t >> 1)][0][foff + (t & 1) * 256]);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][i], b[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][i], b[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
      since += 32;
      if (since >= g.kc && more) {
        flush();
        since = 0;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }
  flush();
  // (C's old values all at once, into the dead accumulators, then arithmetic and stores: see k_gemm_hp16w's epilogue)
  float* const c0p = g.C + (long long)(m0 + wr * 64 + 4 * (lane >> 4)) * g.ldc + n0 + wc * 64 + (lane & 15);
  if (g.beta != 0.0f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = c0p[(long long)(i * 16 + r) * g.ldc + j * 16];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t pk = lo[i][j][r >> 1];
        const float h = hi[i][j][r];
        const float l = (r & 1) ? __uint_as_float(pk & 0xffff0000u) : __uint_as_float(pk << 16);
        const float v = (g.beta != 0.0f) ? fmaf(g.alpha, h, g.beta * acc[i][j][r]) : g.alpha * h;
        acc[i][j][r] = fmaf(g.alpha, l, v);
      }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) c0p[(long long)(i * 16 + r) * g.ldc + j * 16] = acc[i][j][r];
}

// The same product on 256 x 128 tiles with THREE LDS stages and counted waits (round 3).  PMC on k_gemm_hp16 showed the matrix pipe
// 56 % busy: every K32 step ended with s_waitcnt vmcnt(0) + barrier on a two-buffer ring, so a step's LDS-DMA pieces were requested
// one step ahead and waited for in full, and a wave issued 8 pieces per 48 MFMAs.  Here a workgroup of 8 waves (4 x 2, 64 x 64 each)
// owns 256 rows: 48 pieces per K32 step = 6 per wave for the same 48 MFMAs, the pieces of step k + 2 are requested while step k is
// multiplied (three stages of 48 KB = 144 of the 160 KB), and the top of a step waits with vmcnt(6) - for its own pieces of step k,
// two steps old - before the one barrier of the step (which also says that everybody has finished reading the stage about to be
// refilled).  The two 128-row halves of a tile follow the 128-tile rules separately (tri, K clipping: a half that is excluded or
// whose K range has not begun skips the MFMAs, wave-uniformly), so every caller's tile semantics are those of the 128 x 128 kernels.
#ifndef RP_FLUSH_FAST
#define RP_FLUSH_FAST 1  // 0: the six-operation two-sum of rounds 2 and 3
#endif
__global__ void __launch_bounds__(512, 1) k_gemm_hp16w(rp_hp_args q) {
  // [stage][A: plane 2 x block 8 x K16 2 | B: plane 2 x block 4 x K16 2][1 KB]
  extern __shared__ __attribute__((aligned(1024))) unsigned char smw[];
  constexpr int STAGE = 48 * 1024, AOFF = 0, BOFF = 32 * 1024;
  const rp_gemm_args& g = q.g;
  // (the wave index as a SCALAR: with tid >> 6 in a vector register every address, K bound and branch derived from it was vector
  // arithmetic - and vector instructions share the SIMD's issue port with the MFMAs: attribution by removal put the requests, the
  // fragment reads and the flush at 15 / 13 / 8 % of the kernel, ADDING UP, i.e. not hidden behind the matrix pipe)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;  // wave rows 0..3 (64 rows each), columns 0..1
  int tm = blockIdx.y, tn = blockIdx.x;
  if (q.xb) {
    // XCD-aware order (option gemm_xcd_block): workgroups are dealt to the 8 XCDs round-robin by their linear id and an XCD runs 32 of
    // them at a time (one per CU).  In the plain 2-D order those 32 are 2 column tiles of 16 tile rows (T = N11 L21^T) or 32 column
    // tiles of one row (N N^T): 17-33 operand panels per XCD and K step for 32 tiles.  Here the 32 consecutive workgroups of one
    // XCD form a block of xb tile rows x 32 / xb tile columns (4 x 8: 12 panels), blocks go to the XCDs round-robin in row-major
    // order - the XCDs walk the matrix side by side, top (longest K) first.
    const unsigned L = blockIdx.x, x = L & 7u, s8 = L >> 3, beta = s8 >> 5, pp = s8 & 31u;
    const unsigned R = (unsigned)q.xb, Cc = 32u / R;
    const unsigned gy = (unsigned)((g.M + 255) / 256), gx = (unsigned)(g.N / GT);
    const unsigned nbc = (gx + Cc - 1) / Cc;
    const unsigned gb = beta * 8u + x, g_row = gb / nbc, g_col = gb - g_row * nbc;
    tm = (int)(g_row * R + pp / Cc);
    tn = (int)(g_col * Cc + pp % Cc);
    if ((unsigned)tm >= gy || (unsigned)tn >= gx) return;
  }
  const int m0 = tm * 256, n0 = tn * GT;
  const bool has_lo = m0 + GT < g.M;  // the lower 128 rows exist
  // per half: excluded by tri?  K range
  bool act[2];
  int kbh[2], keh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int mh = m0 + h * GT;
    act[h] = (h == 0 || has_lo) && !rp_gemm_skip(g, mh, n0);
    int kb = g.k0, ke = g.k1;
    if (g.kb_m) kb = max(kb, g.m_off + mh);
    if (g.kb_n) kb = max(kb, g.n_off + n0);
    if (g.ke_n) ke = min(ke, g.n_off + n0 + GT);
    kbh[h] = kb & ~31;
    keh[h] = ke;
    if (kbh[h] >= keh[h] && !q.first) act[h] = false;  // nothing to add to C
  }
  if (!act[0] && !act[1]) return;
  const int kb = act[0] && act[1] ? min(kbh[0], kbh[1]) : (act[0] ? kbh[0] : kbh[1]);
  const int ke = act[0] && act[1] ? max(keh[0], keh[1]) : (act[0] ? keh[0] : keh[1]);
  const int my = wr >> 1;  // this wave's half
  const bool mine = act[my];
  const int my_kb = kbh[my], my_ke = keh[my];
  // the scales leave the accumulators at the flush.  With a plane cache A's scale belongs to the K chunk: the accumulators are
  // flushed at every chunk edge and the factor follows
  const float sb_fixed = q.b_tab ? 1.0f : rp_h2_scale(g.amax[1]);
  auto chunk_unscale = [&](int k) __attribute__((always_inline)) {
    const float sa = rp_h2_scale(q.sc_tab[k >> q.sc_shift]);
    return 1.0f / (sa * (q.b_tab ? sa : sb_fixed));
  };
  float unscale = q.sc_tab ? chunk_unscale(my_kb < my_ke ? my_kb : g.k0) : 1.0f / (rp_h2_scale(g.amax[0]) * sb_fixed);

  // Master sums hi + lo; lo as packed bf16 pairs (gfx950 converts two floats with ONE v_cvt_pk_bf16_f32, round to nearest even; the
  // integer rounding of the 128 x 128 kernels was 40 % of the flush's instructions - and an f32 lo spilled: 256 registers).  The
  // accumulators are never zeroed: after a flush the next step's first product of every accumulator takes the constant 0 as its C
  // operand (`fresh`).
  typedef __bf16 rp_bf2 __attribute__((ext_vector_type(2)));
  typedef float rp_f2 __attribute__((ext_vector_type(2)));
  f32x4 acc[4][4], hi[4][4];
  uint32_t lo[4][4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = hi[i][j][r] = 0.0f;
      lo[i][j][0] = lo[i][j][1] = 0u;
    }
  bool fresh = true;  // the accumulators hold nothing (uniform)
  // (round 5) The two waves of a SIMD (w and w + 4) at DIFFERENT PRIORITIES for the whole kernel: waves 4-7, which multiply first and
  // request the next pieces afterwards, always win the issue arbitration; waves 0-3 request first (in the shadow of their partner's
  // MFMAs) and multiply when the partner has gone on to its requests.  At equal priority the two interleaved instruction by instruction
  // and each request (60-180 cycles of a wave's stream) stood between MFMAs of both: 16384^3 364 -> 389 TFLOP/s fp32-equivalent, the
  // 100 000-item solve 3.09 -> 2.88 s on one box.  Measured beside it: the priority raised only around the MFMA rows (-1.3 %, waves
  // 4-7 only: -4 %), waves 0-3 instead of 4-7 (nothing), level 3 instead of 1 (the same), all waves requesting before or after
  // their MFMAs (-3.5 %), without the per-row sched_barrier (-6 % instead of -7).
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  auto flush = [&]() __attribute__((always_inline)) {
    const rp_f2 us2 = {unscale, unscale};
    // the bf16 pairs (1, 0) and (0, 1) from scalar registers: as an immediate the first becomes the inline constant "1.0", which
    // the instruction does not read as this pair (found as an error of 1e-5 in a K = 4096 product)
    uint32_t c10, c01;
    asm("s_mov_b32 %0, 0x3f80" : "=s"(c10));
    asm("s_mov_b32 %0, 0x3f800000" : "=s"(c01));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#if RP_FLUSH_FAST
        // (round 4) The flush is VALU time, ~80 cycles per pair of elements for two waves per SIMD, with nothing beside it (8 % of a
        // long-K product by removal; packed fp32 instructions take two passes per wave64 and buy nothing).  Fewer operations:
        // * the error of hi + p by Dekker's three operations (s = hi + p, e = p - (s - hi)) instead of Knuth's six.  Exact while
        //   |hi| >= |p|; when a block's partial sum p outgrows the running sum (cancellation, the first blocks) the term it misses
        //   is below ulp(p) - less than what the block's own 32 fp32 accumulation steps inside the matrix pipe have already put
        //   on p.  (s, e) never loses more than an uncompensated sum would.
        // * lo + e through v_dot2_f32_bf16 against the constant pairs (1, 0) and (0, 1): the packed bf16 halves are read where they
        //   stand - one instruction per element instead of shift / mask + add.
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const rp_f2 a2 = {acc[i][j][r], acc[i][j][r + 1]}, h2 = {hi[i][j][r], hi[i][j][r + 1]};
          const rp_f2 m1 = {-1.0f, -1.0f};
          const rp_f2 sv = __builtin_elementwise_fma(a2, us2, h2);
          const rp_f2 z = __builtin_elementwise_fma(h2, m1, sv);
          const rp_f2 e = __builtin_elementwise_fma(a2, us2, -z);
          hi[i][j][r] = sv[0];
          hi[i][j][r + 1] = sv[1];
#if RP_FLUSH_FAST == 2
          const uint32_t pk = lo[i][j][r >> 1];
          const rp_f2 l = {__uint_as_float(pk << 16) + e[0], __uint_as_float(pk & 0xffff0000u) + e[1]};
#else
          const rp_bf2 lp = __builtin_bit_cast(rp_bf2, lo[i][j][r >> 1]);
          const rp_f2 l = {__builtin_amdgcn_fdot2_f32_bf16(lp, __builtin_bit_cast(rp_bf2, c10), e[0], false),
                           __builtin_amdgcn_fdot2_f32_bf16(lp, __builtin_bit_cast(rp_bf2, c01), e[1], false)};
#endif
          const rp_bf2 pb = __builtin_convertvector(l, rp_bf2);
          lo[i][j][r >> 1] = __builtin_bit_cast(uint32_t, pb);
        }
#else
        const f32x4 av = acc[i][j] * unscale, h = hi[i][j];
        const f32x4 sv = h + av;
        const f32x4 bv = sv - h;
        const f32x4 dl = (h - (sv - bv)) + (av - bv);
        hi[i][j] = sv;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const uint32_t pk = lo[i][j][r >> 1];
          rp_f2 l = {__uint_as_float(pk << 16) + dl[r], __uint_as_float(pk & 0xffff0000u) + dl[r + 1]};
          const rp_bf2 pb = __builtin_convertvector(l, rp_bf2);
          lo[i][j][r >> 1] = __builtin_bit_cast(uint32_t, pb);
        }
#endif
      }
    fresh = true;
  };

  if (kb < ke) {
    // wave w stages A's 32-row block w (both planes, both K16 pieces: 4 pieces) and B's block w & 3, plane w >> 2 (2 pieces).
    // A block that lies beyond M (half tile) is replaced by the tile's first block: the count of requests per wave stays 6.
    const int ablk = (m0 / 32 + wave) * 32 < g.M ? wave : 0;
    const unsigned char* asrc = q.WA + ((long long)(m0 / 32 + ablk) * q.KS) * 1024 + lane * 16;
    const unsigned char* bsrc = q.WB + (long long)(wave >> 2) * q.psb + ((long long)(n0 / 32 + (wave & 3)) * q.KSb) * 1024 + lane * 16;
    auto issue = [&](int stage, int k) __attribute__((always_inline)) {
      const long long ko = (long long)((k - q.kc0) >> 4) * 1024, kob = (long long)((k - q.kc0b) >> 4) * 1024;
      unsigned char* sa = smw + stage * STAGE + AOFF + (wave * 2) * 1024;
      unsigned char* sb = smw + stage * STAGE + BOFF + (((wave >> 2) * 4 + (wave & 3)) * 2) * 1024;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + p * q.psa + ko + s2 * 1024),
                                           (__attribute__((address_space(3))) void*)(sa + p * 16 * 1024 + s2 * 1024), 16, 0, 0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + kob + s2 * 1024),
                                         (__attribute__((address_space(3))) void*)(sb + s2 * 1024), 16, 0, 0);
    };
    // a request past the K range re-reads the last step (harmless, keeps the counted waits uniform)
    const int k_last = kb + ((ke - kb - 1) / 32) * 32;
    issue(0, kb);
    issue(1, min(kb + 32, k_last));
    const int foff = (lane >> 5) * 1024 + ((((lane >> 4) & 1) * 32) + (lane & 15)) * 16;
    int since = 0, stage = 0;
    for (int k = kb; k < ke; k += 32) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // my pieces of this step have landed (those of the next step may be in flight)
      // everybody's have, and nobody still reads the stage refilled below (the fragment reads of the previous step were consumed by
      // its MFMAs).  The bare barrier: __syncthreads() carries a fence that the compiler turns into vmcnt(0) - the counted wait gone.
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int st2 = stage + 2;
      if (st2 >= 3) st2 -= 3;
      const int kn = min(k + 64, k_last);
      if (mine && k >= my_kb && k < my_ke) {
        const unsigned char* A0 = smw + stage * STAGE + AOFF;
        const unsigned char* B0 = smw + stage * STAGE + BOFF;
        // Issuing a piece holds a wave's instruction stream for 60 - 180 cycles (MI355X_MICROARCH.md): six of them cost as much as
        // the step's 48 MFMAs.  SQ counters of the first version (all eight waves request at the top of the step, or between their
        // MFMAs): matrix pipes 56 % busy, a wave parked a third of its time - the two waves of a SIMD (w and w + 4) ran the same
        // program in step and sat in their request phases together.  So the halves take turns: waves 0-3 request BEFORE their MFMAs,
        // waves 4-7 AFTER theirs - one partner's requests run beside the other's MFMAs.  (A row's A fragments are read while the row
        // before it multiplies.)
        if (wave < 4) issue(st2, kn);
        f16x8 a[2][2], b[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            b[p][t] = *reinterpret_cast<const f16x8*>(B0 + p * 8 * 1024 + ((wc * 2 + (t >> 1)) * 2) * 1024 + foff + (t & 1) * 256);
          a[p][0] = *reinterpret_cast<const f16x8*>(A0 + p * 16 * 1024 + ((wr * 2) * 2) * 1024 + foff);
        }
        // `fresh` is decided ONCE per step, outside the rows: with the branch inside, every row was a basic block of its own and
        // the compiler's wait-count pass, merging the two ways in, put s_waitcnt lgkmcnt(0) in front of each row's MFMAs - the
        // fragment reads issued for the NEXT row were waited for before the current row multiplied (found in the ISA listing)
        auto rows = [&](auto is_fresh) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < 3) {
#pragma unroll
              for (int p = 0; p < 2; ++p)
                a[p][(i + 1) & 1] = *reinterpret_cast<const f16x8*>(A0 + p * 16 * 1024 + ((wr * 2 + ((i + 1) >> 1)) * 2) * 1024 + foff + ((i + 1) & 1) * 256);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f32x4 c = decltype(is_fresh)::value ? (f32x4){0.0f, 0.0f, 0.0f, 0.0f} : acc[i][j];
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][i & 1], b[0][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][i & 1], b[1][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][i & 1], b[0][j], c, 0, 0, 0);
              acc[i][j] = c;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if (fresh)
          rows(std::true_type{});
        else
          rows(std::false_type{});
        fresh = false;
        if (wave >= 4) issue(st2, kn);
        since += 32;
        const bool edge = q.sc_tab && !((k + 32) & q.sc_mask);
        if ((since >= g.kc || edge) && k + 32 < my_ke) {
          flush();
          since = 0;
          if (edge) unscale = chunk_unscale(k + 32);
        }
      } else {
        issue(st2, kn);
      }
      stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the trailing re-reads)
  }
  if (!mine) return;
  if (!fresh) flush();
  // C in two phases: ALL of the tile's old values first (into the accumulators' registers, free now), then the arithmetic and the
  // stores.  One loop (load, fma, store per element) made every load wait for the store before it - the compiler cannot know that
  // the 64 addresses differ - i.e. 64 memory round trips in a row per thread: ~30 us per tile with nothing else on the CU, a fixed
  // 1.0 ms per launch at 16384^2 (tools/gemm_k_sweep.py) that cost a product with K = 2048 a fifth of its rate.
  float* c0p = g.C + (long long)(m0 + wr * 64 + 4 * (lane >> 4)) * g.ldc + n0 + wc * 64 + (lane & 15);
  long long ldc = g.ldc;
  if (g.tri == 4 && rp_gemm_ncol(g, n0) == g.m_off + m0 + my * GT) {
    // (round 5) a DIAGONAL tile of P = N N^T: the tiles of its row still read N's diagonal block, so it goes to the side stack
    // g.Cd [tile][128][128] (copied over the diagonal afterwards) - in the same launch as the rest, on this kernel, instead of a
    // second product on the 128 x 128 kernel (782 tiles with K up to n: 11 ms of a 100 000-item solve on their own)
    c0p = g.Cd + (long long)((g.m_off + m0) / GT + my) * GT * GT + (long long)((wr & 1) * 64 + 4 * (lane >> 4)) * GT + wc * 64 + (lane & 15);
    ldc = GT;
  }
  if (g.beta != 0.0f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = c0p[(long long)(i * 16 + r) * ldc + j * 16];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t pk = lo[i][j][r >> 1];
        const float h = hi[i][j][r];
        const float l = (r & 1) ? __uint_as_float(pk & 0xffff0000u) : __uint_as_float(pk << 16);
        const float v = (g.beta != 0.0f) ? fmaf(g.alpha, h, g.beta * acc[i][j][r]) : g.alpha * h;
        acc[i][j][r] = fmaf(g.alpha, l, v);
      }
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) c0p[(long long)(i * 16 + r) * ldc + j * 16] = acc[i][j][r];
}

// C = beta * C + alpha * sum_s part[s]  (fixed order of the partial sums: deterministic)
__global__ void k_splitk_reduce(const float* __restrict__ part, int S, int M, int N, float alpha, float beta, float* __restrict__ C,
                                long long ldc) {
  const long long total = (long long)M * N / 4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = (e * 4) / N;
    const int col = (int)((e * 4) % N);
    double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;  // float64: one rounding for the whole sum and the update of C
    for (int s2 = 0; s2 < S; ++s2) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((long long)s2 * M + row) * N + col);
      sx += v.x;
      sy += v.y;
      sz += v.z;
      sw += v.w;
    }
    float4* cp = reinterpret_cast<float4*>(C + row * ldc + col);
    const double al = alpha, be = beta;
    sx *= al;
    sy *= al;
    sz *= al;
    sw *= al;
    if (beta != 0.0f) {
      const float4 c = *cp;
      sx += be * c.x;
      sy += be * c.y;
      sz += be * c.z;
      sw += be * c.w;
    }
    *cp = make_float4((float)sx, (float)sy, (float)sz, (float)sw);
  }
}

// ---- diagonal block: Cholesky + inverse of a 128x128 block, float64 in LDS --------------------------------
constexpr int DS = GT + 1;  // LDS row stride in doubles

// Blocked in 16 columns: the first version walked the 128 columns one by one with four workgroup barriers each (and two more per
// column of the inverse): 768 barriers of 16 waves = the 0.3 ms the kernel took, a third of the Cholesky's per-block chain, which
// is what the second half of the factorisation waits for (the trailing updates there are shorter than the chain).  Now per block
// of 16 columns: the 16 x 16 diagonal block in the registers of one wave (shuffles, no barrier), the rows below by forward
// substitution (a thread per row), the rank-16 update of the rest in 4 x 4 register blocks: 3 barriers; the inverse by block rows
// with 2 barriers each.  40 barriers of 8 waves.
constexpr int PB = 16;                                    // block width inside the diagonal block
constexpr int PSTG = GT + 1;                              // row stride of the staging rows
constexpr int PSTC = GT + 2;                              // ... when they hold the Cholesky's current block column transposed (16-byte rows)
constexpr size_t RP_POTRF_LDS = ((size_t)GT * DS + (size_t)PB * PSTC + (size_t)PB * (PB + 1) + PB) * sizeof(double);
constexpr int RP_POTRF_THREADS = 512;

// wt != 0: the block of A receives W^T (upper triangular) instead of L: in the fused solve nothing reads a diagonal block of L again
// (the column below it is solved with W), and W^T is what N = L^-T holds there - one k_transpose_128 launch per block less.
// lane `src` of a wave's double, `src` a compile-time constant (the loops below are fully unrolled): two v_readlane_b32 instead of the two
// ds_bpermute_b32 of __shfl - the 16 x 16 diagonal steps are one wave's serial chain of 304 such broadcasts, each a trip through the LDS
// crossbar (round 6: the kernel is 60 % of the factorisation's chain where the chain is the bound, EXPERIMENTS e57)
__device__ __forceinline__ double rp_bcast_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ void __launch_bounds__(RP_POTRF_THREADS) k_potrf_trtri(float* __restrict__ A, long long lda, float* __restrict__ W, int* __restrict__ info,
                                                                  int wt = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* a = reinterpret_cast<double*>(smem_raw);  // [128][DS]
  double* stg = a + GT * DS;                         // [PB][PSTG] (the inverse) / [PB][PSTC] (the Cholesky's block column, k-major)
  double* wii = stg + PB * PSTC;                     // [PB][PB + 1]
  double* rdg = wii + PB * (PB + 1);                 // [PB]: reciprocals of the current block's diagonal
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NT = RP_POTRF_THREADS;
#pragma unroll 8
  for (int e = tid; e < GT * GT; e += NT) {  // (unrolled: eight of a thread's 32 loads in flight - one after the other they were 9 us)
    const int i = e / GT, k = e % GT;
    a[i * DS + k] = (k <= i) ? (double)A[(long long)i * lda + k] : 0.0;
  }
  __syncthreads();
  // ---- Cholesky, right-looking in blocks of PB columns
  for (int c0 = 0; c0 < GT; c0 += PB) {
    if (wave == 0) {  // the diagonal block: lane r (r < 16; the other lanes mirror them) holds row r
      const int r = lane & (PB - 1);
      double x[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) x[c] = a[(c0 + r) * DS + c0 + c];  // (zero above the diagonal)
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const double d = rp_bcast_f64(x[j], j);
        if (lane == 0 && !(d > 0.0)) atomicExch(info, c0 + j + 1);
        const double djj = sqrt(d > 0.0 ? d : 1.0), rj = 1.0 / djj;
        if (lane == 0) rdg[j] = rj;
        x[j] = (r == j) ? djj : x[j] * rj;
#pragma unroll
        for (int c = j + 1; c < PB; ++c) {
          const double lcj = rp_bcast_f64(x[j], c);
          if (r >= c) x[c] -= x[j] * lcj;
        }
      }
      if (lane < PB) {
#pragma unroll
        for (int c = 0; c < PB; ++c)
          if (c <= r) a[(c0 + r) * DS + c0 + c] = x[c];
      }
    }
    __syncthreads();
    const int nrow = GT - c0 - PB;
    if (tid < nrow) {  // the rows below: x L_dd^T = a[row][c0 : c0 + PB]
      const int row = c0 + PB + tid;
      double x[PB];
#pragma unroll
      for (int k = 0; k < PB; ++k) x[k] = a[row * DS + c0 + k];
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        double t = x[k];
#pragma unroll
        for (int m = 0; m < k; ++m) t -= x[m] * a[(c0 + k) * DS + c0 + m];
        x[k] = t * rdg[k];
      }
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        a[row * DS + c0 + k] = x[k];
        stg[k * PSTC + tid] = x[k];  // (round 6) ... and k-major beside it: what the update below reads
      }
    }
    __syncthreads();
    // the rest: a[i][j] -= sum_k a[i][c0 + k] a[j][c0 + k] for i >= j >= c0 + PB, in 4 x 4 blocks (bi >= bj)
    const int nq = nrow / 4;  // nrow is a multiple of 16
    for (int e = tid; e < nq * (nq + 1) / 2; e += NT) {
      int bi = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while ((bi + 1) * (bi + 2) / 2 <= e) ++bi;
      while (bi * (bi + 1) / 2 > e) --bi;
      const int bj = e - bi * (bi + 1) / 2;
      const int i0 = c0 + PB + 4 * bi, j0 = c0 + PB + 4 * bj;
      double acc[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
      // (round 6) the operands come from the k-major copy: a thread's four rows are 32 contiguous bytes there - two 16-byte reads -
      // where the row-major block had them 4 x 129 doubles apart from the next thread's: eight 8-byte reads per step that met in
      // the same banks eight ways (18 of the kernel's 113 us, tools/potrf_phases.hip).  Same products, same order: the same bits.
#pragma unroll 4
      for (int k = 0; k < PB; ++k) {
        double li[4], lj[4];
        const double2* pi = reinterpret_cast<const double2*>(stg + k * PSTC + 4 * bi);
        const double2* pj = reinterpret_cast<const double2*>(stg + k * PSTC + 4 * bj);
        const double2 i01 = pi[0], i23 = pi[1], j01 = pj[0], j23 = pj[1];
        li[0] = i01.x;
        li[1] = i01.y;
        li[2] = i23.x;
        li[3] = i23.y;
        lj[0] = j01.x;
        lj[1] = j01.y;
        lj[2] = j23.x;
        lj[3] = j23.y;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] += li[u] * lj[v];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (j0 + v <= i0 + u) a[(i0 + u) * DS + j0 + v] -= acc[u][v];
    }
    __syncthreads();
  }
  if (!wt) {
    for (int e = tid; e < GT * GT; e += NT) {
      const int i = e / GT, k = e % GT;
      A[(long long)i * lda + k] = (k <= i) ? (float)a[i * DS + k] : 0.0f;
    }
  }
  // ---- W = L^-1 in place, block row by block row: W[I][J] = -W_II sum_{K = J .. I-1} L[I][K] W[K][J] (rows above I already hold W)
  for (int r0 = 0; r0 < GT; r0 += PB) {
    if (wave == 0) {  // W_II: lane c computes column c of the inverse of the 16 x 16 triangle by forward substitution
      const int c = lane & (PB - 1);
      const double rc = 1.0 / a[(r0 + c) * DS + r0 + c];  // lane c: 1 / l_cc; row r's comes by shuffle
      double w[PB];
#pragma unroll
      for (int r = 0; r < PB; ++r) {
        double t = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < r; ++m) t -= a[(r0 + r) * DS + r0 + m] * w[m];
        w[r] = (r >= c) ? t * rp_bcast_f64(rc, r) : 0.0;
      }
      if (lane < PB) {
#pragma unroll
        for (int r = 0; r < PB; ++r) wii[r * (PB + 1) + c] = w[r];
      }
    } else {  // T = L[I][0 : r0] W[0 : r0][0 : r0] (16 x r0), the other waves
      // (round 6) a thread owns ONE column of T and FOUR of its rows, and a wave walks m in step from the smallest column it holds: the
      // entry of W is read once for four sums (consecutive columns over the lanes: no bank conflict), the four entries of L are the
      // same address in every lane (a broadcast).  Until then a thread owned one entry and started at its own column: two strided
      // reads per product, 46 of the kernel's 143 us (tools/potrf_phases.hip).  Every sum still adds its terms for m = col .. r0 - 1
      // in that order (the steps before a lane's column add 0): the same bits.
      for (int e0 = (wave - 1) * 64; e0 < 4 * r0; e0 += NT - 64) {
        const int e = e0 + lane;
        const bool on = e < 4 * r0;
        const int rg = on ? e / r0 : 0, col = on ? e - rg * r0 : r0;
        int m_lo = col;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m_lo = min(m_lo, __shfl_xor(m_lo, d));
        const double* lrow = a + (r0 + 4 * rg) * DS;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll 8
        for (int m = m_lo; m < r0; ++m) {  // (unrolled: the reads of eight steps are in flight before the first sum needs one)
          const double w = (on && m >= col) ? a[m * DS + col] : 0.0;
          t0 += lrow[m] * w;
          t1 += lrow[DS + m] * w;
          t2 += lrow[2 * DS + m] * w;
          t3 += lrow[3 * DS + m] * w;
        }
        if (on) {
          stg[(4 * rg + 0) * PSTG + col] = t0;
          stg[(4 * rg + 1) * PSTG + col] = t1;
          stg[(4 * rg + 2) * PSTG + col] = t2;
          stg[(4 * rg + 3) * PSTG + col] = t3;
        }
      }
    }
    __syncthreads();
    // W[I][0 : r0] = -W_II T and W[I][I] = W_II: a thread owns a column and four rows (round 6: four independent sums, every entry of T
    // read once for them; one entry per thread, several entries one after the other, was 10 of the kernel's 111 us)
    for (int e = tid; e < 4 * (r0 + PB); e += NT) {
      const int rg = e / (r0 + PB), col = e - rg * (r0 + PB);
      const double* wr = wii + 4 * rg * (PB + 1);
      double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
      if (col < r0) {
#pragma unroll
        for (int m = 0; m < PB; ++m) {  // (wii is zero above its diagonal)
          const double v = stg[m * PSTG + col];
          t0 -= wr[m] * v;
          t1 -= wr[(PB + 1) + m] * v;
          t2 -= wr[2 * (PB + 1) + m] * v;
          t3 -= wr[3 * (PB + 1) + m] * v;
        }
      } else {
        t0 = wr[col - r0];
        t1 = wr[(PB + 1) + col - r0];
        t2 = wr[2 * (PB + 1) + col - r0];
        t3 = wr[3 * (PB + 1) + col - r0];
      }
      a[(r0 + 4 * rg + 0) * DS + col] = t0;
      a[(r0 + 4 * rg + 1) * DS + col] = t1;
      a[(r0 + 4 * rg + 2) * DS + col] = t2;
      a[(r0 + 4 * rg + 3) * DS + col] = t3;
    }
    __syncthreads();
  }
  for (int e = tid; e < GT * GT; e += NT) {
    const int i = e / GT, k = e % GT;
    W[i * GT + k] = (k <= i) ? (float)a[i * DS + k] : 0.0f;
    if (wt) A[(long long)i * lda + k] = (k >= i) ? (float)a[k * DS + i] : 0.0f;
  }
}

// upper triangle of a symmetric matrix from its lower triangle, tile pair by tile pair
__global__ void __launch_bounds__(256) k_mirror_lower(float* __restrict__ P, long long ld, int nb) {
  __shared__ float t[32][33];
  // 32x32 sub-tiles; blockIdx.x enumerates (bi, bj) with bj <= bi over a (nb32 x nb32) grid
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) t[r][tx] = P[(long long)(bi * 32 + r) * ld + bj * 32 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bj * 32 + r, j = bi * 32 + tx;  // element (i, j) = P[j][i]
    if (j > i) P[(long long)i * ld + j] = t[tx][r];
  }
}

// ---- small helpers -------------------------------------------------------------------------------------------
// G = 0 everywhere, diag = lambda (real items) or 1 (padding)
// (a workgroup per row, 16-byte stores: the element-per-thread form with its 64-bit division per element wrote 40 GB in 17 ms)
// deg != NULL (round 5): the Gram kernel that follows writes all n columns of every row whose item has users (rp_gram_dense: a task
// per column tile for every row with a positive degree) - of those rows only the padding columns are cleared here; rows of items
// without users and the padding rows are written whole.  40 GB of stores at 100 000 items (6.4 ms) become a few megabytes.
__global__ void k_ease_init(float* __restrict__ G, long long ld, int n, int n_pad, float lam, const int32_t* __restrict__ deg) {
  for (long long i = blockIdx.x; i < n_pad; i += gridDim.x) {
    float* row = G + i * ld;
    const float d = i < n ? lam : 1.0f;
    const long long j_from = (deg && i < n && deg[i] > 0) ? ((long long)n & ~3LL) : 0;
    for (long long j = j_from + 4LL * threadIdx.x; j < ld; j += 4LL * blockDim.x) {  // (ld is a multiple of 4)
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i == j) v.x = d;
      if (i == j + 1) v.y = d;
      if (i == j + 2) v.z = d;
      if (i == j + 3) v.w = d;
      *reinterpret_cast<float4*>(row + j) = v;
    }
  }
}

__global__ void k_zero_f32(float* __restrict__ p, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) p[e] = 0.0f;
}

// diagonal block b of dst (leading dimension ld) = src[b] (128 x 128, leading dimension 128)
__global__ void __launch_bounds__(256) k_copy_diag_blocks(const float* __restrict__ src, float* __restrict__ dst, long long ld) {
  const long long b0 = (long long)blockIdx.x * GT;
  for (int e = threadIdx.x; e < GT * GT; e += 256) {
    const int i = e / GT, k = e % GT;
    dst[(b0 + i) * ld + b0 + k] = src[(b0 + i) * GT + k];
  }
}

// dst (n x n, leading dimension ldd) = src^T (leading dimension lds), where src is a block on the diagonal of the shared
// L / N matrix: its 128-tiles below the diagonal hold L and count as zero.  n a multiple of 32, block (32, 8), grid (n/32, n/32)
__global__ void k_transpose_block(const float* __restrict__ src, long long lds, float* __restrict__ dst, long long ldd) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const bool below = (by / GT) > (bx / GT);  // (32 divides 128: a 32 x 32 piece lies in one tile)
  for (int r = threadIdx.y; r < 32; r += 8) t[r][threadIdx.x] = below ? 0.0f : src[(long long)(by + r) * lds + bx + threadIdx.x];
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) dst[(long long)(bx + r) * ldd + by + threadIdx.x] = t[threadIdx.x][r];
}

__global__ void k_ease_diag(const float* __restrict__ P, long long ld, int n, const int32_t* __restrict__ deg, float alpha,
                            float* __restrict__ colscale) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const float d = P[(long long)j * ld + j];
    float w = 1.0f;
    if (alpha != 0.0f) w = deg[j] > 0 ? powf((float)deg[j], -alpha) : 0.0f;
    colscale[j] = -w / d;
  }
}

// K7: B_ij = P_ij * colscale_j (= -P_ij / P_jj * w_j), B_ii = 0; in place
// (a workgroup per row: no division per element, the column factors as 16-byte loads; colscale holds ld entries)
// (empty != NULL: rows of B without any stored value are counted on the way - what _check_fit_complete warns about; a pass of its
// own over the matrix before round 4)
__global__ void k_ease_scale(float* __restrict__ P, long long ld, int n, const float* __restrict__ colscale, int* __restrict__ empty) {
  for (long long i = blockIdx.x; i < n; i += gridDim.x) {
    float* row = P + i * ld;
    int nz = 0;
    for (long long j = 4LL * threadIdx.x; j < ld; j += 4LL * blockDim.x) {  // (ld is a multiple of 4)
      const float4 v = *reinterpret_cast<const float4*>(row + j);
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long long jj = j + c;
        o[c] = (jj < n && jj != i) ? o[c] * colscale[jj] : 0.0f;
        nz |= (o[c] != 0.0f) ? 1 : 0;
      }
      *reinterpret_cast<float4*>(row + j) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (empty) {  // (uniform)
      const int any = __syncthreads_or(nz);
      if (threadIdx.x == 0 && !any) atomicAdd(empty, 1);
    }
  }
}

// K7 straight from the LOWER half of P (round 4): B_ij = P_ij * colscale[j] below the diagonal and, from the same 32 x 32 tile,
// B_ji = P_ij * colscale[i] above it - the mirror of the inverse and the scaling in one pass (20 GB read + 40 GB written at 100 000
// items instead of 60 + 60).  Values on the diagonal and in the padding are zero; rownz[i] is set when row i keeps a stored value.
__global__ void __launch_bounds__(256) k_ease_scale_sym(float* __restrict__ P, long long ld, int n, const float* __restrict__ colscale,
                                                        int* __restrict__ rownz) {
  __shared__ float t[32][33];
  __shared__ int nz_lo[32], nz_up[32];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (threadIdx.x < 32) nz_lo[threadIdx.x] = nz_up[threadIdx.x] = 0;
  for (int r = ty; r < 32; r += 8) t[r][tx] = P[(long long)(bi * 32 + r) * ld + bj * 32 + tx];
  __syncthreads();
  // the tile itself: rows bi*32 + r, columns bj*32 + tx (on a diagonal tile only its lower part: the upper part comes from the mirror)
  const int jl = bj * 32 + tx;
  const float cl = jl < n ? colscale[jl] : 0.0f;
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r;
    if (bi == bj && tx > r) continue;
    const float v = (i < n && jl < n && i != jl) ? t[r][tx] * cl : 0.0f;
    P[(long long)i * ld + jl] = v;
    if (v != 0.0f) nz_lo[r] = 1;
  }
  // its mirror: rows bj*32 + r, columns bi*32 + tx, element (i, j) = P[j][i] = t[tx][r]
  const int ju = bi * 32 + tx;
  const float cu = ju < n ? colscale[ju] : 0.0f;
  for (int r = ty; r < 32; r += 8) {
    const int i = bj * 32 + r;
    if (ju <= i) continue;  // (strictly above the diagonal)
    const float v = (i < n && ju < n) ? t[tx][r] * cu : 0.0f;
    P[(long long)i * ld + ju] = v;
    if (v != 0.0f) nz_up[r] = 1;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    if (nz_lo[threadIdx.x] && bi * 32 + threadIdx.x < n) rownz[bi * 32 + threadIdx.x] = 1;
    if (nz_up[threadIdx.x] && bj * 32 + threadIdx.x < n) rownz[bj * 32 + threadIdx.x] = 1;
  }
}

__global__ void k_count_zero_flags(const int* __restrict__ f, int n, int* __restrict__ cnt) {
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += f[i] == 0 ? 1 : 0;
  if (c) atomicAdd(cnt, c);
}

// ---- host ------------------------------------------------------------------------------------------------------
// `st` = stream to launch on (the context's by default); `part_at` = where this product's split-K partials go
// (two products that run concurrently on different streams must not share them)
static bool rp_gemm_wide_ok(const rphip_ctx* ctx, const rp_gemm_args& g) {
  return ctx->gemm_wide && g.M >= 256 && !ctx->gemm_supertile;
}

static int rp_launch_hp16w(rphip_ctx* ctx, rp_hp_args& q, hipStream_t st) {
  constexpr int wide_lds = 3 * 48 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    RP_HIP(hipFuncSetAttribute((const void*)k_gemm_hp16w, hipFuncAttributeMaxDynamicSharedMemorySize, wide_lds));
    attr_set = true;
  }
  q.xb = 0;
  const long long gx = q.g.N / GT, gy = rp_cdiv(q.g.M, 256);
  if (ctx->gemm_xcd_block > 0 && q.g.tri != 3 && gx * gy >= 512) {
    int R = ctx->gemm_xcd_block;
    // (the default, 4, is a rule: the factorisation's trailing updates - lower-triangular products whose A and B planes are one
    // workspace - measured best on 16 x 2 blocks (969 against 977 ms of Cholesky phase), everything else on 4 x 8)
    if (R == 4 && q.g.tri == 1) R = 16;
    while (R > 1 && 32 / R < 2 && gx >= 2) R >>= 1;
    while (32 / R > gx && R < 32) R <<= 1;  // (narrow products: taller blocks)
    q.xb = R;
    const long long Cc = 32 / R, blocks = rp_cdiv(gy, R) * rp_cdiv(gx, Cc);
    hipLaunchKernelGGL(k_gemm_hp16w, dim3((unsigned)(rp_cdiv(blocks, 8) * 8 * 32)), dim3(512), wide_lds, st, q);
    return RPHIP_OK;
  }
  hipLaunchKernelGGL(k_gemm_hp16w, dim3((unsigned)gx, (unsigned)gy), dim3(512), wide_lds, st, q);
  return RPHIP_OK;
}

// A's planes (and B's, with g.cb) from a plane cache: nothing of A is measured or cut here; B is cut with one scale of its own
static int rp_gemm_nt_cached(rphip_ctx* ctx, const rp_gemm_args& g, hipStream_t st) {
  const rp_plane_cache& pc = *g.ca;
  if (!rp_gemm_wide_ok(ctx, g) || ctx->gemm_mode != 2 || g.tri == 3 || g.k0 < 0 || g.k1 > pc.K || g.M > pc.K || (g.cb && g.N > pc.K) ||
      (g.k0 % 32) || (g.k1 % 32) || (pc.w & (pc.w - 1)))
    RP_FAIL(RPHIP_EINVAL, "gemm_nt: this product cannot read its operand from the plane cache");
  rp_hp_args q{};
  q.g = g;
  q.g.kc = ctx->gemm_kc;
  q.g.sw = 0;
  q.g.part = nullptr;
  q.WA = pc.W;
  q.psa = pc.ps;
  q.KS = pc.K / 16;
  q.kc0 = 0;
  q.sc_tab = pc.amax;
  q.sc_shift = __builtin_ctz((unsigned)pc.w);
  q.sc_mask = pc.w - 1;
  q.b_tab = g.cb ? 1 : 0;
  if (g.cb) {
    q.WB = pc.W;
    q.psb = pc.ps;
    q.KSb = q.KS;
    q.kc0b = 0;
    q.first = 1;
    RP_TRY(rp_launch_hp16w(ctx, q, st));
    RP_HIP(hipGetLastError());
    return RPHIP_OK;
  }
  if (!ctx->gemm_amax_ring) RP_HIP(hipMalloc((void**)&ctx->gemm_amax_ring, (size_t)RP_AMAX_SLOTS * 2 * sizeof(unsigned)));
  unsigned* slot = ctx->gemm_amax_ring + 2 * (size_t)(ctx->gemm_amax_seq++ % RP_AMAX_SLOTS);
  RP_HIP(hipMemsetAsync(slot, 0, 2 * sizeof(unsigned), st));
  const int klen = g.k1 - g.k0;
  rp_amax_op oa{nullptr, 0, 0, 0, 0, 0};
  rp_amax_op ob{g.B, g.ldb, g.N / GT, g.kb_n, g.ke_n, g.n_off};
  hipLaunchKernelGGL(k_gemm_amax, dim3(ob.tiles, (unsigned)rp_cdiv(klen, 256)), dim3(256), 0, st, oa, ob, g.k0, g.k1, 0, slot);
  q.g.amax = slot;
  long long Kc = (long long)(ctx->gemm_ws_bytes / ((size_t)g.N * 4)) / 512 * 512;
  Kc = std::max<long long>(512, std::min<long long>(Kc, (klen + 31) / 32 * 32));
  const int slot_b = (g.ws_slot ? g.ws_slot : ((st == ctx->stream) ? 8 : 10)) + 1;
  unsigned char* WB = nullptr;
  size_t cap = (size_t)64 << 20;
  while (cap < (size_t)g.N * Kc * 4) cap <<= 1;
  RP_TRY(rp_scratch(ctx, slot_b, std::max<size_t>((size_t)g.N * Kc * 4, std::min<size_t>(cap, ctx->gemm_ws_bytes)), (void**)&WB));
  for (long long kc0 = g.k0; kc0 < g.k1; kc0 += Kc) {
    const int kc1 = (int)std::min<long long>(g.k1, kc0 + Kc);
    const int KS = (kc1 - (int)kc0) / 16;
    rp_split_args sb{g.B, g.ldb, g.N, (int)kc0, kc1, g.kb_n, g.ke_n, g.n_off, g.k0, g.k1, slot + 1, WB, (long long)g.N / 32 * KS * 1024};
    hipLaunchKernelGGL(k_split_planes, dim3(g.N / 32, (unsigned)rp_cdiv(KS, 16)), dim3(256), 0, st, sb);
    q.g.k0 = (int)kc0;
    q.g.k1 = kc1;
    q.g.beta = (kc0 == g.k0) ? g.beta : 1.0f;
    q.first = (kc0 == g.k0) ? 1 : 0;
    q.WB = WB;
    q.psb = sb.plane_stride;
    q.KSb = KS;
    q.kc0b = (int)kc0;
    RP_TRY(rp_launch_hp16w(ctx, q, st));
  }
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}

int rp_gemm_nt(rphip_ctx* ctx, const rp_gemm_args& g, hipStream_t st, float* part_at) {
  if (!st) st = ctx->stream;
  if (g.M <= 0 || g.N <= 0) return RPHIP_OK;
  if ((g.M % GT) || (g.N % GT) || (g.k0 % GBK) || (g.k1 % GBK)) RP_FAIL(RPHIP_EINVAL, "gemm_nt: sizes must be tile multiples");
  if (g.tri == 4 && !(g.ca && g.cb)) RP_FAIL(RPHIP_EINVAL, "gemm_nt: tri = 4 is the cached N N^T product's");
  if (g.ca) return rp_gemm_nt_cached(ctx, g, st);
  const long long tiles = (long long)(g.M / GT) * (g.N / GT);
  const int klen = g.k1 - g.k0;
  // few tiles and a long K: split K so that the launch fills the chip (partials in scratch, fixed-order reduce)
  int S = 1;
  // (thresholds 2-4 x CUs, targets 3-6 x CUs and minimum chunks of 512-2048 measured within noise of each other)
  if (!g.tri && !g.no_splitk && tiles < 2LL * ctx->n_cu && klen >= 4096 && (g.ldc % 4) == 0) {
    S = (int)std::min<long long>(8, std::min<long long>(klen / 2048, rp_cdiv(3LL * ctx->n_cu, tiles)));
    if (S < 2) S = 1;
  }
  rp_gemm_args h = g;
  h.kc = ctx->gemm_kc;
  dim3 grid(g.tri == 3 ? 1 : g.N / GT, g.M / GT, S);
  h.sw = 0;
  if (g.tri != 3 && ctx->gemm_supertile) {
    const int gx = g.N / GT, gy = g.M / GT;
    h.sw = gx >= 8 ? 8 : (gx >= 4 ? 4 : (gx >= 2 ? 2 : 1));
    const long long ns = rp_cdiv(gx, h.sw) * rp_cdiv(gy, 64 / h.sw);
    grid = dim3((unsigned)(rp_cdiv(ns, 8) * 512), 1, S);
  }
  if (S > 1) {
    float* part = part_at;
    if (!part) RP_TRY(rp_scratch(ctx, 7, (size_t)S * g.M * g.N * sizeof(float), (void**)&part));
    h.part = part;
    h.k_chunk = (int)(rp_cdiv(rp_cdiv(klen, S), GBK) * GBK);
  }
  const int mode = g.f32 ? 0 : ctx->gemm_mode;  // (g.f32: a short product of a launch-bound chain - one launch, no measuring pass)
  if (mode == 2) {
    // the operands' largest magnitudes over what this launch reads -> a slot of the context's ring (launches on different streams
    // overlap: each gets its own two words)
    if (!ctx->gemm_amax_ring) {
      RP_HIP(hipMalloc((void**)&ctx->gemm_amax_ring, (size_t)RP_AMAX_SLOTS * 2 * sizeof(unsigned)));
    }
    unsigned* slot = g.amax_known ? const_cast<unsigned*>(g.amax_known) : ctx->gemm_amax_ring + 2 * (size_t)(ctx->gemm_amax_seq++ % RP_AMAX_SLOTS);
    if (!g.amax_known) RP_HIP(hipMemsetAsync(slot, 0, 2 * sizeof(unsigned), st));
    const int m_tiles = g.M / GT, n_tiles = g.N / GT;
    rp_amax_op oa{g.A, g.lda, m_tiles, g.kb_m, 0, g.m_off, g.m_skip0, g.m_skip1};
    rp_amax_op ob{g.B, g.ldb, n_tiles, g.kb_n, g.ke_n, g.n_off, g.n_skip0, g.n_skip1};
    const int dup = (g.A == g.B && g.lda == g.ldb && m_tiles == n_tiles && g.kb_m == g.kb_n && !g.ke_n && (!g.kb_m || g.m_off == g.n_off)) ? 1 : 0;
    // many tiles and a K worth a pass of its own: planes cut once into the workspace, then the LDS-DMA product (see k_gemm_hp)
    const bool presplit = ctx->gemm_presplit && S == 1 && g.tri != 3 && tiles >= 512 && klen >= 1024 && !(g.k0 % 32) && !(g.k1 % 32);
    // B's rows may be a sub-range of A's (the next panel's columns of the trailing update): then one set of planes - and one scale:
    // max |B| <= max |A| - serves both
    const bool same_cols = g.lda == g.ldb && !g.kb_m && !g.kb_n && !g.ke_n && g.B >= g.A && (g.B - g.A) % g.lda == 0;
    const long long row_shift = same_cols ? (g.B - g.A) / g.lda : -1;
    const bool alias = presplit && (dup || (same_cols && !(row_shift % 32) && row_shift + g.N <= g.M));
    const int one_max = (dup || alias) ? 1 : 0;
    if (one_max) ob.tiles = 0;
    if (!g.amax_known)
      hipLaunchKernelGGL(k_gemm_amax, dim3(oa.tiles + ob.tiles, (unsigned)rp_cdiv(klen, 256)), dim3(256), 0, st, oa, ob, g.k0, g.k1, one_max, slot);
    h.amax = slot;
    if (presplit) {
      const int rows_max = std::max(g.M, alias ? 0 : g.N);
      long long Kc = (long long)(ctx->gemm_ws_bytes / ((size_t)rows_max * 4)) / 512 * 512;
      Kc = std::max<long long>(512, std::min<long long>(Kc, (klen + 31) / 32 * 32));
      const int slot_a = g.ws_slot ? g.ws_slot : ((st == ctx->stream) ? 8 : 10);  // a product on a side stream runs beside one on the main stream
      unsigned char *WA = nullptr, *WB = nullptr;
      auto ws_size = [&](size_t need) {  // grow in powers of two: the products of a solve come in many sizes
        size_t c = (size_t)64 << 20;
        while (c < need) c <<= 1;
        return std::max(need, std::min(c, ctx->gemm_ws_bytes));
      };
      RP_TRY(rp_scratch(ctx, slot_a, ws_size((size_t)g.M * Kc * 4), (void**)&WA));
      if (!alias) RP_TRY(rp_scratch(ctx, slot_a + 1, ws_size((size_t)g.N * Kc * 4), (void**)&WB));
      for (long long kc0 = g.k0; kc0 < g.k1; kc0 += Kc) {
        const int kc1 = (int)std::min<long long>(g.k1, kc0 + Kc);
        const int KS = (kc1 - (int)kc0) / 16;
        rp_split_args sa{g.A, g.lda, g.M, (int)kc0, kc1, g.kb_m, 0, g.m_off, g.k0, g.k1, slot, WA, (long long)g.M / 32 * KS * 1024};
        hipLaunchKernelGGL(k_split_planes, dim3(g.M / 32, (unsigned)rp_cdiv(KS, 16)), dim3(256), 0, st, sa);
        rp_hp_args q{};
        q.g = h;
        q.g.k0 = (int)kc0;
        q.g.k1 = kc1;
        q.g.beta = (kc0 == g.k0) ? g.beta : 1.0f;
        q.first = (kc0 == g.k0) ? 1 : 0;
        q.WA = WA;
        q.psa = sa.plane_stride;
        q.KS = q.KSb = KS;
        q.kc0 = q.kc0b = (int)kc0;
        if (alias) {
          q.WB = WA + (dup ? 0 : row_shift / 32 * KS * 1024);
          q.psb = q.psa;
        } else {
          rp_split_args sb{g.B, g.ldb, g.N, (int)kc0, kc1, g.kb_n, g.ke_n, g.n_off, g.k0, g.k1, slot + 1, WB, (long long)g.N / 32 * KS * 1024};
          hipLaunchKernelGGL(k_split_planes, dim3(g.N / 32, (unsigned)rp_cdiv(KS, 16)), dim3(256), 0, st, sb);
          q.WB = WB;
          q.psb = sb.plane_stride;
        }
        if (rp_gemm_wide_ok(ctx, g)) {
          RP_TRY(rp_launch_hp16w(ctx, q, st));
        } else
          hipLaunchKernelGGL(k_gemm_hp16, grid, dim3(256), 0, st, q);
      }
      RP_HIP(hipGetLastError());
      return RPHIP_OK;
    }
    hipLaunchKernelGGL(k_gemm_nt_h2, grid, dim3(256), 0, st, h);
  } else {
    hipLaunchKernelGGL(k_gemm_nt, grid, dim3(256), 0, st, h);
  }
  RP_HIP(hipGetLastError());
  if (S == 1) return RPHIP_OK;
  const long long quads = (long long)g.M * g.N / 4;
  hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)std::min<long long>(4096, rp_cdiv(quads, 256))), dim3(256), 0, st, h.part, S, g.M,
                     g.N, g.alpha, g.beta, g.C, g.ldc);
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}


extern "C" int rphip_gemm_nt_f32(rphip_ctx* ctx, int64_t M, int64_t N, int64_t K, float alpha, const float* A_dev, int64_t lda,
                                 const float* B_dev, int64_t ldb, float beta, float* C_dev, int64_t ldc) {
  RP_ENTER(ctx);
  if (!ctx || !A_dev || !B_dev || !C_dev) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if ((M % GT) || (N % GT) || (K % GBK) || (lda % 4) || (ldb % 4)) RP_FAIL(RPHIP_EINVAL, "M, N must be multiples of 128, K of 16, lda/ldb of 4");
  RP_TRY(rp_set_device(ctx));
  rp_gemm_args g{};
  g.A = A_dev;
  g.lda = lda;
  g.B = B_dev;
  g.ldb = ldb;
  g.C = C_dev;
  g.ldc = ldc;
  g.M = (int)M;
  g.N = (int)N;
  g.k0 = 0;
  g.k1 = (int)K;
  g.alpha = alpha;
  g.beta = beta;
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  RP_TRY(rp_gemm_nt(ctx, g));
  rp_time_end(ctx, ev, RPHIP_T_CHOLESKY, 2.0 * (double)M * (double)N * (double)K);
  return RPHIP_OK;
}

// In-place inverse of the SPD matrix A (n_pad x n_pad, ld), result in A (full).  ONE n x n matrix: after the Cholesky the
// factor L sits strictly below the diagonal; N = L^-T is built on and above it (a diagonal block of L is only needed as its
// inverse W_k, kept aside); P = N N^T then overwrites L below the diagonal (it reads N only), the diagonal tiles go through a
// side buffer, and the lower half is mirrored over N.  200 000 items (160 GB) fit one 288 GB GPU this way.
// tiles per panel of N = L^-T: ctx->inv_panel (default 16 = 2048 columns; 8 / 16 / 32 tiles measured in round 1: 5.94 / 5.53 / 5.87 s at n = 100 000)

int rp_spd_inverse(rphip_ctx* ctx, float* A, int n_pad, long long ld, int* info_ext, hipStream_t on, bool lower_only) {
  hipStream_t st = on ? on : ctx->stream;  // (every launch below names its stream: the solve may run beside the caller's other streams)
  const bool timed = st == ctx->stream;
  const int nb = n_pad / GT;
  char* wsb = nullptr;
  const int RP_INV_PANEL = ctx->inv_panel;
  const int inv_w = std::min(RP_INV_PANEL * GT, n_pad);  // columns of one panel of the inverse
  const size_t cmax_off = (size_t)nb * GT * GT * 4 + ((size_t)n_pad * inv_w + (size_t)inv_w * inv_w + (size_t)inv_w * GT) * 4 + 256;
  RP_TRY(rp_scratch(ctx, 5, cmax_off + (size_t)(2 * nb + 8) * 2 * sizeof(unsigned), (void**)&wsb));
  float* const Nb = A;  // N shares the matrix with L
  int* info = info_ext ? info_ext : (int*)wsb;
  float* W = (float*)(wsb + 256);                 // [nb][128][128]
  float* T = W + (size_t)nb * GT * GT;            // [n_pad][inv_w]: N11 L21^T of a panel ([c0][128] inside a panel)
  float* NT = T + (size_t)n_pad * inv_w;          // [inv_w][inv_w]: transpose of the panel's diagonal block of N
  float* T2 = NT + (size_t)inv_w * inv_w;         // [inv_w][128]: the side stream's own T for the diagonal blocks of N
  float* part_all = nullptr;  // the split-K partials of the skinny products: sized once for the tallest of them
  RP_TRY(rp_scratch(ctx, 7, (size_t)8 * n_pad * GT * sizeof(float), (void**)&part_all));
  if (!info_ext) RP_HIP(hipMemsetAsync(info, 0, 4, st));
  // Scales of the factorisation's products without measuring passes (round 4): the product that writes a block column of L
  // (L[k+1:, k] = A W_k^T, fp32 kernel) leaves the column's largest magnitude in two pairs of words - its sub-panel's and its
  // panel's - and every product that reads L (the block chain's two skinny updates, the sub-panel and the trailing updates) takes
  // its operands' scale from there: a bound over all rows of the columns it reads.  k_gemm_amax + its memset left the per-block
  // chain (two of each per block: 3 100 launches and 0.5 s of kernel time per 100 000-item solve in round 3).
  unsigned* cmax_sub = (unsigned*)(wsb + cmax_off);  // [nb + 4][2]: by sub-panel index
  unsigned* cmax_pan = cmax_sub + (size_t)(nb + 4) * 2;  // [nb + 4][2]: by panel index
  RP_HIP(hipMemsetAsync(cmax_sub, 0, (size_t)(2 * nb + 8) * 2 * sizeof(unsigned), st));
  const bool known_scales = ctx->gemm_mode == 2 && rp_diag_env("RP_CHOL_MEASURE_SCALES") == nullptr;
  const size_t diag_lds = RP_POTRF_LDS;
  RP_HIP(hipFuncSetAttribute((const void*)k_potrf_trtri, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
  const double n3 = (double)n_pad * n_pad * n_pad;
  hipEvent_t ev;
  // ---- K5: left-looking blocked Cholesky.  The 128x128 diagonal block (its update and its float64 factorisation, one
  // workgroup for 0.26 ms) runs on a side stream while the rest of the panel is updated on the main stream.
  struct side_stream {  // released on every way out of this function
    hipStream_t s = nullptr, t = nullptr, u = nullptr;
    hipEvent_t a = nullptr, b = nullptr, c = nullptr, d = nullptr, f = nullptr, g = nullptr;
    ~side_stream() {
      if (s) hipStreamSynchronize(s);
      if (t) hipStreamSynchronize(t);
      if (u) hipStreamSynchronize(u);
      if (g) hipEventDestroy(g);
      if (u) hipStreamDestroy(u);
      if (f) hipEventDestroy(f);
      if (a) hipEventDestroy(a);
      if (b) hipEventDestroy(b);
      if (c) hipEventDestroy(c);
      if (d) hipEventDestroy(d);
      if (s) hipStreamDestroy(s);
      if (t) hipStreamDestroy(t);
    }
  } side;
  RP_HIP(hipStreamCreateWithFlags(&side.s, hipStreamNonBlocking));
  RP_HIP(hipStreamCreateWithFlags(&side.t, hipStreamNonBlocking));
  RP_HIP(hipEventCreateWithFlags(&side.a, hipEventDisableTiming));
  RP_HIP(hipEventCreateWithFlags(&side.b, hipEventDisableTiming));
  RP_HIP(hipEventCreateWithFlags(&side.c, hipEventDisableTiming));
  RP_HIP(hipEventCreateWithFlags(&side.d, hipEventDisableTiming));
  hipStream_t s2 = side.s, s3 = side.t;
  RP_HIP(hipEventCreateWithFlags(&side.f, hipEventDisableTiming));
  RP_HIP(hipStreamCreateWithFlags(&side.u, hipStreamNonBlocking));
  RP_HIP(hipEventCreateWithFlags(&side.g, hipEventDisableTiming));
  hipStream_t s4 = side.u;
  hipEvent_t e_rows = side.g;
  hipEvent_t e_panel = side.a, e_diag = side.b, e_fact = side.c, e_trail = side.d, e_cols = side.f;
  bool cols_pending = false;  // the later sub-panels' columns of the current panel are still being updated on s3
  bool trail_pending = false;  // a trailing update is in flight on s3
  // diagnostic (RP_CHOL_TRACE=1): per panel, ms since the start at which the panel was factored and its next-panel update done (main
  // stream) and its rest update done (third stream) - shows where the factorisation is chain-bound and where product-bound
  std::vector<hipEvent_t> tr_ev;
  const bool trace = rp_diag_env("RP_CHOL_TRACE") != nullptr;
  auto tr_rec = [&](hipStream_t s) {
    if (!trace) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    hipEventRecord(e, s);
    tr_ev.push_back(e);
  };
  tr_rec(st);
  float* part_diag = part_all + (size_t)8 * (n_pad - GT) * GT;  // the side stream's own split-K partials
  // The diagonal blocks N22 of the inverse's panels (K6a below) are chains of ~30 small launches that depend on nothing but W and L
  // inside the panel's own diagonal block.  They run AHEAD of the inverse's wide products on the side stream (issued in line they were
  // 13 % of the inverse at 60 000 items with the chip idle) - except those of the first `inv_n22_early` panels (round 6), which the
  // inverse's first products would wait for: they are issued during the factorisation, on a stream of their own, as soon as it has
  // passed the panel's last block column (they write tiles above the diagonal, which the factorisation never touches).  At 100 000
  // items: fit -6 ms with 4 panels; with ALL panels early the factorisation loses 40 ms and the inverse gains 16 (EXPERIMENTS e55).
  struct ev_list {
    std::vector<hipEvent_t> v;
    ~ev_list() {
      for (auto e : v) hipEventDestroy(e);
    }
  } n22_ev;
  const int n_inv_panels = (int)rp_cdiv(nb, RP_INV_PANEL);
  n22_ev.v.resize(n_inv_panels, nullptr);
  for (auto& e : n22_ev.v) RP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const int n22_f32 = rp_diag_env("RP_N22_PLANES") == nullptr ? 1 : 0;
  const bool n22_early = n22_f32 == 1 && ctx->inv_n22_early > 0;  // (the plane path shares workspaces with the factorisation's products)
  int n22_issued = 0;
  auto n22 = [&](int P, hipStream_t sn) -> int {
    const int pb = P * RP_INV_PANEL, pe = std::min(nb, pb + RP_INV_PANEL);
    const int j = pb * GT;
    for (int i = pb; i < pe; ++i) {
      const int c0 = i * GT;
      const float* Wi = W + (size_t)i * GT * GT;
      // (N's diagonal block W_i^T is where the factorisation left it: k_potrf_trtri, wt)
      if (i > pb) {
        rp_gemm_args g{};
        g.A = Nb + (long long)j * ld;
        g.lda = ld;
        g.B = A + (long long)c0 * ld;
        g.ldb = ld;
        g.C = T2;
        g.ldc = GT;
        g.M = c0 - j;
        g.N = GT;
        g.k0 = j;
        g.k1 = c0;
        g.kb_m = 1;  // row block r of N starts at column r
        g.m_off = j;
        g.alpha = 1.0f;
        g.beta = 0.0f;
        g.no_splitk = 1;
        g.f32 = n22_f32;  // (round 4: these tiny products - at most 15 x 1 tiles - on the fp32 kernel: one launch each, no measuring pass, no memset)
        RP_TRY(rp_gemm_nt(ctx, g, sn));
        rp_gemm_args h{};
        h.f32 = n22_f32;
        h.A = T2;
        h.lda = GT;
        h.B = Wi;
        h.ldb = GT;
        h.C = Nb + (long long)j * ld + c0;
        h.ldc = ld;
        h.M = c0 - j;
        h.N = GT;
        h.k0 = 0;
        h.k1 = GT;
        h.alpha = -1.0f;
        h.beta = 0.0f;
        h.no_splitk = 1;
        RP_TRY(rp_gemm_nt(ctx, h, sn));
      }
    }
    RP_HIP(hipEventRecord(n22_ev.v[P], sn));
    return RPHIP_OK;
  };
  if (timed) rp_time_begin(ctx, &ev);
  int rc = RPHIP_OK;
  // Blocked in two levels: wide panels of 16 or 32 tiles are factored left-looking with K restricted to the panel
  // (short, skinny products: ~n^2 * panel / 2 flop in total), then the whole trailing matrix takes the panel's
  // rank-(128 * panel) update in ONE lower-triangular product at the square-GEMM rate.  Purely left-looking (K
  // from 0 for every 128-column block) made every product of the factorisation skinny: 92 TFLOP/s at n = 100 000.
  // (measured on MI355X: panels of 8 / 16 / 32 tiles give 91 / 96 / 89 TFLOP/s at n = 50 000 and 105 / 114 / 116 at 100 000)
  const int RP_CHOL_PANEL = ctx->chol_panel ? ctx->chol_panel : (nb >= 640 ? 32 : 16);  // tiles per panel
  // ... and a level between: sub-panels of `chol_sub` tiles.  The left-looking products of a block (one tile column wide: ~100
  // TFLOP/s) only reach back to the sub-panel's first column; a finished sub-panel updates the REST OF ITS PANEL's columns in one
  // wide product (K = 128 * chol_sub).  3/4 of the in-panel flop moves from the skinny kernel to the wide one.
  const int SUB = (ctx->chol_sub > 0 && RP_CHOL_PANEL % ctx->chol_sub == 0) ? ctx->chol_sub : RP_CHOL_PANEL;
  for (int k = 0; k < nb && rc == RPHIP_OK; ++k) {
    const int r0 = k * GT;
    const int pk0 = (k / RP_CHOL_PANEL) * RP_CHOL_PANEL * GT;  // first column of this block's panel
    const int sk0 = (k / SUB) * SUB * GT;                      // ... of its sub-panel
    RP_HIP(hipEventRecord(e_panel, st));  // block k-1 (and everything before) is complete
    RP_HIP(hipStreamWaitEvent(s2, e_panel, 0));
    rp_gemm_args g{};
    g.A = A + (long long)r0 * ld;
    g.lda = ld;
    g.B = A + (long long)r0 * ld;
    g.ldb = ld;
    g.C = A + (long long)r0 * ld + r0;
    g.ldc = ld;
    g.N = GT;
    g.k0 = sk0;
    g.k1 = r0;
    g.alpha = -1.0f;
    g.beta = 1.0f;
    if (known_scales) g.amax_known = cmax_sub + 2 * (size_t)(k / SUB);
    if (r0 > sk0) {  // A[k,k] -= L[k,p:k] L[k,p:k]^T on the side stream (p = the sub-panel's first column)
      g.M = GT;
      rc = rp_gemm_nt(ctx, g, s2, part_diag);
      if (rc != RPHIP_OK) break;
    }
    hipLaunchKernelGGL(k_potrf_trtri, dim3(1), dim3(RP_POTRF_THREADS), diag_lds, s2, A + (long long)r0 * ld + r0, ld, W + (size_t)k * GT * GT, info, 1);
    RP_HIP(hipEventRecord(e_diag, s2));
    if (r0 + GT < n_pad) {
      if (r0 > sk0) {  // A[k+1:,k] -= L[k+1:,p:k] L[k,p:k]^T
        g.A = A + (long long)(r0 + GT) * ld;
        g.C = A + (long long)(r0 + GT) * ld + r0;
        g.M = n_pad - r0 - GT;
        rc = rp_gemm_nt(ctx, g, st, part_all);
        if (rc != RPHIP_OK) break;
      }
      RP_HIP(hipStreamWaitEvent(st, e_diag, 0));
      rp_gemm_args h{};  // L[k+1:,k] = A[k+1:,k] W_k^T
      h.A = A + (long long)(r0 + GT) * ld + r0;
      h.lda = ld;
      h.B = W + (size_t)k * GT * GT;
      h.ldb = GT;
      h.C = A + (long long)(r0 + GT) * ld + r0;
      h.ldc = ld;
      h.M = n_pad - r0 - GT;
      h.N = GT;
      h.k0 = 0;
      h.k1 = GT;
      h.alpha = 1.0f;
      h.beta = 0.0f;
      h.f32 = 1;  // (K = 128 on the per-block chain: the fp32 MFMA kernel in ONE launch instead of memset + amax + plane kernel)
      if (known_scales) {
        h.amax_out = cmax_sub + 2 * (size_t)(k / SUB);
        h.amax_out2 = cmax_pan + 2 * (size_t)(k / RP_CHOL_PANEL);
      }
      rc = rp_gemm_nt(ctx, h, st);
      if (rc != RPHIP_OK) break;
      const int r1 = r0 + GT;
      if ((k + 1) % SUB == 0 && (k + 1) % RP_CHOL_PANEL != 0 && r1 < n_pad) {
        // the sub-panel is done: the panel's remaining columns take its rank-(128 SUB) update, A[r1:, r1:pe] -= L[r1:, s:r1] L[r1:pe, s:r1]^T
        const int pe = std::min(pk0 + RP_CHOL_PANEL * GT, n_pad);
        rp_gemm_args u{};
        u.A = A + (long long)r1 * ld;
        u.lda = ld;
        u.B = A + (long long)r1 * ld;
        u.ldb = ld;
        u.C = A + (long long)r1 * ld + r1;
        u.ldc = ld;
        u.M = n_pad - r1;
        u.N = pe - r1;
        u.k0 = sk0;
        u.k1 = r1;
        u.tri = 1;
        u.alpha = -1.0f;
        u.beta = 1.0f;
        u.no_splitk = 1;
        if (known_scales) u.amax_known = cmax_sub + 2 * (size_t)(k / SUB);
        if (cols_pending) {  // (these columns took the previous panel's update on the third stream)
          RP_HIP(hipStreamWaitEvent(st, e_cols, 0));
          cols_pending = false;
        }
        rc = rp_gemm_nt(ctx, u, st);
        if (rc != RPHIP_OK) break;
      }
      if ((k + 1) % RP_CHOL_PANEL == 0 && r1 < n_pad) {
        // The panel is done: A[r1:, r1:] -= L[r1:, p:r1] L[r1:, p:r1]^T (lower tiles), in two pieces.  The columns of the NEXT
        // panel are updated here, on the main stream; the rest of the trailing matrix - almost all of the flops - on a third
        // stream, so that it runs beside the next panel's factorisation (a chain of small launches and one-workgroup diagonal
        // blocks that leaves the chip idle).  Its columns are not touched by that factorisation; the piece after it waits.
        const int wn = std::min(RP_CHOL_PANEL * GT, n_pad - r1);
        tr_rec(st);
        RP_HIP(hipEventRecord(e_fact, st));
        if (trail_pending) RP_HIP(hipStreamWaitEvent(st, e_trail, 0));  // the previous rest-update wrote these columns
        rp_gemm_args t{};
        t.A = A + (long long)r1 * ld;
        t.lda = ld;
        t.B = A + (long long)r1 * ld;
        t.ldb = ld;
        t.C = A + (long long)r1 * ld + r1;
        t.ldc = ld;
        // (of the next panel's columns only its FIRST sub-panel's are needed before its factorisation can start - the chain
        // that the second half of the factorisation waits for; the others follow on the third stream, in front of the rest,
        // and the next panel's first sub-panel update waits for them)
        const int w1 = (SUB < RP_CHOL_PANEL) ? std::min(SUB * GT, wn) : wn;
        t.M = n_pad - r1;
        t.N = w1;
        t.k0 = pk0;
        t.k1 = r1;
        t.tri = 1;
        t.alpha = -1.0f;
        t.beta = 1.0f;
        if (known_scales) t.amax_known = cmax_pan + 2 * (size_t)(k / RP_CHOL_PANEL);
        rc = rp_gemm_nt(ctx, t, st);
        if (rc != RPHIP_OK) break;
        tr_rec(st);
        if (w1 < wn) {
          const int rc1 = r1 + w1;
          RP_HIP(hipStreamWaitEvent(s3, e_fact, 0));
          t.A = A + (long long)rc1 * ld;
          t.B = A + (long long)rc1 * ld;
          t.C = A + (long long)rc1 * ld + rc1;
          t.M = n_pad - rc1;
          t.N = wn - w1;
          rc = rp_gemm_nt(ctx, t, s3);
          if (rc != RPHIP_OK) break;
          RP_HIP(hipEventRecord(e_cols, s3));
          cols_pending = true;
        }
        const int r2 = r1 + wn;
        if (r2 < n_pad) {
          RP_HIP(hipStreamWaitEvent(s3, e_fact, 0));
          t.A = A + (long long)r2 * ld;
          t.B = A + (long long)r2 * ld;
          t.C = A + (long long)r2 * ld + r2;
          t.M = n_pad - r2;
          t.N = n_pad - r2;
          rc = rp_gemm_nt(ctx, t, s3);
          tr_rec(s3);
          RP_HIP(hipEventRecord(e_trail, s3));
          trail_pending = true;
        }
      }
    } else {
      RP_HIP(hipStreamWaitEvent(st, e_diag, 0));
    }
    if (n22_early && k / RP_INV_PANEL < ctx->inv_n22_early && ((k + 1) % RP_INV_PANEL == 0 || k + 1 == nb)) {  // rows and W of the inverse's panel k / RP_INV_PANEL are final on `st`
      RP_HIP(hipEventRecord(e_rows, st));
      RP_HIP(hipStreamWaitEvent(s4, e_rows, 0));
      rc = n22(k / RP_INV_PANEL, s4);
      if (rc != RPHIP_OK) break;
      ++n22_issued;
    }
  }
  hipStreamSynchronize(s2);
  if (trace) {
    hipDeviceSynchronize();
    for (size_t i = 1; i < tr_ev.size(); ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, tr_ev[0], tr_ev[i]);
      fprintf(stderr, "%s%.2f", (i % 3 == 1) ? "\nchol panel: " : " ", ms);
    }
    fprintf(stderr, "\n");
    for (auto e : tr_ev) hipEventDestroy(e);
  }
  if (trail_pending) hipStreamWaitEvent(st, e_trail, 0);
  if (rc != RPHIP_OK) return rc;
  RP_HIP(hipGetLastError());
  if (timed) rp_time_end(ctx, ev, RPHIP_T_CHOLESKY, n3 / 3.0);
  // ---- K6a: N = L^-T (upper triangular), panel by panel (N12 = -N11 L21^T N22 for the 2 x 2 block form of L).
  // Inside a panel the 128-column blocks are done as before but with rows and K restricted to the panel (tiny products);
  // then ONE product per panel against everything to its left, 2048 columns wide: T = N11 L21^T (K from the row block's
  // first column), N12 = -T N22.  Block column by block column (128 wide, K from 0) every product was skinny:
  // 101 TFLOP/s with split K; the wide ones run at the square-GEMM rate.
  // Plane cache of N (rp_plane_cache): rows and columns [0, Kc) - everything when 4 n^2 bytes are free, else as many whole panels
  // as fit (the products below then come in two parts: K below Kc from the cache, the rest cut per product as before).
  struct cache_guard {
    rphip_ctx* ctx;
    rp_plane_cache pc;
    ~cache_guard() {
      if (pc.W) rp_dev_free(ctx, pc.W);  // (the block carries an event of the context's stream; side-stream users are joined below)
    }
  } cg{ctx, {}};
  rp_plane_cache& pc = cg.pc;
  if (ctx->gemm_plane_cache && st == ctx->stream && ctx->gemm_mode == 2 && ctx->gemm_presplit && ctx->gemm_wide &&
      !ctx->gemm_supertile && n_pad >= 3 * inv_w) {
    {
      // The size is a function of the DEVICE and the matrix, not of what happens to be free: the cached and the freshly cut
      // products round differently (one scale per chunk against one per product), and B should not depend on the memory other
      // users hold.  What the device has, less the matrix itself, less an eighth of it (or 24 GiB) for everything else; when that
      // much cannot be had after all, the allocation below fails and the solve runs without the cache.
      const size_t tot = ctx->hbm_bytes, mat = (size_t)n_pad * (size_t)ld * 4, reserve = std::max<size_t>((size_t)24 << 30, tot / 8);
      size_t avail = tot > mat + reserve ? tot - mat - reserve : 0;
      if (ctx->gemm_plane_cache > 1) avail = std::min<size_t>(avail, (size_t)ctx->gemm_plane_cache);
      long long Kc = (long long)std::sqrt((double)avail / 4.0);
      if (Kc >= n_pad)
        Kc = n_pad;
      else
        Kc = Kc / inv_w * inv_w;
      if (Kc >= 2 * inv_w) {
        const int chunks = (int)rp_cdiv(Kc, inv_w);
        const size_t bytes = (size_t)4 * Kc * Kc + (size_t)chunks * sizeof(unsigned) + 256;
        void* p = nullptr;
        if (rp_dev_alloc(ctx, bytes, &p) == RPHIP_OK) {
          pc.W = (unsigned char*)p;
          pc.ps = 2LL * Kc * Kc;
          pc.K = (int)Kc;
          pc.w = inv_w;
          pc.amax = (unsigned*)(pc.W + (size_t)4 * Kc * Kc);
          RP_HIP(hipMemsetAsync(pc.amax, 0, (size_t)chunks * sizeof(unsigned), st));
        } else {
          rp_set_error("%s", "");  // (no cache: the products cut their operands as before)
        }
      }
    }
  }
  const int Kc = pc.K;
  if (timed) rp_time_begin(ctx, &ev);
  constexpr int N22_AHEAD = 3;  // (only when the chains were not issued during the factorisation: diagnostic plane path)
  if (n22_issued < n_inv_panels) {
    RP_HIP(hipEventRecord(e_panel, st));  // (the factor is complete)
    RP_HIP(hipStreamWaitEvent(s2, e_panel, 0));
  }
  // diagnostic (RP_INV_TRACE=1): where the inverse's main stream spends its time, summed over the panels
  std::vector<hipEvent_t> iv_ev;
  const bool itrace = rp_diag_env("RP_INV_TRACE") != nullptr;
  auto iv_rec = [&]() {
    if (!itrace) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    hipEventRecord(e, st);
    iv_ev.push_back(e);
  };
  for (int pb = 0; pb < nb; pb += RP_INV_PANEL) {
    const int pe = std::min(nb, pb + RP_INV_PANEL);
    const int j = pb * GT, w = (pe - pb) * GT;
    const int P = pb / RP_INV_PANEL;
    for (; n22_issued < std::min(n_inv_panels, P + 1 + N22_AHEAD); ++n22_issued) RP_TRY(n22(n22_issued, s2));
    RP_HIP(hipStreamWaitEvent(st, n22_ev.v[P], 0));
    iv_rec();  // 0: after the wait for N22
    if (j > 0) {
      rp_gemm_args g{};  // T = N11 L21^T
      g.A = Nb;
      g.lda = ld;
      g.B = A + (long long)j * ld;
      g.ldb = ld;
      g.C = T;
      g.ldc = w;
      g.M = j;
      g.N = w;
      g.k0 = 0;
      g.k1 = j;
      g.kb_m = 1;
      g.alpha = 1.0f;
      g.beta = 0.0f;
      if (Kc > 0) {  // K >= Kc first (all rows, beta = 0), then K < Kc from the cache (rows < Kc only have such K)
        const int jc = std::min(j, Kc);
        if (j > jc) {
          g.k0 = jc;
          RP_TRY(rp_gemm_nt(ctx, g, st));
          g.beta = 1.0f;
        }
        g.k0 = 0;
        g.k1 = jc;
        g.M = jc;
        g.ca = &pc;
        g.no_splitk = 1;
      }
      RP_TRY(rp_gemm_nt(ctx, g, st));
      iv_rec();  // 1: T done
      hipLaunchKernelGGL(k_transpose_block, dim3(w / 32, w / 32), dim3(32, 8), 0, st, Nb + (long long)j * ld + j, ld, NT, (long long)w);
      rp_gemm_args h{};  // N12 = -T N22 (N22 is upper triangular: K ends with the column tile)
      h.ke_n = 1;
      h.A = T;
      h.lda = w;
      h.B = NT;
      h.ldb = w;
      h.C = Nb + j;
      h.ldc = ld;
      h.M = j;
      h.N = w;
      h.k0 = 0;
      h.k1 = w;
      h.alpha = -1.0f;
      h.beta = 0.0f;
      RP_TRY(rp_gemm_nt(ctx, h, st));
    } else {
      iv_rec();
    }
    iv_rec();  // 2: N12 done
    if (j < Kc) {  // columns [j, j + w) of N are final: into the cache (rows 0 .. j + w; a row tile's K starts at its own first column)
      const int c = j / inv_w, rows = j + w;
      rp_amax_op oa{Nb, ld, rows / GT, 1, 0, 0}, ob{nullptr, 0, 0, 0, 0, 0};
      hipLaunchKernelGGL(k_gemm_amax, dim3(oa.tiles, (unsigned)rp_cdiv(w, 256)), dim3(256), 0, st, oa, ob, j, j + w, 0, pc.amax + c);
      rp_split_args sa{Nb, ld, rows, j, j + w, 1, 0, 0, j, j + w, pc.amax + c, pc.W, pc.ps, (long long)pc.K / 16, 0};
      hipLaunchKernelGGL(k_split_planes, dim3(rows / 32, (unsigned)rp_cdiv(w / 16, 16)), dim3(256), 0, st, sa);
    }
    iv_rec();  // 3: into the cache
  }
  if (itrace) {
    hipStreamSynchronize(st);
    double sum[4] = {0, 0, 0, 0};
    std::string per;
    for (size_t i = 1; i < iv_ev.size(); ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, iv_ev[i - 1], iv_ev[i]);
      sum[i % 4] += ms;
      if (i % 4 == 1 || i % 4 == 2) {
        char b[32];
        snprintf(b, sizeof b, (i % 4 == 1) ? " | %.1f" : " %.1f", ms);
        per += b;
      }
    }
    fprintf(stderr, "inverse main stream: wait for N22 %.1f ms, T %.1f, N12 %.1f, cache %.1f; per panel (T, N12):%s\n", sum[0], sum[1], sum[2], sum[3], per.c_str());
    for (auto e : iv_ev) hipEventDestroy(e);
    iv_ev.clear();
    iv_rec();
  }
  // ---- K6b: P = N N^T (lower tiles, K from max(row, column) = row), mirrored into the upper half
  {
    rp_gemm_args g{};
    g.A = Nb;
    g.lda = ld;
    g.B = Nb;
    g.ldb = ld;
    g.C = A;
    g.ldc = ld;
    g.M = n_pad;
    g.N = n_pad;
    g.k0 = 0;
    g.k1 = n_pad;
    g.kb_m = 1;
    g.kb_n = 1;
    g.tri = 2;  // P is symmetric: the tiles below the diagonal (they read N on and above it, never what they write) ...
    g.alpha = 1.0f;
    g.beta = 0.0f;
    const bool diag_inline = Kc == n_pad;  // the whole product is ONE launch of the wide kernel: its diagonal tiles go along (tri = 4)
    if (diag_inline) {
      g.tri = 4;
      g.Cd = T;
    }
    if (Kc > 0) {  // as above: K >= Kc cut per product, K < Kc (rows and columns < Kc) straight from the cache
      rp_gemm_args c = g;
      if (Kc < n_pad) {
        c.k0 = Kc;
        RP_TRY(rp_gemm_nt(ctx, c, st));
        c.beta = 1.0f;
      }
      c.k0 = 0;
      c.k1 = Kc;
      c.M = c.N = Kc;
      c.ca = &pc;
      c.cb = 1;
      RP_TRY(rp_gemm_nt(ctx, c, st));
    } else {
      RP_TRY(rp_gemm_nt(ctx, g, st));
    }
    if (itrace) {
      iv_rec();
      hipStreamSynchronize(st);
      float ms = 0.f;
      hipEventElapsedTime(&ms, iv_ev[0], iv_ev[1]);
      fprintf(stderr, "inverse: N N^T (lower tiles) %.1f ms\n", ms);
      for (auto e : iv_ev) hipEventDestroy(e);
      iv_ev.clear();
      iv_rec();
    }
    if (!diag_inline) {
      g.tri = 3;  // ... the diagonal tiles into the (now free) T buffer, copied over N's diagonal blocks afterwards ...
      g.C = T;
      g.ldc = GT;
      RP_TRY(rp_gemm_nt(ctx, g, st));
    }
    hipLaunchKernelGGL(k_copy_diag_blocks, dim3(nb), dim3(256), 0, st, T, A, ld);
    if (!lower_only) hipLaunchKernelGGL(k_mirror_lower, dim3(n_pad / 32, n_pad / 32), dim3(256), 0, st, A, ld, n_pad / 32);  // ... and the mirror
    if (itrace) {
      iv_rec();
      hipStreamSynchronize(st);
      float ms = 0.f;
      hipEventElapsedTime(&ms, iv_ev[0], iv_ev[1]);
      fprintf(stderr, "inverse: diagonal tiles + mirror %.1f ms\n", ms);
      for (auto e : iv_ev) hipEventDestroy(e);
    }
  }
  RP_HIP(hipGetLastError());
  if (timed) rp_time_end(ctx, ev, RPHIP_T_INVERSE, 2.0 * n3 / 3.0);  // n^3/3 (N = L^-T) + n^3/3 (lower half of N N^T)
  if (info_ext) return RPHIP_OK;  // the caller reads the flag when it next waits for the device
  int h_info = 0;
  RP_HIP(hipMemcpyAsync(&h_info, info, 4, hipMemcpyDeviceToHost, st));
  RP_HIP(hipStreamSynchronize(st));
  if (h_info) RP_FAIL(RPHIP_ENUMERIC, "Cholesky broke down (non-positive pivot %d in a diagonal block): G + l2*I is not positive definite", h_info);
  return RPHIP_OK;
}


// ---- building blocks of a multi-GPU solve (recpack_amd/dist.py::ease_fit_sharded drives them) --------------------------
// The matrix lives in caller-owned device memory (a torch tensor per rank); collectives happen between these calls.

// rows [row_begin, row_end) of G + lambda I, with the identity block on the padding (row >= n_items), into a row-major matrix
__global__ void k_ease_init_rows(float* __restrict__ G, long long ld, int n, long long r0, long long r1, long long n_cols, float lam) {
  const long long total = (r1 - r0) * n_cols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i = r0 + e / n_cols, j = e % n_cols;
    G[i * ld + j] = (i == j) ? (i < n ? lam : 1.0f) : 0.0f;
  }
}

int rp_ease_gram_rows(rphip_ctx* ctx, const rphip_csr* X, int64_t row_begin, int64_t row_end, int64_t n_cols, float lam, float* G_dev, int64_t ld) {
  if (row_end > row_begin)
    hipLaunchKernelGGL(k_ease_init_rows, dim3(4096), dim3(256), 0, ctx->stream, G_dev, (long long)ld, (int)X->n_items, (long long)row_begin,
                       (long long)row_end, (long long)n_cols, lam);
  RP_HIP(hipGetLastError());
  const int64_t r1 = std::min<int64_t>(row_end, X->n_items);
  if (r1 > row_begin) RP_TRY(rp_gram_dense(ctx, X, row_begin, r1, G_dev, ld, lam));
  return RPHIP_OK;
}

// A (128 x 128 block, leading dimension lda) := its Cholesky factor (lower, upper part zeroed), W (128 x 128, ld 128) := the
// factor's inverse; *info_dev (device int) is set to the failing pivot + 1 when the block is not positive definite
extern "C" int rphip_potrf_trtri_128(rphip_ctx* ctx, float* A_dev, int64_t lda, float* W_dev, int32_t* info_dev) {
  RP_ENTER(ctx);
  if (!ctx || !A_dev || !W_dev || !info_dev) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  RP_TRY(rp_set_device(ctx));
  const size_t diag_lds = RP_POTRF_LDS;
  RP_HIP(hipFuncSetAttribute((const void*)k_potrf_trtri, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
  hipLaunchKernelGGL(k_potrf_trtri, dim3(1), dim3(RP_POTRF_THREADS), diag_lds, ctx->stream, A_dev, (long long)lda, W_dev, (int*)info_dev);
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}

// rphip_gemm_nt_f32 with the K clipping of the triangular factors: tile (m0, n0) starts its K loop at
// max(k_begin, k_from_row ? m_off + m0 : 0, k_from_col ? n_off + n0 : 0); lower_only skips tiles with n_off + n0 > m_off + m0
extern "C" int rphip_gemm_nt_f32_ex(rphip_ctx* ctx, int64_t M, int64_t N, int64_t k_begin, int64_t k_end, float alpha, const float* A_dev,
                                    int64_t lda, const float* B_dev, int64_t ldb, float beta, float* C_dev, int64_t ldc, int32_t k_from_row,
                                    int32_t k_from_col, int64_t m_off, int64_t n_off, int32_t lower_only) {
  RP_ENTER(ctx);
  if (!ctx || !A_dev || !B_dev || !C_dev) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if ((M % GT) || (N % GT) || (k_begin % GBK) || (k_end % GBK) || (lda % 4) || (ldb % 4) || (m_off % GT) || (n_off % GT))
    RP_FAIL(RPHIP_EINVAL, "M, N, m_off, n_off must be multiples of 128, the K range of 16, lda/ldb of 4");
  RP_TRY(rp_set_device(ctx));
  rp_gemm_args g{};
  g.A = A_dev;
  g.lda = lda;
  g.B = B_dev;
  g.ldb = ldb;
  g.C = C_dev;
  g.ldc = ldc;
  g.M = (int)M;
  g.N = (int)N;
  g.k0 = (int)k_begin;
  g.k1 = (int)k_end;
  g.kb_m = k_from_row ? 1 : 0;
  g.kb_n = k_from_col ? 1 : 0;
  g.m_off = (int)m_off;
  g.n_off = (int)n_off;
  g.tri = lower_only ? 1 : 0;
  g.alpha = alpha;
  g.beta = beta;
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  RP_TRY(rp_gemm_nt(ctx, g));
  rp_time_end(ctx, ev, RPHIP_T_CHOLESKY, 2.0 * (double)M * (double)N * (double)(k_end - k_begin));
  return RPHIP_OK;
}

// ---- building blocks of the COLUMN-SHARDED solve (recpack_amd/dist.py::ease_fit_sharded) ------------------------------------
// Every rank holds a slab of columns [col_begin, col_begin + n_cols) of the n x n matrix, all rows, row-major with leading
// dimension ld (= the slab's width).  The inverse is formed in place by block SWEEPS (Goodnight 1979): for a pivot block K with
// P = A_KK^-1, T = A[:, K] P:   A[:, J] -= T A[K, J] (every other column),  A[K, J] = T[J, :]^T,  A[:, K] = T,  A[K, K] = -P;
// after all blocks the matrix holds -A^-1.  One panel broadcast per block is the only communication, and a rank's work per block
// is one plain product on its slab - the same kernel as everything else here (k_gemm_nt).

// dst[c][r] = src[r][c]  (rows, cols multiples of 32)
__global__ void k_transpose_f32(const float* __restrict__ src, long long lds, float* __restrict__ dst, long long ldd) {
  __shared__ float t[32][33];
  const long long bx = (long long)blockIdx.x * 32, by = (long long)blockIdx.y * 32;  // bx: column of src, by: row of src
  for (int r = threadIdx.y; r < 32; r += 8) t[r][threadIdx.x] = src[(by + r) * lds + bx + threadIdx.x];
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) dst[(bx + r) * ldd + by + threadIdx.x] = t[threadIdx.x][r];
}

int rp_transpose_f32(rphip_ctx* ctx, const float* src, int64_t lds, int64_t rows, int64_t cols, float* dst, int64_t ldd, hipStream_t st) {
  hipLaunchKernelGGL(k_transpose_f32, dim3((unsigned)(cols / 32), (unsigned)(rows / 32)), dim3(32, 8), 0, st ? st : ctx->stream, src, (long long)lds, dst,
                     (long long)ldd);
  RP_HIP(hipGetLastError());
  return RPHIP_OK;
}

// A dense model that holds the COLUMNS [col_begin, col_begin + n_cols) of B (all n_items rows, leading dimension ld) in
// caller-owned device memory: predict against it scores only those columns (B sharded by item columns, a
// local top-Kp per user on every rank, then rphip_predict_merge_topk on the gathered lists).
extern "C" int rphip_model_from_dense_cols(rphip_ctx* ctx, int64_t n_items, int64_t col_begin, int64_t n_cols, float* B_dev, int64_t ld,
                                           rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !B_dev || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (n_items <= 0 || col_begin < 0 || n_cols <= 0 || col_begin + n_cols > n_items || ld < n_cols || (ld % 64))
    RP_FAIL(RPHIP_EINVAL, "columns [%lld,%lld) of %lld items, ld %lld (a multiple of 64)", (long long)col_begin, (long long)(col_begin + n_cols),
            (long long)n_items, (long long)ld);
  RP_TRY(rp_set_device(ctx));
  rphip_model* m = new rphip_model();
  m->ctx = ctx;
  m->kind = RPHIP_MODEL_DENSE;
  m->n_items = n_items;
  m->row_end = n_items;
  m->ld = ld;
  m->B = B_dev;
  m->borrows_B = true;
  m->col_begin = col_begin;
  m->n_cols = n_cols;
  m->empty_rows = -1;  // (not counted on a slab)
  *out = m;
  return RPHIP_OK;
}


// ---- EASE_XY (recpack/algorithms/experimental/ease_xy.py:61-96) --------------------------------------------------------
// B = P (X^T Y) - P diag(diag(P X^T Y) / diag(P)), columns scaled by c_j^-alpha; P = (X^T X + l2 I)^-1 as for EASE.
// X^T Y is needed with its K index contiguous for the NT product, i.e. as rows of Y^T X: the dense Gram kernel with the
// item rows taken from Y and the user rows from X.
__global__ void k_ease_xy_finish(float* __restrict__ Brr, const float* __restrict__ P, long long ld, int n, int n_pad,
                                 const int32_t* __restrict__ deg, float alpha) {
  const long long total = (long long)n_pad * ld;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / ld;
    const int j = (int)(e % ld);
    float v = 0.0f;
    if (i < n && j < n && i != j) {
      const long long dj = (long long)j * ld + j;
      const float d = Brr[dj] / P[dj];
      v = Brr[e] - P[e] * d;
      if (alpha != 0.0f) v *= deg[j] > 0 ? powf((float)deg[j], -alpha) : 0.0f;
    }
    // the diagonal entries are read by every thread of their column: write them in a second pass
    if (i != j || i >= n) Brr[e] = v;
  }
}

__global__ void k_zero_diag(float* __restrict__ B, long long ld, int n) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) B[(long long)j * ld + j] = 0.0f;
}

extern "C" int rphip_ease_xy_fit(rphip_ctx* ctx, const rphip_csr* X, const rphip_csr* Y, double l2, double alpha, rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !X || !Y || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (!(l2 > 0.0)) RP_FAIL(RPHIP_EINVAL, "l2 must be > 0 (got %g): G + l2*I has to be positive definite", l2);
  if (X->n_users != Y->n_users || X->n_items != Y->n_items) RP_FAIL(RPHIP_EINVAL, "X and y must have the same shape");
  if (X->xt_row_end >= 0 || Y->xt_row_end >= 0) RP_FAIL(RPHIP_ESTATE, "EASE_XY needs matrices with all item rows of X^T");
  RP_NEED_DEGREES(X);
  RP_NEED_DEGREES(Y);
  RP_TRY(rp_set_device(ctx));
  const int n = (int)X->n_items;
  if (n <= 0) RP_FAIL(RPHIP_EINVAL, "no items");
  const int n_pad = (int)(rp_cdiv(n, GT) * GT);
  const long long ld = n_pad;
  const size_t bytes = (size_t)n_pad * ld * sizeof(float);
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  if (3 * bytes + (1ull << 30) > free_b)
    RP_FAIL(RPHIP_ENOMEM, "EASE_XY with %d items needs %.1f GB of HBM (three %d x %d f32 matrices), %.1f GB free", n, 3 * bytes / 1e9, n_pad,
            n_pad, free_b / 1e9);
  hipStream_t st = ctx->stream;
  float *Pm = nullptr, *Gyx = nullptr;
  rphip_model* m = new rphip_model();
  m->ctx = ctx;
  m->kind = RPHIP_MODEL_DENSE;
  m->n_items = n;
  m->row_end = n;
  m->ld = ld;
  auto fail = [&](int rc) {
    hipStreamSynchronize(st);
    if (Pm) hipFree(Pm);
    if (Gyx) hipFree(Gyx);
    rphip_model_destroy(m);
    return rc;
  };
  if (rp_dev_alloc(ctx, bytes, (void**)&m->B) != RPHIP_OK || hipMalloc((void**)&Pm, bytes) != hipSuccess ||
      hipMalloc((void**)&Gyx, bytes) != hipSuccess) {
    rp_set_error("hipMalloc of the dense matrices failed");
    return fail(RPHIP_ENOMEM);
  }
  // P = (X^T X + l2 I)^-1 (ease_xy.py:80-83: the identity is float64 there, l2 is used as given; it is float32 here)
  const float lam = (float)l2;
  hipLaunchKernelGGL(k_ease_init, dim3(4096), dim3(256), 0, st, Pm, ld, n, n_pad, lam, X->deg);
  int r = rp_gram_dense(ctx, X, 0, n, Pm, ld, lam);
  if (r == RPHIP_OK) r = rp_spd_inverse(ctx, Pm, n_pad, ld);
  if (r != RPHIP_OK) return fail(r);
  // Y^T X rows, then B_rr = P (X^T Y) = P (Y^T X)^T as an NT product
  hipLaunchKernelGGL(k_zero_f32, dim3(2048), dim3(256), 0, st, Gyx, (long long)n_pad * ld);
  r = rp_gram_dense(ctx, Y, 0, n, Gyx, ld, 0.0f, X);
  if (r != RPHIP_OK) return fail(r);
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  rp_gemm_args g{};
  g.A = Pm;
  g.lda = ld;
  g.B = Gyx;
  g.ldb = ld;
  g.C = m->B;
  g.ldc = ld;
  g.M = n_pad;
  g.N = n_pad;
  g.k0 = 0;
  g.k1 = n_pad;
  g.alpha = 1.0f;
  g.beta = 0.0f;
  r = rp_gemm_nt(ctx, g);
  if (r != RPHIP_OK) return fail(r);
  rp_time_end(ctx, ev, RPHIP_T_INVERSE, 2.0 * (double)n_pad * n_pad * n_pad);
  rp_time_begin(ctx, &ev);
  hipLaunchKernelGGL(k_ease_xy_finish, dim3(4096), dim3(256), 0, st, m->B, Pm, ld, n, n_pad, X->deg, (float)alpha);
  hipLaunchKernelGGL(k_zero_diag, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, st, m->B, ld, n);
  if (hipGetLastError() != hipSuccess) return fail(RPHIP_EHIP);
  rp_time_end(ctx, ev, RPHIP_T_EASE_SCALE, 12.0 * (double)n * (double)n);
  hipStreamSynchronize(st);
  hipFree(Pm);
  hipFree(Gyx);
  Pm = Gyx = nullptr;
  r = rp_dense_count_empty(ctx, m);
  if (r != RPHIP_OK) return fail(r);
  *out = m;
  return RPHIP_OK;
}

// a dense n_pad x n_pad model (B) and, for a solve, the second matrix of the same size
static int rp_ease_alloc(rphip_ctx* ctx, int n, int n_mats, rphip_model** mo, float** Nb) {
  const int n_pad = (int)(rp_cdiv(n, GT) * GT);
  const long long ld = n_pad;
  const size_t bytes = (size_t)n_pad * ld * sizeof(float);
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  // cached blocks count as available when the cache would serve the requests (best fit with splitting, rp_dev_alloc); what it
  // cannot serve must come from the driver - if that does not fit either, the cache is released first
  if (n_mats * bytes + (1ull << 30) > free_b && ctx->cached_bytes > 0) {
    const int served = rp_dev_cache_serves(ctx, bytes, n_mats);
    if ((size_t)(n_mats - served) * bytes + (1ull << 30) > free_b) {
      hipStreamSynchronize(ctx->stream);
      rp_dev_trim(ctx);
      hipMemGetInfo(&free_b, &total_b);
      free_b += (size_t)rp_dev_cache_serves(ctx, bytes, n_mats) * bytes;  // (parts of split regions stay cached)
    } else {
      free_b += (size_t)served * bytes;
    }
  }
  if (n_mats * bytes + (1ull << 30) > free_b)
    RP_FAIL(RPHIP_ENOMEM, "EASE with %d items needs %.1f GB of HBM (%d %d x %d f32 matrices), %.1f GB free", n, n_mats * bytes / 1e9, n_mats,
            n_pad, n_pad, free_b / 1e9);
  rphip_model* m = new rphip_model();
  m->ctx = ctx;
  m->kind = RPHIP_MODEL_DENSE;
  m->n_items = n;
  m->row_begin = 0;
  m->row_end = n;
  m->K = 0;
  m->ld = ld;
  *Nb = nullptr;
  if (rp_dev_alloc(ctx, bytes, (void**)&m->B) != RPHIP_OK || (n_mats > 1 && hipMalloc((void**)Nb, bytes) != hipSuccess)) {
    if (*Nb) hipFree(*Nb);
    *Nb = nullptr;
    rphip_model_destroy(m);
    RP_FAIL(RPHIP_ENOMEM, "hipMalloc of the dense matrices failed");
  }
  *mo = m;
  return RPHIP_OK;
}

// m->B holds G + lambda I (identity on the padding): P = its inverse (K5/K6, in place), then K7
static int rp_ease_solve_scale(rphip_ctx* ctx, const rphip_csr* X, rphip_model* m, double alpha) {
  hipStream_t st = ctx->stream;
  const int n = (int)m->n_items;
  const int n_pad = (int)m->ld;
  const bool sym = (n_pad % 32 == 0) && m->ld == n_pad && rp_diag_env("RP_EASE_SCALE_ROWS") == nullptr;  // mirror + scaling in one pass
  int r = rp_spd_inverse(ctx, m->B, n_pad, m->ld, nullptr, nullptr, sym);
  hipStreamSynchronize(st);
  if (r != RPHIP_OK) return r;
  // ---- K7
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  float* colscale = nullptr;
  RP_TRY(rp_scratch(ctx, 5, (size_t)n_pad * 8 + 64, (void**)&colscale));
  int* rownz = reinterpret_cast<int*>(colscale + n_pad);  // [n_pad]
  int* empty = rownz + n_pad;
  RP_HIP(hipMemsetAsync(rownz, 0, (size_t)n_pad * 4 + 4, st));
  hipLaunchKernelGGL(k_ease_diag, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, st, m->B, m->ld, n, X->deg, (float)alpha, colscale);
  if (sym) {
    hipLaunchKernelGGL(k_ease_scale_sym, dim3(n_pad / 32, n_pad / 32), dim3(256), 0, st, m->B, m->ld, n, colscale, rownz);
    hipLaunchKernelGGL(k_count_zero_flags, dim3(64), dim3(256), 0, st, rownz, n, empty);
  } else {
    hipLaunchKernelGGL(k_ease_scale, dim3(4096), dim3(256), 0, st, m->B, m->ld, n, colscale, empty);
  }
  RP_HIP(hipGetLastError());
  rp_time_end(ctx, ev, RPHIP_T_EASE_SCALE, 8.0 * (double)n * (double)n);
  int h_empty = 0;
  RP_HIP(hipMemcpyAsync(&h_empty, empty, 4, hipMemcpyDeviceToHost, st));
  RP_HIP(hipStreamSynchronize(st));
  m->empty_rows = h_empty;
  return RPHIP_OK;
}

extern "C" int rphip_ease_fit(rphip_ctx* ctx, const rphip_csr* X, double l2, double alpha, rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !X || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (!(l2 > 0.0)) RP_FAIL(RPHIP_EINVAL, "l2 must be > 0 (got %g): G + l2*I has to be positive definite", l2);
  if (X->xt_row_end >= 0) RP_FAIL(RPHIP_ESTATE, "EASE needs a matrix with all item rows of X^T (not rphip_csr_from_device_rows)");
  RP_NEED_DEGREES(X);
  RP_TRY(rp_set_device(ctx));
  const int n = (int)X->n_items;
  if (n <= 0) RP_FAIL(RPHIP_EINVAL, "no items");
  rphip_model* m = nullptr;
  float* none = nullptr;
  RP_TRY(rp_ease_alloc(ctx, n, 1, &m, &none));
  // lambda is rounded to float32 exactly as the reference does (l2 * np.identity(dtype=float32), ease.py:82)
  const float lam = (float)l2;
  hipLaunchKernelGGL(k_ease_init, dim3(4096), dim3(256), 0, ctx->stream, m->B, m->ld, n, (int)m->ld, lam, X->deg);
  int r = rp_gram_dense(ctx, X, 0, n, m->B, m->ld, lam);
  if (r != RPHIP_OK)
    hipStreamSynchronize(ctx->stream);
  else
    r = rp_ease_solve_scale(ctx, X, m, alpha);
  if (r != RPHIP_OK) {
    rphip_model_destroy(m);
    return r;
  }
  *out = m;
  return RPHIP_OK;
}

// ---- the Gram kept for a sweep over l2 (pipelines grid-search l2, e.g. [10, 100, 1000]: the Gram does not depend on it)
__global__ void k_add_diag(float* __restrict__ G, long long ld, int n, float lam) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) G[(long long)i * ld + i] += lam;
}

extern "C" int rphip_ease_gram_create(rphip_ctx* ctx, const rphip_csr* X, rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !X || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (X->xt_row_end >= 0) RP_FAIL(RPHIP_ESTATE, "EASE needs a matrix with all item rows of X^T (not rphip_csr_from_device_rows)");
  RP_NEED_DEGREES(X);
  RP_TRY(rp_set_device(ctx));
  const int n = (int)X->n_items;
  if (n <= 0) RP_FAIL(RPHIP_EINVAL, "no items");
  rphip_model* m = nullptr;
  float* none = nullptr;
  RP_TRY(rp_ease_alloc(ctx, n, 1, &m, &none));
  m->is_gram = true;
  hipLaunchKernelGGL(k_ease_init, dim3(4096), dim3(256), 0, ctx->stream, m->B, m->ld, n, (int)m->ld, 0.0f, X->deg);
  int r = rp_gram_dense(ctx, X, 0, n, m->B, m->ld, 0.0f);
  if (r == RPHIP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) r = RPHIP_EHIP;
  if (r != RPHIP_OK) {
    rphip_model_destroy(m);
    return r;
  }
  *out = m;
  return RPHIP_OK;
}

extern "C" int rphip_ease_fit_gram(rphip_ctx* ctx, const rphip_csr* X, const rphip_model* gram, double l2, double alpha, rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !X || !gram || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (!(l2 > 0.0)) RP_FAIL(RPHIP_EINVAL, "l2 must be > 0 (got %g): G + l2*I has to be positive definite", l2);
  if (!gram->is_gram || gram->kind != RPHIP_MODEL_DENSE) RP_FAIL(RPHIP_ESTATE, "not a Gram from rphip_ease_gram_create");
  if (gram->n_items != X->n_items) RP_FAIL(RPHIP_EINVAL, "the Gram has %lld items, the matrix %lld", (long long)gram->n_items, (long long)X->n_items);
  RP_NEED_DEGREES(X);
  RP_TRY(rp_set_device(ctx));
  const int n = (int)X->n_items;
  rphip_model* m = nullptr;
  float* none = nullptr;
  RP_TRY(rp_ease_alloc(ctx, n, 1, &m, &none));
  int r = RPHIP_OK;
  if (hipMemcpyAsync(m->B, gram->B, (size_t)m->ld * m->ld * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) r = RPHIP_EHIP;
  if (r == RPHIP_OK) {
    // integer counts plus the float32 lambda: the same float32 sum the fused epilogue of rphip_ease_fit produces
    hipLaunchKernelGGL(k_add_diag, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, ctx->stream, m->B, m->ld, n, (float)l2);
    r = rp_ease_solve_scale(ctx, X, m, alpha);
  }
  if (r != RPHIP_OK) {
    rphip_model_destroy(m);
    return r;
  }
  *out = m;
  return RPHIP_OK;
}

// rows of B without any non-zero entry (items that co-occur with nothing): feeds the reference's
// "missing similar items" warning (recpack/algorithms/base.py:272-279) without a host-side nonzero()
__global__ void __launch_bounds__(256) k_count_empty_rows(const float* __restrict__ B, long long ld, int n, int* __restrict__ cnt) {
  __shared__ int any_nz;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    if (threadIdx.x == 0) any_nz = 0;
    __syncthreads();
    int nz = 0;
    for (int j = threadIdx.x; j < n; j += 256) nz |= (B[(long long)i * ld + j] != 0.0f);
    if (nz) any_nz = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !any_nz) atomicAdd(cnt, 1);
    __syncthreads();
  }
}

int rp_dense_count_empty(rphip_ctx* ctx, rphip_model* m) {
  int* cnt = nullptr;
  RP_TRY(rp_scratch(ctx, 5, 64, (void**)&cnt));
  RP_HIP(hipMemsetAsync(cnt, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_count_empty_rows, dim3((unsigned)std::min<int64_t>(m->n_items, 4096)), dim3(256), 0, ctx->stream, m->B, m->ld,
                     (int)m->n_items, cnt);
  int h = 0;
  RP_HIP(hipMemcpyAsync(&h, cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  m->empty_rows = h;
  return RPHIP_OK;
}


// ---- density pruning on the device (ease.py:97-109) -----------------------------------------------------------------
// Keep the `keep` largest |B_ij| of a dense model, zero the rest: an exact radix select over the 31 magnitude bits of
// the floats (11 + 11 + 9 bits, one histogram pass over the matrix per digit), then one pass that zeroes everything
// below the threshold.  Entries EQUAL to the threshold (rare for float32 values) are kept up to the exact count,
// whichever come first to an atomic counter - the reference's np.argpartition leaves that choice open as well.
constexpr int PR_BINS = 2048;

__global__ void __launch_bounds__(256) k_prune_hist(const float* __restrict__ B, long long ld, int n, uint32_t prefix, int shift, int width,
                                                    int hi_shift, unsigned long long* __restrict__ hist) {
  __shared__ uint32_t h[PR_BINS];
  for (int i = threadIdx.x; i < PR_BINS; i += 256) h[i] = 0u;
  __syncthreads();
  const int n4 = (n + 3) >> 2;
  const long long total = (long long)n * n4;
  const uint32_t mask = (1u << width) - 1u;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i = e / n4;
    const int j = (int)(e % n4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(B + i * ld + j);  // ld is a multiple of 4
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t a = __float_as_uint(x[c]) & 0x7fffffffu;
      if (j + c < n && a != 0u && (hi_shift >= 32 || (a >> hi_shift) == prefix)) atomicAdd(&h[(a >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PR_BINS; i += 256)
    if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

__global__ void __launch_bounds__(256) k_prune_apply(float* __restrict__ B, long long ld, int n, uint32_t thr, unsigned long long tie_budget,
                                                     unsigned long long* __restrict__ ties_taken) {
  const int n4 = (n + 3) >> 2;
  const long long total = (long long)n * n4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i = e / n4;
    const int j = (int)(e % n4) * 4;
    float4* p = reinterpret_cast<float4*>(B + i * ld + j);
    float4 v = *p;
    float x[4] = {v.x, v.y, v.z, v.w};
    bool changed = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t a = __float_as_uint(x[c]) & 0x7fffffffu;
      if (j + c >= n || a == 0u) continue;
      bool drop = a < thr;
      if (a == thr) drop = atomicAdd(ties_taken, 1ull) >= tie_budget;
      if (drop) {
        x[c] = 0.0f;
        changed = true;
      }
    }
    if (changed) *p = make_float4(x[0], x[1], x[2], x[3]);
  }
}

extern "C" int rphip_model_dense_changed(rphip_ctx* ctx, rphip_model* m) {
  RP_ENTER(ctx);
  if (!ctx || !m) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (m->kind != RPHIP_MODEL_DENSE || !m->B) RP_FAIL(RPHIP_ESTATE, "not a dense model");
  RP_TRY(rp_set_device(ctx));
  RP_TRY(rp_model_gather_wait(ctx, m, false));
  RP_HIP(hipStreamSynchronize(ctx->stream));
  rp_dev_free(ctx, m->B16);
  rp_dev_free(ctx, m->b16_delta);
  rp_dev_free(ctx, m->b16_rowmax);
  rp_dev_free(ctx, m->b8_scale);
  m->B16 = nullptr;
  m->b16_delta = m->b16_rowmax = m->b8_scale = nullptr;
  m->b16_kind = 0;
  m->b16_failed = false;
  return rp_dense_count_empty(ctx, m);
}

extern "C" int rphip_model_dense_prune(rphip_ctx* ctx, rphip_model* m, int64_t keep, int64_t* nnz_before, int64_t* nnz_after) {
  RP_ENTER(ctx);
  if (!ctx || !m) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (m->kind != RPHIP_MODEL_DENSE) RP_FAIL(RPHIP_ESTATE, "model is not a dense model");
  if (m->n_cols > 0) RP_FAIL(RPHIP_ESTATE, "the model holds a slab of columns: prune the whole matrix");
  if (keep < 0) RP_FAIL(RPHIP_EINVAL, "keep must be >= 0");
  RP_TRY(rp_set_device(ctx));
  RP_TRY(rp_model_gather_wait(ctx, m, true));
  hipStream_t st = ctx->stream;
  const int n = (int)m->n_items;
  unsigned long long* hist = nullptr;
  RP_TRY(rp_scratch(ctx, 5, (size_t)(PR_BINS + 8) * 8, (void**)&hist));
  unsigned long long* ties_taken = hist + PR_BINS;
  std::vector<unsigned long long> h(PR_BINS);
  const int grid = ctx->n_cu * 8;
  // digits from the top of the 31 magnitude bits
  const int shifts[3] = {20, 9, 0}, widths[3] = {11, 11, 9};
  uint32_t prefix = 0;
  unsigned long long need = (unsigned long long)keep, total_nz = 0, above = 0;
  for (int lv = 0; lv < 3; ++lv) {
    RP_HIP(hipMemsetAsync(hist, 0, (size_t)PR_BINS * 8, st));
    const int hi_shift = shifts[lv] + widths[lv];  // bits above this digit must equal the prefix
    hipLaunchKernelGGL(k_prune_hist, dim3(grid), dim3(256), 0, st, m->B, m->ld, n, prefix, shifts[lv], widths[lv], lv == 0 ? 32 : hi_shift, hist);
    RP_HIP(hipGetLastError());
    RP_HIP(hipMemcpyAsync(h.data(), hist, (size_t)PR_BINS * 8, hipMemcpyDeviceToHost, st));
    RP_HIP(hipStreamSynchronize(st));
    const int nb = 1 << widths[lv];
    if (lv == 0) {
      for (int b = 0; b < nb; ++b) total_nz += h[b];
      if (nnz_before) *nnz_before = (int64_t)total_nz;
      if ((unsigned long long)keep >= total_nz) {  // nothing to prune
        if (nnz_after) *nnz_after = (int64_t)total_nz;
        return RPHIP_OK;
      }
      if (keep == 0) break;
    }
    int b = nb - 1;
    for (; b > 0; --b) {
      if (h[b] >= need) break;
      need -= h[b];
      above += h[b];
    }
    // bin b holds the threshold: `need` of its members (from the top) are kept
    prefix = (prefix << widths[lv]) | (uint32_t)b;
  }
  uint32_t thr = prefix;
  unsigned long long budget = need;  // entries equal to the threshold that stay
  if (keep == 0) {
    thr = 0x7fffffffu;
    budget = 0;
  }
  RP_HIP(hipMemsetAsync(ties_taken, 0, 8, st));
  hipLaunchKernelGGL(k_prune_apply, dim3(grid), dim3(256), 0, st, m->B, m->ld, n, thr, budget, ties_taken);
  RP_HIP(hipGetLastError());
  if (nnz_after) *nnz_after = keep;
  (void)above;
  // B has changed: the fp16 copy of the two-pass predict is rebuilt on its next use
  RP_HIP(hipStreamSynchronize(st));
  rp_dev_free(ctx, m->B16);
  rp_dev_free(ctx, m->b16_delta);
  rp_dev_free(ctx, m->b16_rowmax);
  rp_dev_free(ctx, m->b8_scale);
  m->B16 = nullptr;
  m->b16_delta = m->b16_rowmax = m->b8_scale = nullptr;
  m->b16_kind = 0;
  return rp_dense_count_empty(ctx, m);
}

extern "C" int rphip_model_from_dense_host(rphip_ctx* ctx, int64_t n_items, const float* B_host, rphip_model** out) {
  RP_ENTER(ctx);
  if (!ctx || !B_host || !out) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (n_items <= 0) RP_FAIL(RPHIP_EINVAL, "no items");
  RP_TRY(rp_set_device(ctx));
  rphip_model* m = new rphip_model();
  m->ctx = ctx;
  m->kind = RPHIP_MODEL_DENSE;
  m->n_items = n_items;
  m->row_end = n_items;
  m->ld = rp_cdiv(n_items, 4) * 4;
  if (rp_dev_alloc(ctx, (size_t)n_items * m->ld * sizeof(float), (void**)&m->B) != RPHIP_OK) {
    rphip_model_destroy(m);
    RP_FAIL(RPHIP_ENOMEM, "hipMalloc of the dense matrix failed");
  }
  hipError_t e = hipMemcpy2DAsync(m->B, (size_t)m->ld * 4, B_host, (size_t)n_items * 4, (size_t)n_items * 4, (size_t)n_items,
                                  hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
    rphip_model_destroy(m);
    RP_FAIL(RPHIP_EHIP, "upload of the dense matrix failed");
  }
  int r = rp_dense_count_empty(ctx, m);
  if (r != RPHIP_OK) {
    rphip_model_destroy(m);
    return r;
  }
  *out = m;
  return RPHIP_OK;
}

extern "C" int rphip_model_dense_download(rphip_ctx* ctx, const rphip_model* m, float* B_host) {
  RP_ENTER(ctx);
  if (!ctx || !m || !B_host) RP_FAIL(RPHIP_EINVAL, "NULL argument");
  if (m->kind != RPHIP_MODEL_DENSE) RP_FAIL(RPHIP_ESTATE, "model is not dense");
  if (m->n_cols > 0) RP_FAIL(RPHIP_ESTATE, "the model holds a slab of columns (rphip_model_dense_layout), not the whole matrix");
  RP_TRY(rp_set_device(ctx));
  RP_TRY(rp_model_gather_wait(ctx, m, true));
  const size_t n = (size_t)m->n_items;
  hipEvent_t ev;
  rp_time_begin(ctx, &ev);
  RP_HIP(hipMemcpy2DAsync(B_host, n * sizeof(float), m->B, (size_t)m->ld * sizeof(float), n * sizeof(float), n, hipMemcpyDeviceToHost,
                          ctx->stream));
  rp_time_end(ctx, ev, RPHIP_T_D2H, 4.0 * (double)n * (double)n);
  RP_HIP(hipStreamSynchronize(ctx->stream));
  return RPHIP_OK;
}

extern "C" int rphip_model_dense_device(const rphip_model* m, void** B_dev, int64_t* n_items, int64_t* ld) {
  if (!m) RP_FAIL(RPHIP_EINVAL, "model is NULL");
  if (m->kind != RPHIP_MODEL_DENSE) RP_FAIL(RPHIP_ESTATE, "model is not dense");
  if (m->gather_ev && m->ctx) {  // (the caller will read the matrix outside the library's stream: the gather has to be over)
    std::unique_lock<std::recursive_mutex> lk(m->ctx->api_mu);
    RP_TRY(rp_set_device(m->ctx));
    RP_TRY(rp_model_gather_wait(m->ctx, m, false));
  }
  if (B_dev) *B_dev = m->B;
  if (n_items) *n_items = m->n_items;
  if (ld) *ld = m->ld;
  return RPHIP_OK;
}
