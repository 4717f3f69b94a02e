// Round 6 gate (EXPERIMENTS e44): the CORE LOOP of a K9 first pass that serves a BLOCK of 64 users per workgroup on the matrix cores.
//   hipcc --offload-arch=gfx950 -O3 tools/k9_block_gate.hip -o /tmp/k9_block_gate && /tmp/k9_block_gate [blocks] [rows per block] [items]
// The users of a block share the rows of their histories (C4: 2 029 distinct rows per 64 users against 3 177 row reads today), so the
// pass becomes a block-sparse product  scores[64 users][I] = A[64][R] x B8[R rows][I]  with
//   B8  the int8 rows of the model (k_dense_to_i8), row-major in HBM - streamed ONCE per block;
//   A   the users' operand: A[u][k] = scale of row k if user u has it, else 0 (fp16), prepared per block in FRAGMENT order in memory (256 KB
//       per block, read once per pass from L2: a quarter of the int8 bytes again, but no VALU work);
//   the int8 bytes transposed and converted in registers: a lane loads 8 bytes (8 columns) of each of its 8 rows, v_perm_b32 picks byte j
//   of two rows into the low bytes of an fp16 pair, OR 0x6400 makes each 1024 + byte exactly - eight B fragments (columns 8 n + j) of
//   v_mfma_f32_16x16x32_f16 from eight 8-byte loads, one VALU operation per byte; the offset 1024 x (sum of the user's scales) is the
//   same for all columns of a user and leaves at the end.
// What is measured: bytes of B8 per second of this loop (everything else of a first pass - thresholds, candidate lists - is not here).
// A workgroup = 4 waves = 64 users x 512 columns per pass; a wave 64 users x 128 columns = 4 x 8 MFMA tiles, 128 accumulator registers.
// The result of one (user, column) is checked against the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int G = 64;        // users per block
constexpr int KS = 32;       // rows per K step
constexpr int WCOLS = 128;   // columns per wave and pass
constexpr int NWAVE = 4;
constexpr int PCOLS = WCOLS * NWAVE;

template <int PF>  // K steps requested ahead of the one being multiplied
__global__ void __launch_bounds__(256, 2) k_block_pass(const int8_t* __restrict__ B8, long long ld, int n_items, const int32_t* __restrict__ rows,
                                                       const f16x8* __restrict__ Afrag, int R, float* __restrict__ out_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t* s_rows = reinterpret_cast<int32_t*>(smem);  // [R]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int32_t* br = rows + (long long)b * R;
  for (int k = tid; k < R; k += 256) s_rows[k] = br[k];
  const f16x8* Ab = Afrag + (long long)b * (R / KS) * 4 * 64;  // [K step][m tile][lane]
  __syncthreads();
  const int n16 = lane & 15, g = lane >> 4;
  float umax[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
  const int n_pass = (n_items + PCOLS - 1) / PCOLS;
  for (int pass = 0; pass < n_pass; ++pass) {
    const long long col0 = (long long)pass * PCOLS + wave * WCOLS + 8 * n16;
    f32x4 acc[4][8];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[m][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    u32x2 d[PF + 1][8];
    auto issue = [&](int ks, u32x2(&D)[8]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = s_rows[ks * KS + 8 * g + i];
        D[i] = *reinterpret_cast<const u32x2*>(B8 + (long long)r * ld + col0);
      }
    };
    f16x8 an[4];
    auto issue_a = [&](int ks) {
#pragma unroll
      for (int m = 0; m < 4; ++m) an[m] = Ab[((long long)ks * 4 + m) * 64 + lane];
    };
    const int n_ks = R / KS;
#pragma unroll
    for (int q = 0; q < PF; ++q) issue(min(q, n_ks - 1), d[q]);
    issue_a(0);
    for (int ks0 = 0; ks0 < n_ks; ks0 += PF + 1) {
#pragma unroll
      for (int sub = 0; sub <= PF; ++sub) {  // (the ring of PF + 1 register sets, statically indexed)
      const int ks = ks0 + sub;
      if (ks >= n_ks) break;
      u32x2(&D)[8] = d[sub];
      f16x8 a[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = an[m];
      issue(min(ks + PF, n_ks - 1), d[(sub + PF) % (PF + 1)]);
      issue_a(min(ks + 1, n_ks - 1));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // B fragment of column 8 n + j: rows 8 g .. 8 g + 7 as fp16 = 1024 + byte
        u32x4 f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const unsigned lo = (j < 4) ? D[2 * p].x : D[2 * p].y, hi = (j < 4) ? D[2 * p + 1].x : D[2 * p + 1].y;
          const unsigned sel = 0x0c000c00u | ((unsigned)(4 + (j & 3)) << 16) | (unsigned)(j & 3);  // bytes: [lo.bj, 0, hi.bj, 0]
          f[p] = __builtin_amdgcn_perm(hi, lo, sel) | 0x64006400u;
        }
        const f16x8 bf = __builtin_bit_cast(f16x8, f);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], bf, acc[m][j], 0, 0, 0);
      }
      }
    }
    // (stand-in for the candidate extraction: the largest sum of every user of this lane's rows)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) umax[q] = fmaxf(umax[q], acc[m][j][q] + (float)(m * 0));
    if (b == 0 && pass == 0 && wave == 0) {  // the check: user row 4 g + q of m-tile 0, column 8 n + 0
#pragma unroll
      for (int q = 0; q < 4; ++q) out_max[gridDim.x * 4 + (4 * g + q) * 16 + n16] = acc[0][0][q];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) atomicMax(reinterpret_cast<int*>(out_max) + b * 4 + q, __float_as_int(fmaxf(umax[q], 0.0f)));
}

__host__ __device__ inline unsigned fill_byte(unsigned long long pos) {
  unsigned h = (unsigned)pos * 2654435761u + (unsigned)(pos >> 32) * 40503u;
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  return h & 0xffu;
}
__global__ void k_fill(int8_t* p, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = (int8_t)fill_byte((unsigned long long)i);
}

int main(int argc, char** argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 1024;
  const int R = argc > 2 ? atoi(argv[2]) : 2048;
  const int I = argc > 3 ? atoi(argv[3]) : 100096;
  const long long ld = I;
  int8_t* B8;
  int32_t* rows;
  f16x8* Afrag;
  float* out;
  hipMalloc(&B8, (size_t)I * ld);
  hipMalloc(&rows, (size_t)NB * R * 4);
  hipMalloc(&Afrag, (size_t)NB * R * G * 2);
  hipMalloc(&out, (size_t)(NB * 4 + 4096) * 4);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, B8, (long long)I * ld);
  hipDeviceSynchronize();
  std::vector<int8_t> hB((size_t)4096 * 128);  // the first 128 columns of the first 4096 rows, for the check
  for (size_t r = 0; r < 4096; ++r)
    for (size_t c = 0; c < 128; ++c) hB[r * 128 + c] = (int8_t)fill_byte((unsigned long long)(r * ld + c));
  std::vector<int32_t> hrows((size_t)NB * R);
  std::vector<unsigned long long> hmem((size_t)NB * R);
  std::vector<float> hscale(I);
  uint32_t y = 777u;
  auto rnd = [&]() {
    y = y * 1664525u + 1013904223u;
    return y >> 8;
  };
  for (int i = 0; i < I; ++i) hscale[i] = 0.5f + (float)(rnd() % 1024) / 2048.0f;
  for (int b = 0; b < NB; ++b)
    for (int k = 0; k < R; ++k) {
      // half of a block's rows among the 4 096 most "popular" (L2 / MALL hits as in the real pass), half anywhere
      hrows[(size_t)b * R + k] = (k & 1) ? (int)(rnd() % 4096) : (int)(rnd() % I);
      if (b == 0) hrows[k] = k;  // (block 0: rows 0 .. R - 1, for the check)
      unsigned long long m = 0;
      for (int t = 0; t < 2; ++t) m |= 1ull << (rnd() % 64);  // ~1.6 users of the block hold a row
      hmem[(size_t)b * R + k] = m;
    }
  hipMemcpy(rows, hrows.data(), hrows.size() * 4, hipMemcpyHostToDevice);
  {  // the users' operand in fragment order: [block][K step][m tile][lane] x 8 halves: lane l = (user 16 m + (l & 15), k = 8 (l >> 4) + e)
    std::vector<_Float16> hA((size_t)NB * R * G);
    for (int b = 0; b < NB; ++b)
      for (int ks = 0; ks < R / KS; ++ks)
        for (int m = 0; m < 4; ++m)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
              const int k = ks * KS + 8 * (l >> 4) + e, u = 16 * m + (l & 15);
              const float sc = hscale[hrows[(size_t)b * R + k]];
              hA[((((size_t)b * (R / KS) + ks) * 4 + m) * 64 + l) * 8 + e] = ((hmem[(size_t)b * R + k] >> u) & 1ull) ? (_Float16)sc : (_Float16)0.0f;
            }
    hipMemcpy(Afrag, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  }
  const size_t lds = (size_t)R * 4;
  const int PF = argc > 4 ? atoi(argv[4]) : 2;
  auto launch = [&]() {
    if (PF == 1) hipLaunchKernelGGL(k_block_pass<1>, dim3(NB), dim3(256), lds, 0, B8, ld, I, rows, Afrag, R, out);
    else if (PF == 2) hipLaunchKernelGGL(k_block_pass<2>, dim3(NB), dim3(256), lds, 0, B8, ld, I, rows, Afrag, R, out);
    else hipLaunchKernelGGL(k_block_pass<3>, dim3(NB), dim3(256), lds, 0, B8, ld, I, rows, Afrag, R, out);
  };
  printf("prefetch %d K steps; blocks %d, rows per block %d, items %d: %.1f GB of int8 rows per launch, LDS %zu bytes per workgroup\n", PF, NB, R, I, (double)NB * R * I / 1e9, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipMemset(out, 0, (size_t)(NB * 4 + 4096) * 4);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    printf("  run %d: %.2f ms = %.2f TB/s of int8 rows%s\n", it, ms, (double)NB * R * I / (ms * 1e-3) / 1e12, err == hipSuccess ? "" : hipGetErrorString(err));
  }
  // check: block 0 (rows 0 .. R - 1), users 0 .. 15, columns 8 n (n < 16): sum_k A[u][k] (1024 + ubyte)
  std::vector<float> hout((size_t)NB * 4 + 4096);
  hipMemcpy(hout.data(), out, hout.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (int u = 0; u < 16; ++u)
    for (int n = 0; n < 16; ++n) {
      double want = 0.0;
      for (int k = 0; k < R && k < 4096; ++k)
        if ((hmem[k] >> u) & 1ull) want += (double)(float)(_Float16)hscale[k] * (1024.0 + (double)(uint8_t)hB[(size_t)k * 128 + 8 * n]);
      const double got = hout[(size_t)NB * 4 + u * 16 + n];
      worst = std::max(worst, std::abs(got - want) / std::max(1.0, std::abs(want)));
    }
  printf("check (16 users x 16 columns of block 0 against the host): worst relative difference %.2e\n", worst);
  return 0;
}
