// Pairwise divergence on the matrix pipe — DMatrix::from (src/pedigree.rs:210-261) as exact integer Gram products.
//
// Per site k and sample a the byte code is status s in {0 = U, 1 = I, 2 = M}, | 0x80 when the posterior filter drops the
// site for that sample.  With three byte planes per sample
//     v = valid            (0 / 1)
//     i = v [s = 1]        (0 / 1)
//     z = v (1 - s)        (+1 / 0 / -1)
// the reference's two sums over the sites both samples keep (:249-254) are
//     both_ab = sum_k v_a v_b
//     diff_ab = sum_k v_a v_b |s_a - s_b| = sum_k (v_a v_b - i_a i_b - z_a z_b)
// because on the valid states the 3 x 3 table |s_a - s_b| = J - e_1 e_1^T - (e_0 - e_2)(e_0 - e_2)^T (check: U/M ->
// 1 - 0 + 1 = 2, U/I -> 1, equal states -> 0).  So both = V V^T and diff = V V^T - (I I^T + Z Z^T): THREE symmetric
// n x n x L products of signed bytes with 32-bit sums — v_mfma_i32_16x16x64_i8, exact.  This is not the fit path (which
// stays off the matrix pipe); it is the byte scan of Pedigree::build, whose vector-ALU form (popcounts on bit planes,
// abn_pairwise_bits_kernel of rounds 2-3) was bound by vector issue at 0.23-0.27 of the HBM rate.
//
// Data flow: no LDS tile.  The codes are row-major per sample, so the 16 bytes lane l of a wavefront loads from row
// (l & 15) of a 16-sample block at site offset 16 (l >> 4) ARE that lane's A-operand fragment of one K = 64 step after the
// byte conversion — and, the products being A A^T, its B-operand fragment too (any assignment of sites to k slots
// works as long as both operands use the same one).  Conversion, per dword of four sites: y = (x | x >> 5) & 0x07070707
// folds the flag into bit 2 (valid -> s, filtered -> 4 + s) and three v_perm_b32 look the planes up in 8-byte tables
// held in the instruction's two sources (entries 4..7 zero): six vector instructions per four sites and sample.
//
// Work: samples in blocks of 16, blocks in groups of 4 (64 samples); a job = one pair of groups (R <= C: "super-pair")
// x one contiguous chunk of sites.  A workgroup of four wavefronts (one per SIMD; two workgroups per CU for the
// diagonal kernels) deals the chunk's batches round-robin to its wavefronts; a wavefront keeps the 10 (R = C) or 16 tiles
// of 16 x 16 sums for both products in registers over its whole share, loads one batch ahead (8 KiB per wavefront in
// flight), and at the end the four wavefronts add their tiles into LDS and the workgroup writes one packed row
// (both << 32 | diff per element, tile-major) of `partial`.  abn_pairwise_reduce_tiles_kernel sums the rows of a
// super-pair and writes diff / both / D = diff / (2 both) per pair in the reference's pair order.  No global atomics;
// integer sums: exact and independent of the order.  The sample axis is tiled (any n; n <= 64 is one super-pair).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace abn {

typedef int pmx_i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pmx_u32x4 __attribute__((ext_vector_type(4), aligned(4)));  // global loads need dword alignment only

constexpr int kPmxThreads = 256;            // four wavefronts per workgroup
constexpr int kPmxWaves = kPmxThreads / 64;
constexpr int kPmxFrags = 8;                // 16-byte fragments per lane and batch (one batch ahead in flight)
constexpr int kPmxJobElems = 16 * 256;      // packed sums per job: 4 x 4 tiles of 16 x 16

struct PairMxArgs {
  const uint8_t* codes;
  int n;                 // samples
  long long L;           // sites per sample
  int ngroups;           // groups of 64 samples
  int nchunks;           // site chunks per super-pair
  long long first;       // this launch's first super-pair: DIAG: group R = C = first + k; else index first + k among the
                         // pairs R < C in the order (0,1), (0,2), ..., (1,2), ...
  unsigned long long* partial;  // [super-pairs of the launch * nchunks][16 tiles][256]
};

// K steps (64 sites) per batch for NF fragments per step: about kPmxFrags fragments per lane and batch
__host__ __device__ constexpr int pmx_steps(int nf) { return nf >= 8 ? 1 : (nf >= 3 ? 2 : (nf == 2 ? 4 : 8)); }
// k-th pair R < C of g groups -> (R, C)
__host__ __device__ inline void pmx_offdiag(long long k, int g, int& R, int& C) {
  R = 0;
  while (k >= g - 1 - R) {
    k -= g - 1 - R;
    ++R;
  }
  C = R + 1 + (int)k;
}

// four sites (one dword of codes) -> one dword of each byte plane
__device__ __forceinline__ void pmx_planes(uint32_t x, int& v, int& i, int& z) {
  const uint32_t y = (x | (x >> 5)) & 0x07070707u;  // valid: s (0..2); filtered: 4 + s
  // v_perm_b32: selector bytes 0..3 pick from the second source, 4..7 from the first (zero here)
  v = (int)__builtin_amdgcn_perm(0u, 0x00010101u, y);
  i = (int)__builtin_amdgcn_perm(0u, 0x00000100u, y);
  z = (int)__builtin_amdgcn_perm(0u, 0x00ff0001u, y);
}

// 16 bytes of one row at byte offset `off` (fast path: the whole fragment lies inside the buffer with 4 bytes to spare)
template <bool AL4>
__device__ __forceinline__ pmx_u32x4 pmx_load(const uint8_t* codes, size_t off) {
  if constexpr (AL4) {
    return *reinterpret_cast<const pmx_u32x4*>(codes + off);
  } else {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(codes) + off;   // the caller's buffer itself may be misaligned
    const uint8_t* base = reinterpret_cast<const uint8_t*>(addr & ~(uintptr_t)3);
    const unsigned sh = (unsigned)(addr & 3);
    const pmx_u32x4 a = *reinterpret_cast<const pmx_u32x4*>(base);
    const uint32_t e = *reinterpret_cast<const uint32_t*>(base + 16);
    pmx_u32x4 r;
    r[0] = __builtin_amdgcn_alignbyte(a[1], a[0], sh);
    r[1] = __builtin_amdgcn_alignbyte(a[2], a[1], sh);
    r[2] = __builtin_amdgcn_alignbyte(a[3], a[2], sh);
    r[3] = __builtin_amdgcn_alignbyte(e, a[3], sh);
    return r;
  }
}
// the same byte by byte for the ragged end of the rows: sites >= L read as filtered
__device__ __forceinline__ pmx_u32x4 pmx_load_edge(const uint8_t* codes, size_t row_off, long long site, long long L) {
  pmx_u32x4 r;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long s = site + 4 * d + e;
      const uint32_t b = s < L ? codes[row_off + (size_t)s] : 0x80u;
      w |= b << (8 * e);
    }
    r[d] = w;
  }
  return r;
}

// NB: 16-sample blocks of the row group that exist (DIAG: 1..4, the tiles bi <= bj < NB; else 4 x 4 tiles of groups R < C)
template <int NB, bool DIAG, bool AL4>
__global__ __launch_bounds__(kPmxThreads, DIAG ? 2 : 1) void abn_pairwise_mx_kernel(const PairMxArgs a) {
  static_assert(DIAG || NB == 4, "off-diagonal super-pairs are 4 x 4 blocks");
  constexpr int NF = DIAG ? NB : 8;            // fragments per K step: the blocks of the row group (+ of the column group)
  constexpr int DSTEPS = pmx_steps(NF);        // K steps per batch
  constexpr int NT = DIAG ? NB * (NB + 1) / 2 : 16;
  __shared__ unsigned long long red[kPmxJobElems];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int g = a.ngroups;
  // job -> (super-pair, chunk)
  const long long job = blockIdx.x;
  const int chunk = (int)(job % a.nchunks);
  const long long spl = job / a.nchunks;       // super-pair of the launch
  int R, C;
  if constexpr (DIAG) R = C = (int)(a.first + spl);
  else pmx_offdiag(a.first + spl, g, R, C);

  // this lane's byte offset in each block's row: sample (clamped: rows past n give sums nobody reads) x L + 16 q
  size_t roff[NF];
#pragma unroll
  for (int b = 0; b < NF; ++b) {
    const int blk = b < 4 ? 4 * R + b : 4 * C + (b - 4);
    int s = 16 * blk + r;
    s = s < a.n ? s : a.n - 1;
    roff[b] = (size_t)s * (size_t)a.L;
  }

  pmx_i32x4 S1[NT], S2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) S1[t] = S2[t] = pmx_i32x4{0, 0, 0, 0};

  // K steps (64 sites each).  "Inner" steps lie inside every row with four bytes to spare (the unaligned loader reads one
  // dword past the fragment and the last row ends the buffer); the ragged end of the rows — at most two steps of the
  // whole launch — is staged through LDS after the loop.  The inner steps of the super-pair are split evenly over its
  // nchunks x 4 wavefronts (shares differ by at most one step), each wavefront a contiguous range.
  const long long nk_all = (a.L + 63) / 64;
  const long long nk_inner = AL4 ? a.L / 64 : (a.L >= 4 ? (a.L - 4) / 64 : 0);
  // The inner steps are split evenly over the super-pair's chunks (jobs); inside a job the four wavefronts take batches of
  // DSTEPS steps round-robin (together they read 256 DSTEPS contiguous bytes of every row: interleaved batches measured 8 %
  // faster at 32 M sites than a contiguous range per wavefront) and share what is left of the last round evenly, so that no
  // wavefront does more than one step more than another.
  // Chunk boundaries are multiples of two steps: a batch then reads whole 128-byte lines of a row whose start is aligned
  // (boundaries at odd steps split every line between two wavefronts: measured +10 % time at 32 M sites).
  const long long nk2 = nk_inner / 2;
  const long long Ks = 2 * ((long long)chunk * nk2 / a.nchunks);
  const long long Ke = chunk == a.nchunks - 1 ? nk_inner : 2 * ((long long)(chunk + 1) * nk2 / a.nchunks);
  const long long nfull = (Ke - Ks) / (kPmxWaves * DSTEPS);
  const long long R0 = Ks + nfull * (kPmxWaves * DSTEPS);
  const int rem = (int)(Ke - R0);
  const int rem_lo = wave * rem / kPmxWaves, rem_hi = (wave + 1) * rem / kPmxWaves;
  // DSTEPS steps from step k (PART: only the first cnt of them; cnt is uniform in the wavefront: scalar branches)
  auto load_steps = [&](long long k, pmx_u32x4 (&x)[DSTEPS][NF], auto part, int cnt) {
    const size_t k0 = (size_t)(k * 64 + 16 * q);
#pragma unroll
    for (int d = 0; d < DSTEPS; ++d)
      if (!decltype(part)::value || d < cnt) {
#pragma unroll
        for (int f = 0; f < NF; ++f) x[d][f] = pmx_load<AL4>(a.codes, roff[f] + k0 + (size_t)(64 * d));
      }
  };
  auto compute = [&](const pmx_u32x4 (&x)[DSTEPS][NF], auto part, int cnt) {
#pragma unroll
    for (int d = 0; d < DSTEPS; ++d) {
      if (decltype(part)::value && d >= cnt) break;
      pmx_i32x4 V[NF], I[NF], Z[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int v, i, z;
          pmx_planes(x[d][f][e], v, i, z);
          V[f][e] = v;
          I[f][e] = i;
          Z[f][e] = z;
        }
      // D[row][col] = sum_k A[row][k] B[k][col]: A = the row block's fragment, B = the column block's (same registers
      // for a diagonal tile).  The three products of a tile are issued a whole round of tiles apart: no dependent pair
      // of matrix instructions back to back.
      int t = 0;
#pragma unroll
      for (int bi = 0; bi < (DIAG ? NB : 4); ++bi)
#pragma unroll
        for (int bj = (DIAG ? bi : 0); bj < (DIAG ? NB : 4); ++bj, ++t) {
          const int fa = bi, fb = DIAG ? bj : 4 + bj;
          S1[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(V[fa], V[fb], S1[t], 0, 0, 0);
        }
      t = 0;
#pragma unroll
      for (int bi = 0; bi < (DIAG ? NB : 4); ++bi)
#pragma unroll
        for (int bj = (DIAG ? bi : 0); bj < (DIAG ? NB : 4); ++bj, ++t) {
          const int fa = bi, fb = DIAG ? bj : 4 + bj;
          S2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(I[fa], I[fb], S2[t], 0, 0, 0);
        }
      t = 0;
#pragma unroll
      for (int bi = 0; bi < (DIAG ? NB : 4); ++bi)
#pragma unroll
        for (int bj = (DIAG ? bi : 0); bj < (DIAG ? NB : 4); ++bj, ++t) {
          const int fa = bi, fb = DIAG ? bj : 4 + bj;
          S2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Z[fa], Z[fb], S2[t], 0, 0, 0);
        }
    }
  };
  constexpr std::false_type kFull{};
  constexpr std::true_type kPart{};

  // The wavefront's share of the last round is loaded FIRST and computed LAST (a third register set): the full rounds in
  // between run without a branch inside a batch — one batch in flight while the previous one is computed, two register
  // sets, the loop unrolled by two (no copies) — and the partial batch exposes no load latency at the end.
  {
    pmx_u32x4 xt[DSTEPS][NF], xa[DSTEPS][NF], xb[DSTEPS][NF];
    const int ct = rem_hi - rem_lo;
    load_steps(R0 + rem_lo, xt, kPart, ct);
    auto kfull = [&](long long r) { return Ks + (r * kPmxWaves + wave) * DSTEPS; };
    long long r = 0;
    if (nfull > 0) load_steps(kfull(0), xa, kFull, DSTEPS);
    // the workgroup's sums start at zero — cleared behind the first loads (they are in flight meanwhile); the barrier keeps
    // a wavefront that is already at the ragged end (it stages through `red`) from meeting another one's clearing stores
    for (int k = tid; k < kPmxJobElems; k += kPmxThreads) red[k] = 0ull;
    __syncthreads();
    while (r < nfull) {
      if (r + 1 < nfull) load_steps(kfull(r + 1), xb, kFull, DSTEPS);
      compute(xa, kFull, DSTEPS);
      if (++r >= nfull) break;
      if (r + 1 < nfull) load_steps(kfull(r + 1), xa, kFull, DSTEPS);
      compute(xb, kFull, DSTEPS);
      ++r;
    }
    compute(xt, kPart, ct);
  }
  // the ragged end: byte loads with the sites past L read as filtered, staged through this wavefront's share of `red`
  // (run-time indices are fine in LDS; in registers they would move the fragment arrays to scratch memory) and cleared
  // again before the sums go there.  The last wavefront of the super-pair's last chunk takes it.
  if (chunk == a.nchunks - 1 && wave == kPmxWaves - 1) {
    pmx_u32x4* stage = reinterpret_cast<pmx_u32x4*>(red) + wave * (NF * 64);
    for (long long k = nk_inner; k < nk_all; ++k) {
      const long long k0 = k * 64 + 16 * q;
      for (int f = 0; f < NF; ++f) stage[f * 64 + lane] = pmx_load_edge(a.codes, roff[f], k0, a.L);
      pmx_u32x4 xe[DSTEPS][NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) xe[0][f] = stage[f * 64 + lane];
      compute(xe, kPart, 1);
    }
    for (int f = 0; f < NF; ++f) stage[f * 64 + lane] = pmx_u32x4{0u, 0u, 0u, 0u};
  }

  // ---- the workgroup's sums: C/D layout of the 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + register
  __syncthreads();
  {
    int t = 0;
#pragma unroll
    for (int bi = 0; bi < (DIAG ? NB : 4); ++bi)
#pragma unroll
      for (int bj = (DIAG ? bi : 0); bj < (DIAG ? NB : 4); ++bj, ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned both = (unsigned)S1[t][e];
          const unsigned diff = (unsigned)(S1[t][e] - S2[t][e]);
          if (both | diff)
            atomicAdd(&red[(4 * bi + bj) * 256 + (4 * q + e) * 16 + r], ((unsigned long long)both << 32) | diff);
        }
      }
  }
  __syncthreads();
  unsigned long long* row = a.partial + (spl * a.nchunks + chunk) * kPmxJobElems;
  for (int k = tid; k < kPmxJobElems; k += kPmxThreads) {
    const int bi = k >> 10, bj = (k >> 8) & 3;
    const bool used = DIAG ? (bi <= bj && bj < NB) : true;
    if (used) row[k] = red[k];
  }
}

// Rows of `partial` -> diff[p], both[p], dvalue[p] = diff / (2 both) (:257; 0 / 0 = NaN like the reference) in the pair
// order of the reference's nested loops (:214-215).  A workgroup owns one row of one tile (16 elements = 128 contiguous
// bytes per partial row): 64 thread groups sum the chunks of the super-pair (a strided share each), LDS combines them.
constexpr int kPmxReduceGroups = 64;
__global__ __launch_bounds__(16 * kPmxReduceGroups) void abn_pairwise_reduce_tiles_kernel(
    const unsigned long long* partial, int nchunks, int n, int ngroups, int diag, long long first,
    unsigned long long* diff, unsigned long long* both, double* dvalue) {
  __shared__ unsigned long long lo[kPmxReduceGroups][16], hi[kPmxReduceGroups][16];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const long long wg = blockIdx.x;
  const int trow = (int)(wg & 15), t = (int)((wg >> 4) & 15);
  const long long sp = wg >> 8;  // super-pair of the launch (as in the kernel that wrote `partial`)
  {
    int R, C;
    if (diag) R = C = (int)(first + sp);
    else pmx_offdiag(first + sp, ngroups, R, C);
    const int bi = t >> 2, bj = t & 3;
    const long long i = 64ll * R + 16 * bi + trow, j0 = 64ll * C + 16 * bj;
    // the whole workgroup leaves together when its row holds no pair (uniform: nothing below synchronises half a group)
    if (i >= n || j0 >= n || j0 + 15 <= i) return;
    const long long j = j0 + col;
    unsigned long long alo = 0, ahi = 0;
    const unsigned long long* src = partial + (sp * nchunks) * kPmxJobElems + t * 256 + trow * 16 + col;
#pragma unroll 4
    for (int c = grp; c < nchunks; c += kPmxReduceGroups) {
      const unsigned long long v = src[(size_t)c * kPmxJobElems];
      alo += v & 0xffffffffull;
      ahi += v >> 32;
    }
    lo[grp][col] = alo;
    hi[grp][col] = ahi;
    __syncthreads();
    for (int half = kPmxReduceGroups / 2; half >= 1; half >>= 1) {
      if (grp < half) {
        lo[grp][col] += lo[grp + half][col];
        hi[grp][col] += hi[grp + half][col];
      }
      __syncthreads();
    }
    if (grp == 0 && i < j && j < n) {
      const long long p = i * n - i * (i + 1) / 2 + (j - i - 1);
      const unsigned long long d = lo[0][col], cc = hi[0][col];
      if (diff) diff[p] = d;
      if (both) both[p] = cc;
      if (dvalue) dvalue[p] = (double)d / (2.0 * (double)cc);
    }
  }
}

}  // namespace abn
