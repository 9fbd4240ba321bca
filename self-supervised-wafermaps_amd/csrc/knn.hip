// kNN retrieval: blocked pairwise-dot on MFMA + running in-register top-k.
//
// Replaces lightly.utils.benchmarking.knn_predict as called by the reference at
// src/ssl_wafermap/models/knn.py:91-98 (torch.mm -> topk -> gather -> exp -> one-hot -> argsort).
// The reference materialises sim[B,N] in fp32 and re-reads it for topk; here a block streams a
// slice of the bank through LDS once, multiplies it against a resident query tile with MFMA and
// keeps each query's best K in registers, so HBM sees the bank once per query batch and nothing
// else.  Roofline: HBM (algorithmic bytes = n*d*elsize per query batch, SURVEY §8d formula (ii)).
//
// Geometry.  A row of either operand is cut into 256-byte slabs (128 bf16 / 64 f32).  LDS images
// keep 256-byte rows, the 16-byte chunk index XOR-swizzled with (row & 15) so that the
// ds_read_b128 fragment reads (lane (r,h) -> row r, chunk 2s+h) are bank-conflict free
// (cdna_hip_programming.md T2).  MFMA orientation is "swapped": A = 32 bank rows, B = 32 queries,
// so the accumulator puts the QUERY on the lane (col = lane&31) and 16 bank rows in registers —
// the top-k list of a query is lane-local and needs no cross-lane traffic in the hot loop.
//   bf16: v_mfma_f32_32x32x16_bf16, one 16-byte fragment = one MFMA (k = 8h+j)
//   f32 : v_mfma_f32_32x32x2_f32 (exact f32 fmaf chain), one 16-byte fragment = 4 MFMAs
//
// Selection is two-stage so that the streaming loop is branch-free: a lane folds the 16 values an
// accumulator tile gives it into a running MAXIMUM over a fixed group of 64 bank rows (4 chunks x 16
// rows) and offers that one (max, group id) pair to its sorted list once per group.  The k best
// elements always lie inside the k groups with the largest maxima (each group whose max reaches
// the k-th best value holds at least one of the k best), so after the cross-block merge a small
// rescoring kernel recomputes the <= K*64 candidate rows of every query exactly and orders them
// (value descending, bank index ascending).
#include "common.h"
#include <limits.h>
#include <stdlib.h>

namespace {

constexpr int KNN_THREADS = 256;
constexpr int KNN_ROWS = 128;  // bank rows per LDS chunk: 4 waves x 32
constexpr int KNN_SLAB = 256;  // bytes per row per slab
constexpr int KNN_BUF = KNN_ROWS * KNN_SLAB;

__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) {
  return av > bv || (av == bv && ai < bi);
}

// Insert (nv, ni) into a list sorted best-first; the worst entry falls off.
template <int K>
__device__ __forceinline__ void topk_insert(float (&v)[K], int (&ix)[K], float nv, int ni) {
#pragma unroll
  for (int j = K - 1; j >= 1; --j) {
    const bool cj = better(nv, ni, v[j], ix[j]);
    const bool cjm = better(nv, ni, v[j - 1], ix[j - 1]);
    const float tv = cjm ? v[j - 1] : nv;
    const int ti = cjm ? ix[j - 1] : ni;
    v[j] = cj ? tv : v[j];
    ix[j] = cj ? ti : ix[j];
  }
  if (better(nv, ni, v[0], ix[0])) {
    v[0] = nv;
    ix[0] = ni;
  }
}

// (value, id) as one 64-bit key whose unsigned order is the list order (value descending, id ascending):
// high word = the float's bits mapped monotonically to uint32, low word = ~id.  One v_cmp_gt_u64
// replaces the three compares of better(); ids are non-negative, values never -0.0 (an MFMA chain
// starts from +0.0).
__device__ __forceinline__ unsigned long long knn_key(float v, int id) {
  uint32_t u = __float_as_uint(v);
  u ^= (u & 0x80000000u) ? 0xffffffffu : 0x80000000u;
  return ((unsigned long long)u << 32) | (uint32_t)~id;
}
__device__ __forceinline__ void knn_unkey(unsigned long long key, float& v, int& id) {
  uint32_t u = (uint32_t)(key >> 32);
  u ^= (u & 0x80000000u) ? 0x80000000u : 0xffffffffu;
  v = __uint_as_float(u);
  id = (int)~(uint32_t)key;
}

// Best K of two best-first key lists by a bitonic merge: max(a[j], b[K-1-j]) keeps the K best of the 2K
// entries as a bitonic sequence, log2(K) compare-exchange stages sort it (~6x fewer VALU operations than K
// insertions with topk_insert, same total order).
template <int K>
__device__ __forceinline__ void merge_keys(unsigned long long (&a)[K], const unsigned long long (&b)[K]) {
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = b[K - 1 - j] > a[j] ? b[K - 1 - j] : a[j];
#pragma unroll
  for (int s = K / 2; s >= 1; s >>= 1) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if ((j & s) == 0) {
        const bool c = a[j + s] > a[j];
        const unsigned long long hi = c ? a[j + s] : a[j], lo = c ? a[j] : a[j + s];
        a[j] = hi;
        a[j + s] = lo;
      }
    }
  }
}

// Block-level tail of the streaming kernels: every lane holds QT best-first lists (query = t*32 + r) for its
// (wave, half).  Fold the two halves by shuffle, then the four waves through LDS in two levels (waves
// {0,1} and {2,3} in parallel on different threads, then the two results), and store the block's list.
template <int QT, int K>
__device__ __forceinline__ void block_merge_store(float (&lv)[QT][K], int (&li)[QT][K], uint8_t* smem, int tid,
                                                  int q0, int nslices, int slice, float* __restrict__ part_sim,
                                                  int* __restrict__ part_idx) {
  constexpr int QB = QT * 32;
  const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  unsigned long long* mk = reinterpret_cast<unsigned long long*>(smem);  // [4][K][QB], the query fastest
  unsigned long long* m2 = mk + 4 * K * QB;                              // [2][K][QB]
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    unsigned long long mine[K], other[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      mine[j] = knn_key(lv[t][j], li[t][j]);
      other[j] = __shfl(mine[j], lane ^ 32, 64);
    }
    merge_keys<K>(mine, other);
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < K; ++j) mk[(wave * K + j) * QB + t * 32 + r] = mine[j];
    }
  }
  __syncthreads();
  const int q = tid % QB, pairid = tid / QB;  // QB <= 128: threads 0 .. 2*QB-1 exist
  unsigned long long f[K], g[K];
  if (tid < 2 * QB) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      f[j] = mk[((2 * pairid) * K + j) * QB + q];
      g[j] = mk[((2 * pairid + 1) * K + j) * QB + q];
    }
    merge_keys<K>(f, g);
    if (pairid == 1) {
#pragma unroll
      for (int j = 0; j < K; ++j) m2[j * QB + q] = f[j];
    }
  }
  __syncthreads();
  if (tid < QB) {
#pragma unroll
    for (int j = 0; j < K; ++j) g[j] = m2[j * QB + q];
    merge_keys<K>(f, g);
    float fv[K];
    int fi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) knn_unkey(f[j], fv[j], fi[j]);
    const size_t o = ((size_t)(q0 + tid) * nslices + slice) * K;
#pragma unroll
    for (int j = 0; j < K; j += 4) {
      *reinterpret_cast<float4*>(part_sim + o + j) = make_float4(fv[j], fv[j + 1], fv[j + 2], fv[j + 3]);
      *reinterpret_cast<int4*>(part_idx + o + j) = make_int4(fi[j], fi[j + 1], fi[j + 2], fi[j + 3]);
    }
  }
}

__device__ __forceinline__ int acc_row(int reg, int half) {
  // C/D map of the 32x32 MFMA family: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

// 256 zero bytes: the global_load_lds source of bank rows past the end
__device__ __attribute__((aligned(256))) uint8_t knn_zero_page[256];

// Streaming kernel.  The bank slice flows HBM -> LDS by global_load_lds (16 B per lane, 1 KiB per
// wave instruction, no VGPR staging) into a ring of S 32-KB stages; S-1 chunks stay in flight
// behind a counted s_waitcnt vmcnt (cdna_hip_programming.md "Pipelining across barriers"); every
// wave fetches its own 32 rows, so the loop needs no barrier.  LDS rows are 256 B; the DMA destination is lane-linear, so the XOR swizzle is
// applied to the per-lane SOURCE chunk and again on the fragment reads.
template <int DT, int QT, int K, int S>
__global__ __launch_bounds__(KNN_THREADS) void knn_stream(
    const uint8_t* __restrict__ query, const uint8_t* __restrict__ bank, int nq, int n,
    int rowbytes, int chunks_per_slice, int nslices, float* __restrict__ part_sim,
    int* __restrict__ part_idx) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int dbg = chunks_per_slice >> 24;  // timing-only ablation bits (WM_KNN_DEBUG); 0 in production
  constexpr int QB = QT * 32;
  constexpr int PER_STAGE = 8;  // global_load_lds instructions per thread per stage
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nslab = rowbytes / KNN_SLAB;
  const int q0 = blockIdx.y * QB;
  const int slice = blockIdx.x;

  uint8_t* ring = smem;                  // S x KNN_BUF
  uint8_t* qbuf = smem + S * KNN_BUF;    // QB x rowbytes: the query tile, read from LDS throughout
  // (bf16 rows of exactly one slab take knn_stream_b128 below: query fragments in registers)

  // Block b owns chunks b, b + nslices, b + 2*nslices, ...: at any instant the resident blocks read
  // a contiguous window of the bank, which spreads over all HBM channels (contiguous per-block
  // slices would start every block on the same channel).
  const int total_chunks = (n + KNN_ROWS - 1) / KNN_ROWS;
  const int my_chunks = slice < total_chunks ? (total_chunks - slice + nslices - 1) / nslices : 0;
  const int iters = my_chunks * nslab;
  (void)chunks_per_slice;

  // lane geometry of one DMA instruction: 4 rows x 16 chunks.  A thread's 8 source addresses
  // advance by a constant stride from one of its chunks to the next, so they are kept as running
  // pointers; only the last chunk of the bank can hold rows past n (-> zero page).
  const int drow = lane >> 4, dpc = lane & 15;
  const size_t chunk_stride_bytes = (size_t)nslices * KNN_ROWS * rowbytes;
  const uint8_t* srcp[PER_STAGE];
  int soff[PER_STAGE];  // byte offset of the lane's 16-byte piece inside a zero page / slab
#pragma unroll
  for (int i = 0; i < PER_STAGE; ++i) {
    const int row = wave * 32 + i * 4 + drow;  // a wave fetches exactly the 32 rows it multiplies
    soff[i] = (dpc ^ (row & 15)) * 16;  // logical chunk that lands at physical slot dpc
    srcp[i] = bank + ((size_t)slice * KNN_ROWS + row) * rowbytes + soff[i];
  }
  // issue() is called for it = 0, 1, 2, ... in order: (chunk, slab) advance by a carry instead of a division,
  // the ring slot by a wrap-around counter; LDS addresses are plain integers (no pointer cast per piece)
  int issued = 0, it_slab = 0, it_chunk = slice, it_slot = 0;
  const uint32_t ring_base = lds_addr(ring);
  auto issue = [&](int it) {
    (void)it;
    const int slab = it_slab;
    const int nb = it_chunk * KNN_ROWS;
    const bool tail = nb + KNN_ROWS > n;  // block-uniform
    const uint32_t stage = ring_base + (uint32_t)it_slot * KNN_BUF;
#pragma unroll
    for (int i = 0; i < PER_STAGE; ++i) {
      const int row0 = wave * 32 + i * 4;  // wave-uniform first row of this instruction
      const uint8_t* src = srcp[i] + (size_t)slab * KNN_SLAB;
      if (tail && nb + row0 + drow >= n) src = knn_zero_page + soff[i];
      glds16_nt_at(src, stage + row0 * KNN_SLAB);
    }
    if (++it_slab == nslab) {
      it_slab = 0;
      it_chunk += nslices;
#pragma unroll
      for (int i = 0; i < PER_STAGE; ++i) srcp[i] += chunk_stride_bytes;
    }
    if (++it_slot == S) it_slot = 0;
    ++issued;
  };
  (void)issued;

  // prologue: S-1 stages in flight, then the query tile (plain loads; drained before the loop)
#pragma unroll
  for (int p = 0; p < S - 1; ++p)
    if (p < iters && !(dbg & 2)) issue(p);
  {
    const int ppr = rowbytes >> 4;
    const int total = QB * ppr;
    for (int p = tid; p < total; p += KNN_THREADS) {
      const int row = p / ppr, c = p - row * ppr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q0 + row < nq)
        v = *reinterpret_cast<const uint4*>(query + (size_t)(q0 + row) * rowbytes + (size_t)c * 16);
      const int slab = c >> 4, ch = c & 15;
      *reinterpret_cast<uint4*>(qbuf + (size_t)row * rowbytes + slab * KNN_SLAB + ((ch ^ (row & 15)) << 4)) = v;
    }
  }

  float lv[QT][K];
  int li[QT][K];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      lv[t][j] = -INFINITY;
      li[t][j] = INT_MAX;
    }
  f32x16_t acc[QT];
  float gmax[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) gmax[t] = -INFINITY;

  __syncthreads();  // query tile staged

  for (int it = 0; it < iters; ++it) {
    // stage `it` must have landed: in steady state S-2 younger stages may still be in flight
    if (it + S - 1 < iters) {
      if constexpr (S == 5) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if constexpr (S == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (S == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // No barrier: a wave DMA-fetches exactly the rows it consumes, so its own counted vmcnt orders the
    // data for its own ds_reads, and lgkmcnt(0) retires last iteration's fragment reads before the
    // DMA below may overwrite that stage.  The four waves of a block (and all blocks) drift apart,
    // which spreads the HBM requests instead of issuing them in block-wide bursts.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (it + S - 1 < iters && !(dbg & 2)) issue(it + S - 1);  // reuses the stage consumed in iteration it-1
    if (dbg & 1) continue;
    const int slab = it % nslab;
    if (slab == 0) {
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    }
    const uint8_t* abuf = ring + (it % S) * KNN_BUF + (wave * 32 + r) * KNN_SLAB;
    {
      const uint8_t* qrow = qbuf + (size_t)r * rowbytes + slab * KNN_SLAB;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int off = ((2 * s + h) ^ (r & 15)) << 4;
        const uint4 a = *reinterpret_cast<const uint4*>(abuf + off);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          const uint4 b = *reinterpret_cast<const uint4*>(qrow + (size_t)t * 32 * rowbytes + off);
          if constexpr (DT == WM_BF16) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                             __builtin_bit_cast(bf16x8_t, b), acc[t], 0, 0, 0);
          } else {
            const f32x4_t af = __builtin_bit_cast(f32x4_t, a);
            const f32x4_t bf = __builtin_bit_cast(f32x4_t, b);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[t], 0, 0, 0);
          }
        }
      }
    }
    if (slab == nslab - 1) {
      const int crel = it / nslab;  // chunk index inside this slice
      const int nb = (slice + crel * nslices) * KNN_ROWS + wave * 32;
      if (nb + 32 <= n) {  // wave-uniform: every row of this wave's sub-tile is a real bank row
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          float m = gmax[t];
#pragma unroll
          for (int e = 0; e < 16; ++e) m = fmaxf(m, acc[t][e]);
          gmax[t] = m;
        }
      } else {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          float m = gmax[t];
#pragma unroll
          for (int e = 0; e < 16; ++e) m = fmaxf(m, (nb + acc_row(e, h) < n) ? acc[t][e] : -INFINITY);
          gmax[t] = m;
        }
      }
      if ((crel & 3) == 3 || it == iters - 1) {  // group of 4 chunks complete (or slice ends)
        const int gid = ((slice + (crel & ~3) * nslices) << 3) | (wave << 1) | h;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          if (gmax[t] > lv[t][K - 1]) topk_insert<K>(lv[t], li[t], gmax[t], gid);
          gmax[t] = -INFINITY;
        }
      }
    }
  }
  __syncthreads();  // ring is free: reuse it for the cross-wave merge

  block_merge_store<QT, K>(lv, li, smem, tid, q0, nslices, slice, part_sim, part_idx);
}

// bf16 rows of exactly one slab (d = 128, the reference's embedding width): the same stream with the
// instruction stream cut to the bone.  A wave's 32 rows of a chunk are 8 KB contiguous in HBM, so
// a stage is ONE asm statement: M0 once, eight global_load_lds_dwordx4 that share a scalar base
// (advanced by one s_add per chunk) and four loop-invariant per-lane offsets (the XOR swizzle depends
// on i & 3 only); the instruction's immediate offset moves the global source and the LDS
// destination together by 1 KiB per instruction.  The ring slot is a compile-time constant (loop
// unrolled by S), so ds_read addresses are loop-invariant registers + immediates and the query
// fragments stay in registers.
#define KNN_GLDS8(POLICY)                                                                              \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"                                                       \
               "global_load_lds_dwordx4 %1, %5 offset:-4096" POLICY "\n\t"                             \
               "global_load_lds_dwordx4 %2, %5 offset:-3072" POLICY "\n\t"                             \
               "global_load_lds_dwordx4 %3, %5 offset:-2048" POLICY "\n\t"                             \
               "global_load_lds_dwordx4 %4, %5 offset:-1024" POLICY "\n\t"                             \
               "global_load_lds_dwordx4 %1, %5" POLICY "\n\t"                                          \
               "global_load_lds_dwordx4 %2, %5 offset:1024" POLICY "\n\t"                              \
               "global_load_lds_dwordx4 %3, %5 offset:2048" POLICY "\n\t"                              \
               "global_load_lds_dwordx4 %4, %5 offset:3072" POLICY                                     \
               :                                                                                       \
               : "s"(m0v), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase)          \
               : "memory", "m0")

template <int QT, int K, int S, bool NT>
__global__ __launch_bounds__(KNN_THREADS) void knn_stream_b128(
    const uint8_t* __restrict__ query, const uint8_t* __restrict__ bank, int nq, int n, int nslices,
    int dbg, float* __restrict__ part_sim, int* __restrict__ part_idx) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int QB = QT * 32;
  constexpr int RB = KNN_SLAB;  // bytes per row
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.y * QB;
  const int slice = blockIdx.x;
  uint8_t* qbuf = smem + (S - 1) * KNN_BUF;  // query tile staged inside the last ring stage
  // WM_KNN_DEBUG bit 2: 100 MHz timestamps of the block's phases over query q0's partial similarities (timing only)
  unsigned long long stamp[4];
  stamp[0] = __builtin_amdgcn_s_memrealtime();

  const int total_chunks = (n + KNN_ROWS - 1) / KNN_ROWS;
  const int iters = slice < total_chunks ? (total_chunks - slice + nslices - 1) / nslices : 0;

  const int drow = lane >> 4, dpc = lane & 15;
  uint32_t voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voff[i] = (uint32_t)(drow * RB + ((dpc ^ ((4 * i + drow) & 15)) << 4));
  const uint64_t chunk_stride_bytes = (uint64_t)nslices * KNN_ROWS * RB;
  // +4096: the eight immediates are -4096 .. +3072
  uint64_t sbase = reinterpret_cast<uint64_t>(bank) + ((uint64_t)slice * KNN_ROWS + wave * 32) * RB + 4096;
  const uint32_t wave_lds = lds_addr(smem) + (uint32_t)wave * 32 * RB;
  int it_chunk = slice;
  auto issue = [&](int slot) {  // slot: ring stage, a compile-time constant at every call site
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(wave_lds + (uint32_t)slot * KNN_BUF + 4096);
    const int nb = it_chunk * KNN_ROWS;
    if (nb + KNN_ROWS > n) {
      // the bank's last chunk: rows past the end are fetched from row n-1 (never beyond the allocation)
      // and masked out of the maxima below
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int row = nb + wave * 32 + i * 4 + drow;
        row = row < n ? row : n - 1;
        const uint8_t* src = bank + (size_t)row * RB + ((dpc ^ ((4 * i + drow) & 15)) << 4);
        if constexpr (NT) glds16_nt_at(src, m0v - 4096 + i * 1024);
        else glds16_at(src, m0v - 4096 + i * 1024);
      }
    } else {
      if constexpr (NT) KNN_GLDS8(" nt");
      else KNN_GLDS8("");
    }
    it_chunk += nslices;
    sbase += chunk_stride_bytes;
  };

#pragma unroll
  for (int p = 0; p < S - 1; ++p)
    if (p < iters && !(dbg & 2)) issue(p);
  for (int p = tid; p < QB * 16; p += KNN_THREADS) {
    const int row = p >> 4, c = p & 15;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (q0 + row < nq) v = *reinterpret_cast<const uint4*>(query + (size_t)(q0 + row) * RB + c * 16);
    *reinterpret_cast<uint4*>(qbuf + row * RB + ((c ^ (row & 15)) << 4)) = v;
  }

  float lv[QT][K];
  int li[QT][K];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      lv[t][j] = -INFINITY;
      li[t][j] = INT_MAX;
    }
  f32x16_t acc[QT];
  float gmax[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) gmax[t] = -INFINITY;

  uint32_t aoff[8];  // this lane's eight fragment offsets inside a stage (row wave*32 + r)
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) aoff[s8] = (uint32_t)((wave * 32 + r) * RB + (((2 * s8 + h) ^ (r & 15)) << 4));
  // the query tile passes through LDS once (coalesced 16-byte loads; per-lane fragment loads straight from
  // global memory were measured 2 us slower) and then lives in registers
  uint4 qreg[QT][8];
  __syncthreads();  // query tile staged
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
      qreg[t][s8] = *reinterpret_cast<const uint4*>(qbuf + (t * 32 + r) * RB + (((2 * s8 + h) ^ (r & 15)) << 4));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // every wave has its fragments: the stage may now be overwritten
  stamp[1] = __builtin_amdgcn_s_memrealtime();

  for (int it0 = 0; it0 < iters; it0 += S) {
#pragma unroll
    for (int st = 0; st < S; ++st) {
      const int it = it0 + st;
      if (it >= iters) break;
      const bool more = it + S - 1 < iters;
      if (more) {
        if constexpr (S == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr (S == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // no barrier: a wave fetches exactly the rows it multiplies (see knn_stream)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (more && !(dbg & 2)) issue((st + S - 1) % S);
      if (dbg & 1) continue;
      uint4 a[8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) a[s8] = *reinterpret_cast<const uint4*>(smem + st * KNN_BUF + aoff[s8]);
      __builtin_amdgcn_sched_barrier(0);  // all eight fragment reads in flight before the first MFMA
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[s8]),
                                                           __builtin_bit_cast(bf16x8_t, qreg[t][s8]),
                                                           s8 == 0 ? zero : acc[t], 0, 0, 0);
        }
      }
      const int nb = (slice + it * nslices) * KNN_ROWS + wave * 32;
      if (nb + 32 <= n) {  // wave-uniform: all 32 rows of this wave's sub-tile exist
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          float m = gmax[t];
#pragma unroll
          for (int e = 0; e < 16; ++e) m = fmaxf(m, acc[t][e]);
          gmax[t] = m;
        }
      } else {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          float m = gmax[t];
#pragma unroll
          for (int e = 0; e < 16; ++e) m = fmaxf(m, (nb + acc_row(e, h) < n) ? acc[t][e] : -INFINITY);
          gmax[t] = m;
        }
      }
      if ((it & 3) == 3 || it == iters - 1) {  // group of 4 chunks complete (or slice ends)
        const int gid = ((slice + (it & ~3) * nslices) << 3) | (wave << 1) | h;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          if (gmax[t] > lv[t][K - 1]) topk_insert<K>(lv[t], li[t], gmax[t], gid);
          gmax[t] = -INFINITY;
        }
      }
    }
  }
  stamp[2] = __builtin_amdgcn_s_memrealtime();
  __syncthreads();  // ring is free: reuse it for the cross-wave merge
  block_merge_store<QT, K>(lv, li, smem, tid, q0, nslices, slice, part_sim, part_idx);
  if ((dbg & 4) && tid == 0) {
    stamp[3] = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = reinterpret_cast<unsigned long long*>(part_sim + ((size_t)q0 * nslices + slice) * K);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = stamp[j];
  }
}

// One wave per query: merge `parts` sorted lists of `kin` candidates into the best `kout`.
// candidate (p, j) of query q lives at q*stride_q + p*stride_p + j.
template <int K>
__global__ __launch_bounds__(64) void knn_merge_lists(const float* __restrict__ in_sim,
                                                      const int* __restrict__ in_idx, int parts,
                                                      int kin, long long stride_q,
                                                      long long stride_p, int kout,
                                                      float* __restrict__ out_sim,
                                                      int* __restrict__ out_idx) {
  const int q = blockIdx.x;
  const int lane = threadIdx.x;
  float v[K];
  int ix[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    v[j] = -INFINITY;
    ix[j] = INT_MAX;
  }
  const int total = parts * kin;
  for (int c = lane; c < total; c += 64) {
    const int p = c / kin, j = c - p * kin;
    const size_t o = (size_t)q * stride_q + (size_t)p * stride_p + j;
    const float cv = in_sim[o];
    const int ci = in_idx[o];
    if (better(cv, ci, v[K - 1], ix[K - 1])) topk_insert<K>(v, ix, cv, ci);
  }
  for (int t = 0; t < kout; ++t) {
    float bv = v[0];
    int bi = ix[0];
    int bl = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const int ol = __shfl_xor(bl, o, 64);
      if (better(ov, oi, bv, bi) || (ov == bv && oi == bi && ol < bl)) {
        bv = ov;
        bi = oi;
        bl = ol;
      }
    }
    if (lane == bl) {
#pragma unroll
      for (int j = 0; j < K - 1; ++j) {
        v[j] = v[j + 1];
        ix[j] = ix[j + 1];
      }
      v[K - 1] = -INFINITY;
      ix[K - 1] = INT_MAX;
    }
    if (lane == 0) {
      out_sim[(size_t)q * kout + t] = bv;
      out_idx[(size_t)q * kout + t] = bi;
    }
  }
}

// Wave maximum of 64-bit keys on the DPP network (no LDS round trips): two quad permutes and the two row mirrors make
// every lane of a 16-lane row hold the row's maximum, row_bcast15 / row_bcast31 fold the rows into lane 63, which is
// read back as a wave-uniform scalar.  ~35 VALU instructions against six ds_bpermute round trips (x2 words) for
// the __shfl_xor butterfly: the selection kernel below runs 3 K such reductions back to back.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_key(unsigned long long v) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  const uint32_t ol = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  const uint32_t oh = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)oh << 32) | ol;
}
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long v) {
  unsigned long long o;
  o = dpp_key<0xB1, 0xf>(v);  v = o > v ? o : v;   // quad_perm [1,0,3,2]
  o = dpp_key<0x4E, 0xf>(v);  v = o > v ? o : v;   // quad_perm [2,3,0,1]
  o = dpp_key<0x141, 0xf>(v); v = o > v ? o : v;   // row_half_mirror
  o = dpp_key<0x140, 0xf>(v); v = o > v ? o : v;   // row_mirror: the row's 16 lanes agree
  o = dpp_key<0x142, 0xa>(v); v = o > v ? o : v;   // row_bcast15 into rows 1 and 3
  o = dpp_key<0x143, 0xc>(v); v = o > v ? o : v;   // row_bcast31 into rows 2 and 3: lane 63 has the maximum
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, 63), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// The K largest of the wave's P-per-lane keys, best first, as wave-uniform values (0 = none left).  Keys are
// distinct (or 0), so removing the winner is a compare against it.
template <int P, int K>
__device__ __forceinline__ void wave_topk(unsigned long long (&key)[P], unsigned long long (&win)[K]) {
#pragma unroll
  for (int t = 0; t < K; ++t) {
    unsigned long long m = key[0];
#pragma unroll
    for (int p = 1; p < P; ++p) m = key[p] > m ? key[p] : m;
    const unsigned long long w = wave_max_key(m);
#pragma unroll
    for (int p = 0; p < P; ++p) key[p] = key[p] == w ? 0ull : key[p];
    win[t] = w;
  }
}

__device__ __forceinline__ unsigned long long group_key(float v, int g) { return g != INT_MAX ? knn_key(v, g) : 0ull; }
__device__ __forceinline__ int key_id(unsigned long long key) { return key != 0ull ? (int)~(uint32_t)key : INT_MAX; }

// One block per query: merge the per-slice group lists, rescore the winning groups, pick the top k.
//  1. the K best GROUPS can only come from the K runs (slices) with the best heads: every wave picks the K best
//     heads of its share of the `nslices` sorted runs (two per lane), wave 0 the K best of those 4 K; group ids are
//     unique across runs, so (value, group id) as one 64-bit key is a total order and the run of a winner is
//     (group id >> 3) % nslices (a streaming block's chunks are slice, slice + nslices, ...);
//  2. the K x K entries of those runs -> the K best groups (K = 8: one entry per lane of wave 0; K = 16: one per
//     thread, two-level as for the heads);
//  3. recompute the K*64 candidate rows exactly (4 lanes per row, 16-byte pieces, xor-shuffle sum);
//  4. k rounds of wave arg-best (value descending, bank index ascending).
// All arg-best rounds are wave_topk on the DPP network; three block barriers in all (K = 8).  The K = 8 form is kept
// LEAN on purpose -- 64 registers per lane, ~5 KB of LDS, rescoring in four rounds of two passes -- so that its blocks
// fit on CUs that already hold two blocks of the NEXT batch's streaming kernel (214 registers and 64 KB each): with
// batches issued round-robin on streams the selection runs under the following streaming kernel.  (Parking the whole
// run lists in LDS saved the second trip to memory of step 2 but cost 37 KB and 178 registers: same 51 us for one
// call, 40.0 instead of 39.0 us per batch pipelined.)
// TH = 1024 (k <= 8, the default): sixteen waves per query -- one run head per thread and all 512 candidate rows rescored
// in ONE round of dependent loads instead of four, still 64 registers per lane: 51.4 -> 48.4 us per call, 40.1 -> 39.0 us
// per batch pipelined over three streams (tools/bench_knn_pipeline.py, WM_KNN_SELECT_THREADS=256 / 1024 on one box).
template <int DT, int K, int TH = 256>
__global__ __launch_bounds__(TH, (K == 8 && TH == 256) ? 6 : 1) void knn_select(const uint8_t* __restrict__ query,
                                                  const uint8_t* __restrict__ bank, int n, int d,
                                                  int rowbytes, const float* __restrict__ part_sim,
                                                  const int* __restrict__ part_idx, int nslices,
                                                  int chunk_stride, int total_chunks, int index_base,
                                                  int kout, float* __restrict__ out_sim,
                                                  int* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) uint8_t sl_smem[];
  static_assert(TH == 256 || (TH == 1024 && K == 8), "256 threads, or 1024 for k <= 8");
  constexpr int NW = TH / 64;                                               // waves
  unsigned long long* hk = reinterpret_cast<unsigned long long*>(sl_smem);  // [NW K] wave winners
  int* topg = reinterpret_cast<int*>(hk + NW * K);                          // [K] selected runs, then groups
  float* cv = reinterpret_cast<float*>(topg + K);                           // [K*64]
  int* ci = reinterpret_cast<int*>(cv + K * 64);                            // [K*64]
  float* qf = reinterpret_cast<float*>(ci + K * 64);                        // [d]
  const int q = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const size_t qbase = (size_t)q * nslices * K;
  unsigned long long stamp[6];  // WM_KNN_DEBUG bit 2 (kout < 0): phase timestamps, see tools/knn_stamps.py
  const bool stamps = kout < 0;
  if (stamps) kout = -kout;
  stamp[0] = __builtin_amdgcn_s_memrealtime();

  for (int c = tid; c < d; c += TH) {
    if constexpr (DT == WM_BF16) qf[c] = bf2f(reinterpret_cast<const uint16_t*>(query + (size_t)q * rowbytes)[c]);
    else qf[c] = reinterpret_cast<const float*>(query + (size_t)q * rowbytes)[c];
  }
  // 1. heads (and, K = 8, the whole lists) of this lane's two runs
  {
    constexpr int U = 512 / TH > 0 ? 512 / TH : 1;  // runs per lane (nslices <= 512: host-checked)
    unsigned long long head[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int run = tid + TH * u;
      head[u] = 0ull;
      if (run < nslices) head[u] = group_key(part_sim[qbase + (size_t)run * K], part_idx[qbase + (size_t)run * K]);
    }
    unsigned long long win[K];
    wave_topk<U, K>(head, win);
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < K; ++t) hk[wv * K + t] = win[t];
    }
  }
  __syncthreads();
  stamp[1] = __builtin_amdgcn_s_memrealtime();
  // 2. wave 0: the K runs that can hold the K best groups, then (K = 8) the K best of their K x K entries
  if (wv == 0) {
    constexpr int P2 = (NW * K + 63) / 64;  // wave winners per lane
    unsigned long long c[P2];
#pragma unroll
    for (int u = 0; u < P2; ++u) c[u] = lane + 64 * u < NW * K ? hk[lane + 64 * u] : 0ull;
    unsigned long long win[K];
    wave_topk<P2, K>(c, win);
    if constexpr (K * K <= 64) {
      // K = 8: lane L takes entry L % K of the (L / K)-th selected run: K x K = 64 entries, one per lane
      unsigned long long rk = 0ull;
#pragma unroll
      for (int t = 0; t < K; ++t) rk = (lane / K) == t ? win[t] : rk;
      const int rg = key_id(rk);
      unsigned long long e[1] = {0ull};
      if (rg != INT_MAX) {
        const size_t o = qbase + (size_t)((rg >> 3) % nslices) * K + (lane % K);
        e[0] = group_key(part_sim[o], part_idx[o]);
      }
      wave_topk<1, K>(e, win);
      if (lane == 0) {
#pragma unroll
        for (int t = 0; t < K; ++t) topg[t] = key_id(win[t]);
      }
    } else {
      if (lane == 0) {
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const int rg = key_id(win[t]);
          topg[t] = rg != INT_MAX ? (rg >> 3) % nslices : INT_MAX;
        }
      }
    }
  }
  __syncthreads();
  if constexpr (K * K > 64) {
    // K = 16: 256 entries, one per thread, fetched now; two-level arg-best as for the heads
    unsigned long long e[1] = {0ull};
    if (tid < K * K) {
      const int run = topg[tid / K];
      if (run != INT_MAX)
        e[0] = group_key(part_sim[qbase + (size_t)run * K + (tid % K)], part_idx[qbase + (size_t)run * K + (tid % K)]);
    }
    unsigned long long win[K];
    wave_topk<1, K>(e, win);
    __syncthreads();  // every thread has read its run from topg
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < K; ++t) hk[wv * K + t] = win[t];
    }
    __syncthreads();
    if (wv == 0) {
      unsigned long long c[1] = {lane < 4 * K ? hk[lane] : 0ull};
      wave_topk<1, K>(c, win);
      if (lane == 0) {
#pragma unroll
        for (int t = 0; t < K; ++t) topg[t] = key_id(win[t]);
      }
    }
    __syncthreads();
  }
  stamp[2] = __builtin_amdgcn_s_memrealtime();

  // ---- exact rescoring: 4 lanes per row, 64 rows per pass; a lane's pieces are sub, sub+4, ...
  // All loads of PASSES passes are issued before any arithmetic (the work is pure latency).
  const int sub = tid & 3, slot = tid >> 2;
  const int pieces = rowbytes >> 4;  // 16-byte pieces per row
  constexpr int EPP = DT == WM_BF16 ? 8 : 4;  // elements per piece
  // K = 8: two passes in flight (the lean form, see the kernel's header)
  constexpr int PASSES = K == 8 ? 2 : 4;
  constexpr int ncand = K * 64;  // 4 chunks x 16 rows per group
  constexpr int RP = TH / 4;     // rows per pass
  static_assert(ncand % (RP * PASSES) == 0, "whole rescoring rounds");
  for (int c0p = 0; c0p < ncand; c0p += RP * PASSES) {
    int rows[PASSES];
    float accs[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int c = c0p + ps * RP + slot;
      const int gid = topg[c >> 6];
      const int j = c & 63, cc = j >> 4, e = j & 15;
      int row = INT_MAX;
      if (gid != INT_MAX) {
        const int c0 = gid >> 3, w = (gid >> 1) & 3, hh = gid & 1;
        const int chunk = c0 + cc * chunk_stride;  // the streaming block's next chunks
        const int rr = chunk * KNN_ROWS + w * 32 + acc_row(e, hh);
        // (unsigned: a corrupt group id must not turn into an address)
        if ((unsigned)chunk < (unsigned)total_chunks && (unsigned)rr < (unsigned)n) row = rr;
      }
      rows[ps] = row;
      accs[ps] = 0.f;
    }
    for (int pc0 = 0; pc0 < pieces; pc0 += 16) {  // 4 pieces per lane per step
      uint4 u[PASSES][4];
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int pc = pc0 + sub + 4 * x;
          u[ps][x] = (rows[ps] != INT_MAX && pc < pieces)
                         ? *reinterpret_cast<const uint4*>(bank + (size_t)rows[ps] * rowbytes + pc * 16)
                         : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int pc = pc0 + sub + 4 * x;
          if (pc < pieces) {
            const float* qq = qf + pc * EPP;
            const uint32_t ws[4] = {u[ps][x].x, u[ps][x].y, u[ps][x].z, u[ps][x].w};
            if constexpr (DT == WM_BF16) {
#pragma unroll
              for (int y = 0; y < 4; ++y) {
                accs[ps] = fmaf(bf2f((uint16_t)(ws[y] & 0xffff)), qq[2 * y], accs[ps]);
                accs[ps] = fmaf(bf2f((uint16_t)(ws[y] >> 16)), qq[2 * y + 1], accs[ps]);
              }
            } else {
#pragma unroll
              for (int y = 0; y < 4; ++y) accs[ps] = fmaf(__builtin_bit_cast(float, ws[y]), qq[y], accs[ps]);
            }
          }
        }
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      float a = accs[ps];
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      const int c = c0p + ps * RP + slot;
      if (sub == 0) {
        cv[c] = rows[ps] != INT_MAX ? a : -INFINITY;
        ci[c] = rows[ps];
      }
    }
  }
  __syncthreads();
  stamp[3] = __builtin_amdgcn_s_memrealtime();
  if (tid < 64) {
    // one wave: every lane keeps its K candidates as 64-bit keys in registers (0 = none); bank rows are distinct,
    // so the key identifies the winner of a round
    unsigned long long key[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int c = tid + 64 * i;
      const int idx = ci[c];
      key[i] = idx != INT_MAX ? knn_key(cv[c], idx) : 0ull;
    }
    unsigned long long win[K];
    wave_topk<K, K>(key, win);
    if (tid == 0) {
#pragma unroll
      for (int t = 0; t < K; ++t) {
        if (t < kout) {
          float bv = -INFINITY;
          int bi = -1;
          if (win[t] != 0ull) {
            knn_unkey(win[t], bv, bi);
            bi += index_base;
          }
          out_sim[(size_t)q * kout + t] = bv;
          out_idx[(size_t)q * kout + t] = bi;
        }
      }
    }
  }
  if (stamps && tid == 0) {
    stamp[4] = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(part_sim) + qbase);
    for (int j = 0; j < 5; ++j) o[j] = stamp[j];
  }
}

constexpr int VOTE_MAX_CLASSES = 64;

__global__ void knn_vote_kernel(const float* __restrict__ sim, const int* __restrict__ idx,
                                const long long* __restrict__ labels, long long n_labels, int nq, int k,
                                int nc, float t, long long* __restrict__ pred,
                                float* __restrict__ scores_out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  float score[VOTE_MAX_CLASSES];
  for (int c = 0; c < nc; ++c) score[c] = 0.f;
  for (int j = 0; j < k; ++j) {
    // reference order of operations: (sim / t).exp(), then a sum over the k neighbours in order
    const int nb = idx[(size_t)q * k + j];
    if (nb < 0 || nb >= n_labels) continue;  // padding entry of a short shard list (sim = -inf): no vote
    const float w = expf(sim[(size_t)q * k + j] / t);
    const int lab = (int)labels[nb];
    for (int c = 0; c < nc; ++c) score[c] += (c == lab) ? w : 0.f;
  }
  if (scores_out)
    for (int c = 0; c < nc; ++c) scores_out[(size_t)q * nc + c] = score[c];
  // selection sort: descending score, ties -> lower class id
  unsigned long long used = 0ull;
  for (int o = 0; o < nc; ++o) {
    int best = -1;
    for (int c = 0; c < nc; ++c) {
      if ((used >> c) & 1ull) continue;
      if (best < 0 || score[c] > score[best]) best = c;
    }
    pred[(size_t)q * nc + o] = best;
    used |= 1ull << best;
  }
}

inline int pick_qt(int rowbytes, int nq, int kt) {
  int qt = kt > 8 ? 2 : 4;  // K=16 lists at QT=4 would not fit the register file
  while (qt > 1 && qt * 32 * rowbytes > 65536) qt >>= 1;   // query tile <= 64 KB of LDS
  while (qt > 2 && (qt / 2) * 32 >= nq) qt >>= 1;            // do not carry empty query sub-tiles
  return qt;
}

struct KnnPlan {
  int qt, qtiles, nslices, chunks_per_slice, kt;
};

inline KnnPlan make_plan(int nq, int n, int rowbytes, int k) {
  KnnPlan p;
  p.kt = k <= 8 ? 8 : 16;
  p.qt = pick_qt(rowbytes, nq, p.kt);
  p.qtiles = wm_cdiv(nq, p.qt * 32);
  const int total_chunks = wm_cdiv(n, KNN_ROWS);
  int want = (p.qt == 4 ? 256 : 512) / p.qtiles;  // resident blocks per CU: 1 (128-query tiles) or 2
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  static const int forced = [] {  // experiment knob, read once
    const char* e = getenv("WM_KNN_SLICES");
    return e ? atoi(e) : 0;
  }();
  if (forced > 0) want = forced;
  p.nslices = total_chunks < want ? total_chunks : want;
  p.chunks_per_slice = wm_cdiv(total_chunks, p.nslices);
  p.nslices = wm_cdiv(total_chunks, p.chunks_per_slice);
  return p;
}

inline int knn_debug_bits() {
  static int bits = -1;
  if (bits < 0) {
    const char* e = getenv("WM_KNN_DEBUG");
    bits = e ? atoi(e) & 7 : 0;
  }
  return bits;
}

template <int DT, int QT, int K, int S>
int launch_stream(const KnnPlan& p, const void* query, const void* bank, int nq, int n, int rowbytes,
                  float* ps, int* pi, hipStream_t st) {
  const size_t lds = (size_t)S * KNN_BUF + (size_t)QT * 32 * rowbytes;
  static bool attr_set = false;  // idempotent; a race only repeats the call
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_stream<DT, QT, K, S>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(p.nslices, p.qtiles);
  knn_stream<DT, QT, K, S><<<grid, KNN_THREADS, lds, st>>>(
      static_cast<const uint8_t*>(query), static_cast<const uint8_t*>(bank), nq, n, rowbytes,
      p.chunks_per_slice | (knn_debug_bits() << 24), p.nslices, ps, pi);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

template <int QT, int K, int S, bool NT>
int launch_stream_b128(const KnnPlan& p, const void* query, const void* bank, int nq, int n, float* ps, int* pi,
                       hipStream_t st) {
  const size_t lds = (size_t)S * KNN_BUF;
  static bool attr_set = false;  // idempotent; a race only repeats the call
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_stream_b128<QT, K, S, NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(p.nslices, p.qtiles);
  knn_stream_b128<QT, K, S, NT><<<grid, KNN_THREADS, lds, st>>>(
      static_cast<const uint8_t*>(query), static_cast<const uint8_t*>(bank), nq, n, p.nslices, knn_debug_bits(), ps, pi);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

inline bool knn_nt() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("WM_KNN_NT");
    v = e ? atoi(e) != 0 : 1;
  }
  return v != 0;
}

// Instantiations are kept few (each is a large, fully unrolled kernel): query tiles of 64 or 128
// (a smaller batch is zero-padded), K = 8 or 16 (16 only with 64-query tiles); bf16 rows of exactly one
// slab (d = 128) take the specialised kernel, everything else the general one with a 3-stage ring.
template <int DT, int QT, int K>
int launch_block(const KnnPlan& p, const void* query, const void* bank, int nq, int n, int rowbytes,
                 float* ps, int* pi, hipStream_t st) {
  if constexpr (DT == WM_BF16) {
    // 64-query tiles: 2-stage ring, two blocks per CU (2 waves/SIMD overlap issue and waits);
    // 128-query tiles need ~400 registers per lane, i.e. one block per CU: 4-stage ring instead
    if (rowbytes == KNN_SLAB) {
      if constexpr (QT == 4) {
        return knn_nt() ? launch_stream_b128<QT, K, 4, true>(p, query, bank, nq, n, ps, pi, st)
                        : launch_stream_b128<QT, K, 4, false>(p, query, bank, nq, n, ps, pi, st);
      } else {
        return knn_nt() ? launch_stream_b128<QT, K, 2, true>(p, query, bank, nq, n, ps, pi, st)
                        : launch_stream_b128<QT, K, 2, false>(p, query, bank, nq, n, ps, pi, st);
      }
    }
  }
  return launch_stream<DT, QT, K, 3>(p, query, bank, nq, n, rowbytes, ps, pi, st);
}

template <int DT, int K>
int dispatch_qt(const KnnPlan& p, const void* query, const void* bank, int nq, int n, int rowbytes,
                float* ps, int* pi, hipStream_t st) {
  if constexpr (K <= 8) {
    if (p.qt == 4) return launch_block<DT, 4, K>(p, query, bank, nq, n, rowbytes, ps, pi, st);
  }
  if constexpr (DT == WM_F32 && K <= 8) {  // 2048-byte rows (512 float32 features): 32-query tiles
    if (p.qt == 1) return launch_block<DT, 1, K>(p, query, bank, nq, n, rowbytes, ps, pi, st);
  }
  if (p.qt == 1) return WM_EUNSUPPORTED;  // k > 8 with rows above 1 KB is not instantiated
  return launch_block<DT, 2, K>(p, query, bank, nq, n, rowbytes, ps, pi, st);
}

template <int DT, int K>
int launch_select(const KnnPlan& p, const void* query, const void* bank, int n, int d, int rowbytes, int nq,
                  const float* ps, const int* pi, int index_base, int kout, float* out_sim, int* out_idx,
                  hipStream_t st) {
  if (p.nslices > 512) return WM_EUNSUPPORTED;  // two runs per lane of the selection block
  // wave winners (up to 16 waves), selected runs / groups, rescored candidates, the query
  const size_t lds = (16 * K * 8 + K * 4 + (size_t)K * 64 * 8 + (size_t)d * 4 + 15) & ~(size_t)15;
  const int kk = (knn_debug_bits() & 4) ? -kout : kout;
  if constexpr (K == 8) {
    // 1024 threads: the 512 candidate rows are rescored in ONE round of dependent loads instead of four
    // (WM_KNN_SELECT_THREADS=256 keeps the lean form; read per call: A/B switch)
    const char* e = getenv("WM_KNN_SELECT_THREADS");
    if (e == nullptr || atoi(e) == 1024) {
      knn_select<DT, K, 1024><<<nq, 1024, lds, st>>>(static_cast<const uint8_t*>(query), static_cast<const uint8_t*>(bank),
                                                     n, d, rowbytes, ps, pi, p.nslices, p.nslices, wm_cdiv(n, KNN_ROWS),
                                                     index_base, kk, out_sim, out_idx);
      WM_LAUNCH_CHECK();
      return WM_OK;
    }
  }
  knn_select<DT, K><<<nq, 256, lds, st>>>(static_cast<const uint8_t*>(query), static_cast<const uint8_t*>(bank), n, d,
                                          rowbytes, ps, pi, p.nslices, p.nslices, wm_cdiv(n, KNN_ROWS), index_base,
                                          kk, out_sim, out_idx);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

}  // namespace

extern "C" size_t wm_knn_topk_workspace_bytes(int nq, int n, int d, int k) {
  if (nq <= 0 || n <= 0 || d <= 0 || k <= 0 || k > 16) return 0;
  // sized for the wider element type so one workspace serves both dtypes
  const KnnPlan pb = make_plan(nq, n, d * 2, k);
  const KnnPlan pf = make_plan(nq, n, d * 4, k);
  const size_t a = (size_t)pb.qtiles * pb.qt * 32 * pb.nslices * pb.kt * 8;
  const size_t b = (size_t)pf.qtiles * pf.qt * 32 * pf.nslices * pf.kt * 8;
  return (a > b ? a : b) + (size_t)nq * pb.kt * 8 + 256;
}

extern "C" int wm_knn_topk(const void* query, const void* bank, int nq, int n, int d, int dtype,
                           int k, int bank_index_base, float* out_sim, int32_t* out_idx,
                           void* workspace, size_t workspace_bytes, void* stream) {
  WM_REQUIRE(query && bank && out_sim && out_idx && workspace, WM_EINVAL);
  WM_REQUIRE(nq > 0 && n > 0 && d > 0 && k > 0, WM_EINVAL);
  WM_REQUIRE(dtype == WM_F32 || dtype == WM_BF16, WM_EUNSUPPORTED);
  WM_REQUIRE(k <= 16 && k <= n, WM_EUNSUPPORTED);
  const int rowbytes = d * (dtype == WM_BF16 ? 2 : 4);
  WM_REQUIRE(rowbytes % KNN_SLAB == 0 && rowbytes <= (dtype == WM_F32 ? 2048 : 1024), WM_EUNSUPPORTED);
  WM_REQUIRE((reinterpret_cast<uintptr_t>(query) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(bank) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
             WM_EALIGN);
  const KnnPlan p = make_plan(nq, n, rowbytes, k);
  const size_t cand = (size_t)p.qtiles * p.qt * 32 * p.nslices * p.kt;
  WM_REQUIRE(workspace_bytes >= cand * 8 + (size_t)nq * p.kt * 8, WM_EWORKSPACE);
  float* ps = static_cast<float*>(workspace);
  int* pi = reinterpret_cast<int*>(ps + cand);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  if (dtype == WM_BF16) {
    rc = p.kt == 8 ? dispatch_qt<WM_BF16, 8>(p, query, bank, nq, n, rowbytes, ps, pi, st)
                   : dispatch_qt<WM_BF16, 16>(p, query, bank, nq, n, rowbytes, ps, pi, st);
  } else {
    rc = p.kt == 8 ? dispatch_qt<WM_F32, 8>(p, query, bank, nq, n, rowbytes, ps, pi, st)
                   : dispatch_qt<WM_F32, 16>(p, query, bank, nq, n, rowbytes, ps, pi, st);
  }
  if (rc != WM_OK) return rc;
  if (dtype == WM_BF16)
    return p.kt == 8 ? launch_select<WM_BF16, 8>(p, query, bank, n, d, rowbytes, nq, ps, pi, bank_index_base, k, out_sim, out_idx, st)
                     : launch_select<WM_BF16, 16>(p, query, bank, n, d, rowbytes, nq, ps, pi, bank_index_base, k, out_sim, out_idx, st);
  return p.kt == 8 ? launch_select<WM_F32, 8>(p, query, bank, n, d, rowbytes, nq, ps, pi, bank_index_base, k, out_sim, out_idx, st)
                   : launch_select<WM_F32, 16>(p, query, bank, n, d, rowbytes, nq, ps, pi, bank_index_base, k, out_sim, out_idx, st);
}

// Many query batches, whole calls (streaming kernel + selection kernel) round-robin over the caller's streams, all
// launches queued by this one call: the latency-bound selection kernel of a batch overlaps the streaming kernels of
// the batches queued behind it on the other streams.  No events between the lanes: measured on 811 457 x 128 bf16,
// 64 queries per batch (profiles/r02_experiments.md), ordering the streaming kernels with cross-stream events
// (either chained lane to lane, or all on one stream with the selections behind events) cost more than it gained
// (66 / 72 us per batch against 57 for this form and 63 for one stream).
extern "C" int wm_knn_topk_many(const void* query, const void* bank, int nq, int n, int d, int dtype, int k,
                                int bank_index_base, float* out_sim, int32_t* out_idx, int batch, void* workspaces,
                                size_t workspace_bytes_per_lane, void* const* streams, int n_streams) {
  WM_REQUIRE(query && bank && out_sim && out_idx && workspaces && streams, WM_EINVAL);
  WM_REQUIRE(nq > 0 && batch > 0 && n_streams > 0 && n_streams <= 16, WM_EINVAL);
  WM_REQUIRE(dtype == WM_F32 || dtype == WM_BF16, WM_EUNSUPPORTED);
  const size_t rowbytes = (size_t)d * (dtype == WM_BF16 ? 2 : 4);
  WM_REQUIRE(workspace_bytes_per_lane % 256 == 0, WM_EALIGN);
  int bi = 0;
  for (int o = 0; o < nq; o += batch, ++bi) {
    const int lane = bi % n_streams;
    const int m = nq - o < batch ? nq - o : batch;
    const int rc = wm_knn_topk(static_cast<const uint8_t*>(query) + (size_t)o * rowbytes, bank, m, n, d, dtype, k,
                               bank_index_base, out_sim + (size_t)o * k, out_idx + (size_t)o * k,
                               static_cast<uint8_t*>(workspaces) + (size_t)lane * workspace_bytes_per_lane,
                               workspace_bytes_per_lane, streams[lane]);
    if (rc != WM_OK) return rc;
  }
  return WM_OK;
}

extern "C" int wm_knn_merge(const float* in_sim, const int32_t* in_idx, int parts, int nq, int k,
                            float* out_sim, int32_t* out_idx, void* stream) {
  WM_REQUIRE(in_sim && in_idx && out_sim && out_idx, WM_EINVAL);
  WM_REQUIRE(parts > 0 && nq > 0 && k > 0, WM_EINVAL);
  WM_REQUIRE(k <= 16, WM_EUNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long sq = k, sp = (long long)nq * k;
  if (k <= 8)
    knn_merge_lists<8><<<nq, 64, 0, st>>>(in_sim, in_idx, parts, k, sq, sp, k, out_sim, out_idx);
  else
    knn_merge_lists<16><<<nq, 64, 0, st>>>(in_sim, in_idx, parts, k, sq, sp, k, out_sim, out_idx);
  WM_LAUNCH_CHECK();
  return WM_OK;
}

extern "C" int wm_knn_vote(const float* sim, const int32_t* idx, const int64_t* bank_labels, long long n_labels,
                           int nq, int k, int num_classes, float temperature, int64_t* pred_labels,
                           float* scores, void* stream) {
  WM_REQUIRE(sim && idx && bank_labels && pred_labels, WM_EINVAL);
  WM_REQUIRE(n_labels > 0 && nq > 0 && k > 0 && num_classes > 0 && temperature > 0.f, WM_EINVAL);
  WM_REQUIRE(num_classes <= VOTE_MAX_CLASSES, WM_EUNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  knn_vote_kernel<<<wm_cdiv(nq, 64), 64, 0, st>>>(sim, idx,
                                                  reinterpret_cast<const long long*>(bank_labels), n_labels,
                                                  nq, k, num_classes, temperature,
                                                  reinterpret_cast<long long*>(pred_labels), scores);
  WM_LAUNCH_CHECK();
  return WM_OK;
}
