// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32) with fused
// epilogues.  Replaces tf.layers.dense / tf.matmul and their gradients on the
// time-aware path (see include/mtam_hip.h for the reference lines).
//
// Tiling: one 256-thread workgroup (4 waves, one per SIMD) owns a 64x64 tile of
// C; each wave owns a 32x32 quadrant and accumulates it in 16 registers with
// one MFMA per two k.  The fp32 MFMA is an exact k-ordered fmaf chain, so the
// result of one output element does not depend on the tile it falls in.
// Operands are staged global -> registers -> LDS in k-major order
// (As[k][m], Bs[k][n]) so that the MFMA fragment reads (lane = m or n) are
// bank-conflict free.  LDS is double buffered: while k-tile kt is multiplied,
// tile kt+1 moves registers -> LDS and tile kt+2's global loads are issued
// (straight-line 16-B loads on the wave-uniform fast path), one barrier per tile.
#include "common.h"
#include "split_bf16.h"
#include <stdlib.h>

#ifndef GEMM_STAMP          // tools/gemm_lab.hip defines these to read where a kernel's cycles go
#define GEMM_STAMP(i)
#define GEMM_STAMP_DECL
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
// LDS row strides for a tile edge of E rows/columns: E + 1 when the source is k-contiguous (transposing
// write), E + 4 when it is m/n-contiguous (row write, 16-B aligned)

struct GemmArgs {
  const float *A, *B;
  float *C;
  const float *bias, *aux_in;
  float *aux_out;
  int M, N, K, lda, ldb, ldc, ld_aux;
  int k_chunk;
  int vecA, vecB;
  int tiles_n;
  // batched form: grid.y = batch index z -> (z / heads, z % heads), element offsets per operand
  int heads;
  long sA0, sA1, sB0, sB1, sC0, sC1;
  // optional second source accumulated into the same output tile (same transposes; no split-K)
  const float *A2, *B2;
  int K2, lda2, ldb2, vec2;
  // C (and aux_in / aux_out / the bias-row matrix, where the epilogue uses them) 16-byte aligned with
  // leading dimensions that are multiples of 4: the epilogue may move whole float4 row pieces
  int vecC;
};

// Source laid out src[r][k] (k contiguous): tile of 64 rows x 32 k.
// FAST: the whole tile is in range and 16-B aligned (decided per block / per k-tile, wave-uniform),
// so the loads are straight-line dwordx4 with nothing between them for the compiler to wait on.
template <bool FAST, int E>
__device__ __forceinline__ void load_kcontig(const float *__restrict__ src, int ld, int rows, int r0,
                                             int k0, int kend, float4 (&reg)[E / 32]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < E / 32; ++i) {
    const int idx = t + 256 * i;
    const int r = r0 + (idx >> 3);
    const int k = k0 + 4 * (idx & 7);
    if (FAST) {
      reg[i] = *reinterpret_cast<const float4 *>(src + (size_t)r * ld + k);
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows) {
        const float *p = src + (size_t)r * ld + k;
        if (k + 0 < kend) v.x = p[0];
        if (k + 1 < kend) v.y = p[1];
        if (k + 2 < kend) v.z = p[2];
        if (k + 3 < kend) v.w = p[3];
      }
      reg[i] = v;
    }
  }
}
template <int E>
__device__ __forceinline__ void store_kcontig(float *__restrict__ S, const float4 (&reg)[E / 32]) {
  constexpr int LD = E + 1;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < E / 32; ++i) {
    const int idx = t + 256 * i;
    const int r = idx >> 3;
    const int k = 4 * (idx & 7);
    S[(k + 0) * LD + r] = reg[i].x;
    S[(k + 1) * LD + r] = reg[i].y;
    S[(k + 2) * LD + r] = reg[i].z;
    S[(k + 3) * LD + r] = reg[i].w;
  }
}

// Source laid out src[k][c] (c contiguous): tile of 32 k x E columns.
template <bool FAST, int E>
__device__ __forceinline__ void load_ccontig(const float *__restrict__ src, int ld, int cols, int c0,
                                             int k0, int kend, float4 (&reg)[E / 32]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < E / 32; ++i) {
    const int idx = t + 256 * i;
    const int k = k0 + idx / (E / 4);
    const int c = c0 + 4 * (idx % (E / 4));
    if (FAST) {
      reg[i] = *reinterpret_cast<const float4 *>(src + (size_t)k * ld + c);
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kend) {
        const float *p = src + (size_t)k * ld + c;
        if (c + 0 < cols) v.x = p[0];
        if (c + 1 < cols) v.y = p[1];
        if (c + 2 < cols) v.z = p[2];
        if (c + 3 < cols) v.w = p[3];
      }
      reg[i] = v;
    }
  }
}
template <int E>
__device__ __forceinline__ void store_ccontig(float *__restrict__ S, const float4 (&reg)[E / 32]) {
  constexpr int LD = E + 4;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < E / 32; ++i) {
    const int idx = t + 256 * i;
    const int k = idx / (E / 4);
    const int c = 4 * (idx % (E / 4));
    *reinterpret_cast<float4 *>(&S[k * LD + c]) = reg[i];
  }
}

template <int E> constexpr int tile_floats() { return BK * (E + 4); }   // one operand tile in LDS
constexpr int TILE_FLOATS = tile_floats<64>();

// ---- split-bf16 operands (X3 = true in gemm_tile): fp32 products from six bf16 MFMAs (csrc/split_bf16.h), 2.7 x the
// matrix rate of v_mfma_f32_32x32x2_f32.  A staged 64 x 32 operand tile becomes THREE bf16 images of 4 KB, written as
// 8-byte pieces in the orientation the source has and read as the MFMA wants them:
//   source k-contiguous (src[row][k]):  image [64 rows][32 k] (64-byte rows); a fragment (row r, 8 k) is one ds_read_b128
//   source row-contiguous (src[k][c]):  image [32 k][64 c] (128-byte rows); a fragment (column c, 8 k) is gathered DOWN
//                                        the image by two ds_read_b64_tr_b16
// with 16-byte chunks swizzled so that the writes, the row reads and the transposed reads are all conflict-free.
using namespace split_bf16;
constexpr int X3_IMG = 4096;                      // bytes per image
constexpr int X3_TILE_FLOATS = 3 * X3_IMG / 4;    // one operand tile (three images), in floats
__device__ __forceinline__ int kimg_off(int row, int ch) { return 64 * row + 16 * (ch ^ ((row >> 2) & 3)); }     // ch 0..3
__device__ __forceinline__ int cimg_off(int k, int ch) { return 128 * k + 16 * (ch ^ (((k >> 1) & 1) << 2)); }  // ch 0..7
// registers of load_kcontig<., 64> -> images
__device__ __forceinline__ void store_kcontig_x3(unsigned char *img, const float4 (&reg)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 256 * i, r = idx >> 3, piece = idx & 7;
    const float x[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
    bf16x4 q[3];
    split4(x, q);
    unsigned char *const dst = img + kimg_off(r, piece >> 1) + 8 * (piece & 1);
#pragma unroll
    for (int im = 0; im < 3; ++im) *reinterpret_cast<bf16x4 *>(dst + im * X3_IMG) = q[im];
  }
}
// registers of load_ccontig<., 64> -> images
__device__ __forceinline__ void store_ccontig_x3(unsigned char *img, const float4 (&reg)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 256 * i, k = idx >> 4, piece = idx & 15;
    const float x[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
    bf16x4 q[3];
    split4(x, q);
    unsigned char *const dst = img + cimg_off(k, piece >> 1) + 8 * (piece & 1);
#pragma unroll
    for (int im = 0; im < 3; ++im) *reinterpret_cast<bf16x4 *>(dst + im * X3_IMG) = q[im];
  }
}

// One wave's 32x32 accumulator -> C with the fused epilogue.
// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
// Epilogues that read (C, bias rows, aux_in) issue ALL their loads before the first store: written
// as load-compute-store per row, possible aliasing made the compiler wait out a full memory round
// trip per row (16 rows: 12,000 cycles for RELU_ADD, 36,000 for ACCUM2_MASK, tools/gemm_lab.hip).
template <int EPI>
__device__ __forceinline__ float store_acc(const GemmArgs &p, const f32x16 &acc, int row0, int gn, int lane) {
  float sq = 0.f;          // STORE_SQ: this lane's sum of acc^2 over its in-range elements
  if (gn >= p.N) return sq;
  constexpr bool READ_C = EPI == MTAM_EPI_ACCUM || EPI == MTAM_EPI_ACCUM_MASK || EPI == MTAM_EPI_ACCUM2_MASK;
  constexpr bool READ_AUX = EPI == MTAM_EPI_RELU_ADD || EPI == MTAM_EPI_ACCUM_MASK || EPI == MTAM_EPI_ACCUM2_MASK;
  constexpr bool READ_BIAS2 = EPI == MTAM_EPI_ACCUM2_MASK;
  float bias = 0.f;
  if (EPI == MTAM_EPI_BIAS || EPI == MTAM_EPI_BIAS_RELU) bias = p.bias[gn];
  const int rbase = row0 + 4 * (lane >> 5);
  float cv[16], av[16], bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);      // clamped: ragged rows are not stored below
    if (READ_C) cv[r] = p.C[(size_t)gm * p.ldc + gn];
    if (READ_AUX) av[r] = p.aux_in[(size_t)gm * p.ld_aux + gn];
    if (READ_BIAS2) bv[r] = p.bias[(size_t)gm * p.ld_aux + gn];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = rbase + (r & 3) + 8 * (r >> 2);
    if (gm >= p.M || gn >= p.N) continue;
    float v = acc[r];
    float *c = p.C + (size_t)gm * p.ldc + gn;
    const size_t o = (size_t)gm * p.ld_aux + gn;
    if (EPI == MTAM_EPI_STORE) {
      *c = v;
    } else if (EPI == MTAM_EPI_BIAS) {
      *c = v + bias;
    } else if (EPI == MTAM_EPI_BIAS_RELU) {
      *c = fmaxf(v + bias, 0.f);
    } else if (EPI == MTAM_EPI_RELU_ADD) {
      v = fmaxf(v, 0.f);
      p.aux_out[o] = v;
      *c = v + av[r];
    } else if (EPI == MTAM_EPI_ACCUM) {
      *c = cv[r] + v;
    } else if (EPI == MTAM_EPI_ACCUM_MASK) {
      v += cv[r];
      *c = v;
      p.aux_out[o] = (av[r] > 0.f) ? v : 0.f;
    } else if (EPI == MTAM_EPI_ACCUM2_MASK) {
      v += cv[r] + bv[r];
      *c = v;
      p.aux_out[o] = (av[r] > 0.f) ? v : 0.f;
    } else if (EPI == MTAM_EPI_STORE_SQ) {
      *c = v;
      sq += v * v;
    } else {  // MTAM_EPI_ATOMIC
      atomicAdd(c, v);
    }
  }
  return sq;
}

// One 64x64 output tile over one K slice.  LDS: As[2][TILE_FLOATS], Bs[2][TILE_FLOATS]
// (double buffered: the tile for step kt+1 is written while step kt is multiplied; one barrier per step).
// MW = 32-row blocks per wave: 1 -> the 64x64 workgroup tile; 2 -> a 128x64 tile whose waves own 64x32
// (two accumulators: twice the MFMA work per staged byte and two independent chains that issue back
// to back) for problems with thousands of tiles, e.g. the [B, V] scoring products at V >= 1 M.
// The same epilogues in ROW layout: the tile goes through a wave-private LDS scratch (32 x 36 floats) and
// comes back as 4 float4 row pieces per lane (lane -> row 8 i + (lane >> 3), columns 4 (lane & 7) ..), so
// that every global access of the epilogue is 16 bytes wide: 4 store instructions per tile instead of 16.
// The 4-byte stores of the column layout are store-ISSUE bound (tools: a 22.9 MB epilogue cost 12.5 us as
// 4-byte stores, 3.7 us as 16-byte stores).  Needs the full 32-column block in range and p.vecC.
constexpr int T_PITCH = 36;
template <int EPI>
__device__ __forceinline__ float store_acc_rows(const GemmArgs &p, const f32x16 &acc, float *scratch, int row0,
                                                int col0, int lane) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  constexpr bool READ_C = EPI == MTAM_EPI_ACCUM || EPI == MTAM_EPI_ACCUM_MASK || EPI == MTAM_EPI_ACCUM2_MASK;
  constexpr bool READ_AUX = EPI == MTAM_EPI_RELU_ADD || EPI == MTAM_EPI_ACCUM_MASK || EPI == MTAM_EPI_ACCUM2_MASK;
  constexpr bool READ_BIAS2 = EPI == MTAM_EPI_ACCUM2_MASK;
  constexpr bool WRITE_AUX = EPI == MTAM_EPI_RELU_ADD || EPI == MTAM_EPI_ACCUM_MASK || EPI == MTAM_EPI_ACCUM2_MASK;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 16; ++q) scratch[((q & 3) + 8 * (q >> 2) + 4 * h) * T_PITCH + r] = acc[q];
  const int gn = col0 + 4 * (lane & 7);
  v4 bias = {0.f, 0.f, 0.f, 0.f};
  if (EPI == MTAM_EPI_BIAS || EPI == MTAM_EPI_BIAS_RELU) bias = *reinterpret_cast<const v4 *>(p.bias + gn);
  v4 t[4], cv[4], av[4], bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {                       // every load first (clamped rows: ragged rows are not stored)
    const int lr = 8 * i + (lane >> 3);
    const size_t gm = min(row0 + lr, p.M - 1);
    t[i] = *reinterpret_cast<const v4 *>(scratch + lr * T_PITCH + 4 * (lane & 7));
    if (READ_C) cv[i] = *reinterpret_cast<const v4 *>(p.C + gm * p.ldc + gn);
    if (READ_AUX) av[i] = *reinterpret_cast<const v4 *>(p.aux_in + gm * p.ld_aux + gn);
    if (READ_BIAS2) bv[i] = *reinterpret_cast<const v4 *>(p.bias + gm * p.ld_aux + gn);
  }
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = row0 + 8 * i + (lane >> 3);
    if (gm >= p.M) continue;
    v4 v = t[i], aux = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (EPI == MTAM_EPI_BIAS) {
        v[c] += bias[c];
      } else if (EPI == MTAM_EPI_BIAS_RELU) {
        v[c] = fmaxf(v[c] + bias[c], 0.f);
      } else if (EPI == MTAM_EPI_RELU_ADD) {
        aux[c] = fmaxf(v[c], 0.f);
        v[c] = aux[c] + av[i][c];
      } else if (EPI == MTAM_EPI_ACCUM) {
        v[c] += cv[i][c];
      } else if (EPI == MTAM_EPI_ACCUM_MASK) {
        v[c] += cv[i][c];
        aux[c] = (av[i][c] > 0.f) ? v[c] : 0.f;
      } else if (EPI == MTAM_EPI_ACCUM2_MASK) {
        v[c] += cv[i][c] + bv[i][c];
        aux[c] = (av[i][c] > 0.f) ? v[c] : 0.f;
      } else if (EPI == MTAM_EPI_STORE_SQ) {
        sq += v[c] * v[c];
      }
    }
    *reinterpret_cast<v4 *>(p.C + (size_t)gm * p.ldc + gn) = v;
    if (WRITE_AUX) *reinterpret_cast<v4 *>(p.aux_out + (size_t)gm * p.ld_aux + gn) = aux;
  }
  return sq;
}

template <int MW, bool TA, bool TB, int EPI, bool X3 = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs &p, float *As, float *Bs, int tile, int kslice) {
  static_assert(!X3 || MW == 1, "split-bf16 operands: 64 x 64 tiles only");
  constexpr int EA = 64 * MW;                   // A-tile rows
  constexpr int LDA_S = TA ? EA + 4 : EA + 1;
  constexpr int LDB_S = TB ? BN + 1 : BN + 4;
  constexpr int A_FLOATS = X3 ? X3_TILE_FLOATS : tile_floats<EA>(), B_FLOATS = X3 ? X3_TILE_FLOATS : tile_floats<BN>();
  constexpr int BM = EA;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  // the current source of the product: (A, B, K); a second source (A2, B2, K2) is accumulated into the same
  // tile afterwards (C = A B + A2 B2: two matmul gradients that meet in one tensor, one launch)
  const float *Asrc = p.A, *Bsrc = p.B;
  int lda = p.lda, ldb = p.ldb;
  int kbeg = kslice * p.k_chunk;
  int kend = min(p.K, kbeg + p.k_chunk);
  // wave-uniform: whole tile rows/columns in range and vector loads legal
  bool fullA = p.vecA && (m0 + BM <= p.M);
  bool fullB = p.vecB && (n0 + BN <= p.N);

  f32x16 acc, acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = acc2[i] = 0.f;

  float4 ra[EA / 32], rb[BN / 32];
  auto load_tiles = [&](int k0) __attribute__((always_inline)) {
    const bool fullK = (k0 + BK <= kend);
    if (fullA && fullK) {
      if (TA) load_ccontig<true, EA>(Asrc, lda, p.M, m0, k0, kend, ra);
      else    load_kcontig<true, EA>(Asrc, lda, p.M, m0, k0, kend, ra);
    } else {
      if (TA) load_ccontig<false, EA>(Asrc, lda, p.M, m0, k0, kend, ra);
      else    load_kcontig<false, EA>(Asrc, lda, p.M, m0, k0, kend, ra);
    }
    if (fullB && fullK) {
      if (TB) load_kcontig<true, BN>(Bsrc, ldb, p.N, n0, k0, kend, rb);
      else    load_ccontig<true, BN>(Bsrc, ldb, p.N, n0, k0, kend, rb);
    } else {
      if (TB) load_kcontig<false, BN>(Bsrc, ldb, p.N, n0, k0, kend, rb);
      else    load_ccontig<false, BN>(Bsrc, ldb, p.N, n0, k0, kend, rb);
    }
  };
  auto store_set = [&](int buf, const float4 (&qa)[EA / 32], const float4 (&qb)[BN / 32]) __attribute__((always_inline)) {
    float *a = As + buf * A_FLOATS, *b = Bs + buf * B_FLOATS;
    if constexpr (X3) {
      unsigned char *ai = reinterpret_cast<unsigned char *>(a), *bi = reinterpret_cast<unsigned char *>(b);
      if (TA) store_ccontig_x3(ai, qa); else store_kcontig_x3(ai, qa);
      if (TB) store_kcontig_x3(bi, qb); else store_ccontig_x3(bi, qb);
    } else {
      if (TA) store_ccontig<EA>(a, qa); else store_kcontig<EA>(a, qa);
      if (TB) store_kcontig<BN>(b, qb); else store_ccontig<BN>(b, qb);
    }
  };

  for (int src = 0; src < (p.A2 ? 2 : 1); ++src) {
  if (src == 1) {
    Asrc = p.A2; Bsrc = p.B2; lda = p.lda2; ldb = p.ldb2;
    kbeg = 0; kend = p.K2;
    fullA = p.vec2 && (m0 + BM <= p.M);
    fullB = p.vec2 && (n0 + BN <= p.N);
  }
  if (kbeg < kend) {
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int a_off = (lane >> 5) * LDA_S + wm * (32 * MW) + (lane & 31);
    const int b_off = (lane >> 5) * LDB_S + wn * 32 + (lane & 31);
    GEMM_STAMP_DECL;
    // split-bf16 fragment addresses (bytes into an operand's first image): row-read or transposed-read form
    const int cgl = (lane >> 4) & 1, tql = (lane >> 2) & 3, tpl = lane & 3, hl = lane >> 5, rl = lane & 31;
    int xa[2][2], xb[2][2];       // [k-step][half]; the row-read form uses [.][0] only
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (TA) {
        xa[ks][0] = cimg_off(16 * ks + 8 * hl + tql, 4 * wm + 2 * cgl + (tpl >> 1)) + 8 * (tpl & 1);
        xa[ks][1] = cimg_off(16 * ks + 8 * hl + 4 + tql, 4 * wm + 2 * cgl + (tpl >> 1)) + 8 * (tpl & 1);
      } else {
        xa[ks][0] = xa[ks][1] = kimg_off(32 * wm + rl, 2 * ks + hl);
      }
      if (!TB) {
        xb[ks][0] = cimg_off(16 * ks + 8 * hl + tql, 4 * wn + 2 * cgl + (tpl >> 1)) + 8 * (tpl & 1);
        xb[ks][1] = cimg_off(16 * ks + 8 * hl + 4 + tql, 4 * wn + 2 * cgl + (tpl >> 1)) + 8 * (tpl & 1);
      } else {
        xb[ks][0] = xb[ks][1] = kimg_off(32 * wn + rl, 2 * ks + hl);
      }
    }
    auto multiply_x3 = [&](int kt) __attribute__((always_inline)) {
      const unsigned char *ai = reinterpret_cast<const unsigned char *>(As + (kt & 1) * A_FLOATS);
      const unsigned char *bi = reinterpret_cast<const unsigned char *>(Bs + (kt & 1) * B_FLOATS);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Tri fa, fb;
#pragma unroll
        for (int im = 0; im < 3; ++im) {
          if (TA) fa.t[im] = lds_tr8(ai + im * X3_IMG + xa[ks][0], ai + im * X3_IMG + xa[ks][1]);
          else    fa.t[im] = *reinterpret_cast<const bf16x8 *>(ai + im * X3_IMG + xa[ks][0]);
          if (!TB) fb.t[im] = lds_tr8(bi + im * X3_IMG + xb[ks][0], bi + im * X3_IMG + xb[ks][1]);
          else     fb.t[im] = *reinterpret_cast<const bf16x8 *>(bi + im * X3_IMG + xb[ks][0]);
        }
        acc = mfma6(fa, fb, acc);
      }
      GEMM_STAMP(2);
    };
    auto multiply = [&](int kt) __attribute__((always_inline)) {
      if constexpr (X3) {
        multiply_x3(kt);
        return;
      }
      const float *a_s = As + (kt & 1) * A_FLOATS + a_off;
      const float *b_s = Bs + (kt & 1) * B_FLOATS + b_off;
      float fa[BK / 2], fa2[BK / 2], fb[BK / 2];
#pragma unroll
      for (int q = 0; q < BK / 2; ++q) {
        fa[q] = a_s[2 * q * LDA_S];
        if (MW == 2) fa2[q] = a_s[2 * q * LDA_S + 32];
        fb[q] = b_s[2 * q * LDB_S];
      }
      GEMM_STAMP(1);
#pragma unroll
      for (int q = 0; q < BK / 2; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[q], acc, 0, 0, 0);
        if (MW == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa2[q], fb[q], acc2, 0, 0, 0);
      }
      GEMM_STAMP(2);
    };
    if (fullA && fullB && (kend - kbeg) % BK == 0) {
      // Every tile in range: a straight-line loop body.  The 16 MFMAs of a k-tile are one dependent
      // chain (a new one issues every ~84 cycles, tools/gemm_lab.hip), so the next tile's LDS writes
      // and the global loads of the tile after it are placed INTO those gaps instead of after the
      // chain, where a lone wave per SIMD ran them un-overlapped (1,100 of 2,900 cycles per k-tile).
      {
        // (A second register set, each tile's loads issued two iterations ahead of its LDS write, measured no
        // faster on the step-sized products -- 28.7 against 27.0 us for the weight gradients: a k-tile of the
        // split path is paced by its own chain fragment reads -> 12 MFMAs -> split + LDS writes -> barrier with
        // one wave per SIMD, tools/gemm_lab.hip, not by the round trip of its operands.)
        // ds_write2_b32 pairs / ds_write_b128 per k-tile
        constexpr int N_DSW = (TA ? 2 : 4) * MW + (TB ? 4 : 2);
        auto load_one = [&](int k0) __attribute__((always_inline)) {
          if (TA) load_ccontig<true, EA>(Asrc, lda, p.M, m0, k0, kend, ra);
          else    load_kcontig<true, EA>(Asrc, lda, p.M, m0, k0, kend, ra);
          if (TB) load_kcontig<true, BN>(Bsrc, ldb, p.N, n0, k0, kend, rb);
          else    load_ccontig<true, BN>(Bsrc, ldb, p.N, n0, k0, kend, rb);
        };
        load_one(kbeg);
        store_set(0, ra, rb);
        if (nk > 1) load_one(kbeg + BK);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 < nk; ++kt) {
          GEMM_STAMP(0);
          multiply(kt);
          store_set((kt + 1) & 1, ra, rb);
          load_one(kbeg + (kt + 2) * BK);
#pragma unroll
          for (int i = 0; i < (X3 ? 0 : BK / 2); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MW, 0);                      // one MFMA per chain
            if (i >= 2 && i < 2 + N_DSW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            if (i >= 10 && i < 12 + 2 * MW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
          }
          GEMM_STAMP(3);
          __syncthreads();
          GEMM_STAMP(4);
        }
        for (; kt < nk; ++kt) {
          multiply(kt);
          if (kt + 1 < nk) store_set((kt + 1) & 1, ra, rb);
          __syncthreads();
        }
      }
    } else {
      load_tiles(kbeg);
      store_set(0, ra, rb);
      if (nk > 1) load_tiles(kbeg + BK);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
        GEMM_STAMP(0);
        multiply(kt);
        if (kt + 1 < nk) {
          store_set((kt + 1) & 1, ra, rb);                       // registers hold tile kt+1 (loaded one step ago)
          if (kt + 2 < nk) load_tiles(kbeg + (kt + 2) * BK);
        }
        GEMM_STAMP(3);
        __syncthreads();
        GEMM_STAMP(4);
      }
    }
  }

  }   // sources

  float sq;
  if (EPI != MTAM_EPI_ATOMIC && p.vecC && n0 + wn * 32 + 32 <= p.N) {        // wave-uniform
    // the operand tiles are dead (the k-loop ends on a barrier): two waves share each buffer as scratch
    float *scratch = ((wave < 2) ? As : Bs) + (wave & 1) * (32 * T_PITCH);
    sq = store_acc_rows<EPI>(p, acc, scratch, m0 + wm * (32 * MW), n0 + wn * 32, lane);
    if (MW == 2) sq += store_acc_rows<EPI>(p, acc2, scratch, m0 + wm * 64 + 32, n0 + wn * 32, lane);
  } else {
    sq = store_acc<EPI>(p, acc, m0 + wm * (32 * MW), n0 + wn * 32 + (lane & 31), lane);
    if (MW == 2) sq += store_acc<EPI>(p, acc2, m0 + wm * 64 + 32, n0 + wn * 32 + (lane & 31), lane);
  }
  if (EPI == MTAM_EPI_STORE_SQ) {
    sq = wave_sum(sq);
    if (lane == 0) p.aux_out[4 * (size_t)blockIdx.x + wave] = sq;
  }
}

template <int MW, bool TA, bool TB, int EPI, bool X3>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float As[2 * (X3 ? X3_TILE_FLOATS : tile_floats<64 * MW>())];
  __shared__ __attribute__((aligned(16))) float Bs[2 * (X3 ? X3_TILE_FLOATS : TILE_FLOATS)];
  // 1-D tile index (grid.y is capped at 65535; V/64 is not), split-K slice on grid.y
  if (p.heads > 0) {                       // batched: grid.y is the batch index, no split-K
    GemmArgs q = p;
    const long z0 = blockIdx.y / p.heads, z1 = blockIdx.y % p.heads;
    q.A += z0 * p.sA0 + z1 * p.sA1;
    q.B += z0 * p.sB0 + z1 * p.sB1;
    q.C += z0 * p.sC0 + z1 * p.sC1;
    gemm_tile<MW, TA, TB, EPI, X3>(q, As, Bs, blockIdx.x, 0);
    return;
  }
  gemm_tile<MW, TA, TB, EPI, X3>(p, As, Bs, blockIdx.x, blockIdx.y);
}

// Grouped weight-gradient form: up to MTAM_MAX_GROUP independent C += A^T B problems
// (TA, !TB, atomic epilogue) in ONE launch.  blockIdx.x walks the concatenation of every
// problem's (tile, k-slice) list; `first[g]` is the prefix sum of block counts.
struct GroupArgs {
  GemmArgs g[MTAM_MAX_GROUP];
  int first[MTAM_MAX_GROUP + 1];
  int n;
};

// 64 rows x 64 columns of one column-sum job: out[c] += sum over the rows of in[r][c]
__device__ __forceinline__ void colsum_block(const MtamColsumJob &q, int local, float (*part)[64]) {
  const int col_blocks = (q.cols + 63) / 64;
  const int c = (local % col_blocks) * 64 + (threadIdx.x & 63);
  const int rr = threadIdx.x >> 6;
  const int r0 = (local / col_blocks) * 64;
  const int r1 = min(q.rows, r0 + 64);
  float s = 0.f;
  if (c < q.cols) {
    // the 16 rows of this thread are loaded unconditionally (clamped row, masked value) so that they are
    // all in flight together instead of one round trip per row
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = q.in[(size_t)min(r0 + rr + 4 * i, q.rows - 1) * q.ld + c];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (r0 + rr + 4 * i < r1) ? v[i] : 0.f;
  }
  part[rr][threadIdx.x & 63] = s;
  __syncthreads();
  if (rr == 0 && c < q.cols) {
    s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    atomicAdd(q.out + c, s);
  }
}

struct ColsumGroup {
  MtamColsumJob j[MTAM_MAX_GROUP];
  int first[MTAM_MAX_GROUP + 1];
  int n;
};

// Every weight gradient (grouped C += A^T B) AND every bias-like gradient (column sums) of the step in
// ONE launch: workgroups [0, gemm_blocks) walk the GEMM (tile, k-slice) units, the rest the column-sum
// blocks.  tf.gradients w.r.t. kernels and biases, Model/base_model.py:292.
struct WeightGradArgs {
  GroupArgs g;
  ColsumGroup c;
  int gemm_blocks;
  int xcd_order;      // 1: blocks walk the group's order XCD by XCD (xcd_swizzle); MTAM_WGRAD_XCD=0 turns it off
};

// XCD-aware order of a grouped split-K launch.  Blocks are dealt round-robin over the 8 XCDs (blocks b and b + 8 share
// one, each XCD has its own L2), and the group's natural order is problem-major, k-slice-major, tile-minor: the 12-24
// tiles of ONE k-slice -- which read the same rows of both operands -- landed on 8 different L2s, every one of which
// fetched its own copy: 126 MB per launch at the fabric for ~40 MB of operands (profiles/r03_pmc_step_v1.json), which
// at the HBM / Infinity-Cache rate IS the launch's 25 us.  Swizzled, an XCD works through a CONTIGUOUS range of that
// order -- a few whole k-slices -- and its L2 serves the re-reads (cdna_hip_programming.md T1, bijective form: the
// blocks past the last multiple of 8 keep their place).  Speed only: any placement gives the same sums.
__device__ __forceinline__ int xcd_swizzle(int bid, int n) {
  const int n8 = n & ~7;
  return bid < n8 ? (bid & 7) * (n8 >> 3) + (bid >> 3) : bid;
}

template <bool X3>
__global__ __launch_bounds__(256) void weight_grads_kernel(WeightGradArgs wa) {
  __shared__ __attribute__((aligned(16))) float As[2 * (X3 ? X3_TILE_FLOATS : TILE_FLOATS)];
  __shared__ __attribute__((aligned(16))) float Bs[2 * (X3 ? X3_TILE_FLOATS : TILE_FLOATS)];
  if ((int)blockIdx.x >= wa.gemm_blocks) {
    const int bid = blockIdx.x - wa.gemm_blocks;
    int j = 0;
    while (j + 1 < wa.c.n && bid >= wa.c.first[j + 1]) ++j;
    colsum_block(wa.c.j[j], bid - wa.c.first[j], reinterpret_cast<float (*)[64]>(As));
    return;
  }
  const GroupArgs &ga = wa.g;
  const int bid = wa.xcd_order ? xcd_swizzle(blockIdx.x, wa.gemm_blocks) : (int)blockIdx.x;
  int g = 0;
  while (g + 1 < ga.n && bid >= ga.first[g + 1]) ++g;
  const GemmArgs &p = ga.g[g];
  const int local = bid - ga.first[g];
  const int tiles = p.tiles_n * ((p.M + BM - 1) / BM);
  gemm_tile<1, true, false, MTAM_EPI_ATOMIC, X3>(p, As, Bs, local % tiles, local / tiles);
}

template <bool X3>
__global__ __launch_bounds__(256) void gemm_tn_atomic_grouped_kernel(GroupArgs ga) {
  __shared__ __attribute__((aligned(16))) float As[2 * (X3 ? X3_TILE_FLOATS : TILE_FLOATS)];
  __shared__ __attribute__((aligned(16))) float Bs[2 * (X3 ? X3_TILE_FLOATS : TILE_FLOATS)];
  int g = 0;
  while (g + 1 < ga.n && (int)blockIdx.x >= ga.first[g + 1]) ++g;
  const GemmArgs &p = ga.g[g];
  const int local = blockIdx.x - ga.first[g];
  const int tiles = p.tiles_n * ((p.M + BM - 1) / BM);
  gemm_tile<1, true, false, MTAM_EPI_ATOMIC, X3>(p, As, Bs, local % tiles, local / tiles);
}

// MTAM_GEMM_SPLIT=0: every product on v_mfma_f32_32x32x2_f32 (read once)
bool split_enabled() {
  static const bool on = [] {
    const char *e = getenv("MTAM_GEMM_SPLIT");
    return !(e && e[0] == '0');
  }();
  return on;
}

template <int MW, bool TA, bool TB, bool X3>
void launch_epi(int epi, dim3 grid, hipStream_t s, const GemmArgs &a) {
  switch (epi) {
    case MTAM_EPI_STORE: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_STORE, X3>), grid, dim3(256), 0, s, a); break;
    case MTAM_EPI_BIAS: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_BIAS, X3>), grid, dim3(256), 0, s, a); break;
    case MTAM_EPI_BIAS_RELU: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_BIAS_RELU, X3>), grid, dim3(256), 0, s, a); break;
    case MTAM_EPI_RELU_ADD: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_RELU_ADD, X3>), grid, dim3(256), 0, s, a); break;
    case MTAM_EPI_ACCUM: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_ACCUM, X3>), grid, dim3(256), 0, s, a); break;
    case MTAM_EPI_ACCUM_MASK: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_ACCUM_MASK, X3>), grid, dim3(256), 0, s, a); break;
    case MTAM_EPI_ACCUM2_MASK: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_ACCUM2_MASK, X3>), grid, dim3(256), 0, s, a); break;
    case MTAM_EPI_STORE_SQ: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_STORE_SQ, X3>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((gemm_f32_kernel<MW, TA, TB, MTAM_EPI_ATOMIC, X3>), grid, dim3(256), 0, s, a); break;
  }
}
template <int MW, bool X3>
void launch_trans(int trans_a, int trans_b, int epi, dim3 grid, hipStream_t s, const GemmArgs &a) {
  if (trans_a) {
    if (trans_b) launch_epi<MW, true, true, X3>(epi, grid, s, a);
    else         launch_epi<MW, true, false, X3>(epi, grid, s, a);
  } else {
    if (trans_b) launch_epi<MW, false, true, X3>(epi, grid, s, a);
    else         launch_epi<MW, false, false, X3>(epi, grid, s, a);
  }
}

// 128x64 tiles once a problem has this many of them (and at least 128 rows): enough workgroups to fill
// the chip several times over, so the larger tile's higher MFMA work per staged byte wins.
constexpr long TALL_TILE_MIN_UNITS = 2048;
bool use_tall_tile(int M, int N, long batches_or_slices) {
  return M >= 128 && (long)((M + 127) / 128) * ((N + BN - 1) / BN) * batches_or_slices >= TALL_TILE_MIN_UNITS;
}

}  // namespace

extern "C" int mtam_gemm_sq_partials(int M, int N) {
  const int bm = use_tall_tile(M, N, 1) ? 128 : BM;
  return 4 * ((M + bm - 1) / bm) * ((N + BN - 1) / BN);
}

static int gemm_impl(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda, const float *B,
                     int ldb, int K2, const float *A2, int lda2, const float *B2, int ldb2, float *C, int ldc,
                     int epilogue, const float *bias, const float *aux_in, float *aux_out, int ld_aux,
                     int split_k, void *stream) {
  MTAM_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: M, N, K must be positive (got %d %d %d)", M, N, K);
  MTAM_CHECK_ARG(A && B && C, "gemm: null operand");
  const bool x3 = (epilogue & MTAM_GEMM_SPLIT_BF16) != 0 && split_enabled() &&
                  (epilogue & ~MTAM_GEMM_SPLIT_BF16) != MTAM_EPI_STORE_SQ;      // (STORE_SQ partials are sized for the fp32 tiling)
  epilogue &= ~MTAM_GEMM_SPLIT_BF16;
  MTAM_CHECK_ARG(epilogue >= MTAM_EPI_STORE && epilogue <= MTAM_EPI_STORE_SQ, "gemm: bad epilogue %d", epilogue);
  if (epilogue == MTAM_EPI_STORE_SQ) MTAM_CHECK_ARG(aux_out != nullptr, "gemm: STORE_SQ needs aux_out for the partial sums");
  MTAM_CHECK_ARG(lda >= (trans_a ? M : K), "gemm: lda %d too small", lda);
  MTAM_CHECK_ARG(ldb >= (trans_b ? K : N), "gemm: ldb %d too small", ldb);
  MTAM_CHECK_ARG(ldc >= N, "gemm: ldc %d too small", ldc);
  if (epilogue == MTAM_EPI_BIAS || epilogue == MTAM_EPI_BIAS_RELU || epilogue == MTAM_EPI_ACCUM2_MASK)
    MTAM_CHECK_ARG(bias != nullptr, "gemm: bias epilogue without bias");
  if (epilogue == MTAM_EPI_RELU_ADD || epilogue == MTAM_EPI_ACCUM_MASK || epilogue == MTAM_EPI_ACCUM2_MASK)
    MTAM_CHECK_ARG(aux_in && aux_out && ld_aux >= N, "gemm: aux epilogue needs aux_in/aux_out/ld_aux");
  if (split_k < 1) split_k = 1;
  MTAM_CHECK_ARG(split_k == 1 || epilogue == MTAM_EPI_ATOMIC, "gemm: split_k > 1 needs the atomic epilogue");
  int k_chunk = (K + split_k - 1) / split_k;
  k_chunk = ((k_chunk + BK - 1) / BK) * BK;
  split_k = (K + k_chunk - 1) / k_chunk;
  const bool tall = !x3 && use_tall_tile(M, N, split_k);
  const long gx = (N + BN - 1) / BN, gy = tall ? (M + 127) / 128 : (M + BM - 1) / BM;
  MTAM_CHECK_ARG(gx * gy <= 0x7fffffffL && split_k <= 65535, "gemm: grid too large");

  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux_in = aux_in; a.aux_out = aux_out;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ld_aux = ld_aux;
  a.k_chunk = k_chunk;
  a.vecA = (lda % 4 == 0) && mtam_aligned16(A);
  a.vecB = (ldb % 4 == 0) && mtam_aligned16(B);
  {
    const bool uses_aux = epilogue == MTAM_EPI_RELU_ADD || epilogue == MTAM_EPI_ACCUM_MASK || epilogue == MTAM_EPI_ACCUM2_MASK;
    const bool bias_rows = epilogue == MTAM_EPI_ACCUM2_MASK;
    const bool bias_vec = epilogue == MTAM_EPI_BIAS || epilogue == MTAM_EPI_BIAS_RELU;
    a.vecC = (ldc % 4 == 0) && mtam_aligned16(C) &&
             (!uses_aux || (ld_aux % 4 == 0 && mtam_aligned16(aux_in) && mtam_aligned16(aux_out))) &&
             (!bias_rows || mtam_aligned16(bias)) && (!bias_vec || mtam_aligned16(bias));
  }
  a.tiles_n = (int)gx;
  a.heads = 0;
  a.sA0 = a.sA1 = a.sB0 = a.sB1 = a.sC0 = a.sC1 = 0;
  a.A2 = a.B2 = nullptr; a.K2 = a.lda2 = a.ldb2 = a.vec2 = 0;
  if (A2) {
    MTAM_CHECK_ARG(B2 && K2 > 0 && split_k == 1, "gemm_dual: second source needs B2, K2 > 0 and no split-K");
    MTAM_CHECK_ARG(lda2 >= (trans_a ? M : K2) && ldb2 >= (trans_b ? K2 : N), "gemm_dual: bad leading dimension");
    a.A2 = A2; a.B2 = B2; a.K2 = K2; a.lda2 = lda2; a.ldb2 = ldb2;
    a.vec2 = (lda2 % 4 == 0) && (ldb2 % 4 == 0) && mtam_aligned16(A2) && mtam_aligned16(B2);
  }
  dim3 grid((unsigned)(gx * gy), (unsigned)split_k, 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (tall)    launch_trans<2, false>(trans_a, trans_b, epilogue, grid, s, a);
  else if (x3) launch_trans<1, true>(trans_a, trans_b, epilogue, grid, s, a);
  else         launch_trans<1, false>(trans_a, trans_b, epilogue, grid, s, a);
  MTAM_CHECK_LAUNCH("gemm");
  return MTAM_OK;
}

extern "C" int mtam_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                             const float *B, int ldb, float *C, int ldc, int epilogue,
                             const float *bias, const float *aux_in, float *aux_out, int ld_aux,
                             int split_k, void *stream) {
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, 0, nullptr, 0, nullptr, 0, C, ldc, epilogue, bias,
                   aux_in, aux_out, ld_aux, split_k, stream);
}

extern "C" int mtam_gemm_f32_dual(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                  const float *B, int ldb, int K2, const float *A2, int lda2, const float *B2,
                                  int ldb2, float *C, int ldc, int epilogue, const float *bias,
                                  const float *aux_in, float *aux_out, int ld_aux, void *stream) {
  MTAM_CHECK_ARG(A2 && B2, "gemm_dual: null second source");
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, K2, A2, lda2, B2, ldb2, C, ldc, epilogue, bias,
                   aux_in, aux_out, ld_aux, 1, stream);
}


extern "C" int mtam_gemm_f32_batched(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                     long sA0, long sA1, const float *B, int ldb, long sB0, long sB1, float *C,
                                     int ldc, long sC0, long sC1, int batch0, int batch1, int epilogue,
                                     void *stream) {
  MTAM_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch0 > 0 && batch1 > 0, "gemm_batched: bad sizes");
  MTAM_CHECK_ARG(A && B && C, "gemm_batched: null operand");
  const bool x3 = (epilogue & MTAM_GEMM_SPLIT_BF16) != 0 && split_enabled();
  epilogue &= ~MTAM_GEMM_SPLIT_BF16;
  MTAM_CHECK_ARG(epilogue == MTAM_EPI_STORE || epilogue == MTAM_EPI_ACCUM, "gemm_batched: STORE or ACCUM only");
  MTAM_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "gemm_batched: bad leading dimension");
  MTAM_CHECK_ARG((long)batch0 * batch1 <= 65535, "gemm_batched: at most 65535 problems per launch");
  const long gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = nullptr; a.aux_in = nullptr; a.aux_out = nullptr;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ld_aux = 0;
  a.k_chunk = ((K + BK - 1) / BK) * BK;
  // vector loads need every problem's base 16-byte aligned
  a.vecA = (lda % 4 == 0) && mtam_aligned16(A) && sA0 % 4 == 0 && sA1 % 4 == 0;
  a.vecB = (ldb % 4 == 0) && mtam_aligned16(B) && sB0 % 4 == 0 && sB1 % 4 == 0;
  a.tiles_n = (int)gx;
  a.heads = batch1;
  a.sA0 = sA0; a.sA1 = sA1; a.sB0 = sB0; a.sB1 = sB1; a.sC0 = sC0; a.sC1 = sC1;
  a.A2 = a.B2 = nullptr; a.K2 = a.lda2 = a.ldb2 = a.vec2 = 0;
  a.vecC = 0;
  dim3 grid((unsigned)(gx * gy), (unsigned)(batch0 * batch1), 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (x3) launch_trans<1, true>(trans_a, trans_b, epilogue, grid, s, a);
  else    launch_trans<1, false>(trans_a, trans_b, epilogue, grid, s, a);
  MTAM_CHECK_LAUNCH("gemm_batched");
  return MTAM_OK;
}

static int fill_group(int n, const MtamGemmDesc *d, GroupArgs &ga, int &blocks_out) {
  MTAM_CHECK_ARG(n >= 1 && n <= MTAM_MAX_GROUP && d, "gemm_grouped: 1 <= n <= %d problems", MTAM_MAX_GROUP);
  ga.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const MtamGemmDesc &q = d[i];
    MTAM_CHECK_ARG(q.M > 0 && q.N > 0 && q.K > 0 && q.A && q.B && q.C, "gemm_grouped[%d]: bad problem", i);
    MTAM_CHECK_ARG(q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N, "gemm_grouped[%d]: bad leading dimension", i);
    int split = q.split_k < 1 ? 1 : q.split_k;
    int k_chunk = (q.K + split - 1) / split;
    k_chunk = ((k_chunk + BK - 1) / BK) * BK;
    split = (q.K + k_chunk - 1) / k_chunk;
    GemmArgs &a = ga.g[i];
    a.A = q.A; a.B = q.B; a.C = q.C; a.bias = nullptr; a.aux_in = nullptr; a.aux_out = nullptr;
    a.M = q.M; a.N = q.N; a.K = q.K; a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc; a.ld_aux = 0;
    a.k_chunk = k_chunk;
    a.vecA = (q.lda % 4 == 0) && mtam_aligned16(q.A);
    a.vecB = (q.ldb % 4 == 0) && mtam_aligned16(q.B);
    a.tiles_n = (q.N + BN - 1) / BN;
    a.heads = 0;
    a.sA0 = a.sA1 = a.sB0 = a.sB1 = a.sC0 = a.sC1 = 0;
    a.A2 = a.B2 = nullptr; a.K2 = a.lda2 = a.ldb2 = a.vec2 = 0;
    a.vecC = 0;
    ga.first[i] = blocks;
    blocks += a.tiles_n * ((q.M + BM - 1) / BM) * split;
  }
  ga.first[n] = blocks;
  blocks_out = blocks;
  return MTAM_OK;
}

static int fill_colsums(int n, const MtamColsumJob *jobs, ColsumGroup &cg, int &blocks_out) {
  MTAM_CHECK_ARG(n >= 1 && n <= MTAM_MAX_GROUP && jobs, "colsum_multi: 1 <= n <= %d jobs", MTAM_MAX_GROUP);
  cg.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const MtamColsumJob &q = jobs[i];
    MTAM_CHECK_ARG(q.in && q.out && q.rows > 0 && q.cols > 0 && q.ld >= q.cols, "colsum_multi[%d]: bad job", i);
    cg.j[i] = q;
    cg.first[i] = blocks;
    blocks += ((q.cols + 63) / 64) * ((q.rows + 63) / 64);
  }
  cg.first[n] = blocks;
  blocks_out = blocks;
  return MTAM_OK;
}

extern "C" int mtam_gemm_tn_atomic_grouped(int n, const MtamGemmDesc *d, void *stream) {
  GroupArgs ga;
  int blocks = 0;
  const int rc = fill_group(n, d, ga, blocks);
  if (rc) return rc;
  if (split_enabled())
    hipLaunchKernelGGL(gemm_tn_atomic_grouped_kernel<true>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), ga);
  else
    hipLaunchKernelGGL(gemm_tn_atomic_grouped_kernel<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), ga);
  MTAM_CHECK_LAUNCH("gemm_grouped");
  return MTAM_OK;
}

extern "C" int mtam_weight_grads(int n_gemm, const MtamGemmDesc *d, int n_colsum, const MtamColsumJob *jobs,
                                 void *stream) {
  WeightGradArgs wa;
  int gb = 0, cb = 0;
  int rc = fill_group(n_gemm, d, wa.g, gb);
  if (rc) return rc;
  rc = fill_colsums(n_colsum, jobs, wa.c, cb);
  if (rc) return rc;
  wa.gemm_blocks = gb;
  static const bool xcd_on = [] {
    const char *e = getenv("MTAM_WGRAD_XCD");
    return !(e && e[0] == '0');
  }();
  wa.xcd_order = xcd_on ? 1 : 0;
  if (split_enabled())
    hipLaunchKernelGGL(weight_grads_kernel<true>, dim3(gb + cb), dim3(256), 0, static_cast<hipStream_t>(stream), wa);
  else
    hipLaunchKernelGGL(weight_grads_kernel<false>, dim3(gb + cb), dim3(256), 0, static_cast<hipStream_t>(stream), wa);
  MTAM_CHECK_LAUNCH("weight_grads");
  return MTAM_OK;
}

// ---------------------------------------------------------------- column sums
namespace {
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ in, int rows, int cols,
                                                     int ld, int rows_per_block, float *__restrict__ out) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rr = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  if (c < cols)
    for (int r = r0 + rr; r < r1; r += 4) s += in[(size_t)r * ld + c];
  part[rr][threadIdx.x & 63] = s;
  __syncthreads();
  if (rr == 0 && c < cols) {
    s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    atomicAdd(out + c, s);
  }
}
}  // namespace

extern "C" int mtam_colsum_atomic(const float *in, int rows, int cols, int ld, float *out, void *stream) {
  MTAM_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad arguments");
  const int rows_per_block = 64;
  dim3 grid((cols + 63) / 64, (rows + rows_per_block - 1) / rows_per_block);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, rows, cols,
                     ld, rows_per_block, out);
  MTAM_CHECK_LAUNCH("colsum");
  return MTAM_OK;
}

namespace {
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumGroup cg) {
  __shared__ float part[4][64];
  int g = 0;
  while (g + 1 < cg.n && (int)blockIdx.x >= cg.first[g + 1]) ++g;
  colsum_block(cg.j[g], blockIdx.x - cg.first[g], part);
}
}  // namespace

extern "C" int mtam_colsum_atomic_multi(int n, const MtamColsumJob *jobs, void *stream) {
  ColsumGroup cg;
  int blocks = 0;
  const int rc = fill_colsums(n, jobs, cg, blocks);
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_multi_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), cg);
  MTAM_CHECK_LAUNCH("colsum_multi");
  return MTAM_OK;
}
