// Internal interface between gemm.hip (dispatch, C ABI) and gemm_quad.hip (one-wave-per-SIMD kernels).  Not part of the C ABI.
#pragma once
#include "common.h"

struct QuadArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;   // EPI_BIAS / EPI_BIAS_GELU: fp32 [N] input; EPI_DGELU: fp32 [N] OUTPUT receiving the column sums (or null)
  bf16_t* aux;
  long lda, ldb, ldc, ldaux;
  int M, N, K;
  int tiles_m, tiles_n;
  float beta;
  int splitk;          // > 1 (fp32 output only): K cut into slices, slice s stores its partial tile at C + s * slice_stride
  long slice_stride;
  int group_m;
  // paired launch (TN wgrad form): a SECOND problem with the same N and K stacked below the first one's row tiles - tiles [0, tiles_m_split) of the tile rows are
  // problem one, the rest problem two (its own operands and output).  One launch of exactly 256 tiles instead of a 256-tile launch of 192-row tiles plus a
  // split-K launch + reduce (qkv and out-proj weight gradients: 192 + 64 tiles of 256 x 256).  0 = single problem.
  int tiles_m_split = 0;
  // multi-problem launch (TN wgrad form, 256 x 256 tiles): up to four problems C_i[M_i, N_i] = A_i[K, M_i]^T B_i[K, N_i] over the SAME K share one grid (and one
  // K split) - the four few-tile weight gradients of a small DiT block (UniDisc-S: 27 + 9 + 36 + 36 tiles over K = 24 576).  Tiles of problem i are
  // [m_tile_start[i], m_tile_start[i + 1]); 0 = off.
  int nprob = 0;
  int m_tile_start[5] = {0, 0, 0, 0, 0};
  int m_tiles_n[4] = {0, 0, 0, 0};
  const bf16_t* mA[4] = {nullptr, nullptr, nullptr, nullptr};
  const bf16_t* mB[4] = {nullptr, nullptr, nullptr, nullptr};
  void* mC[4] = {nullptr, nullptr, nullptr, nullptr};
  long mlda[4] = {0, 0, 0, 0}, mldb[4] = {0, 0, 0, 0}, mldc[4] = {0, 0, 0, 0};
  const bf16_t* A2 = nullptr;
  const bf16_t* B2 = nullptr;
  void* C2 = nullptr;
  long lda2 = 0, ldb2 = 0, ldc2 = 0;
  unsigned* timeline = nullptr;   // diagnostic build of the asm K loop (make UDM_QUADLOOP=timeline)
};

extern int g_quad_mode;
extern int g_quad_asm;
extern unsigned* g_quad_timeline;
int udm_gemm_cus_available();   // 256, or the cap of udm_gemm_set_cus / UDM_GEMM_CUS: rounds are counted against it
int udm_quad_mode();   // 0 off, 1 auto (shapes that fill the chip), 2 force wherever the shape fits
// TN (wgrad) form: does a quad tile fit (whole tiles, K % 64 == 0)?  *fm receives the tile height / 64 (3, 4 or 5).
bool udm_quad_tn_ok(long M, long N, long K, int* fm);
int udm_quad_launch_tn(const QuadArgs& a, int fm, hipStream_t stream);
// two wgrads C = A^T B, C2 = A2^T B2 with the same N and K in one launch of 256 x 256 tiles (a.M / a.tiles_m_split describe problem one, M2 rows problem two)
int udm_quad_launch_tn_pair(const QuadArgs& a, long M2, hipStream_t stream);
// up to four wgrads over the same K in one launch of 256 x 256 tiles (a.nprob, a.m*: filled by the caller; a.splitk slices per tile)
int udm_quad_launch_tn_multi(const QuadArgs& a, hipStream_t stream);
// NT (forward / dgrad) form
bool udm_quad_nt_ok(long M, long N, long K, int* fm);
int udm_quad_launch_nt(const QuadArgs& a, int fm, int epilogue, int out_f32, hipStream_t stream);
// NN (dgrad from the forward's W shadow) form: C[M, N] bf16 = A[M, K] B[K, N], plain epilogue.  *fm = 3 / 4 / 5 (whole tiles) or -5 (320-row tiles, ragged last tile row)
bool udm_quad_nn_ok(long M, long N, long K, int* fm);
int udm_quad_launch_nn(const QuadArgs& a, int fm, hipStream_t stream);
int udm_quad_launch_nn_f32(const QuadArgs& a, int fm, hipStream_t stream);   // whole tiles (fm = 3 / 4 / 5), fp32 output, a.splitk slices
