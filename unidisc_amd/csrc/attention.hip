// Bidirectional flash attention (forward, dQ, dK/dV) on gfx950 MFMA, bf16 in / fp32 accumulate.
//
// Replaces flash_attn_qkvpacked_func / SDPA / FlexAttention-with-document-mask in the reference
// (models/dit.py:826-829, :843, :784-812; mask semantics model_utils.py:740-771).
//
// Layout idea (wave64, v_mfma_f32_32x32x16_bf16): scores are computed TRANSPOSED, S^T = K·Q^T, so that a
// lane owns ONE query column (lane & 31) and its 16 accumulator registers walk the key rows.  Softmax
// statistics are then lane-local (one __shfl_xor with lane^32 joins the two half-waves), the rescale of
// O^T is lane-local, and P^T feeds the second MFMA as its B operand straight from registers: for a
// 16-key chunk the accumulator registers 8c..8c+7 of a lane are, in order, exactly the 8 k-slots the B
// operand wants once the A operand (V^T) is gathered with the same key permutation
//      k-slot (half h, j) -> key 4h + j (j < 4), 8 + 4h + (j - 4) (j >= 4),
// which is what two ds_read_b64_tr_b16 transposing LDS reads of a row-major V tile deliver.
// K/V (or Q/dO) tiles are staged global -> registers -> XOR-swizzled LDS with the next tile's loads in
// flight under the current tile's MFMAs.  The same swizzle is conflict-free for the ds_read_b128
// fragment reads and for the transposing reads.
#include "attention_common.h"
#include <type_traits>

namespace {

constexpr int BQ = 128;   // query rows per block (4 waves x 32)
constexpr int BKV = 64;   // keys per tile

__global__ __launch_bounds__(256) void attn_doc_ranges_kernel(const int64_t* __restrict__ sid, int* __restrict__ ranges, int L) {
  __shared__ long s_mn[4], s_mx[4];
  __shared__ int s_lo[4], s_hi[4], s_cnt[4], s_pad;
  const int nT = (L + 63) / 64;
  const int b = blockIdx.x / nT, t = blockIdx.x % nT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t* row = sid + (long)b * L;
  long mn = INT64_MAX, mx = -1;
  int pad = 0;   // rows past L do not count as padding: nothing is ever computed for them
  if (tid < 64 && t * 64 + tid < L) {
    const long v = row[t * 64 + tid];
    if (v >= 0) { mn = v; mx = v; } else pad = 1;
  }
  for (int off = 32; off; off >>= 1) {
    mn = min(mn, __shfl_xor(mn, off, 64));
    mx = max(mx, __shfl_xor(mx, off, 64));
    pad |= __shfl_xor(pad, off, 64);
  }
  if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; if (wave == 0) s_pad = pad; }
  __syncthreads();
  mn = s_mn[0];   // only wave 0 read ids
  mx = s_mx[0];
  int lo = L, hi = 0, cnt = 0;
  if (mx >= 0)
    for (int j = tid; j < L; j += 256) {
      const long v = row[j];
      if (v >= mn && v <= mx) { lo = min(lo, j); hi = max(hi, j + 1); ++cnt; }
    }
  for (int off = 32; off; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
    cnt += __shfl_xor(cnt, off, 64);
  }
  if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_cnt[wave] = cnt; }
  __syncthreads();
  if (tid == 0) {
    lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
    hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
    if (hi <= lo) { lo = 0; hi = 0; }
    const bool small = mx >= 0 && mx < (1L << 30);   // ids that do not fit are simply never "uniform"
    int4 r;
    r.x = lo; r.y = hi;
    r.z = (small && !s_pad) ? (int)mn : -1;
    r.w = small ? (int)mx : -2;
    // exact: one document (no padding in the tile) whose rows are exactly the positions [lo, hi) - nothing of another id in between
    const int cnt_all = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const int exact = (r.z >= 0 && r.z == r.w && cnt_all == hi - lo) ? 1 : 0;
    int* out = ranges + ((long)b * nT + t) * DOC_STRIDE;
    *reinterpret_cast<int4*>(out) = r;
    *reinterpret_cast<int4*>(out + 4) = make_int4(exact, 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int D, bool HAS_SID, bool USE_TR, int ABL = 0>   // ABL (timing-only ablations, wrong results): 1 = no softmax VALU, 2 = no MFMAs
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K0 | K1 | V0 | V1 | sidk[2][64]   (one array: keeps LDS-DMA waits exact)
  constexpr int TB = BKV * D * 2;
  long* sid_s = reinterpret_cast<long*>(smem + 4 * TB);
  constexpr int KS = D / 16, DB = D / 32;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid, tile-major: id = tile * (B*H) + (b*H + h).  Blocks are dispatched to XCD id % 8, so every tile of one (b,h) runs on the
  // same XCD and shares its K/V (or Q/dO) through that L2 instead of re-fetching them over the fabric (B*H is a multiple of 8 in practice).
  int bh, tile_x;
  attn_block_to_work(blockIdx.x, a.B * a.H, bh, tile_x);
  const int b = bh / a.H, h = bh % a.H;
  const int qi = tile_x * BQ + wave * 32 + l31;
  const bool q_ok = qi < a.L;
  const long rowbase = (long)b * a.L;

  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = load_frag_global(a.q + (rowbase + qi) * a.q_stride + h * D + ks * 16 + hi * 8, q_ok);
  // Touch the Q fragments here so the compiler's wait for their global loads lands BEFORE the loop: inside it the only outstanding
  // vector-memory operations are the inline-asm LDS-DMA refills, which it must not wait for (see DmaStager).
  // (Sample ids: neither this lane's query id nor the tile's key ids are loaded unless the tile pair needs the element mask - `id_test`, block-uniform,
  // false for every tile pair inside one document.  Kept live / staged per tile they cost 44 spilled registers and a waited load per key tile:
  // packed 4 x 1152 rows 139.6 us against 105.7 us for the same work as separate samples.)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

  f32x16_t oT[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const float c = a.scale_log2;

  const bf16_t* kbase = a.k + rowbase * a.k_stride + h * D;
  const bf16_t* vbase = a.v + rowbase * a.v_stride + h * D;
  using Stg = DmaStager<D, BKV>;
  DmaPlan<D, BKV> plank, planv;
  plank.init(a.k_stride, wave, lane);
  planv.init(a.v_stride, wave, lane);
  const int nkv = (a.L + BKV - 1) / BKV;
  int t_begin = 0, t_end = nkv, blk_id = -1;
  bool doc_pure = false;   // the whole key span of this query block is the block's own document: no tile of the walk needs the element mask (nor its range entry)
  if (HAS_SID) { const DocSpan sp = doc_tile_span(a.doc_ranges, b, a.L, tile_x, nkv); t_begin = sp.t_begin; t_end = sp.t_end; blk_id = sp.blk_id;
                 doc_pure = sp.pure && sp.lo % BKV == 0 && (sp.hi % BKV == 0 || sp.hi == a.L); }   // (key tiles are walked from multiples of BKV: the span must start and end on one)
  if (t_begin < t_end) {
    Stg::issue(kbase, a.k_stride, t_begin * BKV, a.L, smem + (t_begin & 1) * TB, wave, lane);
    Stg::issue(vbase, a.v_stride, t_begin * BKV, a.L, smem + (2 + (t_begin & 1)) * TB, wave, lane);
  }
  // One key tile; ST = t & 1 as a compile-time constant, so that every LDS fragment address of the tile is a hoisted per-lane register plus an
  // IMMEDIATE (with the stage picked at run time each of the 48 fragment reads carried its own v_add_u32: 62 of ~370 instructions per tile and wave
  // in a loop that is bound by instruction issue)
  // IDS = false: the walk of a block whose whole key span is its own document - the tile body is then, instruction for instruction, the one of the kernel
  // without sample ids (with the id code merely branched around, the same single-document work ran 16 % slower: 127 vs 110 us at B = 8, L = 1152)
  auto tile = [&](auto st_c, int t, auto ids_c) {
    constexpr int ST = decltype(st_c)::value;
    constexpr bool IDS = HAS_SID && decltype(ids_c)::value;
    const int kv0 = t * BKV;
    constexpr int st = (ABL & 4) ? 0 : ST;
    const char* Ks = smem + st * TB;
    const char* Vs = smem + (2 + st) * TB;
    const long* sidk = sid_s + st * BKV;
    const bool id_test = IDS && doc_pair_needs_mask(a.doc_ranges, b, a.L, t, blk_id);   // block-uniform
    if (IDS && id_test && tid < BKV) sid_s[st * BKV + tid] = (kv0 + tid < a.L) ? a.sample_ids[rowbase + kv0 + tid] : -2;
    if (!(ABL & 4) || t == 0) {
    wait_all_vmem();   // this wave's share of tile t has landed
    __syncthreads();   // ... and everybody's; all waves are also done with tile t-1, so its stage may be refilled
    }
    if ((ABL & 4) || t + 1 >= t_end) {
    } else if (kv0 + 2 * BKV <= a.L) {   // the next tile is a full one: offsets are precomputed, the tile base is wave-uniform
      plank.issue_full(kbase + (long)(kv0 + BKV) * a.k_stride, smem + (st ^ 1) * TB, wave);
      planv.issue_full(vbase + (long)(kv0 + BKV) * a.v_stride, smem + (2 + (st ^ 1)) * TB, wave);
    } else {
      Stg::issue(kbase, a.k_stride, kv0 + BKV, a.L, smem + (st ^ 1) * TB, wave, lane);
      Stg::issue(vbase, a.v_stride, kv0 + BKV, a.L, smem + (2 + (st ^ 1)) * TB, wave, lane);
    }
    // S^T = K Q^T : [64 keys] x [32 queries per wave].  The two 32-key chains alternate (consecutive MFMAs are independent) and the K
    // fragments are read two k-steps ahead of their MFMAs, so neither the accumulator dependence nor the LDS latency stalls the pipe.
    f32x16_t sT[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[f][r] = 0.f;
    {
      bf16x8_t kq[3][2];
#pragma unroll
      for (int pre = 0; pre < 2; ++pre)
#pragma unroll
        for (int f = 0; f < 2; ++f) kq[pre][f] = lds_frag(Ks, tile_off<D>(f * 32 + l31, pre * 2 + hi));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 2 < KS) {
#pragma unroll
          for (int f = 0; f < 2; ++f) kq[(ks + 2) % 3][f] = lds_frag(Ks, tile_off<D>(f * 32 + l31, (ks + 2) * 2 + hi));
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          if (ABL & 2) { sT[f][ks] += (float)kq[ks % 3][f][0]; continue; }
          sT[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kq[ks % 3][f], qf[ks], sT[f], 0, 0, 0);
        }
      }
    }
    if (id_test || kv0 + BKV > a.L) {
      long sid_q = 0;
      if (IDS && id_test) sid_q = q_ok ? a.sample_ids[rowbase + qi] : -1;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kl = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          bool ok = kv0 + kl < a.L;
          if (IDS) ok = ok && (!id_test || attn_pair_ok(sid_q, sidk[kl]));
          if (!ok) sT[f][r] = -INFINITY;
        }
    }
    float p[2][16];
    if (ABL & 1) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) p[f][r] = sT[f][r];
    } else {
    float mloc = -INFINITY;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sT[f][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // Lazy rescale: m is the REFERENCE exponent of this query, not necessarily its running maximum.  It is only moved (and O^T, l
    // rescaled: 64 + 2 VALU per lane) when some query of the wave saw a score more than 2^8 above its reference; otherwise
    // p = exp2(s c - m c) simply exceeds 1 by at most 2^8, which fp32 sums and bf16 operands hold without loss.  O / l and the
    // LSE m c + log2(l) are exact for any reference.  After the first key tile the branch is practically never taken.
    const bool move = q_ok && (mloc * c > m * c + 8.0f);   // (lanes of query rows past L never vote: their scores depend on the masking path taken)
    if (__builtin_amdgcn_ballot_w64(move) != 0) {
      const float m_new = fmaxf(m, mloc);
      const float alpha = __builtin_amdgcn_exp2f((m - ((m_new == -INFINITY) ? 0.f : m_new)) * c);
      lsum *= alpha;
      m = m_new;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[i][r] *= alpha;
    }
    const float mc = (m == -INFINITY) ? 0.f : m * c;
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[f][r] = __builtin_amdgcn_exp2f(sT[f][r] * c - mc);
        psum += p[f][r];
      }
    lsum += psum;
    }
    // O^T += V^T P^T
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      bf16x8_t pb = pack8(&p[cc >> 1][8 * (cc & 1)]);
#pragma unroll
      for (int i = 0; i < DB; ++i) {
        bf16x8_t vt = lds_frag_T<D, USE_TR>(Vs, cc * 16, i * 32, lane);
        if (ABL & 2) { oT[i][cc] += (float)vt[0] * (float)pb[0]; continue; }
        oT[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt, pb, oT[i], 0, 0, 0);
      }
    }
  };
#define UDM_WALK(IDS_C)                                                                       \
  {                                                                                            \
    int t = t_begin;                                                                           \
    if (t < t_end && (t & 1)) { tile(std::integral_constant<int, 1>{}, t, IDS_C); ++t; }       \
    for (; t + 1 < t_end; t += 2) {                                                            \
      tile(std::integral_constant<int, 0>{}, t, IDS_C);                                        \
      tile(std::integral_constant<int, 1>{}, t + 1, IDS_C);                                    \
    }                                                                                          \
    if (t < t_end) tile(std::integral_constant<int, 0>{}, t, IDS_C);                           \
  }
  if (HAS_SID && !doc_pure) UDM_WALK(std::true_type{}) else UDM_WALK(std::false_type{})
#undef UDM_WALK
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
  if (q_ok && hi == 0) a.lse[((long)b * a.H + h) * a.L + qi] = ltot > 0.f ? __builtin_fmaf(m, c, log2f(ltot)) : INFINITY;
  if constexpr (D == 128) {
    if (a.out_stride % 8 == 0) {   // block-uniform: whole-row stores through the (now idle) K / V stages
      __syncthreads();
      const int q0 = tile_x * BQ + wave * 32;
      store_rows_via_lds_d128(smem + wave * 8192, oT, inv, a.out + (rowbase + q0) * a.out_stride + h * D, a.out_stride, a.L - q0, lane);
      return;
    }
  }
  if (q_ok) {
    bf16_t* op = a.out + (rowbase + qi) * a.out_stride + h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = i * 32 + 8 * rg + 4 * hi;
        *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(oT[i][rg * 4] * inv, oT[i][rg * 4 + 1] * inv), pack2bf(oT[i][rg * 4 + 2] * inv, oT[i][rg * 4 + 3] * inv));
      }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, dQ: block owns 128 queries, walks key tiles.  dQ^T = K^T dS^T (lane owns a query column).
// ------------------------------------------------------------------------------------------------
template <int D, bool HAS_SID, bool USE_TR>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K0 | K1 | V0 | V1 | sidk[2][64]
  constexpr int TB = BKV * D * 2;
  long* sid_s = reinterpret_cast<long*>(smem + 4 * TB);
  constexpr int KS = D / 16, DB = D / 32;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid, tile-major: id = tile * (B*H) + (b*H + h).  Blocks are dispatched to XCD id % 8, so every tile of one (b,h) runs on the
  // same XCD and shares its K/V (or Q/dO) through that L2 instead of re-fetching them over the fabric (B*H is a multiple of 8 in practice).
  int bh, tile_x;
  attn_block_to_work(blockIdx.x, a.B * a.H, bh, tile_x);
  const int b = bh / a.H, h = bh % a.H;
  const int qi = tile_x * BQ + wave * 32 + l31;
  const bool q_ok = qi < a.L;
  const long rowbase = (long)b * a.L;

  bf16x8_t qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = load_frag_global(a.q + (rowbase + qi) * a.q_stride + h * D + ks * 16 + hi * 8, q_ok);
    dof[ks] = load_frag_global(a.dout + (rowbase + qi) * a.do_stride + h * D + ks * 16 + hi * 8, q_ok);
  }
  const long sidx = ((long)b * a.H + h) * a.L + qi;
  float lse_q = q_ok ? a.lse[sidx] : INFINITY;
  // delta = rowsum(dO * O) of this lane's query, from the dO fragments it holds anyway plus one read of the O row: the lane pair (hi = 0, 1) covers
  // the D columns between them.  Stored for the dK/dV kernel, which runs after this one on the same stream (this replaced a separate pass over O, dO).
  float delta_q = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8_t of = load_frag_global(a.o + (rowbase + qi) * a.o_stride + h * D + ks * 16 + hi * 8, q_ok);
#pragma unroll
    for (int e = 0; e < 8; ++e) delta_q += (float)of[e] * (float)dof[ks][e];
  }
  delta_q += __shfl_xor(delta_q, 32, 64);
  if (q_ok && hi == 0) {
    float* dl = const_cast<float*>(a.delta);
    dl[sidx] = delta_q;
    if (a.q_prescaled) {   // negated copies for the wave-specialised dK/dV kernel, whose score chains start from them (attention_dkv_ws.hip): delta | -lse | -delta
      const long plane = (long)a.B * a.H * a.L;
      dl[plane + sidx] = -lse_q;
      dl[2 * plane + sidx] = -delta_q;
    }
  }
  const float c = a.scale_log2;
  // (as in the forward kernel: the compiler's wait for these global loads must sit before the loop, not behind the inline-asm refills)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));
  asm volatile("" : "+v"(lse_q), "+v"(delta_q));

  f32x16_t dqT[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqT[i][r] = 0.f;

  const bf16_t* kbase = a.k + rowbase * a.k_stride + h * D;
  const bf16_t* vbase = a.v + rowbase * a.v_stride + h * D;
  using Stg = DmaStager<D, BKV>;
  DmaPlan<D, BKV> plank, planv;
  plank.init(a.k_stride, wave, lane);
  planv.init(a.v_stride, wave, lane);
  const int nkv = (a.L + BKV - 1) / BKV;
  int t_begin = 0, t_end = nkv, blk_id = -1;
  bool doc_pure = false;   // (see the forward kernel)
  if (HAS_SID) { const DocSpan sp = doc_tile_span(a.doc_ranges, b, a.L, tile_x, nkv); t_begin = sp.t_begin; t_end = sp.t_end; blk_id = sp.blk_id;
                 doc_pure = sp.pure && sp.lo % BKV == 0 && (sp.hi % BKV == 0 || sp.hi == a.L); }
  if (t_begin < t_end) {
    Stg::issue(kbase, a.k_stride, t_begin * BKV, a.L, smem + (t_begin & 1) * TB, wave, lane);
    Stg::issue(vbase, a.v_stride, t_begin * BKV, a.L, smem + (2 + (t_begin & 1)) * TB, wave, lane);
  }
  // (fragment addresses as register + immediate - the forward kernel's compile-time stage index, or per-lane address registers that follow the stage by
  // +-TB per tile - were both tried here: 174-175 us either way against 173-176; not kept)
  for (int t = t_begin; t < t_end; ++t) {
    const int kv0 = t * BKV, st = t & 1;
    const char* Ks = smem + st * TB;
    const char* Vs = smem + (2 + st) * TB;
    const long* sidk = sid_s + st * BKV;
    const bool id_test = HAS_SID && !doc_pure && doc_pair_needs_mask(a.doc_ranges, b, a.L, t, blk_id);   // block-uniform
    if (HAS_SID && id_test && tid < BKV) sid_s[st * BKV + tid] = (kv0 + tid < a.L) ? a.sample_ids[rowbase + kv0 + tid] : -2;   // (see the forward)
    wait_all_vmem();   // this wave's share of tile t has landed
    __syncthreads();   // ... and everybody's; all waves are also done with tile t-1, so its stage may be refilled
    if (t + 1 >= t_end) {
    } else if (kv0 + 2 * BKV <= a.L) {
      plank.issue_full(kbase + (long)(kv0 + BKV) * a.k_stride, smem + (st ^ 1) * TB, wave);
      planv.issue_full(vbase + (long)(kv0 + BKV) * a.v_stride, smem + (2 + (st ^ 1)) * TB, wave);
    } else {
      Stg::issue(kbase, a.k_stride, kv0 + BKV, a.L, smem + (st ^ 1) * TB, wave, lane);
      Stg::issue(vbase, a.v_stride, kv0 + BKV, a.L, smem + (2 + (st ^ 1)) * TB, wave, lane);
    }
    float ds[2][16];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      f32x16_t sT, dpT;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sT[r] = 0.f; dpT[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = tile_off<D>(f * 32 + l31, ks * 2 + hi);
        sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Ks, off), qf[ks], sT, 0, 0, 0);
        dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Vs, off), dof[ks], dpT, 0, 0, 0);
      }
      // per-element masks only for document masks and the ragged last tile: full tiles have no out-of-range keys, and an out-of-range
      // QUERY has lse = +inf (p = 0) and zero operands
      if (id_test || kv0 + BKV > a.L) {
        long sid_q = 0;
        if (HAS_SID && id_test) sid_q = q_ok ? a.sample_ids[rowbase + qi] : -1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kl = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          bool ok = (kv0 + kl < a.L) && q_ok;
          if (HAS_SID) ok = ok && (!id_test || attn_pair_ok(sid_q, sidk[kl]));
          const float pv = ok ? __builtin_amdgcn_exp2f(sT[r] * c - lse_q) : 0.f;
          ds[f][r] = pv * (dpT[r] - delta_q);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[f][r] = __builtin_amdgcn_exp2f(sT[r] * c - lse_q) * (dpT[r] - delta_q);
      }
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      bf16x8_t dsb = pack8(&ds[cc >> 1][8 * (cc & 1)]);
#pragma unroll
      for (int i = 0; i < DB; ++i) {
        bf16x8_t kt = lds_frag_T<D, USE_TR>(Ks, cc * 16, i * 32, lane);
        dqT[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, dsb, dqT[i], 0, 0, 0);
      }
    }
  }
  if constexpr (D == 128) {
    if (a.out_stride % 8 == 0) {   // block-uniform: whole-row stores through the (now idle) K / V stages
      __syncthreads();
      const int q0 = tile_x * BQ + wave * 32;
      store_rows_via_lds_d128(smem + wave * 8192, dqT, a.scale, a.out + (rowbase + q0) * a.out_stride + h * D, a.out_stride, a.L - q0, lane);
      return;
    }
  }
  if (q_ok) {
    bf16_t* op = a.out + (rowbase + qi) * a.out_stride + h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = i * 32 + 8 * rg + 4 * hi;
        *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(dqT[i][rg * 4] * a.scale, dqT[i][rg * 4 + 1] * a.scale),
                                                        pack2bf(dqT[i][rg * 4 + 2] * a.scale, dqT[i][rg * 4 + 3] * a.scale));
      }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, dK/dV: block owns 128 keys (lane owns a key column), walks 64-query tiles.
//   S = Q K^T (regs walk queries), dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
// ------------------------------------------------------------------------------------------------
constexpr int BQT = 64;
// MODE: 1 = dK only, 2 = dV only, 3 = both.  At D = 128 both accumulators (128 registers) plus K/V operands (64) do not fit two
// waves per SIMD, so the backward launches the dK and dV halves separately (each recomputes S; 40 instead of 32 MFMAs per
// 32-query step, but twice the occupancy).
template <int D, bool HAS_SID, bool USE_TR, int MODE, int WAVES>
__global__ __launch_bounds__(256, WAVES) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr bool DO_DK = (MODE & 1) != 0, DO_DV = (MODE & 2) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // Q0 | Q1 | dO0 | dO1 | lse[2][64] | delta[2][64] | sidq[2][64]
  constexpr int TB = BQT * D * 2;
  float* lse_all = reinterpret_cast<float*>(smem + 4 * TB);
  float* delta_all = lse_all + 2 * BQT;
  long* sid_all = reinterpret_cast<long*>(delta_all + 2 * BQT);
  constexpr int KS = D / 16, DB = D / 32;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid, tile-major: id = tile * (B*H) + (b*H + h).  Blocks are dispatched to XCD id % 8, so every tile of one (b,h) runs on the
  // same XCD and shares its K/V (or Q/dO) through that L2 instead of re-fetching them over the fabric (B*H is a multiple of 8 in practice).
  int bh, tile_x;
  attn_block_to_work(blockIdx.x, a.B * a.H, bh, tile_x);
  const int b = bh / a.H, h = bh % a.H;
  const int ki = tile_x * 128 + wave * 32 + l31;
  const bool k_ok = ki < a.L;
  const long rowbase = (long)b * a.L;

  bf16x8_t kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = load_frag_global(a.k + (rowbase + ki) * a.k_stride + h * D + ks * 16 + hi * 8, k_ok);
    if (DO_DK) vf[ks] = load_frag_global(a.v + (rowbase + ki) * a.v_stride + h * D + ks * 16 + hi * 8, k_ok);
  }
  long sid_k = 0;
  if (HAS_SID) sid_k = k_ok ? a.sample_ids[rowbase + ki] : -2;
  const float c = a.scale_log2;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    asm volatile("" : "+v"(kf[ks]));
    if (DO_DK) asm volatile("" : "+v"(vf[ks]));
  }
  if (HAS_SID) asm volatile("" : "+v"(sid_k));

  f32x16_t dkT[DB], dvT[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkT[i][r] = 0.f; dvT[i][r] = 0.f; }

  const bf16_t* qbase = a.q + rowbase * a.q_stride + h * D;
  const bf16_t* dobase = a.dout + rowbase * a.do_stride + h * D;
  const long sbase = ((long)b * a.H + h) * a.L;
  using Stg = DmaStager<D, BQT>;
  const int nq = (a.L + BQT - 1) / BQT;
  int t_begin = 0, t_end = nq, blk_id = -1;
  if (HAS_SID) {
    const DocSpan sp = doc_tile_span(a.doc_ranges, b, a.L, tile_x, nq);
    if (sp.pure && a.doc_pure_split) return;   // block-uniform: this key block belongs to attn_bwd_dkv_ws_kernel (launched beside this one)
    t_begin = sp.t_begin; t_end = sp.t_end; blk_id = sp.blk_id;
  }
  if (t_begin < t_end) {
    Stg::issue(qbase, a.q_stride, t_begin * BQT, a.L, smem + (t_begin & 1) * TB, wave, lane);
    Stg::issue(dobase, a.do_stride, t_begin * BQT, a.L, smem + (2 + (t_begin & 1)) * TB, wave, lane);
  }
  for (int t = t_begin; t < t_end; ++t) {
    const int q0 = t * BQT, st = t & 1;
    const char* Qs = smem + st * TB;
    const char* Os = smem + (2 + st) * TB;
    const float* lse_s = lse_all + st * BQT;
    const float* delta_s = delta_all + st * BQT;
    const long* sidq = sid_all + st * BQT;
    const bool id_test = HAS_SID && doc_pair_needs_mask(a.doc_ranges, b, a.L, t, blk_id);   // block-uniform
    if (tid < BQT) {
      const bool ok = q0 + tid < a.L;
      lse_all[st * BQT + tid] = ok ? a.lse[sbase + q0 + tid] : INFINITY;
      delta_all[st * BQT + tid] = ok ? a.delta[sbase + q0 + tid] : 0.f;
      if (HAS_SID) sid_all[st * BQT + tid] = ok ? a.sample_ids[rowbase + q0 + tid] : -1;
    }
    wait_all_vmem();
    __syncthreads();
    if (t + 1 < t_end) {
      Stg::issue(qbase, a.q_stride, q0 + BQT, a.L, smem + (st ^ 1) * TB, wave, lane);
      Stg::issue(dobase, a.do_stride, q0 + BQT, a.L, smem + (2 + (st ^ 1)) * TB, wave, lane);
    }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = tile_off<D>(qs * 32 + l31, ks * 2 + hi);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Qs, off), kf[ks], s, 0, 0, 0);
        if (DO_DK) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Os, off), vf[ks], dp, 0, 0, 0);
      }
      if (WAVES == 1) {  // one wave per SIMD: nothing else hides LDS latency, so issue all fragment reads up front (registers are free)
        __builtin_amdgcn_sched_group_barrier(0x100, DO_DK ? 2 * KS : KS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, DO_DK ? 2 * KS : KS, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      float p[16], ds[16];
      auto softmax_bwd = [&](auto IDT) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ql0 = qs * 32 + 8 * rg + 4 * hi;
          const float4 l4 = *reinterpret_cast<const float4*>(lse_s + ql0);
          const float4 d4 = *reinterpret_cast<const float4*>(delta_s + ql0);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = rg * 4 + e;
            bool ok = k_ok && (q0 + ql0 + e < a.L);
            if (decltype(IDT)::value) ok = ok && attn_pair_ok(sidq[ql0 + e], sid_k);
            p[r] = ok ? __builtin_amdgcn_exp2f(s[r] * c - lv[e]) : 0.f;
            ds[r] = p[r] * (dp[r] - dv[e]);
          }
        }
      };
      if (HAS_SID && id_test) softmax_bwd(std::true_type{}); else softmax_bwd(std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        bf16x8_t pb = pack8(&p[8 * c2]);
        bf16x8_t dsb = pack8(&ds[8 * c2]);
#pragma unroll
        for (int i = 0; i < DB; ++i) {
          bf16x8_t dot;
          if (DO_DV) dot = lds_frag_T<D, USE_TR>(Os, qs * 32 + c2 * 16, i * 32, lane);
          if (DO_DV) dvT[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pb, dvT[i], 0, 0, 0);
          if (DO_DK) {
            bf16x8_t qt = lds_frag_T<D, USE_TR>(Qs, qs * 32 + c2 * 16, i * 32, lane);
            dkT[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, dsb, dkT[i], 0, 0, 0);
          }
        }
      }
    }
  }
  if (k_ok) {
    bf16_t* kp = a.out2 + (rowbase + ki) * a.out2_stride + h * D;
    bf16_t* vp = a.out3 + (rowbase + ki) * a.out3_stride + h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = i * 32 + 8 * rg + 4 * hi;
        if (DO_DK) *reinterpret_cast<uint2*>(kp + d0) = make_uint2(pack2bf(dkT[i][rg * 4] * a.scale, dkT[i][rg * 4 + 1] * a.scale),
                                                                   pack2bf(dkT[i][rg * 4 + 2] * a.scale, dkT[i][rg * 4 + 3] * a.scale));
        if (DO_DV) *reinterpret_cast<uint2*>(vp + d0) = make_uint2(pack2bf(dvT[i][rg * 4], dvT[i][rg * 4 + 1]), pack2bf(dvT[i][rg * 4 + 2], dvT[i][rg * 4 + 3]));
      }
  }
}

int g_dkv_ws = 1;   // UDM_DKV_WS=0 in the environment selects the single-role dK/dV kernel at head dim 128 too (A/B measurements)

template <typename KernT>
void set_lds(KernT kern, size_t bytes) {
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <int D, bool SID, bool TR>
void launch_fwd(const AttnArgs& a, hipStream_t s) {
  dim3 grid(((a.L + BQ - 1) / BQ) * a.H * a.B);
  const size_t lds = 4 * BKV * D * 2 + 2 * BKV * sizeof(long);
  auto kern = attn_fwd_kernel<D, SID, TR>;
  static bool once = false;
  if (!once) { set_lds(kern, lds); once = true; }
  // head dim 128, no mask, L % 256 == 0, q pre-scaled (the headline path): the one-wave-per-SIMD, 64-queries-per-wave kernel of attention_fwd64.hip
  if (D == 128 && !SID && TR && a.q_prescaled && udm_launch_attn_fwd64(&a, s)) return;
  if (D == 128 && !SID && TR) {   // UDM_ATTN_ABL=1|2: timing-only ablations of the forward kernel (scripts/bench_attn.py)
    static const int abl = [] { const char* e = getenv("UDM_ATTN_ABL"); return e ? atoi(e) : 0; }();
    if (abl == 1) { auto k1 = attn_fwd_kernel<128, false, true, 1>; set_lds(k1, lds); hipLaunchKernelGGL(k1, grid, dim3(256), lds, s, a); return; }
    if (abl == 2) { auto k2 = attn_fwd_kernel<128, false, true, 2>; set_lds(k2, lds); hipLaunchKernelGGL(k2, grid, dim3(256), lds, s, a); return; }
    if (abl == 4) { auto k4 = attn_fwd_kernel<128, false, true, 4>; set_lds(k4, lds); hipLaunchKernelGGL(k4, grid, dim3(256), lds, s, a); return; }   // no refills / barriers
    if (abl == 5) { auto k5 = attn_fwd_kernel<128, false, true, 5>; set_lds(k5, lds); hipLaunchKernelGGL(k5, grid, dim3(256), lds, s, a); return; }   // ... and no softmax
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
}
template <int D, bool SID, bool TR>
void launch_bwd(const AttnArgs& a, hipStream_t s) {
  dim3 gq(((a.L + BQ - 1) / BQ) * a.H * a.B), gk(((a.L + 127) / 128) * a.H * a.B);
  const size_t lds_q = 4 * BKV * D * 2 + 2 * BKV * sizeof(long);
  const size_t lds_k = 4 * BQT * D * 2 + 4 * BQT * sizeof(float) + 2 * BQT * sizeof(long);
  auto kq = attn_bwd_dq_kernel<D, SID, TR>;
  // D = 128: both accumulators (128 registers) plus K/V operands (64) only fit one wave per SIMD in the single-role kernel, which is then
  // bound by that one wave's instruction issue (0.39 ms at B8 H16 L1280).  Without a document mask the wave-specialised kernel of
  // attention_dkv_ws.hip (two waves per SIMD with different roles, 0.27 ms) is used instead.
  constexpr int W = (D == 128) ? 1 : 2;
  auto kk = attn_bwd_dkv_kernel<D, SID, TR, 3, W>;
  static bool once = false;
  if (!once) { set_lds(kq, lds_q); set_lds(kk, lds_k); once = true; }
  // D = 128, no mask, L % 256 == 0, q pre-scaled: both passes as generated one-wave-per-SIMD programs (round 6: attention_dq64.hip, attention_dkv64.hip)
  if (!(D == 128 && !SID && TR && a.q_prescaled && udm_launch_attn_bwd_dq64(&a, s))) hipLaunchKernelGGL(kq, gq, dim3(256), lds_q, s, a);
  if (D == 128 && !SID && TR && a.q_prescaled && udm_launch_attn_bwd_dkv64(&a, s)) return;   // the one-wave-per-SIMD, 64-keys-per-wave kernel (round 6)
  if (D == 128 && !SID && TR && g_dkv_ws) udm_launch_attn_bwd_dkv_ws(&a, s);
  else if (D == 128 && SID && TR && g_dkv_ws && a.doc_ranges) {
    // packed documents: key blocks that lie inside one document and whose query span is exactly that document go to the wave-specialised
    // kernel (no id test needed anywhere); the blocks at document boundaries / with padding stay with the single-role kernel
    AttnArgs a2 = a;
    a2.doc_pure_split = 1;
    udm_launch_attn_bwd_dkv_ws(&a2, s);
    hipLaunchKernelGGL(kk, gk, dim3(256), lds_k, s, a2);
  } else hipLaunchKernelGGL(kk, gk, dim3(256), lds_k, s, a);
}

#define ATTN_DISPATCH(FN, a, D, sid, tr, s)                                    \
  do {                                                                         \
    if (D == 128) { if (sid) { if (tr) FN<128, true, true>(a, s); else FN<128, true, false>(a, s); } else { if (tr) FN<128, false, true>(a, s); else FN<128, false, false>(a, s); } } \
    else if (D == 64) { if (sid) { if (tr) FN<64, true, true>(a, s); else FN<64, true, false>(a, s); } else { if (tr) FN<64, false, true>(a, s); else FN<64, false, false>(a, s); } } \
    else { if (sid) { if (tr) FN<32, true, true>(a, s); else FN<32, true, false>(a, s); } else { if (tr) FN<32, false, true>(a, s); else FN<32, false, false>(a, s); } } \
  } while (0)

int g_use_tr = 1;

int check_common(const char* name, int64_t B, int64_t H, int64_t L, int64_t D, int64_t qs, int64_t ks, int64_t vs) {
  UDM_CHECK_ARG(B > 0 && H > 0 && L > 0, "%s: empty problem", name);
  UDM_CHECK_ARG(D == 32 || D == 64 || D == 128, "%s: head_dim %ld unsupported (32, 64, 128)", name, (long)D);
  UDM_CHECK_ARG(qs % 8 == 0 && ks % 8 == 0 && vs % 8 == 0, "%s: row strides must be multiples of 8 elements", name);
  UDM_CHECK_ARG(B * H * ((L + 127) / 128) < (1LL << 31), "%s: grid too large", name);
  return 0;
}
}  // namespace

extern "C" __attribute__((visibility("hidden"))) int udm_attention_set_tr_read(int enable) {
  g_use_tr = enable ? 1 : 0;
  return 0;
}

extern "C" int udm_attention_doc_ranges(const int64_t* sample_ids, int64_t B, int64_t L, int32_t* ranges, hipStream_t stream) {
  UDM_CHECK_ARG(sample_ids && ranges, "udm_attention_doc_ranges: null pointer");
  UDM_CHECK_ARG(B > 0 && L > 0 && B * ((L + 63) / 64) < (1LL << 31) && L < (1LL << 31), "udm_attention_doc_ranges: bad shape");
  hipLaunchKernelGGL(attn_doc_ranges_kernel, dim3((unsigned)(B * ((L + 63) / 64))), dim3(256), 0, stream, sample_ids, ranges, (int)L);
  UDM_CHECK_LAUNCH("udm_attention_doc_ranges");
  return 0;
}

extern "C" int udm_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int64_t* sample_ids, const int32_t* doc_ranges, int64_t B, int64_t H, int64_t L,
                                 int64_t D, int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride, int64_t flags, hipStream_t stream) {
  UDM_CHECK_ARG(q && k && v && o && lse, "udm_attention_fwd: null pointer");
  UDM_CHECK_ARG((flags & ~(int64_t)UDM_ATTN_Q_PRESCALED) == 0, "udm_attention_fwd: unknown flags %ld", (long)flags);
  if (int rc = check_common("udm_attention_fwd", B, H, L, D, q_stride, k_stride, v_stride)) return rc;
  UDM_CHECK_ARG(o_stride % 4 == 0, "udm_attention_fwd: o_stride must be a multiple of 4");
  UDM_CHECK_ARG(sample_ids || !doc_ranges, "udm_attention_fwd: doc_ranges without sample_ids");
  AttnArgs a{};
  a.exp = udm_exp_flags();
  a.doc_ranges = doc_ranges;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse; a.sample_ids = sample_ids;
  a.q_stride = q_stride; a.k_stride = k_stride; a.v_stride = v_stride; a.out_stride = o_stride;
  a.B = (int)B; a.H = (int)H; a.L = (int)L;
  a.scale = 1.0f / sqrtf((float)D);
  a.scale_log2 = a.scale * 1.4426950408889634f;
  if (flags & UDM_ATTN_Q_PRESCALED) { a.q_prescaled = 1; a.scale_log2 = 1.0f; }   // q already carries log2(e) / sqrt(D): the scores ARE the base-2 exponents
  ATTN_DISPATCH(launch_fwd, a, D, sample_ids != nullptr, g_use_tr, stream);
  UDM_CHECK_LAUNCH("udm_attention_fwd");
  return 0;
}

extern "C" int udm_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta, void* dq, void* dk,
                                 void* dv, const int64_t* sample_ids, const int32_t* doc_ranges, int64_t B, int64_t H, int64_t L, int64_t D, int64_t q_stride, int64_t k_stride,
                                 int64_t v_stride, int64_t o_stride, int64_t do_stride, int64_t dq_stride, int64_t dk_stride, int64_t dv_stride,
                                 int64_t flags, hipStream_t stream) {
  UDM_CHECK_ARG(q && k && v && o && dout && lse && delta && dq && dk && dv, "udm_attention_bwd: null pointer");
  UDM_CHECK_ARG((flags & ~(int64_t)UDM_ATTN_Q_PRESCALED) == 0, "udm_attention_bwd: unknown flags %ld", (long)flags);
  if (int rc = check_common("udm_attention_bwd", B, H, L, D, q_stride, k_stride, v_stride)) return rc;
  UDM_CHECK_ARG(o_stride % 8 == 0 && do_stride % 8 == 0 && dq_stride % 4 == 0 && dk_stride % 4 == 0 && dv_stride % 4 == 0, "udm_attention_bwd: bad strides");
  UDM_CHECK_ARG(sample_ids || !doc_ranges, "udm_attention_bwd: doc_ranges without sample_ids");
  AttnArgs a{};
  a.exp = udm_exp_flags();
  a.doc_ranges = doc_ranges;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dout = (const bf16_t*)dout;
  a.out = (bf16_t*)dq; a.out2 = (bf16_t*)dk; a.out3 = (bf16_t*)dv; a.lse = const_cast<float*>(lse); a.delta = delta; a.sample_ids = sample_ids;
  a.q_stride = q_stride; a.k_stride = k_stride; a.v_stride = v_stride; a.o_stride = o_stride; a.do_stride = do_stride;
  a.out_stride = dq_stride; a.out2_stride = dk_stride; a.out3_stride = dv_stride;
  a.B = (int)B; a.H = (int)H; a.L = (int)L;
  a.scale = 1.0f / sqrtf((float)D);
  a.scale_log2 = a.scale * 1.4426950408889634f;
  // q~ = q log2(e) / sqrt(D): scores are base-2 exponents as they come; dq~ = ln2 dS K and dk = ln2 dS^T q~ (dS wrt the natural-log scores): the factor
  // the kernels put on dQ / dK is ln 2 instead of 1 / sqrt(D)
  if (flags & UDM_ATTN_Q_PRESCALED) { a.q_prescaled = 1; a.scale_log2 = 1.0f; a.scale = 0.6931471805599453f; }
  static const bool env_once = [] { if (const char* e = getenv("UDM_DKV_WS")) g_dkv_ws = atoi(e); return true; }();
  (void)env_once;
  // (delta is computed and stored by the dQ kernel, which launch_bwd runs first)
  ATTN_DISPATCH(launch_bwd, a, D, sample_ids != nullptr, g_use_tr, stream);
  UDM_CHECK_LAUNCH("udm_attention_bwd");
  return 0;
}
