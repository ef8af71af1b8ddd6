// Fused multi-head attention for gfx950: forward, dQ and dK/dV kernels on
// v_mfma_f32_32x32x16_f16 with in-register online softmax.  No N x N score matrix ever
// reaches HBM (the reference materialises it: models/xti_attention_processor.py:48-49).
//
// The XTI contract (models/xti_attention_processor.py:16-42) is native here: keys and values
// are separate tensors (K from CONTEXT_TENSOR_l, V from CONTEXT_TENSOR_BYPASS_l), and the
// backward returns dK and dV separately so they flow to the two context gradients.
//
// Operand orientation.  Scores are computed TRANSPOSED, S^T[key][q] = K.Q^T, so that in the
// 32x32 MFMA accumulator layout (col = lane&31, rows = 8*(r>>2) + 4*(lane>>5) + (r&3)) every
// lane owns one query column: row max / row sum are in-lane reductions plus one cross-half
// shuffle.  The accumulator registers 8j..8j+7 of a lane are then *directly* the B operand
// (k = 8 keys) of the second MFMA  O^T[d][q] += V^T[d][key] . P^T[key][q]  if the A operand
// is gathered with the same key order.  That A operand (V^T here; K^T, Q^T, dO^T in the backward
// kernels) is read TRANSPOSED out of the row-major [key][d] LDS tile with ds_read_b64_tr_b16
// (load_tr below): callers pass plain row-major Q/K/V/dO views, no transposed copies exist.
// Head dims 40/80 are zero padded to the MFMA k-step (16) for contractions over d and to 32 for
// output rows.
#include "common.h"
#include "../../include/vneti.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int TROW64 = 144;  // bytes per row of a [d][64 keys] transposed LDS tile (64*2 + 16 pad)
constexpr int TROW32 = 80;   // bytes per row of a [d][32 q] transposed LDS tile (32*2 + 16 pad)
// dK/dV kernel: 32-row query halves per LDS stage (2 where two blocks of 64-row stages still fit one CU's LDS)
constexpr int DKV_QH(int D) { return D <= 64 ? 2 : 1; }

template <int D>
struct Cfg {
  static constexpr int KS = (D + 15) / 16;   // k-steps when contracting over d
  static constexpr int DPAD = KS * 16;
  static constexpr int DB = (D + 31) / 32;   // 32-row output blocks over d
  static constexpr int DCH = D / 8;          // valid 16-byte chunks per row
  static constexpr int ROW = DPAD * 2 + 16;  // bytes per LDS row of a [n][d] tile (odd multiple of 16)
};

struct AttnArgs {
  const half_t *Q, *K, *V, *dO;
  half_t *O, *dQ, *dK, *dV;
  float* lse;
  const float* lse_in;
  const float* delta;
  float* delta_out;      // dQ kernel with O given: delta is computed in-kernel and published here
  const half_t* O_in;    //   (saves the separate delta launch; the dK/dV kernel then runs after dQ)
  long long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int Bn, H, Nq, Nk, D;
  float scale;
  int causal;
  int qsplit;   // dK/dV kernel: number of query ranges per key block (f32 partials -> ws)
  float* ws;    // [qsplit][2][Bn*Nk][H*D] f32
};

// raw v_exp_f32: the libm exp2 adds a 5-instruction denormal-range fix-up per element, which made
// the softmax the bottleneck; flushing results below 2^-126 to zero is harmless here.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// two f32 lanes per VALU instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): the softmax element work is what these
// kernels spend their VALU cycles on, and the compiler does not pair the score scaling by itself.  Same roundings as the
// scalar forms (one fused multiply-add; a subtract and a multiply).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// online-softmax rescale threshold (natural-log units, expressed in log2 units at the use site):
// the running max is only advanced when some row's max grew by more than this, so in steady state the
// O/l rescale is skipped; P stays <= e^8 (fits f16, sums in f32).
constexpr float RESCALE_THR_LOG2 = 8.0f * 1.4426950408889634f;

__device__ __forceinline__ void zero_lds(char* smem, int bytes) {
  u32x4 z = {0u, 0u, 0u, 0u};
  for (int i = threadIdx.x * 16; i < bytes; i += 256 * 16) *reinterpret_cast<u32x4*>(smem + i) = z;
}

__device__ __forceinline__ half8 cvt8(const f32x16& v, int j) {
  half8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = (half_t)v[8 * j + e];
  return h;
}

// A operand (rows = d, k-slots = the 8 keys/queries a lane owns) read TRANSPOSED out of a ROW-MAJOR
// [key|query][d] LDS tile with gfx950's ds_read_b64_tr_b16: within a 16-lane group source lane i = 4*jj + qq
// supplies the address of 4 consecutive d of row (n0 + jj) and the hardware hands lane i the 4 rows jj = 0..3 of
// column d0 + i (semantics measured with tools/lab/tr_probe.hip).  Lanes 16..31 take the next 16 d.  Two reads
// give the 8 k-slots (rows n0..n0+3 and n0+8..n0+11) that the P / dS accumulator layout pairs with, so no
// transposed operand copies (V^T, K^T, Q^T, dO^T) exist anywhere any more.
typedef short vn_short4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ half8 load_tr(const char* tile, int row_bytes, int d0, int n0, int lane) {
  const int i = lane & 15;
  const char* p = tile + (n0 + (i >> 2)) * row_bytes + (d0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
  typedef __attribute__((address_space(3))) vn_short4* lds_ptr;
  vn_short4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  vn_short4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 8 * row_bytes));
  half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
  half8 r = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  return r;
}

// ============================================================================================
// forward
// ============================================================================================
template <int D, int QW>
// min 2 blocks/CU caps the register budget at 256, which makes the compiler keep MFMA accumulators in
// arch VGPRs (no v_accvgpr copies around the softmax VALU work)
__global__ __launch_bounds__(256, (D <= 40 && QW == 1 ? 4 : 2)) void attn_fwd_kernel(AttnArgs a) {
  using C = Cfg<D>;
  // each wave owns QW independent 32-query sub-tiles: K/V fragments are read from LDS once and
  // used QW times, and the two softmax/MFMA dependency chains interleave inside the wave.  One MFMA per fragment read
  // (QW = 1) sits exactly on the LDS roofline (1 KB per 32x32x16 MFMA = 128 B/clk/CU at the MFMA rate).
  // when D is padded up to a multiple of 32, column D of the V tile is set to ones so the PV MFMA
  // accumulates the softmax denominator for free (it rescales with O as well)
  constexpr bool ONES = C::DB * 32 > D;
  constexpr int L_DB = D / 32, L_R = ((D % 32) & 3) + 4 * ((D % 32) >> 3), L_H2 = ((D % 32) >> 2) & 1;
  constexpr int KS_BYTES = 64 * C::ROW;
  constexpr int VS_BYTES = 64 * C::ROW + 128;  // V tile row-major like K (+ slack: the last rows' d-padding is read)
  constexpr int STAGE = KS_BYTES + VS_BYTES;  // two stages: the next tile is written while this one is read
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h2 = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qb0 = blockIdx.x * 128 * QW;
  int q[QW];
  bool qok[QW];
#pragma unroll
  for (int u = 0; u < QW; ++u) {
    q[u] = qb0 + (wave * QW + u) * 32 + l31;
    qok[u] = q[u] < a.Nq;
  }

  __amdgpu_buffer_rsrc_t rsQ = vn_make_rsrc(a.Q, (uint32_t)((long long)a.Bn * a.Nq * a.ldq * 2));
  __amdgpu_buffer_rsrc_t rsK = vn_make_rsrc(a.K, (uint32_t)((long long)a.Bn * a.Nk * a.ldk * 2));
  __amdgpu_buffer_rsrc_t rsV = vn_make_rsrc(a.V, (uint32_t)((long long)a.Bn * a.Nk * a.ldv * 2));

  zero_lds(smem, 2 * STAGE);

  half8 qf[QW][C::KS];
#pragma unroll
  for (int u = 0; u < QW; ++u)
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      int ch = ks * 2 + h2;
      uint32_t off = (qok[u] && ch < C::DCH)
                         ? (uint32_t)((((long long)b * a.Nq + q[u]) * a.ldq + h * D + ch * 8) * 2)
                         : VN_OOB;
      qf[u][ks] = as_half8(vn_buf_load16(rsQ, off));
    }

  constexpr int KIT = (64 * C::DCH + 255) / 256;
  u32x4 kreg[KIT], vreg[KIT];

  auto issue = [&](int kt) {
    const int key0 = kt * 64;
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      int idx = tid + 256 * i;
      int r = idx / C::DCH, c = idx - r * C::DCH;
      int key = key0 + r;
      const bool ok = idx < 64 * C::DCH && key < a.Nk;
      kreg[i] = vn_buf_load16(rsK, ok ? (uint32_t)((((long long)b * a.Nk + key) * a.ldk + h * D + c * 8) * 2) : VN_OOB);
      vreg[i] = vn_buf_load16(rsV, ok ? (uint32_t)((((long long)b * a.Nk + key) * a.ldv + h * D + c * 8) * 2) : VN_OOB);
    }
  };
  auto commit = [&](int buf) {
    char* Ks = smem + buf * STAGE;
    char* Vs = Ks + KS_BYTES;
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      int idx = tid + 256 * i;
      int r = idx / C::DCH, c = idx - r * C::DCH;
      if (idx < 64 * C::DCH) {
        *reinterpret_cast<u32x4*>(Ks + r * C::ROW + c * 16) = kreg[i];
        *reinterpret_cast<u32x4*>(Vs + r * C::ROW + c * 16) = vreg[i];
      }
    }
  };

  f32x16 o[QW][C::DB];
  float m[QW], l[QW];
#pragma unroll
  for (int u = 0; u < QW; ++u) {
    m[u] = -INFINITY;
    l[u] = 0.f;
#pragma unroll
    for (int i = 0; i < C::DB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[u][i][e] = 0.f;
  }
  const float c = a.scale * LOG2E;

  int nkt = cdiv_dev(a.Nk, 64);
  if (a.causal) {
    int lim = (min(qb0 + 128 * QW - 1, a.Nq - 1)) / 64 + 1;
    nkt = min(nkt, lim);
  }

  issue(0);
  __syncthreads();  // LDS zero-fill visible
  if constexpr (ONES) {  // column D of every key row of the V tile (padding that commit() never touches)
    if (tid < 128)
      *reinterpret_cast<half_t*>(smem + (tid >> 6) * STAGE + KS_BYTES + (tid & 63) * C::ROW + D * 2) = (half_t)1.f;
  }
  commit(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) issue(kt + 1);
    const int key0 = kt * 64;
    const char* Ks = smem + (kt & 1) * STAGE;
    const char* Vs = Ks + KS_BYTES;

    f32x16 s[QW][2];
#pragma unroll
    for (int u = 0; u < QW; ++u)
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[u][aa][e] = 0.f;
#pragma unroll
    for (int aa = 0; aa < 2; ++aa) {
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        half8 kf = as_half8(
            *reinterpret_cast<const u32x4*>(Ks + (aa * 32 + l31) * C::ROW + (ks * 2 + h2) * 16));
#pragma unroll
        for (int u = 0; u < QW; ++u) {
          s[u][aa] = VN_MFMA_32x32x16(kf, qf[u][ks], s[u][aa], 0, 0, 0);
        }
      }
    }
    if ((key0 + 64 > a.Nk) || a.causal) {  // wave-uniform: only the tail / diagonal tiles pay for masking
#pragma unroll
      for (int u = 0; u < QW; ++u)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int key = key0 + aa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
            bool ok = key < a.Nk && (!a.causal || key <= q[u]);
            if (!ok) s[u][aa][r] = -INFINITY;
          }
    }
#pragma unroll
    for (int u = 0; u < QW; ++u) {
      float mx = -INFINITY;
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][aa][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (!__all((mx - m[u]) * c <= RESCALE_THR_LOG2)) {
        // some row's max grew a lot (always true on the first tile): advance the running max and
        // rescale everything accumulated so far, exactly once
        const float m_new = fmaxf(m[u], mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2((m[u] - m_use) * c);
        if constexpr (!ONES) l[u] *= alpha;
        m[u] = m_use;
#pragma unroll
        for (int i = 0; i < C::DB; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[u][i][e] *= alpha;
      }
      const float mc = m[u] * c;
      float psum = 0.f;
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 t = pk_fma(f32x2{s[u][aa][r], s[u][aa][r + 1]}, f32x2{c, c}, f32x2{-mc, -mc});
          const float p0 = fast_exp2(t.x), p1 = fast_exp2(t.y);
          s[u][aa][r] = p0;
          s[u][aa][r + 1] = p1;
          if constexpr (!ONES) psum += p0 + p1;
        }
      }
      if constexpr (!ONES) l[u] += psum;
    }

#pragma unroll
    for (int aa = 0; aa < 2; ++aa) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half8 pf[QW];
#pragma unroll
        for (int u = 0; u < QW; ++u) pf[u] = cvt8(s[u][aa], j);
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
          half8 vf = load_tr(Vs, C::ROW, db * 32, aa * 32 + 16 * j + 4 * h2, lane);
#pragma unroll
          for (int u = 0; u < QW; ++u) {
            o[u][db] = VN_MFMA_32x32x16(vf, pf[u], o[u][db], 0, 0, 0);
          }
        }
      }
    }
    if (kt + 1 < nkt) commit((kt + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int u = 0; u < QW; ++u) {
    float lt;
    if constexpr (ONES) lt = __shfl(o[u][L_DB][L_R], l31 + 32 * L_H2, 64);
    else lt = l[u] + __shfl_xor(l[u], 32, 64);
    const float inv = (lt > 0.f) ? 1.f / lt : 0.f;
    if (qok[u]) {
      half_t* orow = a.O + ((long long)b * a.Nq + q[u]) * a.ldo + h * D;
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          int d = db * 32 + 8 * qd + 4 * h2;
          if (d < D) {
            half4 v = {(half_t)(o[u][db][4 * qd] * inv), (half_t)(o[u][db][4 * qd + 1] * inv),
                       (half_t)(o[u][db][4 * qd + 2] * inv), (half_t)(o[u][db][4 * qd + 3] * inv)};
            *reinterpret_cast<half4*>(orow + d) = v;
          }
        }
      }
      if (h2 == 0 && a.lse) a.lse[((long long)b * a.H + h) * a.Nq + q[u]] = m[u] * a.scale + logf(lt);
    }
  }
}

// ============================================================================================
// backward: delta[b][h][q] = sum_d dO * O
// ============================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const half_t* __restrict__ dO, long long lddo,
                                                         const half_t* __restrict__ O, long long ldo,
                                                         float* __restrict__ delta, int Bn, int H, int Nq,
                                                         int D) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)Bn * Nq * H;
  if (gid >= total) return;
  int h = (int)(gid % H);
  long long row = gid / H;  // b*Nq + q
  const half_t* po = O + row * ldo + h * D;
  const half_t* pd = dO + row * lddo + h * D;
  float s = 0.f;
  for (int c = 0; c < D; c += 8) {
    half8 x = *reinterpret_cast<const half8*>(po + c);
    half8 y = *reinterpret_cast<const half8*>(pd + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)x[j] * (float)y[j];
  }
  int b = (int)(row / Nq), q = (int)(row - (long long)b * Nq);
  delta[((long long)b * H + h) * Nq + q] = s;
}

// ============================================================================================
// backward: dQ.   per q-block of 128 (4 waves x 32 q), loop over 64-key tiles.
//   S^T = K.Q^T ; P^T = exp(S^T*scale - lse) ; dP^T = V.dO^T ; dS^T = P^T o (dP^T - delta)
//   dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
// ============================================================================================
template <int D>
// min 2 blocks/CU caps the register budget at 256, which makes the compiler keep MFMA accumulators in
// arch VGPRs (no v_accvgpr copies around the softmax VALU work)
// (4 blocks per CU for D = 40 was measured: 128 VGPRs + 28 B/lane of scratch, 251-257 us vs 240 us at N = 4096)
__global__ __launch_bounds__(256, (D >= 160 ? 1 : 2)) void attn_dq_kernel(AttnArgs a) {
  using C = Cfg<D>;
  constexpr int KS_BYTES = 64 * C::ROW;
  constexpr int STAGE = 2 * KS_BYTES + 128;  // K and V tiles (row-major) + slack for the transposed reads' d-padding
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h2 = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qb0 = blockIdx.x * 128;
  const int q = qb0 + wave * 32 + l31;
  const bool qok = q < a.Nq;

  __amdgpu_buffer_rsrc_t rsQ = vn_make_rsrc(a.Q, (uint32_t)((long long)a.Bn * a.Nq * a.ldq * 2));
  __amdgpu_buffer_rsrc_t rsdO = vn_make_rsrc(a.dO, (uint32_t)((long long)a.Bn * a.Nq * a.lddo * 2));
  __amdgpu_buffer_rsrc_t rsK = vn_make_rsrc(a.K, (uint32_t)((long long)a.Bn * a.Nk * a.ldk * 2));
  __amdgpu_buffer_rsrc_t rsV = vn_make_rsrc(a.V, (uint32_t)((long long)a.Bn * a.Nk * a.ldv * 2));

  zero_lds(smem, 2 * STAGE);

  half8 qf[C::KS], dof[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    int ch = ks * 2 + h2;
    bool ok = qok && ch < C::DCH;
    uint32_t offq = ok ? (uint32_t)((((long long)b * a.Nq + q) * a.ldq + h * D + ch * 8) * 2) : VN_OOB;
    uint32_t offd = ok ? (uint32_t)((((long long)b * a.Nq + q) * a.lddo + h * D + ch * 8) * 2) : VN_OOB;
    qf[ks] = as_half8(vn_buf_load16(rsQ, offq));
    dof[ks] = as_half8(vn_buf_load16(rsdO, offd));
  }
  const float c = a.scale * LOG2E;
  float lse2 = INFINITY, dlt = 0.f;
  if (a.O_in) {
    // delta[q] = sum_d dO[q][d] * O[q][d]: the lane already holds its half of the dO row as MFMA fragments
    __amdgpu_buffer_rsrc_t rsO = vn_make_rsrc(a.O_in, (uint32_t)((long long)a.Bn * a.Nq * a.ldo * 2));
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      int ch = ks * 2 + h2;
      uint32_t offo = (qok && ch < C::DCH)
                          ? (uint32_t)((((long long)b * a.Nq + q) * a.ldo + h * D + ch * 8) * 2)
                          : VN_OOB;
      half8 of = as_half8(vn_buf_load16(rsO, offo));
#pragma unroll
      for (int j = 0; j < 8; ++j) part += (float)dof[ks][j] * (float)of[j];
    }
    dlt = part + __shfl_xor(part, 32, 64);
    if (qok && h2 == 0) a.delta_out[((long long)b * a.H + h) * a.Nq + q] = dlt;
  }
  if (qok) {
    lse2 = a.lse_in[((long long)b * a.H + h) * a.Nq + q] * LOG2E;
    if (!a.O_in) dlt = a.delta[((long long)b * a.H + h) * a.Nq + q];
  }

  constexpr int KIT = (64 * C::DCH + 255) / 256;
  u32x4 kreg[KIT], vreg[KIT];

  auto issue = [&](int kt) {
    const int key0 = kt * 64;
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      int idx = tid + 256 * i;
      int r = idx / C::DCH, cc = idx - r * C::DCH;
      int key = key0 + r;
      bool ok = idx < 64 * C::DCH && key < a.Nk;
      uint32_t offk = ok ? (uint32_t)((((long long)b * a.Nk + key) * a.ldk + h * D + cc * 8) * 2) : VN_OOB;
      uint32_t offv = ok ? (uint32_t)((((long long)b * a.Nk + key) * a.ldv + h * D + cc * 8) * 2) : VN_OOB;
      kreg[i] = vn_buf_load16(rsK, offk);
      vreg[i] = vn_buf_load16(rsV, offv);
    }
  };
  auto commit = [&](int buf) {
    // V first, K last: the transposed reads of the K tile run a few halves past its rows' d-padding
    char* Vs = smem + buf * STAGE;
    char* Ks = Vs + KS_BYTES;
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      int idx = tid + 256 * i;
      int r = idx / C::DCH, cc = idx - r * C::DCH;
      if (idx < 64 * C::DCH) {
        *reinterpret_cast<u32x4*>(Ks + r * C::ROW + cc * 16) = kreg[i];
        *reinterpret_cast<u32x4*>(Vs + r * C::ROW + cc * 16) = vreg[i];
      }
    }
  };

  f32x16 dq[C::DB];
#pragma unroll
  for (int i = 0; i < C::DB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[i][e] = 0.f;

  int nkt = cdiv_dev(a.Nk, 64);
  if (a.causal) nkt = min(nkt, (min(qb0 + 127, a.Nq - 1)) / 64 + 1);

  issue(0);
  __syncthreads();
  commit(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) issue(kt + 1);
    const int key0 = kt * 64;
    const char* Vs = smem + (kt & 1) * STAGE;
    const char* Ks = Vs + KS_BYTES;
    const bool need_mask = (key0 + 64 > a.Nk) || a.causal;
#pragma unroll
    for (int aa = 0; aa < 2; ++aa) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] = 0.f;
        dp[e] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int off = (aa * 32 + l31) * C::ROW + (ks * 2 + h2) * 16;
        half8 kf = as_half8(*reinterpret_cast<const u32x4*>(Ks + off));
        half8 vf = as_half8(*reinterpret_cast<const u32x4*>(Vs + off));
        s = VN_MFMA_32x32x16(kf, qf[ks], s, 0, 0, 0);
        dp = VN_MFMA_32x32x16(vf, dof[ks], dp, 0, 0, 0);
      }
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = key0 + aa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
          bool ok = key < a.Nk && (!a.causal || key <= q);
          if (!ok) s[r] = -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = pk_fma(f32x2{s[r], s[r + 1]}, f32x2{c, c}, f32x2{-lse2, -lse2});
        const f32x2 g = f32x2{dp[r], dp[r + 1]} - f32x2{dlt, dlt};
        const f32x2 ds = f32x2{fast_exp2(t.x), fast_exp2(t.y)} * g;
        s[r] = ds.x;
        s[r + 1] = ds.y;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half8 pf = cvt8(s, j);
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
          half8 tf = load_tr(Ks, C::ROW, db * 32, aa * 32 + 16 * j + 4 * h2, lane);
          dq[db] = VN_MFMA_32x32x16(tf, pf, dq[db], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < nkt) commit((kt + 1) & 1);
    __syncthreads();
  }

  if (qok) {
    half_t* orow = a.dQ + ((long long)b * a.Nq + q) * a.lddq + h * D;
#pragma unroll
    for (int db = 0; db < C::DB; ++db) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        int d = db * 32 + 8 * qd + 4 * h2;
        if (d < D) {
          half4 v = {(half_t)(dq[db][4 * qd] * a.scale), (half_t)(dq[db][4 * qd + 1] * a.scale),
                     (half_t)(dq[db][4 * qd + 2] * a.scale), (half_t)(dq[db][4 * qd + 3] * a.scale)};
          *reinterpret_cast<half4*>(orow + d) = v;
        }
      }
    }
  }
}

// ============================================================================================
// backward: dK, dV.   per key-block of 128 (4 waves x 32 keys), loop over 32-query tiles.
//   S[q][key] = Q.K^T ; P = exp(S*scale - lse) ; dP = dO.V^T ; dS = P o (dP - delta)
//   dV^T[d][key] += dO^T[d][q] . P[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]
// ============================================================================================
template <int D>
// min 2 blocks/CU caps the register budget at 256, which makes the compiler keep MFMA accumulators in
// arch VGPRs (no v_accvgpr copies around the softmax VALU work)
// d = 40: TWO blocks per CU on purpose.  At three (168 VGPRs, 24 B/lane of scratch) the 1024-block grid of the 64x64
// self-attention takes 1.33 residency rounds; at two (179 VGPRs, no scratch) it is exactly two rounds: 405 -> 358 us.
__global__ __launch_bounds__(256, (D >= 160 ? 1 : 2)) void attn_dkv_kernel(AttnArgs a) {
  using C = Cfg<D>;
  // query tile per barrier interval: 64 rows (two 32-row halves computed back to back with the same
  // registers) where two such blocks still fit a CU's LDS, else 32
  constexpr int QH = DKV_QH(D), QR = 32 * QH;
  constexpr int QS_BYTES = QR * C::ROW + 128;  // row-major Q / dO tiles (+ slack: transposed reads touch the d-padding)
  constexpr int STAGE = 2 * QS_BYTES + 8 * QR;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h2 = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int nkb = cdiv_dev(a.Nk, 128);
  const int qs = blockIdx.x / nkb;
  const int kb0 = (blockIdx.x - qs * nkb) * 128;
  const int key = kb0 + wave * 32 + l31;
  const bool kok = key < a.Nk;

  __amdgpu_buffer_rsrc_t rsQ = vn_make_rsrc(a.Q, (uint32_t)((long long)a.Bn * a.Nq * a.ldq * 2));
  __amdgpu_buffer_rsrc_t rsdO = vn_make_rsrc(a.dO, (uint32_t)((long long)a.Bn * a.Nq * a.lddo * 2));
  __amdgpu_buffer_rsrc_t rsK = vn_make_rsrc(a.K, (uint32_t)((long long)a.Bn * a.Nk * a.ldk * 2));
  __amdgpu_buffer_rsrc_t rsV = vn_make_rsrc(a.V, (uint32_t)((long long)a.Bn * a.Nk * a.ldv * 2));

  zero_lds(smem, 2 * STAGE);

  half8 kf[C::KS], vf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    int ch = ks * 2 + h2;
    bool ok = kok && ch < C::DCH;
    uint32_t offk = ok ? (uint32_t)((((long long)b * a.Nk + key) * a.ldk + h * D + ch * 8) * 2) : VN_OOB;
    uint32_t offv = ok ? (uint32_t)((((long long)b * a.Nk + key) * a.ldv + h * D + ch * 8) * 2) : VN_OOB;
    kf[ks] = as_half8(vn_buf_load16(rsK, offk));
    vf[ks] = as_half8(vn_buf_load16(rsV, offv));
  }
  const float c = a.scale * LOG2E;

  constexpr int QIT = (QR * C::DCH + 255) / 256;
  u32x4 qreg[QIT], doreg[QIT];
  float lreg = INFINITY, dreg = 0.f;

  auto issue = [&](int qt) {
    const int q0 = qt * QR;
#pragma unroll
    for (int i = 0; i < QIT; ++i) {
      int idx = tid + 256 * i;
      int r = idx / C::DCH, cc = idx - r * C::DCH;
      int qq = q0 + r;
      bool ok = idx < QR * C::DCH && qq < a.Nq;
      uint32_t o1 = ok ? (uint32_t)((((long long)b * a.Nq + qq) * a.ldq + h * D + cc * 8) * 2) : VN_OOB;
      uint32_t o2 = ok ? (uint32_t)((((long long)b * a.Nq + qq) * a.lddo + h * D + cc * 8) * 2) : VN_OOB;
      qreg[i] = vn_buf_load16(rsQ, o1);
      doreg[i] = vn_buf_load16(rsdO, o2);
    }
    if (tid < QR) {
      int qq = q0 + tid;
      if (qq < a.Nq) {
        lreg = a.lse_in[((long long)b * a.H + h) * a.Nq + qq] * LOG2E;
        dreg = a.delta[((long long)b * a.H + h) * a.Nq + qq];
      } else {
        lreg = INFINITY;
        dreg = 0.f;
      }
    }
  };
  auto commit = [&](int buf) {
    char* Qs = smem + buf * STAGE;
    char* dOs = Qs + QS_BYTES;
    float* lses = reinterpret_cast<float*>(dOs + QS_BYTES);
    float* dels = lses + QR;
#pragma unroll
    for (int i = 0; i < QIT; ++i) {
      int idx = tid + 256 * i;
      int r = idx / C::DCH, cc = idx - r * C::DCH;
      if (idx < QR * C::DCH) {
        *reinterpret_cast<u32x4*>(Qs + r * C::ROW + cc * 16) = qreg[i];
        *reinterpret_cast<u32x4*>(dOs + r * C::ROW + cc * 16) = doreg[i];
      }
    }
    if (tid < QR) {
      lses[tid] = lreg;
      dels[tid] = dreg;
    }
  };

  f32x16 dk[C::DB], dv[C::DB];
#pragma unroll
  for (int i = 0; i < C::DB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      dk[i][e] = 0.f;
      dv[i][e] = 0.f;
    }

  const int nqt_all = cdiv_dev(a.Nq, QR);
  const int per = cdiv_dev(nqt_all, a.qsplit);
  const int nqt = min(nqt_all, (qs + 1) * per);
  const int qt0 = max(qs * per, a.causal ? min(kb0 / QR, nqt_all) : 0);

  if (qt0 < nqt) issue(qt0);
  __syncthreads();
  if (qt0 < nqt) commit(0);
  __syncthreads();
  for (int qt = qt0; qt < nqt; ++qt) {
    if (qt + 1 < nqt) issue(qt + 1);
    const int q0 = qt * QR;
    const char* Qs = smem + ((qt - qt0) & 1) * STAGE;
    const char* dOs = Qs + QS_BYTES;
    const float* lses = reinterpret_cast<const float*>(dOs + QS_BYTES);
    const float* dels = lses + QR;

#pragma unroll
    for (int hq = 0; hq < QH; ++hq) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] = 0.f;
        dp[e] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int off = (hq * 32 + l31) * C::ROW + (ks * 2 + h2) * 16;
        half8 qfr = as_half8(*reinterpret_cast<const u32x4*>(Qs + off));
        half8 dofr = as_half8(*reinterpret_cast<const u32x4*>(dOs + off));
        s = VN_MFMA_32x32x16(qfr, kf[ks], s, 0, 0, 0);
        dp = VN_MFMA_32x32x16(dofr, vf[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 l4 = *reinterpret_cast<const f32x4*>(&lses[hq * 32 + 8 * qd + 4 * h2]);
        f32x4 d4 = *reinterpret_cast<const f32x4*>(&dels[hq * 32 + 8 * qd + 4 * h2]);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const int r = 4 * qd + e;
          const f32x2 t = pk_fma(f32x2{s[r], s[r + 1]}, f32x2{c, c}, -f32x2{l4[e], l4[e + 1]});
          f32x2 p = f32x2{fast_exp2(t.x), fast_exp2(t.y)};
          if (a.causal) {
            int qq = q0 + hq * 32 + 8 * qd + 4 * h2 + e;
            if (key > qq) p.x = 0.f;
            if (key > qq + 1) p.y = 0.f;
          }
          const f32x2 ds = p * (f32x2{dp[r], dp[r + 1]} - f32x2{d4[e], d4[e + 1]});
          s[r] = p.x;
          s[r + 1] = p.y;
          dp[r] = ds.x;
          dp[r + 1] = ds.y;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half8 pf = cvt8(s, j);
        half8 dsf = cvt8(dp, j);
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
          half8 dot = load_tr(dOs, C::ROW, db * 32, hq * 32 + 16 * j + 4 * h2, lane);
          half8 qtf = load_tr(Qs, C::ROW, db * 32, hq * 32 + 16 * j + 4 * h2, lane);
          dv[db] = VN_MFMA_32x32x16(dot, pf, dv[db], 0, 0, 0);
          dk[db] = VN_MFMA_32x32x16(qtf, dsf, dk[db], 0, 0, 0);
        }
      }
    }
    if (qt + 1 < nqt) commit((qt + 1 - qt0) & 1);
    __syncthreads();
  }

  if (a.qsplit > 1) {
    // f32 partials: ws[qs][0 = dK (unscaled) | 1 = dV][b*Nk + key][h*D + d]
    if (kok) {
      const long long plane = (long long)a.Bn * a.Nk * a.H * D;
      const __amdgpu_buffer_rsrc_t rsP = vn_make_rsrc(a.ws, 0xfffffff0u);
#pragma unroll
      for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          int d = db * 32 + 8 * qd + 4 * h2;
          if (d < D) {
            f32x4 k4 = {dk[db][4 * qd], dk[db][4 * qd + 1], dk[db][4 * qd + 2], dk[db][4 * qd + 3]};
            f32x4 v4 = {dv[db][4 * qd], dv[db][4 * qd + 1], dv[db][4 * qd + 2], dv[db][4 * qd + 3]};
            // (write-through: 126 MB of f32 partials at 64 x 64 must not sit dirty in L2 at the kernel boundary)
            const uint32_t po = (uint32_t)(((((long long)qs * 2) * plane + ((long long)b * a.Nk + key) * (a.H * D) + h * D + d)) * 4);
            vn_st16_wt(rsP, po, k4);
            vn_st16_wt(rsP, po + (uint32_t)(plane * 4), v4);
          }
        }
    }
    return;
  }
  if (kok) {
    half_t* krow = a.dK + ((long long)b * a.Nk + key) * a.lddk + h * D;
    half_t* vrow = a.dV + ((long long)b * a.Nk + key) * a.lddv + h * D;
#pragma unroll
    for (int db = 0; db < C::DB; ++db) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        int d = db * 32 + 8 * qd + 4 * h2;
        if (d < D) {
          half4 k4 = {(half_t)(dk[db][4 * qd] * a.scale), (half_t)(dk[db][4 * qd + 1] * a.scale),
                      (half_t)(dk[db][4 * qd + 2] * a.scale), (half_t)(dk[db][4 * qd + 3] * a.scale)};
          half4 v4 = {(half_t)dv[db][4 * qd], (half_t)dv[db][4 * qd + 1], (half_t)dv[db][4 * qd + 2],
                      (half_t)dv[db][4 * qd + 3]};
          *reinterpret_cast<half4*>(krow + d) = k4;
          *reinterpret_cast<half4*>(vrow + d) = v4;
        }
      }
    }
  }
}

// ============================================================================================
// backward for short sequences (Nq = Nk <= 96: the 77 tokens of the CLIP text encoder, 64 x 12 (sequence, head) pairs per
// step): one block per (sequence, head) keeps Q, K, V, dO of the head in LDS and produces dQ, dK and dV in ONE launch —
// the dQ and dK/dV kernels above are each a ~15-20 us latency chain at this size (zero-fill, global loads, two staged
// tiles, epilogue) for 0.1 GFLOP.  Wave w owns rows 32w..32w+31 twice: as queries (the dQ body above over the key blocks,
// computing delta on the way) and, after a barrier, as keys (the dK/dV body over the query blocks).  Same arithmetic per
// element as the separate kernels (S recomputed in both roles), so results are bit-identical to them.
// ============================================================================================
template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_small_kernel(AttnArgs a) {
  using C = Cfg<D>;
  constexpr int NR = 96;
  constexpr int TILE = NR * C::ROW + 128;  // row-major [row][d] (+ slack: the transposed reads touch the d-padding)
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE + 2 * NR * 4];
  char* Qs = smem;
  char* Ks = smem + TILE;
  char* Vs = smem + 2 * TILE;
  char* dOs = smem + 3 * TILE;
  float* lses = reinterpret_cast<float*>(smem + 4 * TILE);
  float* dels = lses + NR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h2 = lane >> 5;
  const int b = blockIdx.y, h = blockIdx.x;
  const int N = a.Nq;

  __amdgpu_buffer_rsrc_t rsQ = vn_make_rsrc(a.Q, (uint32_t)((long long)a.Bn * N * a.ldq * 2));
  __amdgpu_buffer_rsrc_t rsdO = vn_make_rsrc(a.dO, (uint32_t)((long long)a.Bn * N * a.lddo * 2));
  __amdgpu_buffer_rsrc_t rsK = vn_make_rsrc(a.K, (uint32_t)((long long)a.Bn * N * a.ldk * 2));
  __amdgpu_buffer_rsrc_t rsV = vn_make_rsrc(a.V, (uint32_t)((long long)a.Bn * N * a.ldv * 2));
  __amdgpu_buffer_rsrc_t rsO = vn_make_rsrc(a.O_in, (uint32_t)((long long)a.Bn * N * a.ldo * 2));

  // ---- the four operand tiles: requested first, stored after the zero-fill (pad rows and the d-padding stay zero) ----
  constexpr int NIT = (NR * C::DCH + 255) / 256;
  u32x4 rq[NIT], rk[NIT], rv[NIT], rd[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx / C::DCH, cc = idx - r * C::DCH;
    const bool ok = idx < NR * C::DCH && r < N;
    const long long row = (long long)b * N + r;
    rq[i] = vn_buf_load16(rsQ, ok ? (uint32_t)((row * a.ldq + h * D + cc * 8) * 2) : VN_OOB);
    rk[i] = vn_buf_load16(rsK, ok ? (uint32_t)((row * a.ldk + h * D + cc * 8) * 2) : VN_OOB);
    rv[i] = vn_buf_load16(rsV, ok ? (uint32_t)((row * a.ldv + h * D + cc * 8) * 2) : VN_OOB);
    rd[i] = vn_buf_load16(rsdO, ok ? (uint32_t)((row * a.lddo + h * D + cc * 8) * 2) : VN_OOB);
  }
  float lreg = INFINITY;
  if (tid < NR && tid < N) lreg = a.lse_in[((long long)b * a.H + h) * N + tid] * LOG2E;
  zero_lds(smem, 4 * TILE);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx / C::DCH, cc = idx - r * C::DCH;
    if (idx < NR * C::DCH) {
      *reinterpret_cast<u32x4*>(Qs + r * C::ROW + cc * 16) = rq[i];
      *reinterpret_cast<u32x4*>(Ks + r * C::ROW + cc * 16) = rk[i];
      *reinterpret_cast<u32x4*>(Vs + r * C::ROW + cc * 16) = rv[i];
      *reinterpret_cast<u32x4*>(dOs + r * C::ROW + cc * 16) = rd[i];
    }
  }
  if (tid < NR) lses[tid] = lreg;
  __syncthreads();

  const float c = a.scale * LOG2E;
  const int row = wave * 32 + l31;  // this lane's query (first role) / key (second role)
  const bool rok = row < N;
  // ---- role 1: dQ of queries 32w..32w+31 (attn_dq_kernel's body over 32-key blocks) ----
  if (wave < 3) {
    half8 qf[C::KS], dof[C::KS];
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int ch = ks * 2 + h2;
      qf[ks] = as_half8(*reinterpret_cast<const u32x4*>(Qs + row * C::ROW + ch * 16));
      dof[ks] = as_half8(*reinterpret_cast<const u32x4*>(dOs + row * C::ROW + ch * 16));
      const uint32_t offo = (rok && ch < C::DCH) ? (uint32_t)((((long long)b * N + row) * a.ldo + h * D + ch * 8) * 2) : VN_OOB;
      const half8 of = as_half8(vn_buf_load16(rsO, offo));
#pragma unroll
      for (int j = 0; j < 8; ++j) part += (float)dof[ks][j] * (float)of[j];
    }
    const float dlt = part + __shfl_xor(part, 32, 64);  // delta[q] = sum_d dO[q][d] * O[q][d]
    if (h2 == 0) dels[row] = dlt;
    const float lse2 = lses[row];
    f32x16 dq[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[i][e] = 0.f;
    const int nkb = a.causal ? wave + 1 : 3;
    for (int aa = 0; aa < nkb; ++aa) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] = 0.f;
        dp[e] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int off = (aa * 32 + l31) * C::ROW + (ks * 2 + h2) * 16;
        half8 kf = as_half8(*reinterpret_cast<const u32x4*>(Ks + off));
        half8 vf = as_half8(*reinterpret_cast<const u32x4*>(Vs + off));
        s = VN_MFMA_32x32x16(kf, qf[ks], s, 0, 0, 0);
        dp = VN_MFMA_32x32x16(vf, dof[ks], dp, 0, 0, 0);
      }
      if (aa * 32 + 32 > N || (a.causal && aa == wave)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = aa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
          const bool ok = key < N && (!a.causal || key <= row);
          if (!ok) s[r] = -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = pk_fma(f32x2{s[r], s[r + 1]}, f32x2{c, c}, f32x2{-lse2, -lse2});
        const f32x2 g = f32x2{dp[r], dp[r + 1]} - f32x2{dlt, dlt};
        const f32x2 ds = f32x2{fast_exp2(t.x), fast_exp2(t.y)} * g;
        s[r] = ds.x;
        s[r + 1] = ds.y;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half8 pf = cvt8(s, j);
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
          half8 tf = load_tr(Ks, C::ROW, db * 32, aa * 32 + 16 * j + 4 * h2, lane);
          dq[db] = VN_MFMA_32x32x16(tf, pf, dq[db], 0, 0, 0);
        }
      }
    }
    if (rok) {
      half_t* orow = a.dQ + ((long long)b * N + row) * a.lddq + h * D;
#pragma unroll
      for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d = db * 32 + 8 * qd + 4 * h2;
          if (d < D) {
            half4 v = {(half_t)(dq[db][4 * qd] * a.scale), (half_t)(dq[db][4 * qd + 1] * a.scale),
                       (half_t)(dq[db][4 * qd + 2] * a.scale), (half_t)(dq[db][4 * qd + 3] * a.scale)};
            *reinterpret_cast<half4*>(orow + d) = v;
          }
        }
    }
  }
  __syncthreads();  // every query's delta is in LDS
  // ---- role 2: dK, dV of keys 32w..32w+31 (attn_dkv_kernel's body over 32-query blocks) ----
  if (wave < 3) {
    half8 kf[C::KS], vf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int ch = ks * 2 + h2;
      kf[ks] = as_half8(*reinterpret_cast<const u32x4*>(Ks + row * C::ROW + ch * 16));
      vf[ks] = as_half8(*reinterpret_cast<const u32x4*>(Vs + row * C::ROW + ch * 16));
    }
    f32x16 dk[C::DB], dv[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        dk[i][e] = 0.f;
        dv[i][e] = 0.f;
      }
    for (int hq = a.causal ? wave : 0; hq < 3; ++hq) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] = 0.f;
        dp[e] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int off = (hq * 32 + l31) * C::ROW + (ks * 2 + h2) * 16;
        half8 qfr = as_half8(*reinterpret_cast<const u32x4*>(Qs + off));
        half8 dofr = as_half8(*reinterpret_cast<const u32x4*>(dOs + off));
        s = VN_MFMA_32x32x16(qfr, kf[ks], s, 0, 0, 0);
        dp = VN_MFMA_32x32x16(dofr, vf[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 l4 = *reinterpret_cast<const f32x4*>(&lses[hq * 32 + 8 * qd + 4 * h2]);
        f32x4 d4 = *reinterpret_cast<const f32x4*>(&dels[hq * 32 + 8 * qd + 4 * h2]);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const int r = 4 * qd + e;
          const f32x2 t = pk_fma(f32x2{s[r], s[r + 1]}, f32x2{c, c}, -f32x2{l4[e], l4[e + 1]});
          f32x2 p = f32x2{fast_exp2(t.x), fast_exp2(t.y)};
          if (a.causal) {
            const int qq = hq * 32 + 8 * qd + 4 * h2 + e;
            if (row > qq) p.x = 0.f;
            if (row > qq + 1) p.y = 0.f;
          }
          const f32x2 ds = p * (f32x2{dp[r], dp[r + 1]} - f32x2{d4[e], d4[e + 1]});
          s[r] = p.x;
          s[r + 1] = p.y;
          dp[r] = ds.x;
          dp[r + 1] = ds.y;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half8 pf = cvt8(s, j);
        half8 dsf = cvt8(dp, j);
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
          half8 dot = load_tr(dOs, C::ROW, db * 32, hq * 32 + 16 * j + 4 * h2, lane);
          half8 qtf = load_tr(Qs, C::ROW, db * 32, hq * 32 + 16 * j + 4 * h2, lane);
          dv[db] = VN_MFMA_32x32x16(dot, pf, dv[db], 0, 0, 0);
          dk[db] = VN_MFMA_32x32x16(qtf, dsf, dk[db], 0, 0, 0);
        }
      }
    }
    if (rok) {
      half_t* krow = a.dK + ((long long)b * N + row) * a.lddk + h * D;
      half_t* vrow = a.dV + ((long long)b * N + row) * a.lddv + h * D;
#pragma unroll
      for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d = db * 32 + 8 * qd + 4 * h2;
          if (d < D) {
            half4 k4 = {(half_t)(dk[db][4 * qd] * a.scale), (half_t)(dk[db][4 * qd + 1] * a.scale),
                        (half_t)(dk[db][4 * qd + 2] * a.scale), (half_t)(dk[db][4 * qd + 3] * a.scale)};
            half4 v4 = {(half_t)dv[db][4 * qd], (half_t)dv[db][4 * qd + 1], (half_t)dv[db][4 * qd + 2],
                        (half_t)dv[db][4 * qd + 3]};
            *reinterpret_cast<half4*>(krow + d) = k4;
            *reinterpret_cast<half4*>(vrow + d) = v4;
          }
        }
    }
  }
}

// sum the q-split partials of the dK/dV kernel, apply the softmax scale to dK, round to f16
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(AttnArgs a) {
  const int Cc = a.H * a.D;
  const long long rows = (long long)a.Bn * a.Nk;
  const long long plane = rows * Cc;  // < 2^31 (77-key cross attention: a few hundred thousand): 32-bit index arithmetic
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= (unsigned)(plane / 4)) return;
  const unsigned e32 = gid * 4u;
  const unsigned row32 = e32 / (unsigned)Cc;
  const int c = (int)(e32 - row32 * (unsigned)Cc);
  const long long e = e32, row = row32;
  f32x4 k = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.qsplit; ++s) {
    k += *reinterpret_cast<const f32x4*>(a.ws + ((long long)s * 2) * plane + e);
    v += *reinterpret_cast<const f32x4*>(a.ws + ((long long)s * 2 + 1) * plane + e);
  }
  half4 k4 = {(half_t)(k[0] * a.scale), (half_t)(k[1] * a.scale), (half_t)(k[2] * a.scale), (half_t)(k[3] * a.scale)};
  half4 v4 = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  *reinterpret_cast<half4*>(a.dK + row * a.lddk + c) = k4;
  *reinterpret_cast<half4*>(a.dV + row * a.lddv + c) = v4;
}

int check_common(int Bn, int H, int Nq, int Nk, int D) {
  if (Bn <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return -1;
  if (!(D == 40 || D == 64 || D == 80 || D == 160)) return -2;
  return 0;
}

#define DISPATCH_D(KERNEL, grid, st, args)                                                   \
  switch (args.D) {                                                                          \
    case 40: hipLaunchKernelGGL((KERNEL<40>), grid, dim3(256), 0, st, args); break;          \
    case 64: hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), 0, st, args); break;          \
    case 80: hipLaunchKernelGGL((KERNEL<80>), grid, dim3(256), 0, st, args); break;          \
    default: hipLaunchKernelGGL((KERNEL<160>), grid, dim3(256), 0, st, args); break;         \
  }

#define DISPATCH_D1(KERNEL, grid, st, args)                                                     \
  switch (args.D) {                                                                             \
    case 40: hipLaunchKernelGGL((KERNEL<40, 1>), grid, dim3(256), 0, st, args); break;          \
    case 64: hipLaunchKernelGGL((KERNEL<64, 1>), grid, dim3(256), 0, st, args); break;          \
    case 80: hipLaunchKernelGGL((KERNEL<80, 1>), grid, dim3(256), 0, st, args); break;          \
    default: hipLaunchKernelGGL((KERNEL<160, 1>), grid, dim3(256), 0, st, args); break;         \
  }

}  // namespace

extern "C" int vneti_attn_fwd(const void* Q, long long ldq, const void* K, long long ldk, const void* V,
                              long long ldv, void* O, long long ldo, float* lse, int Bn, int H, int Nq, int Nk,
                              int D, float scale, int causal, void* stream) {
  int rc = check_common(Bn, H, Nq, Nk, D);
  VN_REQUIRE(rc == 0, "attn_fwd: unsupported shape B=%d H=%d Nq=%d Nk=%d D=%d", Bn, H, Nq, Nk, D);
  VN_REQUIRE(Q && K && V && O, "attn_fwd: null pointer");
  VN_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attn_fwd: bad strides");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = (const half_t*)Q;
  a.K = (const half_t*)K;
  a.V = (const half_t*)V;
  a.O = (half_t*)O;
  a.lse = lse;
  a.ldq = ldq;
  a.ldk = ldk;
  a.ldv = ldv;
  a.ldo = ldo;
  a.Bn = Bn;
  a.H = H;
  a.Nq = Nq;
  a.Nk = Nk;
  a.D = D;
  a.scale = scale;
  a.causal = causal;
  hipStream_t st = (hipStream_t)stream;
  // (QW = 2 — two query sub-tiles per wave, half the K/V fragment reads per FLOP at half the waves per SIMD — measured
  // 176-183 us against 176-178 for QW = 1 on the N = 4096, d = 40 self-attention; not instantiated)
  dim3 grid(cdiv(Nq, 128), H, Bn);
  DISPATCH_D1(attn_fwd_kernel, grid, st, a);
  return vneti_check_launch("attn_fwd");
}

extern "C" int vneti_attn_bwd_delta(const void* dO, long long lddo, const void* O, long long ldo, float* delta,
                                    int Bn, int H, int Nq, int D, void* stream) {
  VN_REQUIRE(dO && O && delta && Bn > 0 && H > 0 && Nq > 0 && D % 8 == 0, "attn_bwd_delta: bad arguments");
  long long total = (long long)Bn * Nq * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)dO, lddo, (const half_t*)O, ldo, delta, Bn, H, Nq, D);
  return vneti_check_launch("attn_bwd_delta");
}

extern "C" int vneti_attn_bwd_dq(const void* Q, long long ldq, const void* K, long long ldk, const void* V,
                                 long long ldv, const void* dO, long long lddo,
                                 const float* lse, float* delta, const void* O, long long ldo, void* dQ,
                                 long long lddq, int Bn, int H, int Nq, int Nk, int D, float scale, int causal,
                                 void* stream) {
  int rc = check_common(Bn, H, Nq, Nk, D);
  VN_REQUIRE(rc == 0, "attn_bwd_dq: unsupported shape B=%d H=%d Nq=%d Nk=%d D=%d", Bn, H, Nq, Nk, D);
  VN_REQUIRE(Q && K && V && dO && lse && delta && dQ, "attn_bwd_dq: null pointer");
  VN_REQUIRE(!O || ldo % 8 == 0, "attn_bwd_dq: ldo must be a multiple of 8");
  VN_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0,
             "attn_bwd_dq: bad strides");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = (const half_t*)Q;
  a.K = (const half_t*)K;
  a.V = (const half_t*)V;
  a.dO = (const half_t*)dO;
  a.dQ = (half_t*)dQ;
  a.lse_in = lse;
  a.delta = delta;
  a.delta_out = delta;
  a.O_in = (const half_t*)O;
  a.ldo = ldo;
  a.ldq = ldq;
  a.ldk = ldk;
  a.ldv = ldv;
  a.lddo = lddo;
  a.lddq = lddq;
  a.Bn = Bn;
  a.H = H;
  a.Nq = Nq;
  a.Nk = Nk;
  a.D = D;
  a.scale = scale;
  a.causal = causal;
  dim3 grid(cdiv(Nq, 128), H, Bn);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_D(attn_dq_kernel, grid, st, a);
  return vneti_check_launch("attn_bwd_dq");
}

extern "C" int vneti_attn_bwd_dkv(const void* Q, long long ldq, const void* K, long long ldk, const void* V,
                                  long long ldv, const void* dO, long long lddo, const float* lse, const float* delta,
                                  void* dK, long long lddk, void* dV, long long lddv, int Bn, int H, int Nq,
                                  int Nk, int D, float scale, int causal, float* ws, long long ws_floats,
                                  void* stream) {
  int rc = check_common(Bn, H, Nq, Nk, D);
  VN_REQUIRE(rc == 0, "attn_bwd_dkv: unsupported shape B=%d H=%d Nq=%d Nk=%d D=%d", Bn, H, Nq, Nk, D);
  VN_REQUIRE(Q && K && V && dO && lse && delta && dK && dV, "attn_bwd_dkv: null pointer");
  VN_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && lddk % 4 == 0 && lddv % 4 == 0,
             "attn_bwd_dkv: bad strides");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = (const half_t*)Q;
  a.K = (const half_t*)K;
  a.V = (const half_t*)V;
  a.dO = (const half_t*)dO;
  a.dK = (half_t*)dK;
  a.dV = (half_t*)dV;
  a.lse_in = lse;
  a.delta = delta;
  a.ldq = ldq;
  a.ldk = ldk;
  a.ldv = ldv;
  a.lddo = lddo;
  a.lddk = lddk;
  a.lddv = lddv;
  a.Bn = Bn;
  a.H = H;
  a.Nq = Nq;
  a.Nk = Nk;
  a.D = D;
  a.scale = scale;
  a.causal = causal;
  // few key blocks (cross-attention: Nk = 77) => split the query range so the chip is filled
  const int nkb = cdiv(Nk, 128), nqt = cdiv(Nq, 32 * DKV_QH(D));
  long long blocks = (long long)nkb * H * Bn;
  int qsplit = 1;
  // (also at exactly one 4-wave block per CU, the 32x32-latent self-attention: one wave per SIMD is latency-bound,
  //  64 -> 57 us with the query range split in two, reduce launch included)
  if (ws && !causal && blocks <= 256 && nqt >= 8) {
    qsplit = (int)(512 / blocks);
    if (qsplit > nqt / 4) qsplit = nqt / 4;
    if (qsplit > 32) qsplit = 32;
    const long long plane = (long long)Bn * Nk * H * D;
    // (the partials are stored through 32-bit buffer offsets: keep the slab under 4 GiB as well)
    while (qsplit > 1 && (2LL * qsplit * plane > ws_floats || 8LL * qsplit * plane >= (1LL << 32) - 64)) --qsplit;
    if (qsplit < 2) qsplit = 1;
  }
  a.qsplit = qsplit;
  a.ws = ws;
  dim3 grid(nkb * qsplit, H, Bn);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_D(attn_dkv_kernel, grid, st, a);
  if (qsplit > 1) {
    long long n4 = (long long)Bn * Nk * H * D / 4;
    VN_REQUIRE(n4 * 4 < 0x7fffffffLL, "attn_bwd_dkv: q-split reduce indexes with 32 bits");
    hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)cdivl(n4, 256)), dim3(256), 0, st, a);
  }
  return vneti_check_launch("attn_bwd_dkv");
}

extern "C" int vneti_attn_bwd_small(const void* Q, long long ldq, const void* K, long long ldk, const void* V, long long ldv,
                                    const void* dO, long long lddo, const void* O, long long ldo, const float* lse, void* dQ,
                                    long long lddq, void* dK, long long lddk, void* dV, long long lddv, int Bn, int H, int N,
                                    int D, float scale, int causal, void* stream) {
  VN_REQUIRE(Q && K && V && dO && O && lse && dQ && dK && dV, "attn_bwd_small: null pointer");
  VN_REQUIRE(Bn > 0 && H > 0 && N > 0, "attn_bwd_small: bad shape B=%d H=%d N=%d", Bn, H, N);
  if (N > 96 || D != 64) return VNETI_EUNSUP;  // longer sequences / other head sizes: vneti_attn_bwd_dq + vneti_attn_bwd_dkv
  VN_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && ldo % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 &&
                 lddv % 4 == 0, "attn_bwd_small: bad strides");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = (const half_t*)Q;
  a.K = (const half_t*)K;
  a.V = (const half_t*)V;
  a.dO = (const half_t*)dO;
  a.O_in = (const half_t*)O;
  a.lse_in = lse;
  a.dQ = (half_t*)dQ;
  a.dK = (half_t*)dK;
  a.dV = (half_t*)dV;
  a.ldq = ldq;
  a.ldk = ldk;
  a.ldv = ldv;
  a.lddo = lddo;
  a.ldo = ldo;
  a.lddq = lddq;
  a.lddk = lddk;
  a.lddv = lddv;
  a.Bn = Bn;
  a.H = H;
  a.Nq = N;
  a.Nk = N;
  a.D = D;
  a.scale = scale;
  a.causal = causal;
  hipLaunchKernelGGL((attn_bwd_small_kernel<64>), dim3(H, Bn), dim3(256), 0, (hipStream_t)stream, a);
  return vneti_check_launch("attn_bwd_small");
}
