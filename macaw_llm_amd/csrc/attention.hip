// Fused (flash-style) attention forward for gfx950: softmax(scale * Q K^T + mask) V without
// materialising the score matrix (at S = 2048 the fp32 score tensor is 537 MB per
// sample per layer).  bf16 in/out, fp32 softmax state, head_dim 64 (CLIP / Whisper) or 128
// (LLaMA).  Replaces modeling.py:197-215 and the HF encoder attention for the no-grad paths
// (frozen towers, inference); the training path of the LLaMA layers keeps the GEMM + softmax
// formulation until the fused backward lands.
//
// Work decomposition: block = 4 waves, each wave owns 32 query rows; the block walks the keys
// in tiles of 64 staged in LDS (K: [key][d] XOR-swizzled for ds_read_b128; V: [key][d] read
// back TRANSPOSED with ds_read_b64_tr_b16).  Both products use v_mfma_f32_32x32x16_bf16 with
// the operands swapped so that a lane owns ONE query column:
//   S^T[key][q] = K Q^T      lane (q = l&31, half = l>>5) holds 16 keys of its row
//   O^T[d][q]  += V^T P^T    P^T fragments are exactly the lane's own S registers (the MFMA
//                            k-slot <-> key mapping is chosen to match, no cross-lane shuffle)
// Row max / sum need one __shfl_xor(.., 32) with the partner half.
#include "common.h"
#include "../../include/macaw_hip.h"

namespace {

struct FlashArgs {
  const bf16* q; const bf16* k; const bf16* v; bf16* o; float* lse; const int32_t* kmask;
  int B, H, Lq, Lk;
  long q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs;
  float scale;
};

template <int HD>
MK_DEV int k_off(int row, int chunk) {  // K tile [64][HD] bf16, 16-B chunk swizzle
  if constexpr (HD == 128) return row * 256 + ((chunk ^ (row & 15)) << 4);
  else return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
template <int HD>
MK_DEV int v_off(int key, int chunk) {  // V tile [64][HD] bf16, read through tr_b16
  if constexpr (HD == 128) return key * 256 + ((chunk ^ (4 * (key & 3))) << 4);
  else return key * 128 + ((chunk ^ (4 * ((key >> 1) & 1))) << 4);
}

// Loads are issued UNCONDITIONALLY from a clamped row and zeroed by a select afterwards: a load
// under `if (row < n)` sits in its own basic block, the compiler then waits (vmcnt(0)) for each
// 16-byte chunk before issuing the next one -- eight serialised HBM round trips per tile, which
// was ~2/3 of the run time of every fused-attention kernel at S = 144.
MK_DEV uint4 ld16_or_zero(const bf16* rowptr_clamped, bool ok) {
  uint4 v = *reinterpret_cast<const uint4*>(rowptr_clamped);
  if (!ok) v = make_uint4(0, 0, 0, 0);
  return v;
}
MK_DEV bf16x8 ld8_or_zero(const bf16* rowptr_clamped, bool ok) {
  const uint4 v = ld16_or_zero(rowptr_clamped, ok);
  return __builtin_bit_cast(bf16x8, v);
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(FlashArgs a) {
  constexpr int KT = 64;                 // keys per LDS tile
  constexpr int NKD = HD / 16;           // MFMA k-steps over d for Q K^T
  constexpr int NDB = HD / 32;           // 32-wide d blocks of the output
  constexpr int CPR = HD / 8;            // 16-B chunks per row
  __shared__ __attribute__((aligned(16))) char lds[2 * KT * HD * 2 + KT * 4];
  char* ldsK = lds;
  char* ldsV = lds + KT * HD * 2;
  int* ldsM = reinterpret_cast<int*>(lds + 2 * KT * HD * 2);   // key validity of the tile
  // grid = (H, B, blocks): the sequence-block index is the SLOWEST dimension, so blocks of equal
  // cost are dispatched together (heaviest first) instead of one heavy + one light block per
  // (b, h) alternating -- at S = 144 the second block has 16 rows and a quarter of the work, and
  // pairing them made every dispatch round as long as a heavy block (LPT order: -35 %).
  const int b = blockIdx.y, h = blockIdx.x;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = l >> 5, lq = l & 31;
  const int blk = CAUSAL ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
  const int q0 = blk * 128 + w * 32;
  const int qg = q0 + lq;                // this lane's query row
  const bf16* Q = a.q + (long)b * a.q_bs + (long)h * HD;
  const bf16* K = a.k + (long)b * a.k_bs + (long)h * HD;
  const bf16* V = a.v + (long)b * a.v_bs + (long)h * HD;
  const int32_t* km = a.kmask ? a.kmask + (long)b * a.Lk : nullptr;

  // Q fragments (B operand of K Q^T): lane holds Q[qg][16*kd + 8*half .. +8]
  bf16x8 qf[NKD];
#pragma unroll
  for (int kd = 0; kd < NKD; ++kd) {
    qf[kd] = ld8_or_zero(Q + (long)min(qg, a.Lq - 1) * a.q_ld + 16 * kd + 8 * half, qg < a.Lq);
  }
  f32x16 oacc[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  // Online softmax state in the exp2 domain: m_run = running max of t = s * scale * log2(e), l_run =
  // sum of 2^(t - m_run).  (v_exp_f32 IS exp2: folding scale and log2(e) into one FMA with the max
  // removes a multiply per score; the natural-log lse is recovered at the end.)
  float m_run = -INFINITY, l_run = 0.f;
  const float c2 = a.scale * 1.4426950408889634f;
  // Lazy rescaling (cdna_hip_programming.md T13): the accumulators are rescaled only when some row's
  // maximum grew by more than 2^DEFER since the last rescale; otherwise the old maximum is kept and
  // p = 2^(t - m_old) <= 2^DEFER.  Every quantity still at the old maximum (O, l) is rescaled in the
  // same place, before this sub-block's P exists, and all earlier P V products are complete (the
  // code is sequential per wave): the "textbook order", no pending-tile hazard.  DEFER = 0 rescales
  // on every growth (the round-2 behaviour).
  constexpr float DEFER = 6.0f;          // 2^6 = 64: p, l stay far inside fp32 / bf16 range

  // keys this block has to visit (causal: up to the last query row of the block)
  const int shift = a.Lk - a.Lq;         // query i may see keys <= i + shift
  int k_end = a.Lk;
  if (CAUSAL) k_end = min(a.Lk, blk * 128 + 128 + shift);
  const int ntiles = (k_end + KT - 1) / KT;

  for (int kt = 0; kt < ntiles; ++kt) {
    const int kbase = kt * KT;
    __syncthreads();                     // previous tile fully consumed
    // ---- stage K and V tiles (zero fill beyond Lk) ----
    {
      constexpr int NCH = (KT * CPR) / 256;
      uint4 kv4[NCH], vv4[NCH];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {    // all 2 * NCH loads in flight, then one wait
        const int c = threadIdx.x + 256 * i;
        const int row = c / CPR, ch = c % CPR;
        const int kg = kbase + row;
        const int kc = min(kg, a.Lk - 1);
        kv4[i] = ld16_or_zero(K + (long)kc * a.k_ld + ch * 8, kg < a.Lk);
        vv4[i] = ld16_or_zero(V + (long)kc * a.v_ld + ch * 8, kg < a.Lk);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = threadIdx.x + 256 * i;
        const int row = c / CPR, ch = c % CPR;
        *reinterpret_cast<uint4*>(ldsK + k_off<HD>(row, ch)) = kv4[i];
        *reinterpret_cast<uint4*>(ldsV + v_off<HD>(row, ch)) = vv4[i];
      }
    }
    if (threadIdx.x < KT) {
      const int kg = kbase + threadIdx.x;
      ldsM[threadIdx.x] = (kg < a.Lk) && (!km || km[kg] != 0);
    }
    __syncthreads();
    if (q0 >= a.Lq) continue;            // wave has no rows (still takes part in the barriers)
    // Interior tile: every key of the tile is valid and visible to every query row of this wave --
    // no mask arithmetic at all (at S = 2048 all but the last two tiles of a block).  Wave-uniform.
    bool interior = (__builtin_amdgcn_ballot_w64(ldsM[l] != 0) == ~0ull);
    if (CAUSAL) interior = interior && (kbase + KT - 1 <= q0 + shift);

#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      // ---- S^T = K Q^T for 32 keys x 32 queries ----
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + k_off<HD>(sb * 32 + lq, 2 * kd + half));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kd], s, 0, 0, 0);
      }
      // ---- mask + online softmax (this lane: query qg, keys key(r)) ----
      float mx = -INFINITY;
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] *= c2; mx = fmaxf(mx, s[r]); }
      } else {
        int mk[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int4 m4 = *reinterpret_cast<const int4*>(ldsM + sb * 32 + 8 * g4 + 4 * half);
          mk[4 * g4] = m4.x; mk[4 * g4 + 1] = m4.y; mk[4 * g4 + 2] = m4.z; mk[4 * g4 + 3] = m4.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kg = kbase + sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          bool ok = mk[r] != 0;
          if (CAUSAL) ok = ok && (kg <= qg + shift);
          s[r] = ok ? s[r] * c2 : -INFINITY;
          mx = fmaxf(mx, s[r]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // rescale only if some row of the wave needs it (NaN-safe: -inf - -inf compares false -> rescale)
      if (!__all((mx - m_run <= DEFER) ? 1 : 0)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
      }
      float ps = 0.f;
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_run); ps += s[r]; }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = (s[r] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(s[r] - m_run);
          ps += s[r];
        }
      }
      ps += __shfl_xor(ps, 32, 64);
      l_run += ps;
      bf16x8 pf[2];
#pragma unroll
      for (int e = 0; e < 8; ++e) { pf[0][e] = (bf16)s[e]; pf[1][e] = (bf16)s[8 + e]; }
      // ---- O^T += V^T P^T ----
      const int li = l & 15;
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const int col = d * 32 + 16 * ((l >> 4) & 1) + 4 * (li & 3);   // d index of this lane group
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 vf;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int key = sb * 32 + ks * 16 + 8 * r + 4 * half + (li >> 2);
            const int off = v_off<HD>(key, col >> 3) + ((col & 7) << 1);
            const bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(ldsV + off));
            vf[4 * r] = t[0]; vf[4 * r + 1] = t[1]; vf[4 * r + 2] = t[2]; vf[4 * r + 3] = t[3];
          }
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], oacc[d], 0, 0, 0);
        }
      }
    }
  }
  if (qg >= a.Lq) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  bf16* O = a.o + (long)b * a.o_bs + (long)qg * a.o_ld + (long)h * HD;
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      bf16x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = (bf16)(oacc[d][4 * q4 + e] * inv);
      *reinterpret_cast<bf16x4*>(O + d * 32 + 8 * q4 + 4 * half) = ov;
    }
  if (a.lse && half == 0)   // natural-log lse = (m2 + log2(l)) * ln(2)
    a.lse[((long)b * a.H + h) * a.Lq + qg] =
        (l_run > 0.f) ? (m_run + __log2f(l_run)) * 0.6931471805599453f : -INFINITY;
}

}  // namespace

extern "C" int mk_flash_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                                 const int32_t* kmask, int32_t B, int32_t H, int32_t Lq,
                                 int32_t Lk, int32_t hd, int64_t q_ld, int64_t q_bs, int64_t k_ld,
                                 int64_t k_bs, int64_t v_ld, int64_t v_bs, int64_t o_ld,
                                 int64_t o_bs, float scale, int32_t causal, int32_t dtype,
                                 void* stream) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return MK_ERR_BAD_ARG;
  if (dtype != MK_BF16 || (hd != 64 && hd != 128)) return MK_ERR_UNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                       reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o);
  if ((al & 15) || (q_ld % 8) || (k_ld % 8) || (v_ld % 8) || (o_ld % 8) || (q_bs % 8) ||
      (k_bs % 8) || (v_bs % 8) || (o_bs % 8))
    return MK_ERR_UNSUPPORTED;
  FlashArgs a;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o;
  a.lse = lse; a.kmask = kmask;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.q_ld = q_ld; a.q_bs = q_bs; a.k_ld = k_ld; a.k_bs = k_bs; a.v_ld = v_ld; a.v_bs = v_bs;
  a.o_ld = o_ld; a.o_bs = o_bs;
  a.scale = scale;
  dim3 grid(H, B, mk_cdiv(Lq, 128)), block(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // algorithmic FLOPs: QK^T and PV, 2 * Lq * Lk * hd each per (b, h); causal counts the lower
  // triangle only (what a masked dense formulation would not skip is not work)
  const double pairs = causal ? ((double)Lq * Lk - 0.5 * (double)min(Lq, Lk) * (min(Lq, Lk) - 1)) : (double)Lq * Lk;
  const int prof = mkp::begin(st, 1, 4.0 * pairs * hd * B * H, Lq, Lk, hd, B * H, causal, 0);
  if (hd == 128) {
    if (causal) MK_LAUNCH((flash_fwd_kernel<128, true>), grid, block, 0, st, a);
    else MK_LAUNCH((flash_fwd_kernel<128, false>), grid, block, 0, st, a);
  } else {
    if (causal) MK_LAUNCH((flash_fwd_kernel<64, true>), grid, block, 0, st, a);
    else MK_LAUNCH((flash_fwd_kernel<64, false>), grid, block, 0, st, a);
  }
  mkp::end(prof, st);
  return mk_check_launch();
}

// =====================================================================================
// Fused attention BACKWARD (recompute form): given q, k, v, o, do and the forward's
// log-sum-exp, produce dq, dk, dv without ever storing P.  Three kernels:
//   flash_bwd_prep : Dv[b,h,q] = sum_d do[q,d] * o[q,d]                       (HBM-bound)
//   flash_bwd_dq   : block = 128 queries, walks the key tiles  (lane <-> query, as forward)
//   flash_bwd_dkv  : block = 128 keys,    walks the query tiles (lane <-> key)
// In both MFMA kernels every product is arranged so that the softmax-shaped tile (P, dS) is
// consumed as the B operand straight out of the accumulator registers that produced it:
//   dq kernel : S^T = K Q^T, dP^T = V dO^T  -> dS^T (lane = query)  -> dQ^T += K^T dS^T
//   dkv kernel: S   = Q K^T, dP   = dO V^T  -> P, dS (lane = key)   -> dV^T += dO^T P,
//                                                                     dK^T += Q^T dS
// with the transposed operands (K^T, dO^T, Q^T) read from row-major LDS tiles through
// ds_read_b64_tr_b16 using the same k-slot <-> row mapping as the accumulator layout.
// =====================================================================================
namespace {

struct FlashBwdArgs {
  const bf16* q; const bf16* k; const bf16* v; const bf16* o; const bf16* dout;
  bf16* dq; bf16* dk; bf16* dv;
  const float* lse; float* dvec; const int32_t* kmask;
  int B, H, Lq, Lk;
  long q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs;  // dq/dk/dv/do share q/k/v/o geometry
  float scale;
};

// D[b, h, q] = sum_d dO * O.  HD / 8 lanes per row, 16-byte loads; consecutive lane groups take
// consecutive HEADS of one token (contiguous in memory), so a wave reads whole 128-byte lines.
template <int HD>
__global__ __launch_bounds__(256) void flash_bwd_prep_kernel(FlashBwdArgs a) {
  constexpr int LPR = HD / 8;                       // lanes per (token, head) row
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;   // (b, q, h) flattened, h fastest
  const int c = threadIdx.x % LPR;
  const long total = (long)a.B * a.Lq * a.H;
  const bool ok = g < total;
  const long gc = ok ? g : total - 1;
  const int h = (int)(gc % a.H);
  const long bq = gc / a.H;
  const int qi = (int)(bq % a.Lq);
  const long b = bq / a.Lq;
  const long off = b * a.o_bs + (long)qi * a.o_ld + (long)h * HD + c * 8;
  float o[8], d[8];
  VecIO<bf16>::load(a.o + off, o);
  VecIO<bf16>::load(a.dout + off, d);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += o[e] * d[e];
#pragma unroll
  for (int m = LPR / 2; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
  if (ok && c == 0) a.dvec[(b * a.H + h) * a.Lq + qi] = s;
}

// tile stored twice: [rows][HD] with the b128 swizzle (k_off) and with the tr swizzle (v_off)
template <int HD, int ROWS>
MK_DEV void stage_rows(const bf16* src, long ld, int row0, int nrows_valid, char* lds_b128,
                       char* lds_tr) {
  constexpr int CPR = HD / 8, N = (ROWS * CPR) / 256;
  uint4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {          // all loads in flight before the first LDS write
    const int c = threadIdx.x + 256 * i;
    const int row = c / CPR, ch = c % CPR;
    v[i] = ld16_or_zero(src + (long)min(row0 + row, nrows_valid - 1) * ld + ch * 8,
                        row0 + row < nrows_valid);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int row = c / CPR, ch = c % CPR;
    if (lds_b128) *reinterpret_cast<uint4*>(lds_b128 + k_off<HD>(row, ch)) = v[i];
    if (lds_tr) *reinterpret_cast<uint4*>(lds_tr + v_off<HD>(row, ch)) = v[i];
  }
}
// two tensors at once (one wait for both)
template <int HD, int ROWS>
MK_DEV void stage_rows2(const bf16* s0, long ld0, char* b0, char* t0, const bf16* s1, long ld1,
                        char* b1, char* t1, int row0, int nrows_valid) {
  constexpr int CPR = HD / 8, N = (ROWS * CPR) / 256;
  uint4 v0[N], v1[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int row = c / CPR, ch = c % CPR;
    const long r = min(row0 + row, nrows_valid - 1);
    v0[i] = ld16_or_zero(s0 + r * ld0 + ch * 8, row0 + row < nrows_valid);
    v1[i] = ld16_or_zero(s1 + r * ld1 + ch * 8, row0 + row < nrows_valid);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int row = c / CPR, ch = c % CPR;
    if (b0) *reinterpret_cast<uint4*>(b0 + k_off<HD>(row, ch)) = v0[i];
    if (t0) *reinterpret_cast<uint4*>(t0 + v_off<HD>(row, ch)) = v0[i];
    if (b1) *reinterpret_cast<uint4*>(b1 + k_off<HD>(row, ch)) = v1[i];
    if (t1) *reinterpret_cast<uint4*>(t1 + v_off<HD>(row, ch)) = v1[i];
  }
}

// A-operand fragment X^T[d-block rows][k-slots <-> rows r0 + kappa] from a row-major LDS tile
template <int HD>
MK_DEV bf16x8 tr_frag(const char* lds_tr, int rowbase16, int dblk) {
  const int l = threadIdx.x & 63, li = l & 15, half = l >> 5;
  const int col = dblk * 32 + 16 * ((l >> 4) & 1) + 4 * (li & 3);
  bf16x8 f;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = rowbase16 + 8 * r + 4 * half + (li >> 2);
    const int off = v_off<HD>(row, col >> 3) + ((col & 7) << 1);
    const bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)(lds_tr + off));
    f[4 * r] = t[0]; f[4 * r + 1] = t[1]; f[4 * r + 2] = t[2]; f[4 * r + 3] = t[3];
  }
  return f;
}

// ------------------------------------------------------------------ dQ kernel --
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void flash_bwd_dq_kernel(FlashBwdArgs a) {
  constexpr int KT = 64, NKD = HD / 16, NDB = HD / 32;
  __shared__ __attribute__((aligned(16))) char lds[3 * KT * HD * 2 + KT * 4];
  int* ldsM = reinterpret_cast<int*>(lds + 3 * KT * HD * 2);
  char* ldsK = lds;                      // K tile, b128 image (A operand of S^T = K Q^T)
  char* ldsKt = lds + KT * HD * 2;       // K tile, tr image   (K^T for dQ^T += K^T dS^T)
  char* ldsV = lds + 2 * KT * HD * 2;    // V tile, b128 image (A operand of dP^T = V dO^T)
  // grid = (H, B, blocks), heaviest sequence blocks dispatched first (see flash_fwd_kernel)
  const int b = blockIdx.y, h = blockIdx.x;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = l >> 5, lq = l & 31;
  const int blk = CAUSAL ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
  const int q0 = blk * 128 + w * 32;
  const int qg = q0 + lq;
  const bf16* Q = a.q + (long)b * a.q_bs + (long)h * HD;
  const bf16* dO = a.dout + (long)b * a.o_bs + (long)h * HD;
  const bf16* K = a.k + (long)b * a.k_bs + (long)h * HD;
  const bf16* V = a.v + (long)b * a.v_bs + (long)h * HD;
  const int32_t* km = a.kmask ? a.kmask + (long)b * a.Lk : nullptr;
  bf16x8 qf[NKD], dof[NKD];
#pragma unroll
  for (int kd = 0; kd < NKD; ++kd) {
    const int qc = min(qg, a.Lq - 1);
    qf[kd] = ld8_or_zero(Q + (long)qc * a.q_ld + 16 * kd + 8 * half, qg < a.Lq);
    dof[kd] = ld8_or_zero(dO + (long)qc * a.o_ld + 16 * kd + 8 * half, qg < a.Lq);
  }
  const long rowid = ((long)b * a.H + h) * a.Lq + min(qg, a.Lq - 1);
  const float lse = a.lse[rowid], dv_ = a.dvec[rowid];
  f32x16 acc[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
  const int shift = a.Lk - a.Lq;
  int k_end = a.Lk;
  if (CAUSAL) k_end = min(a.Lk, blk * 128 + 128 + shift);
  const int ntiles = (k_end + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int kbase = kt * KT;
    __syncthreads();
    stage_rows2<HD, KT>(K, a.k_ld, ldsK, ldsKt, V, a.v_ld, ldsV, nullptr, kbase, a.Lk);
    if (threadIdx.x < KT) {
      const int kg = kbase + threadIdx.x;
      ldsM[threadIdx.x] = (kg < a.Lk) && (!km || km[kg] != 0);
    }
    __syncthreads();
    if (q0 >= a.Lq) continue;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + k_off<HD>(sb * 32 + lq, 2 * kd + half));
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(ldsV + k_off<HD>(sb * 32 + lq, 2 * kd + half));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kd], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[kd], dp, 0, 0, 0);
      }
      bf16x8 dsf[2];
      int mk[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int4 m4 = *reinterpret_cast<const int4*>(ldsM + sb * 32 + 8 * g4 + 4 * half);
        mk[4 * g4] = m4.x; mk[4 * g4 + 1] = m4.y; mk[4 * g4 + 2] = m4.z; mk[4 * g4 + 3] = m4.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kg = kbase + sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        bool ok = mk[r] != 0;
        if (CAUSAL) ok = ok && (kg <= qg + shift);
        const float e = __expf(fminf(s[r] * a.scale - lse, 30.f));
        const float p = ok ? e : 0.f;
        const float ds = p * (dp[r] - dv_) * a.scale;
        dsf[r >> 3][r & 7] = (bf16)ds;
      }
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              tr_frag<HD>(ldsKt, sb * 32 + ks * 16, d), dsf[ks], acc[d], 0, 0, 0);
    }
  }
  if (qg >= a.Lq) return;
  bf16* DQ = a.dq + (long)b * a.q_bs + (long)qg * a.q_ld + (long)h * HD;
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      bf16x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = (bf16)acc[d][4 * q4 + e];
      *reinterpret_cast<bf16x4*>(DQ + d * 32 + 8 * q4 + 4 * half) = ov;
    }
}

// ---------------------------------------------------------------- dK/dV kernel --
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(FlashBwdArgs a) {
  constexpr int QT = 64, NKD = HD / 16, NDB = HD / 32;
  __shared__ __attribute__((aligned(16))) char lds[4 * QT * HD * 2 + 2 * QT * 4];
  char* ldsQ = lds;                      // Q tile b128 image  (A operand of S  = Q K^T)
  char* ldsQt = lds + QT * HD * 2;       // Q tile tr image    (Q^T for dK^T += Q^T dS)
  char* ldsD = lds + 2 * QT * HD * 2;    // dO tile b128 image (A operand of dP = dO V^T)
  char* ldsDt = lds + 3 * QT * HD * 2;   // dO tile tr image   (dO^T for dV^T += dO^T P)
  float* ldsLse = reinterpret_cast<float*>(lds + 4 * QT * HD * 2);
  float* ldsDv = ldsLse + QT;
  // grid = (H, B, blocks), heaviest sequence blocks dispatched first (see flash_fwd_kernel)
  const int b = blockIdx.y, h = blockIdx.x;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = l >> 5, lk = l & 31;
  const int blk = blockIdx.z;
  const int k0 = blk * 128 + w * 32;
  const int kg = k0 + lk;                // this lane's key
  const bf16* Q = a.q + (long)b * a.q_bs + (long)h * HD;
  const bf16* dO = a.dout + (long)b * a.o_bs + (long)h * HD;
  const bf16* K = a.k + (long)b * a.k_bs + (long)h * HD;
  const bf16* V = a.v + (long)b * a.v_bs + (long)h * HD;
  const int32_t* km = a.kmask ? a.kmask + (long)b * a.Lk : nullptr;
  bf16x8 kf[NKD], vf[NKD];               // B operands: lane holds K/V[kg][16*kd + 8*half .. +8]
#pragma unroll
  for (int kd = 0; kd < NKD; ++kd) {
    const int kc = min(kg, a.Lk - 1);
    kf[kd] = ld8_or_zero(K + (long)kc * a.k_ld + 16 * kd + 8 * half, kg < a.Lk);
    vf[kd] = ld8_or_zero(V + (long)kc * a.v_ld + 16 * kd + 8 * half, kg < a.Lk);
  }
  const bool key_ok = (kg < a.Lk) && (!km || km[kg < a.Lk ? kg : 0] != 0);
  f32x16 dka[NDB], dva[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dka[d][e] = 0.f; dva[d][e] = 0.f; }
  const int shift = a.Lk - a.Lq;
  // queries that can see this block's keys: q >= key - shift
  int q_begin = 0;
  if (CAUSAL) q_begin = max(0, blk * 128 - shift) / QT * QT;
  const long rowbase = ((long)b * a.H + h) * a.Lq;
  // This kernel runs one workgroup per CU (354 registers), so nothing else hides the latency of
  // the next query tile: its Q / dO chunks and lse / D values are fetched into registers while
  // the current tile is being multiplied, and only written to LDS at the top of the next round.
  constexpr int CPRq = HD / 8, NCH = (QT * CPRq) / 256;
  uint4 pq[NCH], pd[NCH];
  float plse = 0.f, pdv = 0.f;
  auto prefetch = [&](int qt) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + 256 * i;
      const int row = c / CPRq, ch = c % CPRq;
      const long r = min(qt + row, a.Lq - 1);
      pq[i] = ld16_or_zero(Q + r * a.q_ld + ch * 8, qt + row < a.Lq);
      pd[i] = ld16_or_zero(dO + r * a.o_ld + ch * 8, qt + row < a.Lq);
    }
    const int qi = qt + (threadIdx.x & (QT - 1));
    const long rid = rowbase + min(qi, a.Lq - 1);
    plse = qi < a.Lq ? a.lse[rid] : 0.f;     // select after an unconditional load (clamped row)
    pdv = qi < a.Lq ? a.dvec[rid] : 0.f;
  };
  prefetch(q_begin);
  for (int qt = q_begin; qt < a.Lq; qt += QT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + 256 * i;
      const int row = c / CPRq, ch = c % CPRq;
      *reinterpret_cast<uint4*>(ldsQ + k_off<HD>(row, ch)) = pq[i];
      *reinterpret_cast<uint4*>(ldsQt + v_off<HD>(row, ch)) = pq[i];
      *reinterpret_cast<uint4*>(ldsD + k_off<HD>(row, ch)) = pd[i];
      *reinterpret_cast<uint4*>(ldsDt + v_off<HD>(row, ch)) = pd[i];
    }
    if (threadIdx.x < QT) {
      ldsLse[threadIdx.x] = plse;
      ldsDv[threadIdx.x] = pdv;
    }
    __syncthreads();
    if (qt + QT < a.Lq) prefetch(qt + QT);   // in flight during this tile's MFMAs
    if (k0 >= a.Lk) continue;
#pragma unroll 1
    for (int sb = 0; sb < 2; ++sb) {
      // S = Q K^T and dP = dO V^T for 32 queries x 32 keys: D[i = q][j = key]
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
      for (int kd = 0; kd < NKD; ++kd) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ldsQ + k_off<HD>(sb * 32 + lk, 2 * kd + half));
        const bf16x8 da = *reinterpret_cast<const bf16x8*>(ldsD + k_off<HD>(sb * 32 + lk, 2 * kd + half));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[kd], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[kd], dp, 0, 0, 0);
      }
      bf16x8 pf[2], dsf[2];
      // per-query lse / D loaded unconditionally as float4 (a load under the mask predicate
      // cannot be speculated and turned every element into its own branch)
      float lse_r[16], dv_r[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 l4 = *reinterpret_cast<const float4*>(ldsLse + sb * 32 + 8 * g4 + 4 * half);
        const float4 d4 = *reinterpret_cast<const float4*>(ldsDv + sb * 32 + 8 * g4 + 4 * half);
        lse_r[4 * g4] = l4.x; lse_r[4 * g4 + 1] = l4.y; lse_r[4 * g4 + 2] = l4.z; lse_r[4 * g4 + 3] = l4.w;
        dv_r[4 * g4] = d4.x; dv_r[4 * g4 + 1] = d4.y; dv_r[4 * g4 + 2] = d4.z; dv_r[4 * g4 + 3] = d4.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;   // query row inside the tile
        const int qi = qt + ql;
        bool ok = key_ok & (qi < a.Lq);
        if (CAUSAL) ok = ok & (kg <= qi + shift);
        // masked scores can be arbitrarily large: keep the exponent finite, then select
        const float e = __expf(fminf(s[r] * a.scale - lse_r[r], 30.f));
        const float p = ok ? e : 0.f;
        const float ds = p * (dp[r] - dv_r[r]) * a.scale;
        pf[r >> 3][r & 7] = (bf16)p;
        dsf[r >> 3][r & 7] = (bf16)ds;
      }
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          dva[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              tr_frag<HD>(ldsDt, sb * 32 + ks * 16, d), pf[ks], dva[d], 0, 0, 0);
          dka[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              tr_frag<HD>(ldsQt, sb * 32 + ks * 16, d), dsf[ks], dka[d], 0, 0, 0);
        }
    }
  }
  if (kg >= a.Lk) return;
  bf16* DK = a.dk + (long)b * a.k_bs + (long)kg * a.k_ld + (long)h * HD;
  bf16* DV = a.dv + (long)b * a.v_bs + (long)kg * a.v_ld + (long)h * HD;
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      bf16x4 ok_, ov_;
#pragma unroll
      for (int e = 0; e < 4; ++e) { ok_[e] = (bf16)dka[d][4 * q4 + e]; ov_[e] = (bf16)dva[d][4 * q4 + e]; }
      *reinterpret_cast<bf16x4*>(DK + d * 32 + 8 * q4 + 4 * half) = ok_;
      *reinterpret_cast<bf16x4*>(DV + d * 32 + 8 * q4 + 4 * half) = ov_;
    }
}

}  // namespace

extern "C" int mk_flash_attn_bwd(const void* q, const void* k, const void* v, const void* o,
                                 const void* dout, const float* lse, float* dvec, void* dq,
                                 void* dk, void* dv, const int32_t* kmask, int32_t B, int32_t H,
                                 int32_t Lq, int32_t Lk, int32_t hd, int64_t q_ld, int64_t q_bs,
                                 int64_t k_ld, int64_t k_bs, int64_t v_ld, int64_t v_bs,
                                 int64_t o_ld, int64_t o_bs, float scale, int32_t causal,
                                 int32_t dtype, void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !dvec || !dq || !dk || !dv || B <= 0 || H <= 0 ||
      Lq <= 0 || Lk <= 0)
    return MK_ERR_BAD_ARG;
  if (dtype != MK_BF16 || (hd != 64 && hd != 128)) return MK_ERR_UNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                       reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o) |
                       reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dq) |
                       reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv);
  if ((al & 15) || (q_ld % 8) || (k_ld % 8) || (v_ld % 8) || (o_ld % 8) || (q_bs % 8) ||
      (k_bs % 8) || (v_bs % 8) || (o_bs % 8))
    return MK_ERR_UNSUPPORTED;
  FlashBwdArgs a;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)o;
  a.dout = (const bf16*)dout; a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv;
  a.lse = lse; a.dvec = dvec; a.kmask = kmask;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.q_ld = q_ld; a.q_bs = q_bs; a.k_ld = k_ld; a.k_bs = k_bs; a.v_ld = v_ld; a.v_bs = v_bs;
  a.o_ld = o_ld; a.o_bs = o_bs;
  a.scale = scale;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long rows = (long)B * H * Lq;
  dim3 gq(H, B, mk_cdiv(Lq, 128)), gk(H, B, mk_cdiv(Lk, 128)), block(256);
  // backward: dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q (4 products) + the recomputed
  // QK^T: counted as the 4 algorithmic ones = 2x the forward
  const double pairs = causal ? ((double)Lq * Lk - 0.5 * (double)min(Lq, Lk) * (min(Lq, Lk) - 1)) : (double)Lq * Lk;
  const int prof = mkp::begin(st, 2, 8.0 * pairs * hd * B * H, Lq, Lk, hd, B * H, causal, 0);
#define MK_FB(HDV, CZ)                                                                         \
  do {                                                                                         \
    MK_LAUNCH((flash_bwd_prep_kernel<HDV>), dim3((unsigned)((rows * (HDV / 8) + 255) / 256)), block, 0, st, a); \
    MK_LAUNCH((flash_bwd_dq_kernel<HDV, CZ>), gq, block, 0, st, a);                            \
    MK_LAUNCH((flash_bwd_dkv_kernel<HDV, CZ>), gk, block, 0, st, a);                           \
  } while (0)
  if (hd == 128) { if (causal) MK_FB(128, true); else MK_FB(128, false); }
  else { if (causal) MK_FB(64, true); else MK_FB(64, false); }
#undef MK_FB
  mkp::end(prof, st);
  return mk_check_launch();
}
