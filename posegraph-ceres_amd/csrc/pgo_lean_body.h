// pgo_lean_body.h — the lean linearisation as a BODY (one lane per incidence slot, the per-incidence algebra of pgo_lin_lean.h), for the
// kernels that run it: k_linearize_lean (pgo_lean_kernels.hip: a symmetric-form session, blocks to their places g.sym_dst[t] of the form)
// and the LIN launch of the resident universal stream (pgo_uni_resident.h k_res_lin_lean: blocks to their own incidence slots of the BSR,
// TO_BSR).  Included INSIDE namespace pgo { namespace { of both translation units (after pgo_kernels.h, pgo_lin_lean.h, pgo_wave.h).
// Both destinations share the packed 27-entry slot layout (pgo_kernels.h), so the only difference is where a slot's block goes.
constexpr int LEAN_NV = 27;           // 21 diagonal-block entries + 6 gradient entries per incidence

struct LeanSink {
  double* lpair;       // the 27 doubles of this lane's even/odd pair in LDS
  double* val;         // the form (uniform)
  unsigned off;        // byte offset of this slot's first pair (the pairs of a slot are 64 pairs = 1 KiB apart)
  bool store, odd, nt;
  double wv[28];
  __device__ __forceinline__ void blk(int k, double x) { wv[k] = x; }
  // lane 2i + 1 adds lane 2i's value (row_shr:1) and writes the pair's sum
  __device__ __forceinline__ void dia(int k, double x) {
    const double y = x + dpp_shifted<0x111, 0xf>(x);
    if (odd) lpair[k] = y;
  }
  // quadrants complete in the order [9, 18), [0, 9), [18, 27): the pairs (2 kk, 2 kk + 1) that became whole
  __device__ __forceinline__ void blk_ready(int lo, int) {
    const int k0 = lo == 9 ? 5 : lo == 0 ? 0 : 9, k1 = lo == 9 ? 9 : lo == 0 ? 5 : 14;
    if (lo == 18) wv[27] = 0.0;
    if (store) {
#pragma unroll
      for (int kk = 0; kk < BLK_PAIRS_PACKED; ++kk)
        if (kk >= k0 && kk < k1) {
          double* q = reinterpret_cast<double*>(reinterpret_cast<char*>(val) + (off + 1024u * kk));
          // nontemporal: 310 MB written once and next read from HBM by the CG product anyway (measured 179 -> 163 us against plain
          // stores; nontemporal LOADS of the streamed inputs change nothing)
          typedef double v2d __attribute__((ext_vector_type(2)));
          v2d x;
          x.x = wv[2 * kk]; x.y = wv[2 * kk + 1];
          if (nt) __builtin_nontemporal_store(x, reinterpret_cast<v2d*>(q));
          else *reinterpret_cast<v2d*>(q) = x;
        }
    }
  }
};

// a slot's indices
struct LeanIdx {
  uint8_t side;
  int row, col, dst;
};
// everything else the slot's arithmetic reads: measurement and information (addressed by the slot), the poses of the edge a -> b
// (BEGIN slots sit in row a, END slots in row b), scales and constant masks of the row's and the column's pose (addressed by the indices)
template <int INFO>
struct LeanData {
  double ms[7];
  double wp[INFO == 2 ? 6 : 3], wr[INFO == 2 ? 6 : 3];
  double2 a0, a1, a2, a3, b0, b1, b2, b3, sr0, sr1, sr2, sc0, sc1, sc2;
  uint8_t m_own, m_oth;
};
// uniform base + 32-bit byte offset: the global_load form with the base in scalar registers and ONE address register per lane
// (a 64-bit address per plane and lane costs the kernel 30 registers it does not have)
template <class T>
__device__ __forceinline__ T ld_at(const void* base, unsigned byte_off) { return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off); }
template <bool TO_BSR>
__device__ __forceinline__ LeanIdx lean_load_idx(const DeviceGraph& g, int t) {
  LeanIdx l;
  l.side = ld_at<uint8_t>(g.slot_side, (unsigned)t);
  l.row = ld_at<int>(g.slot_row, 4u * (unsigned)t); l.col = ld_at<int>(g.slot_col, 4u * (unsigned)t);
  l.dst = TO_BSR ? t : ld_at<int>(g.sym_dst, 4u * (unsigned)t);
  return l;
}
template <int INFO>
__device__ __forceinline__ LeanData<INFO> lean_load_data(const DeviceGraph& g, int t, const LeanIdx& l) {
  LeanData<INFO> m;
  const size_t ns = (size_t)g.n_slots;
  const bool edge = l.side <= SIDE_END, begin = l.side == SIDE_BEGIN;
  const unsigned row = edge ? (unsigned)l.row : 0u, col = edge ? (unsigned)l.col : 0u;
  const unsigned oa = (unsigned)(POSE_STRIDE * sizeof(double)) * (begin ? row : col), ob = (unsigned)(POSE_STRIDE * sizeof(double)) * (begin ? col : row);
  m.a0 = ld_at<double2>(g.pose_x, oa); m.a1 = ld_at<double2>(g.pose_x, oa + 16); m.a2 = ld_at<double2>(g.pose_x, oa + 32); m.a3 = ld_at<double2>(g.pose_x, oa + 48);
  m.b0 = ld_at<double2>(g.pose_x, ob); m.b1 = ld_at<double2>(g.pose_x, ob + 16); m.b2 = ld_at<double2>(g.pose_x, ob + 32); m.b3 = ld_at<double2>(g.pose_x, ob + 48);
  // plane k of the measurement / information arrays: one base, byte offset 8 (k n_slots + t) (launch_linearize_lean: 21 planes are below 4 GiB)
  const unsigned tb = 8u * (unsigned)t, pb = 8u * (unsigned)ns;
#pragma unroll
  for (int k = 0; k < 7; ++k) m.ms[k] = ld_at<double>(g.smeas, tb + (unsigned)k * pb);
  if (INFO == 3) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { m.wp[i] = ld_at<double>(g.sW, tb + (unsigned)lean_upper(i, i) * pb); m.wr[i] = ld_at<double>(g.sW, tb + (unsigned)lean_upper(3 + i, 3 + i) * pb); }
  } else if (INFO == 2) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) { m.wp[k] = ld_at<double>(g.sW, tb + (unsigned)lean_upper(i, j) * pb); m.wr[k] = ld_at<double>(g.sW, tb + (unsigned)lean_upper(3 + i, 3 + j) * pb); ++k; }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) { m.wp[i] = 1.0; m.wr[i] = 1.0; }
  }
  const unsigned osr = 48u * row, osc = 48u * col;
  m.sr0 = ld_at<double2>(g.scale, osr); m.sr1 = ld_at<double2>(g.scale, osr + 16); m.sr2 = ld_at<double2>(g.scale, osr + 32);
  m.sc0 = ld_at<double2>(g.scale, osc); m.sc1 = ld_at<double2>(g.scale, osc + 16); m.sc2 = ld_at<double2>(g.scale, osc + 32);
  m.m_own = ld_at<uint8_t>(g.cmask, row); m.m_oth = ld_at<uint8_t>(g.cmask, col);
  return m;
}

// the arithmetic of one slot: block to the form, the 27 row values (pair sums) to LDS.  EVERY lane runs it — the DPP pair sum needs
// both lanes of a pair in the same instruction: a diagonal or padding slot computes on pose 0 against itself with the identity
// measurement its arrays hold (finite numbers) and zero scales, so its 27 values are exact zeros.
template <int INFO, bool TO_BSR>
__device__ __forceinline__ void lean_slot(const DeviceGraph& g, const LeanIdx& c1, const LeanData<INFO>& c2, double* lpair, bool odd) {
  const bool edge = c1.side <= SIDE_END, begin = c1.side == SIDE_BEGIN;
  const V3 pa{c2.a0.x, c2.a0.y, c2.a1.x}, pb{c2.b0.x, c2.b0.y, c2.b1.x};
  const Q4 qa{c2.a1.y, c2.a2.x, c2.a2.y, c2.a3.x}, qb{c2.b1.y, c2.b2.x, c2.b2.y, c2.b3.x};
  // constant parameter blocks: scale 0 (the scales are finite: 1 / (1 + sqrt(d)))
  const double op = (!edge || (c2.m_own & 1)) ? 0.0 : 1.0, oq = (!edge || (c2.m_own & 2)) ? 0.0 : 1.0;
  const double tp = (c2.m_oth & 1) ? 0.0 : 1.0, tq = (c2.m_oth & 2) ? 0.0 : 1.0;
  const double so[6] = {op * c2.sr0.x, op * c2.sr0.y, op * c2.sr1.x, oq * c2.sr1.y, oq * c2.sr2.x, oq * c2.sr2.y};
  const double st[6] = {tp * c2.sc0.x, tp * c2.sc0.y, tp * c2.sc1.x, tq * c2.sc1.y, tq * c2.sc2.x, tq * c2.sc2.y};
  const double mo[6] = {op, op, op, oq, oq, oq};
  LeanSink sink;
  sink.lpair = lpair;
  sink.odd = odd;
  sink.store = edge && c1.dst >= 0 && !PGO_ABLATION(g, 2);      // (bit 2: timing ablation, no block stores)
  const int d0 = sink.store ? c1.dst : 0;
  sink.val = TO_BSR ? g.bsr_val : g.sym_val;
  // (the form or the BSR of a large graph is written once and next read from HBM: nontemporal; the BSR of a universal-stream session —
  // at most 600 k slots, pgo_lm.cpp universal_wanted — is read back within microseconds: plain stores)
  sink.nt = !TO_BSR || g.n_slots > 600000;
  sink.off = (unsigned)(d0 >> 6) * (unsigned)(TILE_DOUBLES * sizeof(double)) + (unsigned)(d0 & 63) * 16u;      // (launch_linearize_lean: the form is below 4 GiB)
  lean_incidence<INFO>(begin, pa, qa, pb, qb, V3{c2.ms[0], c2.ms[1], c2.ms[2]}, Q4{c2.ms[3], c2.ms[4], c2.ms[5], c2.ms[6]}, c2.wp, c2.wr, so, st, mo,
                       g.loss_kind, g.loss_a, sink);
}

// value k of the row's 27: upper-triangle entry (i, j) of the diagonal block (unit diagonal in constant dimensions: keeps them
// decoupled and the block SPD) or gradient entry k - 21
__device__ __forceinline__ void lean_store_row_value(const DeviceGraph& g, int rw, int k, double s, uint8_t cmk) {
  if (k < 21) {
    int i = 0, base = 0;
    while (k >= base + (6 - i)) { base += 6 - i; ++i; }
    const int j = i + (k - base);
    if (i == j && ((i < 3) ? (cmk & 1) : (cmk & 2))) s = 1.0;
    g.Hdiag[36 * (size_t)rw + 6 * i + j] = s;
    g.Hdiag[36 * (size_t)rw + 6 * j + i] = s;
  } else {
    g.grad[6 * (size_t)rw + (k - 21)] = s;
  }
}

// the sum of value k over the pairs [p0, p1] of the chunk
__device__ __forceinline__ double lean_pair_sum(const double* lds, int p0, int p1, int k) {
  const double* q = lds + (p0 * LEAN_NV + k);
  double s0 = 0.0, s1 = 0.0;
  int j = p0;
  for (; j < p1; j += 2, q += 2 * LEAN_NV) { s0 += q[0]; s1 += q[LEAN_NV]; }
  if (j == p1) s0 += q[0];
  return s0 + s1;
}

// lds: LEAN_NV doubles per lane PAIR (odd stride: conflict-free 8-byte accesses) = LEAN_NV * blockDim.x / 2 doubles
template <int INFO, bool TO_BSR>
__device__ __forceinline__ void lean_linearize_body(const DeviceGraph& g, double* lds) {
  const int B = blockDim.x, tid = threadIdx.x, wg = blockIdx.x;
  double* lpair = lds + (size_t)(tid >> 1) * LEAN_NV;
  const int s_begin = g.wg_slot_begin[wg], s_end = g.wg_slot_begin[wg + 1];
  const int r0 = g.wg_row_begin[wg], nrows = g.wg_row_begin[wg + 1] - r0;
  const bool single = (s_end - s_begin) == B;
  int pre_rb = 0, pre_rc = 0;      // the row bookkeeping of this lane's first sum, requested before anything else
  uint8_t pre_cm = 0;
  if (tid < nrows * LEAN_NV) { const int row = r0 + tid / LEAN_NV; pre_rb = g.row_slot_begin[row]; pre_rc = g.row_slot_cnt[row]; pre_cm = g.cmask[row]; }
  double acc = 0.0;
  for (int cb = s_begin; cb < s_end; cb += B) {
    const LeanIdx c1 = lean_load_idx<TO_BSR>(g, cb + tid);
    const LeanData<INFO> c2 = lean_load_data<INFO>(g, cb + tid, c1);
    lean_slot<INFO, TO_BSR>(g, c1, c2, lpair, tid & 1);
    if (PGO_ABLATION(g, 4)) continue;      // (timing ablation: no row sums)
    __syncthreads();
    for (int idx = tid; idx < nrows * LEAN_NV; idx += B) {
      const int rl = idx / LEAN_NV, k = idx - rl * LEAN_NV, rw = r0 + rl;
      const bool first = idx == tid && cb == s_begin;
      const int rb = first ? pre_rb : g.row_slot_begin[rw], rc = first ? pre_rc : g.row_slot_cnt[rw];
      // slots [sb, se) of this chunk; a pair that holds the row's first slot in its odd lane belongs to the row before
      const int sb = max(rb, cb) - cb, se = min(rb + rc, cb + B) - cb;
      const double sum = lean_pair_sum(lds, (sb + 1) >> 1, (se - 1) >> 1, k);
      if (single) lean_store_row_value(g, rw, k, sum, first ? pre_cm : g.cmask[rw]);
      else acc += sum;
    }
    __syncthreads();
  }
  if (!single && tid < LEAN_NV) lean_store_row_value(g, r0, tid, acc, g.cmask[r0]);      // a row that fills several chunks: the work-group holds this one row
}

