// 16-bit (f16 / bf16) convolution weight-gradient kernel for gfx950.
//
//   dw[co][ci][tap] = sum over (b, y, x) of dy[b,y,x,co] * in[b,y+dy-1,x+dx-1,ci]        (aten conv backward-weight)
//
// GEMM view: D[row = cout][col = cin] per tap, K = pixels.  Both MFMA operands need 8 consecutive K (= pixels) per lane
// while the NHWC tiles in LDS are [pixel][channel]; the fragments are fetched with the gfx950 transposed LDS read
// ds_read_b64_tr_b16 (lane i of a 16-lane group receives column i of a 4-row x 16-column block; probe-verified).
//
// Structure (one 256-thread block, 4 waves, one wave per SIMD):
//   * block tile: 64 couts x BCI cins x all taps; K loop over the pixel tiles assigned to the block (split-K over the grid)
//   * LDS tiles are stored as 32-channel sub-tiles [c/32][pixel][32] (64-byte rows): every transposed read of a 32-lane
//     half touches all 64 banks exactly once (conflict-free), the 3x3 halo tile is shared by all 9 taps
//   * double-buffered LDS + register prefetch: global loads of tile t+1 are issued before the MFMAs of tile t and written
//     to the other LDS buffer afterwards; one barrier per tile
//   * each block writes its fp32 partial tile to a slab [split][tap][cout][cin] with coalesced stores; a second kernel
//     sums the slabs in a fixed order (bitwise reproducible, no float atomics) and adds into the fp32 gradient
#include <cstdio>
#include <type_traits>
#include <cstdlib>

#include "pu_kernels.h"

namespace pu {

// 16 zero bytes: what an out-of-image / out-of-range staging vector is loaded from (no select after the load)
__device__ __attribute__((aligned(16))) unsigned int g_wg_zero[4] = {0u, 0u, 0u, 0u};

typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4v* lds_s4_ptr;

template <typename T> struct MMW;
template <> struct MMW<f16> {
  typedef f16x8 Frag;
  __device__ static __forceinline__ f32x16 mfma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct MMW<bf16> {
  typedef bf16x8 Frag;
  __device__ static __forceinline__ f32x16 mfma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

typedef short s16x8v __attribute__((ext_vector_type(8)));
// one MFMA fragment = two transposed 8-byte LDS reads; the halves are joined with a vector shuffle (a register sequence, no copies -
// the union form cost a v_mov_b64 per fragment) and the pointers are lane-constant bases plus compile-time element offsets, which
// the backend folds into the instructions' offset fields (the first form re-derived every address on the VALU: ~200 integer
// instructions per tile and wave in an issue-bound loop)
template <typename Frag>
__device__ __forceinline__ Frag tr_frag(const uint16_t* p0, const uint16_t* p1) {
  const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p0));
  const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p1));
  const s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(Frag, v);
}

// BCO = couts per block (64, or 32 for the Cout <= 32 layers of the 256 x 256 level: half of a 64-cout tile would be zero padding)
// resident blocks per CU the register allocation is held to: the 4-wave forms run two blocks per CU (three for the 32 x 32 tile of the
// 256 x 256 level) - the interleaved K loop would otherwise take 264-284 registers and halve that
constexpr int wg16_min_blocks(int TH, int TW, int BCI, int NW, int BCO) {
  return NW != 4 || TH * TW < 128 || (BCI == 64 && BCO == 64) ? 1 : (BCI == 32 && BCO == 32 ? 3 : 2);
}
template <typename T, int KS, int TH, int TW, int BCI, int NW, int BCO>
__global__ __launch_bounds__(64 * NW, wg16_min_blocks(TH, TW, BCI, NW, BCO)) void conv_wgrad16_kernel(WgradArgs a) {
  typedef MMW<T> M;
  constexpr int NTH = 64 * NW;
  constexpr int TAPS = KS * KS, PADP = KS / 2;
  constexpr int BM = TH * TW;
  constexpr int IH = TH + 2 * PADP, IW = TW + 2 * PADP, NPH = IH * IW;
  constexpr int CIS = BCI / 32;                       // cin sub-tiles
  constexpr int COS = BCO / 32;                       // cout sub-tiles
  constexpr int BUF = (COS * BM + CIS * NPH) * 32;    // elements per LDS buffer
  constexpr int NVD_TOT = BM * (BCO / 8), NVD = (NVD_TOT + NTH - 1) / NTH;   // 16-byte vectors of the dy tile per thread
  constexpr int NVA_TOT = NPH * (BCI / 8);
  constexpr int NVA = (NVA_TOT + NTH - 1) / NTH;
  // wave roles: (cout sub-tile, cin sub-tile) x a tap group; the NW / (COS * CIS) tap groups split the taps evenly (two waves per
  // SIMD hide each other's LDS waits in the 8-wave forms; a group beyond the last tap only helps with the staging loads)
  constexpr int GROUPS = NW / (COS * CIS);
  constexpr int NJ = (TAPS + GROUPS - 1) / GROUPS;
  static_assert(NW % (COS * CIS) == 0 && GROUPS >= 1 && GROUPS <= 4, "wave roles");
  constexpr bool PHASED = NW == 8 && GROUPS == 2 && TAPS > NJ;   // both tap groups own taps: they alternate between multiplying and staging (tile loop)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint16_t* lds = reinterpret_cast<uint16_t*>(smem_raw);

  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  const int ct = wave % COS;
  const int it = (wave / COS) % CIS;
  const int tap0 = (wave / (COS * CIS)) * NJ;
  const int co0 = blockIdx.y * BCO, ci0 = blockIdx.z * BCI;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  const int ntiles = tiles_x * tiles_y * a.B;
  const uint16_t* dy = reinterpret_cast<const uint16_t*>(a.dy);
  const uint16_t* in = reinterpret_cast<const uint16_t*>(a.in);

  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  typedef int wg_i32x4 __attribute__((ext_vector_type(4)));   // (whole-vector values: the {w[4]} struct form was split into scalars and re-joined
  wg_i32x4 rd[NVD], ra[NVA];                                   //  with v_mov copies behind an s_waitcnt vmcnt right after the loads)
  // Tile-invariant staging plan of this thread (the loop is instruction-issue bound: SQ_ACTIVE_INST_ANY 45 % of the wave cycles at two
  // waves per SIMD, MFMA pipe 28 % busy - every integer instruction removed from the tile loop is time given to the matrix pipe):
  //   goff_d / goff_a: element offset of the vector relative to the tile's first pixel, 32-bit (pu_create bounds the tensors);
  //   loff_d / loff_a: its element offset inside an LDS buffer;  vmask: bit k / 16 + k = the slot exists (and its channels do);
  //   aedge: per halo vector the image edges it lies beyond when the tile touches them (bit 0 top, 1 bottom, 2 left, 3 right).
  // Out-of-image / out-of-range vectors are LOADED from a zero page: no branches, nothing post-processes a loaded vector.
  int goff_d[NVD], goff_a[NVA]; int loff_d[NVD], loff_a[NVA]; unsigned vmask = 0; uint64_t aedge = 0;
#pragma unroll
  for (int k = 0; k < NVD; ++k) {
    const int i = tid + k * NTH;
    const int ii = i < NVD_TOT ? i : 0;
    const int pix = ii / (BCO / 8), cv = ii % (BCO / 8);
    goff_d[k] = ((pix / TW) * a.W + (pix % TW)) * a.dy_ld + co0 + cv * 8;
    loff_d[k] = ((cv >> 2) * BM + pix) * 32 + (cv & 3) * 8;
    if (i < NVD_TOT && co0 + cv * 8 < a.Cout) vmask |= 1u << k;
  }
#pragma unroll
  for (int k = 0; k < NVA; ++k) {
    const int i = tid + k * NTH;
    const int ii = i < NVA_TOT ? i : 0;
    const int hp = ii / (BCI / 8), cv = ii % (BCI / 8);
    const int hy = hp / IW, hx = hp % IW;
    goff_a[k] = ((hy - PADP) * a.W + (hx - PADP)) * a.in_ld + ci0 + cv * 8;
    loff_a[k] = COS * BM * 32 + ((cv >> 2) * NPH + hp) * 32 + (cv & 3) * 8;
    if (i < NVA_TOT && ci0 + cv * 8 < a.Cin) vmask |= 1u << (16 + k);
    aedge |= (uint64_t)((hy < PADP ? 1u : 0u) | (hy >= TH + PADP ? 2u : 0u) | (hx < PADP ? 4u : 0u) | (hx >= TW + PADP ? 8u : 0u)) << (4 * k);
  }
  static_assert(NVD <= 16 && NVA <= 16, "staging plan bit fields");
  const ptrdiff_t zoff_d = reinterpret_cast<const uint16_t*>(g_wg_zero) - dy, zoff_a = reinterpret_cast<const uint16_t*>(g_wg_zero) - in;
  auto gload = [&](int tile) __attribute__((always_inline)) {
    int pt = tile;
    const int tx0 = (pt % tiles_x) * TW; pt /= tiles_x;
    const int ty0 = (pt % tiles_y) * TH; pt /= tiles_y;
    const int b = pt;
    const unsigned pbase = (unsigned)((b * a.H + ty0) * a.W + tx0);
    const unsigned dbase = pbase * (unsigned)a.dy_ld, abase = pbase * (unsigned)a.in_ld;
    const unsigned omask = (ty0 == 0 ? 1u : 0u) | (ty0 + TH >= a.H ? 2u : 0u) | (tx0 == 0 ? 4u : 0u) | (tx0 + TW >= a.W ? 8u : 0u);
#pragma unroll
    for (int k = 0; k < NVD; ++k) {
      const bool ok = (vmask >> k) & 1u;
      rd[k] = *reinterpret_cast<const wg_i32x4*>(dy + (ok ? (ptrdiff_t)(dbase + (unsigned)goff_d[k]) : zoff_d));
    }
#pragma unroll
    for (int k = 0; k < NVA; ++k) {
      const bool ok = ((vmask >> (16 + k)) & 1u) && (((unsigned)(aedge >> (4 * k)) & omask) == 0);
      ra[k] = *reinterpret_cast<const wg_i32x4*>(in + (ok ? (ptrdiff_t)(abase + (unsigned)goff_a[k]) : zoff_a));
    }
  };
  // one staged vector -> LDS: part p < NVD is a dy vector, the rest are halo-tile vectors (slots beyond the tile write nothing)
  auto lstore_part = [&](int buf, int p) __attribute__((always_inline)) {
    uint16_t* sb = lds + buf * BUF;
    if (p < NVD) {
      if (tid + p * NTH < NVD_TOT) *reinterpret_cast<wg_i32x4*>(sb + loff_d[p]) = rd[p];
    } else {
      const int k = p - NVD;
      if (tid + k * NTH < NVA_TOT) *reinterpret_cast<wg_i32x4*>(sb + loff_a[k]) = ra[k];
    }
  };
  constexpr int NPARTS = NVD + NVA;
  auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPARTS; ++p) lstore_part(buf, p);
  };

  // lane roles of the transposed reads
  const int g = l >> 4, h = l >> 5, q = (l & 15) >> 2, p = l & 3;
  const int cb = 16 * (g & 1) + 4 * p;

  // element offsets (inside one LDS buffer) of the two 4-pixel rows this lane supplies to a transposed read at K-step 0
  const int lq0 = 8 * h + q, lq1 = lq0 + 4;
  const int lane_d0 = (ct * BM + lq0) * 32 + cb, lane_d1 = (ct * BM + lq1) * 32 + cb;
  int lane_a0[NJ], lane_a1[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int t = tap0 + j;
    const int tt = t < TAPS ? t : 0;
    const int toff = (tt / KS) * IW + (tt % KS);
    lane_a0[j] = COS * BM * 32 + (it * NPH + (lq0 / TW) * IW + (lq0 % TW) + toff) * 32 + cb;
    lane_a1[j] = COS * BM * 32 + (it * NPH + (lq1 / TW) * IW + (lq1 % TW) + toff) * 32 + cb;
  }
  static_assert(TW == 8 || TW % 16 == 0, "K-step offsets of the halo tile");

  // fused bias gradient (ci-tile 0 blocks): thread (co = tid % BCO, part = tid / BCO) sums its pixels of every dy tile
  constexpr int BPARTS = NTH / BCO;
  const bool do_bias = a.dbias0 != nullptr && blockIdx.z == 0;
  float bsum = 0.f;

  int tile = blockIdx.x;
  int cur = 0;
  // the MFMA part of one tile out of LDS buffer `cur`, NV = taps this wave really owns (+ its share of the fused bias column sums)
  auto ktile = [&](int cur, auto nv, bool live) __attribute__((always_inline)) {
    constexpr int NV = decltype(nv)::value;
    // lane-constant fragment bases of this tile's buffer; a K-step adds a compile-time offset (pixel kk * 16 + lq of the tile is pixel lq
    // shifted by whole rows / a multiple of 16 columns: (kk * 16 + lq) % TW == kk * 16 % TW + lq for every tile shape used here)
    const uint16_t* bufp = lds + cur * BUF;
    const uint16_t* pD0 = bufp + lane_d0; const uint16_t* pD1 = bufp + lane_d1;
    const uint16_t* pA0[NV > 0 ? NV : 1]; const uint16_t* pA1[NV > 0 ? NV : 1];
#pragma unroll
    for (int j = 0; j < NV; ++j) { pA0[j] = bufp + lane_a0[j]; pA1[j] = bufp + lane_a1[j]; }
    if constexpr (NV > 0) {
      // software-pipelined over the K-steps: the transposed reads of step kk+1 are issued as one block before the MFMAs of step
      // kk (sched_barrier keeps the blocks apart), so they land under those MFMAs instead of stalling their own consumers
      typename M::Frag fa[2], fb[2][NV];
      auto load_step = [&](int kk, int slot) __attribute__((always_inline)) {
        const int dd = kk * 16 * 32;                                                        // dy tile: pixel rows of 32 elements
        const int da = (TW >= 16 ? ((kk * 16 / TW) * IW + (kk * 16 % TW)) : (kk * 16 / TW) * IW) * 32;   // halo tile
        fa[slot] = tr_frag<typename M::Frag>(pD0 + dd, pD1 + dd);
#pragma unroll
        for (int j = 0; j < NV; ++j) fb[slot][j] = tr_frag<typename M::Frag>(pA0[j] + da, pA1[j] + da);
      };
      // MFMAs of step kk and the transposed reads of step kk+1 are issued INTERLEAVED (one MFMA, then its share of the reads, under that
      // MFMA's 32 pipe cycles): a wave issues in order, so a block of 12 reads between two blocks of MFMAs leaves the matrix pipe idle
      // for the ~100 cycles the reads take to issue whenever the SIMD's other wave is not multiplying at that moment.
      if (live) {                                 // (scalar: a tap group entirely beyond the last tap only helps with the staging)
        constexpr int NRD = 2 + 2 * NV, RPM = (NRD + NV - 1) / NV;
        load_step(0, 0);
#pragma unroll
        for (int kk = 0; kk < BM / 16; ++kk) {
          __builtin_amdgcn_sched_barrier(0);
          if (kk + 1 < BM / 16) load_step(kk + 1, (kk + 1) & 1);
#pragma unroll
          for (int j = 0; j < NV; ++j) acc[j] = M::mfma(fa[kk & 1], fb[kk & 1][j], acc[j]);
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);          // its share of the next step's LDS reads
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (do_bias) {
      const int bc = tid % BCO, bp = tid / BCO;
      const uint16_t* col = lds + cur * BUF + ((bc >> 5) * BM) * 32 + (bc & 31);
#pragma unroll 8
      for (int pp = bp * (BM / BPARTS); pp < (bp + 1) * (BM / BPARTS); ++pp) {
        T v; *reinterpret_cast<uint16_t*>(&v) = col[pp * 32];
        bsum += ET<T>::ld(&v);
      }
    }
  };
  // taps of this wave's group (the guard `if (tap0 + j < TAPS)` that used to sit on every MFMA depends on the wave index, which the
  // compiler treats as divergent: each MFMA ended up in its own exec-masked basic block; the group is now picked once, by a scalar branch)
  const int grp = __builtin_amdgcn_readfirstlane(wave / (COS * CIS));
  auto ktile_grp = [&](int cur) __attribute__((always_inline)) {
    if constexpr (PHASED) {                       // two groups: exact tap counts (9 = 5 + 4), one instantiation each
      if (grp == 0) ktile(cur, std::integral_constant<int, (NJ < TAPS ? NJ : TAPS)>{}, true);
      else ktile(cur, std::integral_constant<int, (TAPS - NJ > 0 ? TAPS - NJ : 0)>{}, true);
    } else {
      // one instantiation (register allocation as before): every live group multiplies all NJ slots - a slot beyond the last tap
      // re-reads tap 0 and its result is never stored; a group entirely beyond the last tap (the fourth group of 9 = 3 + 3 + 3)
      // only helps with the staging
      ktile(cur, std::integral_constant<int, NJ>{}, grp * NJ < TAPS);
    }
  };
  const int G = gridDim.x;
  if constexpr (PHASED) {
    // Two tap groups = the two waves of every SIMD.  Stamps of the one-phase loop (all eight waves: load burst -> K-steps -> LDS stores
    // -> barrier) showed the matrix pipe busy only during the K-steps, about half of each tile: 17 % of the wave cycles went into
    // ISSUING the load burst (64 KB through the CU's 64-byte-per-clock vector memory path while nobody multiplies), 7 % into the LDS
    // stores, 22 % into the barrier.  Here the groups alternate: while group A multiplies tile t, group B writes its staged vectors of
    // tile t + 1 to the other LDS buffer and issues its loads of tile t + 2, and vice versa; two barriers per tile, every SIMD always
    // has one wave in its MFMA phase, and a load has a whole tile to arrive.
    if (tile < ntiles) { gload(tile); lstore(0); }
    if (tile + G < ntiles) gload(tile + G);
    __syncthreads();
    // (two separate loops - one per group, same barrier count - rather than `if (grp == 0) multiply else stage` twice inside one
    //  loop: with the branches inside the loop the register allocator put the staged vectors and the MFMA fragments of the two
    //  branches on the same registers and copied the loaded vectors away behind an s_waitcnt vmcnt right after issuing the loads)
    if (grp == 0) {
      for (; tile < ntiles; tile += G) {
        ktile_grp(cur);
        __syncthreads();
        if (tile + G < ntiles) lstore(cur ^ 1);
        if (tile + 2 * G < ntiles) gload(tile + 2 * G);
        __syncthreads();
        cur ^= 1;
      }
    } else {
      for (; tile < ntiles; tile += G) {
        if (tile + G < ntiles) lstore(cur ^ 1);
        if (tile + 2 * G < ntiles) gload(tile + 2 * G);
        __syncthreads();
        ktile_grp(cur);
        __syncthreads();
        cur ^= 1;
      }
    }
  } else {
    if (tile < ntiles) { gload(tile); lstore(0); }
    __syncthreads();
    for (; tile < ntiles; tile += G) {
      const int nxt = tile + G;
      if (nxt < ntiles) gload(nxt);
      ktile_grp(cur);
      if (nxt < ntiles) lstore(cur ^ 1);          // (the next tile's vectors in one burst after the K-steps; spreading them between the MFMA
                                                  //  groups of the last K-steps measured neutral to 5 % slower, profiles/r2_ab_runs.txt)
      __syncthreads();
      cur ^= 1;
    }
  }
  if (do_bias) {                                  // combine the pixel parts, one partial row per split
    float* red = reinterpret_cast<float*>(smem_raw);
    red[tid] = bsum;
    __syncthreads();
    if (tid < BCO) {
      const int cout_pad_b = gridDim.y * BCO;
      float* bslab = a.slab + (size_t)gridDim.x * TAPS * cout_pad_b * (gridDim.z * BCI) + (size_t)blockIdx.x * cout_pad_b;
      float sacc = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < BPARTS; ++w8) sacc += red[tid + BCO * w8];
      bslab[co0 + tid] = sacc;
    }
  }

  // ---- partial tile -> slab [split][tap][cout_pad][cin_pad] (cin contiguous: 128-byte coalesced rows)
  const int cout_pad = gridDim.y * BCO, cin_pad = gridDim.z * BCI;
  float* slab = a.slab + (size_t)blockIdx.x * TAPS * cout_pad * cin_pad;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int t = tap0 + j;
    if (t < TAPS) {
      const int ci = ci0 + it * 32 + (l & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        slab[((size_t)t * cout_pad + co) * cin_pad + ci] = acc[j][r];
      }
    }
  }
}

template <typename T, int KS, int TH, int TW, int BCI, int NW = 4, int WBCO = 64>
static hipError_t launch_wg16(const WgradArgs& a, hipStream_t s, WgradReduce* red) {
  constexpr int PADP = KS / 2, BM = TH * TW, NPH = (TH + 2 * PADP) * (TW + 2 * PADP), TAPS = KS * KS;
  constexpr size_t lds_tiles = (size_t)2 * ((WBCO / 32) * BM + (BCI / 32) * NPH) * 32 * 2;
  constexpr size_t lds = lds_tiles > (size_t)64 * NW * 4 ? lds_tiles : (size_t)64 * NW * 4;     // the bias combine reuses the buffer
  auto kern = conv_wgrad16_kernel<T, KS, TH, TW, BCI, NW, WBCO>;
  static AttrOnce attr_once;
  if (!attr_once.cur()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_once.cur() = true;
  }
  const int ntiles = (a.W / TW) * (a.H / TH) * a.B;
  const int gy = cdiv(a.Cout, WBCO), gz = cdiv(a.Cin, BCI);
  const long per_split = (long)TAPS * gy * WBCO * gz * BCI + (long)gy * WBCO;   // + one bias row
  int split = cdiv(lds > 80 * 1024 ? 256 : 512, gy * gz);   // resident blocks per CU: 1 (85 KB LDS) / 2 (<= 80 KB); fewer, longer
                                                         // K ranges keep the slab traffic (split x tile) below the operand traffic
  if (split > ntiles) split = ntiles;
  if ((long)split * per_split > a.slab_floats) split = (int)(a.slab_floats / per_split);
  if (split < 1) return hipErrorInvalidValue;
  WgradArgs b = a;
  char tag[128];
  const bool prof = prof_enabled();
  if (prof) {
    snprintf(tag, sizeof tag, "conv_wgrad16_kernel<%s,%d,%d,%d,%d,%d,%d>", ET<T>::DT == 1 ? "f16" : "bf16", KS, TH, TW, BCI, NW, WBCO);
    const double px = (double)a.B * a.H * a.W;
    prof_record(tag, 2.0 * px * a.Cout * (double)a.Cin * a.taps, px * (a.Cin + a.Cout) * 2 + (double)a.Cout * a.Cin * a.taps * 4, s, true);
  }
  hipLaunchKernelGGL(kern, dim3(split, gy, gz), dim3(64 * NW), lds, s, b);
  if (prof) prof_record(tag, 0, 0, s, false);
  red->slab = a.slab; red->split = split; red->taps = TAPS; red->cout_pad = gy * WBCO; red->cin_pad = gz * BCI; red->Cout = a.Cout; red->Cin = a.Cin;
  red->dw = a.dw; red->inv_scale = a.inv_scale; red->db0 = a.dbias0; red->db1 = a.dbias1; red->inv_dev = a.inv_scale_dev; red->overwrite = a.overwrite;
  return hipGetLastError();
}

// dw[co][ci][tap] += inv_scale * sum_s slab[s][tap][co][ci]   (fixed order -> reproducible).  Second form of this reduction; what the
// profile of the first one (64 units x 4 split lanes per block, one unit per tap, bias rows in 64-cout x 4-lane blocks) showed:
//  * layers with small tiles and deep splits (32 x 32 x 9 tile, 512 slabs) ran 37 blocks whose threads each walked 128 dependent
//    loads (31.7 us for 18.9 MB);
//  * layers with large weight tensors (512 x 512 x 9 and up) spent their time in the 4-byte read-modify-write of dw at stride `taps`:
//    one 64-byte line of dw[co][ci][tap] was touched by nine different blocks (80 us for a 75 MB slab set).
// Here a block is UB = 256 / L units x L split lanes (L = 4 or 16, picked per launch so that the grid fills the chip), and with
// ALLT a unit is one (cout, 4 x cin) position whose thread walks ALL taps: its nine float4 sums are the 36 consecutive floats
// dw[co][ci .. ci+3][0 .. 8], written as nine 16-byte read-modify-writes.  Sums stay in a fixed order (lane-serial over the slabs,
// then a fixed pairwise tree over the lanes), so results are reproducible run to run.
template <int TAPS, int L, bool ALLT>
__global__ __launch_bounds__(256) void wgrad_reduce2_kernel(const float* __restrict__ slab, int split, int cout_pad, int cin_pad, int Cout, int Cin,
                                                             float* __restrict__ dw, float inv_scale, float* db0, float* db1,
                                                             const float* __restrict__ inv_dev, unsigned nwb, int vec_ok, int overwrite) {
  constexpr int UB = 256 / L, NT = ALLT ? TAPS : 1;
  __shared__ f32x4 red[NT][256];
  if (inv_dev) inv_scale *= inv_dev[0];            // device-chosen scale of this sub-graph (latent encoders, f16)
  const long per_tap = (long)cout_pad * cin_pad;
  const size_t stride = (size_t)TAPS * per_tap;
  if (blockIdx.x >= nwb) {
    // bias rows (slab tail [split][cout_pad]): 16 couts x 16 split lanes per block, eight loads in flight per thread.  (One block of
    // 64 couts x 4 lanes walking split / 4 dependent-latency loads was what a 512-slab launch actually waited for: 32 us.)
    float* bred = reinterpret_cast<float*>(&red[0][0]);
    const int co = (int)(blockIdx.x - nwb) * 16 + (threadIdx.x & 15);
    const int bl = threadIdx.x >> 4;
    const float* bs = slab + (size_t)split * stride + (co < cout_pad ? co : 0);
    float sb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sb[i] = 0.f;
    int k = bl;
    for (; k + 7 * 16 < split; k += 8 * 16) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = bs[(size_t)(k + i * 16) * cout_pad];
      __builtin_amdgcn_sched_barrier(0);            // loads first: left alone, the scheduler re-uses ONE register quad and issues load, wait, add eight times
#pragma unroll
      for (int i = 0; i < 8; ++i) sb[i] += v[i];
    }
    // the tail in ONE batch of loads from clamped rows (selected where they are added): `sb[0] += bs[...]` per remaining row was a chain of
    // dependent memory round trips - at the usual split of 64 (four rows per lane) the whole reduction, 10 us, was four latencies long
    if (k < split) {
      float v[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) { const int kk = k + i * 16; v[i] = bs[(size_t)(kk < split ? kk : k) * cout_pad]; }
      __builtin_amdgcn_sched_barrier(0);            // loads first: left alone, the scheduler re-uses ONE register quad and issues load, wait, add eight times
#pragma unroll
      for (int i = 0; i < 7; ++i) sb[i] += (k + i * 16 < split) ? v[i] : 0.f;
    }
    bred[threadIdx.x] = ((sb[0] + sb[1]) + (sb[2] + sb[3])) + ((sb[4] + sb[5]) + (sb[6] + sb[7]));
    __syncthreads();
    if (bl == 0 && co < Cout) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) v += bred[j * 16 + threadIdx.x];
      v *= inv_scale;
      if (overwrite) { db0[co] = v; if (db1) db1[co] = v; }
      else { db0[co] += v; if (db1) db1[co] += v; }
    }
    return;
  }
  const int u = threadIdx.x % UB, lane = threadIdx.x / UB;
  const long unit = (long)blockIdx.x * UB + u;
  const long nunits = (ALLT ? per_tap : (long)TAPS * per_tap) / 4;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int ci = 0, co = 0, t0 = 0;
  bool live = false;
  if (unit < nunits) {
    const long e = unit * 4;
    ci = (int)(e % cin_pad);
    const long q = e / cin_pad;
    co = (int)(q % cout_pad); t0 = (int)(q / cout_pad);      // t0 == 0 with ALLT
    live = co < Cout && ci < Cin;
    if (live) {
      const float* p = slab + e;
      if (ALLT && NT > 1) {
        int k = lane;
        for (; k + L < split; k += 2 * L) {                  // two slabs = 2 NT independent 16-byte loads in flight per trip
          const float* pk = p + (size_t)k * stride;
          const float* pk2 = pk + (size_t)L * stride;
          f32x4 v[NT], w[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) v[t] = *reinterpret_cast<const f32x4*>(pk + (size_t)t * per_tap);
#pragma unroll
          for (int t = 0; t < NT; ++t) w[t] = *reinterpret_cast<const f32x4*>(pk2 + (size_t)t * per_tap);
          __builtin_amdgcn_sched_barrier(0);            // loads first: left alone, the scheduler re-uses ONE register quad and issues load, wait, add eight times
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] += v[t];
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] += w[t];
        }
        if (k < split) {
          const float* pk = p + (size_t)k * stride;
          f32x4 v[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) v[t] = *reinterpret_cast<const f32x4*>(pk + (size_t)t * per_tap);
          __builtin_amdgcn_sched_barrier(0);            // loads first: left alone, the scheduler re-uses ONE register quad and issues load, wait, add eight times
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] += v[t];
        }
      } else {
        f32x4 sa[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sa[i] = acc[0];
        int k = lane;
        for (; k + 7 * L < split; k += 8 * L) {              // eight independent 16-byte loads in flight per trip
          f32x4 v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(k + i * L) * stride);
          __builtin_amdgcn_sched_barrier(0);            // loads first: left alone, the scheduler re-uses ONE register quad and issues load, wait, add eight times
#pragma unroll
          for (int i = 0; i < 8; ++i) sa[i] += v[i];
        }
        if (k < split) {                                     // tail: one batch from clamped slabs (see the bias rows above)
          f32x4 v[7];
#pragma unroll
          for (int i = 0; i < 7; ++i) { const int kk = k + i * L; v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(kk < split ? kk : k) * stride); }
          __builtin_amdgcn_sched_barrier(0);            // loads first: left alone, the scheduler re-uses ONE register quad and issues load, wait, add eight times
#pragma unroll
          for (int i = 0; i < 7; ++i) { if (k + i * L < split) sa[i] += v[i]; }
        }
        acc[0] = ((sa[0] + sa[1]) + (sa[2] + sa[3])) + ((sa[4] + sa[5]) + (sa[6] + sa[7]));
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) red[t][threadIdx.x] = acc[t];
  // pairwise tree over the split lanes (lane j += lane j + h for h = L/2 .. 1): every thread of the upper levels takes part, where a
  // serial walk by the L == 64 blocks' four lane-0 threads (576 dependent LDS reads) cost more than the slab reads themselves
#pragma unroll
  for (int h = L / 2; h >= 1; h >>= 1) {
    __syncthreads();
    if (lane < h) {
#pragma unroll
      for (int t = 0; t < NT; ++t) { acc[t] += red[t][threadIdx.x + h * UB]; if (h > 1) red[t][threadIdx.x] = acc[t]; }
    }
  }
  if (lane != 0 || !live) return;
  f32x4 v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) v[t] = acc[t] * inv_scale;
  if (ALLT) {
    float* d = dw + ((size_t)co * Cin + ci) * TAPS;
    if (vec_ok) {                                  // Cin % 4 == 0 and dw 16-byte aligned: TAPS float4 read-modify-writes of 4 * TAPS consecutive floats
#pragma unroll
      for (int m = 0; m < TAPS; ++m) {
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!overwrite) o = *reinterpret_cast<f32x4*>(d + 4 * m);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int f = 4 * m + j; o[j] += v[f % TAPS][f / TAPS]; }
        *reinterpret_cast<f32x4*>(d + 4 * m) = o;
      }
    } else {
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4)
        if (ci + e4 < Cin) {
#pragma unroll
          for (int t = 0; t < TAPS; ++t) d[(size_t)e4 * TAPS + t] = (overwrite ? 0.f : d[(size_t)e4 * TAPS + t]) + v[t][e4];
        }
    }
  } else {
    float* d = dw + ((size_t)co * Cin + ci) * TAPS + t0;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4)
      if (ci + e4 < Cin) d[(size_t)e4 * TAPS] = (overwrite ? 0.f : d[(size_t)e4 * TAPS]) + v[0][e4];
  }
}

template <int TAPS, int L, bool ALLT>
static void launch_reduce2(const WgradReduce& r, hipStream_t s) {
  const long per_tap = (long)r.cout_pad * r.cin_pad;
  const long nunits = (ALLT ? per_tap : (long)TAPS * per_tap) / 4;
  constexpr int UB = 256 / L;
  const unsigned nwb = (unsigned)((nunits + UB - 1) / UB);
  const unsigned nbias_blocks = r.db0 ? (unsigned)cdiv(r.Cout, 16) : 0u;
  const int vec_ok = (r.Cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(r.dw) & 15) == 0);
  hipLaunchKernelGGL((wgrad_reduce2_kernel<TAPS, L, ALLT>), dim3(nwb + nbias_blocks), dim3(256), 0, s, r.slab, r.split, r.cout_pad, r.cin_pad,
                     r.Cout, r.Cin, r.dw, r.inv_scale, r.db0, r.db1, r.inv_dev, nwb, vec_ok, r.overwrite);
}
template <int TAPS, bool ALLT>
static void launch_reduce2_l(const WgradReduce& r, int L, hipStream_t s) {
  if (L == 4) launch_reduce2<TAPS, 4, ALLT>(r, s);
  else launch_reduce2<TAPS, 16, ALLT>(r, s);
}

hipError_t launch_wgrad16_reduce(const WgradReduce& r, hipStream_t s) {
  if (r.taps != 9 && r.taps != 1) return hipErrorInvalidValue;
  // Split lanes per unit: 4 or 16 (a lane row of a block then reads >= 256 contiguous bytes of one slab; 64 lanes x 64-byte rows measured
  // 36 us where this form takes ~10), 16 only when the split is that deep and 4 lanes would leave fewer than 1024 blocks.  All taps per
  // thread only where that still gives >= 1024 blocks (weights of 256 x 256 x 9 and up - the layers whose dw does not stay in L2).
  const long units_all = (long)r.cout_pad * r.cin_pad / 4;
  const int lmax = r.split >= 16 ? 16 : 4;
  auto nblocks = [](long units, int L) { return (units + 256 / L - 1) / (256 / L); };
  const bool allt = r.taps == 1 || nblocks(units_all, lmax) >= 1024;
  const long units = allt ? units_all : units_all * r.taps;
  const int L = (nblocks(units, 4) < 1024 && lmax == 16) ? 16 : 4;
  if (r.taps == 1) launch_reduce2_l<1, true>(r, L, s);
  else if (allt) launch_reduce2_l<9, true>(r, L, s);
  else launch_reduce2_l<9, false>(r, L, s);
  return hipGetLastError();
}

template <typename T, int KS>
static hipError_t launch_wg16_ks(const WgradArgs& a, hipStream_t s, WgradReduce* red) {
  static const bool wg_narrow = getenv("PU_WG_NARROW") != nullptr;   // diagnostic: 32-cin tiles, 4 waves, two blocks per CU for every layer
  const bool wide = a.Cin > 32 && !wg_narrow;
  static const bool wg_big = getenv("PU_WG_BIG") != nullptr, wg_4w = getenv("PU_WG_4W") != nullptr;     // diagnostic switches, read once
  if (a.W % 32 == 0 && a.H % 8 == 0 && wide && wg_big) return launch_wg16<T, KS, 8, 32, 64>(a, s, red);   // 256-pixel K tiles: +4 % alone,
                                                                                            // but 152 KB LDS blocks co-residency with conv3
  static const bool no_bco32 = getenv("PU_WG_NO_BCO32") != nullptr;
  if (a.W % 32 == 0 && a.H % 4 == 0 && a.Cout <= 32 && !no_bco32)      // 256 x 256 level: no zero-padded cout half; (cin sub-tile) x tap groups
    return wide ? launch_wg16<T, KS, 4, 32, 64, 4, 32>(a, s, red) : launch_wg16<T, KS, 4, 32, 32, 4, 32>(a, s, red);
  if (a.W % 32 == 0 && a.H % 4 == 0) return wide ? (wg_4w ? launch_wg16<T, KS, 4, 32, 64>(a, s, red) : launch_wg16<T, KS, 4, 32, 64, 8>(a, s, red)) : launch_wg16<T, KS, 4, 32, 32>(a, s, red);
  // 16 x 16 level: 8 waves on 64-cin tiles.  32-cin tiles in 4-wave blocks (56 KB of LDS, two blocks per CU; PU_WG_NARROW=1) are 10-13 %
  // faster in the micro-benchmark (64.6 / 119 us against 74.6 / 133 us at 512 -> 512 / 1024 -> 512) but not in the step, where the
  // kernel shares the CUs with the main stream: 898.2 against 900.8 pairs/s, same box, alternating
  static const bool wg16_4w = getenv("PU_WG16_4W") != nullptr;
  if (a.W % 16 == 0 && a.H % 8 == 0) return wide ? (wg16_4w ? launch_wg16<T, KS, 8, 16, 64>(a, s, red) : launch_wg16<T, KS, 8, 16, 64, 8>(a, s, red)) : launch_wg16<T, KS, 8, 16, 32>(a, s, red);
  if (a.W % 8 == 0 && a.H % 8 == 0) return wide ? launch_wg16<T, KS, 8, 8, 64>(a, s, red) : launch_wg16<T, KS, 8, 8, 32>(a, s, red);
  return hipErrorInvalidValue;
}

// the split-K main kernel only; the caller runs launch_wgrad16_reduce(*red) afterwards (any stream ordered after `s`)
template <typename T>
hipError_t launch_wgrad16_main(const WgradArgs& a, hipStream_t s, WgradReduce* red) {
  if (!a.slab || !red) return hipErrorInvalidValue;
  if (a.taps == 9) return launch_wg16_ks<T, 3>(a, s, red);
  if (a.taps == 1) return launch_wg16_ks<T, 1>(a, s, red);
  return hipErrorInvalidValue;
}
template <typename T>
hipError_t launch_wgrad16(const WgradArgs& a, hipStream_t s) {
  WgradReduce r;
  hipError_t e = launch_wgrad16_main<T>(a, s, &r);
  return e != hipSuccess ? e : launch_wgrad16_reduce(r, s);
}
template hipError_t launch_wgrad16<f16>(const WgradArgs&, hipStream_t);
template hipError_t launch_wgrad16<bf16>(const WgradArgs&, hipStream_t);
template hipError_t launch_wgrad16_main<f16>(const WgradArgs&, hipStream_t, WgradReduce*);
template hipError_t launch_wgrad16_main<bf16>(const WgradArgs&, hipStream_t, WgradReduce*);

}  // namespace pu
