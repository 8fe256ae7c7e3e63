// Fused HMC transition for a diagonal-Normal log-joint on gfx950 (MI355X):
// the LDS-DMA ring variant for one-wave-per-chain rows (n_data > 128, rows
// 16-B aligned).  Same arithmetic, same Philox counters and the same per-lane
// summation order as hmc_fused_normal.hip (results are bit-identical); what
// changes is how rows travel:
//
//   * each wave owns a ring of K row slots in LDS and keeps K rows of q in
//     flight with `global_load_lds_dwordx4` (HBM -> LDS without touching
//     VGPRs).  With one register-prefetched row per wave the kernel was
//     latency-bound (3 waves/SIMD x 4 KiB = 48 KiB/CU in flight only part of
//     the time); the ring holds K x 4 KiB per wave in flight all the time and
//     gives the 4*NCH prefetch registers back, which buys a fourth wave/SIMD.
//   * ONE workgroup per CU (all the waves the register budget allows) owns
//     every nblk-th run of 16 chains and its waves draw the next chain from
//     an LDS ticket counter.  With a static split the oldest wave of each SIMD
//     won the VALU arbitration and finished at ~55 % of the kernel, leaving
//     the SIMD to one latency-bound wave at the end (waves ended between 55 %
//     and 100 % of the span); tickets make all waves of a CU finish within
//     one chain of each other.
//   * rejected chains cost no write: the row store is predicated by EXEC
//     inside the asm statement (never branched around), so every trip issues
//     exactly NCH DMA loads + NCH stores and the `s_waitcnt vmcnt(N)` that
//     guards a slot can be a compile-time count (gfx9 returns VMEM in order;
//     EXEC=0 VMEM still passes through the counter -- probed on hardware,
//     tools/archive/glds_probe.hip).
//
// vmcnt ledger (per wave; D(i) = NCH DMA loads of the row of trip i, S(i) =
// NCH row stores + 5 HMCInfo scalar stores, all EXEC-predicated asm):
//   prologue D0..D(K-1) | trip i: wait D(i); ds_read slot; D(i+K); compute;
//   S(i).  Issued after D(i) when trip i waits:
//   S(i-K) D(i+1) S(i-K+1) ... D(i+K-1) S(i-1) = (K-1)*NCH + K*(NCH+5), i >= K
//   and at least (K-1)*NCH for i < K (the conservative count used there).
// The kernel body must compile without scratch spills and without any
// compiler-issued VMEM inside the trip loop (tests/test_build_resources.py).
//
// Reference semantics: zhusuan/hmc.py:21-61, :348-372, :479-498;
// distributions/univariate.py:174-181; distributions/base.py:302-304.
#include <stdlib.h>

#include "common.h"
#include "fused_args.h"
#include "philox.h"

namespace zshmc {

// cache-policy suffixes of the row traffic (rows are read once and written
// once per launch, so neither side wants to stay in L2)
// ZS_LD_POL / ZS_ST_POL: 0 = default, 1 = nt, 2 = sc0 sc1, 3 = sc1
// Round 3, with the seven-round generator (the launch sits closer to its
// memory bound than in round 2, where no policy moved it by more than 1 %):
// row STORES non-temporal: 0.0910 -> 0.0877..0.0891 ms without a mass vector,
// 0.0908 -> 0.0886 with one, 0.0931 -> 0.0901 with a mean tile; `sc0 sc1`
// stores within 0.5 % of that; nt LOADS alone 0.0883, but nt loads AND nt
// stores together 0.1033 (profiles/archive/r03n_cache_policy_kbench.txt).
#ifndef ZS_LD_POL
#define ZS_LD_POL 0
#endif
#ifndef ZS_ST_POL
#define ZS_ST_POL 1
#endif
#define ZS_POL_STR_0 ""
#define ZS_POL_STR_1 " nt"
#define ZS_POL_STR_2 " sc0 sc1"
#define ZS_POL_STR_3 " sc1"
#define ZS_POL_CAT(x) ZS_POL_STR_##x
#define ZS_POL(x) ZS_POL_CAT(x)
#define ZS_LD_POLICY ZS_POL(ZS_LD_POL)
#define ZS_ST_POLICY ZS_POL(ZS_ST_POL)
#ifndef ZS_RING_GRANULE
#define ZS_RING_GRANULE 16  // consecutive chains per workgroup turn (16 = one
#endif                      // 64-B line of each HMCInfo array)
#ifndef ZS_RING_K4
#define ZS_RING_K4 1  // ring depth at NCH = 4 (n_data 772..1024): one row ahead is
                      // enough to hide HBM latency behind a ~6 us trip, and a
                      // shallower ring shortens the read -> write-back distance
                      // of a row (0.0951 -> 0.0925 ms)
#endif
#ifndef ZS_RNG_PAIR
#define ZS_RNG_PAIR 1  // generator runs two chunks interleaved (A/B knob)
#endif
#ifndef ZS_RING_WAVES
#define ZS_RING_WAVES 4  // waves per SIMD at NCH = 4 without mass
#endif

// ---- hand-counted VMEM primitives -----------------------------------------
// global -> LDS, 16 B per lane: LDS[m0 + OFF + lane*16] <- sbase[voff + OFF].
// The 5 wait states in front cover "VALU wrote an SGPR that VMEM reads".
// EXEC masks travel as two 32-bit SGPR halves (readfirstlane results), which
// is the only form hipcc reliably keeps in scalar registers.
struct Mask {
  uint32_t lo, hi;
};

template <int OFF>
__device__ __forceinline__ void dma16(uint32_t voff, const float* sbase,
                                      uint32_t lds_addr, Mask mask) {
  uint32_t keep;
  uint64_t saved;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b64 %1, exec\n\t"
      "s_and_b32 exec_lo, exec_lo, %5\n\t"
      "s_and_b32 exec_hi, exec_hi, %6\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %3 offset:%7" ZS_LD_POLICY "\n\t"
      "s_mov_b64 exec, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(saved)
      : "v"(voff), "s"(sbase), "s"(lds_addr), "s"(mask.lo), "s"(mask.hi),
        "n"(OFF)
      : "memory", "scc");
}

template <int OFF>
__device__ __forceinline__ void store16(uint32_t voff, f4 data, float* sbase,
                                        Mask mask) {
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_and_b32 exec_lo, exec_lo, %4\n\t"
      "s_and_b32 exec_hi, exec_hi, %5\n\t"
      "s_nop 1\n\t"
      "global_store_dwordx4 %1, %2, %3 offset:%6" ZS_ST_POLICY "\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved)
      : "v"(voff), "v"(data), "s"(sbase), "s"(mask.lo), "s"(mask.hi), "n"(OFF)
      : "memory", "scc");
}

// the five per-chain HMCInfo scalars: lane 0 only, array k written iff its
// pointer is non-null and commit01 == 1; always 5 VMEM instructions, so they
// sit in the ledger like S(i).  (The enables are derived from the kernel
// arguments inside the statement: hipcc moves pre-computed 0/1 flags to VGPRs.)
__device__ __forceinline__ void store_info5(uint32_t vc, float v0, float v1,
                                            float v2, float v3, float v4,
                                            float* p0, float* p1, float* p2,
                                            float* p3, float* p4,
                                            uint32_t commit01) {
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b32 exec_hi, 0\n\t"
      "s_cmp_lg_u64 %7, 0\n\t"
      "s_cselect_b32 exec_lo, %12, 0\n\t"
      "s_nop 1\n\t"
      "global_store_dword %1, %2, %7\n\t"
      "s_cmp_lg_u64 %8, 0\n\t"
      "s_cselect_b32 exec_lo, %12, 0\n\t"
      "s_nop 1\n\t"
      "global_store_dword %1, %3, %8\n\t"
      "s_cmp_lg_u64 %9, 0\n\t"
      "s_cselect_b32 exec_lo, %12, 0\n\t"
      "s_nop 1\n\t"
      "global_store_dword %1, %4, %9\n\t"
      "s_cmp_lg_u64 %10, 0\n\t"
      "s_cselect_b32 exec_lo, %12, 0\n\t"
      "s_nop 1\n\t"
      "global_store_dword %1, %5, %10\n\t"
      "s_cmp_lg_u64 %11, 0\n\t"
      "s_cselect_b32 exec_lo, %12, 0\n\t"
      "s_nop 1\n\t"
      "global_store_dword %1, %6, %11\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved)
      : "v"(vc), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "s"(p0), "s"(p1),
        "s"(p2), "s"(p3), "s"(p4), "s"(commit01)
      : "memory", "scc");
}
constexpr int kInfoStores = 5;

// next ticket of the workgroup: LDS atomic issued by lane 0 only, NOT waited
// for -- the caller's next `s_waitcnt lgkmcnt(0)` (which must name the result
// as an in/out operand) lands it; read it with uni32 afterwards.
__device__ __forceinline__ uint32_t ticket_issue(uint32_t lds_addr) {
  uint32_t t;
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "v_mov_b32 %0, 1\n\t"
      "s_nop 0\n\t"
      "ds_add_rtn_u32 %0, %2, %0\n\t"
      "s_mov_b64 exec, %1"
      : "=&v"(t), "=&s"(saved)
      : "v"(lds_addr)
      : "memory");
  return t;
}

// pin a wave-uniform value into SGPRs (the "s" asm constraint does not insert
// the readfirstlane itself)
__device__ __forceinline__ uint32_t uni32(uint32_t x) {
  return __builtin_amdgcn_readfirstlane(x);
}
__device__ __forceinline__ int64_t uni64(int64_t x) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)x >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
// all-ones / all-zeros EXEC mask from a wave-uniform condition
// mask ? a : b / mask ? v : 0 as v_cndmask_b32_e64 with the lane mask in an
// SGPR pair that is not VCC.  hipcc's hazard recognizer does not look inside
// inline asm: the wait state a VALU read of a transcendental result needs on
// gfx940+ (v_exp_f32 feeds keep_if) is spelled out in the string.
__device__ __forceinline__ float select_e64(uint64_t mask, float a, float b) {
  float r;
  asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3"
      : "=v"(r)
      : "v"(b), "v"(a), "s"(mask));
  return r;
}
__device__ __forceinline__ float keep_if(float v, uint64_t mask) {
  float r;
  asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask));
  return r;
}

// A/B probes only: VMEM issued with EXEC = 0 (ZS_NO_MEM: compute-only time;
// ZS_NO_LD / ZS_NO_ST: one direction of the row traffic)
template <bool LOAD>
__device__ __forceinline__ Mask mask_if(bool c) {
  const uint32_t f = uni32(c ? 0xFFFFFFFFu : 0u);
  return Mask{f, f};
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// chunk k of a row: imm offsets reach 4095 B, so chunks 4..7 go through a
// second base 4096 B further on (global and LDS side alike)
template <int NCH, int k = 0>
__device__ __forceinline__ void issue_row(uint32_t voff, uint32_t voff_last,
                                          const float* srow, uint32_t slot_addr,
                                          Mask on) {
  if constexpr (k < NCH - 1) {
    dma16<(k & 3) * 1024>(voff, srow + (k >> 2) * 1024,
                          slot_addr + (k >> 2) * 4096, on);
    issue_row<NCH, k + 1>(voff, voff_last, srow, slot_addr, on);
  } else {
    // last chunk: lanes beyond n_data read the row's first 16 B (in bounds)
    dma16<0>(voff_last, srow, slot_addr + (NCH - 1) * 1024, on);
  }
}

template <int NCH, int k = 0>
__device__ __forceinline__ void store_row(uint32_t voff, uint32_t voff_last,
                                          const f4* out, float* srow, Mask on,
                                          Mask on_last) {
  if constexpr (k < NCH - 1) {
    store16<(k & 3) * 1024>(voff, out[k], srow + (k >> 2) * 1024, on);
    store_row<NCH, k + 1>(voff, voff_last, out, srow, on, on_last);
  } else {
    store16<0>(voff_last, out[k], srow, on_last);
  }
}

// Waves per SIMD = the register budget that compiles WITHOUT scratch spills
// (a spill is a VMEM instruction the vmcnt ledger does not know about;
// tests/test_build_resources.py asserts "VGPRs Spill: 0" per instantiation).
// COLSTATS instantiations (column sums of the end state, see the kernel): a
// chain's 4*NCH (q' - m) and (q' - m)^2 per lane go straight into the
// workgroup's double tile, one ds_add_f64 per element -- no extra registers,
// the plain instantiation's waves per SIMD; the sums differ from run to run
// only by the order of the chains, at the 1e-16 level.  Register accumulators
// (double: two waves per SIMD; float: not bit-stable under dynamic tickets,
// spills under static ones) were measured and lost: docs/LABNOTES.md 3.2, 10.
// timing builds (tools/build_ring_variants.sh, docs/LABNOTES.md 12) of the
// COLSTATS instantiations, with WRONG column sums (the launch reports mean 0
// and variance 1 so that the chains keep moving): bit 0 no accumulation at
// all, bit 1 the arithmetic without the LDS atomics
#ifndef ZS_CS_SKIP
#define ZS_CS_SKIP 0
#endif
constexpr int ring_waves_for(int nch, bool has_mass, bool colstats = false) {
  const int w = nch <= 3 ? 4
                         : (nch == 4 ? (has_mass ? 3 : ZS_RING_WAVES)
                                     : (nch == 5 && !has_mass ? 3 : 2));
  (void)colstats;  // (the column sums cost no registers: same occupancy)
  return w;
}

// STAGE: the workgroup's per-chain scalars (MH uniform in, five HMCInfo values
// out) live in LDS under the chain's ticket (info_cap of them fit); otherwise
// the uniform is drawn on the scalar unit and the five values are stored from
// the trip loop (hand-counted, in the ledger).
// ZERO_MEAN: a.mean == NULL (every mean is 0, the usual prior): no mean tile
// in LDS, no subtraction after the slot read, no re-addition before the store.
// COLSTATS: the launch also leaves, per workgroup, the column sums over its
// chains of (q' - m) and (q' - m)^2 of the state the transition ENDS in (q' =
// the proposal if accepted, else the start row) around the EWMV mean m --
// what the NEXT iteration's mass update consumes (hmc.py:130-148 uses the
// state an iteration starts from, :288), so mass adaptation costs no read
// pass of its own.  The start row of a rejected chain is still needed at the
// end of its trip: the ring gets one more physical slot per wave (the row of
// trip i stays in slot i % (K+1) until trip i+1 refills it) and the sums are
// accumulated with ds_add_f64 into one [2][row] double tile per workgroup.
template <int NCH, int K, bool HAS_MASS, bool STAGE, bool ZERO_MEAN,
          bool COLSTATS>
__global__ __launch_bounds__(256 * ring_waves_for(NCH, HAS_MASS, COLSTATS)) void
hmc_diag_normal_ring_kernel(FusedArgs a) {
  constexpr int kLedgerInfo = STAGE ? 0 : kInfoStores;
  constexpr int kRow = NCH * 256;  // padded row length (floats) of one chain
  constexpr int kRowB = kRow * 4;
  constexpr int kWavesPerBlock =
      4 * ring_waves_for(NCH, HAS_MASS, COLSTATS);  // a CU
  constexpr int kSlots = COLSTATS ? K + 1 : K;  // physical slots per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kTiles =
      (ZERO_MEAN ? 0 : 1) + (HAS_MASS ? 1 : 0) + (COLSTATS ? 1 : 0);
  float* __restrict__ s_mean = reinterpret_cast<float*>(smem);  // !ZERO_MEAN
  float* __restrict__ s_sqrtm = s_mean + (ZERO_MEAN ? 0 : kRow);  // HAS_MASS
  float* __restrict__ s_cm = s_sqrtm + (HAS_MASS ? kRow : 0);     // COLSTATS
  float* __restrict__ s_ring = s_mean + kTiles * kRow;
  // COLSTATS: [2][NCH*4][64] doubles, element (k, j) of lane l at
  // (k*4 + j)*64 + l (lanes contiguous: conflict-free ds_add_f64)
  double* __restrict__ s_cs =
      reinterpret_cast<double*>(s_ring + kWavesPerBlock * kSlots * kRow);
  double* __restrict__ s_acc = s_cs + (COLSTATS ? 2 * kRow : 0);
  int* __restrict__ s_bad = reinterpret_cast<int*>(s_acc + kWavesPerBlock);
  int* __restrict__ s_ticket = s_bad + 1;
  // [2] = "the update's acceptance-independent half is in s_prep"
  TunerPrep* __restrict__ s_prep = reinterpret_cast<TunerPrep*>(s_bad + 4);
  float* __restrict__ s_info = reinterpret_cast<float*>(s_bad + 12);  // [5][cap]

#ifdef ZS_TIMING
  const unsigned long long t_start = wall_clock64();
  const unsigned long long c_start = clock64();
#endif
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
#ifdef ZS_TIMING
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + wib;
#endif
  const int64_t D = a.n_data;
  const int64_t C = a.n_chains;
  // the iteration word of the Philox counters: the argument, or -- a launch
  // replayed from a hipGraph -- the argument plus a device counter; pinned
  // into an SGPR here, i.e. its load is waited for before any DMA is issued
  const uint32_t iteration = uni32(link_iteration(a.link, a.iteration));

  // ---- this workgroup's chains; tickets index into them -------------------
  const int64_t nblk = gridDim.x, blk = blockIdx.x;
  // Workgroups take turns of G consecutive chains: ticket t of workgroup b is
  // chain ((t / G) * nblk + b) * G + t % G.  At any moment the CUs sweep one
  // narrow band of q together (the access pattern of a grid-stride copy; a
  // contiguous range per CU measured 4-8 % slower and less even across CUs),
  // and G = 16 keeps whole 64-B lines of the HMCInfo arrays in one workgroup.
  constexpr int G = ZS_RING_GRANULE;
  const int64_t round = (int64_t)G * nblk;
  const int64_t full = C / round, tail = C % round - blk * G;
  const int count = (int)uni32(
      (uint32_t)(full * G + (tail < 0 ? 0 : (tail > G ? G : tail))));
  const int64_t start = blk * G;
#define ZS_CHAIN_OF(t) \
  (start + (int64_t)((t) / G) * round + (int64_t)((t) % G))
  const int64_t last_row = C - 1;
  // MH uniforms (hmc.py:485) of all this workgroup's chains, one Philox call
  // per lane, parked in LDS under the chain's ticket
  float* __restrict__ s_u = s_info + 5 * a.info_cap;
  if (STAGE)
    for (int t = threadIdx.x; t < count; t += blockDim.x)
      s_u[t] = uniform_chain((uint32_t)(ZS_CHAIN_OF(t) + a.chain_offset),
                             iteration, a.k0, a.k1);

  // ---- stage mean / sqrt(mass) in LDS (zero padding beyond n_data) --------
  if (!ZERO_MEAN || HAS_MASS || COLSTATS)
    for (int d = threadIdx.x; d < kRow; d += blockDim.x) {
      if (!ZERO_MEAN) s_mean[d] = d < D ? a.mean[d] : 0.f;
      if (HAS_MASS) s_sqrtm[d] = d < D ? sqrtf(a.mass[d]) : 0.f;
      if (COLSTATS) {
        s_cm[d] = d < D ? a.link.cs_mean[d] : 0.f;
        s_cs[d] = 0.0;
        s_cs[kRow + d] = 0.0;
      }
    }
  if (threadIdx.x == 0) {
    *s_bad = 0;
    *s_ticket = 0;
  }

  // the step size of THIS transition: the pending dual-averaging update of
  // the previous one is applied here (hmc.py:501-505 moved across the launch
  // boundary), so an adaptive transition is still one launch
  const float eps = link_step_size(a.link, a.step_size_host);
  const bool moving = eps != 0.f;
  const float se = moving ? eps : 1.f;
  const float inv_se = 1.0f / se;

  // ---- per-latent parameters kept in registers ---------------------------
  f4 nep[NCH];  // -se * exp(-2*logstd)   (precision: univariate.py:178)
  f4 eim[NCH];  //  se / mass             (only if HAS_MASS)
  float logz_part = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int64_t d0 = (int64_t)(k * kWave + lane) * 4;
    const bool valid = d0 < D;  // n_data % 4 == 0: a chunk is all in or out
    f4 ls = f4{0.f, 0.f, 0.f, 0.f}, m = f4{1.f, 1.f, 1.f, 1.f};
    if (valid) {
      ls = *reinterpret_cast<const f4*>(a.logstd + d0);
      if (HAS_MASS) m = *reinterpret_cast<const f4*>(a.mass + d0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // padding gets precision 0 and inverse mass 0: no contribution anywhere
      nep[k][j] = valid ? -se * expf(-2.0f * ls[j]) : 0.f;
      logz_part += valid ? (kHalfLog2PiNeg - ls[j]) : 0.f;
      if (HAS_MASS) eim[k][j] = valid ? se / m[j] : 0.f;
    }
  }
  const float logz = wave_total_dpp(logz_part);
  // the drift's scalar factor: from an SGPR (v_pk_fma_f32 with a scalar
  // operand) or pinned in a VGPR (A/B knob, tools/kbench.py)
  const float eps_l = eps;
  const int Lr = moving ? a.n_leapfrogs : 0;
  const float hk = moving ? 0.5f : 0.f;    // first half kick
  const float hk2 = Lr >= 1 ? 0.5f : 0.f;  // taken back from the last
  // every compiler-issued load above has been consumed: from here on the
  // loop's VMEM traffic is the hand-counted asm only
  __syncthreads();  // LDS tile ready
  // the acceptance-independent half of the step-size update this launch
  // retires with: one lane of the last wave, off everybody else's path (its
  // wave joins the ticket queue a microsecond late); read after the
  // barriers at the end of the kernel
  if (wib == kWavesPerBlock - 1 && lane == 0)
    s_bad[2] = link_prepare(a.link, s_prep) ? 1 : 0;

  auto draw = [&]() -> int {  // next ticket of this workgroup (wave-uniform)
    int t = 0;
    if (lane == 0) t = atomicAdd(s_ticket, 1);
    return (int)uni32((uint32_t)t);
  };

  const uint32_t ticket_addr =
      (uint32_t)reinterpret_cast<uintptr_t>(s_ticket);
  float* __restrict__ ring_w = s_ring + wib * kSlots * kRow;
  const uint32_t ring_addr = __builtin_amdgcn_readfirstlane(
      (uint32_t)reinterpret_cast<uintptr_t>(ring_w));
  const uint32_t voff = (uint32_t)lane * 16u;
  const bool valid_last = (int64_t)((NCH - 1) * kWave + lane) * 4 < D;
  const uint32_t voff_last =
      valid_last ? (uint32_t)((NCH - 1) * kWave + lane) * 16u : 0u;
  const uint64_t b_last = __ballot(valid_last);
  const Mask m_last{uni32((uint32_t)b_last), uni32((uint32_t)(b_last >> 32))};

  // ---- prologue: draw K tickets, fill the ring ---------------------------
  int tk[K];  // tk[0] = the chain this trip works on, tk[j] = j trips ahead
#pragma unroll
  for (int j = 0; j < K; ++j) {
    tk[j] = draw();
    int64_t row = ZS_CHAIN_OF(tk[j]);
    row = row < last_row ? row : last_row;
    issue_row<NCH>(voff, voff_last, a.q + uni64(row * D),
                   ring_addr + j * kRowB, mask_if<true>(tk[j] < count));
  }

  double acc_local = 0.0;
  bool bad_old = false;
#ifdef ZS_TIMING
  unsigned long long n_done = 0;
#endif
  int slot = 0;
  for (int it = 0; tk[0] < count; ++it) {
    const int t_cur = tk[0];
    const int64_t chain = ZS_CHAIN_OF(t_cur);
    float* __restrict__ qrow = a.q + uni64(chain * D);
    const uint32_t gchain = (uint32_t)(chain + a.chain_offset);

    // the ticket for the slot this trip frees (LDS atomic: its latency hides
    // behind the wait and the slot read)
    uint32_t nt_raw = ticket_issue(ticket_addr);

    // ---- wait for D(it), r = q - mean, refill the slot with D(it+K) -------
    if (it < K)
      wait_vmcnt<(K - 1) * NCH>();
    else
      wait_vmcnt<(K - 1) * NCH + K * (NCH + kLedgerInfo)>();
    f4 r[NCH], p[NCH];
    // refill of the slot this trip frees: issued right after the slot is in
    // registers (later in the trip -- a shorter distance between a row's read
    // and its write-back -- measured no better); always before this trip's
    // stores, so the vmcnt ledger does not change
    const float* late_src;
    uint32_t late_dst;
    bool late_on;
    // COLSTATS: the row read now stays in its slot to the end of the trip;
    // the refill goes to the slot the PREVIOUS trip read
    const int slot_rd = slot;
    {
      const float* __restrict__ sl = ring_w + slot * kRow;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        r[k] = *reinterpret_cast<const f4*>(sl + (k * kWave + lane) * 4);
        if (!ZERO_MEAN)
          r[k] -= *reinterpret_cast<const f4*>(s_mean + (k * kWave + lane) * 4);
      }
      // the slot must be in registers before the DMA may overwrite it (the
      // same wait lands the ticket)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nt_raw)::"memory");
      const int nt = (int)uni32(nt_raw);
      int64_t nrow = ZS_CHAIN_OF(nt);
      nrow = nrow < last_row ? nrow : last_row;
      late_src = a.q + uni64(nrow * D);
      late_dst = uni32(ring_addr +
                       (uint32_t)(COLSTATS ? (slot == 0 ? K : slot - 1) : slot) *
                           kRowB);
      late_on = nt < count;
      issue_row<NCH>(voff, voff_last, late_src, late_dst,
                     mask_if<true>(late_on));
      slot = slot + 1 == kSlots ? 0 : slot + 1;
#pragma unroll
      for (int j = 0; j + 1 < K; ++j) tk[j] = tk[j + 1];
      tk[K - 1] = nt;
    }

    // ---- momentum resample (hmc.py:21-23, :458), initial energies and the
    // first half kick (trip i = 0 of hmc.py:352-364), chunk by chunk.
    // grad log p = -prec * r; with nep = -eps*prec a kick of s2 is
    // p += (s2/eps) * (nep * r), a drift is r += eps * p / m.
    uint32_t key0 = a.k0, key1 = a.k1;
    asm volatile("" : "+s"(key0), "+s"(key1));
    f4 ko = f4{0.f, 0.f, 0.f, 0.f}, uo = ko;
    // one chunk's share of: p0 = z * sqrt(mass), K0, U0, first half kick
    auto start_chunk = [&](int k, float z0, float z1, float z2, float z3) {
      p[k] = f4{z0, z1, z2, z3};
      if (HAS_MASS) {
        p[k] = p[k] *
               *reinterpret_cast<const f4*>(s_sqrtm + (k * kWave + lane) * 4);
        ko += (p[k] * p[k]) * eim[k];
      } else {
        if (k == NCH - 1 && !valid_last) p[k] = f4{0.f, 0.f, 0.f, 0.f};
        ko += p[k] * p[k];
      }
      const f4 t = nep[k] * r[k];
      uo += t * r[k];
      p[k] += hk * t;
    };
    // the generator runs two chunks at a time where the register budget has
    // room (philox.h: normal4x2), one otherwise
    constexpr int kPair = (ZS_RNG_PAIR && NCH <= 4) ? 2 : 1;
#pragma unroll
    for (int k = 0; k < NCH; k += kPair) {
      uint32_t group = (uint32_t)(k * kWave + lane);
      // hipcc hoists the first Philox round's M0 * group out of the trip loop
      // (2 VGPRs per chunk held for the whole kernel); where the budget has
      // no room for that the counter word is made opaque per trip
      if (NCH >= 7) asm volatile("" : "+v"(group));
      if (kPair == 2 && k + 1 < NCH) {
        float za[4], zb[4];
        normal4x2(group, group + kWave, gchain, iteration, kStreamMomentum,
                  key0, key1, za, zb);
        start_chunk(k, za[0], za[1], za[2], za[3]);
        start_chunk(k + 1, zb[0], zb[1], zb[2], zb[3]);
      } else {
        float z0, z1, z2, z3;
        normal4(group, gchain, iteration, kStreamMomentum, key0, key1, z0,
                z1, z2, z3);
        start_chunk(k, z0, z1, z2, z3);
      }
    }

    // ---- leapfrog (hmc.py:348-372): L full drifts + full kicks; half of the
    // last kick is taken back below ----------------------------------------
    // (bottom-tested: a top-tested loop makes hipcc copy r and p into fresh
    // registers on the zero-trip edge, 12-16 v_mov per chain)
    if (Lr > 0) {
      int i = Lr;
      do {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          if (HAS_MASS)
            r[k] += eim[k] * p[k];
          else
            r[k] += eps_l * p[k];
          p[k] += nep[k] * r[k];
        }
      } while (--i > 0);
    }

    // ---- Hamiltonians (hmc.py:30-35) and acceptance (hmc.py:46-61) -------
    f4 kn = f4{0.f, 0.f, 0.f, 0.f}, un = kn;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const f4 t = nep[k] * r[k];
      un += t * r[k];
      p[k] -= hk2 * t;
      if (HAS_MASS)
        kn += (p[k] * p[k]) * eim[k];
      else
        kn += p[k] * p[k];
    }
    float k_old = (ko[0] + ko[1]) + (ko[2] + ko[3]);
    float u_old = (uo[0] + uo[1]) + (uo[2] + uo[3]);
    float k_new = (kn[0] + kn[1]) + (kn[2] + kn[3]);
    float u_new = (un[0] + un[1]) + (un[2] + un[3]);
    wave_total4_swap(k_old, u_old, k_new, u_new);
    if (HAS_MASS) {  // sum p^2/m = (1/se) sum p^2 * (se/m)
      k_old *= inv_se;
      k_new *= inv_se;
    }
    const float lp_old = logz + 0.5f * inv_se * u_old;
    const float lp_new = logz + 0.5f * inv_se * u_new;
    const float h_old = -lp_old + 0.5f * k_old;
    const float h_new = -lp_new + 0.5f * k_new;
    const float dh = h_old - h_new;
    // exp(min(dh, 0)) on v_exp_f32 (argument <= 0: no overflow, underflow
    // flushes to 0), then the guards of hmc.py:56-59.  fminf drops a NaN
    // operand, so NaN is tested explicitly; exp of a non-positive finite
    // number is finite.  The select is forced into its e64 form with the lane
    // mask in a plain SGPR pair: the e32 form hipcc picks (mask in VCC) issues
    // ~5x slower than any other VALU op on gfx950 (tools/instr_bench.hip).
    const float acc = keep_if(
        __builtin_amdgcn_exp2f(fminf(dh, 0.0f) * 1.4426950408889634f),
        __builtin_amdgcn_ballot_w64((dh == dh) && isfinite(lp_new)));
    if (!isfinite(lp_old)) bad_old = true;

    float u;
    if (STAGE)
      u = s_u[t_cur];
    else
      u = uniform_chain(gchain, iteration, key0, key1);
    const bool accept = u < acc;  // strict, hmc.py:486

    // STAGE: the sum is taken from the staged values after the loop, in
    // ticket order (which wave ran which chain varies from run to run)
    if (!STAGE && lane == 0) acc_local += (double)acc;
#ifdef ZS_TIMING
    ++n_done;
#endif

    // ---- MH select: the accepted row goes back in place (hmc.py:487-497);
    // EXEC-predicated, so a rejected chain issues the same NCH (empty) stores
    {
      const Mask m_acc = mask_if<false>(accept && a.commit != 0);
      // q' = r + mean in place (r is dead after the store): all the LDS reads
      // of the mean tile go out together, ahead of the asm statements
      if (!ZERO_MEAN) {
#pragma unroll
        for (int k = 0; k < NCH; ++k)
          r[k] += *reinterpret_cast<const f4*>(s_mean + (k * kWave + lane) * 4);
      }
      store_row<NCH>(voff, voff_last, r, qrow, m_acc,
                     Mask{m_acc.lo & m_last.lo, m_acc.hi & m_last.hi});
    }

    // ---- the five HMCInfo scalars of this chain (lane 0; hmc.py:508-517) --
    // Staged in LDS under the chain's ticket and written back as whole lines
    // after the loop (STAGE), or stored from here as 5 ledger entries.
    const float lp_sel =
        select_e64(__builtin_amdgcn_ballot_w64(accept), lp_new, lp_old);
    if (STAGE) {
      if (lane == 0) {
        const int cap = a.info_cap;
        s_info[t_cur] = acc;
        s_info[cap + t_cur] = h_old;
        s_info[2 * cap + t_cur] = h_new;
        s_info[3 * cap + t_cur] = lp_old;
        s_info[4 * cap + t_cur] = lp_sel;
      }
    } else {
      store_info5((uint32_t)chain * 4u, acc, h_old, h_new, lp_old, lp_sel,
                  a.acceptance_rate, a.orig_hamiltonian, a.hamiltonian,
                  a.orig_log_prob, a.log_prob, a.commit_direct);
    }

    // ---- COLSTATS: this chain's END state into the workgroup's column
    // sums.  r holds q' (mean re-added above); a rejected chain's state is
    // the start row, still in its ring slot.  LDS traffic only (no VMEM:
    // the ledger does not change); `accept` is wave-uniform.
    if (COLSTATS && !(ZS_CS_SKIP & 1)) {
      const bool took = accept && a.commit != 0;
      const float* __restrict__ sl = ring_w + slot_rd * kRow;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (k == NCH - 1 && !valid_last) continue;  // padding lanes
        f4 v = r[k];
        if (!took) v = *reinterpret_cast<const f4*>(sl + (k * kWave + lane) * 4);
        const f4 d = v - *reinterpret_cast<const f4*>(s_cm + (k * kWave + lane) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = (k * 4 + j) * kWave + lane;
          if (ZS_CS_SKIP & 2) {
            asm volatile("" ::"v"((double)d[j]), "v"((double)d[j] * (double)d[j]));
            continue;
          }
          __hip_atomic_fetch_add(&s_cs[idx], (double)d[j], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&s_cs[kRow + idx],
                                 (double)d[j] * (double)d[j], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  // no DMA may outlive the wave (its LDS would be handed to another block)
  wait_vmcnt<0>();
#ifdef ZS_TIMING
  if (a.timing && lane == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* t = a.timing + wave_id * 4;
    t[0] = t_start;
    t[1] = wall_clock64();
    t[2] = (xcc & 0xf) | ((clock64() - c_start) << 8);
    t[3] = (unsigned long long)n_done;
  }
#endif

  // ---- sum of acceptance rates of this workgroup, then the order-fixed
  // total over workgroups (link_retire) --------------------------------------
  if (bad_old) *s_bad = 1;
  __syncthreads();  // all waves done: s_info complete, s_bad final
  if (COLSTATS) {
    // this workgroup's row of the partials: [sum (q'-m) | sum (q'-m)^2]
    double* __restrict__ out = a.link.cs_parts + (int64_t)blockIdx.x * 2 * D;
    for (int d = threadIdx.x; d < (int)D; d += blockDim.x) {
      const int chunk = d >> 2, j = d & 3;
      const int idx = ((chunk / kWave) * 4 + j) * kWave + chunk % kWave;
      out[d] = s_cs[idx];
      out[D + d] = (ZS_CS_SKIP & 3) ? (double)count : s_cs[kRow + idx];
    }
  }
  if (STAGE) {
    acc_local = 0.0;
    for (int t = threadIdx.x; t < count; t += blockDim.x)
      acc_local += (double)s_info[t];
  }
  const double w = wave_sum_f64(acc_local);
  if (lane == 0) s_acc[wib] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < kWavesPerBlock; ++i) tot += s_acc[i];
    if (*s_bad && a.flags) atomicOr(a.flags, ZSHMC_FLAG_OLD_LOGPROB_NONFINITE);
    link_retire(a.link, tot, a.flags, s_bad[2] ? s_prep : nullptr);
  }
  // ---- staged HMCInfo scalars -> global, G consecutive chains per line ----
  if (STAGE && a.commit) {
    const int cap = a.info_cap;
    for (int t = threadIdx.x; t < count; t += blockDim.x) {
      const int64_t c = ZS_CHAIN_OF(t);
      if (a.acceptance_rate) a.acceptance_rate[c] = s_info[t];
      if (a.orig_hamiltonian) a.orig_hamiltonian[c] = s_info[cap + t];
      if (a.hamiltonian) a.hamiltonian[c] = s_info[2 * cap + t];
      if (a.orig_log_prob) a.orig_log_prob[c] = s_info[3 * cap + t];
      if (a.log_prob) a.log_prob[c] = s_info[4 * cap + t];
    }
  }
#undef ZS_CHAIN_OF
}

constexpr size_t kLdsLimit = 160 * 1024;

// ring depth of the COLSTATS instantiations (one more physical slot per wave
// has to fit in LDS next to the double tile): NCH <= 2 keep K = 3, the wider
// rows run one row ahead like NCH = 4 does anyway
constexpr int ring_cs_k(int nch) {
  return nch <= 2 ? 3 : 1;
}
constexpr int kRingCsMaxNch = 6;  // 7, 8: LDS / VGPR budget exhausted

constexpr size_t ring_lds_base(int nch, int k, bool has_mass, bool zero_mean,
                               bool colstats) {
  const int waves = 4 * ring_waves_for(nch, has_mass, colstats);
  return (size_t)((zero_mean ? 0 : 1) + (has_mass ? 1 : 0) +
                  (colstats ? 1 : 0) + waves * (colstats ? k + 1 : k)) *
             nch * 1024 +
         (colstats ? (size_t)nch * 4096 : 0) + waves * sizeof(double) + 48;
}

template <int NCH, int K, bool HAS_MASS, bool ZERO_MEAN, bool COLSTATS>
static int launch_ring_cfg(const FusedArgs& a_in, hipStream_t stream) {
  constexpr int kWaves = 4 * ring_waves_for(NCH, HAS_MASS, COLSTATS);
  constexpr size_t lds_base =
      ring_lds_base(NCH, K, HAS_MASS, ZERO_MEAN, COLSTATS);
  static_assert(lds_base <= kLdsLimit, "ring does not fit in LDS");
  static bool ready = false;
  if (!ready) {
    // the ring needs more than the default 64 KiB dynamic-LDS cap
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(
            hmc_diag_normal_ring_kernel<NCH, K, HAS_MASS, true, ZERO_MEAN,
                                        COLSTATS>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(
          reinterpret_cast<const void*>(
              hmc_diag_normal_ring_kernel<NCH, K, HAS_MASS, false, ZERO_MEAN,
                                          COLSTATS>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
    if (e != hipSuccess) return check_hip(e, "ring kernel: LDS size attribute");
    ready = true;
  }
  const int64_t grid = fused_ring_grid(a_in.n_chains);
  // largest per-workgroup share (workgroup 0); stage the per-chain scalars in
  // LDS when they fit beside the ring
  constexpr int G = ZS_RING_GRANULE;
  const int64_t round = G * grid;
  const int64_t tail = a_in.n_chains % round;
  const int64_t share = (a_in.n_chains / round) * G + (tail > G ? G : tail);
  FusedArgs a = a_in;
  // 6 floats per chain: 5 HMCInfo scalars + the MH uniform
  // (the in-loop path when the share does not fit: > ~300k chains at
  // D = 1024; tests/test_gpu_fused.py compares the two chain by chain)
  const bool stage = lds_base + (size_t)share * 24 <= kLdsLimit;
  a.info_cap = stage ? (int)share : 0;
  a.commit_direct = (a.commit && !stage) ? 1u : 0u;
  const size_t lds = lds_base + (stage ? (size_t)share * 24 : 0);
  const dim3 gdim(grid > 0 ? (unsigned)grid : 1u), bdim(64 * kWaves);
  if (stage)
    hipLaunchKernelGGL(
        (hmc_diag_normal_ring_kernel<NCH, K, HAS_MASS, true, ZERO_MEAN,
                                     COLSTATS>),
        gdim, bdim, lds, stream, a);
  else
    hipLaunchKernelGGL(
        (hmc_diag_normal_ring_kernel<NCH, K, HAS_MASS, false, ZERO_MEAN,
                                     COLSTATS>),
        gdim, bdim, lds, stream, a);
  ZS_LAUNCH_CHECK("hmc_diag_normal_ring_kernel launch");
  return ZSHMC_OK;
}

template <int NCH, int K>
static int launch_ring_k(const FusedArgs& a, hipStream_t stream) {
  if constexpr (NCH <= kRingCsMaxNch) {
    if (a.link.cs_parts) {
      constexpr int KC = ring_cs_k(NCH);
      if (a.mean)
        return a.mass ? launch_ring_cfg<NCH, KC, true, false, true>(a, stream)
                      : launch_ring_cfg<NCH, KC, false, false, true>(a, stream);
      return a.mass ? launch_ring_cfg<NCH, KC, true, true, true>(a, stream)
                    : launch_ring_cfg<NCH, KC, false, true, true>(a, stream);
    }
  }
  if (a.mean)
    return a.mass ? launch_ring_cfg<NCH, K, true, false, false>(a, stream)
                  : launch_ring_cfg<NCH, K, false, false, false>(a, stream);
  return a.mass ? launch_ring_cfg<NCH, K, true, true, false>(a, stream)
                : launch_ring_cfg<NCH, K, false, true, false>(a, stream);
}

// one workgroup per CU; fewer when there are not even that many turns
int64_t fused_ring_grid(int64_t n_chains) {
  constexpr int G = ZS_RING_GRANULE;
  const int64_t cus = device_cu_count();
  const int64_t turns = (n_chains + G - 1) / G;
  return turns < cus ? turns : cus;
}

bool fused_ring_colstats(int64_t D, bool has_mass, bool zero_mean) {
  int nch = 0, k = 0;
  return fused_ring_enabled() &&
         fused_ring_config(D, has_mass, zero_mean, &nch, &k) &&
         nch <= kRingCsMaxNch;
}

bool fused_ring_config(int64_t D, bool has_mass, bool zero_mean, int* nch_out,
                       int* k_out) {
  if (D % 4 != 0 || D <= 128 || D > 2048) return false;
  // NCH = ceil(D / 256) exactly: only the LAST 1 KiB chunk of a row may be
  // ragged (that is what voff_last / m_last handle)
  const int nch = (int)((D + 255) / 256);
  // 8 chunks + mass does not fit 256 VGPRs without scratch spills (VMEM the
  // ledger cannot count): that shape stays on the register-prefetch kernel
  if (nch == 8 && has_mass) return false;
  // likewise 7 chunks + mass + a mean tile (one spilled VGPR)
  if (nch == 7 && has_mass && !zero_mean) return false;
  *nch_out = nch;
  *k_out = nch <= 3 ? 3 : (nch == 4 ? ZS_RING_K4 : 2);
  return true;
}

bool fused_ring_enabled() { return true; }

int launch_fused_ring(const FusedArgs& a, hipStream_t stream) {
  int nch = 0, k = 0;
  const bool ok = fused_ring_config(a.n_data, a.mass != nullptr,
                                    a.mean == nullptr, &nch, &k) &&
                  ((reinterpret_cast<uintptr_t>(a.q) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(a.logstd) & 15) == 0) &&
                  (!a.mass || (reinterpret_cast<uintptr_t>(a.mass) & 15) == 0) &&
                  a.n_chains + 0 < (1ll << 30);  // 4*chain fits 32 bits
  if (!ok) return ZSHMC_ERR_UNSUPPORTED;
  // column statistics were asked for (the caller checked
  // zshmc_fused_colstats_rows): only the instantiations that produce them
  if (a.link.cs_parts && nch > kRingCsMaxNch) return ZSHMC_ERR_UNSUPPORTED;
  switch (nch) {
    case 1: return launch_ring_k<1, 3>(a, stream);
    case 2: return launch_ring_k<2, 3>(a, stream);
    case 3: return launch_ring_k<3, 3>(a, stream);
    case 4: return launch_ring_k<4, ZS_RING_K4>(a, stream);
    case 5: return launch_ring_k<5, 2>(a, stream);
    case 6: return launch_ring_k<6, 2>(a, stream);
    case 7:
      if (a.mass && a.mean) return ZSHMC_ERR_UNSUPPORTED;  // (not reached)
      if (a.mean)
        return launch_ring_cfg<7, 2, false, false, false>(a, stream);
      return a.mass ? launch_ring_cfg<7, 2, true, true, false>(a, stream)
                    : launch_ring_cfg<7, 2, false, true, false>(a, stream);
    default:
      return a.mean ? launch_ring_cfg<8, 2, false, false, false>(a, stream)
                    : launch_ring_cfg<8, 2, false, true, false>(a, stream);
  }
}

}  // namespace zshmc
