// Argument block shared by the fused diag-Normal HMC kernels
// (hmc_fused_normal.hip: register-prefetch kernel for small / ragged rows;
//  hmc_fused_ring.hip: LDS-DMA ring kernel for 16-B aligned rows > 512 B).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zshmc {

constexpr float kHalfLog2PiNeg = -0.91893853320467274178f;  // -0.5*log(2*pi)

struct FusedArgs {
  float* q;
  const float* mean;
  const float* logstd;
  const float* mass;
  const float* step_size_dev;
  float step_size_host;
  int64_t n_chains;
  int64_t n_data;
  int64_t chain_offset;
  int n_leapfrogs;
  uint32_t k0, k1;
  uint32_t iteration;
  int commit;
  float* acceptance_rate;
  float* orig_hamiltonian;
  float* hamiltonian;
  float* orig_log_prob;
  float* log_prob;
  double* acc_sum;
  uint32_t* flags;
  // ring kernel only (set by its launcher): HMCInfo scalars staged in LDS
  // (info_cap chains per workgroup) or stored straight from the trip loop
  int info_cap;          // 0 = no staging
  uint32_t commit_direct;  // commit && !staging: enables the in-loop stores
#ifdef ZS_TIMING
  unsigned long long* timing;  // [n_waves][4]: start, end, xcc, chains
#endif
};

typedef float f4 __attribute__((ext_vector_type(4)));

// hmc_fused_ring.hip.  Returns ZSHMC_ERR_UNSUPPORTED (without setting the
// error string) when the shape is not one it covers.
int launch_fused_ring(const FusedArgs& a, hipStream_t stream);
// "NCH,K,mass" of the ring instantiation for this shape; false if the shape
// is not covered (alignment aside).
bool fused_ring_config(int64_t n_data, bool has_mass, int* nch, int* k);
bool fused_ring_enabled();  // ZSHMC_FUSED_RING != 0

}  // namespace zshmc
