// The step-size link of the fused transition (zhusuan_amd/csrc/fused_args.h:
// link_step_size, link_retire, tuner_persist -- the code the HIP kernels run)
// compiled for the host, one call standing for "the kernel's prologue" and
// one for "the last workgroup's epilogue".  TEST INFRASTRUCTURE: it lets the
// CPU tests of zhusuan_amd/hmc.py's orchestration (tests/fake_zshmc.py) apply
// pending / retired dual-averaging updates with the product's own update
// code instead of a restatement of it.
//   clang++ -O2 -ffp-contract=off -DZS_HOST_ONLY -shared -fPIC \
//       tests/host_link/zs_link_host.cpp -o tests/_build/libzs_link_host.so
#include <string.h>

#include "../../zhusuan_amd/csrc/fused_args.h"

using namespace zshmc;

static AdaptLink to_device_view(const zshmc_adapt_link* link,
                                unsigned long long* accum) {
  AdaptLink k;
  memset(&k, 0, sizeof(k));
  k.state = link->state;
  k.stats = link->stats;
  k.accum = link->stats ? accum : nullptr;
  // the launcher's rule (zshmc_hmc_diag_normal_step): n * 2^shift < 2^kSumBits
  int shift = 40;
  while (shift > 0 && (double)(link->n_chains_global + 1) *
                              (double)(1ull << shift) >=
                          (double)(1ull << kSumBits))
    --shift;
  k.fx_scale = (double)(1ull << shift);
  k.fx_inv_scale = 1.0 / k.fx_scale;
  k.inv_chains =
      link->n_chains_global > 0 ? 1.0 / (double)link->n_chains_global : 0.0;
  k.pending = link->pending;
  k.retire = link->retire_update;
  k.fresh = link->fresh_start ? 1.0f : 0.0f;
  k.used_step_size = link->used_step_size;
  k.tuner = TunerCfg{link->delta, link->gamma, link->t0, link->kappa, link->mu};
  return k;
}

extern "C" {

// the step size a launch with this link integrates with (kernel prologue)
float zs_host_link_step_size(const zshmc_adapt_link* link,
                             float step_size_host) {
  unsigned long long accum = 0;
  const AdaptLink k = to_device_view(link, &accum);
  return link_step_size(k, step_size_host);
}

// the epilogue of a launch whose chains' acceptance rates sum to `total`:
// link_retire as ONE workgroup (gridDim.x = 1, so it is the one that retires
// last): fixed-point sum, pending update from the OLD stats[0], this
// transition's own update, publication of the sum and the flag
void zs_host_link_retire(const zshmc_adapt_link* link, double total,
                         unsigned int flags_value) {
  unsigned long long accum = 0;
  const AdaptLink k = to_device_view(link, &accum);
  uint32_t flags = flags_value;
  link_retire(k, total, &flags);
}

// zshmc_stepsize_flush
void zs_host_link_flush(const zshmc_adapt_link* link) {
  unsigned long long accum = 0;
  const AdaptLink k = to_device_view(link, &accum);
  if (k.state && k.pending != ZSHMC_PEND_NONE)
    tuner_persist(k, k.pending, k.stats[0]);
}

}  // extern "C"
