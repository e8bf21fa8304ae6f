// lgar_tangent_nl.hip -- forward-mode tangent kernels for ONE soil-layer count (compiled with -DLGAR_NL=<n>); see
// lgar_tangent_body.hpp.
#include <hip/hip_runtime.h>

#include "lgar_host.hpp"
#include "lgar_launch.hpp"
#include "lgar_tangent_body.hpp"

#ifndef LGAR_NL
#error "compile with -DLGAR_NL=<number of soil layers>"
#endif

namespace lgar {

__host__ __device__ inline unsigned columns_per_block(int share) { return share >= 2 ? (unsigned)((WAVE / share) * share) : (unsigned)WAVE; }

// Waves per SIMD the register allocator is held to.  The 8-front fp32 kernel needs 259 registers on its own -- three too many
// for two waves -- and a single resident wave gets an issue slot only every ~7 cycles: held to 256, it
// runs two waves per SIMD (backward pass of the 100 000-column ensemble 61.6 -> 44.7 ms).  The fp64 kernels stay at one wave:
// their dual-number front table alone is 40 KB of LDS per wave.
#ifndef LGAR_TAN_F32_WAVES
#define LGAR_TAN_F32_WAVES 2
#endif
template <typename R, int CAP> struct TangentOccupancy {
  static constexpr int waves = (sizeof(R) == 4 && CAP == LGAR_CAP_SMALL) ? LGAR_TAN_F32_WAVES : 1;
};
template <typename R, int NL, int CAP, int MODE>
__global__ __launch_bounds__(WAVE, (TangentOccupancy<R, CAP>::waves)) void lgar_tangent_kernel(TArgs<R> a) {
  __shared__ WaveLDS<Dual<R>, CAP, 1> lds;
  __shared__ R xchg[LGAR_XCHG_WORDS];  // tangent_share: the lanes of a column exchange trapezoid nodes through it (lgar_dual.hpp)
  const int lane = threadIdx.x;
  // the argument block is read in place (kernarg segment), see LGAR_KARG in lgar_device.hpp
  const LGAR_KARG TArgs<R> *ap = (const LGAR_KARG TArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr();
  if (ap->pending_in != nullptr && *ap->pending_in == 0u) return;  // no column was handed over to this kernel
  const size_t N = (size_t)ap->N;
  unsigned *ticket = ap->ticket;
  // A block = the columns of one wavefront: 64, or -- W lanes sharing a column (tangent_share) -- floor(64 / W) groups of W.
  // With a ticket counter: persistent waves, as in the forward kernels.
  const unsigned cpb = columns_per_block(ap->share);
  const unsigned nblocks = (unsigned)((N + cpb - 1) / cpb);
  for (bool first = true;; first = false) {
    unsigned blk = blockIdx.x;
    if (ticket != nullptr) {
      if (lane == 0) blk = atomicAdd(ticket, 1u);
      blk = __builtin_amdgcn_readfirstlane(blk);
      if (blk >= nblocks) break;
    } else if (!first) {
      break;
    }
    const size_t c = (size_t)blk * cpb + lane;
    if ((unsigned)lane < cpb && c < N) tangent_lane<R, NL, CAP, MODE>(ap, c, lane, lds, &xchg[0]);
  }
}

template <typename R, int NL, int CAP, int MODE>
static void launch_one(TArgs<R> &a, unsigned nblocks, unsigned *ticket, hipStream_t st) {
  a.ticket = ticket;
  unsigned grid = nblocks;
  if (ticket != nullptr) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    // resident one-wave workgroups per CU for THIS kernel (registers and LDS decide: 4 for the fp64 dual-number kernels --
    // 256 VGPRs + ~180 AGPRs, one wave per SIMD -- 8 for the fp32 ones)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lgar_tangent_kernel<R, NL, CAP, MODE>, WAVE, 0) != hipSuccess || per_cu <= 0)
      per_cu = 4;
    const unsigned slots = (unsigned)cus * (unsigned)per_cu;
    grid = nblocks < slots ? nblocks : slots;
  }
  hipLaunchKernelGGL((lgar_tangent_kernel<R, NL, CAP, MODE>), dim3(grid), dim3(WAVE), 0, st, a);
}

template <typename R, int NL>
static int tangent_typed(const LgarDims *dims, const LgarParams *params, const LgarParams *direction, const LgarForcing *forcing,
                         const void *w_runoff, const void *w_perc, void *grad_out, void *tangent_runoff, int32_t *status,
                         hipStream_t st, unsigned *tickets) {
  const unsigned cpb = columns_per_block(dims->tangent_share);
  const unsigned nblocks = ((unsigned)dims->n_columns + cpb - 1) / cpb;
  if (tickets != nullptr && hipMemsetAsync(tickets, 0, LGAR_NTICKETS * sizeof(unsigned), st) != hipSuccess) return LGAR_E_LAUNCH;
  TArgs<R> a{dims->n_columns, dims->n_steps, forcing_columns(dims), forcing_group(dims), dims->tangent_share, front_slots(dims),
             nullptr, nullptr, nullptr, 1, 1, (const R *)params->alpha, (const R *)params->n, (const R *)params->ksat,
             (const R *)params->theta_e, (const R *)params->theta_r, (const R *)params->thickness,
             (const R *)direction->alpha, (const R *)direction->n, (const R *)direction->ksat,
             (const R *)forcing->precip, (const R *)forcing->pet, (const R *)w_runoff, (const R *)w_perc,
             (R *)grad_out, (R *)tangent_runoff, status, make_glob<R>(dims)};
  if (dims->search_mode == 0) {
    launch_one<R, NL, LGAR_FMAX, 0>(a, nblocks, tickets, st);
    return launch_status();
  }
  const bool chain = (NL + dims->num_subcycles + 2 <= LGAR_CAP_SMALL) && (nblocks > 1024u || dims->search_mode == 2);
  if (chain) {
    a.chain_first = 1; a.chain_last = 0;
    a.pending_out = tickets ? tickets + 4 : nullptr;
    launch_one<R, NL, LGAR_CAP_SMALL, 1>(a, nblocks, tickets, st);
    int rc = launch_status();
    if (rc) return rc;
    a.chain_first = 0; a.chain_last = 1;
    a.pending_in = a.pending_out;
    a.pending_out = nullptr;
  }
  launch_one<R, NL, LGAR_FMAX, 1>(a, nblocks, tickets ? tickets + 1 : nullptr, st);
  return launch_status();
}

template <int NL>
int launch_tangent_nl(const LgarDims *dims, const LgarParams *params, const LgarParams *direction, const LgarForcing *forcing,
                      const void *w_runoff, const void *w_perc, void *grad_out, void *tangent_runoff, int32_t *status,
                      int dtype, hipStream_t st, unsigned *tickets) {
  if (dtype == LGAR_F64)
    return tangent_typed<double, NL>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st, tickets);
  if (dtype == LGAR_F32)
    return tangent_typed<float, NL>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st, tickets);
  return LGAR_E_ARG;
}

#if defined(LGAR_MEASURE) && defined(LGAR_CLOCKS)
// measurement builds: the cycle attribution of this translation unit's kernels (lgar_measure.hpp LGAR_POINT_CLK)
extern "C" int lgar_debug_clocks_tangent(unsigned long long *out, int reset) {
  unsigned long long z[64] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lgar_dbg_clk), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(lgar_dbg_clk), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

template int launch_tangent_nl<LGAR_NL>(const LgarDims *, const LgarParams *, const LgarParams *, const LgarForcing *,
                                        const void *, const void *, void *, void *, int32_t *, int, hipStream_t, unsigned *);

}  // namespace lgar
