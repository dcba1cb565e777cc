// lqer_clock_probe: the shader clock the chip holds WHILE other work runs.  A few one-wave workgroups, launched on a stream of
// their own beside the kernels under study, stamp s_memtime (shader cycles) and s_memrealtime (100 MHz ticks), sleep until
// `duration_us` of real time has passed (stamping once more after the first quarter: the interval that counts), and stamp again: clock = d(cycles) / d(ticks) x 100 MHz (MI355X_MICROARCH.md, DVFS
// give-back, item 6).  bench.py quotes the median as roofline.sustained_mhz: a kernel whose main loop runs at 97 % of MFMA issue
// still shows ~0.5 of the peak that is priced at 2.4 GHz when the chip holds 1.6 GHz under that load - this makes the factor
// checkable from the bench line.  The wait is bounded twice (real time and an iteration cap): every wave exits.
#include "common.h"

namespace lqer {

__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* __restrict__ out, unsigned long long ticks) {
  unsigned long long c0, r0, c1, r1, cq = 0, rq = 0;
  bool quarter = false;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  r1 = r0, c1 = c0;
  for (int it = 0; it < (1 << 22); ++it) {  // (<= ~4 M sleeps of 64 x 64 cycles: minutes - the real-time bound ends it first)
    __builtin_amdgcn_s_sleep(64);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if (!quarter && r1 - r0 >= ticks / 4) quarter = true, cq = c1, rq = r1;  // (the first quarter is not counted: the load behind the probe ramps up)
    if (r1 - r0 >= ticks) break;
  }
  if (!quarter) cq = c0, rq = r0;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - cq;
    out[2 * blockIdx.x + 1] = r1 - rq;
  }
}

}  // namespace lqer

extern "C" int lqer_clock_probe(unsigned long long* out_pairs, int nblocks, int64_t duration_us, void* stream) {
  using namespace lqer;
  if (!out_pairs || nblocks < 1 || nblocks > 64 || duration_us < 1 || duration_us > 5000000) {
    set_error("clock_probe: bad argument (1..64 blocks, 1 us .. 5 s)");
    return LQER_E_INVALID;
  }
  k_clock_probe<<<(unsigned)nblocks, 64, 0, (hipStream_t)stream>>>(out_pairs, (unsigned long long)duration_us * 100ull);
  return check_launch("clock_probe");
}
