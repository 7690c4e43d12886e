// pg_unit_kernel_fast_wide: the fast unit kernel with every effect kind (one kernel per translation unit; the launchers are in pg_kernels.hip).
#include "pg_unit_body.inl"

__global__ void __launch_bounds__(256, PG_FAST_WAVES) pg_unit_kernel_fast_wide(PgLaunch L) {
  PgUnitCarry carry;
  carry.resident = 0; carry.fx_valid = 0;
  pg_unit_body<true, PG_KMASK_ALL>(L, (int)blockIdx.x, 0, carry);
  for (int c = 1; c < L.n_chunks; ++c) { __syncthreads(); pg_unit_body<true, PG_KMASK_ALL>(L, (int)blockIdx.x, c, carry); }
}
