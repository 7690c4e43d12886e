// pg_stage1/2/3_kernel: one launch per stage (profiling the stages in isolation) (one kernel per translation unit; the launchers are in pg_kernels.hip).
#include "pg_stage_body.inl"

__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage1_kernel(PgLaunch L) {
  if ((int)blockIdx.x >= L.n_units || !stage_unit_staged<1>(L, blockIdx.x)) return;
  (void)stage1_run<1, false>(L, blockIdx.x, stage_slot_info(L, blockIdx.x));
}
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage2_kernel(PgLaunch L) {
  if ((int)blockIdx.x >= L.n_units || !stage_unit_staged<1>(L, blockIdx.x)) return;
  bool deferred; const int flags = stage_unit_flags(L, blockIdx.x, deferred);
  if (!deferred) stage2_run<1, false>(L, blockIdx.x, flags);
}
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage3_kernel(PgLaunch L) {
  if ((int)blockIdx.x >= L.n_units || !stage_unit_staged<1>(L, blockIdx.x)) return;
  bool deferred; const int flags = stage_unit_flags(L, blockIdx.x, deferred);
  if (!deferred) stage3_run<1, false>(L, blockIdx.x, flags, pg_smem, 0, make_int4(0, 0, 0, 0));
}
