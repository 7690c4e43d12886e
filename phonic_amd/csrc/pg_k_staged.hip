// pg_stage_fused_kernel: the staged single launch of [Gain|Panning]* -> Reverb units (the headline's dominant kernel) (one kernel per translation unit; the launchers are in pg_kernels.hip).
#include "pg_stage_body.inl"

__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage_fused_kernel(PgLaunch L) { stage_fused_body<1, 2>(L); }
