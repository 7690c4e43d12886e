// pg_stage_fused_wide_kernel (one kernel per translation unit; the launchers are in pg_kernels.hip).
#include "pg_stage_body.inl"

// The same single launch for reverb units whose leading effects go beyond Gain / Panning (Filter, Eq5, Delay, Distortion:
// C5's per-voice Filter -> Eq5 -> Delay -> Reverb). A kernel of its own so that the lean one keeps its register allocation.
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage_fused_wide_kernel(PgLaunch L) { stage_fused_body<2, 3>(L); }
