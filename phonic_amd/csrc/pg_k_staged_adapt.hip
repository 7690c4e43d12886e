// pg_stage_fused_adapt_kernel (one kernel per translation unit; the launchers are in pg_kernels.hip).
#include "pg_stage_body.inl"

// The wide single launch with the source adapters compiled into its source stage: reverb-terminated units whose voice sits behind a
// ResampledSource (src/source/resampled.rs:27-152: a second cubic resampler and the two 512-frame TempBuffers) or is fed by the host
// (pg_graph_add_stream_voice). Until round 5 such units took the fused fast kernel at two workgroups per CU (1024 of them: 0.31 ms per block;
// here 0.17). A kernel of its own: inside pg_stage_fused_wide_kernel the adapter code cost C5 a second spilled register (-0.6 %).
__global__ void __launch_bounds__(256, PG_STAGE_WAVES) pg_stage_fused_adapt_kernel(PgLaunch L) { stage_fused_body<3, 4>(L); }
