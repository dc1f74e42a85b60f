// dompc_plant_args.h - kernel argument block of the batched plant integrator, shared by the generic host runtime
// (dompc_runtime.cpp) and the per-model device code (dompc_plant.hip).  Plain data, no model-dependent sizes.
#pragma once
#include <stdint.h>

namespace dompc_plantk {
struct Args {
  const double *x, *u, *tvp, *p, *w, *v;     // [B][nx], then per-sample or shared (stride 0) rows of u, tvp, p, w, v
  double *x_next, *y;                        // [B][nx], [B][ny] (y may be null)
  int32_t* status;                           // [B] (may be null): bit 0 = step limit reached / NaN; steps taken in status >> 8
  int32_t batch, stride_u, stride_tvp, stride_p, stride_w, stride_v;
  int32_t max_steps, pad;
  double t_step, rtol, atol;
};
}  // namespace dompc_plantk
