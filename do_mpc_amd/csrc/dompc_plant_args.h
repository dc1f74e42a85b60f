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
  int32_t max_steps;
  int32_t method;                            // 0: explicit pair, samples that turn out stiff repeat their interval with the implicit method;
                                             // 1: explicit Dormand-Prince 5(4) only; 2: implicit SDIRK 4(3) only
  double t_step, rtol, atol;
  double* z_guess;                           // [B][nz] or null: Newton start of the algebraic states per sample (in), their values at x_next (out)
  int32_t explicit_limit, pad;               // method 0: steps of the explicit pair after which a sample counts as stiff
};
}  // namespace dompc_plantk
