// terra_fz_api.hpp -- what terra_hip.hip calls in terra_fz.hip (the contraction-allowed build of the noise kernels): launchers with untyped struct arguments, internal to
// the library (hidden visibility).  Each returns the hipError_t of its launch.
#pragma once
#include <stdint.h>
#include <stddef.h>
extern "C" {
__attribute__((visibility("hidden"))) int terra_fz_noise_grid(int mode, void const *grid_job, void const *noise_consts, void const *sin_lut, float const *smx, float const *smy, float *out, uint32_t *mm, uint32_t const *nlut, void *stream);
__attribute__((visibility("hidden"))) int terra_fz_noise_tiles(void const *refs, uint32_t n, uint32_t nux, float const *d_sm, float const *d_m0, void const *grid_job, void const *noise_consts, void const *sin_lut, float *zvals, uint32_t tw, uint32_t const *nlut, void *stream);
}
