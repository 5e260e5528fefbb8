/*
 * rlshaders_amd_diag.h -- measurement aids of librlshaders_amd.so.  NOT part of the drop-in surface.
 *
 * Nothing here has a counterpart in shihchinw/rlShaders and nothing here is needed to replace its closure layer:
 * include/rlshaders_amd.h is the boundary a plugin binds.  These entry points exist for bench.py and for the profiling
 * scripts under tools/ (DESIGN.md sections 5 and 6); they are exported by the same library so that what is measured is
 * the library that ships.  A build with RLS_DIAGNOSTICS=0 (cmake -DRLS_DIAGNOSTICS=OFF; python -m rlshaders_amd.build
 * --variant nodiag -DRLS_DIAGNOSTICS=0) leaves all of it out: that library exports the drop-in surface only, holds no
 * diagnostic kernel, and its product kernels are the default build's up to address literals
 * (tests/test_diagnostics_option.py).
 */
#ifndef RLSHADERS_AMD_DIAG_H
#define RLSHADERS_AMD_DIAG_H

#include "rlshaders_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* In-kernel clock stamps.  Between _begin and _end the kernels of the four BASELINE configurations --
 * rls_ggx_reflect_refract with every parameter streamed, rls_sss_probe_ray with per-point scatter distances,
 * rls_skin_sample_eval_pdf with every parameter streamed, rls_disney_integrate at one lane per point -- are launched as
 * their DIAGNOSTIC instantiation: the same kernel body bracketed by reads of the shader-clock counter (s_memtime) and of
 * the constant 100 MHz counter (s_memrealtime) by the first wave of every workgroup.  The product kernels contain no
 * stamp.  Every stamped launch clears the slots first (on the launch stream), so _read copies out, per workgroup slot,
 * four words {memtime at entry, at exit, memrealtime at entry, at exit} of the LAST stamped launch, whatever grids the
 * earlier launches of the bracket used (stamps_host holds 4 * capacity words; *count = slots there are; a slot whose
 * workgroup did not run is all zero), and synchronises.  Effective shader clock of a workgroup's lifetime =
 * (w1 - w0) / (w3 - w2) x 100 MHz; bench.py reports the median as roofline.effective_clock_ghz.
 * Not for production use: a stamped launch takes ~2 % longer than the product's, and a context on which _begin is left
 * in force keeps launching the diagnostic instantiations until _end (or rls_context_destroy).  A bracket and a launch
 * graph recording exclude each other: _begin and _read fail while rls_graph_begin_capture is in force, and
 * rls_graph_begin_capture fails between _begin and _end (a recorded stamped launch would write stamps on every replay).
 * One context, one host thread. */
rls_status  rls_diag_clock_stamps_begin(rls_context *ctx);
rls_status  rls_diag_clock_stamps_read(rls_context *ctx, int64_t capacity, uint64_t *stamps_host, int64_t *count);
rls_status  rls_diag_clock_stamps_end(rls_context *ctx);

#ifdef __cplusplus
}
#endif

#endif /* RLSHADERS_AMD_DIAG_H */
