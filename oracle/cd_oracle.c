/*
 * TEST INFRASTRUCTURE ONLY.  CPU oracle for the consistent_depth fine-tuning hot path
 * (geometric-consistency loss + its depth gradient, bilinear `sample`, Adam).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (consistent_depth_amd/) never imports it and has no CPU
 * fallback: it fails loudly when the HIP library is missing.
 *
 * Parity pinning: the reference has no golden vectors or tests of its own
 * (SURVEY.md section 4).  This restatement is pinned against OUTPUTS OF THE REFERENCE
 * ITSELF: oracle/gen_golden.py imports /root/reference's loss.consistency_loss /
 * utils.geometry (unmodified) in the build container, runs them in fp64 and fp32 with
 * torch autograd on seeded inputs and commits the vectors to tests/golden/;
 * tests/test_oracle.py checks this file against those vectors.
 *
 * The body is compiled twice (float, double) from cd_oracle_body.inc.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define FN(x) x##_f32
#define FLOOR floorf
#define SQRT sqrtf
#define FABS fabsf
#include "cd_oracle_body.inc"
#undef REAL
#undef FN
#undef FLOOR
#undef SQRT
#undef FABS

#define REAL double
#define FN(x) x##_f64
#define FLOOR floor
#define SQRT sqrt
#define FABS fabs
#include "cd_oracle_body.inc"
