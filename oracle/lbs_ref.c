/* TEST INFRASTRUCTURE ONLY -- see lbs_ref_impl.inc for what this restates.
 * Built by oracle/Makefile into oracle/liboracle.so; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it. */
#include <math.h>
#include <stdlib.h>
#include <stddef.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FN(name) CAT(name, _f32)
#include "lbs_ref_impl.inc"
#undef REAL
#undef FN

#define REAL double
#define FN(name) CAT(name, _f64)
#include "lbs_ref_impl.inc"
#undef REAL
#undef FN
