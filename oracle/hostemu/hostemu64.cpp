// TEST INFRASTRUCTURE ONLY -- the stepper's per-lane code (phc_amd/csrc/phc_aba.h, the very functions k_sim_step is made of) compiled at DOUBLE precision.
//
// Why: the dense fp64 oracle (oracle/dyn_oracle.py) states the FRESH scheme in a formulation that shares nothing with the kernel's recursion, and the fp32 kernel is
// compared with it.  The LAGGED scheme (phc_sim_params_t.inertia_lag: the sub-steps behind the first one of a simulate() call keep its articulated inertias I^A and
// joint-space inverses D^-1 and redo only the bias-force recursion with the CURRENT levers, rotations and velocity products) has no dense counterpart: its result depends
// on the elimination order (stale I^A of a parent contains the child's I^a shifted by the OLD lever, the force hand-over uses the NEW one), so "H(q_old)^-1 b(q_new)" is a
// different scheme that differs from it at the order of the lag error itself.  What pins the lagged kernel instead:
//   1. this build == the dense oracle on the fresh scheme to ~1e-9 (tests/test_dynamics.py): the double-precision recursion IS the scheme, exactly;
//   2. the fp32 kernel / fp32 host emulation == this build with inertia_lag = 1 at the tolerances the fresh scheme is held to against the dense oracle;
//   3. both schemes converge to the continuous model at first order (tests/test_stepper_options.py).
// How: every `float` of the headers becomes `double` and the single-precision libm calls their double versions; the system headers are included first (their
// include guards keep them out of the macro's reach).  The C structs of include/phc_amd.h change layout with it: oracle/hostemu_util.py mirrors them (`*64`).
#include <math.h>
#include <cmath>
#include <stdint.h>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <omp.h>
#define float double
#define sqrtf sqrt
#define fminf fmin
#define fmaxf fmax
#define expf exp
#define sinf sin
#define cosf cos
#define atan2f atan2
#define acosf acos
#define fabsf fabs
#define rintf rint
#define log1pf log1p
#define logf log
#define floorf floor
#include "hostemu.cpp"
