// The parallel pass for batches too small to fill the device, as a compilation unit of its own (see the end of the
// batch kernels in infl_kernels.hip for why).  Diagnostic cycle accounting is not compiled into this unit.
#undef DEFL_PAR_PROFILE
#undef DEFL_PAR_PROFILE_SETUP
#undef DEFL_PAR_PROFILE_STEPS
#define DEFL_SMALL_BATCH_UNIT 1
#include "infl_kernels.hip"
