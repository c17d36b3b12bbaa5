// tuning.h -- the one gate in front of every measurement knob of the library and the CLI.
//
// The kernels, the batch pipeline, the index upload and the communicator carry a few dozen knobs that exist for A/B runs
// (blocks per CU, unroll factors, queue grouping, pruning margin, ramp of the streamed sub-batches, ...; DESIGN.md section 4
// "Measurement knobs").  None changes a result, all of them change the performance profile -- and a library that a host
// process loads must not pick such a thing up from a stray variable in the user's environment.  So the environment is
// consulted only when TAXOR_TUNING=1 is set as well; without it tune_env() answers "not set" for every name and the
// defaults (the measured optimum) apply.  What a caller may legitimately want to choose per searcher is a field of
// taxor_gpu_search_params (flags), not a variable.  GPU_MAX_HW_QUEUES is the HIP runtime's own variable, not one of these.
#pragma once
#include <cstdlib>

namespace taxor {

inline bool tuning_enabled()
{
    const char *e = getenv("TAXOR_TUNING");
    return e && atoi(e) != 0;
}

inline const char *tune_env(const char *name) { return tuning_enabled() ? getenv(name) : nullptr; }

} // namespace taxor
