// placeholder: masked, batched small-d engine (ExperimentImpute) -- implemented next
#include "../../include/psmf_hip.h"
extern "C" int psmf_impute_run(const psmf_impute_config*, const double*, const uint8_t*, const uint8_t*, double*, double*,
                               const double*, const double*, const double*, double, double*, double*, double*, double*,
                               double*, double*, float*) {
  return PSMF_ERR_ARG;
}
