/* TEST INFRASTRUCTURE: Matlab's matrix.h, as far as the gateways need it (everything lives in the mock mex.h). */
#include "mex.h"
