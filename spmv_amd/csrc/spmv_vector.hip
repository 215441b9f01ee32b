// spmv_vector.hip -- second translation unit of the HIP shim: the executors of the CSR-vector family (shim/launch_vector.hpp).
// Nothing else lives here; spmv_shim.hip declares launch_vector_any / launch_rows_any and links against these instantiations.
#include <hip/hip_runtime.h>

#include <atomic>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "spmv_shim.h"
#include "kernels/common.hpp"
#include "kernels/csr_vector4.hpp"
#include "kernels/xwindows.hpp"
#include "kernels/csr5.hpp"
#include "kernels/blocked.hpp"
#include "kernels/split.hpp"
#include "kernels/rcm.hpp"
#include "kernels/csr_vector_tile.hpp"

using namespace spmv;

#define SPMV_TU_SECONDARY // state.hpp: types and helpers only, the extern "C" entry points belong to spmv_shim.hip
#include "shim/state.hpp"

// the form selectors of launch.hpp (kept in step by the static_assert-free rule "one definition": both units include this list)
#include "shim/vector_forms.hpp"
#include "shim/launch_vector.hpp"
