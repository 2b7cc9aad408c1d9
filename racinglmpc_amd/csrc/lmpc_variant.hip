// racinglmpc_amd/csrc/lmpc_variant.hip -- translation unit of ONE (N, numSS_points) variant of the solve kernels, built on demand as
// liblmpc_var_N<N>_S<S>.so (see lmpc_variant.hip.h).  hipcc ... -DLMPC_VAR_N=16 -DLMPC_VAR_S=36 -DLMPC_VARIANT_TU
#include "lmpc_variant.hip.h"

extern "C" int lmpc_variant_get(lmpc_variant_api *v, int abi_version) {
    if (!v || abi_version != LMPC_VARIANT_ABI) return -1;
    return lmpc_variant_fill<LMPC_VAR_N, LMPC_VAR_S>(v) ? 0 : -2;
}
