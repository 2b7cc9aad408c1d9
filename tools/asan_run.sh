#!/bin/bash
# Memory-safety pass: the GPU tests that reallocate stores under live contexts, dlopen variant objects and run device-resident
# rollout sessions, against liblmpc_hip_asan.so -- host side of the library under AddressSanitizer, a 256-byte guard zone behind every device buffer and
# between the work-buffer ranges of the slabs (-DLMPC_GUARD, checked when a buffer is freed).
#   in the build container:  python -c 'from racinglmpc_amd import build; build.build_asan()'        (the .so travels with the gpurun snapshot)
#   on the GPU box:          tools/asan_run.sh > gpurun_out/r4_asan.log 2>&1
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAVOUR=${1:-asan}                      # asan: host side under AddressSanitizer + guard zones;  guard: guard zones only (liblmpc_hip_guard.so, build.build_guard())
LIB=$ROOT/racinglmpc_amd/liblmpc_hip_$FLAVOUR.so
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
[ -f "$LIB" ] || { echo "$LIB not built"; exit 2; }
cd "$ROOT"
export LMPC_LIB=$LIB LMPC_GUARD_REPORT=1
# detect_leaks=0: the interpreter itself leaks by design; protect_shadow_gap=0: the ROCm runtime maps device apertures into ASan's shadow gap
# HSA_XNACK=1: the ROCm build of the ASan runtime intercepts hsa_amd_memory_pool_allocate and needs page-fault retry on the device for its own mappings
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=1
PRE=""; if [ "$FLAVOUR" = "asan" ]; then PRE=$RT; export HSA_XNACK=1; fi
LD_PRELOAD=$PRE timeout 1500 python -m pytest tests/test_gpu_stores.py tests/test_gpu_retry.py tests/test_gpu_dropin_main.py tests/test_gpu_parity.py tests/test_gpu_configs.py \
    -m gpu -q --timeout=1200 -k "stores or retry or dropin or rollout or generations or status or 30_lap or horizons or edge" -p no:cacheprovider
echo "pytest exit code $?"
