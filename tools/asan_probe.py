"""Developer probe: create a context and run one step under the ASan flavour, stderr not captured (tools/asan_run.sh sets the environment)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
print("probe: importing", flush=True)
from racinglmpc_amd import _capi
print("probe: library", _capi.LIB_PATH, flush=True)
lib = _capi.load()
print("probe: loaded", flush=True)
from tests import common
g = common.load_lmpc_golden()
ctx, par = common.make_lmpc_ctx(g, 4, max_batch=4)
print("probe: context made", flush=True)
res = common.run_golden_step_check(max_records=4)
print("probe: step ok", {k: v for k, v in res.items() if k != "status"}, "guard failures", lib.lmpc_debug_guard_failures(), flush=True)
ctx.close()
print("probe: done, guard failures", lib.lmpc_debug_guard_failures(), flush=True)
