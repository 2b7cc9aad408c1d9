"""python tools/exec_prologue_check.py <.so | .s> ...  -- see racinglmpc_amd/isa_check.py (the guard racinglmpc_amd.build applies to every library it produces)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runpy.run_module("racinglmpc_amd.isa_check", run_name="__main__")
