"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/lmpc_hip.h declares."""
import os
import re

from tests import common


def test_library_builds_and_exports_all_declared_symbols(built):
    from racinglmpc_amd import _capi
    lib = _capi.load()
    header = open(os.path.join(common.ROOT, "include", "lmpc_hip.h")).read()
    declared = set(re.findall(r"\b(lmpc_[a-z_0-9]+)\s*\(", header))
    declared -= {"lmpc_ctx"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export %s" % name
    assert set(_capi.EXPORTS) == declared
    assert lib.lmpc_version() >= 100


def test_config_struct_layout_matches_header(built):
    """sizeof(lmpc_config) as seen by ctypes equals the C side (checked through lmpc_config_default round trip)."""
    from racinglmpc_amd import _capi
    cfg = _capi.default_config()
    assert cfg.N == 12 and cfg.numSS_points == 48 and cfg.maxNumPoint == 7 and cfg.max_iter == 40
    assert abs(cfg.tol_gap - 1e-11) < 1e-20 and abs(cfg.reg_lambda - 1e-6) < 1e-15
    assert list(cfg.bu) == [0.5, 0.5, 10.0, 10.0] and cfg.QtermSlack[35] == 500.0


def test_no_cpu_fallback_in_product():
    """The product package never imports the oracle."""
    pkg = os.path.join(common.ROOT, "racinglmpc_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no CPU oracle", ""), f


def test_variant_libraries_build_and_export_their_entry_point(built):
    """build() leaves one shared object per extra (N, numSS_points) variant next to the library, each exporting lmpc_variant_get."""
    import ctypes as C
    from racinglmpc_amd import build
    for n, s in build.EXTRA_VARIANTS:
        path = build.variant_path(n, s)
        assert os.path.exists(path), path
        lib = C.CDLL(path)
        assert hasattr(lib, "lmpc_variant_get")
    assert not (set(build.EXTRA_VARIANTS) & build.BUILTIN)
