"""CPU suite: libilsx.so loads and exports every symbol include/ilsx.h declares; the ctypes table covers
the header; the product path fails loudly without a GPU (no fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ilsx.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ilsx_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from ilswiss_amd import _lib
    return _lib


def test_header_declares_symbols():
    syms = declared_symbols()
    assert "ilsx_sac_train_step" in syms and "ilsx_replay_sample" in syms and len(syms) >= 40


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/ilsx.h but not exported: {missing}"


def test_ctypes_table_matches_header(built):
    assert sorted(built.PROTOTYPES) == declared_symbols()
    lib = built.load()
    assert lib.ilsx_abi_version() == 1


def test_struct_sizes_match_header(built, tmp_path):
    """Every struct that crosses the ABI: sizeof() as gcc lays include/ilsx.h out == the ctypes mirror in _lib.py."""
    import subprocess
    pairs = dict(ilsx_mlp_cfg="MlpCfg", ilsx_sac_cfg="SacCfg", ilsx_sac_stats="SacStats", ilsx_disc_cfg="DiscCfg",
                 ilsx_disc_stats="DiscStats", ilsx_ppo_cfg="PpoCfg", ilsx_td3_cfg="Td3Cfg", ilsx_td3_stats="Td3Stats",
                 ilsx_sacv_cfg="SacvCfg", ilsx_sacv_stats="SacvStats", ilsx_bc_cfg="BcCfg", ilsx_planar_model="PlanarModel", ilsx_spatial_model="SpatialModel",
                 ilsx_opt_meta="OptMeta")
    src = tmp_path / "sizes.c"
    body = "".join(f'  printf("{c} %zu\\n", sizeof({c}));\n' for c in pairs)
    src.write_text('#include <stdio.h>\n#include "ilsx.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes = dict(zip(out[::2], map(int, out[1::2])))
    for c, py in pairs.items():
        assert ctypes.sizeof(getattr(built, py)) == sizes[c], (c, ctypes.sizeof(getattr(built, py)), sizes[c])


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ilswiss_amd
    with pytest.raises(RuntimeError, match="libilsx error"):
        ilswiss_amd.Context(0)


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "ilswiss_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"


def test_weight_gradient_row_ranges_cover_the_batch(built):
    """Host arithmetic of the large-batch weight-gradient launches (ilsx_debug_dw_split): whatever the batch size, the row ranges are whole
    numbers of the kernel's row steps, none is empty, and together they cover the batch exactly once.  (An empty trailing range — e.g.
    10241 rows in 20 ranges of 544 — once left a stale slab in the sum.)"""
    lib = built.load()
    for big, unit, cap in ((1, 32, 64), (0, 128, 32)):
        for rows in list(range(1024, 12000, 37)) + [10241, 16500, 32768, 32769, 65536, 100003, 1 << 20]:
            s, rps = ctypes.c_int(), ctypes.c_int()
            assert lib.ilsx_debug_dw_split(rows, big, ctypes.byref(s), ctypes.byref(rps)) == 0
            s, rps = s.value, rps.value
            assert 1 <= s <= cap and rps % unit == 0
            assert (s - 1) * rps < rows <= s * rps, (rows, big, s, rps)     # the last range starts inside the batch, the ranges reach its end
    assert lib.ilsx_debug_dw_split(0, 1, ctypes.byref(ctypes.c_int()), ctypes.byref(ctypes.c_int())) != 0
