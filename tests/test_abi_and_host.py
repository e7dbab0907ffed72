"""CPU: the C-ABI library loads here (no GPU) and exports every symbol that
include/svt_hip_dsp.h declares; the host layer fails LOUDLY without a device (no
CPU fallback); the package never touches oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "svt_hip_dsp.h")
PKG = os.path.join(ROOT, "cidana-svt-av1_amd")


def declared_symbols():
    txt = open(HDR).read()
    names = set(re.findall(r"\b(svt_hip_[a-z0-9_]+)\s*\(", txt))
    # macro-generated drop-ins
    for m in re.finditer(r"SVT_HIP_DECL_FWD\((\d+), (\d+)\)", txt):
        names.add(f"svt_hip_av1_fwd_txfm2d_{m.group(1)}x{m.group(2)}")
    for m in re.finditer(r"SVT_HIP_DECL_INV_(?:SQ|R1|R2)\((\d+), (\d+)\)", txt):
        names.add(f"svt_hip_av1_inv_txfm2d_add_{m.group(1)}x{m.group(2)}")
    for m in re.finditer(r"SVT_HIP_DECL_QUANT\((svt_hip_[a-z0-9_]+)\)", txt):
        names.add(m.group(1))
    names = {n for n in names if not n.endswith("_") and "##" not in n}
    names.discard("svt_hip_av1_fwd_txfm2d_")
    names.discard("svt_hip_av1_inv_txfm2d_add_")
    return sorted(names)


def test_library_loads_and_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = declared_symbols()
    assert len(names) >= 19 + 19 + 6 + 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "svt_hip_dsp.h"\nint main(void){ svt_txfm_param p; (void)p; return sizeof(svt_hip_rtcd_table) > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_txfm_param_layout_matches_reference():
    """TxfmParam is 24 bytes with eob at offset 20 (EbDefinitions.h:764-776)."""
    code = '#include <stddef.h>\n#include <stdio.h>\n#include "svt_hip_dsp.h"\nint main(void){printf("%zu %zu %zu", sizeof(svt_txfm_param), offsetof(svt_txfm_param, bd), offsetof(svt_txfm_param, eob));return 0;}\n'
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(code)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert out == ["24", "8", "20"]


def test_no_device_is_a_loud_error_not_a_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.SvtHipError):
        pkg.SvtHipDsp(0)
    lib = pkg.load_library()
    rc = lib.svt_hip_fwd_txfm2d_batch(None, 8, 64, None, 1, 1, 0, 8, None)
    assert rc != 0 and lib.svt_hip_last_error()


def test_product_does_not_reference_the_oracle():
    """Nothing under the package (sources or the built .so) may import, link or name oracle/."""
    for dp, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "svt_oracle" not in txt and "libsvt_oracle" not in txt and "libsvtref" not in txt, os.path.join(dp, f)
    so = os.path.join(PKG, "libsvt_hip_dsp.so")
    deps = subprocess.check_output(["ldd", so]).decode()
    assert "oracle" not in deps and "svtref" not in deps


def test_log_scale_rule(pkg):
    want = {0: 0, 1: 0, 2: 0, 3: 1, 4: 2, 9: 1, 10: 1, 11: 2, 12: 2, 15: 0, 17: 1, 18: 1}
    for s, ls in want.items():
        assert pkg.tx_log_scale(s) == ls


def test_host_tables_match_oracle():
    """cidana-svt-av1_amd/tables.py (what bench.py and the tools feed the quantiser with) against the oracle's
    av1_build_quantizer / scan restatements, which tests/test_oracle_golden.py (test_scan_tables, test_quantizer_tables:
    tests/golden/tables.npz = the reference's own arrays) pins to the reference."""
    import numpy as np
    import __graft_entry__ as ge
    import svtlibs
    pkg = ge.load_package()
    for bd in (8, 10, 12):
        a, b = pkg.tables.quant_tables(bd), svtlibs.quant_tables(bd)
        for k in b:
            assert np.array_equal(a[k], b[k]), (bd, k)
    for s in range(19):
        for t in range(16):
            if not svtlibs.txfm_allowed(s, t):
                continue
            sa, ia = pkg.tables.scan_tables(s, t)
            sb, ib = svtlibs.scan_tables(s, t)
            assert np.array_equal(sa, sb) and np.array_equal(ia, ib), (s, t)
