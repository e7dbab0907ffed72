"""CPU: integration/asm_hip.patch (the ASM_HIP backend row) applies cleanly to a scratch copy of the reference
files it touches and the patched sources still COMPILE - with the HIP backend (-DSVT_HIP_BACKEND, include/svt_hip_dsp.h: the
drop-ins must have the exact types of the table slots, -Werror=incompatible-pointer-types) and without it.  The reference
tree is read-only and is never modified; without /root/reference the test is skipped (the patch itself is committed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "asm_hip.patch")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Source")), reason="needs /root/reference to apply the patch to a scratch copy")


def patched_files():
    return re.findall(r"^\+\+\+ b/(\S+)", open(PATCH, encoding="utf-8", errors="surrogateescape").read(), flags=re.M)


@pytest.fixture(scope="module")
def scratch(tmp_path_factory):
    td = tmp_path_factory.mktemp("asm_hip")
    for rel in patched_files():
        os.makedirs(os.path.dirname(td / rel), exist_ok=True)
        shutil.copy(os.path.join(REF, rel), td / rel)
    pr = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", PATCH], cwd=td, capture_output=True, text=True)
    assert pr.returncode == 0, pr.stdout + pr.stderr
    assert "FAILED" not in pr.stdout and "fuzz" not in pr.stdout, pr.stdout
    return td


def includes(td):
    inc = []
    for d in ("Source/API", "Source/Lib/Common/Codec", "Source/Lib/Encoder/Codec"):
        inc += ["-I", str(td / d)]                        # patched copies first
    for d in ("Source/API", "Source/Lib/Common/Codec", "Source/Lib/Common/C_DEFAULT", "Source/Lib/Common/ASM_SSE2", "Source/Lib/Common/ASM_SSSE3",
              "Source/Lib/Common/ASM_SSE4_1", "Source/Lib/Common/ASM_AVX2", "Source/Lib/Encoder/Codec"):
        inc += ["-I", os.path.join(REF, d)]
    return inc + ["-I", os.path.join(ROOT, "include")]


def test_patch_is_current():
    """the committed patch is what tools/make_asm_hip_patch.py generates from this reference snapshot"""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tools", "make_asm_hip_patch.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    with tempfile.TemporaryDirectory() as td:
        patch, names = mk.build(td)
    assert patch == open(PATCH, encoding="utf-8", errors="surrogateescape").read()
    assert len(names) == 58 and sum(n in mk.HIP_ROWS for n in names) == 8      # 51 tables indexed [asm_type] first + 7 indexed [n][asm_type]


def test_every_asm_type_table_has_three_rows(scratch):
    """ASM_TYPE_TOTAL becomes 3: a table left with two initialisers would dispatch through NULL under -asm 2"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tools", "make_asm_hip_patch.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    n = 0
    for rel in patched_files():
        txt = open(scratch / rel, encoding="utf-8", errors="surrogateescape").read()
        n += txt.count("// HIP (libsvt_hip_dsp")
    assert n == 51 + 3 * 9 + 4 * 2          # 51 tables indexed [asm_type] first; [9][asm_type] x 3 and [2][asm_type] x 4: one entry per inner group
    defs = open(scratch / "Source/Lib/Common/Codec/EbDefinitions.h", errors="surrogateescape").read()
    assert re.search(r"ASM_AVX2,\s*ASM_HIP,[^\n]*\n\s*ASM_TYPE_TOTAL", defs)
    # untouched reference files must not hold further tables
    out = subprocess.run(["grep", "-rl", "--include=*.h", "--include=*.c", r"ASM_TYPE_TOTAL\]", os.path.join(REF, "Source")], capture_output=True, text=True).stdout.split()
    # (EbMcp.h only DECLARES the tables EbMcpTables.c defines: their size follows ASM_TYPE_TOTAL by itself)
    assert {os.path.relpath(f, REF) for f in out} - {"Source/Lib/Common/Codec/EbMcp.h"} <= set(patched_files())


@pytest.mark.parametrize("hip", [True, False])
def test_patched_sources_compile(scratch, hip):
    flags = ["gcc", "-std=gnu99", "-fsyntax-only", "-w", "-Werror=incompatible-pointer-types", "-Werror=implicit-function-declaration", "-mavx2"]
    if hip:
        flags.append("-DSVT_HIP_BACKEND")
    # (1) the dispatch header in the translation unit that defines the RTCD globals, with every patched table header
    tu = scratch / ("tu_hip.c" if hip else "tu_plain.c")
    tu.write_text('#define RTCD_C\n#include "EbDefinitions.h"\n#include "aom_dsp_rtcd.h"\n#include "EbComputeSAD.h"\n#include "EbPictureOperators.h"\n'
                  '#include "EbMeSadCalculation.h"\n#include "EbComputeMean.h"\n#include "EbPackUnPack.h"\n#include "EbTransforms.h"\n#include "EbMcp.h"\n'
                  '#include "EbAvcStyleMcp.h"\n#include "EbIntraPrediction.h"\n#include "EbPictureAnalysisProcess.h"\n'
                  'int use(void) { setup_rtcd_internal(ASM_HIP); return (int)ASM_TYPE_TOTAL; }\n')
    pr = subprocess.run(flags + includes(scratch) + [str(tu)], capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr[-4000:]
    # (2) the patched .c files themselves (tables inside EbMotionEstimation.c / EbProductCodingLoop.c / EbMcpTables.c, EbEncHandle.c)
    for rel in patched_files():
        if rel.endswith(".c"):
            pr = subprocess.run(flags + includes(scratch) + [str(scratch / rel)], capture_output=True, text=True)
            assert pr.returncode == 0, (rel, pr.stderr[-4000:])
