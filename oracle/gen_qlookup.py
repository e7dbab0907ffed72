#!/usr/bin/env python3
"""Emit oracle/qlookup_data.h: the AV1 Dc_Qlookup / Ac_Qlookup tables (AV1 spec
7.12.2) for bit depths 8/10/12, read out of the reference build by CALLING
av1_dc_quant_Q3 / av1_ac_quant_Q3 (EbModeDecisionConfigurationProcess.c:279-298)
in oracle/_ref/libsvtref.so.  The tables are normative data with no closed form.
Run here (needs oracle/_ref); the emitted header is committed. TEST INFRASTRUCTURE."""
import ctypes, os, sys
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "_ref", "libsvtref.so"))
L.av1_dc_quant_Q3.restype = ctypes.c_int16
L.av1_ac_quant_Q3.restype = ctypes.c_int16
lines = ["/* oracle/qlookup_data.h - AV1 spec Dc_Qlookup/Ac_Qlookup [bd 8,10,12][qindex].",
         " * Emitted by oracle/gen_qlookup.py from the reference build. TEST INFRASTRUCTURE. */",
         "#ifndef ORC_QLOOKUP_DATA_H", "#define ORC_QLOOKUP_DATA_H", "#include <stdint.h>"]
for name, fn in (("k_dc_qlookup", L.av1_dc_quant_Q3), ("k_ac_qlookup", L.av1_ac_quant_Q3)):
    lines.append(f"static const int16_t {name}[3][256] = {{")
    for bd in (8, 10, 12):
        vals = [int(fn(q, 0, bd)) for q in range(256)]
        lines.append("    {" + ", ".join(map(str, vals)) + "},")
    lines.append("};")
lines.append("#endif")
open(os.path.join(here, "qlookup_data.h"), "w").write("\n".join(lines) + "\n")
print("wrote qlookup_data.h")
