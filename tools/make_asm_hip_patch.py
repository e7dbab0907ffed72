#!/usr/bin/env python3
"""Generate integration/asm_hip.patch: the `ASM_HIP` backend row for the reference's dispatch.

The reference selects kernels in two ways: ~545 RTCD globals filled by setup_rtcd_internal(EbAsm), and 62 static
`xxx[ASM_TYPE_TOTAL]...` function-pointer tables in 13 files indexed at the call sites by `asm_type`.  A third backend
therefore needs (1) a new EbAsm value, (2) a third row in EVERY such table (a table is per translation unit and has no
runtime registration), (3) the RTCD overrides, (4) the encoder accepting `-asm 2`.

This script edits a scratch COPY of the touched reference files and writes the unified diff; nothing under /root/reference
is modified, and no reference source is kept in this repository beyond the diff's own context lines.

  * EbDefinitions.h        ASM_HIP = 2, ASM_TYPE_TOTAL = 3
  * every [ASM_TYPE_TOTAL] table: third row = the AVX2 row, except where libsvt_hip_dsp has the drop-in of that exact type
                           (HIP_ROWS below): NxMSadKernel / SubSampled, NxMSadLoopKernel, NxMSadAveragingKernel,
                           spatial_full_distortion_kernel, full_distortion_kernel32_bits / _cbf_zero32_bits
  * aom_dsp_rtcd.h         HIP implies the AVX2 flags for the slots the library does not cover; then every slot of the
                           library's registry is overridden by name through the X-macro lists of svt_hip_dsp.h
  * EbEncHandle.c          asm_type 2 accepted, svt_hip_init() at handle initialisation (before any worker thread exists)

Run here (needs /root/reference):  python tools/make_asm_hip_patch.py
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REF = os.environ.get("SVT_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "integration", "asm_hip.patch")

# table name -> how to build the HIP row from the AVX2 row: a function applied to each non-null entry
HIP_ROWS = {
    "NxMSadKernelSubSampled_funcPtrArray": "svt_hip_nxm_sad_kernel",
    "NxMSadKernel_funcPtrArray": "svt_hip_nxm_sad_kernel",
    "NxMSadAveragingKernel_funcPtrArray": "svt_hip_combined_averaging_sad",
    "NxMSadLoopKernel_funcPtrArray": "svt_hip_sad_loop_kernel",
    "spatial_full_distortion_kernel_func_ptr_array": "svt_hip_spatial_full_distortion_kernel",
    "full_distortion_kernel32_bits_func_ptr_array": "svt_hip_full_distortion_kernel32_bits",
    "full_distortion_kernel_cbf_zero32_bits_func_ptr_array": "svt_hip_full_distortion_kernel_cbf_zero32_bits",
    "compute4x4SAD_funcPtrArray": "svt_hip_nxm_sad_kernel",
}


def strip_comments_keep_len(text):
    """comments and preprocessor lines replaced by spaces (same length), so indices stay valid"""
    def blank(m):
        return re.sub(r"[^\n]", " ", m.group(0))
    text = re.sub(r"/\*.*?\*/", blank, text, flags=re.S)
    text = re.sub(r"//[^\n]*", blank, text)
    return re.sub(r"(?m)^[ \t]*#[^\n]*", blank, text)


def add_third_rows(src):
    """returns (new_text, [table names]) with a third top-level initialiser appended to every [ASM_TYPE_TOTAL] table"""
    clean = strip_comments_keep_len(src)
    out, pos, names = [], 0, []
    for m in re.finditer(r"(\w+)\s*\[\s*ASM_TYPE_TOTAL\s*\]((?:\s*\[[^\]]*\])*)\s*=", clean):
        brace = clean.find("{", m.end())
        if brace < 0 or clean[m.end():brace].strip():
            continue
        depth, i, elems, start = 0, brace, [], None
        while True:
            ch = clean[i]
            if ch == "{":
                depth += 1
                if depth == 1:
                    start = i + 1
            elif ch == "}":
                depth -= 1
                if depth == 0:
                    if clean[start:i].strip():
                        elems.append((start, i))
                    end = i
                    break
            elif ch == "," and depth == 1:
                if clean[start:i].strip():
                    elems.append((start, i))
                start = i + 1
            i += 1
        name = m.group(1)
        if "#" in src[brace:end]:
            # a table whose rows sit in preprocessor conditionals (compute4x4SAD_funcPtrArray): the third row goes in front of
            # the closing brace, outside the conditionals; it repeats the last row unless the library has the drop-in
            last = src[elems[-1][0]:elems[-1][1]].strip().split("\n")[-1].strip()
            entry = f"SVT_HIP_OR({HIP_ROWS[name]}, {last.rstrip(',')})" if name in HIP_ROWS else last.rstrip(",")
            k = end
            while src[k - 1] in " \t\n":
                k -= 1
            out.append(src[pos:k] + f"\n    // HIP (libsvt_hip_dsp{'' if name in HIP_ROWS else ': no drop-in of this type'})\n    {entry},")
            pos = k
            names.append(name)
            continue
        assert len(elems) == 2, (name, len(elems))
        a0, a1 = elems[1]
        avx2_row = src[a0:a1].rstrip()
        # drop the row's leading comment lines ("// AVX2") and indentation baseline
        body = re.sub(r"^\s*//[^\n]*\n", "", avx2_row.lstrip("\n"), count=1)
        indent = re.match(r"[ \t]*", body).group(0)
        if name in HIP_ROWS:
            fn = HIP_ROWS[name]
            def sub(mm):
                tok = mm.group(0)
                return tok if tok in ("0", "NULL") or "VoidFunc" in tok or tok.startswith("EB_") else f"SVT_HIP_OR({fn}, {tok})"
            # identifiers that are table entries: bare identifiers followed by ',' / '}' / end (casts keep their type names)
            body = re.sub(r"\b[A-Za-z_]\w*\b(?=\s*(?:,|\}|$))", sub, strip_comments_keep_len(body)).rstrip()
            body = "\n".join(l.rstrip() for l in body.split("\n") if l.strip())
        third = f"\n{indent}// HIP (libsvt_hip_dsp{'' if name in HIP_ROWS else ': no drop-in of this type, AVX2 kernel'})\n{body.rstrip().rstrip(',')},"
        # insert after the second element (after its trailing comma if any)
        tail = clean[a1:end]
        comma = tail.find(",")
        ins = a1 + comma + 1 if comma >= 0 and not tail[:comma].strip() else a1
        sep = "" if ins != a1 else ","
        out.append(src[pos:ins] + sep + third)
        pos = ins
        names.append(name)
    out.append(src[pos:])
    src = "".join(out)
    # tables with asm_type as the INNER index (ComputeMeanFunc[2][ASM_TYPE_TOTAL]): a third entry in every inner group
    clean = strip_comments_keep_len(src)
    out, pos = [], 0
    for m in re.finditer(r"(\w+)\s*\[[^\]]+\]\s*\[\s*ASM_TYPE_TOTAL\s*\]\s*=\s*\{", clean):
        i, depth = m.end() - 1, 0
        while True:
            ch = clean[i]
            if ch == "{":
                depth += 1
                if depth == 2:
                    gstart = i
            elif ch == "}":
                if depth == 2:
                    inner = clean[gstart + 1:i]
                    parts = [p_ for p_ in inner.split(",") if p_.strip()]
                    assert len(parts) == 2, (m.group(1), parts)
                    k = i
                    while src[k - 1] in " \t\n":
                        k -= 1
                    indent = re.search(r"\n([ \t]*)\S[^\n]*$", src[gstart:k]).group(1)
                    ck = i
                    while clean[ck - 1] in " \t\n":
                        ck -= 1
                    sep = "" if clean[ck - 1] == "," else ","          # the group may already end in a comma
                    out.append(src[pos:k] + f"{sep}\n{indent}// HIP (libsvt_hip_dsp: no drop-in of this type, AVX2 kernel)\n{indent}{parts[1].strip()}")
                    pos = k
                depth -= 1
                if depth == 0:
                    break
            i += 1
        names.append(m.group(1))
    out.append(src[pos:])
    return "".join(out), names


RTCD_TAIL = r'''
#ifdef SVT_HIP_BACKEND
        /* ASM_HIP: every dispatch slot libsvt_hip_dsp implements is overridden BY NAME (its registry holds a drop-in of the
         * exact signature); slots it does not implement keep the AVX2 kernels chosen above.  The X-macro lists come from
         * svt_hip_dsp.h.  aom_highbd_paeth_predictor_* are #defines to the C functions in this header, not pointers.
         * svt_hip_rtcd_override_slot refuses (non-zero, pointer untouched) when the device is unusable: a refused slot keeps
         * its AVX2 kernel, which is this encoder's own fallback, and the refusals are counted and logged. */
        if (asm_type == ASM_HIP) {
            int svt_hip_refused = 0;
#define SVT_HIP_OVR(slot) if (svt_hip_rtcd_override_slot(#slot, (void **)&slot) != SVT_HIP_OK) svt_hip_refused++;
#define SVT_HIP_OVR_TX(A, B, W, H) SVT_HIP_OVR(av1_fwd_txfm2d_##W##x##H) SVT_HIP_OVR(av1_inv_txfm2d_add_##W##x##H)
#define SVT_HIP_OVR_PRED(mode, MODE, W, H) SVT_HIP_OVR(aom_##mode##_predictor_##W##x##H)
#define SVT_HIP_OVR_HPRED(mode, MODE, W, H) SVT_HIP_OVR(aom_highbd_##mode##_predictor_##W##x##H)
#define SVT_HIP_OVR_SAD(W, H) SVT_HIP_OVR(aom_sad##W##x##H) SVT_HIP_OVR(aom_sad##W##x##H##x4d)
#define SVT_HIP_HIGHBD_MODES(SIZES, X) \
    SIZES(X, dc, 0) SIZES(X, dc_top, 0) SIZES(X, dc_left, 0) SIZES(X, dc_128, 0) SIZES(X, v, 0) SIZES(X, h, 0) \
    SIZES(X, smooth, 0) SIZES(X, smooth_v, 0) SIZES(X, smooth_h, 0)
            SVT_HIP_BLOCK_SIZES_2(SVT_HIP_OVR_TX, 0, 0)
            SVT_HIP_OVR(av1_inv_txfm_add)
            SVT_HIP_OVR(aom_quantize_b) SVT_HIP_OVR(aom_quantize_b_32x32) SVT_HIP_OVR(aom_quantize_b_64x64)
            SVT_HIP_OVR(aom_highbd_quantize_b) SVT_HIP_OVR(aom_highbd_quantize_b_32x32) SVT_HIP_OVR(aom_highbd_quantize_b_64x64)
            SVT_HIP_OVR(ResidualKernel)
            SVT_HIP_INTRA_MODES(SVT_HIP_BLOCK_SIZES_2, SVT_HIP_OVR_PRED)
            SVT_HIP_HIGHBD_MODES(SVT_HIP_BLOCK_SIZES_2, SVT_HIP_OVR_HPRED)
            SVT_HIP_OVR(eb_smooth_v_predictor) SVT_HIP_OVR(eb_smooth_h_predictor)
            SVT_HIP_OVR(av1_dr_prediction_z1) SVT_HIP_OVR(av1_dr_prediction_z2) SVT_HIP_OVR(av1_dr_prediction_z3)
            SVT_HIP_OVR(av1_highbd_dr_prediction_z1) SVT_HIP_OVR(av1_highbd_dr_prediction_z2) SVT_HIP_OVR(av1_highbd_dr_prediction_z3)
            SVT_HIP_OVR(av1_filter_intra_edge) SVT_HIP_OVR(av1_filter_intra_edge_high) SVT_HIP_OVR(av1_upsample_intra_edge)
            SVT_HIP_OVR(subtract_average) SVT_HIP_OVR(cfl_predict_lbd) SVT_HIP_OVR(cfl_predict_hbd) SVT_HIP_OVR(av1_txb_init_levels)
            SVT_HIP_SAD_SIZES(SVT_HIP_OVR_SAD)
#undef SVT_HIP_OVR
#undef SVT_HIP_OVR_TX
#undef SVT_HIP_OVR_PRED
#undef SVT_HIP_OVR_HPRED
#undef SVT_HIP_OVR_SAD
#undef SVT_HIP_HIGHBD_MODES
            if (svt_hip_refused)
                SVT_LOG("Warning: -asm 2: %d dispatch slots kept their AVX2 kernels (%s)\n", svt_hip_refused, svt_hip_last_error());
        }
#endif
'''


def edit(path, text):
    rel = path.replace("\\", "/")
    if rel.endswith("Codec/EbDefinitions.h"):
        old = "    ASM_AVX2,\n    ASM_TYPE_TOTAL,"
        assert old in text
        text = text.replace(old, "    ASM_AVX2,\n    ASM_HIP,        // MI355X (gfx950) kernels of libsvt_hip_dsp behind the same dispatch surface (-asm 2)\n    ASM_TYPE_TOTAL,")
        # the HIP rows of the dispatch tables name the library's drop-ins only when the build links it
        anchor = "/** Assembly Types\n"
        assert text.count(anchor) == 1
        text = text.replace(anchor, "#ifdef SVT_HIP_BACKEND\n#include \"svt_hip_dsp.h\"   /* C ABI of libsvt_hip_dsp.so + the X-macro lists of its slot families */\n"
                                    "#define SVT_HIP_OR(hip_kernel, other) hip_kernel\n#else\n#define SVT_HIP_OR(hip_kernel, other) other\n#endif\n\n" + anchor)
    if rel.endswith("Codec/aom_dsp_rtcd.h"):
        old = "        if (asm_type == ASM_AVX2)\n            flags |= HAS_AVX2;"
        assert old in text
        text = text.replace(old, "        if (asm_type == ASM_AVX2 || asm_type == ASM_HIP)    // HIP: AVX2 for every slot the library does not cover\n            flags |= HAS_AVX2;")
        # end of setup_rtcd_internal: the last closing brace before the matching #endif of RTCD_C
        i = text.index("static void setup_rtcd_internal(EbAsm asm_type)")
        depth, j = 0, text.index("{", i)
        while True:
            if text[j] == "{":
                depth += 1
            elif text[j] == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        text = text[:j] + RTCD_TAIL.lstrip("\n") + "    " + text[j:]
    if rel.endswith("Codec/EbEncHandle.c"):
        old = "    if (((int32_t)(config->asm_type) < -1) || ((int32_t)(config->asm_type) != 1)) {"
        assert old in text
        text = text.replace(old, "#ifdef SVT_HIP_BACKEND\n    if (((int32_t)(config->asm_type) != 1) && ((int32_t)(config->asm_type) != ASM_HIP)) {   // 2: HIP backend\n#else\n" + old + "\n#endif")
        old = "    setup_rtcd_internal(encHandlePtr->sequence_control_set_instance_array[0]->encode_context_ptr->asm_type);"
        assert old in text
        ectx = "encHandlePtr->sequence_control_set_instance_array[0]->encode_context_ptr"
        text = text.replace(old, "#ifdef SVT_HIP_BACKEND\n    /* -asm 2 without a usable gfx950 device: svt_hip_init says so, and the encoder keeps its AVX2 kernels - the asm_type\n"
                                 "     * every *_funcPtrArray[asm_type] table and setup_rtcd_internal see goes back to ASM_AVX2 (single-threaded here: no\n"
                                 "     * worker thread exists yet, EbEncHandle.c creates them further down).  A HIP error in the middle of a run is a\n"
                                 "     * different matter: the drop-ins have no error channel and abort (include/svt_hip_dsp.h). */\n"
                                 f"    if ({ectx}->asm_type == ASM_HIP && svt_hip_init(0) != SVT_HIP_OK) {{\n"
                                 "        SVT_LOG(\"Warning: -asm 2: no usable gfx950 device (%s); keeping the AVX2 kernels\\n\", svt_hip_last_error());\n"
                                 f"        {ectx}->asm_type = ASM_AVX2;\n    }}\n#endif\n" + old)
    new, names = add_third_rows(text)
    return new, names


def touched_files():
    cmd = ["grep", "-rl", "--include=*.h", "--include=*.c", "ASM_TYPE_TOTAL\\]", os.path.join(REF, "Source")]
    files = set(subprocess.check_output(cmd).decode().split())
    for extra in ("Source/Lib/Common/Codec/EbDefinitions.h", "Source/Lib/Common/Codec/aom_dsp_rtcd.h", "Source/Lib/Encoder/Codec/EbEncHandle.c"):
        files.add(os.path.join(REF, extra))
    return sorted(files)


def build(scratch):
    """scratch/a = pristine copies, scratch/b = edited copies; returns (patch text, table names)"""
    all_names = []
    for f in touched_files():
        rel = os.path.relpath(f, REF)
        for side in ("a", "b"):
            os.makedirs(os.path.dirname(os.path.join(scratch, side, rel)), exist_ok=True)
        shutil.copy(f, os.path.join(scratch, "a", rel))
        text = open(f, encoding="utf-8", errors="surrogateescape").read()
        new, names = edit(rel, text)
        all_names += names
        open(os.path.join(scratch, "b", rel), "w", encoding="utf-8", errors="surrogateescape").write(new)
    chunks = []
    for f in touched_files():
        rel = os.path.relpath(f, REF)
        pr = subprocess.run(["diff", "-U2", "--label", "a/" + rel, "--label", "b/" + rel, os.path.join("a", rel), os.path.join("b", rel)],
                            cwd=scratch, stdout=subprocess.PIPE)
        assert pr.returncode in (0, 1)
        chunks.append(pr.stdout.decode("utf-8", errors="surrogateescape"))
    return "".join(chunks), all_names


def main():
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not present")
    with tempfile.TemporaryDirectory(prefix="asm_hip_") as td:
        patch, names = build(td)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w", encoding="utf-8", errors="surrogateescape").write(patch)
    print(f"wrote {OUT}: {len(patch.splitlines())} lines, {len(names)} tables got a third row "
          f"({sum(n in HIP_ROWS for n in names)} with libsvt_hip_dsp drop-ins)")


if __name__ == "__main__":
    main()
