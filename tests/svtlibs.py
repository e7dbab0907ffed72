"""Loaders for the three native libraries used by the tests.

  oracle()  -> oracle/libsvt_oracle.so   our scalar-C restatement (checker)
  ref()     -> oracle/_ref/libsvtref.so  the reference's own sources compiled
               here (None when absent: it cannot be rebuilt without
               /root/reference, but the prebuilt .so travels to the GPU box)
  product() -> cidana-svt-av1_amd/libsvt_hip_dsp.so  the C-ABI under test
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cidana-svt-av1_amd")

TX_SIZES = ["4X4", "8X8", "16X16", "32X32", "64X64", "4X8", "8X4", "8X16", "16X8", "16X32", "32X16",
            "32X64", "64X32", "4X16", "16X4", "8X32", "32X8", "16X64", "64X16"]
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
TX_TYPES = ["DCT_DCT", "ADST_DCT", "DCT_ADST", "ADST_ADST", "FLIPADST_DCT", "DCT_FLIPADST",
            "FLIPADST_FLIPADST", "ADST_FLIPADST", "FLIPADST_ADST", "IDTX", "V_DCT", "H_DCT", "V_ADST",
            "H_ADST", "V_FLIPADST", "H_FLIPADST"]


def txfm_allowed(tx_size, tx_type):
    """test/TxfmCommon.h:172-181"""
    m = max(TX_W[tx_size], TX_H[tx_size])
    if m == 64:
        return tx_type == 0
    if m == 32:
        return tx_type in (0, 9)
    return True


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_cache = {}


def oracle():
    if "oracle" not in _cache:
        path = os.path.join(ROOT, "oracle", "libsvt_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        L = ctypes.CDLL(path)
        L.svt_oracle_sad.restype = ctypes.c_uint32
        L.svt_oracle_sse.restype = ctypes.c_uint64
        L.svt_oracle_sad_avg.restype = ctypes.c_uint32
        L.svt_oracle_fwd_txfm2d_pack64.restype = ctypes.c_uint64
        _cache["oracle"] = L
    return _cache["oracle"]


def ref():
    if "ref" not in _cache:
        path = os.path.join(ROOT, "oracle", "_ref", "libsvtref.so")
        L = None
        if os.path.exists(path):
            L = ctypes.CDLL(path)
            L.ref_get_scan.restype = ctypes.POINTER(ctypes.c_int16)
            L.ref_bench_fwd_quant_sad.restype = ctypes.c_double
            L.fast_loop_nx_m_sad_kernel.restype = ctypes.c_uint32
            L.spatial_full_distortion_kernel.restype = ctypes.c_uint64
        _cache["ref"] = L
    return _cache["ref"]


def product():
    if "product" not in _cache:
        path = os.path.join(PKG, "libsvt_hip_dsp.so")
        _cache["product"] = ctypes.CDLL(path)
    return _cache["product"]


def quant_tables(bd=8):
    """y-plane tables from the oracle's av1_build_quantizer restatement: dict of int16[256][8]"""
    L = oracle()
    t = {k: np.zeros((256, 8), np.int16) for k in ("zbin", "round", "quant", "quant_shift", "dequant")}
    L.svt_oracle_build_quantizer(bd, ptr(t["zbin"]), ptr(t["round"]), ptr(t["quant"]),
                                 ptr(t["quant_shift"]), ptr(t["dequant"]))
    return t


def scan_tables(tx_size, tx_type):
    L = oracle()
    sc = np.zeros(1024, np.int16)
    isc = np.zeros(1024, np.int16)
    n = L.svt_oracle_get_scan(tx_size, tx_type, ptr(sc), ptr(isc))
    return sc[:n].copy(), isc[:n].copy()


class HmeParams(ctypes.Structure):
    """svt_oracle_hme_params / svt_hip_hme_params (same layout)"""
    _fields_ = [(n, ctypes.c_int32) for n in ("search_area_width", "search_area_height", "x_origin_offset", "y_origin_offset",
                                              "pad_width", "pad_height", "ref_width", "ref_height", "round_down", "mv_shift")]


def hme_params(level, hme_w, hme_h, region_w, region_h, mult_x, mult_y, pad, ref_w, ref_h):
    p = HmeParams()
    oracle().svt_oracle_hme_params_for_level(level, ptr(hme_w), ptr(hme_h), region_w, region_h, int(hme_w.sum()), int(hme_h.sum()),
                                             mult_x, mult_y, pad, pad, ref_w, ref_h, ctypes.byref(p))
    return p
