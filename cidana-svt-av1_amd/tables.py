"""Host-side tables a caller of the quantiser entry points needs when it is not the encoder itself (bench.py, tools):
the y-plane quantiser rows of av1_build_quantizer (EbModeDecisionConfigurationProcess.c:301-330, 429-520) and the
coefficient scan orders av1_scan_orders[tx_size][tx_type] (EbTransforms.h:3349-3870).  Inside the encoder these come
from its own Quants / SCAN_ORDER structures and are passed to the C ABI as they are."""
import numpy as np

from .qlookup_data import AC_QLOOKUP, DC_QLOOKUP

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
_V_TYPES = (10, 12, 14)          # V_DCT, V_ADST, V_FLIPADST: row-major scan
_H_TYPES = (11, 13, 15)          # H_DCT, H_ADST, H_FLIPADST: column-major scan


def quant_tables(bd=8):
    """dict of int16 [256][8] arrays: zbin, round, quant, quant_shift, dequant (entry 0 = DC, 1..7 = AC), per qindex."""
    bi = {8: 0, 10: 1, 12: 2}[bd]
    thr = {8: 148, 10: 592, 12: 2368}[bd]
    t = {k: np.zeros((256, 8), np.int16) for k in ("zbin", "round", "quant", "quant_shift", "dequant")}
    for q in range(256):
        dcq = DC_QLOOKUP[bi][q]
        zbin_factor = 64 if q == 0 else (84 if dcq < thr else 80)         # get_qzbin_factor
        round_factor = 64 if q == 0 else 48
        for i in range(8):
            d = dcq if i == 0 else AC_QLOOKUP[bi][q]
            l = d.bit_length() - 1                                         # invert_quant
            m = 1 + (1 << (16 + l)) // d
            t["quant"][q, i] = m - (1 << 16)                               # in (-32768, 1]
            t["quant_shift"][q, i] = 1 << (16 - l)
            t["zbin"][q, i] = (zbin_factor * d + 64) >> 7
            t["round"][q, i] = (round_factor * d) >> 7
            t["dequant"][q, i] = d
    return t


def scan_tables(tx_size, tx_type):
    """(scan, iscan) int16 arrays over the kept min(W,32) x min(H,32) coefficients."""
    w, h = min(TX_W[tx_size], 32), min(TX_H[tx_size], 32)
    n = w * h
    if tx_type in _V_TYPES:
        sc = list(range(n))
    elif tx_type in _H_TYPES:
        sc = [r * w + c for c in range(w) for r in range(h)]
    else:                      # diagonal scan: square blocks alternate direction, tall blocks walk down, wide blocks up
        sc = []
        for d in range(w + h - 1):
            rows = range(max(0, d - (w - 1)), min(d, h - 1) + 1)
            ascending = True if h > w else (False if w > h else bool(d & 1))
            sc += [r * w + (d - r) for r in (rows if ascending else reversed(rows))]
    scan = np.array(sc, np.int16)
    iscan = np.zeros(n, np.int16)
    iscan[scan] = np.arange(n, dtype=np.int16)
    return scan, iscan
