"""cidana-svt-av1_amd — MI355X-native SVT-AV1 block-DSP hot path.

The product is ``libsvt_hip_dsp.so`` (C ABI: include/svt_hip_dsp.h, hand-written
gfx950 HIP kernels under csrc/).  This Python module is only the host-side
mirror used by the tests and the benchmark: it binds the C ABI with ctypes and
passes raw device pointers taken from torch tensors (torch is plumbing for
device memory, streams and torch.distributed — nothing is computed in torch).

There is deliberately NO fallback: if the shared library is missing or the HIP
device cannot be initialised, construction raises.

The directory name contains '-', so import it through
``__graft_entry__.load_package()`` (importlib) rather than ``import``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_int16, c_int32, c_size_t, c_uint32, c_void_p

from . import tables  # noqa: F401  (host-side quantiser / scan tables for callers outside the encoder)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libsvt_hip_dsp.so")

TX_SIZE_NAMES = ["TX_4X4", "TX_8X8", "TX_16X16", "TX_32X32", "TX_64X64", "TX_4X8", "TX_8X4", "TX_8X16",
                 "TX_16X8", "TX_16X32", "TX_32X16", "TX_32X64", "TX_64X32", "TX_4X16", "TX_16X4", "TX_8X32",
                 "TX_32X8", "TX_16X64", "TX_64X16"]
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
TX_TYPE_NAMES = ["DCT_DCT", "ADST_DCT", "DCT_ADST", "ADST_ADST", "FLIPADST_DCT", "DCT_FLIPADST",
                 "FLIPADST_FLIPADST", "ADST_FLIPADST", "FLIPADST_ADST", "IDTX", "V_DCT", "H_DCT", "V_ADST",
                 "H_ADST", "V_FLIPADST", "H_FLIPADST"]
TX_32X32 = 3
DCT_DCT = 0

SVT_HIP_OK = 0


class SvtHipError(RuntimeError):
    pass


def tx_log_scale(tx_size: int) -> int:
    """av1_get_tx_scale (EbTransforms.h:317-329)."""
    pels = TX_W[tx_size] * TX_H[tx_size]
    return 2 if pels > 1024 else (1 if pels > 256 else 0)


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise SvtHipError(f"{path} not built - run `python __graft_entry__.py build` (no CPU fallback exists)")
    L = ctypes.CDLL(path)
    L.svt_hip_last_error.restype = ctypes.c_char_p
    L.svt_hip_device_name.restype = ctypes.c_char_p
    L.svt_hip_malloc.restype = c_void_p
    L.svt_hip_malloc.argtypes = [c_size_t]
    L.svt_hip_free.argtypes = [c_void_p]
    L.svt_hip_memcpy_h2d.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    L.svt_hip_memcpy_d2h.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    L.svt_hip_stream_sync.argtypes = [c_void_p]
    L.svt_hip_malloc_spread.argtypes = [c_void_p, c_int, c_size_t, c_void_p]
    L.svt_hip_me_setup_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_uint32] + [c_void_p] * 7 + [c_size_t, c_void_p]
    L.svt_hip_me_fullpel_search_areas_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                        c_void_p, c_void_p, c_uint32, c_size_t, c_void_p]
    L.svt_hip_me_bipred_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_uint32] + [c_void_p] * 5 + [c_uint32, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_size_t, c_void_p]
    L.svt_hip_fwd_txfm2d_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]
    L.svt_hip_pack64_batch.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p]
    L.svt_hip_inv_txfm2d_add_batch.argtypes = [c_void_p, c_void_p, c_int, c_int32, c_size_t, c_void_p, c_size_t,
                                               c_int, c_int, c_int, c_void_p]
    L.svt_hip_quantize_b_batch.argtypes = [c_void_p, c_size_t, c_int] + [c_void_p] * 4 + [c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_int, c_size_t, c_void_p]
    L.svt_hip_encode_recon_planes_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_size_t,
                                                    c_int, c_int, c_int, c_int] + [c_void_p] * 5 + [c_void_p] * 7
    L.svt_hip_encode_recon_batch.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_int] + [c_void_p] * 5 + [c_void_p] * 8
    L.svt_hip_fwd_quant_sad_batch.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_int] + [c_void_p] * 5 + \
                                             [c_void_p] * 6 + [c_void_p]
    for name in ("svt_hip_sad_batch", "svt_hip_sse_batch"):
        getattr(L, name).argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_size_t, c_uint32, c_uint32,
                                     c_void_p, c_size_t, c_void_p]
    L.svt_hip_residual_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_size_t, c_void_p, c_uint32,
                                         c_size_t, c_uint32, c_uint32, c_size_t, c_void_p]
    L.svt_hip_sad_search_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_uint32, c_size_t,
                                           c_uint32, c_uint32, c_int16, c_int16, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_void_p]
    L.svt_hip_intra_pred_batch.argtypes = [c_void_p, c_int32, c_size_t, c_void_p, c_void_p, c_void_p, c_int32, c_int,
                                           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_size_t, c_void_p]
    L.svt_hip_filter_intra_edge_batch.argtypes = [c_void_p, c_int32, c_int, c_int, c_int, c_size_t, c_void_p]
    L.svt_hip_upsample_intra_edge_batch.argtypes = [c_void_p, c_int32, c_int, c_int, c_int, c_size_t, c_void_p]
    L.svt_hip_full_distortion32_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_size_t, c_uint32,
                                                  c_uint32, c_int, c_void_p, c_size_t, c_void_p]
    L.svt_hip_fwd_quant_batch.argtypes = [c_void_p, c_size_t, c_int, c_int, c_int] + [c_void_p] * 5 + [c_void_p] * 5 + [c_void_p]
    L.svt_hip_fwd_quant_planes_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_size_t, c_int, c_int,
                                                 c_int, c_int] + [c_void_p] * 5 + [c_void_p] * 7 + [c_void_p]
    L.svt_hip_me_sb_search_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_size_t, c_int, c_int,
                                             c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    L.svt_hip_me_sb_search_planes_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_void_p, c_uint32, c_void_p, c_int, c_int,
                                                    c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    L.svt_hip_sad_search_planes_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_void_p, c_uint32, c_uint32, c_void_p,
                                                  c_uint32, c_uint32, c_int16, c_int16, c_void_p, c_void_p, c_void_p,
                                                  c_size_t, c_void_p]
    L.svt_hip_cfl_luma_subsampling_420_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_int, c_void_p, c_uint32,
                                                         c_size_t, c_uint32, c_uint32, c_int, c_size_t, c_void_p]
    L.svt_hip_subtract_average_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_uint32, c_uint32, c_int32, c_int32, c_size_t,
                                                 c_void_p]
    L.svt_hip_cfl_predict_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_void_p, c_uint32, c_void_p,
                                            c_void_p, c_int, c_uint32, c_uint32, c_int, c_size_t, c_void_p]
    L.svt_hip_ois_work_bytes.argtypes = [c_uint32, c_int, c_size_t]
    L.svt_hip_ois_work_bytes.restype = c_size_t
    L.svt_hip_ois_search_batch.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_uint32, c_void_p, c_void_p, c_int,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]
    L.svt_hip_txb_init_levels_batch.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_uint32, c_uint32, c_size_t, c_void_p]
    L.svt_hip_encode_recon_frame.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.svt_hip_encode_recon_frame_ex.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.svt_hip_sad_planes_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_void_p, c_uint32, c_void_p, c_uint32, c_uint32, c_void_p,
                                           c_size_t, c_void_p]
    L.svt_hip_sad_x4d_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_void_p, c_uint32, c_void_p, c_uint32, c_uint32,
                                        c_void_p, c_size_t, c_void_p]
    L.svt_hip_sad_avg_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_size_t,
                                        c_uint32, c_uint32, c_void_p, c_size_t, c_void_p]
    L.svt_hip_residual16_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_size_t, c_void_p, c_uint32, c_size_t,
                                           c_uint32, c_uint32, c_size_t, c_void_p]
    L.svt_hip_picture_full_distortion32_batch.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_uint32, c_uint32, c_void_p, c_int,
                                                          c_void_p, c_size_t, c_void_p]
    L.svt_hip_hme_level_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p]
    L.svt_hip_hme_level_regions_batch.argtypes = [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                                  c_void_p, c_void_p, c_size_t, c_void_p]
    L.svt_hip_hme_level_params.argtypes = [c_int, c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32,
                                           c_uint32, c_uint32, c_uint32, c_void_p]
    L.svt_hip_me_fullpel_search_batch.argtypes = [c_void_p, c_uint32, c_size_t, c_void_p, c_void_p, c_uint32, c_size_t, c_void_p,
                                                  c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_uint32,
                                                  c_size_t, c_void_p]
    L.svt_hip_build_intra_predictors_batch.argtypes = [c_void_p, c_int32, c_size_t, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int,
                                                       c_int, c_int, c_size_t, c_void_p]
    L.svt_hip_ois_search_frame.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_int, c_void_p]
    L.svt_hip_intra_neighbor_px.argtypes = [c_void_p, c_void_p]
    L.svt_hip_intra_has_top_right.argtypes = [c_int] * 12
    L.svt_hip_intra_has_bottom_left.argtypes = [c_int] * 12
    L.svt_hip_y4m_parse_header.argtypes = [ctypes.c_char_p, c_void_p]
    L.svt_hip_y4m_frame_bytes.argtypes = [c_void_p]
    L.svt_hip_y4m_frame_bytes.restype = c_size_t
    L.svt_hip_y4m_open.argtypes = [ctypes.c_char_p, c_void_p, c_void_p]
    L.svt_hip_y4m_read_frame.argtypes = [c_void_p, c_void_p, c_size_t]
    L.svt_hip_y4m_close.argtypes = [c_void_p]
    L.svt_hip_picture_import.argtypes = [c_void_p, c_uint32, c_uint32, c_int, c_int, c_int, c_void_p, c_uint32, c_void_p, c_uint32, c_void_p,
                                         c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, c_void_p]
    L.svt_hip_picture_pad.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, c_int, c_void_p]
    L.svt_hip_picture_luma8.argtypes = [c_void_p, c_uint32, c_void_p, c_uint32, c_uint32, c_uint32, c_int, c_void_p]
    L.svt_hip_picture_decimate.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_uint32,
                                           c_uint32, c_uint32, c_void_p]
    return L


class Y4mInfo(ctypes.Structure):
    """svt_hip_y4m_info: what read_y4m_header (Source/App/EncApp/EbAppInputy4m.c:35) takes from the header line"""
    _fields_ = [("width", c_uint32), ("height", c_uint32), ("fr_n", c_uint32), ("fr_d", c_uint32), ("bit_depth", c_uint32),
                ("interlaced", c_uint32), ("chroma", ctypes.c_char * 8), ("scan_type", ctypes.c_char)]


def y4m_parse_header(lib, line):
    """HOST: the header line that follows the "YUV4MPEG2" signature -> Y4mInfo, or SvtHipError where the reference's application
    rejects the file (svt_hip_y4m_parse_header; needs no device)"""
    info = Y4mInfo()
    rc = lib.svt_hip_y4m_parse_header(line if isinstance(line, bytes) else line.encode(), ctypes.addressof(info))
    if rc != 0:
        raise SvtHipError(f"svt_hip_y4m_parse_header: {lib.svt_hip_last_error().decode()}")
    return info


class Y4mReader:
    """HOST: a y4m file through svt_hip_y4m_open / _read_frame / _close (check_if_y4m, read_y4m_header, read_y4m_frame_delimiter).
    read_into(buffer) fills any writable buffer object (numpy array, pinned torch tensor via .numpy()) with the next frame's planes
    as the file holds them (Y, Cb, Cr back to back) and returns False at the end of the file."""

    def __init__(self, lib, path):
        self.lib = lib
        self.info = Y4mInfo()
        h = c_void_p()
        rc = lib.svt_hip_y4m_open(os.fsencode(path), ctypes.byref(h), ctypes.addressof(self.info))
        if rc != 0:
            raise SvtHipError(f"svt_hip_y4m_open: {lib.svt_hip_last_error().decode()}")
        self.h = h
        self.frame_bytes = lib.svt_hip_y4m_frame_bytes(ctypes.addressof(self.info))

    def read_into(self, arr):
        import numpy as np
        a = np.asarray(arr)
        rc = self.lib.svt_hip_y4m_read_frame(self.h, a.ctypes.data, a.nbytes)
        if rc < 0:
            raise SvtHipError(f"svt_hip_y4m_read_frame: {self.lib.svt_hip_last_error().decode()}")
        return rc == 1

    def close(self):
        if self.h:
            self.lib.svt_hip_y4m_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _np16(a):
    import numpy as np
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int16))
    assert a.size >= 2
    return a


class SvtHipDsp:
    """Batched device API on torch CUDA(HIP) tensors.  Names follow the reference's
    dispatch slots (aom_dsp_rtcd.h): fwd_txfm2d <-> av1_fwd_txfm2d_WxH, ..."""

    def __init__(self, device: int = 0):
        import torch
        self.torch = torch
        self.lib = load_library()
        rc = self.lib.svt_hip_init(int(device))
        if rc != SVT_HIP_OK:
            raise SvtHipError(f"svt_hip_init({device}) = {rc}: {self.lib.svt_hip_last_error().decode()}")
        self.device = torch.device("cuda", device)

    # -- helpers ----------------------------------------------------------------
    def _check(self, rc, what):
        if rc != SVT_HIP_OK:
            raise SvtHipError(f"{what} = {rc}: {self.lib.svt_hip_last_error().decode()}")

    def _stream(self):
        return c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return c_void_p(t.data_ptr())

    def device_name(self):
        return self.lib.svt_hip_device_name().decode()

    def alloc_spread(self, specs, gap_bytes=32 << 30):
        """torch tensors that lie far apart in device memory (what svt_hip_malloc_spread does for C callers, through torch's
        allocator so that the tensors are ordinary tensors): a temporary spacer between consecutive allocations, freed again.
        specs: [(shape, dtype), ...].  Arrays a kernel writes at the same time run 20 - 25 % faster this way."""
        t = self.torch
        t.cuda.empty_cache()                                  # the tensors below must be fresh allocations, not cached blocks
        out, spacers = [], []
        for k, (shape, dt) in enumerate(specs):
            out.append(t.empty(shape, dtype=dt, device=self.device))
            if k + 1 < len(specs):
                try:
                    spacers.append(t.empty(gap_bytes, dtype=t.uint8, device=self.device))
                except RuntimeError:                          # no room for a spacer: the next tensor follows directly
                    pass
        del spacers
        t.cuda.empty_cache()
        return out

    def membw_probe(self, mode, dst, src=None, nbytes=None):
        """svt_hip_membw_probe: mode 0 fill / 1 copy / 2 the fused kernel's 1 : 6 read / write mix; enqueues one kernel"""
        if nbytes is None:
            nbytes = (src if mode else dst).numel() * (src if mode else dst).element_size()
        self.lib.svt_hip_membw_probe.argtypes = [c_int, c_void_p, c_void_p, c_size_t, c_void_p]
        self._check(self.lib.svt_hip_membw_probe(mode, self._p(dst), self._p(src) if src is not None else None, nbytes,
                                                 self._stream()), "svt_hip_membw_probe")

    def membw_probe_chain(self, in0, in1, outs, nblocks):
        """svt_hip_membw_probe_chain: the fused 32x32 chain's traffic alone on the caller's arrays (2 x 1 KiB in, 3 x 4 KiB out per block)"""
        self.lib.svt_hip_membw_probe_chain.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        self._check(self.lib.svt_hip_membw_probe_chain(self._p(in0), self._p(in1), self._p(outs[0]), self._p(outs[1]), self._p(outs[2]), nblocks,
                                                       self._stream()), "svt_hip_membw_probe_chain")

    # -- K1 ---------------------------------------------------------------------
    def fwd_txfm2d(self, residual, tx_size, tx_type, bd=8, out=None):
        """residual: int16 [n, H, W] contiguous (dense blocks). -> int32 [n, H*W]"""
        t = self.torch
        w, h = TX_W[tx_size], TX_H[tx_size]
        n = residual.shape[0]
        assert residual.dtype == t.int16 and residual.is_contiguous() and tuple(residual.shape[1:]) == (h, w)
        if out is None:
            out = t.empty((n, h * w), dtype=t.int32, device=residual.device)
        self._check(self.lib.svt_hip_fwd_txfm2d_batch(self._p(residual), w, w * h, self._p(out), n, tx_size, tx_type,
                                                       bd, self._stream()), "svt_hip_fwd_txfm2d_batch")
        return out

    def pack64(self, coeff, tx_size, want_energy=True):
        t = self.torch
        n = coeff.shape[0]
        energy = t.zeros(n, dtype=t.int64, device=coeff.device) if want_energy else None
        self._check(self.lib.svt_hip_pack64_batch(self._p(coeff), self._p(energy) if want_energy else None, n, tx_size,
                                                   self._stream()), "svt_hip_pack64_batch")
        return energy

    # -- K2 ---------------------------------------------------------------------
    def inv_txfm2d_add(self, coeff, dst, tx_size, tx_type, bd=8, dst_stride=None, dst_block_pitch=None, offsets=None):
        """coeff int32 [n, min(W,32)*min(H,32)]; dst uint8/int16(as uint16) tensor updated in place."""
        t = self.torch
        w, h = TX_W[tx_size], TX_H[tx_size]
        n = coeff.shape[0]
        is16 = 0 if dst.dtype == t.uint8 else 1
        if dst_stride is None:
            dst_stride, dst_block_pitch = w, w * h
        self._check(self.lib.svt_hip_inv_txfm2d_add_batch(self._p(coeff), self._p(dst), is16, dst_stride,
                                                           dst_block_pitch or 0,
                                                           self._p(offsets) if offsets is not None else None, n,
                                                           tx_size, tx_type, bd, self._stream()),
                    "svt_hip_inv_txfm2d_add_batch")
        return dst

    # -- K3 ---------------------------------------------------------------------
    def quantize_b(self, coeff, qrow, iscan, log_scale, skip_block=0):
        """coeff int32 [n, ncoef]; qrow: dict of int16[8] rows (zbin, round, quant, quant_shift, dequant);
        iscan: int16 device tensor [ncoef]. -> (qcoeff, dqcoeff, eob[uint16 as int16 view])"""
        t = self.torch
        n, nc = coeff.shape
        q = t.empty_like(coeff)
        dq = t.empty_like(coeff)
        eob = t.zeros(n, dtype=t.int16, device=coeff.device)
        tabs = [_np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        self._check(self.lib.svt_hip_quantize_b_batch(self._p(coeff), nc, skip_block, tabs[0].ctypes.data,
                                                       tabs[1].ctypes.data, tabs[2].ctypes.data, tabs[3].ctypes.data,
                                                       self._p(q), self._p(dq), tabs[4].ctypes.data, self._p(eob),
                                                       self._p(iscan), log_scale, n, self._stream()),
                    "svt_hip_quantize_b_batch")
        return q, dq, eob

    # -- headline chain -----------------------------------------------------------
    def fwd_quant_sad(self, src, pred, tx_size, tx_type, qrow, iscan, outs=None, want_sad=True):
        """src, pred: uint8 [n, H, W].  -> coeff, qcoeff, dqcoeff (int32 [n, ncoef]), eob, sad"""
        t = self.torch
        n = src.shape[0]
        nc = min(TX_W[tx_size], 32) * min(TX_H[tx_size], 32)
        if outs is None:
            outs = (t.empty((n, nc), dtype=t.int32, device=src.device), t.empty((n, nc), dtype=t.int32, device=src.device),
                    t.empty((n, nc), dtype=t.int32, device=src.device), t.zeros(n, dtype=t.int16, device=src.device),
                    t.zeros(n, dtype=t.int32, device=src.device))
        co, q, dq, eob, sad = outs
        tabs = [_np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        self._check(self.lib.svt_hip_fwd_quant_sad_batch(self._p(src), self._p(pred), n, tx_size, tx_type,
                                                          tabs[0].ctypes.data, tabs[1].ctypes.data, tabs[2].ctypes.data,
                                                          tabs[3].ctypes.data, tabs[4].ctypes.data, self._p(iscan),
                                                          self._p(co), self._p(q), self._p(dq), self._p(eob),
                                                          self._p(sad) if want_sad else None, self._stream()),
                    "svt_hip_fwd_quant_sad_batch")
        return outs

    def encode_recon(self, src, pred, tx_size, tx_type, qrow, iscan, keep_coeff=True, want_sad=True, outs=None):
        """Encode-pass chain (Av1EncodeLoop): src, pred uint8 [n, H, W] ->
        dict(coeff, qcoeff, dqcoeff, eob, sad, recon); coeff/dqcoeff are None when keep_coeff is False.
        outs = (qcoeff, eob, recon): caller-placed outputs (see alloc_spread)."""
        t = self.torch
        n = src.shape[0]
        nc = min(TX_W[tx_size], 32) * min(TX_H[tx_size], 32)
        mk = lambda: t.empty((n, nc), dtype=t.int32, device=src.device)
        co, dq = (mk(), mk()) if keep_coeff else (None, None)
        q = outs[0] if outs else mk()
        eob = outs[1] if outs else t.zeros(n, dtype=t.int16, device=src.device)
        sad = t.zeros(n, dtype=t.int32, device=src.device) if want_sad else None
        recon = outs[2] if outs else t.empty_like(pred)
        tabs = [_np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        self._check(self.lib.svt_hip_encode_recon_batch(self._p(src), self._p(pred), n, tx_size, tx_type,
                                                         tabs[0].ctypes.data, tabs[1].ctypes.data, tabs[2].ctypes.data,
                                                         tabs[3].ctypes.data, tabs[4].ctypes.data, self._p(iscan),
                                                         self._p(co) if keep_coeff else None, self._p(q),
                                                         self._p(dq) if keep_coeff else None, self._p(eob),
                                                         self._p(sad) if want_sad else None, self._p(recon), self._stream()),
                    "svt_hip_encode_recon_batch")
        return {"coeff": co, "qcoeff": q, "dqcoeff": dq, "eob": eob, "sad": sad, "recon": recon}

    def encode_recon_planes(self, src, src_stride, pred, pred_stride, recon, recon_stride, xy, tx_size, tx_type, qrow, iscan,
                            keep_coeff=False, want_sad=False, bd=8):
        """The encode-pass chain on uint8 (bd 8) or int16-as-uint16 (bd 10) planes: xy = int32 tensor of (y << 16) | x
        block origins; recon (may be pred itself) is written in place.  -> dict(coeff, qcoeff, dqcoeff, eob, sad)"""
        t = self.torch
        n = xy.shape[0]
        is16 = 0 if src.dtype == t.uint8 else 1
        nc = min(TX_W[tx_size], 32) * min(TX_H[tx_size], 32)
        mk = lambda: t.empty((n, nc), dtype=t.int32, device=src.device)
        co, dq = (mk(), mk()) if keep_coeff else (None, None)
        q = mk()
        eob = t.zeros(n, dtype=t.int16, device=src.device)
        sad = t.zeros(n, dtype=t.int32, device=src.device) if want_sad else None
        tabs = [_np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        self._check(self.lib.svt_hip_encode_recon_planes_batch(self._p(src), src_stride, self._p(pred), pred_stride, self._p(recon),
                                                                recon_stride, self._p(xy), n, is16, bd, tx_size, tx_type,
                                                                tabs[0].ctypes.data, tabs[1].ctypes.data, tabs[2].ctypes.data,
                                                                tabs[3].ctypes.data, tabs[4].ctypes.data, self._p(iscan),
                                                                self._p(co) if keep_coeff else None, self._p(q),
                                                                self._p(dq) if keep_coeff else None, self._p(eob),
                                                                self._p(sad) if want_sad else None, self._stream()),
                    "svt_hip_encode_recon_planes_batch")
        return {"coeff": co, "qcoeff": q, "dqcoeff": dq, "eob": eob, "sad": sad}

    # -- K4 / K7 / K8 ---------------------------------------------------------------
    def sad(self, a, b):
        """a, b: uint8 [n, H, W] dense -> int32 [n] (uint32 values)"""
        t = self.torch
        n, h, w = a.shape
        out = t.zeros(n, dtype=t.int32, device=a.device)
        self._check(self.lib.svt_hip_sad_batch(self._p(a), w, w * h, self._p(b), w, w * h, w, h, self._p(out), n,
                                                self._stream()), "svt_hip_sad_batch")
        return out

    def sse(self, a, b):
        t = self.torch
        n, h, w = a.shape
        out = t.zeros(n, dtype=t.int64, device=a.device)
        self._check(self.lib.svt_hip_sse_batch(self._p(a), w, w * h, self._p(b), w, w * h, w, h, self._p(out), n,
                                                self._stream()), "svt_hip_sse_batch")
        return out

    def residual(self, src, pred, out=None):
        t = self.torch
        n, h, w = src.shape
        if out is None:
            out = t.empty((n, h, w), dtype=t.int16, device=src.device)
        self._check(self.lib.svt_hip_residual_batch(self._p(src), w, w * h, self._p(pred), w, w * h, self._p(out), w,
                                                     w * h, w, h, n, self._stream()), "svt_hip_residual_batch")
        return out

    # -- K5 -----------------------------------------------------------------------
    def sad_search(self, src, ref, search_w, search_h, ref_stride=None, ref_stride_raw=None):
        """src uint8 [n, H, W]; ref uint8 [n, RH, RW] private windows. -> best_sad int64, x int16, y int16"""
        t = self.torch
        n, h, w = src.shape
        _, rh, rw = ref.shape
        rs = rw if ref_stride is None else ref_stride
        rraw = rw if ref_stride_raw is None else ref_stride_raw
        best = t.zeros(n, dtype=t.int64, device=src.device)
        x = t.zeros(n, dtype=t.int16, device=src.device)
        y = t.zeros(n, dtype=t.int16, device=src.device)
        self._check(self.lib.svt_hip_sad_search_batch(self._p(src), w, w * h, self._p(ref), rs, rraw, rw * rh, w, h,
                                                       search_w, search_h, self._p(best), self._p(x), self._p(y), n,
                                                       self._stream()), "svt_hip_sad_search_batch")
        return best, x, y

    def sad_search_planes(self, src_plane, src_stride, src_offsets, ref_plane, ref_stride, ref_offsets, width, height,
                          search_w, search_h, ref_stride_raw=None):
        """Frame-level form (HME): uint8 planes + int32 (uint32 values) per-block byte offsets of the source
        block and of the search window origin. -> best_sad int64, x int16, y int16"""
        t = self.torch
        n = src_offsets.shape[0]
        best = t.zeros(n, dtype=t.int64, device=src_plane.device)
        x = t.zeros(n, dtype=t.int16, device=src_plane.device)
        y = t.zeros(n, dtype=t.int16, device=src_plane.device)
        self._check(self.lib.svt_hip_sad_search_planes_batch(self._p(src_plane), src_stride, self._p(src_offsets),
                                                              self._p(ref_plane), ref_stride,
                                                              ref_stride if ref_stride_raw is None else ref_stride_raw,
                                                              self._p(ref_offsets), width, height, search_w, search_h,
                                                              self._p(best), self._p(x), self._p(y), n, self._stream()),
                    "svt_hip_sad_search_planes_batch")
        return best, x, y

    def me_sb_search_planes(self, src_plane, src_stride, src_offsets, ref_plane, ref_stride, ref_offsets, search_w, search_h,
                            origins=None, x_origin=0, y_origin=0, best_sad=None, best_mv=None):
        """Frame-level K6: all SBs (x reference pictures) of a segment in one launch, addressed by byte offsets."""
        t = self.torch
        n = src_offsets.shape[0]
        if best_sad is None:
            best_sad = t.full((n, 85), self.MAX_SAD_VALUE, dtype=t.int32, device=src_plane.device)
            best_mv = t.zeros((n, 85), dtype=t.int32, device=src_plane.device)
        self._check(self.lib.svt_hip_me_sb_search_planes_batch(self._p(src_plane), src_stride, self._p(src_offsets),
                                                                self._p(ref_plane), ref_stride, self._p(ref_offsets),
                                                                search_w, search_h,
                                                                self._p(origins) if origins is not None else None,
                                                                x_origin, y_origin, self._p(best_sad), self._p(best_mv), n,
                                                                self._stream()), "svt_hip_me_sb_search_planes_batch")
        return best_sad, best_mv

    # -- K7 coefficient domain ---------------------------------------------------------
    def full_distortion32(self, coeff, recon, width, height, cbf_zero=False):
        """coeff, recon: int32 [n, height, width] dense -> int64 [n, 2] (uint64 values)"""
        t = self.torch
        n = coeff.shape[0]
        out = t.zeros((n, 2), dtype=t.int64, device=coeff.device)
        self._check(self.lib.svt_hip_full_distortion32_batch(self._p(coeff), width, width * height,
                                                              self._p(recon) if recon is not None else None, width,
                                                              width * height, width, height, 1 if cbf_zero else 0,
                                                              self._p(out), n, self._stream()),
                    "svt_hip_full_distortion32_batch")
        return out

    # -- open-loop intra search (open_loop_intra_search_sb) --------------------------------
    @staticmethod
    def ois_candidates(bsize, temporal_layer_index=0, intra_pred_mode=0, is_used_as_reference=True, is_16bit=False):
        """The candidate list the reference's loop enumerates for one block size (EbMotionEstimation.c:8747-8846,
        is_16bit = encoder_bit_depth > 8: PAETH_PRED is left out, the search itself stays on the 8-bit picture):
        (modes uint8[], angle_deltas int8[]) in AV1 PredictionMode numbering."""
        import numpy as np
        last = 11 if is_16bit else 12
        nd = 1 if intra_pred_mode >= 5 else (5 if bsize >= 8 else 1)
        no_angular = temporal_layer_index > 0 or bsize > 16
        if no_angular:
            nd = 1
        if not is_used_as_reference and intra_pred_mode >= 4:
            last = 0
        modes, deltas = [], []
        for m in range(last + 1):
            if 1 <= m <= 8:
                if no_angular:
                    continue
                for k in range(nd):
                    modes.append(m); deltas.append(0 if nd == 1 else k - (nd >> 1))
            else:
                modes.append(m); deltas.append(0)
        return np.array(modes, np.uint8), np.array(deltas, np.int8)

    def ois_search(self, pic, stride, width, height, xy, bsize, modes, angle_deltas):
        """pic: uint8 tensor whose data_ptr() is picture sample (0, 0) (a view into the padded plane is fine);
        xy: int32 [n] (x | y << 16).  -> (distortion int32 [n, ncand], best_index int8 [n])"""
        import numpy as np
        t = self.torch
        n = xy.shape[0]
        modes = np.ascontiguousarray(modes, np.uint8); angle_deltas = np.ascontiguousarray(angle_deltas, np.int8)
        nc = int(modes.shape[0])
        dist = t.empty((n, nc), dtype=t.int32, device=xy.device)        # (the search writes every entry)
        best = t.empty(n, dtype=t.int8, device=xy.device)
        wb = self.lib.svt_hip_ois_work_bytes(bsize, nc, n)
        work = t.empty(max(wb, 1), dtype=t.uint8, device=xy.device)
        self._check(self.lib.svt_hip_ois_search_batch(pic.data_ptr(), stride, width, height, self._p(xy), bsize,
                                                       modes.ctypes.data, angle_deltas.ctypes.data, nc, self._p(dist),
                                                       self._p(best), self._p(work), wb, n, self._stream()),
                    "svt_hip_ois_search_batch")
        return dist, best

    class OisGroup(ctypes.Structure):
        """svt_hip_ois_group"""
        _fields_ = [("d_xy", ctypes.c_void_p), ("bsize", ctypes.c_uint32), ("modes", ctypes.c_void_p), ("angle_deltas", ctypes.c_void_p),
                    ("ncand", ctypes.c_int32), ("d_distortion", ctypes.c_void_p), ("d_best_index", ctypes.c_void_p), ("d_work", ctypes.c_void_p),
                    ("work_bytes", ctypes.c_size_t), ("nblocks", ctypes.c_size_t)]

    def ois_search_frame(self, pic, stride, width, height, groups):
        """The open-loop intra search of a picture, every block size in ONE call (svt_hip_ois_search_frame).  groups: list of
        (xy int32 [n], bsize, modes, angle_deltas) -> list of (distortion int32 [n, ncand], best_index int8 [n])"""
        import numpy as np
        t = self.torch
        arr = (self.OisGroup * len(groups))()
        keep, outs = [], []
        for i, (xy, bsize, modes, deltas) in enumerate(groups):
            n = xy.shape[0]
            modes = np.ascontiguousarray(modes, np.uint8); deltas = np.ascontiguousarray(deltas, np.int8)
            nc = int(modes.shape[0])
            dist = t.empty((n, nc), dtype=t.int32, device=xy.device)        # (the search writes every entry)
            best = t.empty(n, dtype=t.int8, device=xy.device)
            wb = self.lib.svt_hip_ois_work_bytes(bsize, nc, n)
            work = t.empty(max(wb, 1), dtype=t.uint8, device=xy.device)
            arr[i] = self.OisGroup(self._p(xy), bsize, modes.ctypes.data, deltas.ctypes.data, nc, self._p(dist), self._p(best), self._p(work), wb, n)
            keep.append((xy, modes, deltas, work))
            outs.append((dist, best))
        self._check(self.lib.svt_hip_ois_search_frame(pic.data_ptr(), stride, width, height, arr, len(groups), self._stream()),
                    "svt_hip_ois_search_frame")
        self._ois_keep = keep          # host lists and work buffers outlive the enqueued kernels
        return outs

    # -- K11 chroma from luma + level map ---------------------------------------------------
    CFL_BUF_LINE = 32

    def cfl_luma_subsampling_420(self, luma, luma_stride, width, height, xy=None, luma_block_pitch=0, n=None,
                                 subtract_average=False, q3=None):
        """luma: uint8 / int16(uint16 values) plane or dense blocks; width x height = LUMA block.
        -> int16 [n, 32, 32] Q3 buffers in the reference's layout (rows CFL_BUF_LINE apart)"""
        t = self.torch
        if n is None:
            n = xy.shape[0]
        if q3 is None:
            q3 = t.zeros((n, 32, 32), dtype=t.int16, device=luma.device)
        self._check(self.lib.svt_hip_cfl_luma_subsampling_420_batch(self._p(luma), luma_stride, luma_block_pitch,
                                                                     self._p(xy) if xy is not None else None,
                                                                     0 if luma.dtype == t.uint8 else 1, self._p(q3), 32, 1024,
                                                                     width, height, 1 if subtract_average else 0, n,
                                                                     self._stream()), "svt_hip_cfl_luma_subsampling_420_batch")
        return q3

    def subtract_average(self, q3, width, height, round_offset, num_pel_log2):
        n = q3.shape[0]
        self._check(self.lib.svt_hip_subtract_average_batch(self._p(q3), q3.shape[2], q3.shape[1] * q3.shape[2], width, height,
                                                             round_offset, num_pel_log2, n, self._stream()),
                    "svt_hip_subtract_average_batch")
        return q3

    def cfl_predict(self, ac_q3, pred, pred_stride, dst, dst_stride, alpha_q3, bd, width, height, xy=None):
        t = self.torch
        n = ac_q3.shape[0]
        self._check(self.lib.svt_hip_cfl_predict_batch(self._p(ac_q3), ac_q3.shape[2], ac_q3.shape[1] * ac_q3.shape[2],
                                                        self._p(pred), pred_stride, self._p(dst), dst_stride,
                                                        self._p(xy) if xy is not None else None, self._p(alpha_q3), bd, width,
                                                        height, 0 if pred.dtype == t.uint8 else 1, n, self._stream()),
                    "svt_hip_cfl_predict_batch")
        return dst

    def txb_init_levels(self, coeff, width, height, levels_buf=None):
        """coeff int32 [n, height*width] -> uint8 [n, pitch] whole padded level buffers"""
        t = self.torch
        n = coeff.shape[0]
        size = (width + 4) * (height + 6) + 16
        if levels_buf is None:
            levels_buf = t.empty((n, size), dtype=t.uint8, device=coeff.device)
        self._check(self.lib.svt_hip_txb_init_levels_batch(self._p(coeff), coeff.shape[1] if coeff.dim() == 2 else width * height,
                                                            self._p(levels_buf), levels_buf.shape[1], width, height, n,
                                                            self._stream()), "svt_hip_txb_init_levels_batch")
        return levels_buf

    # -- K9 / K10 ------------------------------------------------------------------------
    NB_ORIGIN = 16

    def intra_pred(self, above, left, mode, bw, bh, bd=8, up_above=0, up_left=0, dx=1, dy=1, out=None):
        """above, left: [n, nb_pitch] uint8 or int16(as uint16) neighbour rows (position p at NB_ORIGIN + p).
        -> [n, bh, bw] prediction"""
        t = self.torch
        n, pitch = above.shape
        is16 = 0 if above.dtype == t.uint8 else 1
        if out is None:
            out = t.empty((n, bh, bw), dtype=above.dtype, device=above.device)
        self._check(self.lib.svt_hip_intra_pred_batch(self._p(out), bw, bw * bh, None, self._p(above), self._p(left), pitch,
                                                       mode, bw, bh, up_above, up_left, dx, dy, is16, bd, n,
                                                       self._stream()), "svt_hip_intra_pred_batch")
        return out

    def filter_intra_edge(self, edges, sz, strength):
        t = self.torch
        n, pitch = edges.shape
        self._check(self.lib.svt_hip_filter_intra_edge_batch(self._p(edges), pitch, sz, strength,
                                                              0 if edges.dtype == t.uint8 else 1, n, self._stream()),
                    "svt_hip_filter_intra_edge_batch")
        return edges

    def upsample_intra_edge(self, edges, sz, bd=8):
        t = self.torch
        n, pitch = edges.shape
        self._check(self.lib.svt_hip_upsample_intra_edge_batch(self._p(edges), pitch, sz,
                                                                0 if edges.dtype == t.uint8 else 1, bd, n,
                                                                self._stream()), "svt_hip_upsample_intra_edge_batch")
        return edges

    # -- K6 ---------------------------------------------------------------------------------
    MAX_SAD_VALUE = 128 * 128 * 255     # EbMotionEstimation.h:79

    def me_sb_search(self, src, ref, search_w, search_h, x_origin=0, y_origin=0, origins=None, best_sad=None, best_mv=None):
        """src uint8 [n,64,64]; ref uint8 [n, 64+sh-1 (or more), RW] private windows.
        -> (best_sad, best_mv) int32 [n,85] (uint32 values), updated in place when given."""
        t = self.torch
        n = src.shape[0]
        _, rh, rw = ref.shape
        if best_sad is None:
            best_sad = t.full((n, 85), self.MAX_SAD_VALUE, dtype=t.int32, device=src.device)
            best_mv = t.zeros((n, 85), dtype=t.int32, device=src.device)
        self._check(self.lib.svt_hip_me_sb_search_batch(self._p(src), 64, 64 * 64, self._p(ref), rw, rw * rh, search_w,
                                                         search_h, self._p(origins) if origins is not None else None,
                                                         x_origin, y_origin, self._p(best_sad), self._p(best_mv), n,
                                                         self._stream()), "svt_hip_me_sb_search_batch")
        return best_sad, best_mv

    # -- frame-level encode pass: all (plane, size) groups of a frame in one call --------------------
    class FrameGroup(ctypes.Structure):
        _fields_ = [("d_src", c_void_p), ("src_stride", c_uint32), ("d_pred", c_void_p), ("pred_stride", c_uint32),
                    ("d_recon", c_void_p), ("recon_stride", c_uint32), ("d_xy", c_void_p), ("d_offsets", c_void_p),
                    ("nblocks", c_uint32), ("tx_size", ctypes.c_int32), ("tx_type", ctypes.c_int32), ("d_iscan", c_void_p),
                    ("d_qcoeff", c_void_p), ("d_eob", c_void_p), ("d_coeff", c_void_p), ("d_dqcoeff", c_void_p)]

    def make_frame_groups(self, groups):
        """groups: list of dicts with tensors src, pred, recon (planes), xy, iscan, qcoeff, eob and optional offsets, coeff,
        dqcoeff, plus src_stride / pred_stride / recon_stride, tx_size, tx_type.  -> ctypes array (keep the tensors alive!)"""
        arr = (self.FrameGroup * len(groups))()
        for i, g in enumerate(groups):
            P = lambda k: self._p(g[k]) if g.get(k) is not None else None
            arr[i] = self.FrameGroup(P("src"), g["src_stride"], P("pred"), g["pred_stride"], P("recon"), g["recon_stride"], P("xy"),
                                     P("offsets"), g["xy"].numel(), g["tx_size"], g["tx_type"], P("iscan"), P("qcoeff"), P("eob"),
                                     P("coeff"), P("dqcoeff"))
        return arr

    def encode_recon_frame(self, group_array, qrow, is_16bit=False, bd=8):
        tabs = [_np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        self._check(self.lib.svt_hip_encode_recon_frame(group_array, len(group_array), 1 if is_16bit else 0, bd, tabs[0].ctypes.data,
                                                        tabs[1].ctypes.data, tabs[2].ctypes.data, tabs[3].ctypes.data, tabs[4].ctypes.data,
                                                        self._stream()), "svt_hip_encode_recon_frame")

    class FrameCflGroup(ctypes.Structure):
        _fields_ = [("d_luma_recon", c_void_p), ("luma_stride", c_uint32), ("d_pred_cb", c_void_p), ("pred_stride_cb", c_uint32),
                    ("d_pred_cr", c_void_p), ("pred_stride_cr", c_uint32), ("d_xy", c_void_p), ("d_alpha_q3_cb", c_void_p),
                    ("d_alpha_q3_cr", c_void_p), ("width", c_uint32), ("height", c_uint32), ("nblocks", c_uint32)]

    class FrameLevels(ctypes.Structure):
        _fields_ = [("d_levels_buf", c_void_p), ("levels_block_pitch", ctypes.c_size_t)]

    def make_frame_cfl_groups(self, groups):
        """groups: list of dicts with tensors luma_recon, pred_cb, pred_cr (planes), xy, alpha_cb, alpha_cr (int32 per block), the
        strides luma_stride / cb_stride / cr_stride and the chroma block's width, height"""
        arr = (self.FrameCflGroup * max(len(groups), 1))()
        for i, g in enumerate(groups):
            arr[i] = self.FrameCflGroup(self._p(g["luma_recon"]), g["luma_stride"], self._p(g["pred_cb"]), g["cb_stride"], self._p(g["pred_cr"]),
                                        g["cr_stride"], self._p(g["xy"]), self._p(g["alpha_cb"]), self._p(g["alpha_cr"]), g["width"], g["height"],
                                        g["xy"].numel())
        return arr

    def make_frame_levels(self, level_bufs):
        """level_bufs: one uint8 [nblocks, pitch] tensor (or None) per group of the call"""
        arr = (self.FrameLevels * max(len(level_bufs), 1))()
        for i, b in enumerate(level_bufs):
            arr[i] = self.FrameLevels(self._p(b) if b is not None else None, b.shape[1] if b is not None else 0)
        return arr

    def encode_recon_frame_ex(self, group_array, qrow, first_chroma_group=0, cfl_array=None, ncfl=0, levels_array=None, is_16bit=False, bd=8):
        """svt_hip_encode_recon_frame_ex: groups [0, first_chroma_group) -> chroma-from-luma prediction of the cfl groups -> the other
        groups -> av1_txb_init_levels of every group with a level buffer, all enqueued on the current stream"""
        tabs = [_np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        self._check(self.lib.svt_hip_encode_recon_frame_ex(group_array, len(group_array), first_chroma_group, cfl_array, ncfl, levels_array,
                                                           1 if is_16bit else 0, bd, tabs[0].ctypes.data, tabs[1].ctypes.data,
                                                           tabs[2].ctypes.data, tabs[3].ctypes.data, tabs[4].ctypes.data, self._stream()),
                    "svt_hip_encode_recon_frame_ex")

    # -- hierarchical ME: one level for all SBs, clipping on the device ----------------------------
    class HmeParams(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in ("search_area_width", "search_area_height", "x_origin_offset", "y_origin_offset",
                                                  "pad_width", "pad_height", "ref_width", "ref_height", "round_down", "mv_shift")]

    def hme_level_params(self, level, hme_w, hme_h, region_w, region_h, total_w, total_h, mult_x, mult_y, ref_origin_x, ref_origin_y,
                         ref_width, ref_height):
        p = self.HmeParams()
        w, h = _np16(hme_w).view("uint16"), _np16(hme_h).view("uint16")
        self._check(self.lib.svt_hip_hme_level_params(level, w.ctypes.data, h.ctypes.data, region_w, region_h, total_w, total_h, mult_x,
                                                      mult_y, ref_origin_x, ref_origin_y, ref_width, ref_height, ctypes.byref(p)),
                    "svt_hip_hme_level_params")
        return p

    def hme_level(self, src_pic, src_stride, ref_pic00, ref_stride, sb_origin, sb_size, centers, center_shift, params):
        """src_pic: uint8 tensor whose data_ptr() is sample (0, 0) of the level's source picture; ref_pic00: likewise for the
        padded reference (a view into the padded buffer).  sb_origin int16 [n, 2], sb_size int16 [n, 2] (uint16 values),
        centers int16 [n, 2] or None.  -> (best_sad int64 [n], mv int16 [n, 2])"""
        t = self.torch
        n = sb_origin.shape[0]
        best = t.empty(n, dtype=t.int64, device=sb_origin.device)         # (the kernel writes every task's entries)
        mv = t.empty((n, 2), dtype=t.int16, device=sb_origin.device)
        self._check(self.lib.svt_hip_hme_level_batch(self._p(src_pic), src_stride, self._p(ref_pic00), ref_stride, self._p(sb_origin),
                                                     self._p(sb_size), self._p(centers) if centers is not None else None, center_shift,
                                                     ctypes.byref(params), self._p(best), self._p(mv), n, self._stream()),
                    "svt_hip_hme_level_batch")
        return best, mv

    # ---- picture input ----
    def picture_import(self, frame, width, height, planes, origin_x, origin_y, pad_right=0, pad_bottom=0, ss_x=1, ss_y=1):
        """frame: 1-D uint8 / int16 (uint16 values) device tensor holding Y, Cb, Cr back to back as a y4m frame does; planes: the three
        padded plane buffers (2-D tensors [rows, stride]; chroma may be None).  One launch: copy + right / bottom extension + borders
        (svt_hip_picture_import)."""
        t = self.torch
        is16 = frame.dtype != t.uint8
        y, cb, cr = planes
        self._check(self.lib.svt_hip_picture_import(self._p(frame), width, height, ss_x, ss_y, int(is16), self._p(y), y.stride(0),
                                                    self._p(cb) if cb is not None else None, cb.stride(0) if cb is not None else 0,
                                                    self._p(cr) if cr is not None else None, cr.stride(0) if cr is not None else 0,
                                                    origin_x, origin_y, pad_right, pad_bottom, self._stream()), "svt_hip_picture_import")

    def picture_pad(self, buf, width, height, pad_w, pad_h):
        """generate_padding{,16_bit} in place on a 2-D buffer tensor [height + 2 pad_h, stride] (svt_hip_picture_pad)"""
        self._check(self.lib.svt_hip_picture_pad(self._p(buf), buf.stride(0), width, height, pad_w, pad_h, int(buf.dtype != self.torch.uint8),
                                                 self._stream()), "svt_hip_picture_pad")

    def picture_luma8(self, plane16, out8, cols, rows, bd=10):
        """the 8-bit plane (v >> (bd - 8)) of a 16-bit padded plane buffer, buffer to buffer (svt_hip_picture_luma8)"""
        self._check(self.lib.svt_hip_picture_luma8(self._p(plane16), plane16.stride(0), self._p(out8), out8.stride(0), cols, rows, bd, self._stream()),
                    "svt_hip_picture_luma8")

    def picture_decimate(self, luma_origin, luma_stride, width, height, quarter=None, q_origin=(0, 0), sixteenth=None, s_origin=(0, 0)):
        """DecimateInputPicture: luma_origin = tensor view whose data_ptr() is the luma picture's origin sample; quarter / sixteenth:
        2-D padded buffers or None (svt_hip_picture_decimate)"""
        self._check(self.lib.svt_hip_picture_decimate(self._p(luma_origin), luma_stride, width, height,
                                                      self._p(quarter) if quarter is not None else None,
                                                      quarter.stride(0) if quarter is not None else 0, q_origin[0], q_origin[1],
                                                      self._p(sixteenth) if sixteenth is not None else None,
                                                      sixteenth.stride(0) if sixteenth is not None else 0, s_origin[0], s_origin[1],
                                                      self._stream()), "svt_hip_picture_decimate")

    def hme_level_regions(self, src_pic, src_stride, ref_pic00, ref_stride, sb_origin, sb_size, centers, center_shift, params_list):
        """hme_level for 1 .. 4 search regions in one launch (svt_hip_hme_level_regions_batch).  centers: int16 [regions, n, 2] or
        None.  -> (best_sad int64 [regions, n], mv int16 [regions, n, 2])"""
        t = self.torch
        n, nr = sb_origin.shape[0], len(params_list)
        arr = (self.HmeParams * nr)(*params_list)
        best = t.empty((nr, n), dtype=t.int64, device=sb_origin.device)
        mv = t.empty((nr, n, 2), dtype=t.int16, device=sb_origin.device)
        self._check(self.lib.svt_hip_hme_level_regions_batch(self._p(src_pic), src_stride, self._p(ref_pic00), ref_stride, self._p(sb_origin),
                                                             self._p(sb_size), self._p(centers) if centers is not None else None, center_shift,
                                                             ctypes.addressof(arr), nr, self._p(best), self._p(mv), n, self._stream()),
                    "svt_hip_hme_level_regions_batch")
        return best, mv

    class IntraPos(ctypes.Structure):
        """svt_hip_intra_pos: where one prediction block sits (the arguments of av1_predict_intra_block, EbIntraPrediction.c:4078)"""
        _fields_ = [(n, ctypes.c_int32) for n in ("is_16bit", "sb_size_mi", "mi_rows", "mi_cols", "tile_mi_row_start", "tile_mi_row_end",
                                                  "tile_mi_col_start", "tile_mi_col_end", "partition", "bsize", "tx_size", "plane",
                                                  "bl_org_x_pict", "bl_org_y_pict", "col_off", "row_off", "wpx", "hpx")]

    class IntraBlk(ctypes.Structure):
        """svt_hip_intra_blk: per-block descriptor of svt_hip_build_intra_predictors_batch"""
        _fields_ = [("mode", ctypes.c_uint8), ("angle_delta", ctypes.c_int8), ("filt_type", ctypes.c_uint8),
                    ("disable_edge_filter", ctypes.c_uint8), ("n_top_px", ctypes.c_uint8), ("n_topright_px", ctypes.c_uint8),
                    ("n_left_px", ctypes.c_uint8), ("n_bottomleft_px", ctypes.c_uint8)]

    def intra_neighbor_px(self, pos):
        """HOST: (n_top_px, n_topright_px, n_left_px, n_bottomleft_px) of one prediction block (svt_hip_intra_neighbor_px)"""
        blk = self.IntraBlk()
        self._check(self.lib.svt_hip_intra_neighbor_px(ctypes.addressof(pos), ctypes.addressof(blk)), "svt_hip_intra_neighbor_px")
        return blk.n_top_px, blk.n_topright_px, blk.n_left_px, blk.n_bottomleft_px

    def build_intra_predictors(self, top_neigh, left_neigh, blocks, tx_size, bd=8, dst=None, dst_stride=None, dst_offsets=None, order=None):
        """build_intra_predictors{,_high} on a batch (svt_hip_build_intra_predictors_batch).  top_neigh / left_neigh: uint8 or int16
        (uint16 values) [n, pitch] with element 0 = the corner sample; blocks: uint8 [n, 8] descriptors (IntraBlk layout).
        -> dst [n, h, w] (dense) unless dst / dst_offsets address a picture."""
        t = self.torch
        n = top_neigh.shape[0]
        is16 = top_neigh.dtype != t.uint8
        w, h = (TX_W[tx_size], TX_H[tx_size]) if 0 <= tx_size < 19 else (4, 4)        # the library reports a bad tx_size
        if dst is None:
            dst = t.empty((n, h, w), dtype=top_neigh.dtype, device=top_neigh.device)
            dst_stride, pitch = w, w * h
        else:
            pitch = 0 if dst_offsets is not None else h * dst_stride
        if order is not None:
            self.lib.svt_hip_build_intra_predictors_ordered_batch.argtypes = [c_void_p, c_int32, c_size_t] + [c_void_p] * 3 + [c_int32, c_void_p, c_void_p,
                                                                              c_int, c_int, c_int, c_size_t, c_void_p]
            self._check(self.lib.svt_hip_build_intra_predictors_ordered_batch(self._p(dst), dst_stride, pitch, self._p(dst_offsets) if dst_offsets is not None else None,
                                                                              self._p(top_neigh), self._p(left_neigh), top_neigh.shape[1], self._p(blocks),
                                                                              self._p(order), tx_size, int(is16), bd, n, self._stream()),
                        "svt_hip_build_intra_predictors_ordered_batch")
            return dst
        self._check(self.lib.svt_hip_build_intra_predictors_batch(self._p(dst), dst_stride, pitch, self._p(dst_offsets) if dst_offsets is not None else None,
                                                                  self._p(top_neigh), self._p(left_neigh), top_neigh.shape[1], self._p(blocks),
                                                                  tx_size, int(is16), bd, n, self._stream()),
                    "svt_hip_build_intra_predictors_batch")
        return dst

    def intra_order_blocks(self, blocks, tx_size):
        """svt_hip_intra_order_blocks_batch: the batch's block indices grouped by predictor kind inside tiles of 4 096 blocks (device)
        -> int32 [n]"""
        t = self.torch
        n = blocks.shape[0]
        order = t.empty(n, dtype=t.int32, device=blocks.device)
        self.lib.svt_hip_intra_order_blocks_batch.argtypes = [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p]
        work = getattr(self, "_order_work", None)             # unused by the library since round 3 (older builds: 32 counters)
        if work is None or work.device != blocks.device:
            work = self._order_work = t.empty(32, dtype=t.int32, device=blocks.device)
        self._check(self.lib.svt_hip_intra_order_blocks_batch(self._p(blocks), tx_size, n, self._p(order), self._p(work), self._stream()),
                    "svt_hip_intra_order_blocks_batch")
        return order

    ME_PUS_ALL = 209
    FLAVOUR_C, FLAVOUR_AVX2 = 0, 1

    def me_fullpel_search(self, src, ref, search_w, search_h, x_origin=0, y_origin=0, origins=None, flavour=0, nsq=False,
                          best_sad=None, best_mv=None, src_stride=64, src_offsets=None, ref_stride=None, ref_offsets=None,
                          n=None):
        """K6 in the reference's p_sb_best_sad / p_sb_best_mv layout (svt_hip_me_fullpel_search_batch).  Dense form: src uint8
        [n, 64, 64], ref uint8 [n, rows, RW] private windows.  Plane form: src / ref are planes, *_offsets int32 byte offsets.
        -> (best_sad, best_mv) int32 [n, 85 or 209] (uint32 values), updated in place when given."""
        t = self.torch
        if src_offsets is None:
            n = src.shape[0]
            _, rh, rw = ref.shape
            ref_stride, spitch, rpitch = rw, 64 * 64, rw * rh
        else:
            n = n if n is not None else src_offsets.numel()
            spitch = rpitch = 0
        npu = self.ME_PUS_ALL if nsq else 85
        if best_sad is None:
            best_sad = t.full((n, npu), self.MAX_SAD_VALUE, dtype=t.int32, device=src.device)
            best_mv = t.zeros((n, npu), dtype=t.int32, device=src.device)
        self._check(self.lib.svt_hip_me_fullpel_search_batch(
            self._p(src), src_stride, spitch, self._p(src_offsets) if src_offsets is not None else None, self._p(ref), ref_stride,
            rpitch, self._p(ref_offsets) if ref_offsets is not None else None, search_w, search_h,
            self._p(origins) if origins is not None else None, x_origin, y_origin, flavour, 1 if nsq else 0, self._p(best_sad),
            self._p(best_mv), best_sad.shape[1], n, self._stream()), "svt_hip_me_fullpel_search_batch")
        return best_sad, best_mv

    # -- MotionEstimateLcu's glue: set-up, per-SB areas, bi-prediction + result rows ------------------------
    class MeSetupParams(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in ("picture_width", "picture_height", "ref_width", "ref_height", "search_area_width",
                                                  "search_area_height", "regions_w", "regions_h", "second_best", "zz_check")]

    class MeResult(ctypes.Structure):
        _fields_ = [("x_mv_l0", ctypes.c_int16), ("y_mv_l0", ctypes.c_int16), ("x_mv_l1", ctypes.c_int16), ("y_mv_l1", ctypes.c_int16),
                    ("distortion", ctypes.c_uint32 * 3), ("direction", ctypes.c_uint8 * 3), ("total_me_candidate_index", ctypes.c_uint8)]

    def me_setup(self, src_pic00, src_stride, ref_pic00, ref_stride, sb_origin, sb_size, hme_sad, hme_mv, params):
        """svt_hip_me_setup_batch: search centre (best HME region, CheckZeroZeroCenter) and clipped search area per task.
        src_pic00 / ref_pic00: uint8 views whose data_ptr() is sample (0, 0) of the padded planes; sb_origin / sb_size int16 [n, 2];
        hme_sad int64 [regions, n] and hme_mv int16 [regions, n, 2] (or None).  -> (center int16 [n, 2], area int16 [n, 4])"""
        t = self.torch
        n = sb_origin.shape[0]
        center = t.empty((n, 2), dtype=t.int16, device=sb_origin.device)      # (every entry is written by the kernel)
        area = t.empty((n, 4), dtype=t.int16, device=sb_origin.device)
        self._check(self.lib.svt_hip_me_setup_batch(self._p(src_pic00), src_stride, self._p(ref_pic00), ref_stride, self._p(sb_origin),
                                                    self._p(sb_size), self._p(hme_sad) if hme_sad is not None else None,
                                                    self._p(hme_mv) if hme_mv is not None else None, ctypes.byref(params), self._p(center),
                                                    self._p(area), n, self._stream()), "svt_hip_me_setup_batch")
        return center, area

    def me_fullpel_search_areas(self, src, src_stride, src_offsets, ref, ref_stride, ref_offsets, areas, max_w, max_h, flavour=0, nsq=False,
                                best_sad=None, best_mv=None):
        """svt_hip_me_fullpel_search_areas_batch: one search area per SB, read on the device from `areas` (int16 [n, 4]).
        ref_offsets: the SB's co-located byte offset in the reference plane.  -> (best_sad, best_mv) int32 [n, 85 | 209]"""
        t = self.torch
        n = src_offsets.numel()
        npu = self.ME_PUS_ALL if nsq else 85
        if best_sad is None:
            best_sad = t.full((n, npu), self.MAX_SAD_VALUE, dtype=t.int32, device=src.device)
            best_mv = t.zeros((n, npu), dtype=t.int32, device=src.device)
        self._check(self.lib.svt_hip_me_fullpel_search_areas_batch(self._p(src), src_stride, self._p(src_offsets), self._p(ref), ref_stride,
                                                                   self._p(ref_offsets), self._p(areas), max_w, max_h, flavour, 1 if nsq else 0,
                                                                   self._p(best_sad), self._p(best_mv), best_sad.shape[1], n, self._stream()),
                    "svt_hip_me_fullpel_search_areas_batch")
        return best_sad, best_mv

    def me_bipred(self, src_pic00, src_stride, ref0_pic00, ref0_stride, ref1_pic00, ref1_stride, sb_origin, best_sad0, best_mv0, best_sad1=None,
                  best_mv1=None, npus=209, bipred_all_pus=True, sub_sad=True):
        """svt_hip_me_bipred_batch -> (bipred_sad int32 [n, pu_pitch] in storage order, results uint8 [n, npus, 24] = svt_hip_me_result
        rows in raster PU order)"""
        t = self.torch
        n = sb_origin.shape[0]
        pitch = best_sad0.shape[1]
        bip = t.zeros((n, pitch), dtype=t.int32, device=sb_origin.device)
        res = t.empty((n, npus, ctypes.sizeof(self.MeResult)), dtype=t.uint8, device=sb_origin.device)      # (every row is written)
        two = best_sad1 is not None
        self._check(self.lib.svt_hip_me_bipred_batch(self._p(src_pic00), src_stride, self._p(ref0_pic00) if two else None, ref0_stride,
                                                     self._p(ref1_pic00) if two else None, ref1_stride, self._p(sb_origin), self._p(best_sad0),
                                                     self._p(best_mv0), self._p(best_sad1) if two else None, self._p(best_mv1) if two else None,
                                                     pitch, npus, int(bipred_all_pus), int(sub_sad), self._p(bip), self._p(res), n,
                                                     self._stream()), "svt_hip_me_bipred_batch")
        return bip, res

    @staticmethod
    def me_results_as_rows(res):
        """uint8 [n, npus, 24] svt_hip_me_result rows -> int64 numpy [n, npus, 11] in the column order of oracle/ref_me.c's results:
        xMvL0 yMvL0 xMvL1 yMvL1 dist0 dir0 dist1 dir1 dist2 dir2 totalMeCandidateIndex"""
        import numpy as np
        a = res.cpu().numpy()
        n, npus, _ = a.shape
        out = np.zeros((n, npus, 11), np.int64)
        out[..., 0:4] = a[..., 0:8].copy().view(np.int16).reshape(n, npus, 4)
        d = a[..., 8:20].copy().view(np.uint32).reshape(n, npus, 3)
        for k in range(3):
            out[..., 4 + 2 * k] = d[..., k]
            out[..., 5 + 2 * k] = a[..., 20 + k]
        out[..., 10] = a[..., 23]
        return out

    # -- general fused chain on planes ------------------------------------------------------
    def fwd_quant_planes(self, src, src_stride, pred, pred_stride, xy, tx_size, tx_type, qrow, iscan, bd=8,
                         want_sad=False, want_energy=False):
        """src, pred: uint8 / int16(as uint16) planes (any shape, row strides given in elements);
        xy: int32 tensor of (y << 16) | x block origins.  -> coeff, q, dq [n, KW*KH], eob, sad|None, energy|None"""
        t = self.torch
        n = xy.shape[0]
        nc = min(TX_W[tx_size], 32) * min(TX_H[tx_size], 32)
        is16 = 0 if src.dtype == t.uint8 else 1
        co = t.empty((n, nc), dtype=t.int32, device=src.device)
        q = t.empty_like(co); dq = t.empty_like(co)
        eob = t.zeros(n, dtype=t.int16, device=src.device)
        sad = t.zeros(n, dtype=t.int32, device=src.device) if want_sad else None
        en = t.zeros(n, dtype=t.int64, device=src.device) if want_energy else None
        tabs = [_np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        self._check(self.lib.svt_hip_fwd_quant_planes_batch(self._p(src), src_stride, self._p(pred), pred_stride,
                                                             self._p(xy), n, is16, bd, tx_size, tx_type,
                                                             tabs[0].ctypes.data, tabs[1].ctypes.data, tabs[2].ctypes.data,
                                                             tabs[3].ctypes.data, tabs[4].ctypes.data, self._p(iscan),
                                                             self._p(co), self._p(q), self._p(dq), self._p(eob),
                                                             self._p(sad) if want_sad else None,
                                                             self._p(en) if want_energy else None, self._stream()),
                    "svt_hip_fwd_quant_planes_batch")
        return co, q, dq, eob, sad, en

    # -- configs[1]: FwdTxfm2d + quantize on a residual batch ----------------------------------
    def fwd_quant(self, residual, tx_size, tx_type, qrow, iscan, bd=8, outs=None):
        """residual: int16 [n, H, W] -> coeff, qcoeff, dqcoeff (int32 [n, W*H]), eob"""
        t = self.torch
        n = residual.shape[0]
        nc = TX_W[tx_size] * TX_H[tx_size]
        if outs is None:
            outs = (t.empty((n, nc), dtype=t.int32, device=residual.device), t.empty((n, nc), dtype=t.int32, device=residual.device),
                    t.empty((n, nc), dtype=t.int32, device=residual.device), t.zeros(n, dtype=t.int16, device=residual.device))
        co, q, dq, eob = outs
        tabs = [_np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        self._check(self.lib.svt_hip_fwd_quant_batch(self._p(residual), n, tx_size, tx_type, bd, tabs[0].ctypes.data,
                                                      tabs[1].ctypes.data, tabs[2].ctypes.data, tabs[3].ctypes.data,
                                                      tabs[4].ctypes.data, self._p(iscan), self._p(co), self._p(q),
                                                      self._p(dq), self._p(eob), self._stream()), "svt_hip_fwd_quant_batch")
        return outs
