"""Host-side mirror of the reference's similarity-matrix interface, on top of the C-ABI.

``compute_similarity_matrix`` keeps the name, argument order, argument meaning and error
behaviour of the reference's ``computeSimilarityMatrix`` (reference: similarity_matrix.hpp:51-60);
``SimilarityMatrixPlan`` exposes the staged device-resident interface (prepare once, accumulate a
tile range, finalize) that bench.py and the multi-GPU driver use. All arithmetic happens in
libsecedo_simmat.so on the GPU; this module only moves arrays and pointers.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import InvalidNormalization, SecedoError  # noqa: F401  (re-exported)
from .pileup import FlatPileup, PosData, flatten

NORMALIZATIONS = ("ADD_MIN", "EXPONENTIATE", "SCALE_MAX_1")  # similarity_matrix.hpp:9-17


def to_enum(normalization: str) -> int:
    """reference: to_enum, similarity_matrix.cpp:256-266."""
    if not isinstance(normalization, str):
        raise InvalidNormalization("Invalid normalization: %r" % (normalization,))
    rc = _lib.lib().secedo_simmat_normalization_from_string(normalization.encode())
    if rc < 0:
        raise InvalidNormalization("Invalid normalization: " + normalization)
    return rc


def _as_flat(pos_data) -> FlatPileup:
    if isinstance(pos_data, FlatPileup):
        return pos_data
    return flatten(pos_data)


def _id_arrays(p: FlatPileup):
    """(id_base16, id_base32): the reference's u16 packing when every id fits 14 bits."""
    if p.n_entries == 0 or int(p.id_base.max()) <= 0xFFFF:
        return np.ascontiguousarray(p.id_base.astype(np.uint16)), None
    return None, p.id_base


def compute_similarity_matrix(pos_data: Union[FlatPileup, Sequence[Sequence[PosData]]],
                              num_cells: int,
                              max_fragment_length: int,
                              group_id_to_pos: Optional[Sequence[int]],
                              mutation_rate: float,
                              homozygous_rate: float,
                              seq_error_rate: float,
                              num_threads: int,
                              marker: str = "",
                              normalization: str = "ADD_MIN") -> np.ndarray:
    """num_cells x num_cells float64 similarity matrix (symmetric, zero diagonal).

    Same contract as the reference function: ``num_threads`` is the reference's thread count and
    matters only through the flush threshold 4*num_threads (similarity_matrix.cpp:354-356);
    ``marker`` is unused there (similarity_matrix.cpp:303) and here; an unknown ``normalization``
    raises InvalidNormalization (std::logic_error in the reference, :264).
    """
    del marker
    norm = to_enum(normalization)
    p = _as_flat(pos_data)
    g2p = np.arange(num_cells, dtype=np.uint32) if group_id_to_pos is None \
        else np.ascontiguousarray(group_id_to_pos, dtype=np.uint32)
    id16, id32 = _id_arrays(p)
    out = np.empty((num_cells, num_cells), dtype=np.float64)
    rc = _lib.lib().secedo_simmat_compute(
        _lib.ptr(p.chr_locus_off), p.n_chr, _lib.ptr(p.locus_pos), _lib.ptr(p.locus_entry_off),
        _lib.ptr(p.read_ids), _lib.ptr(id16), _lib.ptr(id32), _lib.ptr(g2p), len(g2p), num_cells,
        max_fragment_length, mutation_rate, homozygous_rate, seq_error_rate, num_threads, norm,
        _lib.ptr(out))
    _lib.check(rc)
    return out


def set_devices(device_ids: Optional[Sequence[int]] = None) -> None:
    """The devices compute_similarity_matrix (secedo_simmat_compute) spreads one matrix over: a list of HIP
    device ids -- a device may be listed more than once, every entry is a lane of its own (how the N-device path
    is rehearsed on fewer GPUs); None / [] returns to the environment (SECEDO_GPUS, else SECEDO_DEVICE, else 0).
    The matrix is bit-identical for any number of devices (secedo_simmat_set_devices)."""
    ids = [int(d) for d in (device_ids or [])]
    arr = (C.c_int * max(len(ids), 1))(*ids)
    _lib.check(_lib.lib().secedo_simmat_set_devices(arr if ids else None, len(ids)))


def get_devices():
    """The devices the next compute_similarity_matrix call uses (secedo_simmat_get_devices)."""
    arr = (C.c_int * 16)()
    n = _lib.lib().secedo_simmat_get_devices(arr, 16)
    if n < 0:
        _lib.check(n)
    return [int(arr[i]) for i in range(min(n, 16))]


def llr(x_s: int, x_d: int, mutation_rate: float, homozygous_rate: float,
        seq_error_rate: float) -> float:
    """D(x_s, x_d) = log P_diff - log P_same as the matrix path adds it (host-only): what the reference's nested
    sums return, uint64 wrap of its binomial products included, at any x_s + x_d (secedo_simmat_llr)."""
    return float(_lib.lib().secedo_simmat_llr(x_s, x_d, mutation_rate, homozygous_rate, seq_error_rate))


def llr_closed_form(x_s: int, x_d: int, mutation_rate: float, homozygous_rate: float,
                    seq_error_rate: float) -> float:
    """The reference's formula (similarity_matrix.cpp:117-170) in exact arithmetic, any x_s, x_d."""
    return float(_lib.lib().secedo_simmat_llr_closed_form(x_s, x_d, mutation_rate, homozygous_rate,
                                                          seq_error_rate))


class SimilarityMatrixPlan:
    """Staged interface: device-resident packed pileup, explicit stream, tile ranges.

    Device memory for the accumulator and the output matrix comes from torch (plumbing only);
    the kernels run on torch's current stream.
    """

    def __init__(self, device: int = 0):
        import torch
        self._torch = torch
        self.device = device
        self._h = C.c_void_p()
        _lib.check(_lib.lib().secedo_simmat_create(C.byref(self._h), device))
        self.num_cells = 0
        self._keep = None

    def close(self):
        if self._h:
            _lib.lib().secedo_simmat_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- prepare ---------------------------------------------------------------------------
    def set_packing(self, mode: str = "auto"):
        """Where prepare packs: "auto" (GPU, host when the pileup requires it), "host", "device"."""
        _lib.check(_lib.lib().secedo_simmat_set_packing(self._h, {"auto": 0, "host": 1, "device": 2}[mode]))
        return self

    @property
    def used_device_packing(self) -> bool:
        return bool(_lib.lib().secedo_simmat_used_device_packing(self._h))

    def upload(self, pos_data, group_id_to_pos=None, num_cells=None):
        """Raw flat pileup -> HBM (torch tensors); returns the resident pileup for prepare_resident."""
        p = _as_flat(pos_data)
        t = self._torch
        dev = "cuda:%d" % self.device
        if group_id_to_pos is None:
            n = num_cells if num_cells is not None else (int(p.id_base.max() >> 2) + 1 if p.n_entries else 1)
            group_id_to_pos = np.arange(n, dtype=np.uint32)
        g2p = np.ascontiguousarray(group_id_to_pos, dtype=np.uint32)
        id16, id32 = _id_arrays(p)

        def dev_tensor(a, view):
            a = np.ascontiguousarray(a)
            if a.size == 0:  # an empty pileup (e.g. a rank without chromosomes) still needs real pointers
                a = np.zeros(1, dtype=a.dtype)
            return t.from_numpy(a.view(view)).to(dev)

        res = dict(
            chr=dev_tensor(p.chr_locus_off, np.int32), pos=dev_tensor(p.locus_pos, np.int32),
            off=dev_tensor(p.locus_entry_off, np.int64), rid=dev_tensor(p.read_ids, np.int32),
            idb=dev_tensor(id16, np.int16) if id16 is not None else dev_tensor(id32, np.int32),
            idb_is16=id16 is not None, g2p=dev_tensor(g2p, np.int32), n_chr=p.n_chr, n_loci=p.n_loci,
            n_entries=p.n_entries, n_groups=len(g2p))
        return res

    def prepare_resident(self, res, num_cells, max_fragment_length, num_threads=8, block_cells=0):
        """prepare() from a pileup already resident in HBM (see upload): no host data touched unless
        the pileup needs the host packing path."""
        L = _lib.lib()
        idb = C.c_void_p(res["idb"].data_ptr())
        _lib.check(L.secedo_simmat_set_pileup_device(
            self._h, C.c_void_p(res["chr"].data_ptr()), res["n_chr"], C.c_void_p(res["pos"].data_ptr()),
            C.c_void_p(res["off"].data_ptr()), C.c_void_p(res["rid"].data_ptr()),
            idb if res["idb_is16"] else None, None if res["idb_is16"] else idb,
            C.c_void_p(res["g2p"].data_ptr()), res["n_groups"], res["n_loci"], res["n_entries"]))
        _lib.check(L.secedo_simmat_prepare(self._h, num_cells, max_fragment_length, num_threads, block_cells,
                                           self._stream()))
        self.num_cells = num_cells
        return self

    def prepare(self, pos_data, num_cells, max_fragment_length, group_id_to_pos=None,
                num_threads=8, block_cells=0):
        p = _as_flat(pos_data)
        g2p = np.arange(num_cells, dtype=np.uint32) if group_id_to_pos is None \
            else np.ascontiguousarray(group_id_to_pos, dtype=np.uint32)
        id16, id32 = _id_arrays(p)
        L = _lib.lib()
        _lib.check(L.secedo_simmat_set_pileup(
            self._h, _lib.ptr(p.chr_locus_off), p.n_chr, _lib.ptr(p.locus_pos),
            _lib.ptr(p.locus_entry_off), _lib.ptr(p.read_ids), _lib.ptr(id16), _lib.ptr(id32),
            _lib.ptr(g2p), len(g2p)))
        _lib.check(L.secedo_simmat_prepare(self._h, num_cells, max_fragment_length, num_threads,
                                           block_cells, self._stream()))
        self.num_cells = num_cells
        return self

    @property
    def num_tiles(self) -> int:
        return int(_lib.lib().secedo_simmat_num_tiles(self._h))

    @property
    def block_cells(self) -> int:
        return int(_lib.lib().secedo_simmat_block_cells(self._h))

    @property
    def acc_elems(self) -> int:
        return int(_lib.lib().secedo_simmat_acc_elems(self._h))

    @property
    def num_entries(self) -> int:
        return int(_lib.lib().secedo_simmat_num_entries(self._h))

    @property
    def num_reads(self) -> int:
        return int(_lib.lib().secedo_simmat_num_reads(self._h))

    @property
    def num_loci(self) -> int:
        return int(_lib.lib().secedo_simmat_num_loci(self._h))

    @property
    def pair_bound(self) -> int:
        """Upper bound on the (read pair, shared locus) incidences one cell pair can collect from the
        prepared pileup; decides the fixed-point scale of the accumulator (secedo_simmat.h)."""
        return int(_lib.lib().secedo_simmat_pair_bound(self._h))

    def set_pair_bound(self, pair_bound: int):
        """The bound of everything that will be summed into one accumulator (chromosome shards on several
        ranks: the sum of the shards' bounds), so that all of them quantise with the same scale."""
        _lib.check(_lib.lib().secedo_simmat_set_pair_bound(self._h, int(pair_bound)))
        return self

    @property
    def max_read_entries(self) -> int:
        """Kept entries of the longest read: no read pair shares more loci."""
        return int(_lib.lib().secedo_simmat_max_read_entries(self._h))

    def cell_squares(self):
        """Per matrix row the sum over loci of (kept entries of the row at the locus)^2, an int64 CUDA tensor
        of num_cells elements; pair_bound is its maximum. Shards that are added up sum these vectors."""
        t = self._torch
        out = t.empty(max(self.num_cells, 1), dtype=t.int64, device="cuda:%d" % self.device)
        _lib.check(_lib.lib().secedo_simmat_cell_squares(self._h, C.c_void_p(out.data_ptr()), self._stream()))
        return out[:self.num_cells]

    def set_scale_bounds(self, pair_bound: int, max_read_entries: int):
        """Pair bound and longest read of everything that will be summed into one accumulator (chromosome
        shards on several ranks), so that all of them quantise the same table with the same scale."""
        _lib.check(_lib.lib().secedo_simmat_set_scale_bounds(self._h, int(pair_bound), int(max_read_entries)))
        return self

    @property
    def scale_bounds_state(self) -> int:
        """0: no shared bounds set; 1: in force for the pileup held; 2: they were set and a later pileup (other
        sizes or other arrays) took them away -- accumulators of such a handle must not be added to other ranks'."""
        return int(_lib.lib().secedo_simmat_scale_bounds_state(self._h))

    @property
    def scale_log2(self) -> int:
        """log2 of the fixed-point scale of the last accumulate()."""
        return int(_lib.lib().secedo_simmat_scale_log2(self._h))

    # -- device work -----------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def new_acc(self, pad_tiles_to: int = 1):
        """int64 accumulator tensor; padded to a multiple of pad_tiles_to tiles (all-gather)."""
        b2 = self.block_cells ** 2
        tiles = -(-self.num_tiles // pad_tiles_to) * pad_tiles_to
        return self._torch.zeros(tiles * b2, dtype=self._torch.int64, device="cuda:%d" % self.device)

    def zero_acc(self, acc):
        _lib.check(_lib.lib().secedo_simmat_zero_acc(self._h, C.c_void_p(acc.data_ptr()), self._stream()))

    def accumulate(self, acc, mutation_rate, homozygous_rate, seq_error_rate, tile_begin=0,
                   tile_end=None, overwrite=False):
        """acc[tile] += the tile's sums for tiles [tile_begin, tile_end); overwrite=True stores them instead
        (secedo_simmat_assign): the tiles need not be zeroed first."""
        tile_end = self.num_tiles if tile_end is None else tile_end
        assert acc.dtype == self._torch.int64 and acc.is_contiguous() and acc.numel() >= self.acc_elems
        fn = _lib.lib().secedo_simmat_assign if overwrite else _lib.lib().secedo_simmat_accumulate
        _lib.check(fn(self._h, mutation_rate, homozygous_rate, seq_error_rate, tile_begin, tile_end,
                      C.c_void_p(acc.data_ptr()), self._stream()))

    def assign_finalize(self, acc, mutation_rate, homozygous_rate, seq_error_rate, normalization="ADD_MIN", out=None):
        """All tiles stored into `acc` (no zeroing needed) and normalised into `out`, one call: the single-GPU
        form of the reference function after prepare (secedo_simmat_assign_finalize)."""
        norm = to_enum(normalization)
        if out is None:
            out = self._torch.empty((self.num_cells, self.num_cells), dtype=self._torch.float64,
                                    device="cuda:%d" % self.device)
        assert acc.dtype == self._torch.int64 and acc.is_contiguous() and acc.numel() >= self.acc_elems
        assert out.dtype == self._torch.float64 and out.is_contiguous()
        _lib.check(_lib.lib().secedo_simmat_assign_finalize(
            self._h, mutation_rate, homozygous_rate, seq_error_rate, norm, C.c_void_p(acc.data_ptr()),
            C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def finalize(self, acc, normalization="ADD_MIN", out=None):
        norm = to_enum(normalization)
        if out is None:
            out = self._torch.empty((self.num_cells, self.num_cells), dtype=self._torch.float64,
                                    device="cuda:%d" % self.device)
        assert out.dtype == self._torch.float64 and out.is_contiguous()
        _lib.check(_lib.lib().secedo_simmat_finalize(
            self._h, norm, C.c_void_p(acc.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def finalize_rows(self, acc, row_begin, row_end, normalization="ADD_MIN", out=None):
        """Rows [row_begin, row_end) of the normalised matrix as a (rows x num_cells) tensor: the row
        block a rank keeps when the matrix stays sharded (BASELINE config 5)."""
        norm = to_enum(normalization)
        if out is None:
            out = self._torch.empty((row_end - row_begin, self.num_cells), dtype=self._torch.float64,
                                    device="cuda:%d" % self.device)
        assert out.dtype == self._torch.float64 and out.is_contiguous()
        if row_end == row_begin:
            return out
        _lib.check(_lib.lib().secedo_simmat_finalize_rows(
            self._h, norm, C.c_void_p(acc.data_ptr()), row_begin, row_end, C.c_void_p(out.data_ptr()),
            self._stream()))
        return out

    def tiles_of_rows(self, row_begin, row_end):
        """Global indices of the tiles that touch rows [row_begin, row_end) with their row block or their
        column block: what a rank accumulates itself to own that row block without any exchange."""
        n = C.c_uint32(0)
        _lib.check(_lib.lib().secedo_simmat_tiles_of_rows(self._h, row_begin, row_end, None, C.byref(n)))
        ids = np.empty(n.value, dtype=np.uint32)
        _lib.check(_lib.lib().secedo_simmat_tiles_of_rows(self._h, row_begin, row_end, _lib.ptr(ids), C.byref(n)))
        return ids

    def accumulate_list(self, acc, mutation_rate, homozygous_rate, seq_error_rate, tile_ids, overwrite=False):
        ids = np.ascontiguousarray(tile_ids, dtype=np.uint32)
        fn = _lib.lib().secedo_simmat_assign_list if overwrite else _lib.lib().secedo_simmat_accumulate_list
        _lib.check(fn(self._h, mutation_rate, homozygous_rate, seq_error_rate, _lib.ptr(ids), len(ids),
                      C.c_void_p(acc.data_ptr()), self._stream()))

    def max_of_tiles(self, acc, tile_ids) -> float:
        ids = np.ascontiguousarray(tile_ids, dtype=np.uint32)
        out = C.c_double(0.0)
        _lib.check(_lib.lib().secedo_simmat_max_of_tiles(
            self._h, C.c_void_p(acc.data_ptr()), _lib.ptr(ids), len(ids), C.byref(out), self._stream()))
        return out.value

    def finalize_rows_max(self, acc, row_begin, row_end, max_value, normalization="ADD_MIN", out=None):
        """finalize_rows with the maximum of D given by the caller (max-reduced over the ranks)."""
        norm = to_enum(normalization)
        if out is None:
            out = self._torch.empty((row_end - row_begin, self.num_cells), dtype=self._torch.float64,
                                    device="cuda:%d" % self.device)
        if row_end == row_begin:
            return out  # a rank without rows (more ranks than cell blocks)
        _lib.check(_lib.lib().secedo_simmat_finalize_rows_max(
            self._h, norm, C.c_void_p(acc.data_ptr()), row_begin, row_end, max_value, C.c_void_p(out.data_ptr()),
            self._stream()))
        return out

    def finalize_raw(self, acc, out=None):
        if out is None:
            out = self._torch.empty((self.num_cells, self.num_cells), dtype=self._torch.float64,
                                    device="cuda:%d" % self.device)
        _lib.check(_lib.lib().secedo_simmat_finalize_raw(
            self._h, C.c_void_p(acc.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def last_counts(self):
        u, r = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().secedo_simmat_last_counts(self._h, C.byref(u), C.byref(r)))
        return int(u.value), int(r.value)

    def last_pair_kernel_ms(self):
        """Device time of accumulate_counts in the last accumulate (None when another kernel variant ran)."""
        ms = C.c_float(0.0)
        if _lib.lib().secedo_simmat_last_pair_kernel_ms(self._h, C.byref(ms)) != 0:
            return None
        return float(ms.value)

    @property
    def pair_kernel(self) -> str:
        """The pair kernel the prepared pileup runs (secedo_simmat_pair_kernel)."""
        return _lib.lib().secedo_simmat_pair_kernel(self._h).decode()

    def last_accumulate_ms(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.lib().secedo_simmat_last_accumulate_ms(self._h, C.byref(ms)))
        return float(ms.value)
