"""Row-band sharding of a picture's 64x64 blocks across ranks, and the layout of the compact per-rank ME result
buffer that is all-gathered (RCCL on GPUs, gloo in the CPU tests).

Open-loop ME has no dependency between 64x64 blocks (reference: Docs/Appendix-Open-Loop-Motion-Estimation.md;
SURVEY §8e), so every rank processes a contiguous band of b64 rows of the same picture against fully replicated
reference pyramids, and the only exchange is the all-gather of per-b64 results.
"""
import numpy as np

from . import abi

# what the host's entropy coder / MD consumes: the MeSbResults arrays and the per-b64 scalars (~1 KB per b64).  The
# search-level arrays (p_sb_best_sad / mv, 5.4 KB per b64, mostly padding for unused reference slots) stay on the rank.
GATHER_FIELDS = tuple(n for n, _, _ in abi.RESULT_FIELDS if n not in ("hme_sc", "hme_sad", "do_ref", "sb_best_sad", "sb_best_mv"))


def band(h64, rank, world):
    """[row0, row1) of b64 rows owned by `rank`."""
    return (h64 * rank) // world, (h64 * (rank + 1)) // world


def rows_max(h64, world):
    return max(band(h64, r, world)[1] - band(h64, r, world)[0] for r in range(world))


class BandLayout:
    """Byte layout of one rank's compact result buffer for `n_pictures` pictures: [picture][field][rows_max * w64][elems]."""

    def __init__(self, w64, h64, world, n_pu, max_refs, max_cand, n_pictures=1):
        self.w64, self.h64, self.world, self.n_pictures = w64, h64, world, n_pictures
        self.fields = [(n, np.dtype(dt), c(n_pu, max_refs, max_cand)) for n, dt, c in abi.RESULT_FIELDS if n in GATHER_FIELDS]
        self.nbb = rows_max(h64, world) * w64
        self.bytes_per_b64 = sum(dt.itemsize * c for _, dt, c in self.fields)
        self.nbytes = n_pictures * self.nbb * self.bytes_per_b64

    def field_offsets(self, picture):
        """name -> (byte offset of the field's first band row, bytes per b64)."""
        off = picture * self.nbb * self.bytes_per_b64
        out = {}
        for n, dt, c in self.fields:
            out[n] = (off, dt.itemsize * c)
            off += self.nbb * dt.itemsize * c
        return out

    def results_struct(self, base_ptr, picture, rank):
        """abi.MeResults whose pointers are biased so that absolute b64 indices of `rank`'s band land in its compact buffer."""
        first = band(self.h64, rank, self.world)[0] * self.w64
        res = abi.MeResults()
        for n, (off, per) in self.field_offsets(picture).items():
            setattr(res, n, base_ptr + off - first * per)
        return res

    def unpack(self, gathered, picture):
        """gathered: uint8 array [world, nbytes] (all ranks' compact buffers).  Returns name -> full-picture array [n_b64, elems]."""
        gathered = np.asarray(gathered).reshape(self.world, self.nbytes)
        out = {}
        for n, dt, c in self.fields:
            off, per = self.field_offsets(picture)[n]
            full = np.zeros((self.w64 * self.h64, c), dt)
            for r in range(self.world):
                r0, r1 = band(self.h64, r, self.world)
                nb = (r1 - r0) * self.w64
                full[r0 * self.w64:r1 * self.w64] = gathered[r, off:off + nb * per].view(dt).reshape(nb, c)
            out[n] = full
        return out
